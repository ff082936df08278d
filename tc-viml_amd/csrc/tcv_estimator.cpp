// Native mirror of the estimator's per-frame window management around the hot path (include/tcv_estimator.h).
// Reference: vins_estimator/src/estimator.cpp (processIMU :191-228, processImagewithLine :230-383, solveOdometry :1476-1490,
// double2vector :1537-1627, failureDetection :1629-1675, slideWindowWithLinesFoV :2121-2259, prior chaining :2027-2044 / :2083-2113)
// and feature_manager.cpp (addFeaturesCheckParallax :260-334, setDepth :379-397, removeFailures :399-408, triangulate :440-492,
// removeLineOutlier :494-534, removeBackShiftDepth :559-616, removeFront :655-696, compensatedParallax2 :698-734).
// Host code only: every numerical step of the hot path goes through the C-ABI of tcv.h (HIP kernels); there is no CPU solver here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tcv_estimator.h"
namespace tcv { hipStream_t util_stream(); hipStream_t aux_stream(); }      // (tcv_capi.hip: the calling thread's utility stream and its second stream)
#include <functional>
#include "tcv_packed.h"      // parallel_run, HostOp: the persistent host worker threads of the packer

namespace tcv { void set_error(const std::string &s); }
int tcv_marg_layout_n(const tcv_batch *b, int window);      // (tcv_marg.hip: n of the prior a window's attached marginalisation problem makes, known before the kernel runs)

namespace {

constexpr int W = 10;                      // WINDOW_SIZE (parameters.h:20)
enum { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
typedef std::array<double, 3> V3;
typedef std::array<double, 9> M3;          // row-major

inline V3 add(const V3 &a, const V3 &b) { return {a[0] + b[0], a[1] + b[1], a[2] + b[2]}; }
inline V3 sub(const V3 &a, const V3 &b) { return {a[0] - b[0], a[1] - b[1], a[2] - b[2]}; }
inline V3 scl(const V3 &a, double s) { return {a[0] * s, a[1] * s, a[2] * s}; }
inline double nrm(const V3 &a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
inline V3 mv(const M3 &m, const V3 &v) { return {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]}; }
inline M3 mm(const M3 &a, const M3 &b) {
    M3 c;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
    return c;
}
inline M3 tr(const M3 &a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
inline M3 eye() { return {1, 0, 0, 0, 1, 0, 0, 0, 1}; }
// Eigen toRotationMatrix of q = (x y z w), no normalisation
inline M3 q2R(const double q[4]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    return {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
}
// Eigen `Quaterniond q{R}` (vector2double, estimator.cpp:1499), x y z w
inline void R2q(const M3 &m, double q[4]) {
    auto M = [&](int r, int c) { return m[3 * r + c]; };
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (M(2, 1) - M(1, 2)) * t; q[1] = (M(0, 2) - M(2, 0)) * t; q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0); q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (M(k, j) - M(j, k)) * t; q[j] = (M(j, i) + M(i, j)) * t; q[k] = (M(k, i) + M(i, k)) * t;
    }
}
// Utility::deltaQ(theta).toRotationMatrix() (utility.h:15-28; not normalised, like the reference)
inline M3 deltaQ_R(const V3 &th) { const double q[4] = {th[0] / 2, th[1] / 2, th[2] / 2, 1.0}; return q2R(q); }

// smallest right singular vector of A (rows x 4): one-sided Jacobi on the columns (what JacobiSVD in triangulate() delivers)
void smallest_right_singular_vector(std::vector<std::array<double, 4>> A, double v_out[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    const int m = (int)A.size();
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double a = 0, b = 0, c = 0;
                for (int i = 0; i < m; i++) { a += A[i][p] * A[i][p]; b += A[i][q] * A[i][q]; c += A[i][p] * A[i][q]; }
                if (std::fabs(c) <= 1e-300 || std::fabs(c) <= 2.3e-16 * std::sqrt(a * b)) continue;
                rotated = true;
                const double zeta = (b - a) / (2.0 * c);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
                for (int i = 0; i < m; i++) { const double x = A[i][p], y = A[i][q]; A[i][p] = cs * x - sn * y; A[i][q] = sn * x + cs * y; }
                for (int i = 0; i < 4; i++) { const double x = V[i][p], y = V[i][q]; V[i][p] = cs * x - sn * y; V[i][q] = sn * x + cs * y; }
            }
        if (!rotated) break;
    }
    int best = 0;
    double bn = -1;
    for (int j = 0; j < 4; j++) {
        double s = 0;
        for (int i = 0; i < m; i++) s += A[i][j] * A[i][j];
        if (bn < 0 || s < bn) { bn = s; best = j; }
    }
    for (int i = 0; i < 4; i++) v_out[i] = V[i][best];
}

struct Feature {
    int id, start;
    std::vector<V3> obs;
    double depth = -1.0;       // FeaturePerId ctor: estimated_depth(-1.0)
    int solve_flag = 0;
    int end() const { return start + (int)obs.size() - 1; }
};
struct LineObs {
    double vec[4], abc[3], world[6];
    double errA = -1, errD = -1, overlap = -1;
    bool credible_line = true, use_flag = false;
};
struct LineFeature { int id, start; std::vector<LineObs> obs; bool credible_matching = true; };
struct GivenLine { double d[9]; };          // 3D start, 3D end (world), A B C
struct ImuBuf {
    bool valid = false;
    V3 acc0, gyr0, ba, bg;
    std::vector<V3> acc, gyr;
};

}  // namespace

// A lock-step frame whose marginalisation was left running when tcv_estimators_optimize returned (device-resident state): its batch,
// shared by the estimators of the frame; each asks for the status of its own window at its next frame (or lets go of it when
// it is reset / destroyed), the last one to let go destroys the batch.
// kernel-time accounting of the lock-step frames (tcv_estimators_kernel_profile): HIP-event durations of the solve and marginalisation
// launches, the windows they held and those windows' ALGORITHMIC bytes x linearisations -- what bench.py --mode replay
// prices against the HBM roofline
std::mutex g_kmu;
double g_kern[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // solve ms, solve launches, marginalisation ms, marginalisation launches, windows, bytes x linearisations, linearisations, marginalised windows
void kern_add(int slot, double v) { std::lock_guard<std::mutex> g(g_kmu); g_kern[slot] += v; }

struct EstInflight {
    tcv_batch *b = nullptr;
    int n_marg = 0;
    // deferred launch (the default): the frame that solved the batch returns WITHOUT launching its marginalisation; the first of its
    // estimators to come back for its next frame launches it -- behind that frame's 2D-3D association, whose device round trip would
    // otherwise queue behind the marginalisation kernel on the thread's stream (0.3 - 0.5 ms of a lock-step frame, whatever the number
    // of windows) -- and every estimator takes its own prior handle.  The kernel then runs while the host builds that frame's windows.
    bool deferred = false, launched = false;
    int launch_rc = TCV_OK;
    std::string launch_msg;
    std::vector<tcv_prior *> handles;      // per window of the batch: the new prior, until its estimator takes it
    int launch(void *stream) {      // (any estimator of the batch, any thread: once)
        std::lock_guard<std::mutex> g(mu);
        if (launched) return launch_rc;
        launched = true;
        const int n = tcv_batch_size(b);
        handles.assign(n, nullptr);
        // (test hook: TCV_EST_INJECT_LAUNCH_FAIL=q makes the q-th deferred launch of the process fail -- the library has no natural one to offer)
        static std::atomic<int> launches{0};
        static const int inject = getenv("TCV_EST_INJECT_LAUNCH_FAIL") ? atoi(getenv("TCV_EST_INJECT_LAUNCH_FAIL")) : -1;
        if (inject >= 0 && launches.fetch_add(1) == inject) { tcv::set_error("injected launch failure"); launch_rc = TCV_ERR_HIP; }
        else launch_rc = tcv_batch_marginalize(b, stream);
        if (launch_rc == TCV_OK) launch_rc = tcv_batch_get_priors_device_async(b, handles.data(), n);
        if (launch_rc != TCV_OK) launch_msg = tcv_last_error();
        return launch_rc;
    }
    tcv_prior *take(int k) { std::lock_guard<std::mutex> g(mu); tcv_prior *p = (k >= 0 && k < (int)handles.size()) ? handles[k] : nullptr; if (p) handles[k] = nullptr; return p; }
    std::vector<int> status;
    bool have_status = false;
    std::mutex mu;
    int status_of(int k) {
        std::lock_guard<std::mutex> g(mu);
        if (!have_status) {
            const int n = tcv_batch_size(b);
            status.assign(n, -9);
            if (tcv_batch_marg_status(b, status.data(), n) != TCV_OK) status.assign(n, -9);      // (waits for the batch; -9: the question itself failed)
            have_status = true;
        }
        // (test hook: TCV_EST_INJECT_MARG_FAIL=q reports the q-th status asked for in the process as a sweep-cap failure)
        static std::atomic<int> asked{0};
        static const int inject = getenv("TCV_EST_INJECT_MARG_FAIL") ? atoi(getenv("TCV_EST_INJECT_MARG_FAIL")) : -1;
        if (inject >= 0 && asked.fetch_add(1) == inject) return 1;
        return (k >= 0 && k < (int)status.size()) ? status[k] : -9;
    }
    ~EstInflight() {
        for (tcv_prior *p : handles) if (p) tcv_prior_destroy(p);
        if (!b) return;
        double ms = 0;
        if (tcv_batch_synchronize(b) == TCV_OK && tcv_batch_stats(b, nullptr, nullptr, &ms) == TCV_OK && ms > 0) { kern_add(2, ms); kern_add(3, 1); kern_add(7, n_marg); }
        tcv_batch_destroy(b);      // (waits for work in flight)
    }
};

// snapshot of the window an estimator handed to the solver (tcv_estimator_set_window_tap)
struct WindowTap {
    bool on = false, have = false;
    std::vector<double> pose_in, sb_in, ex_in, feat_in, pose_out, sb_out, ex_out, feat_out, pts, ld, x0, J0, r0;
    std::vector<tcv_imu_preintegration> imu;
    std::vector<int> imu_i, imu_j, pi, pj, pl, lf, pk, pidx, psize, pcol;
    double Ric[9];
    int marg_flag = 0, prior_m = 0, prior_n = 0, iterations = 0, applied = 0;
    double final_cost = 0;
};

struct tcv_estimator {
    tcv_estimator_config cfg;
    WindowTap tap;
    std::shared_ptr<EstInflight> prev;       // the frame whose marginalisation produced `prior` and may still be running; prev_k: this estimator's window in it
    int prev_k = -1;
    std::shared_ptr<EstInflight> pend;       // the frame whose marginalisation has not been launched yet (EstInflight::deferred): its prior is taken at the start of the next tcv_estimators_optimize
    int pend_k = -1, pend_flag = MARGIN_OLD;
    V3 Ps[W + 1], Vs[W + 1], Bas[W + 1], Bgs[W + 1];
    M3 Rs[W + 1];
    V3 tic;
    M3 ric;
    ImuBuf bufs[W + 1];
    tcv_imu_preintegration pre[W + 1];       // host copy (TCV_EST_HOST_PREINT=1), otherwise only sum_dt is filled in
    std::shared_ptr<tcv_preint> preh[W + 1]; // pre_integrations[] on the device (estimator.h: pre_integrations[WINDOW_SIZE + 1]); slots share a handle while a slide copies them
    bool pre_valid[W + 1];
    std::vector<Feature> features;
    std::vector<GivenLine> line_obs[W + 1];
    bool assoc = false;
    std::vector<double> map_lines;           // n x 6
    std::shared_ptr<tcv_line_map> map_dev;   // the same lines resident on the device (created with set_line_map; null: uploaded per call)
    int n_map = 0;
    M3 Rbw;
    V3 Tbw;
    std::vector<LineFeature> linefeatures;
    std::vector<unsigned char> fov[W + 1];   // WorldLinesInFOV[i] as a mask over the map (empty: not set)
    bool fov_ready = false;
    tcv_prior *prior = nullptr;
    std::vector<std::pair<int, int>> prior_blocks;      // (kind 0 pose / 1 sb / 2 ex, index)
    int frame_count = 0, marg_flag = MARGIN_OLD;
    bool have_acc0 = false, have_last = false;
    V3 acc_0, gyr_0, last_P;
    tcv_estimator_stats stats;
    // the window handed to the solver (parameter blocks are identified by address: these arrays live as long as the estimator)
    double para_pose[(W + 1) * 7], para_sb[(W + 1) * 9], para_ex[7];
    std::vector<double> para_feature;
    std::vector<int> sel;                    // indices into `features` of the landmarks of the current window
    // factor lists of the current window
    std::vector<tcv_imu_preintegration> w_imu;
    std::vector<const tcv_preint *> w_imu_dev, m_imu_dev;      // the same factors as device-resident handles (empty: host pre-integrations)
    std::vector<int> w_imu_i, w_imu_j, w_pi, w_pj, w_pl, w_lf;
    std::vector<double> w_pts, w_ld;
    double w_Ric[9];
    std::vector<int> w_pk, w_pidx;
    // marginalisation sub-problem
    std::vector<tcv_imu_preintegration> m_imu;
    std::vector<int> m_imu_i, m_imu_j, m_pi, m_pj, m_pl;
    std::vector<double> m_pts;
    std::vector<double *> m_drop;
    int n_line_obs_total = 0;
    int phase = 0;                           // 0: between frames, 1: window full, waiting for the optimisation, 2: optimised, waiting for finish_frame
    int opt_failed = 0;                      // != TCV_OK: this estimator's window failed in the last lock-step batch (reported by finish_frame)
    std::string opt_msg;
    int pend_failed = 0;                     // != TCV_OK: the previous frame's (deferred) marginalisation could not be launched: this frame's window is solved without
    std::string pend_msg;                    // its prior, nothing of it is applied and finish_frame reports the failure (the caller resets, like after failureDetection)
};
// the device-resident line map is bound to the device that was current at tcv_estimator_set_line_map: an estimator optimised with another
// current device uses the host copy it still holds (uploaded with the call) instead of failing the association (round-5 advisor finding)
extern "C" int tcv_line_map_device(const tcv_line_map *m);
static inline const tcv_line_map *map_on_current_device(const tcv_estimator *e) {
    if (!e->map_dev) return nullptr;
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    return tcv_line_map_device(e->map_dev.get()) == cur ? e->map_dev.get() : nullptr;
}

namespace {

void process_imu(tcv_estimator *e, int n, const double *acc, const double *gyr) {
    const int j = e->frame_count;
    auto row = [](const double *a, int i) { return V3{a[3 * i], a[3 * i + 1], a[3 * i + 2]}; };
    if (!e->have_acc0) { e->acc_0 = row(acc, 0); e->gyr_0 = row(gyr, 0); e->have_acc0 = true; }
    ImuBuf &b = e->bufs[j];
    if (!b.valid) { b.valid = true; b.acc0 = e->acc_0; b.gyr0 = e->gyr_0; b.ba = e->Bas[j]; b.bg = e->Bgs[j]; b.acc.clear(); b.gyr.clear(); }
    const double dt = e->cfg.imu_dt;
    const V3 G = {e->cfg.gravity[0], e->cfg.gravity[1], e->cfg.gravity[2]};
    for (int i = 1; i <= n; i++) {
        const V3 a = row(acc, i), w = row(gyr, i);
        if (j != 0) {
            b.acc.push_back(a); b.gyr.push_back(w);
            const V3 un_acc_0 = sub(mv(e->Rs[j], sub(e->acc_0, e->Bas[j])), G);            // estimator.cpp:217-224
            const V3 un_gyr = sub(scl(add(e->gyr_0, w), 0.5), e->Bgs[j]);
            e->Rs[j] = mm(e->Rs[j], deltaQ_R(scl(un_gyr, dt)));
            const V3 un_acc_1 = sub(mv(e->Rs[j], sub(a, e->Bas[j])), G);
            const V3 un_acc = scl(add(un_acc_0, un_acc_1), 0.5);
            e->Ps[j] = add(add(e->Ps[j], scl(e->Vs[j], dt)), scl(un_acc, 0.5 * dt * dt));
            e->Vs[j] = add(e->Vs[j], scl(un_acc, dt));
        }
        e->acc_0 = a; e->gyr_0 = w;
    }
    e->pre_valid[j] = false;
}

bool add_features_check_parallax(tcv_estimator *e, int n_points, const int *ids, const double *pts, int n_lines, const int *line_ids, const double *lines) {
    const int fc = e->frame_count;
    std::unordered_map<int, int> by_id;
    for (size_t k = 0; k < e->features.size(); k++) by_id[e->features[k].id] = (int)k;
    int last_track_num = 0;
    for (int k = 0; k < n_points; k++) {
        auto it = by_id.find(ids[k]);
        int idx;
        if (it == by_id.end()) {
            Feature f; f.id = ids[k]; f.start = fc;
            e->features.push_back(f); idx = (int)e->features.size() - 1; by_id[ids[k]] = idx;
        } else { idx = it->second; last_track_num++; }
        e->features[idx].obs.push_back(V3{pts[3 * k], pts[3 * k + 1], pts[3 * k + 2]});
    }
    if (e->assoc) {      // the line tracker's (id, end points): addFeaturesCheckParallax :291-311
        std::unordered_map<int, int> by_lid;
        for (size_t k = 0; k < e->linefeatures.size(); k++) by_lid[e->linefeatures[k].id] = (int)k;
        for (int k = 0; k < n_lines; k++) {
            auto it = by_lid.find(line_ids[k]);
            int idx;
            if (it == by_lid.end()) {
                LineFeature lf; lf.id = line_ids[k]; lf.start = fc;
                e->linefeatures.push_back(lf); idx = (int)e->linefeatures.size() - 1; by_lid[line_ids[k]] = idx;
            } else idx = it->second;
            LineObs ob;
            const double *v = lines + 4 * k;
            for (int i = 0; i < 4; i++) ob.vec[i] = v[i];
            ob.abc[0] = v[3] - v[1]; ob.abc[1] = v[0] - v[2]; ob.abc[2] = v[2] * v[1] - v[0] * v[3];      // feature_manager.cpp:11-13
            for (int i = 0; i < 6; i++) ob.world[i] = 0.0;
            e->linefeatures[idx].obs.push_back(ob);
        }
    } else {
        e->line_obs[fc].clear();
        for (int k = 0; k < n_lines; k++) { GivenLine g; std::memcpy(g.d, lines + 9 * k, sizeof g.d); e->line_obs[fc].push_back(g); }
    }
    if (fc < 2 || last_track_num < 20) return true;
    double s = 0;
    int n = 0;
    for (auto &f : e->features)
        if (f.start <= fc - 2 && f.end() >= fc - 1) {
            const V3 &pi = f.obs[fc - 2 - f.start], &pj = f.obs[fc - 1 - f.start];      // compensatedParallax2 (:698-734)
            const double du = pi[0] / pi[2] - pj[0], dv = pi[1] / pi[2] - pj[1];
            s += std::sqrt(du * du + dv * dv); n++;
        }
    return n == 0 ? true : (s / n >= e->cfg.min_parallax);
}

bool selected(const Feature &f) { return f.obs.size() >= 2 && f.start < W - 2; }

void triangulate(tcv_estimator *e) {
    for (auto &f : e->features) {
        if (!selected(f) || f.depth > 0) continue;
        const int i = f.start;
        const V3 t0 = add(e->Ps[i], mv(e->Rs[i], e->tic));
        const M3 R0 = mm(e->Rs[i], e->ric);
        std::vector<std::array<double, 4>> A;
        for (size_t k = 0; k < f.obs.size(); k++) {
            const int j = i + (int)k;
            const V3 t1 = add(e->Ps[j], mv(e->Rs[j], e->tic));
            const M3 R1 = mm(e->Rs[j], e->ric);
            const V3 t = mv(tr(R0), sub(t1, t0));
            const M3 R = mm(tr(R0), R1), Rt = tr(R);
            const V3 mt = scl(mv(Rt, t), -1.0);
            double P[3][4];
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) P[r][c] = Rt[3 * r + c]; P[r][3] = mt[r]; }
            const V3 fn = scl(f.obs[k], 1.0 / nrm(f.obs[k]));
            std::array<double, 4> r0, r1;
            for (int c = 0; c < 4; c++) { r0[c] = fn[0] * P[2][c] - fn[2] * P[0][c]; r1[c] = fn[1] * P[2][c] - fn[2] * P[1][c]; }
            A.push_back(r0); A.push_back(r1);
        }
        double v[4];
        smallest_right_singular_vector(A, v);
        f.depth = v[2] / v[3];
        if (f.depth < 0.1) f.depth = e->cfg.init_depth;
    }
}

void poses_of(const tcv_estimator *e, double pose[(W + 1) * 7], double ex[7]) {
    for (int i = 0; i <= W; i++) { for (int c = 0; c < 3; c++) pose[7 * i + c] = e->Ps[i][c]; R2q(e->Rs[i], pose + 7 * i + 3); }
    for (int c = 0; c < 3; c++) ex[c] = e->tic[c];
    R2q(e->ric, ex + 3);
}

// UpdateLinesInFoV / initialLineFoVWindow, updateLinePairInWindow and removeLineOutlier, as processImagewithLine runs them before
// solveOdometry (estimator.cpp:328-336, :385-497; feature_manager.cpp:494-534)
// The 2D-3D association of one estimator, in two halves around the device call so that the estimators of a lock-step frame share ONE
// tcv_match_lines_batch (one upload, one download, one wait for all of them instead of a round trip each).
struct AssocJob {
    double pose[(W + 1) * 7], ex[7];
    std::vector<int> det_frame, match;
    std::vector<double> det;
    std::vector<LineObs *> where;
    std::vector<unsigned char> given;
    std::vector<float> err;
    bool one_call = false, call = false;
    tcv_match_lines_args args;
};
int assoc_prepare(tcv_estimator *e, AssocJob &J) {
    poses_of(e, J.pose, J.ex);
    const int nm = e->n_map;
    for (auto &lf : e->linefeatures)
        for (size_t k = 0; k < lf.obs.size(); k++) { J.det_frame.push_back(lf.start + (int)k); J.det.insert(J.det.end(), lf.obs[k].vec, lf.obs[k].vec + 4); J.where.push_back(&lf.obs[k]); }
    J.one_call = e->fov_ready;      // steady state: UpdateLinesInFoV(frame_count) and the matching in ONE device call
    if (!e->fov_ready) {
        std::vector<unsigned char> fov_now((size_t)(W + 1) * nm, 0);
        const int rc = tcv_match_lines(W + 1, J.pose, J.ex, e->Rbw.data(), e->Tbw.data(), e->cfg.K, e->cfg.width, e->cfg.height, W, nm, e->map_lines.data(), 0, nullptr, nullptr,
                                       e->cfg.angle_th, e->cfg.overlap_th, 0, fov_now.data(), nullptr, nullptr, nullptr);
        if (rc != TCV_OK) return rc;
        for (int i = 0; i <= W; i++) e->fov[i].assign(fov_now.begin() + (size_t)i * nm, fov_now.begin() + (size_t)(i + 1) * nm);      // initialLineFoVWindow (:483-497)
        e->fov_ready = true;
    }
    J.given.assign((size_t)(W + 1) * nm, 0);
    for (int i = 0; i <= W; i++) if (!e->fov[i].empty()) std::copy(e->fov[i].begin(), e->fov[i].end(), J.given.begin() + (size_t)i * nm);
    const int nd = (int)J.where.size();
    J.match.assign(std::max(nd, 1), 0);
    J.err.assign((size_t)std::max(nd, 1) * 3, 0.0f);
    J.call = nd > 0 || J.one_call;
    tcv_match_lines_args &a = J.args;
    a.n_frames = W + 1; a.poses = J.pose; a.ex_pose = J.ex; a.Rbw = e->Rbw.data(); a.Tbw = e->Tbw.data(); a.K = e->cfg.K; a.width = e->cfg.width; a.height = e->cfg.height;
    a.window_size = W; a.n_map = nm; a.lines3d = e->map_lines.data(); a.map_device = map_on_current_device(e); a.n_det = nd; a.det_frame = nd ? J.det_frame.data() : nullptr; a.det_lines = nd ? J.det.data() : nullptr;
    a.angle_th = e->cfg.angle_th; a.overlap_th = e->cfg.overlap_th; a.fov_given = J.one_call ? 2 + W : 1; a.in_fov = J.given.data();
    a.match_index = nd ? J.match.data() : nullptr; a.err = nd ? J.err.data() : nullptr; a.projected = nullptr;
    return TCV_OK;
}
void assoc_finish(tcv_estimator *e, AssocJob &J) {
    const int nm = e->n_map, nd = (int)J.where.size();
    if (J.call && J.one_call) e->fov[W].assign(J.given.begin() + (size_t)W * nm, J.given.begin() + (size_t)(W + 1) * nm);                          // UpdateLinesInFoV(frame_count)
    if (nd > 0) {
        for (int q = 0; q < nd; q++) {
            LineObs &ob = *J.where[q];
            ob.errA = (double)J.err[3 * q]; ob.errD = (double)J.err[3 * q + 1]; ob.overlap = (double)J.err[3 * q + 2];
            const unsigned char *g = J.given.data() + (size_t)J.det_frame[q] * nm;
            int first = -1;
            for (int i = 0; i < nm && first < 0; i++) if (g[i]) first = i;
            if (J.match[q] >= 0) std::memcpy(ob.world, e->map_lines.data() + 6 * J.match[q], sizeof ob.world);
            else if (first >= 0) std::memcpy(ob.world, e->map_lines.data() + 6 * first, sizeof ob.world);          // `linesInThisFov[0]` (:871-877)
            else { const double fake[6] = {ob.vec[0], ob.vec[1], 1.0, ob.vec[0], ob.vec[1], 1.0}; std::memcpy(ob.world, fake, sizeof fake); }      // fake_line (:707-709)
            ob.use_flag = true;
            ob.credible_line = J.err[3 * q] != -1.0f;
        }
    }
    for (auto &lf : e->linefeatures) {                    // removeLineOutlier
        if (lf.obs.empty()) continue;
        const double *first = lf.obs[0].world;
        int count = 0;
        for (auto &ob : lf.obs) {
            double d2 = 0;
            for (int c = 0; c < 3; c++) { const double d = (ob.world[3 + c] - ob.world[c]) - (first[3 + c] - first[c]); d2 += d * d; }
            ob.credible_line = !((float)std::sqrt(d2) > 0.1f);
            count += ob.credible_line ? 0 : 1;
        }
        lf.credible_matching = !((count / (int)lf.obs.size()) >= 0.5);
    }
}

// vector2double (estimator.cpp:1492-1535) + the factor lists OptimizationWithLine walks (:1683-1846)
int build_window(tcv_estimator *e) {
    poses_of(e, e->para_pose, e->para_ex);
    for (int i = 0; i <= W; i++) for (int c = 0; c < 3; c++) { e->para_sb[9 * i + c] = e->Vs[i][c]; e->para_sb[9 * i + 3 + c] = e->Bas[i][c]; e->para_sb[9 * i + 6 + c] = e->Bgs[i][c]; }
    e->sel.clear();
    for (size_t k = 0; k < e->features.size(); k++) if (selected(e->features[k])) e->sel.push_back((int)k);
    e->para_feature.resize(std::max<size_t>(1, e->sel.size()));
    for (size_t l = 0; l < e->sel.size(); l++) e->para_feature[l] = 1.0 / e->features[e->sel[l]].depth;
    e->w_imu.clear(); e->w_imu_i.clear(); e->w_imu_j.clear(); e->w_imu_dev.clear();
    for (int k = 0; k < W; k++)
        if (e->pre[k + 1].sum_dt <= 10.0) {      // estimator.cpp:1726
            e->w_imu.push_back(e->pre[k + 1]); e->w_imu_i.push_back(k); e->w_imu_j.push_back(k + 1);
            e->w_imu_dev.push_back(e->preh[k + 1].get());
        }
    e->w_pi.clear(); e->w_pj.clear(); e->w_pl.clear(); e->w_pts.clear();
    for (size_t l = 0; l < e->sel.size(); l++) {      // estimator.cpp:1737-1771
        const Feature &f = e->features[e->sel[l]];
        for (size_t k = 1; k < f.obs.size(); k++) {
            e->w_pi.push_back(f.start); e->w_pj.push_back(f.start + (int)k); e->w_pl.push_back((int)l);
            e->w_pts.insert(e->w_pts.end(), f.obs[0].begin(), f.obs[0].end()); e->w_pts.insert(e->w_pts.end(), f.obs[k].begin(), f.obs[k].end());
        }
    }
    e->w_lf.clear(); e->w_ld.clear();
    e->n_line_obs_total = 0;
    if (!e->assoc) {
        for (int i = 0; i <= W; i++) for (auto &g : e->line_obs[i]) { e->w_lf.push_back(i); e->w_ld.insert(e->w_ld.end(), g.d, g.d + 9); }
    } else {      // estimator.cpp:1786-1846
        for (auto &lf : e->linefeatures) {
            e->n_line_obs_total += (int)lf.obs.size();
            if (!(lf.obs.size() >= 2 && lf.start < W - 2) || !lf.credible_matching) continue;
            for (size_t k = 0; k < lf.obs.size(); k++) {
                const LineObs &ob = lf.obs[k];
                if (!ob.credible_line || !ob.use_flag || ob.errD > e->cfg.dist_th) continue;
                e->w_lf.push_back(lf.start + (int)k);
                const V3 ps = add(mv(e->Rbw, V3{ob.world[0], ob.world[1], ob.world[2]}), e->Tbw), pe = add(mv(e->Rbw, V3{ob.world[3], ob.world[4], ob.world[5]}), e->Tbw);
                e->w_ld.insert(e->w_ld.end(), ps.begin(), ps.end()); e->w_ld.insert(e->w_ld.end(), pe.begin(), pe.end());
                e->w_ld.insert(e->w_ld.end(), ob.abc, ob.abc + 3);
            }
        }
    }
    double qn[4];
    const double n4 = std::sqrt(e->para_ex[3] * e->para_ex[3] + e->para_ex[4] * e->para_ex[4] + e->para_ex[5] * e->para_ex[5] + e->para_ex[6] * e->para_ex[6]);
    for (int c = 0; c < 4; c++) qn[c] = e->para_ex[3 + c] / n4;
    const M3 Ric = q2R(qn);
    std::memcpy(e->w_Ric, Ric.data(), sizeof e->w_Ric);
    e->w_pk.clear(); e->w_pidx.clear();
    for (auto &b : e->prior_blocks) { e->w_pk.push_back(b.first); e->w_pidx.push_back(b.second); }
    return TCV_OK;
}

void fill_desc(const tcv_estimator *e, tcv_window_desc &d, bool marg, int flag) {
    std::memset(&d, 0, sizeof d);
    tcv_estimator *m = const_cast<tcv_estimator *>(e);
    d.n_frames = W + 1; d.n_landmarks = (int)e->sel.size(); d.estimate_extrinsic = e->cfg.estimate_extrinsic;
    d.para_pose = m->para_pose; d.para_speedbias = m->para_sb; d.para_ex_pose = m->para_ex; d.para_feature = m->para_feature.data();
    d.proj_sqrt_info = e->cfg.focal_length / 1.5; d.proj_loss_a = 1.0; d.line_loss_a = 1.0;
    std::memcpy(d.line_K, e->cfg.K, sizeof d.line_K); std::memcpy(d.line_Ric, e->w_Ric, sizeof d.line_Ric);
    for (int c = 0; c < 3; c++) { d.line_Tic[c] = e->tic[c]; d.gravity[c] = e->cfg.gravity[c]; }
    d.line_exact_jacobian = e->cfg.line_exact_jacobian;
    d.prior = e->prior; d.prior_block_kind = e->w_pk.data(); d.prior_block_index = e->w_pidx.data();
    if (!marg) {
        d.n_imu = (int)e->w_imu.size(); d.imu = e->w_imu.data(); d.imu_frame_i = e->w_imu_i.data(); d.imu_frame_j = e->w_imu_j.data();
        d.imu_device = e->w_imu_dev.empty() ? nullptr : e->w_imu_dev.data();      // (entries may be null: that factor's host copy is used)
        d.n_proj = (int)e->w_pi.size(); d.proj_frame_i = e->w_pi.data(); d.proj_frame_j = e->w_pj.data(); d.proj_feature = e->w_pl.data(); d.proj_pts = e->w_pts.data();
        d.n_line = (int)e->w_lf.size(); d.line_frame = e->w_lf.data(); d.line_data = e->w_ld.data();
    } else if (flag == MARGIN_OLD) {
        d.n_imu = (int)e->m_imu.size(); d.imu = e->m_imu.data(); d.imu_frame_i = e->m_imu_i.data(); d.imu_frame_j = e->m_imu_j.data();
        d.imu_device = e->m_imu_dev.empty() ? nullptr : e->m_imu_dev.data();
        d.n_proj = (int)e->m_pi.size(); d.proj_frame_i = e->m_pi.data(); d.proj_frame_j = e->m_pj.data(); d.proj_feature = e->m_pl.data(); d.proj_pts = e->m_pts.data();
    }
}

// factor set and drop sets MarginalizationInfo receives: estimator.cpp:1911-1986 (MARGIN_OLD), :2047-2063 (MARGIN_SECOND_NEW)
void build_marg(tcv_estimator *e, int flag) {
    e->m_imu.clear(); e->m_imu_dev.clear(); e->m_imu_i.clear(); e->m_imu_j.clear(); e->m_pi.clear(); e->m_pj.clear(); e->m_pl.clear(); e->m_pts.clear(); e->m_drop.clear();
    if (flag == MARGIN_OLD) {
        for (size_t k = 0; k < e->w_imu.size(); k++)
            if (e->w_imu_i[k] == 0 && e->w_imu[k].sum_dt < 10.0) { e->m_imu.push_back(e->w_imu[k]); e->m_imu_dev.push_back(e->w_imu_dev[k]); e->m_imu_i.push_back(0); e->m_imu_j.push_back(e->w_imu_j[k]); }
        std::vector<int> lms;
        for (size_t k = 0; k < e->w_pi.size(); k++)
            if (e->w_pi[k] == 0) {
                e->m_pi.push_back(0); e->m_pj.push_back(e->w_pj[k]); e->m_pl.push_back(e->w_pl[k]);
                e->m_pts.insert(e->m_pts.end(), e->w_pts.begin() + 6 * k, e->w_pts.begin() + 6 * k + 6);
                lms.push_back(e->w_pl[k]);
            }
        std::sort(lms.begin(), lms.end()); lms.erase(std::unique(lms.begin(), lms.end()), lms.end());
        e->m_drop.push_back(e->para_pose); e->m_drop.push_back(e->para_sb);
        for (int l : lms) e->m_drop.push_back(e->para_feature.data() + l);
    } else e->m_drop.push_back(e->para_pose + 7 * (W - 1));
}

// double2vector (:1565-1581; the gauge fix itself ran on the device), setDepth (feature_manager.cpp:379-397)
void apply_states(tcv_estimator *e) {
    for (int i = 0; i <= W; i++) {
        for (int c = 0; c < 3; c++) { e->Ps[i][c] = e->para_pose[7 * i + c]; e->Vs[i][c] = e->para_sb[9 * i + c]; e->Bas[i][c] = e->para_sb[9 * i + 3 + c]; e->Bgs[i][c] = e->para_sb[9 * i + 6 + c]; }
        e->Rs[i] = q2R(e->para_pose + 7 * i + 3);
    }
    for (int c = 0; c < 3; c++) e->tic[c] = e->para_ex[c];
    e->ric = q2R(e->para_ex + 3);
    for (size_t l = 0; l < e->sel.size(); l++) {
        Feature &f = e->features[e->sel[l]];
        f.depth = 1.0 / e->para_feature[l];
        f.solve_flag = f.depth < 0 ? 2 : 1;
    }
}

// getParameterBlocks(addr_shift) (marginalization_factor.cpp:301-321; estimator.cpp:2027-2039 / :2084-2104)
int take_prior(tcv_estimator *e, tcv_prior *np, int flag) {
    int m, n, nb, xs;
    int rc = tcv_prior_dims(np, &m, &n, &nb, &xs);
    if (rc != TCV_OK) return rc;
    std::vector<double *> addr(nb);
    rc = tcv_prior_keep_block_addresses(np, addr.data());
    if (rc != TCV_OK) return rc;
    std::vector<std::pair<int, int>> blocks;
    for (int k = 0; k < nb; k++) {
        int kind = -1, idx = 0;
        if (addr[k] >= e->para_pose && addr[k] < e->para_pose + (W + 1) * 7) { kind = 0; idx = (int)(addr[k] - e->para_pose) / 7; }
        else if (addr[k] >= e->para_sb && addr[k] < e->para_sb + (W + 1) * 9) { kind = 1; idx = (int)(addr[k] - e->para_sb) / 9; }
        else if (addr[k] == e->para_ex) { kind = 2; idx = 0; }
        else { tcv::set_error("estimator: kept block is not a pose / speed-bias / extrinsic block"); return TCV_ERR_INVALID; }
        if (kind < 2) {
            if (flag == MARGIN_OLD) idx -= 1;
            else if (idx == W) idx -= 1;
        }
        blocks.push_back({kind, idx});
    }
    if (e->prior) tcv_prior_destroy(e->prior);
    e->prior = np;
    e->prior_blocks = blocks;
    e->stats.prior_n = n;
    return TCV_OK;
}

// the window build_window() just made, copied for the tap (device-resident inputs are materialised: tcv_preint_export / tcv_prior_export)
int tap_window(tcv_estimator *e) {
    WindowTap &T = e->tap;
    T.have = false; T.applied = 0; T.iterations = 0; T.final_cost = 0;
    T.pose_in.assign(e->para_pose, e->para_pose + (W + 1) * 7); T.sb_in.assign(e->para_sb, e->para_sb + (W + 1) * 9);
    T.ex_in.assign(e->para_ex, e->para_ex + 7); T.feat_in.assign(e->para_feature.begin(), e->para_feature.begin() + e->sel.size());
    T.pose_out.assign((W + 1) * 7, 0.0); T.sb_out.assign((W + 1) * 9, 0.0); T.ex_out.assign(7, 0.0); T.feat_out.assign(e->sel.size(), 0.0);
    T.imu = e->w_imu; T.imu_i = e->w_imu_i; T.imu_j = e->w_imu_j;
    for (size_t k = 0; k < T.imu.size(); k++)
        if (k < e->w_imu_dev.size() && e->w_imu_dev[k]) { const int rc = tcv_preint_export(e->w_imu_dev[k], &T.imu[k]); if (rc != TCV_OK) return rc; }
    T.pi = e->w_pi; T.pj = e->w_pj; T.pl = e->w_pl; T.pts = e->w_pts; T.lf = e->w_lf; T.ld = e->w_ld;
    std::memcpy(T.Ric, e->w_Ric, sizeof T.Ric);
    T.marg_flag = e->marg_flag;
    T.pk = e->w_pk; T.pidx = e->w_pidx; T.psize.clear(); T.pcol.clear(); T.x0.clear(); T.J0.clear(); T.r0.clear();
    T.prior_m = T.prior_n = 0;
    if (e->prior) {
        int m, nn, nb, xs;
        int rc = tcv_prior_dims(e->prior, &m, &nn, &nb, &xs);
        if (rc != TCV_OK) return rc;
        T.psize.resize(nb); T.pcol.resize(nb); T.x0.resize(xs); T.J0.resize((size_t)nn * nn); T.r0.resize(nn);
        rc = tcv_prior_export(e->prior, T.psize.data(), T.pcol.data(), T.x0.data(), T.J0.data(), T.r0.data());
        if (rc != TCV_OK) return rc;
        T.prior_m = m; T.prior_n = nn;
    }
    T.have = true;
    return TCV_OK;
}

bool failure_detection(const tcv_estimator *e) {
    if (nrm(e->Bas[W]) > 2.5 || nrm(e->Bgs[W]) > 1.0) return true;
    if (e->have_last) {
        if (nrm(sub(e->Ps[W], e->last_P)) > 5 || std::fabs(e->Ps[W][2] - e->last_P[2]) > 1) return true;
    }
    return false;
}

void new_buf(tcv_estimator *e, int j) {
    ImuBuf &b = e->bufs[j];
    b.valid = true; b.acc0 = e->acc_0; b.gyr0 = e->gyr_0; b.ba = e->Bas[j]; b.bg = e->Bgs[j]; b.acc.clear(); b.gyr.clear();
}

void slide_window(tcv_estimator *e) {
    if (e->frame_count != W) return;
    if (e->marg_flag == MARGIN_OLD) {
        const M3 back_R0 = e->Rs[0];
        const V3 back_P0 = e->Ps[0];
        for (int i = 0; i < W; i++) {      // the swaps of :2131-2153 followed by the copy of slot W-1 into W
            e->Ps[i] = e->Ps[i + 1]; e->Rs[i] = e->Rs[i + 1]; e->Vs[i] = e->Vs[i + 1]; e->Bas[i] = e->Bas[i + 1]; e->Bgs[i] = e->Bgs[i + 1];
            // (moves, not copies: this runs once per frame and estimator on the host's critical path; slot W is refilled right below --
            // except the line bookkeeping, whose slot W keeps its content: WorldLinesInFOV[W] = WorldLinesInFOV[W-1], :2158)
            e->bufs[i] = std::move(e->bufs[i + 1]); e->pre[i] = e->pre[i + 1]; e->preh[i] = std::move(e->preh[i + 1]); e->pre_valid[i] = e->pre_valid[i + 1];
            if (i + 1 < W) { e->line_obs[i] = std::move(e->line_obs[i + 1]); e->fov[i] = std::move(e->fov[i + 1]); }
            else { e->line_obs[i] = e->line_obs[i + 1]; e->fov[i] = e->fov[i + 1]; }
        }
        e->pre_valid[W] = false;
        new_buf(e, W);
        // slideWindowOld (:2242-2259) -> removeBackShiftDepth (feature_manager.cpp:559-616)
        const M3 R0 = mm(back_R0, e->ric), R1 = mm(e->Rs[0], e->ric);
        const V3 P0 = add(back_P0, mv(back_R0, e->tic)), P1 = add(e->Ps[0], mv(e->Rs[0], e->tic));
        std::vector<Feature> kept;
        for (auto &f : e->features) {
            if (f.start != 0) { f.start -= 1; kept.push_back(std::move(f)); continue; }
            const V3 uv_i = f.obs.front();
            f.obs.erase(f.obs.begin());
            if (f.obs.size() < 2) continue;
            const V3 pts_j = mv(tr(R1), sub(add(mv(R0, scl(uv_i, f.depth)), P0), P1));
            f.depth = pts_j[2] > 0 ? pts_j[2] : e->cfg.init_depth;
            kept.push_back(std::move(f));
        }
        e->features.swap(kept);
        std::vector<LineFeature> keptl;                // line features: feature_manager.cpp:598-614
        for (auto &lf : e->linefeatures) {
            if (lf.start != 0) { lf.start -= 1; keptl.push_back(std::move(lf)); continue; }
            lf.obs.erase(lf.obs.begin());
            if (!lf.obs.empty()) keptl.push_back(std::move(lf));
        }
        e->linefeatures.swap(keptl);
    } else {
        // MARGIN_SECOND_NEW (:2189-2230): the newest frame replaces the second newest, their IMU buffers are concatenated
        ImuBuf &a = e->bufs[W - 1], &b = e->bufs[W];
        a.acc.insert(a.acc.end(), b.acc.begin(), b.acc.end()); a.gyr.insert(a.gyr.end(), b.gyr.begin(), b.gyr.end());
        e->pre_valid[W - 1] = false;
        e->Ps[W - 1] = e->Ps[W]; e->Rs[W - 1] = e->Rs[W]; e->Vs[W - 1] = e->Vs[W]; e->Bas[W - 1] = e->Bas[W]; e->Bgs[W - 1] = e->Bgs[W];
        e->line_obs[W - 1] = e->line_obs[W];
        new_buf(e, W);
        e->pre_valid[W] = false;
        std::vector<Feature> kept;                     // slideWindowNew -> removeFront(frame_count) (feature_manager.cpp:655-675)
        for (auto &f : e->features) {
            if (f.start == W) { f.start -= 1; kept.push_back(std::move(f)); continue; }
            if (f.end() < W - 1) { kept.push_back(std::move(f)); continue; }
            f.obs.erase(f.obs.begin() + (W - 1 - f.start));
            if (!f.obs.empty()) kept.push_back(std::move(f));
        }
        e->features.swap(kept);
        std::vector<LineFeature> keptl;                // feature_manager.cpp:677-695
        for (auto &lf : e->linefeatures) {
            if (lf.start == W) { lf.start -= 1; keptl.push_back(std::move(lf)); continue; }
            if (lf.start + (int)lf.obs.size() - 1 < W - 1) { keptl.push_back(std::move(lf)); continue; }
            lf.obs.erase(lf.obs.begin() + (W - 1 - lf.start));
            if (!lf.obs.empty()) keptl.push_back(std::move(lf));
        }
        e->linefeatures.swap(keptl);
        e->fov[W - 1] = e->fov[W];
    }
    std::vector<Feature> ok;                           // removeFailures (:399-408)
    for (auto &f : e->features) if (f.solve_flag != 2) ok.push_back(std::move(f));
    e->features.swap(ok);
}

}  // namespace

static void clear_state(tcv_estimator *e) {
    const tcv_estimator_config *cfg = &e->cfg;
    for (int i = 0; i <= W; i++) {
        e->Ps[i] = e->Vs[i] = e->Bas[i] = e->Bgs[i] = V3{0, 0, 0}; e->Rs[i] = eye(); e->pre_valid[i] = false; std::memset(&e->pre[i], 0, sizeof e->pre[i]); e->preh[i].reset();
        e->bufs[i] = ImuBuf(); e->line_obs[i].clear(); e->fov[i].clear();
    }
    for (int c = 0; c < 3; c++) e->tic[c] = cfg->tic[c];
    std::memcpy(e->ric.data(), cfg->ric, sizeof(double) * 9);
    std::memset(&e->stats, 0, sizeof e->stats);
    e->features.clear(); e->linefeatures.clear(); e->fov_ready = false;
    if (e->prior) { tcv_prior_destroy(e->prior); e->prior = nullptr; }
    e->prev.reset(); e->prev_k = -1;
    e->pend.reset(); e->pend_k = -1;
    e->prior_blocks.clear();
    e->frame_count = 0; e->marg_flag = MARGIN_OLD;
    e->have_acc0 = false; e->have_last = false;
    e->acc_0 = e->gyr_0 = e->last_P = V3{0, 0, 0};
    e->para_feature.clear(); e->sel.clear();
    e->n_line_obs_total = 0; e->phase = 0; e->opt_failed = 0; e->opt_msg.clear(); e->pend_failed = 0; e->pend_msg.clear();
}
extern "C" int tcv_estimator_create(tcv_estimator **out, const tcv_estimator_config *cfg) {
    if (!out || !cfg || !(cfg->imu_dt > 0) || !(cfg->focal_length > 0) || cfg->num_iterations < 1) { tcv::set_error("estimator_create: bad configuration"); return TCV_ERR_INVALID; }
    tcv_estimator *e = new tcv_estimator();
    e->cfg = *cfg;
    clear_state(e);
    *out = e;
    return TCV_OK;
}
// Estimator::clearState() + setParameter() (estimator.cpp:126-189, :39-52): what estimator_node.cpp does after failureDetection fired
// (:437-446) or on a restart request -- window, IMU buffers, pre-integrations, feature and line tracks, the marginalisation prior and the
// biases are dropped, the extrinsic goes back to the configured one; the line map (setParameters, :54-124) stays.
extern "C" int tcv_estimator_reset(tcv_estimator *e) {
    if (!e) return TCV_ERR_INVALID;
    clear_state(e);
    return TCV_OK;
}
extern "C" void tcv_estimator_destroy(tcv_estimator *e) {
    if (!e) return;
    if (e->prior) tcv_prior_destroy(e->prior);
    delete e;
}
extern "C" int tcv_estimator_set_biases(tcv_estimator *e, const double ba[3], const double bg[3]) {
    if (!e || !ba || !bg) return TCV_ERR_INVALID;
    for (int i = 0; i <= W; i++) for (int c = 0; c < 3; c++) { e->Bas[i][c] = ba[c]; e->Bgs[i][c] = bg[c]; }
    return TCV_OK;
}
extern "C" int tcv_estimator_set_line_map(tcv_estimator *e, int n, const double *lines3d, const double Rbw[9], const double Tbw[3]) {
    if (!e || n <= 0 || !lines3d || !Rbw || !Tbw) { tcv::set_error("estimator_set_line_map: bad argument"); return TCV_ERR_INVALID; }
    e->assoc = true; e->n_map = n;
    e->map_lines.assign(lines3d, lines3d + (size_t)6 * n);
    {      // (best effort: without the handle the association uploads the map per call, same results)
        tcv_line_map *md = nullptr;
        e->map_dev.reset();
        if (!getenv("TCV_EST_HOST_MAP") && tcv_line_map_create(&md, n, lines3d) == TCV_OK) e->map_dev = std::shared_ptr<tcv_line_map>(md, tcv_line_map_destroy);
    }
    std::memcpy(e->Rbw.data(), Rbw, sizeof(double) * 9);
    for (int c = 0; c < 3; c++) e->Tbw[c] = Tbw[c];
    return TCV_OK;
}

extern "C" int tcv_estimator_begin_frame(tcv_estimator *e, int n_imu, const double *acc, const double *gyr, int n_points, const int *point_ids,
                                         const double *points, int n_lines, const int *line_ids, const double *lines, const double *truth, int *ready) {
    if (!e || !ready || n_imu < 0 || n_points < 0 || n_lines < 0 || (n_points > 0 && (!point_ids || !points)) || (n_lines > 0 && !lines) ||
        (n_lines > 0 && e->assoc && !line_ids) || (n_imu > 0 && (!acc || !gyr))) { tcv::set_error("estimator_begin_frame: bad argument"); return TCV_ERR_INVALID; }
    if (e->phase != 0) { tcv::set_error("estimator_begin_frame: the previous frame has not been optimised and finished"); return TCV_ERR_INVALID; }
    if (acc && gyr) process_imu(e, n_imu, acc, gyr);
    e->marg_flag = add_features_check_parallax(e, n_points, point_ids, points, n_lines, line_ids, lines) ? MARGIN_OLD : MARGIN_SECOND_NEW;
    const int fc = e->frame_count;
    if (truth) { for (int c = 0; c < 3; c++) { e->Ps[fc][c] = truth[c]; e->Vs[fc][c] = truth[12 + c]; } std::memcpy(e->Rs[fc].data(), truth + 3, sizeof(double) * 9); }
    if (fc < W) {
        e->frame_count++;
        const int j = e->frame_count;
        e->Bas[j] = e->Bas[j - 1]; e->Bgs[j] = e->Bgs[j - 1]; e->Ps[j] = e->Ps[j - 1]; e->Rs[j] = e->Rs[j - 1]; e->Vs[j] = e->Vs[j - 1];
        *ready = 0;
        return TCV_OK;
    }
    *ready = 1;
    e->phase = 1;
    return TCV_OK;
}

// host-side time accounting of tcv_estimators_optimize (seconds, cumulative; read and cleared by tcv_estimators_profile):
// 0 pre-integration, 1 association + triangulation + window, 2 problem construction, 3 batch_create (pack + H2D), 4 kernels (launch to
// sync), 5 downloads (states, summaries, priors), 6 apply / prior chaining, 7 calls
// Process-wide state of tcv_estimators_optimize, shared by host threads that drive estimators on the same or on different GPUs: the
// profile accumulators.
namespace {
std::mutex g_mu;
double g_prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
void prof_add(int slot, double v) { std::lock_guard<std::mutex> g(g_mu); g_prof[slot] += v; }
// TCV_DEBUG_EST_CPU=1 (developer): CPU time of the whole process (caller, worker threads, runtime threads) between the laps of the same slots,
// printed by tcv_estimators_profile -- drive the estimators from ONE host thread to read it
double g_prof_cpu[8] = {0, 0, 0, 0, 0, 0, 0, 0};
bool prof_cpu_on() { static const bool on = getenv("TCV_DEBUG_EST_CPU") != nullptr; return on; }
double cpu_now_s() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
void prof_cpu_add(int slot, double v) { std::lock_guard<std::mutex> g(g_mu); g_prof_cpu[slot] += v; }
// per-estimator host work of a lock-step frame (association bookkeeping, triangulation, window and problem construction) on the packer's
// persistent worker threads: the estimators are independent objects, every task touches its own
void for_each_estimator(int n, const std::function<void(int)> &fn) {
    // TCV_EST_SERIAL_MAX = m (experiment, default 0): up to m estimators per call the tasks run on the caller.  A task is 10 - 20 us and a fork / join
    // ~50 us, so at four windows per call the association sections are faster alone (0.18 against 0.22 ms) -- but the problems then all come
    // out of the CALLER's malloc arena while the worker threads free last frame's behind its back, and their construction goes from 0.06 to
    // 0.13 ms (0.07 -> 0.3 ms at eight per call): 8 streams the same, 16 streams 6.4 K -> 5.6 K windows/s (profiles/r05_replay_host_workers.txt)
    static const int serial_max = [] { const char *e = getenv("TCV_EST_SERIAL_MAX"); return e ? atoi(e) : 0; }();
    if (n <= serial_max) { for (int i = 0; i < n; i++) fn(i); return; }
    tcv::HostOp op;
    const int nth = op.threads(n);
    if (nth <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
    tcv::parallel_items(n, nth, [&](int i, int) { fn(i); });
}
}  // namespace
static inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
extern "C" int tcv_estimators_begin_frames(tcv_estimator *const *es, int n, const tcv_frame_input *in, int *ready, int *rc_out) {
    if (!es || n <= 0 || !in || !ready) { tcv::set_error("estimators_begin_frames: bad argument"); return TCV_ERR_INVALID; }
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (es[i] == es[j]) { tcv::set_error("estimators_begin_frames: the same estimator twice"); return TCV_ERR_INVALID; }
    std::vector<int> rcs(n, TCV_OK);
    std::vector<std::string> msgs(n);
    for_each_estimator(n, [&](int i) {
        const tcv_frame_input &f = in[i];
        ready[i] = 0;
        rcs[i] = tcv_estimator_begin_frame(es[i], f.n_imu, f.acc, f.gyr, f.n_points, f.point_ids, f.points, f.n_lines, f.line_ids, f.lines, f.truth, &ready[i]);
        if (rcs[i] != TCV_OK) msgs[i] = tcv_last_error();      // (the text is per thread)
    });
    if (rc_out) for (int i = 0; i < n; i++) rc_out[i] = rcs[i];
    for (int i = 0; i < n; i++) if (rcs[i] != TCV_OK) { tcv::set_error(msgs[i]); return rcs[i]; }
    return TCV_OK;
}
extern "C" int tcv_estimators_kernel_profile(double *out8) {
    if (!out8) return TCV_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_kmu);
    for (int i = 0; i < 8; i++) { out8[i] = g_kern[i]; g_kern[i] = 0; }
    return TCV_OK;
}
extern "C" int tcv_estimators_profile(double *out8) {
    if (!out8) return TCV_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_mu);
    if (prof_cpu_on() && g_prof[7] > 0) {
        static const char *nm[7] = {"preintegrate", "assoc+triangulate+window", "problems", "batch_create", "kernels", "downloads", "apply"};
        for (int i = 0; i < 7; i++) fprintf(stderr, "[est cpu] %-26s wall %8.3f ms  process cpu %8.3f ms per call (%.0f calls)\n", nm[i], 1e3 * g_prof[i] / g_prof[7], 1e3 * g_prof_cpu[i] / g_prof[7], g_prof[7]);
    }
    for (int i = 0; i < 8; i++) { out8[i] = g_prof[i]; g_prof[i] = 0; g_prof_cpu[i] = 0; }
    return TCV_OK;
}

struct OptGroup {
    std::vector<int> idx;
    std::vector<char> dm;             // per window of the group: it marginalises
    bool any_marg = false;
    std::vector<tcv_problem *> P, M;
    std::vector<double *const *> drops;
    std::vector<int> ndrop;
    std::vector<tcv_solver_summary> sum;
    std::vector<tcv_prior *> newp;
    std::vector<int> est_rc;          // per estimator: TCV_ERR_NUMERIC when its own marginalisation failed
    std::string est_msg;
    tcv_batch *b = nullptr;
    int rc = TCV_OK;
    bool deferred = false;            // the marginalisation of this frame is launched by its estimators' next frame (EstInflight)
    bool dl_begun = false, marg_launched = false;      // the copy of the states / the marginalisation were enqueued behind the solve (marg_off_path)
};

struct OptRun {
    std::vector<tcv_estimator *> esv;
    int n = 0;
    OptGroup G[2];
    hipStream_t g_streams[2] = {nullptr, nullptr};
    bool host_priors = false, marg_off_path = false, marg_aux = false;
    int defer_from = 12, rc_all = TCV_OK;
    double t_mark = 0.0, t_begin0 = 0.0, t_end0 = 0.0;
    ~OptRun() { for (auto &g : G) { if (g.b) tcv_batch_destroy(g.b); for (auto *p : g.P) if (p) tcv_problem_destroy(p); for (auto *p : g.M) if (p) tcv_problem_destroy(p); for (auto *p : g.newp) if (p) tcv_prior_destroy(p); } }
};
// ---- a lock-step frame in two halves (tcv_estimators_optimize_begin / _end): everything up to the last command on the device, and the wait for the
// states with what follows.  OptRun carries what the second half needs.
static int optimize_begin(OptRun &R) {
    tcv_estimator *const *es = R.esv.data();
    const int n = R.n;
    typedef OptGroup Group;
    double &t_mark = R.t_mark;
    t_mark = now_s();
    R.t_begin0 = t_mark;
    double c_mark = prof_cpu_on() ? cpu_now_s() : 0.0;
    auto lap = [&](int slot) { const double t = now_s(); prof_add(slot, t - t_mark); t_mark = t; if (prof_cpu_on()) { const double c = cpu_now_s(); prof_cpu_add(slot, c - c_mark); c_mark = c; } };
    prof_add(7, 1);
    for (int i = 0; i < n; i++) if (!es[i] || es[i]->phase != 1) { tcv::set_error("estimators_optimize: an estimator has no full window waiting (begin_frame must report ready)"); return TCV_ERR_INVALID; }
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (es[i] == es[j]) { tcv::set_error("estimators_optimize: the same estimator twice"); return TCV_ERR_INVALID; }
    // solveOdometry up to the solver call: association, triangulation, vector2double + graph.  The association's device round trip comes
    // first and the pre-integration of the new IMU buffers is issued behind it (nobody on the host waits for that kernel: its results are
    // spliced into the batch on the device), so the kernel runs while the host builds the windows and problems.
    std::vector<AssocJob> jobs(n);
    {
        static const bool dbg1 = getenv("TCV_DEBUG_EST") != nullptr;      // developer: where this lap goes
        const double ta0 = now_s();
        std::vector<tcv_match_lines_args> calls;
        {      // (per estimator: poses, the detections of its line tracks, its field-of-view sets -- independent objects, the worker threads share them)
            std::vector<int> rcs(n, TCV_OK);
            std::vector<std::string> msgs(n);
            bool warm = true;
            for (int i = 0; i < n; i++) if (es[i]->assoc && !es[i]->fov_ready) warm = false;      // (the first frame's own device call per estimator stays on this thread)
            auto one = [&](int i) { if (es[i]->assoc) { rcs[i] = assoc_prepare(es[i], jobs[i]); if (rcs[i] != TCV_OK) msgs[i] = tcv_last_error(); } };
            if (warm) for_each_estimator(n, one); else for (int i = 0; i < n; i++) one(i);
            for (int i = 0; i < n; i++) if (rcs[i] != TCV_OK) { tcv::set_error(msgs[i]); return rcs[i]; }
            for (int i = 0; i < n; i++) if (es[i]->assoc && jobs[i].call) calls.push_back(jobs[i].args);
        }
        const double ta1 = now_s();
        if (!calls.empty()) { const int rc = tcv_match_lines_batch((int)calls.size(), calls.data()); if (rc != TCV_OK) return rc; }
        if (dbg1) fprintf(stderr, "[est] n %d: association prepare %.3f ms, device round trip (%d calls) %.3f ms\n", n, 1e3 * (ta1 - ta0), (int)calls.size(), 1e3 * (now_s() - ta1));
    }
    lap(1);
    // the previous frame's marginalisations (deferred: see EstInflight) go on the device NOW, behind the association's round trip, and every
    // estimator takes the prior it will build this frame's window on (getParameterBlocks, marginalization_factor.cpp:301-321)
    for (int i = 0; i < n; i++) {
        tcv_estimator *e = es[i];
        if (!e->pend) continue;
        std::shared_ptr<EstInflight> fl = e->pend;
        const int rcl = fl->launch((void *)tcv::util_stream());
        if (rcl != TCV_OK) {
            // the launch failed for the whole batch of the previous frame (EstInflight::launch keeps the verdict): every estimator of that batch loses
            // its new prior -- and only those: nothing is sticky, the rest of THIS call's estimators go on, and the affected ones report the failure
            // through finish_frame instead of blocking every later call until a reset (round-5 advisor finding)
            e->pend_failed = rcl; e->pend_msg = fl->launch_msg;
            e->pend.reset(); e->pend_k = -1;
            if (e->prior) { tcv_prior_destroy(e->prior); e->prior = nullptr; }
            e->prior_blocks.clear(); e->stats.prior_n = 0;
            continue;
        }
        tcv_prior *np = fl->take(e->pend_k);
        if (!np) { tcv::set_error("estimators_optimize: the previous frame's marginalisation left no prior for this estimator"); return TCV_ERR_INVALID; }
        const int rct = take_prior(e, np, e->pend_flag);
        if (rct != TCV_OK) { tcv_prior_destroy(np); return rct; }
        e->prev = fl; e->prev_k = e->pend_k;
        e->pend.reset(); e->pend_k = -1;
    }
    // one pre-integration call for every stale IMU buffer of every estimator
    {
        std::vector<int> first, count;
        std::vector<double> samples, init;
        std::vector<std::pair<int, int>> who;
        for (int i = 0; i < n; i++)
            for (int j = 1; j <= W; j++) {
                tcv_estimator *e = es[i];
                if (e->pre_valid[j]) continue;
                const ImuBuf &b = e->bufs[j];
                if (!b.valid) { tcv::set_error("estimators_optimize: missing IMU buffer"); return TCV_ERR_INVALID; }
                who.push_back({i, j});
                first.push_back((int)samples.size() / 7); count.push_back((int)b.acc.size());
                for (size_t k = 0; k < b.acc.size(); k++) { samples.push_back(e->cfg.imu_dt); samples.insert(samples.end(), b.acc[k].begin(), b.acc[k].end()); samples.insert(samples.end(), b.gyr[k].begin(), b.gyr[k].end()); }
                init.insert(init.end(), b.acc0.begin(), b.acc0.end()); init.insert(init.end(), b.gyr0.begin(), b.gyr0.end());
                init.insert(init.end(), b.ba.begin(), b.ba.end()); init.insert(init.end(), b.bg.begin(), b.bg.end());
            }
        if (!who.empty()) {
            const tcv_estimator_config &c = es[0]->cfg;
            const double noise[4] = {c.acc_n, c.gyr_n, c.acc_w, c.gyr_w};
            if (samples.empty()) samples.push_back(0.0);
            // pre_integrations[] stay on the device (the reference keeps them alive between frames, too): a handle per buffer, sum_dt on the
            // host; nobody waits for the kernel (tcv_preintegrate_device).  TCV_EST_HOST_PREINT=1: round 3's round trip (3.7 KB down per buffer,
            // 2.3 KB up per factor and window), same bits
            if (getenv("TCV_EST_HOST_PREINT") || getenv("TCV_EST_HOST_STATE")) {
                std::vector<tcv_imu_preintegration> out(who.size());
                const int rc = tcv_preintegrate((int)who.size(), first.data(), count.data(), samples.data(), (int)samples.size() / 7, init.data(), noise, out.data());
                if (rc != TCV_OK) return rc;
                for (size_t k = 0; k < who.size(); k++) { tcv_estimator *e = es[who[k].first]; e->pre[who[k].second] = out[k]; e->preh[who[k].second].reset(); e->pre_valid[who[k].second] = true; }
            } else {
                std::vector<tcv_preint *> hd(who.size(), nullptr);
                const int rc = tcv_preintegrate_device((int)who.size(), first.data(), count.data(), samples.data(), (int)samples.size() / 7, init.data(), noise, hd.data());
                if (rc != TCV_OK) return rc;
                for (size_t k = 0; k < who.size(); k++) {
                    tcv_estimator *e = es[who[k].first];
                    const int j = who[k].second;
                    e->preh[j] = std::shared_ptr<tcv_preint>(hd[k], tcv_preint_destroy);
                    std::memset(&e->pre[j], 0, sizeof e->pre[j]);
                    e->pre[j].sum_dt = tcv_preint_sum_dt(hd[k]);
                    e->pre_valid[j] = true;
                }
            }
        }
    }
    lap(0);
    {
        for_each_estimator(n, [&](int i) {
            tcv_estimator *e = es[i];
            if (e->assoc) assoc_finish(e, jobs[i]);
            triangulate(e);
            build_window(e);
        });
        for (int i = 0; i < n; i++) if (es[i]->tap.on) { const int rc = tap_window(es[i]); if (rc != TCV_OK) return rc; }
    }
    lap(1);
    // One device batch per frame: the windows that marginalise (MARGIN_OLD, or MARGIN_SECOND_NEW with para_Pose[WINDOW_SIZE - 1] in the prior,
    // estimator.cpp:2049-2050) carry a marginalisation problem, the others a NULL entry (tcv_batch_create).  TCV_EST_TWO_BATCHES=1: round 3's
    // split into a batch that marginalises and one that only solves (two tcv_batch_create calls, two launches side by side) -- same bits.
    static const bool two_batches = getenv("TCV_EST_TWO_BATCHES") != nullptr;
    std::vector<char> do_marg(n);
    for (int i = 0; i < n; i++) {
        bool has = false;
        for (auto &b : es[i]->prior_blocks) if (b.first == 0 && b.second == W - 1) has = true;
        do_marg[i] = es[i]->marg_flag == MARGIN_OLD || (es[i]->prior && has);
    }
    // The two batches of a frame are independent: both are created first, then their kernels are launched on two HIP streams and run
    // side by side (a lock-step frame is latency bound: a handful of windows on a handful of CUs), then both are collected.
    OptGroup (&G)[2] = R.G;
    // Every kernel of the frame goes on the CALLING THREAD's utility stream -- the stream tcv_batch_create's uploads, the device-to-device
    // splices and the downloads of this thread use anyway: host threads that drive their own estimators overlap on the device (a stream pair
    // shared by the threads of a device, as until round 4, serialises their kernels: 8 streams on 4 host threads 1 000 against 2 400 windows/s),
    // and with one stream per thread no small copy of one thread waits behind another thread's 2 ms solve kernel on a shared hardware queue.
    // (TCV_EST_TWO_BATCHES: the two batches of a frame then run one after the other.)
    hipStream_t (&g_streams)[2] = R.g_streams;
    g_streams[0] = g_streams[1] = tcv::util_stream();
    // last_marginalization_info stays on the device (estimator.h:176-177: the reference keeps it alive between frames, too): the new
    // priors are handles on the batch's result buffer, the next frame's tcv_batch_create splices them device-to-device.
    // TCV_EST_HOST_PRIORS=1: round 3's host round trip (62 KB down, 46 KB up per window), the A/B partner -- same bits.
    // Faster at every batch size once every host thread launches on its own stream (512-window passes: 171 K against 117 K windows/s; eight
    // replay streams on two host threads: 2 370 - 2 480 against 2 250 - 2 280 windows/s).  TCV_EST_HOST_STATE=1 (both) / TCV_EST_HOST_PRIORS=1 /
    // TCV_EST_HOST_PREINT=1 take the host round trip.
    const bool host_priors = getenv("TCV_EST_HOST_PRIORS") != nullptr || getenv("TCV_EST_HOST_STATE") != nullptr;
    // With the priors on the device the host needs nothing of the marginalisation but its status: the call returns once the solve and the
    // gauge fix are done and the states are applied; the marginalisation is launched behind them and runs while the caller finishes the
    // frame and starts the next one (tcv_batch_get_priors_device_async).  Its status is read at each estimator's NEXT frame, after that
    // frame's solve: a window solved on a prior whose marginalisation failed is not applied and reports the failure (finish_frame), one
    // frame late.  TCV_EST_MARG_WAIT=1: wait for it as before (same bits).
    const bool marg_off_path = !host_priors && !two_batches && getenv("TCV_EST_MARG_WAIT") == nullptr;
    int &rc_all = R.rc_all;
    rc_all = TCV_OK;
    for (int group = 1; group >= 0; group--) {
        Group &g = G[group];
        for (int i = 0; i < n; i++) if ((two_batches ? (int)do_marg[i] : 1) == group) { g.idx.push_back(i); g.dm.push_back(do_marg[i]); g.any_marg = g.any_marg || do_marg[i]; }
        if (g.idx.empty()) continue;
        const int nb = (int)g.idx.size();
        g.P.assign(nb, nullptr); g.M.assign(nb, nullptr); g.drops.assign(nb, nullptr); g.ndrop.assign(nb, 0);
        {
            std::vector<int> rcs(nb, TCV_OK);
            std::vector<std::string> msgs(nb);
            for_each_estimator(nb, [&](int k) {      // (the error text is per thread: a worker's is carried over)
                tcv_estimator *e = es[g.idx[k]];
                tcv_window_desc d;
                fill_desc(e, d, false, e->marg_flag);
                rcs[k] = tcv_problem_from_window(&d, &g.P[k]);
                if (rcs[k] == TCV_OK && g.dm[k] && !marg_off_path) {
                    build_marg(e, e->marg_flag);
                    fill_desc(e, d, true, e->marg_flag);
                    rcs[k] = tcv_problem_from_window(&d, &g.M[k]);
                    g.drops[k] = e->m_drop.data(); g.ndrop[k] = (int)e->m_drop.size();
                }
                if (rcs[k] != TCV_OK) msgs[k] = tcv_last_error();
            });
            for (int k = 0; k < nb && g.rc == TCV_OK; k++) if (rcs[k] != TCV_OK) { g.rc = rcs[k]; tcv::set_error(msgs[k]); }
        }
        lap(2);
        // (marg_off_path: the batch is created without its marginalisation problems -- they are built and attached below, while the solve runs)
        const bool with_marg = g.any_marg && !marg_off_path;
        if (g.rc == TCV_OK) g.rc = tcv_batch_create(&g.b, g.P.data(), with_marg ? g.M.data() : nullptr, with_marg ? g.drops.data() : nullptr, with_marg ? g.ndrop.data() : nullptr, nb);
        lap(3);
    }
    for (int group = 1; group >= 0; group--) {
        Group &g = G[group];
        if (g.idx.empty() || g.rc != TCV_OK) continue;
        tcv_solver_options o;
        tcv_solver_options_default(&o);
        o.max_num_iterations = es[0]->cfg.num_iterations; o.fixed_iterations = es[0]->cfg.fixed_iterations;
        if (es[0]->cfg.solver_time > 0.0 && !o.fixed_iterations) {      // estimator.cpp:1894-1897 (one budget per batch: include/tcv_estimator.h)
            bool any_old = false;
            for (int i : g.idx) any_old = any_old || es[i]->marg_flag == MARGIN_OLD;
            o.max_solver_time_in_seconds = es[0]->cfg.solver_time * (any_old ? 4.0 / 5.0 : 1.0);
        }
        void *st = (void *)g_streams[group];
        // double2vector() in the solve kernel's epilogue (the tcv_batch_gauge_fix below is then a no-op): one launch and its gap less per frame.
        // TCV_EST_SEPARATE_GAUGE=1: the stand-alone kernel, as up to round 5 (same bits)
        static const bool separate_gauge = getenv("TCV_EST_SEPARATE_GAUGE") != nullptr;
        if (!separate_gauge) g.rc = tcv_batch_set_fused_gauge_fix(g.b, 1);
        if (g.rc == TCV_OK) g.rc = tcv_batch_solve(g.b, &o, st);
        if (g.rc == TCV_OK) g.rc = tcv_batch_gauge_fix(g.b, st);
        if (g.rc == TCV_OK && g.any_marg && !marg_off_path) g.rc = tcv_batch_marginalize(g.b, st);
        // the copy of the states (and of the summary heads) goes on the stream right behind the gauge fix -- BEFORE the upload of the marginalisation
        // problems attached below, which used to sit between them (a lock-step frame's GPU timeline: 10 us of upload, a fill and their launch gaps,
        // ~35 us before the states left; tools/gpu_calls.md#r05_gpu_z43)
        if (g.rc == TCV_OK && marg_off_path) { g.rc = tcv_batch_download_states_begin(g.b, st); g.dl_begun = g.rc == TCV_OK; }
    }
    if (marg_off_path) {      // the solve is on the device: the marginalisation problems of the frame, their packing and upload meanwhile
        for (int group = 1; group >= 0; group--) {
            Group &g = G[group];
            if (g.idx.empty() || g.rc != TCV_OK || !g.any_marg) continue;
            const int nb = (int)g.idx.size();
            std::vector<int> rcs(nb, TCV_OK);
            std::vector<std::string> msgs(nb);
            for_each_estimator(nb, [&](int k) {
                if (!g.dm[k]) return;
                tcv_estimator *e = es[g.idx[k]];
                tcv_window_desc d;
                build_marg(e, e->marg_flag);
                fill_desc(e, d, true, e->marg_flag);
                rcs[k] = tcv_problem_from_window(&d, &g.M[k]);
                g.drops[k] = e->m_drop.data(); g.ndrop[k] = (int)e->m_drop.size();
                if (rcs[k] != TCV_OK) msgs[k] = tcv_last_error();
            });
            for (int k = 0; k < nb && g.rc == TCV_OK; k++) if (rcs[k] != TCV_OK) { g.rc = rcs[k]; tcv::set_error(msgs[k]); }
            if (g.rc == TCV_OK) g.rc = tcv_batch_attach_marginalization(g.b, g.M.data(), g.drops.data(), g.ndrop.data());
        }
    }
    // Deferred marginalisation (EstInflight::launch) for frames of many windows, whose host side is long enough to hide the kernel behind the NEXT
    // frame's window construction: 128 streams on two host threads 17.4 K against 15.7 K windows/s.  A frame of a few windows enqueues it here,
    // right behind the copy of its states -- deferred, its 0.45 ms would end up in front of the next frame's upload (8 streams: 3 050 against
    // 3 250 windows/s).  TCV_EST_MARG_DEFER=n: defer from n windows per call on (0: never, 1: always).
    const char *e_defer = getenv("TCV_EST_MARG_DEFER");      // (read per call: the tests switch it)
    const int defer_from = e_defer ? atoi(e_defer) : 12;
    // the marginalisation of a frame of few windows runs on the thread's SECOND stream (ordered behind the copy of the states by an event): the next
    // frame's association round trip on the main stream then does not queue behind the kernel (0.45 -> 0.2 ms of a 2.4 ms frame: 8 streams on one
    // host thread 3 200 -> 3 700 windows/s; on two host threads only if the runtime has a hardware queue per stream, GPU_MAX_HW_QUEUES >= 8:
    // 3 400 -> 3 750; otherwise no change).  TCV_EST_MARG_AUX=0: on the main stream, as until round 5.
    static const bool marg_aux = !(getenv("TCV_EST_MARG_AUX") && atoi(getenv("TCV_EST_MARG_AUX")) == 0);
    if (marg_off_path) {
        // everything the frame still needs from the device goes on the stream NOW, while the solve runs: the copy of the states (and of the
        // summary heads), and behind it the marginalisation with its no-wait prior handles -- the kernel starts the moment the states have
        // left instead of a host round trip (wake-up, unpacking, launch) later, which the next frame's association would wait out
        for (int group = 1; group >= 0; group--) {
            Group &g = G[group];
            if (g.idx.empty() || g.rc != TCV_OK) continue;
            const int nb = (int)g.idx.size();
            void *st = (void *)g_streams[group];
            if (!g.dl_begun) { g.rc = tcv_batch_download_states_begin(g.b, st); g.dl_begun = g.rc == TCV_OK; }
            g.newp.assign(nb, nullptr);
            const bool eager = defer_from <= 0 || nb < defer_from;
            if (g.rc == TCV_OK && g.any_marg && eager) {
                // (TCV_EST_MARG_AUX=1: on the thread's SECOND stream, ordered behind the copy of the states by an event -- the next frame's
                // association round trip on the main stream then does not queue behind the kernel)
                g.rc = tcv_batch_marginalize(g.b, marg_aux ? (void *)tcv::aux_stream() : st);
                if (g.rc == TCV_OK) g.rc = tcv_batch_get_priors_device_async(g.b, g.newp.data(), nb);
                g.marg_launched = g.rc == TCV_OK;
            }
        }
    }
    R.host_priors = host_priors; R.marg_off_path = marg_off_path; R.defer_from = defer_from; R.marg_aux = marg_aux;
    if (getenv("TCV_DEBUG_PIPE")) fprintf(stderr, "[pipe] %p begin  n %d  %.3f -> %.3f ms\n", (void *)tcv::util_stream(), n, 1e3 * (R.t_begin0 - 1.7e9 * 0), 1e3 * now_s());
    return TCV_OK;
}

static int optimize_end(OptRun &R) {
    R.t_end0 = now_s();
    tcv_estimator *const *es = R.esv.data();
    typedef OptGroup Group;
    double &t_mark = R.t_mark;
    double c_mark = prof_cpu_on() ? cpu_now_s() : 0.0;
    auto lap = [&](int slot) { const double t = now_s(); prof_add(slot, t - t_mark); t_mark = t; if (prof_cpu_on()) { const double c = cpu_now_s(); prof_cpu_add(slot, c - c_mark); c_mark = c; } };
    OptGroup (&G)[2] = R.G;
    hipStream_t (&g_streams)[2] = R.g_streams;
    int &rc_all = R.rc_all;
    const bool host_priors = R.host_priors, marg_off_path = R.marg_off_path, marg_aux = R.marg_aux;
    const int defer_from = R.defer_from;
    for (int group = 1; group >= 0; group--) {
        Group &g = G[group];
        if (g.idx.empty()) continue;
        const int nb = (int)g.idx.size();
        g.sum.resize(nb);
        if (g.dl_begun) {      // waits for the copy (an event behind the solve and the gauge fix), not for the marginalisation behind it
            std::vector<int> its(nb, 0), tms(nb, 0);
            std::vector<double> fcs(nb, 0.0);
            const int rcd = tcv_batch_download_states_end(g.b, its.data(), tms.data(), fcs.data());
            if (g.rc == TCV_OK) g.rc = rcd;
            for (int k = 0; k < nb; k++) { std::memset(&g.sum[k], 0, sizeof g.sum[k]); g.sum[k].num_iterations = its[k]; g.sum[k].termination = tms[k]; g.sum[k].final_cost = fcs[k]; }
        } else if (g.rc == TCV_OK) g.rc = tcv_batch_synchronize(g.b);
        double ms = 0;
        if (g.rc == TCV_OK && tcv_batch_stats(g.b, nullptr, &ms, nullptr) == TCV_OK && ms > 0) { kern_add(0, ms); kern_add(1, 1); }
        if (getenv("TCV_DEBUG_PIPE")) fprintf(stderr, "[pipe] %p end    n %d  entered %.3f  states at %.3f ms (solve kernel %.3f ms)\n", (void *)tcv::util_stream(), nb, 1e3 * R.t_end0, 1e3 * now_s(), ms);
    }
    lap(4);
    for (int group = 1; group >= 0; group--) {
        Group &g = G[group];
        if (g.idx.empty()) continue;
        const int nb = (int)g.idx.size();
        if ((int)g.newp.size() != nb) g.newp.assign(nb, nullptr);
        static const bool dbg_dl = getenv("TCV_DEBUG_EST") != nullptr;      // developer: where the "downloads" lap goes
        const double td0 = now_s();
        if (!g.dl_begun) {      // (the variants that wait for the marginalisation inside the call: states + the three summary numbers, one device round trip)
            std::vector<int> its(nb, 0), tms(nb, 0);
            std::vector<double> fcs(nb, 0.0);
            if (g.rc == TCV_OK) g.rc = tcv_batch_download_states_brief(g.b, its.data(), tms.data(), fcs.data());
            for (int k = 0; k < nb; k++) { std::memset(&g.sum[k], 0, sizeof g.sum[k]); g.sum[k].num_iterations = its[k]; g.sum[k].termination = tms[k]; g.sum[k].final_cost = fcs[k]; }
        }
        const double td1 = now_s();
        const double td2 = td1;
        bool have_dev = false;
        const bool marg_eager = defer_from <= 0 || nb < defer_from;
        const bool defer_marg = g.rc == TCV_OK && g.any_marg && marg_off_path && !marg_eager;
        if (defer_marg) have_dev = true;      // (nothing to fetch now: EstInflight::launch at the estimators' next frame)
        if (g.marg_launched) have_dev = true; // (enqueued behind the copy of the states, above; the handles are in g.newp)
        else if (g.rc == TCV_OK && g.any_marg && marg_off_path && marg_eager) {      // TCV_EST_MARG_AUX=1: on the thread's second stream, now that the states are on the host
            // (one host thread 2 570 - 2 720 against 2 590 windows/s, two host threads 2 490 - 2 550 against 2 940: two streams per thread share
            // the runtime's four hardware queues again)
            g.rc = tcv_batch_marginalize(g.b, marg_aux ? (void *)tcv::aux_stream() : (void *)g_streams[group]);
            if (g.rc == TCV_OK) g.rc = tcv_batch_get_priors_device_async(g.b, g.newp.data(), nb);
            have_dev = g.rc == TCV_OK;
        }
        if (g.rc == TCV_OK && g.any_marg && !host_priors && !marg_off_path) have_dev = tcv_batch_get_priors_device(g.b, g.newp.data(), nb) == TCV_OK;      // (a window that failed: the per-window path below says which)
        if (g.rc == TCV_OK && g.any_marg && !have_dev) g.rc = tcv_batch_download_priors_compact(g.b);
        g.est_rc.assign(nb, TCV_OK);
        if (g.rc == TCV_OK) {      // algorithmic bytes of the frame's windows x linearisations (the initial one + one per iteration)
            double bytes = 0, lin = 0;
            for (int k = 0; k < nb; k++) {
                const tcv_estimator *e = es[g.idx[k]];
                const double L = (double)e->sel.size(), n = e->prior ? (double)e->stats.prior_n : 0.0, nl = (double)std::max(1, g.sum[k].num_iterations);      // (ceres numbering: iteration 0 is the initial linearisation)
                const double per = 8.0 * ((77 + 99 + 7 + L) + 287.0 * e->w_imu.size() + 6.0 * e->w_pi.size() + 9.0 * e->w_lf.size() + 21 + (n > 0 ? n * n + n + 86 : 0))
                                   + 4.0 * (4.0 * e->w_imu.size() + 4.0 * e->w_pi.size() + e->w_lf.size()) + 8.0 * ((171 + L) + 1);
                bytes += per * nl; lin += nl;
            }
            kern_add(4, nb); kern_add(5, bytes); kern_add(6, lin);
        }
        if (g.rc == TCV_OK)
            for (int k = 0; k < nb; k++) {     // ceres::Solve's FAILURE (no valid step / a cooperative group that timed out): the window's states are not applied
                if (g.sum[k].termination == 5 || !(g.sum[k].final_cost == g.sum[k].final_cost)) { g.est_rc[k] = TCV_ERR_NUMERIC; g.est_msg = "solver failure (no valid step, NaN cost or workgroup time-out)"; }
                // the marginalisation that made this window's prior was still running when the previous frame returned: its verdict now (it
                // finished before this frame's solve started)
                tcv_estimator *e = es[g.idx[k]];
                if (e->pend_failed != TCV_OK) {
                    g.est_rc[k] = e->pend_failed; g.est_msg = "the previous frame's marginalisation could not be launched (" + e->pend_msg + "): this window was solved without its prior";
                    e->pend_failed = TCV_OK; e->pend_msg.clear();
                }
                if (e->prev) {
                    const int stp = e->prev->status_of(e->prev_k);
                    if (stp != 0 && stp != 2) { g.est_rc[k] = TCV_ERR_NUMERIC; g.est_msg = "the previous frame's marginalisation failed (eigen-solver sweep cap or NaN): this window was solved on an invalid prior"; }
                    e->prev.reset(); e->prev_k = -1;      // (the last estimator of that frame to let go destroys its batch)
                }
            }
        if (g.rc == TCV_OK && g.any_marg) {
            std::vector<std::string> msgs(nb);
            int cur_dev = 0;
            (void)hipGetDevice(&cur_dev);
            for_each_estimator(nb, [&](int k) {
                // a window whose marginalisation did not converge (TCV_ERR_NUMERIC) fails alone: the other estimators of the lock-step
                // batch are applied, this one reports the failure from tcv_estimator_finish_frame
                if (g.est_rc[k] != TCV_OK) { if (g.newp[k]) { tcv_prior_destroy(g.newp[k]); g.newp[k] = nullptr; } return; }
                if (have_dev || !g.dm[k]) return;
                (void)hipSetDevice(cur_dev);      // (a worker thread's current device is its own: the fallback copy of a window must see the batch's)
                g.est_rc[k] = tcv_batch_get_prior(g.b, k, &g.newp[k]);      // (host path: the window's prior out of the downloaded blob)
                if (g.est_rc[k] != TCV_OK) msgs[k] = tcv_last_error();
            });
            for (int k = 0; k < nb; k++) {
                if (g.est_rc[k] == TCV_OK || msgs[k].empty()) continue;
                if (g.est_rc[k] != TCV_ERR_NUMERIC) { g.rc = g.est_rc[k]; tcv::set_error(msgs[k]); break; }
                g.est_msg = msgs[k];
            }
        }
        const double td3 = now_s();
        if (g.rc == TCV_OK && g.any_marg && marg_off_path && have_dev) {      // the batch lives on until its marginalisation has been asked about
            // (the batch alone: its problems were read for the last time by tcv_batch_download_states / tcv_batch_attach_marginalization and
            // go now, like the old prior they point to)
            auto fl = std::make_shared<EstInflight>();
            fl->b = g.b;
            fl->deferred = defer_marg;
            for (int k = 0; k < nb; k++) fl->n_marg += g.dm[k] ? 1 : 0;
            g.b = nullptr;
            for (int k = 0; k < nb; k++)
                if (g.dm[k] && g.est_rc[k] == TCV_OK) {
                    tcv_estimator *e = es[g.idx[k]];
                    if (defer_marg) { e->pend = fl; e->pend_k = k; e->pend_flag = e->marg_flag; const int pn = tcv_marg_layout_n(fl->b, k); if (pn >= 0) e->stats.prior_n = pn; }      // (the new prior's n is part of the attached problem's layout)
                    else { e->prev = fl; e->prev_k = k; }
                }
            g.deferred = defer_marg;
        }
        if (g.b) { tcv_batch_destroy(g.b); g.b = nullptr; }      // (the ticket destroys what is left in it)
        const double td4 = now_s();
        {      // the frame's problems have been read for the last time: destroyed on a worker thread, behind the caller's back
            auto dead = std::make_shared<std::vector<tcv_problem *>>();
            dead->reserve(2 * (size_t)nb);
            for (int k = 0; k < nb; k++) { if (g.P[k]) dead->push_back(g.P[k]); if (g.M[k]) dead->push_back(g.M[k]); g.P[k] = g.M[k] = nullptr; }
            tcv::async_run([dead] { for (tcv_problem *q : *dead) tcv_problem_destroy(q); });
        }
        if (dbg_dl) fprintf(stderr, "[est] group %d n %d: states %.3f ms, summaries %.3f ms, priors %.3f ms, batch destroy %.3f ms, problems destroy %.3f ms\n", group, nb,
                            1e3 * (td1 - td0), 1e3 * (td2 - td1), 1e3 * (td3 - td2), 1e3 * (td4 - td3), 1e3 * (now_s() - td4));
        if (g.rc != TCV_OK && rc_all == TCV_OK) rc_all = g.rc;
    }
    lap(5);
    for (int group = 1; group >= 0; group--) {
        Group &g = G[group];
        if (g.idx.empty()) continue;
        const int nb = (int)g.idx.size();
        if (rc_all != TCV_OK) { for (auto *&p : g.newp) if (p) { tcv_prior_destroy(p); p = nullptr; } continue; }
        for (int k = 0; k < nb; k++) {
            tcv_estimator *e = es[g.idx[k]];
            e->opt_failed = TCV_OK;
            if (g.est_rc[k] != TCV_OK) {      // this estimator's window failed numerically: nothing is applied, finish_frame reports it
                e->opt_failed = g.est_rc[k]; e->opt_msg = g.est_msg; e->phase = 2;
                continue;
            }
            apply_states(e);
            if (e->tap.on && e->tap.have) {
                WindowTap &T = e->tap;
                T.pose_out.assign(e->para_pose, e->para_pose + (W + 1) * 7); T.sb_out.assign(e->para_sb, e->para_sb + (W + 1) * 9);
                T.ex_out.assign(e->para_ex, e->para_ex + 7); T.feat_out = e->para_feature;
                T.iterations = g.sum[k].num_iterations; T.final_cost = g.sum[k].final_cost; T.applied = 1;
            }
            e->stats.marg_flag = e->marg_flag; e->stats.n_landmarks = (int)e->sel.size(); e->stats.n_proj = (int)e->w_pi.size(); e->stats.n_line = (int)e->w_lf.size();
            e->stats.n_line_obs = e->n_line_obs_total; e->stats.iterations = g.sum[k].num_iterations; e->stats.termination = g.sum[k].termination; e->stats.final_cost = g.sum[k].final_cost;
            if (g.dm[k] && g.deferred) { }      // (the new prior is taken at the start of the next frame; stats.prior_n was set with the hand-over above)
            else if (g.dm[k]) {
                const int rc = take_prior(e, g.newp[k], e->marg_flag);
                if (rc != TCV_OK) {      // (take_prior keeps the old prior on failure: the new one and the ones not handed over yet are released)
                    for (int k2 = k; k2 < nb; k2++) if (g.newp[k2]) { tcv_prior_destroy(g.newp[k2]); g.newp[k2] = nullptr; }
                    rc_all = rc; break;
                }
                g.newp[k] = nullptr;
            }
            else e->stats.prior_n = e->prior ? e->stats.prior_n : 0;
            e->phase = 2;
        }
        lap(6);
    }
    return rc_all;
}


// The frame in two calls, for callers that overlap the host side of one group of estimators with the device side of another (bench.py --mode
// replay: every host thread alternates between two halves of its streams): _begin returns when the frame's last command is on the device,
// _end waits for the states and applies them.  tcv_estimators_optimize = _begin + _end.  Between the two calls the estimators of the ticket
// must not be touched; the ticket is consumed by _end (also on failure).
extern "C" int tcv_estimators_optimize_begin(tcv_estimator *const *es, int n, tcv_opt_ticket **out) {
    if (!es || n <= 0 || !out) return TCV_ERR_INVALID;
    OptRun *R = new OptRun();
    R->esv.assign(es, es + n); R->n = n;
    const int rc = optimize_begin(*R);
    if (rc != TCV_OK) { delete R; *out = nullptr; return rc; }
    *out = reinterpret_cast<tcv_opt_ticket *>(R);
    return TCV_OK;
}
extern "C" int tcv_estimators_optimize_end(tcv_opt_ticket *t) {
    if (!t) return TCV_ERR_INVALID;
    OptRun *R = reinterpret_cast<OptRun *>(t);
    const int rc = optimize_end(*R);
    delete R;
    return rc;
}
extern "C" int tcv_estimators_optimize(tcv_estimator *const *es, int n) {
    tcv_opt_ticket *t = nullptr;
    const int rc = tcv_estimators_optimize_begin(es, n, &t);
    if (rc != TCV_OK) return rc;
    return tcv_estimators_optimize_end(t);
}

extern "C" int tcv_estimator_finish_frame(tcv_estimator *e, double P[3], double q[4], double V[3]) {
    if (!e || !P || !q || !V) return TCV_ERR_INVALID;
    if (e->phase != 2) { tcv::set_error("estimator_finish_frame: the window has not been optimised"); return TCV_ERR_INVALID; }
    e->phase = 0;
    if (e->opt_failed != TCV_OK) {      // the window's own solve / marginalisation failed in the lock-step batch: like failureDetection, the caller resets
        const int rc = e->opt_failed; e->opt_failed = TCV_OK;
        tcv::set_error("estimator: optimisation of this window failed: " + e->opt_msg);
        return rc;
    }
    if (failure_detection(e)) { tcv::set_error("failure detection (estimator.cpp:1629-1675): the estimator diverged"); return TCV_ERR_NUMERIC; }
    for (int c = 0; c < 3; c++) { P[c] = e->Ps[W][c]; V[c] = e->Vs[W][c]; }
    R2q(e->Rs[W], q);
    slide_window(e);
    e->last_P = e->Ps[W]; e->have_last = true;
    return TCV_OK;
}
extern "C" int tcv_estimators_finish_frames(tcv_estimator *const *es, int n, double *P, double *q, double *V, int *rc, tcv_estimator_stats *stats) {
    if (!es || n <= 0 || !P || !q || !V || !rc) return TCV_ERR_INVALID;
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (es[i] == es[j]) { tcv::set_error("estimators_finish_frames: the same estimator twice"); return TCV_ERR_INVALID; }
    std::vector<std::string> msgs(n);
    for_each_estimator(n, [&](int i) {
        if (stats && es[i]) stats[i] = es[i]->stats;
        rc[i] = tcv_estimator_finish_frame(es[i], P + 3 * i, q + 4 * i, V + 3 * i);
        if (rc[i] != TCV_OK) msgs[i] = tcv_last_error();      // (the text is per thread)
    });
    for (int i = 0; i < n; i++) if (rc[i] != TCV_OK) { tcv::set_error(msgs[i]); return rc[i]; }
    return TCV_OK;
}
extern "C" int tcv_estimator_set_window_tap(tcv_estimator *e, int on) {
    if (!e) return TCV_ERR_INVALID;
    e->tap.on = on != 0;
    if (!on) e->tap = WindowTap();
    return TCV_OK;
}
extern "C" int tcv_estimator_get_window_snapshot(const tcv_estimator *e, tcv_window_snapshot *o) {
    if (!e || !o) return TCV_ERR_INVALID;
    const WindowTap &T = e->tap;
    if (!T.on || !T.have) { tcv::set_error("estimator_get_window_snapshot: no snapshot (tap off, or no window optimised since it was switched on)"); return TCV_ERR_INVALID; }
    std::memset(o, 0, sizeof *o);
    o->n_frames = W + 1; o->n_landmarks = (int)T.feat_in.size(); o->n_imu = (int)T.imu.size(); o->n_proj = (int)T.pi.size(); o->n_line = (int)T.lf.size();
    o->marg_flag = T.marg_flag; o->estimate_extrinsic = e->cfg.estimate_extrinsic; o->line_exact_jacobian = e->cfg.line_exact_jacobian;
    o->pose_in = T.pose_in.data(); o->speedbias_in = T.sb_in.data(); o->ex_pose_in = T.ex_in.data(); o->feature_in = T.feat_in.data();
    o->pose_out = T.pose_out.data(); o->speedbias_out = T.sb_out.data(); o->ex_pose_out = T.ex_out.data(); o->feature_out = T.feat_out.data();
    o->imu = T.imu.data(); o->imu_frame_i = T.imu_i.data(); o->imu_frame_j = T.imu_j.data();
    o->proj_frame_i = T.pi.data(); o->proj_frame_j = T.pj.data(); o->proj_feature = T.pl.data(); o->proj_pts = T.pts.data();
    o->line_frame = T.lf.data(); o->line_data = T.ld.data();
    std::memcpy(o->line_K, e->cfg.K, sizeof o->line_K); std::memcpy(o->line_Ric, T.Ric, sizeof o->line_Ric);
    for (int c = 0; c < 3; c++) { o->line_Tic[c] = T.ex_in[c]; o->gravity[c] = e->cfg.gravity[c]; }
    o->proj_sqrt_info = e->cfg.focal_length / 1.5;
    o->prior_m = T.prior_m; o->prior_n = T.prior_n; o->prior_nblk = (int)T.psize.size();
    o->prior_block_kind = T.pk.data(); o->prior_block_index = T.pidx.data(); o->prior_block_size = T.psize.data(); o->prior_block_idx = T.pcol.data();
    o->prior_x0 = T.x0.data(); o->prior_J0 = T.J0.data(); o->prior_r0 = T.r0.data();
    o->iterations = T.iterations; o->applied = T.applied; o->final_cost = T.final_cost;
    return TCV_OK;
}
extern "C" int tcv_estimator_get_stats(const tcv_estimator *e, tcv_estimator_stats *out) {
    if (!e || !out) return TCV_ERR_INVALID;
    *out = e->stats;
    return TCV_OK;
}
