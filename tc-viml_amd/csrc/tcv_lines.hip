// 2D-3D line association: the step that feeds the line factors.
//   Estimator::UpdateLinesInFoV           estimator.cpp:385-447   which prior map lines can be seen from a frame
//   Estimator::LineCorrespondenceInFrame  estimator.cpp:671-885   nearest projected map line for one detected 2D line
//   Estimator::CalAngleDist / CalEulerDist  :602-669,  Line2D  feature_manager.cpp:4-73
// One wavefront per detected line scans the map (an HBM/L2-bound read of 48 B per map line), every lane scoring a strided
// subset exactly like the reference's loop body; the best (smallest distance, first index on ties, as the reference's
// strict `<` in map order) is reduced across the lanes.  The reference mixes float and double here (`float xx, yy`,
// `float min_dist`, `Eigen::Vector3f error`); the float roundings are reproduced.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "tcv_host.h"
#include "tcv_packed.h"      // parallel_run, host_threads
#include "tcv_math.h"

namespace tcv {

struct L2 {      // Line2D(const Eigen::Vector4d &), feature_manager.cpp:4-16
    double sx, sy, ex, ey, len, dx, dy, A, B, C, A2B2;
};
TCV_HD L2 make_l2(double sx, double sy, double ex, double ey) {
    L2 l;
    l.sx = sx; l.sy = sy; l.ex = ex; l.ey = ey;
    const double vx = ex - sx, vy = ey - sy;
    l.len = sqrt(vx * vx + vy * vy);
    l.dx = vx / l.len; l.dy = vy / l.len;
    l.A = ey - sy; l.B = sx - ex; l.C = ex * sy - sx * ey;
    l.A2B2 = sqrt(l.A * l.A + l.B * l.B);
    return l;
}
// Line2D::Point2Flined, feature_manager.cpp:48-73
TCV_HD void point2flined(const L2 &l, double px, double py, double &ox, double &oy) {
    const double tsx = px - l.sx, tsy = py - l.sy, d1 = sqrt(tsx * tsx + tsy * tsy);
    const double tex = px - l.ex, tey = py - l.ey, d2 = sqrt(tex * tex + tey * tey);
    const double A_ = l.B, B_ = -l.A;
    const double C_ = -1 * (A_ * px + B_ * py);
    const double det = l.A * B_ - l.B * A_;
    const double invdet = 1.0 / det;
    const double ix = (B_ * invdet) * (-l.C) + (-l.B * invdet) * (-C_);
    const double iy = (-A_ * invdet) * (-l.C) + (l.A * invdet) * (-C_);
    if ((ix - l.sx) * (ix - l.ex) >= 0) {
        if (d1 < d2) { ox = l.sx; oy = l.sy; } else { ox = l.ex; oy = l.ey; }
    } else { ox = ix; oy = iy; }
}
// CalEulerDist, estimator.cpp:615-669
TCV_HD void euler_dist(const L2 &proj, const L2 &det, double &dist, double &overlap) {
    const int sampleNum = 10;
    const bool det_short = det.len <= proj.len;
    const L2 &l1 = det_short ? det : proj, &l2 = det_short ? proj : det;
    double ax, ay, bx, by;
    point2flined(l2, l1.sx, l1.sy, ax, ay);
    point2flined(l2, l1.ex, l1.ey, bx, by);
    overlap = sqrt((ax - bx) * (ax - bx) + (ay - by) * (ay - by)) / l2.len;
    const double step_x = (l1.sx - l1.ex) / sampleNum, step_y = (l1.sy - l1.ey) / sampleNum;
    double d = 0.0;
    for (int i = 0; i < sampleNum; ++i) {
        const double x = l1.sx + i * step_x, y = l1.sy + i * step_y;
        d = d + fabs(l2.A * x + l2.B * y + l2.C) / l2.A2B2;
    }
    d = d + 1 * fabs(l2.A * l1.sx + l2.B * l1.sy + l2.C) / l2.A2B2;
    d = d + 1 * fabs(l2.A * l1.ex + l2.B * l1.ey + l2.C) / l2.A2B2;
    d = d / (sampleNum + 2);
    if (d != d || overlap != overlap) { d = 10000.0; overlap = 0.0; }
    dist = d;
}

struct LineCam { M3 R; V3 T; };
// R = Ric^T Rbi^T Rbw, T = Ric^T (Rbi^T (Tbw - Tbi) - Tic)   (estimator.cpp:388-402, :679-692)
TCV_HD LineCam line_cam(const double *pose, const double *ex, const double *Rbw, const double *Tbw) {
    const M3 RicT = transpose(to_matrix(normalized(Quat(ex + 3)))), RbiT = transpose(to_matrix(normalized(Quat(pose + 3))));
    LineCam c;
    c.R = (RicT * RbiT) * m3_load(Rbw);
    c.T = RicT * (RbiT * (V3(Tbw) - V3(pose)) - V3(ex));
    return c;
}

struct LineArgs {
    const double *poses, *ex, *Rbw, *Tbw, *K, *map, *det;
    const int *det_frame;
    int n_frames, n_map, n_det, width, height, window_size;
    int only_frame;             // >= 0: the FoV kernel computes this frame's row only (the other rows are given)
    double angle_th, overlap_th;
    unsigned char *in_fov;      // n_frames x n_map
    int *match;                 // n_det
    float *err;                 // n_det x 3
    double *proj;               // n_det x 4
};

// The calls of a batch share two launches (FoV rows, then the matching): a table of the calls' arguments and the first block of every call
// (n + 1 entries, ascending) sit in the uploaded blob; a block looks its call up.  Sixteen tiny launches back to back on one stream cost a
// lock-step frame of eight estimators 0.58 ms, two cost 0.1.
__device__ __forceinline__ int call_of_block(const int *first, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (first[mid] <= blk) lo = mid; else hi = mid - 1; }
    return lo;
}

// UpdateLinesInFoV: one thread per (frame, map line)
__global__ void lines_fov_kernel(const LineArgs *tab, const int *first, int ncall) {
    const int cidx = call_of_block(first, ncall, (int)blockIdx.x);
    const LineArgs A = tab[cidx];
    int i = ((int)blockIdx.x - first[cidx]) * blockDim.x + threadIdx.x;
    if (i >= (A.only_frame >= 0 ? A.n_map : A.n_frames * A.n_map)) return;
    if (A.only_frame >= 0) i += A.only_frame * A.n_map;
    const int f = i / A.n_map, j = i - f * A.n_map;
    const LineCam c = line_cam(A.poses + 7 * f, A.ex, A.Rbw, A.Tbw);
    const double *l = A.map + 6 * (size_t)j;
    const V3 ps = c.R * V3(l) + c.T, pe = c.R * V3(l + 3) + c.T;
    const int hu = -2 * A.window_size, hd = 2 * A.window_size + A.height, wl = -2 * A.window_size, wr = 2 * A.window_size + A.width;
    bool s = false, e = false;
    if (ps.z > 0 && pe.z > 0) {
        const double xx = A.K[0] * ps.x / ps.z + A.K[2], yy = A.K[4] * ps.y / ps.z + A.K[5];
        const double xx_ = A.K[0] * pe.x / pe.z + A.K[2], yy_ = A.K[4] * pe.y / pe.z + A.K[5];
        s = xx > wl && xx < (wr - 1) && yy > hu && yy < hd;
        e = xx_ > wl && xx_ < (wr - 1) && yy_ > hu && yy_ < hd;
    }
    A.in_fov[i] = (s || e) ? 1 : 0;
}

// LineCorrespondenceInFrame: four wavefronts per detected line (a map of ~900 segments: 3 - 4 per lane instead of 14 -- the association is
// a host round trip of every frame, and its kernel was 60 us alone, 270 us beside the solves of a 128-stream replay).  Each lane keeps the first
// minimum of its own ascending walk; lanes, then wavefronts, are merged by (smallest distance, then smallest map index) = the first hit of the
// reference's sequential scan, whatever the partition.
enum { MATCH_NT = 256 };
__global__ void __launch_bounds__(MATCH_NT) lines_match_kernel(const LineArgs *tab, const int *first, int ncall) {
    const int cidx = call_of_block(first, ncall, (int)blockIdx.x);
    const LineArgs A = tab[cidx];
    const int q = (int)blockIdx.x - first[cidx], lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ float s_best[MATCH_NT / 64], s_angle[MATCH_NT / 64], s_ov[MATCH_NT / 64];
    __shared__ int s_j[MATCH_NT / 64];
    __shared__ double s_p[MATCH_NT / 64][4];
    const int f = A.det_frame[q];
    const double *dv = A.det + 4 * (size_t)q;
    const L2 det = make_l2(dv[0], dv[1], dv[2], dv[3]);
    const LineCam c = line_cam(A.poses + 7 * f, A.ex, A.Rbw, A.Tbw);
    const double fx = A.K[0], cx = A.K[2], fy = A.K[4], cy = A.K[5];
    const int W = A.width, H = A.height;
    float best = 10000.0f, b_angle = -1.0f, b_ov = -1.0f;
    int best_j = 0x7fffffff;
    double bp[4] = {dv[0], dv[1], dv[2], dv[3]};
    int any_fov = 0;
    for (int j = (int)threadIdx.x; j < A.n_map; j += MATCH_NT) {
        if (!A.in_fov[(size_t)f * A.n_map + j]) continue;
        any_fov = 1;
        const double *l = A.map + 6 * (size_t)j;
        const V3 ps = c.R * V3(l) + c.T, pe = c.R * V3(l + 3) + c.T;
        bool sflag = false, eflag = false;
        float xx = 0, yy = 0, xx_ = 0, yy_ = 0;
        if (ps.z > 0 && pe.z > 0) {
            xx = (float)(fx * ps.x / ps.z + cx); yy = (float)(fy * ps.y / ps.z + cy);
            xx_ = (float)(fx * pe.x / pe.z + cx); yy_ = (float)(fy * pe.y / pe.z + cy);
            sflag = xx > 0 && xx < W - 1 && yy > 0 && yy < H - 1;
            eflag = xx_ > 0 && xx_ < W - 1 && yy_ > 0 && yy_ < H - 1;
        }
        double cand[4];
        bool have = false;
        if (sflag && eflag) { cand[0] = xx; cand[1] = yy; cand[2] = xx_; cand[3] = yy_; have = true; }
        else if (sflag || eflag) {      // walk the hidden end point back towards the visible one: t = 0.9, 0.8, ... (:765-791, :813-839)
            const V3 a = sflag ? ps : pe, b = sflag ? pe : ps;
            const V3 dir = b - a;
            double t = 0.9, x = 0.0, y = 0.0;
            bool found = false;
            while (t > 0) {
                const V3 p = a + t * dir;
                if (p.z > 0) {
                    x = fx * p.x / p.z + cx; y = fy * p.y / p.z + cy;
                    if (x > 0 && x < (W - 1) && y > 0 && y < (H - 1)) { found = true; break; }
                }
                t = t - 0.1;
            }
            if (found) {
                if (sflag) { cand[0] = xx; cand[1] = yy; cand[2] = x; cand[3] = y; }
                else { cand[0] = x; cand[1] = y; cand[2] = xx_; cand[3] = yy_; }
                have = true;
            }
        }
        if (!have) continue;
        const L2 tl = make_l2(cand[0], cand[1], cand[2], cand[3]);
        double angle = acos(fabs(det.dx * tl.dx + det.dy * tl.dy));      // CalAngleDist
        if (angle != angle) angle = 3.14159265358979323846;
        if (angle > A.angle_th) continue;
        double dist, ov;
        euler_dist(tl, det, dist, ov);
        const float distance = (float)dist, overlap = (float)ov;
        if ((double)overlap < A.overlap_th) continue;
        if (distance < best) {      // lanes walk their lines in ascending j, so the first minimum of the lane is kept
            best = distance; best_j = j; b_angle = (float)angle; b_ov = overlap;
            bp[0] = cand[0]; bp[1] = cand[1]; bp[2] = cand[2]; bp[3] = cand[3];
        }
    }
    // wave reduction: smallest distance, then smallest map index (= first hit of the reference's sequential scan)
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oj = __shfl_xor(best_j, o);
        const float oa = __shfl_xor(b_angle, o), oo = __shfl_xor(b_ov, o);
        double op[4];
#pragma unroll
        for (int k = 0; k < 4; k++) op[k] = __shfl_xor(bp[k], o);
        any_fov |= __shfl_xor(any_fov, o);
        if (oj != 0x7fffffff && (best_j == 0x7fffffff || ob < best || (ob == best && oj < best_j))) {
            best = ob; best_j = oj; b_angle = oa; b_ov = oo;
#pragma unroll
            for (int k = 0; k < 4; k++) bp[k] = op[k];
        }
    }
    if (lane == 0) { s_best[wave] = best; s_j[wave] = best_j; s_angle[wave] = b_angle; s_ov[wave] = b_ov; for (int k = 0; k < 4; k++) s_p[wave][k] = bp[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < MATCH_NT / 64; w++) {
            const float ob = s_best[w]; const int oj = s_j[w];
            if (oj != 0x7fffffff && (best_j == 0x7fffffff || ob < best || (ob == best && oj < best_j))) {
                best = ob; best_j = oj; b_angle = s_angle[w]; b_ov = s_ov[w];
                for (int k = 0; k < 4; k++) bp[k] = s_p[w][k];
            }
        }
        const bool hit = best_j != 0x7fffffff;
        A.match[q] = hit ? best_j : -1;
        A.err[3 * q] = hit ? b_angle : -1.0f; A.err[3 * q + 1] = hit ? best : -1.0f; A.err[3 * q + 2] = hit ? b_ov : -1.0f;
        for (int k = 0; k < 4; k++) A.proj[4 * (size_t)q + k] = hit ? bp[k] : dv[k];
        (void)any_fov;
    }
}

}  // namespace tcv
using namespace tcv;

// n independent association problems in one device round trip: one pinned staging buffer and one device blob
//   [inputs of every call: poses, extrinsic, Rbw, Tbw, K, map, detections (doubles) | frame of every detection (ints)]
//   [FoV rows of every call (bytes): given rows go up, computed rows come back]
//   [outputs of every call: projected segments 4 n_det (doubles) | match index n_det (ints) | errA, errD, overlap 3 n_det (floats)]
// one copy in ([inputs | FoV]), the kernels of all calls back to back on the calling thread's own stream, one copy out ([FoV | outputs]), one wait
struct tcv_line_map { double *d = nullptr; int n = 0, dev = 0; };
extern "C" int tcv_line_map_create(tcv_line_map **out, int n_map, const double *lines3d) {
    if (!out || n_map <= 0 || !lines3d) { set_error("line_map_create: bad argument"); return TCV_ERR_INVALID; }
    if (int rc = device_ready()) return rc;
    tcv_line_map *m = new tcv_line_map();
    m->n = n_map;
    (void)hipGetDevice(&m->dev);
    hipError_t e = tcv::dev_malloc((void **)&m->d, sizeof(double) * 6 * (size_t)n_map);
    if (e == hipSuccess) e = hipMemcpy(m->d, lines3d, sizeof(double) * 6 * (size_t)n_map, hipMemcpyHostToDevice);
    if (e != hipSuccess) { tcv::dev_free(m->d); delete m; return hip_fail(e, "line_map_create"); }
    *out = m;
    return TCV_OK;
}
extern "C" int tcv_line_map_device(const tcv_line_map *m) { return m ? m->dev : -1; }      // (library-internal: tcv_estimator.cpp)
extern "C" void tcv_line_map_destroy(tcv_line_map *m) {
    if (!m) return;
    // An association that reads the map may still be in flight -- on the stream of the thread that launched it.  Every tcv_match_lines_batch
    // call waits for its own results before it returns (the matches are a host round trip), so by the time a caller can destroy the map no
    // launch of ITS threads reads it any more; the calling thread's own stream is drained below.  No device-wide synchronisation: it stalled every host thread's stream of a multi-threaded replay on each estimator
    // teardown (round-5 advisor finding).
    { int cur = -1; if (hipGetDevice(&cur) == hipSuccess && cur == m->dev) { hipStream_t st = tcv::util_stream(); if (st) (void)hipStreamSynchronize(st); } }
    tcv::dev_free(m->d);
    delete m;
}

extern "C" int tcv_match_lines_batch(int n, const tcv_match_lines_args *args) {
    if (n < 0 || (n > 0 && !args)) { set_error("match_lines_batch: bad argument"); return TCV_ERR_INVALID; }
    if (n == 0) return TCV_OK;
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    struct Lay { size_t o_in, o_det, o_fov, o_out, o_match, o_err, tot; };
    std::vector<Lay> L(n);
    size_t in_total = 0, fov_total = 0, out_total = 0;
    for (int c = 0; c < n; c++) {
        const tcv_match_lines_args &a = args[c];
        if (a.n_frames <= 0 || a.n_map <= 0 || a.n_det < 0 || !a.poses || !a.ex_pose || !a.Rbw || !a.Tbw || !a.K || !a.lines3d || (a.n_det > 0 && (!a.det_frame || !a.det_lines))) {
            set_error("match_lines: bad argument"); return TCV_ERR_INVALID;
        }
        if (a.fov_given && !a.in_fov) { set_error("match_lines: fov_given needs in_fov"); return TCV_ERR_INVALID; }
        if (a.fov_given >= 2 && a.fov_given - 2 >= a.n_frames) { set_error("match_lines: fov_given names a frame outside the window"); return TCV_ERR_INVALID; }
        for (int i = 0; i < a.n_det; i++) if (a.det_frame[i] < 0 || a.det_frame[i] >= a.n_frames) { set_error("match_lines: frame index out of range"); return TCV_ERR_INVALID; }
        if (a.map_device && a.map_device->n != a.n_map) { set_error("match_lines: map_device holds another number of lines"); return TCV_ERR_INVALID; }
        const size_t nd_in = (size_t)7 * a.n_frames + 7 + 9 + 3 + 9 + (a.map_device ? 0 : (size_t)6 * a.n_map) + (size_t)4 * a.n_det;
        L[c].tot = (size_t)a.n_frames * a.n_map;
        L[c].o_in = in_total; L[c].o_det = in_total + up16(sizeof(double) * nd_in);
        in_total = up16(L[c].o_det + sizeof(int) * std::max(1, a.n_det));
        L[c].o_fov = fov_total; fov_total += up16(L[c].tot);
        L[c].o_out = out_total; L[c].o_match = out_total + up16(sizeof(double) * 4 * std::max(1, a.n_det));
        L[c].o_err = up16(L[c].o_match + sizeof(int) * std::max(1, a.n_det));
        out_total = up16(L[c].o_err + sizeof(float) * 3 * std::max(1, a.n_det));
    }
    // [.. inputs | argument table (LineArgs per call) | first FoV block per call (n + 1) | first matching block per call (n + 1)]
    const size_t o_tab = in_total, o_ff = up16(o_tab + sizeof(LineArgs) * (size_t)n), o_mf = up16(o_ff + sizeof(int) * (size_t)(n + 1));
    in_total = up16(o_mf + sizeof(int) * (size_t)(n + 1));
    if (int rc = device_ready()) return rc;
    const size_t total = in_total + fov_total + out_total;
    char *h = (char *)tcv::host_staging_acquire(total);
    if (!h) { set_error("hipHostMalloc (staging) failed"); return TCV_ERR_HIP; }
    char *dv = nullptr;
    hipError_t e = tcv::dev_malloc((void **)&dv, total);
    hipStream_t st = tcv::util_stream();
    bool in_flight = false;
    std::vector<size_t> o_pose(n), o_ex(n), o_R(n), o_T(n), o_K(n), o_map(n), o_dl(n);
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    for (int c = 0; c < n && e == hipSuccess; c++) if (args[c].map_device && args[c].map_device->dev != cur_dev) { tcv::host_staging_release(h); tcv::dev_free(dv); set_error("match_lines: map_device lives on another device"); return TCV_ERR_INVALID; }
    if (e == hipSuccess) {
        // the staging buffer is filled by the worker threads (the calls' slices are disjoint): 64 calls with their maps are 3.4 MB of memcpy
        auto fill = [&](int c) {
            const tcv_match_lines_args &a = args[c];
            double *hd = (double *)(h + L[c].o_in);
            size_t o = 0;
            auto put = [&](const double *p, size_t k) { std::memcpy(hd + o, p, sizeof(double) * k); o += k; return L[c].o_in / sizeof(double) + o - k; };
            o_pose[c] = put(a.poses, (size_t)7 * a.n_frames); o_ex[c] = put(a.ex_pose, 7); o_R[c] = put(a.Rbw, 9); o_T[c] = put(a.Tbw, 3); o_K[c] = put(a.K, 9);
            o_map[c] = a.map_device ? 0 : put(a.lines3d, (size_t)6 * a.n_map);
            o_dl[c] = a.n_det ? put(a.det_lines, (size_t)4 * a.n_det) : L[c].o_in / sizeof(double) + o;
            if (a.n_det) std::memcpy(h + L[c].o_det, a.det_frame, sizeof(int) * a.n_det);
            if (a.fov_given) std::memcpy(h + in_total + L[c].o_fov, a.in_fov, L[c].tot);
            else std::memset(h + in_total + L[c].o_fov, 0, L[c].tot);
        };
        const int nth = n >= 8 ? tcv::host_threads(std::min(n / 4, 8)) : 1;
        if (nth > 1) tcv::parallel_run(nth, [&](int t) { for (int c = t; c < n; c += nth) fill(c); });
        else for (int c = 0; c < n; c++) fill(c);
    }
    int rc = TCV_OK;
    int fov_blocks = 0, det_blocks = 0;
    if (e == hipSuccess) {
        LineArgs *tab = (LineArgs *)(h + o_tab);
        int *ff = (int *)(h + o_ff), *mf = (int *)(h + o_mf);
        for (int c = 0; c < n; c++) {
            const tcv_match_lines_args &a = args[c];
            double *dd = (double *)dv;
            char *dout = dv + in_total + fov_total + L[c].o_out;
            const int fov_frame = a.fov_given >= 2 ? a.fov_given - 2 : -1;      // this frame's row is computed here, the others are given
            LineArgs A;
            A.poses = dd + o_pose[c]; A.ex = dd + o_ex[c]; A.Rbw = dd + o_R[c]; A.Tbw = dd + o_T[c]; A.K = dd + o_K[c]; A.map = a.map_device ? a.map_device->d : dd + o_map[c]; A.det = dd + o_dl[c];
            A.det_frame = (int *)(dv + L[c].o_det); A.n_frames = a.n_frames; A.n_map = a.n_map; A.n_det = a.n_det; A.width = a.width; A.height = a.height; A.window_size = a.window_size;
            A.angle_th = a.angle_th; A.overlap_th = a.overlap_th; A.in_fov = (unsigned char *)(dv + in_total + L[c].o_fov);
            A.match = (int *)(dv + in_total + fov_total + L[c].o_match); A.err = (float *)(dv + in_total + fov_total + L[c].o_err);
            A.proj = (double *)dout;
            A.only_frame = fov_frame;
            tab[c] = A;
            ff[c] = fov_blocks; mf[c] = det_blocks;
            if (!a.fov_given) fov_blocks += ((int)L[c].tot + 255) / 256;
            else if (fov_frame >= 0) fov_blocks += (a.n_map + 255) / 256;
            det_blocks += a.n_det;
        }
        ff[n] = fov_blocks; mf[n] = det_blocks;
        in_flight = true;
        e = hipMemcpyAsync(dv, h, in_total + fov_total, hipMemcpyHostToDevice, st);
    }
    if (e == hipSuccess) {
        // (a call without blocks has first[c] == first[c + 1]: the search returns the LAST call whose first block is <= the block, i.e. the one that owns it)
        if (fov_blocks > 0) hipLaunchKernelGGL(lines_fov_kernel, dim3(fov_blocks), dim3(256), 0, st, (const LineArgs *)(dv + o_tab), (const int *)(dv + o_ff), n);
        if (det_blocks > 0) hipLaunchKernelGGL(lines_match_kernel, dim3(det_blocks), dim3(MATCH_NT), 0, st, (const LineArgs *)(dv + o_tab), (const int *)(dv + o_mf), n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h + in_total, dv + in_total, fov_total + out_total, hipMemcpyDeviceToHost, st);
    if (in_flight) { const hipError_t es = st ? hipStreamSynchronize(st) : hipDeviceSynchronize(); if (e == hipSuccess) e = es; }
    if (e == hipSuccess)
        for (int c = 0; c < n; c++) {
            const tcv_match_lines_args &a = args[c];
            const int fov_frame = a.fov_given >= 2 ? a.fov_given - 2 : -1;
            const char *hf = h + in_total + L[c].o_fov, *ho = h + in_total + fov_total;
            if (a.in_fov && !a.fov_given) std::memcpy(a.in_fov, hf, L[c].tot);
            else if (fov_frame >= 0) std::memcpy(a.in_fov + (size_t)fov_frame * a.n_map, hf + (size_t)fov_frame * a.n_map, (size_t)a.n_map);
            if (a.n_det && a.match_index) std::memcpy(a.match_index, ho + L[c].o_match, sizeof(int) * a.n_det);
            if (a.n_det && a.err) std::memcpy(a.err, ho + L[c].o_err, sizeof(float) * 3 * a.n_det);
            if (a.n_det && a.projected) std::memcpy(a.projected, ho + L[c].o_out, sizeof(double) * 4 * a.n_det);
        }
    if (e != hipSuccess) rc = hip_fail(e, "match_lines");
    tcv::host_staging_release(h);
    tcv::dev_free(dv);
    return rc;
}

extern "C" int tcv_match_lines(int n_frames, const double *poses, const double *ex_pose, const double *Rbw, const double *Tbw, const double *K,
                               int width, int height, int window_size, int n_map, const double *lines3d, int n_det, const int *det_frame,
                               const double *det_lines, double angle_th, double overlap_th, int fov_given, unsigned char *in_fov, int *match_index,
                               float *err, double *projected) {
    tcv_match_lines_args a;
    a.n_frames = n_frames; a.poses = poses; a.ex_pose = ex_pose; a.Rbw = Rbw; a.Tbw = Tbw; a.K = K; a.width = width; a.height = height; a.window_size = window_size;
    a.n_map = n_map; a.lines3d = lines3d; a.n_det = n_det; a.det_frame = det_frame; a.det_lines = det_lines; a.angle_th = angle_th; a.overlap_th = overlap_th;
    a.fov_given = fov_given; a.in_fov = in_fov; a.match_index = match_index; a.err = err; a.projected = projected; a.map_device = nullptr;
    return tcv_match_lines_batch(1, &a);
}
