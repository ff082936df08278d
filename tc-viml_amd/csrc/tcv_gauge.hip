// Gauge fix after the solve: Estimator::double2vector() (reference vins_estimator/src/estimator.cpp:1537-1581)
// followed by what the next vector2double() (:1492-1512) makes of its result.  Yaw of frame 0 and its position are
// unobservable in a visual-inertial window, so the reference rotates the whole window back about the vertical by
// the yaw drift of frame 0 and re-anchors it at the original Ps[0].  One thread per frame; in batch mode the
// states are rewritten in place in HBM so that the marginalisation kernel linearises at the gauge-fixed states
// exactly as the reference does (:1905 double2vector, :1915 vector2double).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "tcv_host.h"
#include "tcv_math.h"

namespace tcv {

#define TCV_PI 3.14159265358979323846   // M_PI

// Utility::R2ypr, utility.h:70-85 (degrees)
TCV_HD void r2ypr(const M3 &R, double ypr[3]) {
    const double n0 = R(0, 0), n1 = R(1, 0), n2 = R(2, 0), o0 = R(0, 1), o1 = R(1, 1), a0 = R(0, 2), a1 = R(1, 2);
    const double y = atan2(n1, n0);
    const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
    const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
    ypr[0] = y / TCV_PI * 180.0; ypr[1] = p / TCV_PI * 180.0; ypr[2] = r / TCV_PI * 180.0;
}
// Utility::ypr2R, utility.h:87-112 (degrees): Rz * Ry * Rx
TCV_HD M3 ypr2R(double yd, double pd, double rd) {
    const double y = yd / 180.0 * TCV_PI, p = pd / 180.0 * TCV_PI, r = rd / 180.0 * TCV_PI;
    M3 Rz = m3_zero(), Ry = m3_zero(), Rx = m3_zero();
    Rz(0, 0) = cos(y); Rz(0, 1) = -sin(y); Rz(1, 0) = sin(y); Rz(1, 1) = cos(y); Rz(2, 2) = 1.0;
    Ry(0, 0) = cos(p); Ry(0, 2) = sin(p); Ry(1, 1) = 1.0; Ry(2, 0) = -sin(p); Ry(2, 2) = cos(p);
    Rx(0, 0) = 1.0; Rx(1, 1) = cos(r); Rx(1, 2) = -sin(r); Rx(2, 1) = sin(r); Rx(2, 2) = cos(r);
    return (Rz * Ry) * Rx;
}
// `Quaterniond q{R}` of vector2double (estimator.cpp:1499): Eigen's rotation-matrix -> quaternion conversion
TCV_HD Quat r2q(const M3 &m) {
    double q[4];   // x y z w
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t; q[1] = (m(0, 2) - m(2, 0)) * t; q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
    }
    return Quat(q[0], q[1], q[2], q[3]);
}
// rot_diff of estimator.cpp:1548-1563
TCV_HD M3 gauge_rot_diff(const M3 &R0, const double *pose0) {
    double a[3], b[3];
    r2ypr(R0, a);
    const M3 R00 = to_matrix(Quat(pose0 + 3));
    r2ypr(R00, b);
    const double y_diff = a[0] - b[0];
    M3 rot = ypr2R(y_diff, 0.0, 0.0);
    if (fabs(fabs(a[1]) - 90.0) < 1.0 || fabs(fabs(b[1]) - 90.0) < 1.0) rot = R0 * transpose(R00);   // "euler singular point"
    return rot;
}
// one frame of the loop :1565-1581
TCV_HD void gauge_frame(const M3 &rot, const double *P0, const double *pose0, const double *pose_i, const double *vel_i, M3 &Rs, V3 &Ps, V3 &Vs) {
    Rs = rot * to_matrix(normalized(Quat(pose_i + 3)));
    Ps = rot * V3(pose_i[0] - pose0[0], pose_i[1] - pose0[1], pose_i[2] - pose0[2]) + V3(P0);
    Vs = vel_i ? rot * V3(vel_i) : V3();
}

__global__ void gauge_kernel(int n, const double *R0, const double *P0, const double *pose, const double *sb, double *Rs, double *Ps,
                             double *Vs, double *pose_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const M3 rot = gauge_rot_diff(m3_load(R0), pose);
    M3 R; V3 P, V;
    gauge_frame(rot, P0, pose, pose + 7 * i, sb + 9 * i, R, P, V);
    for (int k = 0; k < 9; k++) Rs[9 * i + k] = R.m[k];
    Ps[3 * i] = P.x; Ps[3 * i + 1] = P.y; Ps[3 * i + 2] = P.z;
    Vs[3 * i] = V.x; Vs[3 * i + 1] = V.y; Vs[3 * i + 2] = V.z;
    if (pose_out) {
        const Quat q = r2q(R);
        double *o = pose_out + 7 * i;
        o[0] = P.x; o[1] = P.y; o[2] = P.z; o[3] = q.x; o[4] = q.y; o[5] = q.z; o[6] = q.w;
    }
}

// batch: one thread per (window, frame); every thread recomputes rot_diff of its window from the un-fixed pose 0, which
// is only overwritten after all frames of the window have read it (one workgroup per window, barrier in between)
__global__ void __launch_bounds__(64) gauge_batch_kernel(const WinHdr *win, const PlanHdr *plans, const long long *plan_base, const int *ipool,
                                                         const double *dpool, double *state, int state_stride) {
    const int w = blockIdx.x, i = threadIdx.x;
    const WinHdr &W = win[w];
    const PlanHdr &P = plans[W.plan];
    const int *ft = ipool + plan_base[W.plan] + P.o_frames;
    double *x = state + (size_t)w * state_stride;
    const double *x_init = dpool + W.dbase + W.d_x;
    const bool act = i < P.n_frames && P.n_frames > 0 && ft[0] >= 0 && ft[2 * i] >= 0;
    M3 R; V3 Pn, V;
    int go = 0, so = -1;
    if (act) {
        const int g0 = ft[0];
        go = ft[2 * i]; so = ft[2 * i + 1];
        const M3 R0 = to_matrix(Quat(x_init + g0 + 3));      // Rs[0] before the solve: what vector2double() turned into para_Pose[0]
        const M3 rot = gauge_rot_diff(R0, x + g0);
        gauge_frame(rot, x_init + g0, x + g0, x + go, so >= 0 ? x + so : nullptr, R, Pn, V);
    }
    __syncthreads();
    if (act) {
        const Quat q = r2q(R);
        x[go] = Pn.x; x[go + 1] = Pn.y; x[go + 2] = Pn.z; x[go + 3] = q.x; x[go + 4] = q.y; x[go + 5] = q.z; x[go + 6] = q.w;
        if (so >= 0) { x[so] = V.x; x[so + 1] = V.y; x[so + 2] = V.z; }
    }
}

}  // namespace tcv
using namespace tcv;

extern "C" int tcv_gauge_fix(int n, const double *R0, const double *P0, const double *pose, const double *sb, double *Rs, double *Ps,
                             double *Vs, double *pose_out) {
    if (n <= 0 || n > 4096 || !R0 || !P0 || !pose || !sb || !Rs || !Ps || !Vs) { set_error("gauge_fix: bad argument"); return TCV_ERR_INVALID; }
    if (int rc = device_ready()) return rc;
    for (int i = 0; i < 7 * n; i++) if (!(pose[i] == pose[i])) { set_error("gauge_fix: NaN in poses"); return TCV_ERR_NUMERIC; }
    const size_t nin = 12 + (size_t)16 * n, nout = (size_t)22 * n;
    std::vector<double> h(nin);
    std::memcpy(h.data(), R0, 72); std::memcpy(h.data() + 9, P0, 24);
    std::memcpy(h.data() + 12, pose, sizeof(double) * 7 * n); std::memcpy(h.data() + 12 + 7 * n, sb, sizeof(double) * 9 * n);
    double *d = nullptr;
    hipError_t e = tcv::dev_malloc((void **)&d, sizeof(double) * (nin + nout));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    int rc = TCV_OK;
    e = hipMemcpy(d, h.data(), sizeof(double) * nin, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        double *o = d + nin;
        hipStream_t st = tcv::util_stream();
        hipLaunchKernelGGL(gauge_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, d, d + 9, d + 12, d + 12 + 7 * n, o, o + 9 * n, o + 12 * n,
                           o + 15 * n);
        e = hipGetLastError();
        if (e == hipSuccess) e = st ? hipStreamSynchronize(st) : hipDeviceSynchronize();
        std::vector<double> ho(nout);
        if (e == hipSuccess) e = hipMemcpy(ho.data(), o, sizeof(double) * nout, hipMemcpyDeviceToHost);
        if (e == hipSuccess) {
            std::memcpy(Rs, ho.data(), sizeof(double) * 9 * n); std::memcpy(Ps, ho.data() + 9 * n, sizeof(double) * 3 * n);
            std::memcpy(Vs, ho.data() + 12 * n, sizeof(double) * 3 * n);
            if (pose_out) std::memcpy(pose_out, ho.data() + 15 * n, sizeof(double) * 7 * n);
        }
    }
    if (e != hipSuccess) rc = hip_fail(e, "gauge_fix");
    tcv::dev_free(d);
    return rc;
}

extern "C" int tcv_batch_gauge_fix(tcv_batch *b, void *hip_stream) {
    if (!b || !b->solved) { set_error("batch_gauge_fix: batch has not been solved"); return TCV_ERR_INVALID; }
    if (hip_stream == TCV_STREAM_THREAD) hip_stream = (void *)tcv::util_stream();
    for (auto &H : b->plans)
        if (H.n_frames <= 0 || H.n_frames > 64) { set_error("batch_gauge_fix: problem carries no frame table (tcv_problem_set_frames)"); return TCV_ERR_INVALID; }
    if (int rc = tcv_batch_enter_stream(b, hip_stream)) return rc;
    hipLaunchKernelGGL(gauge_batch_kernel, dim3(b->n), dim3(64), 0, (hipStream_t)hip_stream, b->d_win, b->d_plans, b->d_plan_base, b->d_ipool,
                       b->d_dpool, b->d_state, b->state_stride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gauge kernel launch");
    b->gauge_fixed = true;
    return TCV_OK;
}
