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
#include "tcv_gauge.h"

namespace tcv {

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
    if (b->gauge_in_solve) return TCV_OK;      // the last solve applied it in its epilogue (tcv_batch_set_fused_gauge_fix): same states, same bits
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

// The gauge fix in the solve kernel's epilogue (the states are in LDS there: one launch boundary and its gap less per frame): from the next
// tcv_batch_solve on, and tcv_batch_gauge_fix becomes a no-op behind such a solve.  Needs the frame tables (tcv_problem_set_frames /
// tcv_problem_from_window) like tcv_batch_gauge_fix does.
extern "C" int tcv_batch_set_fused_gauge_fix(tcv_batch *b, int on) {
    if (!b) return TCV_ERR_INVALID;
    if (on)
        for (auto &H : b->plans)
            if (H.n_frames <= 0 || H.n_frames > 64) { set_error("batch_set_fused_gauge_fix: problem carries no frame table (tcv_problem_set_frames)"); return TCV_ERR_INVALID; }
    b->fuse_gauge = on != 0;
    return TCV_OK;
}
