// Batched IMU pre-integration on the GPU: IntegrationBase(acc_0, gyr_0, ba, bg) followed by push_back(dt, acc, gyr)
// for every buffered sample (reference vins_estimator/src/factor/integration_base.h:13-158: constructor :13-28,
// push_back :30-36, propagate :130-158, midPointIntegration :54-128).  repropagate (:38-52) is the same computation
// with new linearisation biases, so it maps onto the same entry point.
//
// One three-wavefront workgroup per pre-integration.  Per sample: the state wave builds the five 3x3 base matrices every block
// of F (15x15) and V (15x18) is a multiple of (R0 = R(delta_q), R1 = R(result_delta_q), M0 = R0 [a0]x, M1 = R1 [a1]x,
// M1B = M1 (I - [w]x dt)) and advances delta_p / delta_q / delta_v (mid-point rule); the 128 lanes of the two matrix waves own
// the entries of jacobian <- F jacobian and covariance <- F covariance F' + V noise V' (noise is diagonal, :21-27).  preint_kernel below.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tcv_factors.h"
#include "tcv_host.h"
#include "tcv_dev.h"

namespace tcv {

struct PreintArgs {
    const int *first, *count;
    const double *samples;   // rows of 7: dt, acc xyz, gyr xyz
    const double *init;      // n x 12: acc_0, gyr_0, linearized_ba, linearized_bg
    double noise[4];         // ACC_N, GYR_N, ACC_W, GYR_W
    double *out;             // n x 466: delta_p 3, delta_q 4 (xyzw), delta_v 3, ba 3, bg 3, sum_dt, jacobian 225, covariance 225 (row-major)
    int n;
};
enum { PREINT_OUT = 467 };

// entry (r, c) of F from the base matrices B[5][9] = R0, R1, M0, M1, M1B  (integration_base.h:90-103)
__device__ __forceinline__ double F_entry(const lds_d *B, int r, int c, double dt) {
    const int br = r / 3, bc = c / 3, i = r - 3 * br, j = c - 3 * bc, e = 3 * i + j;
    const double id = (i == j) ? 1.0 : 0.0;
    if (br == 0) {
        if (bc == 0) return id;
        if (bc == 1) return -0.25 * B[18 + e] * dt * dt + -0.25 * B[36 + e] * dt * dt;
        if (bc == 2) return id * dt;
        if (bc == 3) return -0.25 * (B[e] + B[9 + e]) * dt * dt;
        return -0.25 * B[27 + e] * dt * dt * -dt;
    }
    if (br == 1) {
        if (bc == 1) return B[45 + e];                 // I - [w]x dt
        if (bc == 4) return -1.0 * id * dt;
        return 0.0;
    }
    if (br == 2) {
        if (bc == 1) return -0.5 * B[18 + e] * dt + -0.5 * B[36 + e] * dt;
        if (bc == 2) return id;
        if (bc == 3) return -0.5 * (B[e] + B[9 + e]) * dt;
        if (bc == 4) return -0.5 * B[27 + e] * dt * -dt;
        return 0.0;
    }
    return (br == bc) ? id : 0.0;
}
// entry (r, k) of V, k < 18  (integration_base.h:106-118)
__device__ __forceinline__ double V_entry(const lds_d *B, int r, int k, double dt) {
    const int br = r / 3, bk = k / 3, i = r - 3 * br, j = k - 3 * bk, e = 3 * i + j;
    const double id = (i == j) ? 1.0 : 0.0;
    if (br == 0) {
        if (bk == 0) return 0.25 * B[e] * dt * dt;
        if (bk == 1 || bk == 3) return 0.25 * -B[27 + e] * dt * dt * 0.5 * dt;
        if (bk == 2) return 0.25 * B[9 + e] * dt * dt;
        return 0.0;
    }
    if (br == 1) return (bk == 1 || bk == 3) ? 0.5 * id * dt : 0.0;
    if (br == 2) {
        if (bk == 0) return 0.5 * B[e] * dt;
        if (bk == 1 || bk == 3) return 0.5 * -B[27 + e] * dt * 0.5 * dt;
        if (bk == 2) return 0.5 * B[9 + e] * dt;
        return 0.0;
    }
    if (br == 3) return (bk == 4) ? id * dt : 0.0;
    return (bk == 5) ? id * dt : 0.0;
}

// Round 6: the per-sample chain re-cut.  Until round 5 a sample was five workgroup barriers of a 256-thread workgroup in which 45 lanes, then
// 225, then ONE (the state), then 225 had work: ~5 us per sample on the device, 303 us per frame of a 128-stream replay
// (profiles/r05_replay128_timeline.txt).  Now three wavefronts with roles and TWO barriers per sample:
//   state wave    (wave 2, one lane of it) runs one sample ahead: mid-point update of delta_p / delta_q / delta_v, then the base matrices of the
//                 NEXT sample into the other half of a double buffer;
//   matrix waves  (waves 0, 1; 128 lanes over the 225 entries): F, V of this sample from the base matrices | barrier | J' = F J and T = F P
//                 (J and F / V double-buffered, so nothing is overwritten that a slower lane still reads) | barrier | P = T F' + V N V'.
// Every entry is the same expression, summed in the same order, as before: the same bits.
enum { PRE_MW = 2, PRE_NT = 64 * (PRE_MW + 1), PRE_ML = 64 * PRE_MW };
__global__ void __launch_bounds__(PRE_NT) preint_kernel(PreintArgs A) {
    enum { SMP_CHUNK = 64 };
    __shared__ double sh[2 * 225 + 225 + 225 + 2 * 225 + 2 * 270 + 2 * 64 + 32 + 7 * SMP_CHUNK];
    lds_d *S = (lds_d *)sh;
    lds_d *Jb = S, *P = S + 450, *T = S + 675, *Fb = S + 900, *Vb = S + 1350, *Bb = S + 1890, *st = S + 2018, *SB = S + 2050;
    // st: delta_p 0..2, delta_q 3..6, delta_v 7..9, acc_0 10..12, gyr_0 13..15, ba 16..18, bg 19..21, sum_dt 22
    const int tid = threadIdx.x;
    const bool mat = tid < PRE_ML;
    const int sl = tid - PRE_ML;      // lane of the state wave (>= 0 there)
    const double n2[6] = {A.noise[0] * A.noise[0], A.noise[1] * A.noise[1], A.noise[0] * A.noise[0], A.noise[1] * A.noise[1],
                          A.noise[2] * A.noise[2], A.noise[3] * A.noise[3]};      // :21-27 (ACC_N / GYR_N for k and k+1)
    for (int it = blockIdx.x; it < A.n; it += gridDim.x) {
        const double *init = A.init + (size_t)it * 12;
        const int s0 = A.first[it], ns = A.count[it];
        if (mat) for (int e = tid; e < 225; e += PRE_ML) { Jb[e] = (e / 15 == e % 15) ? 1.0 : 0.0; P[e] = 0.0; }
        if (sl == 0) {
            for (int i = 0; i < 10; i++) st[i] = 0.0;
            st[6] = 1.0;
            for (int i = 0; i < 12; i++) st[10 + i] = init[i];
            st[22] = 0.0;
        }
        // the samples of the buffer in LDS, chunk by chunk (a read from global memory at the top of every step of this serial chain costs a memory
        // round trip per sample)
        { const int cnt = min((int)SMP_CHUNK, ns) * 7; for (int i = tid; i < cnt; i += PRE_NT) SB[i] = A.samples[(size_t)s0 * 7 + i]; }
        __syncthreads();
        // base matrices of a sample from the state in st (integration_base.h:63-66, :90-118).  ONE lane computes all six 3x3 matrices and stores them
        // with literal indices: a lane per entry looked parallel but indexed the matrices by (lane / 3, lane % 3) -- a run-time index into
        // registers, i.e. the matrices went through scratch memory (a memory round trip per access, five times per sample, on the one chain of
        // the kernel that cannot be hidden: 224 B of scratch per lane until round 5)
        auto base = [&](const double *smp_dt_acc_gyr, lds_d *B) {
            const double dt = smp_dt_acc_gyr[0];
            const V3 a1(smp_dt_acc_gyr[1], smp_dt_acc_gyr[2], smp_dt_acc_gyr[3]), g1(smp_dt_acc_gyr[4], smp_dt_acc_gyr[5], smp_dt_acc_gyr[6]);
            const V3 a0(st[10], st[11], st[12]), g0(st[13], st[14], st[15]), ba(st[16], st[17], st[18]), bg(st[19], st[20], st[21]);
            const Quat dq(st[3], st[4], st[5], st[6]);
            const V3 un_gyr = 0.5 * (g0 + g1) - bg;                                             // :65
            const Quat rq = dq * Quat(un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2, 1.0);   // :66 (not normalised here)
            const M3 R0 = to_matrix(dq), R1 = to_matrix(rq);
            const M3 A0 = skew(a0 - ba), A1 = skew(a1 - ba), Rw = skew(un_gyr);
            const M3 M0 = R0 * A0, M1 = R1 * A1, Bm = m3_identity() - dt * Rw, M1B = M1 * Bm;
#pragma unroll
            for (int e = 0; e < 9; e++) { B[e] = R0.m[e]; B[9 + e] = R1.m[e]; B[18 + e] = M0.m[e]; B[27 + e] = M1.m[e]; B[36 + e] = M1B.m[e]; B[45 + e] = Bm.m[e]; }
        };
        auto sample_at = [&](int k, double (&o)[7]) {      // sample k straight from memory: the state wave looking one sample ahead across a chunk boundary
#pragma unroll
            for (int i = 0; i < 7; i++) o[i] = A.samples[(size_t)(s0 + k) * 7 + i];
        };
        if (sl == 0 && ns > 0) { double sm[7]; for (int i = 0; i < 7; i++) sm[i] = SB[i]; base(sm, Bb); }
        __syncthreads();
        for (int k = 0; k < ns; k++) {
            const int kc = k % SMP_CHUNK, cur = k & 1, nxt = cur ^ 1;
            if (kc == 0 && k > 0) {      // next chunk of samples (every 64 steps): nobody reads SB between these two barriers
                __syncthreads();
                const int cnt = min((int)SMP_CHUNK, ns - k) * 7;
                for (int i = tid; i < cnt; i += PRE_NT) SB[i] = A.samples[(size_t)(s0 + k) * 7 + i];
                __syncthreads();
            }
            const double dt = SB[kc * 7];
            lds_d *F = Fb + 225 * cur, *V = Vb + 270 * cur, *J = Jb + 225 * cur, *Jn = Jb + 225 * nxt;
            const lds_d *B = Bb + 64 * cur;
            if (mat) {
                for (int e = tid; e < 225; e += PRE_ML) F[e] = F_entry(B, e / 15, e % 15, dt);
                for (int e = tid; e < 270; e += PRE_ML) V[e] = V_entry(B, e / 18, e % 18, dt);
            } else {
                // the state wave, one sample ahead: state k -> k + 1 (mid-point rule :63-71, propagate :148-156), then the base matrices of sample k + 1
                double sm[7];
#pragma unroll
                for (int i = 0; i < 7; i++) sm[i] = SB[kc * 7 + i];
                if (sl == 0) {
                    const V3 a1(sm[1], sm[2], sm[3]), g1(sm[4], sm[5], sm[6]);
                    const V3 a0(st[10], st[11], st[12]), g0(st[13], st[14], st[15]), ba(st[16], st[17], st[18]), bg(st[19], st[20], st[21]);
                    const Quat dq(st[3], st[4], st[5], st[6]);
                    const V3 un_gyr = 0.5 * (g0 + g1) - bg;
                    const Quat rq = dq * Quat(un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2, 1.0);
                    const V3 un_acc_0 = rotate(dq, a0 - ba);
                    const V3 un_acc_1 = rotate(rq, a1 - ba);
                    const V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
                    const V3 dp(st[0], st[1], st[2]), dv(st[7], st[8], st[9]);
                    const V3 rp = dp + dv * dt + 0.5 * un_acc * dt * dt;
                    const V3 rv = dv + un_acc * dt;
                    const Quat qn = normalized(rq);                                                      // :153
                    st[0] = rp.x; st[1] = rp.y; st[2] = rp.z; st[3] = qn.x; st[4] = qn.y; st[5] = qn.z; st[6] = qn.w;
                    st[7] = rv.x; st[8] = rv.y; st[9] = rv.z;
                    st[10] = a1.x; st[11] = a1.y; st[12] = a1.z; st[13] = g1.x; st[14] = g1.y; st[15] = g1.z;
                    st[22] += dt;
                }
                if (k + 1 < ns && sl == 0) {
                    double sn[7];
                    if (kc + 1 < SMP_CHUNK) {
#pragma unroll
                        for (int i = 0; i < 7; i++) sn[i] = SB[(kc + 1) * 7 + i];
                    } else sample_at(k + 1, sn);
                    base(sn, Bb + 64 * nxt);
                }
            }
            __syncthreads();
            if (mat) {
                for (int e = tid; e < 225; e += PRE_ML) {
                    const int r = e / 15, c = e - 15 * r;
                    double jn = 0, tn = 0;
#pragma unroll
                    for (int q = 0; q < 15; q++) { jn += F[r * 15 + q] * J[q * 15 + c]; tn += F[r * 15 + q] * P[q * 15 + c]; }
                    Jn[e] = jn; T[e] = tn;
                }
            }
            __syncthreads();
            if (mat) {
                for (int e = tid; e < 225; e += PRE_ML) {
                    const int r = e / 15, c = e - 15 * r;
                    double pn = 0;
#pragma unroll
                    for (int q = 0; q < 15; q++) pn += T[r * 15 + q] * F[c * 15 + q];
                    double vn = 0;
#pragma unroll
                    for (int q = 0; q < 18; q++) vn += V[r * 18 + q] * n2[q / 3] * V[c * 18 + q];
                    P[e] = pn + vn;
                }
            }
            // (no barrier here: the next step's F / V / J' go to the other halves of their buffers, and its first barrier stands between this
            // step's writes of P and the next step's reads)
        }
        __syncthreads();
        const lds_d *J = Jb + 225 * (ns & 1);
        double *o = A.out + (size_t)it * PREINT_OUT;
        if (tid < 10) o[tid] = st[tid];
        if (tid < 6) o[10 + tid] = st[16 + tid];
        if (tid == 0) o[16] = st[22];
        for (int e = tid; e < 225; e += PRE_NT) { o[17 + e] = J[e]; o[242 + e] = P[e]; }
        __syncthreads();
    }
}

}  // namespace tcv
using namespace tcv;

// IntegrationBase ctor + push_back x count[i] (integration_base.h:13-36, :130-158); repropagate (:38-52) = same call with new biases.
// One pinned staging buffer, one copy in, everything on the calling thread's own stream: [init 12 n | samples 7 ns | first n, count n
// (ints)] -> device; the kernel's output records (PREINT_OUT doubles each) either come back through the same staging buffer
// (tcv_preintegrate) or stay where they are, in a buffer of their own that the returned handles share (tcv_preintegrate_device).
static int preintegrate_core(int n, const int *first, const int *count, const double *samples7, int num_samples, const double *acc0_gyr0_ba_bg,
                             const double noise[4], tcv_imu_preintegration *out, tcv_preint **handles) {
    if (n <= 0 || !first || !count || !samples7 || num_samples < 0 || !acc0_gyr0_ba_bg || !noise || (!out && !handles)) return TCV_ERR_INVALID;
    for (int i = 0; i < n; i++)
        if (first[i] < 0 || count[i] < 0 || first[i] + count[i] > num_samples) { set_error("preintegrate: sample range out of bounds"); return TCV_ERR_INVALID; }
    if (int rc = device_ready()) return rc;
    PreintArgs a;
    a.n = n;
    for (int i = 0; i < 4; i++) a.noise[i] = noise[i];
    const size_t ns = (size_t)std::max(1, num_samples);
    const size_t in_d = 12 * (size_t)n + 7 * ns, in_bytes = sizeof(double) * in_d + sizeof(int) * 2 * (size_t)n, out_bytes = sizeof(double) * PREINT_OUT * (size_t)n;
    char *h = (char *)tcv::host_staging_acquire(in_bytes + (out ? out_bytes : 0));
    if (!h) { set_error("hipHostMalloc (staging) failed"); return TCV_ERR_HIP; }
    char *d = nullptr, *d_res = nullptr;      // inputs (+ outputs of the host variant); the device variant's outputs
    hipStream_t st = tcv::util_stream();
    bool in_flight = false;
    auto done = [&](int rc) {
        if (in_flight) (void)(st ? hipStreamSynchronize(st) : hipDeviceSynchronize());      // the pinned buffer and the blobs go back to pools
        tcv::host_staging_release(h); (void)tcv::dev_free(d);
        if (rc != TCV_OK) (void)tcv::dev_free(d_res);
        return rc;
    };
    static const bool dbg = getenv("TCV_DEBUG_EST") != nullptr;      // developer: where a call's time goes
    auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = dbg ? now_us() : 0.0;
    const size_t out_off = (in_bytes + 15) & ~(size_t)15;
    hipError_t e = tcv::dev_malloc((void **)&d, out_off + (handles ? 0 : out_bytes) + 16);
    if (e == hipSuccess && handles) e = tcv::dev_malloc((void **)&d_res, out_bytes);
    if (e != hipSuccess) return done(hip_fail(e, "hipMalloc"));
    double *hd = (double *)h;
    std::memcpy(hd, acc0_gyr0_ba_bg, sizeof(double) * 12 * (size_t)n);
    if (num_samples) std::memcpy(hd + 12 * (size_t)n, samples7, sizeof(double) * 7 * (size_t)num_samples);
    int *hi = (int *)(hd + in_d);
    std::memcpy(hi, first, sizeof(int) * n); std::memcpy(hi + n, count, sizeof(int) * n);
    in_flight = true;
    const double t1 = dbg ? now_us() : 0.0;
    if ((e = hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st)) != hipSuccess) return done(hip_fail(e, "hipMemcpyAsync"));
    a.init = (const double *)d; a.samples = (const double *)d + 12 * (size_t)n; a.first = (const int *)((const double *)d + in_d); a.count = a.first + n;
    a.out = handles ? (double *)d_res : (double *)(d + out_off);
    hipLaunchKernelGGL(preint_kernel, dim3(std::min(n, 4096)), dim3(tcv::PRE_NT), 0, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return done(hip_fail(e, "preint kernel launch"));
    char *ho = h + in_bytes;
    if (out && (e = hipMemcpyAsync(ho, (const char *)a.out, out_bytes, hipMemcpyDeviceToHost, st)) != hipSuccess) return done(hip_fail(e, "hipMemcpyAsync"));
    const double t2 = dbg ? now_us() : 0.0;
    // device variant: nobody on the host needs the result -- the commands stay in flight on this thread's stream, the handles' blob carries
    // an event for consumers on other streams / host readers, the staging buffers are released at the thread's next wait on the stream
    hipEvent_t ready = nullptr;
    static const bool no_defer = getenv("TCV_PREINT_SYNC") != nullptr;      // A/B: wait as the host variant does
    if (handles && st && !no_defer && hipEventCreateWithFlags(&ready, hipEventDisableTiming) == hipSuccess) {
        if (hipEventRecord(ready, st) != hipSuccess) { (void)hipEventDestroy(ready); ready = nullptr; }
    }
    if (!ready) {
        if ((e = (st ? hipStreamSynchronize(st) : hipDeviceSynchronize())) != hipSuccess) return done(hip_fail(e, "hipStreamSynchronize"));
        in_flight = false;
    }
    if (dbg) fprintf(stderr, "[preint] n %d (%s): allocations + staging %.0f us, issue %.0f us, wait %.0f us\n", n, handles ? "device" : "host", t1 - t0, t2 - t1, now_us() - t2);
    if (out)
        for (int i = 0; i < n; i++) {
            const double *o = (const double *)ho + (size_t)i * PREINT_OUT;
            tcv_imu_preintegration &p = out[i];
            std::memcpy(p.delta_p, o, 24); std::memcpy(p.delta_q, o + 3, 32); std::memcpy(p.delta_v, o + 7, 24);
            std::memcpy(p.linearized_ba, o + 10, 24); std::memcpy(p.linearized_bg, o + 13, 24); p.sum_dt = o[16];
            std::memcpy(p.jacobian, o + 17, 225 * 8); std::memcpy(p.covariance, o + 242, 225 * 8);
        }
    if (handles) {
        auto blob = std::make_shared<tcv::DevBlob>();
        blob->p = d_res; (void)hipGetDevice(&blob->dev);
        blob->ready = ready;
        for (int i = 0; i < n; i++) {
            tcv_preint *q = new tcv_preint();
            q->dev = blob; q->d_out = (const double *)d_res + (size_t)i * PREINT_OUT;
            double sdt = 0.0;      // IntegrationBase::sum_dt: the kernel starts at 0.0 and adds the dt column row by row -- the same sum, the same bits
            for (int k = 0; k < count[i]; k++) sdt += samples7[7 * (size_t)(first[i] + k)];
            q->sum_dt = sdt;
            handles[i] = q;
        }
    }
    if (ready) { tcv::defer_release(h, d, st); return TCV_OK; }
    (void)done(TCV_OK);
    return TCV_OK;
}
extern "C" int tcv_preintegrate(int n, const int *first, const int *count, const double *samples7, int num_samples,
                                const double *acc0_gyr0_ba_bg, const double noise[4], tcv_imu_preintegration *out) {
    if (!out) return TCV_ERR_INVALID;
    return preintegrate_core(n, first, count, samples7, num_samples, acc0_gyr0_ba_bg, noise, out, nullptr);
}
extern "C" int tcv_preintegrate_device(int n, const int *first, const int *count, const double *samples7, int num_samples,
                                       const double *acc0_gyr0_ba_bg, const double noise[4], tcv_preint **out) {
    if (!out) return TCV_ERR_INVALID;
    for (int i = 0; i < std::max(n, 0); i++) out[i] = nullptr;
    return preintegrate_core(n, first, count, samples7, num_samples, acc0_gyr0_ba_bg, noise, nullptr, out);
}
int tcv_preint_host(const tcv_preint *pre) {
    if (!pre) return TCV_ERR_INVALID;
    std::lock_guard<std::mutex> g(pre->mu);
    if (pre->host) return TCV_OK;
    double o[PREINT_OUT];
    int cur = 0;
    const bool sw = hipGetDevice(&cur) == hipSuccess && pre->dev && cur != pre->dev->dev;
    if (sw) (void)hipSetDevice(pre->dev->dev);
    if (pre->dev) if (const int rcw = pre->dev->sync_ready()) { if (sw) (void)hipSetDevice(cur); return rcw; }
    const hipError_t e = hipMemcpy(o, pre->d_out, sizeof o, hipMemcpyDeviceToHost);
    if (sw) (void)hipSetDevice(cur);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy D2H (device-resident pre-integration)");
    tcv_imu_preintegration &p = pre->pod;
    std::memcpy(p.delta_p, o, 24); std::memcpy(p.delta_q, o + 3, 32); std::memcpy(p.delta_v, o + 7, 24);
    std::memcpy(p.linearized_ba, o + 10, 24); std::memcpy(p.linearized_bg, o + 13, 24); p.sum_dt = o[16];
    std::memcpy(p.jacobian, o + 17, 225 * 8); std::memcpy(p.covariance, o + 242, 225 * 8);
    pre->host = true;
    return TCV_OK;
}
extern "C" double tcv_preint_sum_dt(const tcv_preint *pre) { return pre ? pre->sum_dt : 0.0; }
extern "C" int tcv_preint_export(const tcv_preint *pre, tcv_imu_preintegration *out) {
    if (!pre || !out) return TCV_ERR_INVALID;
    if (int rc = tcv_preint_host(pre)) return rc;
    *out = pre->pod;
    return TCV_OK;
}
extern "C" void tcv_preint_destroy(tcv_preint *pre) { delete pre; }
