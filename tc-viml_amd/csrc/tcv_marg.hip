// Marginalisation on the GPU: MarginalizationInfo::preMarginalize + marginalize
// (reference vins_estimator/src/factor/marginalization_factor.cpp:110-129, :174-299) for a batch of
// independent windows, one workgroup per window.
//
//   1. every factor handed to addResidualBlockInfo is evaluated once at the current states
//      (ResidualBlockInfo::Evaluate, :3-69, loss corrector included), one lane per factor;
//   2. A = sum J'J and b = sum J'r are accumulated in LDS (packed lower triangle) factor by factor in
//      a fixed order (the reference deals factors round-robin to 4 pthreads, :232-261);
//   3. Amm = V diag(lambda) V' by a parallel cyclic Jacobi sweep in LDS, pseudo-inverse with
//      eigenvalues <= eps zeroed (:267-272);
//   4. Schur complement A' = Arr - Arm Amm^+ Amr, b' = brr - Arm Amm^+ bmm (:275-282), formed as
//      Arr - Z'Z with Z = diag(sqrt(lambda^+)) V' Amr;
//   5. A' = V2 diag(S) V2'  ->  linearized_jacobians = diag(sqrt S) V2', linearized_residuals =
//      diag(1/sqrt S) V2' b' with S thresholded at eps (:284-293), eigenvalues ascending like
//      Eigen::SelfAdjointEigenSolver.
//
// Block order (the reference's is unordered_map / address dependent, :176-194): dropped blocks in the
// order they were added to the problem, then kept blocks in that order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include "tcv_factors.h"
#include "tcv_host.h"

namespace tcv {

enum { MARG_MAX_M = 64, MARG_MAX_N = 80, MARG_MAX_POS = MARG_MAX_M + MARG_MAX_N, MARG_MAX_X = 320, MARG_NT = 256 };
enum { MARG_OUT_J0 = 0, MARG_OUT_R0 = 6400, MARG_OUT_AS = 6480, MARG_OUT_BS = 12880, MARG_OUT_X = 12960, MARG_OUT_STRIDE = 13312 };
enum { MARG_SCR_Z = 0, MARG_SCR_PR = MARG_MAX_M * MARG_MAX_N, MARG_SCR_SQ = MARG_SCR_PR + 256, MARG_SCR_STRIDE = MARG_SCR_SQ + 16 * 225 + 450 * 16 };

struct MargHdr {
    int nblk, pos, m, n, nx;
    int n_imu, n_proj, prior_n, prior_nblk, prior_xsize;
    int o_blk;     // nblk x 5: gsize, goff, mloc (-1 constant), kind, xsrc (offset in the solve state, -1 none)
    int o_imu;     // n_imu x 4
    int o_proj;    // n_proj x 4
    int o_prior;   // prior_nblk x 4: blk, idx, gsize, x0 offset
    int o_pcol;    // prior_n: mloc index of every J0 column (-1 constant)
    int d_x, d_imu, d_proj, d_prior, d_misc;
    long long ibase, dbase;
    int solve_window, pad;
};

struct MargArgs {
    const MargHdr *hdr;
    const int *ipool;
    const double *dpool;
    const double *solve_state;   // may be null
    double *out;                 // per window MARG_OUT_STRIDE
    int *out_status;             // per window: 0 ok
    double *scratch;             // per workgroup MARG_SCR_STRIDE
    int nwin, state_stride, use_solved_state, pad;
};

__device__ __forceinline__ int pidx(int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

// Parallel cyclic Jacobi eigen-decomposition of the symmetric matrix M (row-major, leading dimension ld,
// padded to the even size de with a zero row/column), V = eigenvectors.  A round-robin pairing splits M
// into (de/2)^2 independent 2x2 blocks B_kl <- J_k' B_kl J_l, so one round is: angles | barrier | every
// block and the V columns in place | barrier.  Returns the number of sweeps used.
__device__ int jacobi_eig(double *M, double *V, int d, int ld, double *rot, int *cnt, int tid) {
    const int de = d + (d & 1), half = de / 2;
    int *rp = reinterpret_cast<int *>(rot);          // [0..half) p, [half..2half) q
    double *rc = rot + 64, *rs = rot + 64 + 48;      // half <= 40
    double md = 0;
    for (int i = 0; i < d; i++) md = fmax(md, fabs(M[i * ld + i]));
    const double tiny = 1e-19 * md;
    for (int i = tid; i < de * de; i += MARG_NT) { const int r = i / de, c = i - r * de; V[r * ld + c] = (r == c) ? 1.0 : 0.0; }
    if (de > d) for (int i = tid; i < de; i += MARG_NT) { M[i * ld + d] = 0.0; M[d * ld + i] = 0.0; }
    __syncthreads();
    int sweep = 0;
    for (; sweep < 24; sweep++) {
        if (tid == 0) *cnt = 0;
        __syncthreads();
        for (int r = 0; r < de - 1; r++) {
            if (tid < half) {
                int a = (r + tid) % (de - 1), b = (r - tid + de - 1) % (de - 1);
                if (tid == 0) b = de - 1;
                const int p = min(a, b), q = max(a, b);
                double c = 1.0, s = 0.0;
                const double apq = M[p * ld + q], app = M[p * ld + p], aqq = M[q * ld + q];
                if (fabs(apq) > tiny && fabs(apq) > 1e-15 * sqrt(fabs(app) * fabs(aqq))) {
                    const double tau = (aqq - app) / (2.0 * apq);
                    const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    s = t * c;
                    atomicAdd(cnt, 1);
                }
                rp[tid] = p; rp[half + tid] = q; rc[tid] = c; rs[tid] = s;
            }
            __syncthreads();
            for (int w = tid; w < half * half; w += MARG_NT) {
                const int k = w / half, l = w - k * half;
                const double ck = rc[k], sk = rs[k], cl = rc[l], sl = rs[l];
                if (sk == 0.0 && sl == 0.0) continue;
                const int pk = rp[k], qk = rp[half + k], pl = rp[l], ql = rp[half + l];
                const double b00 = M[pk * ld + pl], b01 = M[pk * ld + ql], b10 = M[qk * ld + pl], b11 = M[qk * ld + ql];
                const double t00 = cl * b00 - sl * b01, t01 = sl * b00 + cl * b01;
                const double t10 = cl * b10 - sl * b11, t11 = sl * b10 + cl * b11;
                double n00 = ck * t00 - sk * t10, n01 = ck * t01 - sk * t11, n10 = sk * t00 + ck * t10, n11 = sk * t01 + ck * t11;
                if (k == l) { n01 = 0.0; n10 = 0.0; }
                M[pk * ld + pl] = n00; M[pk * ld + ql] = n01; M[qk * ld + pl] = n10; M[qk * ld + ql] = n11;
            }
            for (int w = tid; w < d * half; w += MARG_NT) {
                const int k = w / d, i = w - k * d;
                const double c = rc[k], s = rs[k];
                if (s == 0.0) continue;
                const int p = rp[k], q = rp[half + k];
                const double vp = V[i * ld + p], vq = V[i * ld + q];
                V[i * ld + p] = c * vp - s * vq; V[i * ld + q] = s * vp + c * vq;
            }
            __syncthreads();
        }
        if (*cnt == 0) break;
        __syncthreads();
    }
    __syncthreads();
    return sweep;
}

__global__ void __launch_bounds__(MARG_NT) marg_kernel(MargArgs Aarg) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    double *scr = Aarg.scratch + (size_t)blockIdx.x * MARG_SCR_STRIDE;
    for (int win = blockIdx.x; win < Aarg.nwin; win += gridDim.x) {
        const MargHdr &H = Aarg.hdr[win];
        const int *ip = Aarg.ipool + H.ibase;
        const double *dp = Aarg.dpool + H.dbase;
        const int pos = H.pos, m = H.m, n = H.n;
        const int npk = pos * (pos + 1) / 2;
        // LDS carve
        double *Apk = lds;                                   // packed lower triangle of A (later: V2)
        const int me = m + (m & 1), ne = n + (n & 1);
        const int r1 = max(npk, ne * (ne + 1));
        double *R2 = lds + ((r1 + 1) & ~1);                  // Amm + V (later A')
        const int r2 = max(2 * me * (me + 1), ne * (ne + 1));
        double *bv = R2 + ((r2 + 1) & ~1);                   // pos
        double *x = bv + MARG_MAX_POS;                       // nx
        double *rot = x + MARG_MAX_X;                        // 160
        double *lam = rot + 160;                             // MARG_MAX_N
        double *stage = lam + MARG_MAX_N;                    // 64 proj records or 1 imu record
        int *cnt = reinterpret_cast<int *>(rot + 158);
        double *out = Aarg.out + (size_t)win * MARG_OUT_STRIDE;

        const int *blk = ip + H.o_blk;
        for (int b = tid; b < H.nblk; b += MARG_NT) {
            const int gs = blk[b * 5], go = blk[b * 5 + 1], xs = blk[b * 5 + 4];
            for (int i = 0; i < gs; i++)
                x[go + i] = (Aarg.use_solved_state && xs >= 0 && Aarg.solve_state)
                                ? Aarg.solve_state[(size_t)H.solve_window * Aarg.state_stride + xs + i]
                                : dp[H.d_x + go + i];
        }
        for (int i = tid; i < npk; i += MARG_NT) Apk[i] = 0.0;
        for (int i = tid; i < pos; i += MARG_NT) bv[i] = 0.0;
        const double *misc = dp + H.d_misc;
        const double G3[3] = {misc[0], misc[1], misc[2]};
        __syncthreads();

        // ---- prior factor (MarginalizationFactor::Evaluate, :335-384)
        if (H.prior_n > 0) {
            const int np = H.prior_n;
            const double *J0 = dp + H.d_prior, *r0 = J0 + np * np, *x0 = r0 + np;
            double *pdx = scr + MARG_SCR_PR, *pr = pdx + 128;
            if (tid < H.prior_nblk) {
                const int *pb = ip + H.o_prior + tid * 4;
                prior_block_dx(x + blk[pb[0] * 5 + 1], x0 + pb[3], pb[2], pdx + pb[1]);
            }
            __syncthreads();
            if (tid < np) {
                double r = r0[tid];
                for (int j = 0; j < np; j++) r += J0[tid + np * j] * pdx[j];
                pr[tid] = r;
            }
            __syncthreads();
            const int *pcol = ip + H.o_pcol;
            for (int e = tid; e < np * np; e += MARG_NT) {
                const int a = e / np, b = e - a * np;
                if (b > a) continue;
                const int ia = pcol[a], ib = pcol[b];
                if (ia < 0 || ib < 0) continue;
                double s = 0;
                for (int i = 0; i < np; i++) s += J0[i + np * a] * J0[i + np * b];
                Apk[pidx(ia, ib)] += s;
            }
            if (tid < np && pcol[tid] >= 0) {
                double s = 0;
                for (int i = 0; i < np; i++) s += J0[i + np * tid] * pr[i];
                bv[pcol[tid]] += s;
            }
            __syncthreads();
        }
        // ---- IMU factors, one at a time (only the factor touching the marginalised frame is passed in)
        for (int f = 0; f < H.n_imu; f++) {
            const int *b = ip + H.o_imu + f * 4;
            double *S = stage + 1500;
            if (tid < 16) (void)imu_sqrt_info_group(dp + H.d_imu + f * IMU_CONST + IMU_COV, S, stage + 512, stage + 512 + 225, tid);
            if (tid == 64) {
                imu_raw(x + blk[b[0] * 5 + 1], x + blk[b[1] * 5 + 1], x + blk[b[2] * 5 + 1], x + blk[b[3] * 5 + 1],
                        dp + H.d_imu + f * IMU_CONST, G3, stage + 30, IMU_STRIDE_J, stage, IMU_STRIDE_J);
            }
            __syncthreads();
            if (tid < 31) {
                double *rec = stage + tid;
                double v[15];
                for (int r = 0; r < 15; r++) v[r] = rec[r * IMU_STRIDE_J];
                for (int r = 0; r < 15; r++) {
                    double a = 0;
                    for (int s2 = r; s2 < 15; s2++) a += S[r * 15 + s2] * v[s2];
                    rec[r * IMU_STRIDE_J] = a;
                }
            }
            __syncthreads();
            const int colc[4] = {0, 6, 15, 21}, colw[4] = {6, 9, 6, 9};
            for (int e = tid; e < 30 * 31; e += MARG_NT) {
                const int ca = e / 31, cb = e - ca * 31;   // cb == 30: residual column
                if (cb < 30 && cb > ca) continue;
                int sa = 0, sb = 0;
                while (sa < 3 && ca >= colc[sa + 1]) sa++;
                const int la = blk[b[sa] * 5 + 2];
                if (la < 0) continue;
                const int ia = la + ca - colc[sa];
                double s = 0;
                for (int r = 0; r < 15; r++) s += stage[r * IMU_STRIDE_J + ca] * stage[r * IMU_STRIDE_J + cb];
                if (cb == 30) { bv[ia] += s; continue; }
                while (sb < 3 && cb >= colc[sb + 1]) sb++;
                const int lb = blk[b[sb] * 5 + 2];
                if (lb < 0) continue;
                Apk[pidx(ia, lb + cb - colc[sb])] += s;
            }
            (void)colw;
            __syncthreads();
        }
        // ---- projection factors in chunks of 64: evaluate in parallel, accumulate one factor at a time
        for (int f0 = 0; f0 < H.n_proj; f0 += 64) {
            const int fn = min(64, H.n_proj - f0);
            if (tid < fn) {
                const int *pf = ip + H.o_proj + (f0 + tid) * 4;
                double *rec = stage + tid * PROJ_REC;
                double r[2];
                proj_eval(x + blk[pf[0] * 5 + 1], x + blk[pf[1] * 5 + 1], x + blk[pf[2] * 5 + 1], x[blk[pf[3] * 5 + 1]],
                          dp + H.d_proj + (f0 + tid) * 6, misc[3], r, rec, PROJ_STRIDE);
                (void)loss_correct2(r, rec, 19, PROJ_STRIDE, misc[4]);
                rec[19] = r[0]; rec[PROJ_STRIDE + 19] = r[1];
            }
            __syncthreads();
            for (int f = 0; f < fn; f++) {
                const int *pf = ip + H.o_proj + (f0 + f) * 4;
                const double *rec = stage + f * PROJ_REC;
                for (int e = tid; e < 19 * 20; e += MARG_NT) {
                    const int ca = e / 20, cb = e - ca * 20;   // cb == 19: residual column
                    if (cb < 19 && cb > ca) continue;
                    const int sa = ca == 18 ? 3 : ca / 6;
                    const int la = blk[pf[sa] * 5 + 2];
                    if (la < 0) continue;
                    const int ia = la + (ca == 18 ? 0 : ca - 6 * sa);
                    const double s = rec[ca] * rec[cb] + rec[PROJ_STRIDE + ca] * rec[PROJ_STRIDE + cb];
                    if (cb == 19) { bv[ia] += s; continue; }
                    const int sb = cb == 18 ? 3 : cb / 6;
                    const int lb = blk[pf[sb] * 5 + 2];
                    if (lb < 0) continue;
                    Apk[pidx(ia, lb + (cb == 18 ? 0 : cb - 6 * sb))] += s;
                }
                __syncthreads();
            }
        }
        // ---- Amm = V diag(lam) V'
        const int ldm = me + 1;
        double *Mm = R2, *Vm = R2 + me * ldm;
        for (int i = tid; i < m * m; i += MARG_NT) { const int r = i / m, c = i - r * m; Mm[r * ldm + c] = Apk[pidx(r, c)]; }
        __syncthreads();
        const int sweeps1 = jacobi_eig(Mm, Vm, m, ldm, rot, cnt, tid);
        if (tid < m) { const double l = Mm[tid * ldm + tid]; lam[tid] = l > 1e-8 ? sqrt(1.0 / l) : 0.0; }
        __syncthreads();
        // Z = diag(sqrt(lam^+)) V' Amr  (m x n) and zb = diag(sqrt(lam^+)) V' bmm, kept in global scratch
        double *Z = scr + MARG_SCR_Z, *zb = scr + MARG_SCR_PR;
        for (int e = tid; e < m * n; e += MARG_NT) {
            const int k = e / n, j = e - k * n;
            double s = 0;
            for (int p = 0; p < m; p++) s += Vm[p * ldm + k] * Apk[pidx(m + j, p)];
            Z[k * n + j] = lam[k] * s;
        }
        if (tid < m) {
            double s = 0;
            for (int p = 0; p < m; p++) s += Vm[p * ldm + tid] * bv[p];
            zb[tid] = lam[tid] * s;
        }
        __syncthreads();
        // A' = Arr - Z'Z, b' = brr - Z' zb
        const int ldn = ne + 1;
        double *As = R2, *V2 = Apk;
        double keepA[ (MARG_MAX_N * MARG_MAX_N + MARG_NT - 1) / MARG_NT ];
        {
            int q = 0;
            for (int e = tid; e < n * n; e += MARG_NT, q++) {
                const int i = e / n, j = e - i * n;
                double s = Apk[pidx(m + i, m + j)];
                for (int k = 0; k < m; k++) s -= Z[k * n + i] * Z[k * n + j];
                keepA[q] = s;
            }
        }
        double bprime = 0;
        if (tid < n) {
            bprime = bv[m + tid];
            for (int k = 0; k < m; k++) bprime -= Z[k * n + tid] * zb[k];
        }
        __syncthreads();   // every read of Vm / Apk is done: R2 and Apk can be overwritten
        {
            int q = 0;
            for (int e = tid; e < n * n; e += MARG_NT, q++) {
                const int i = e / n, j = e - i * n;
                As[i * ldn + j] = keepA[q];
                out[MARG_OUT_AS + i * n + j] = keepA[q];
            }
        }
        if (tid < n) { bv[tid] = bprime; out[MARG_OUT_BS + tid] = bprime; }
        __syncthreads();
        const int sweeps2 = jacobi_eig(As, V2, n, ldn, rot, cnt, tid);
        // ascending order like SelfAdjointEigenSolver, eps thresholding, outputs (J0 column-major n x n)
        if (tid < n) lam[tid] = As[tid * ldn + tid];
        __syncthreads();
        if (tid < n) {
            const double l = lam[tid];
            int rank = 0;
            for (int j = 0; j < n; j++) rank += (lam[j] < l || (lam[j] == l && j < tid)) ? 1 : 0;
            const double S = l > 1e-8 ? l : 0.0, Sinv = l > 1e-8 ? 1.0 / l : 0.0;
            const double ss = sqrt(S), si = sqrt(Sinv);
            double rb = 0;
            for (int j = 0; j < n; j++) {
                const double v = V2[j * ldn + tid];
                out[MARG_OUT_J0 + rank + n * j] = ss * v;
                rb += v * bv[j];
            }
            out[MARG_OUT_R0 + rank] = si * rb;
        }
        for (int i = tid; i < H.nx; i += MARG_NT) out[MARG_OUT_X + i] = x[i];
        if (tid == 0) Aarg.out_status[win] = (sweeps1 >= 24 || sweeps2 >= 24) ? 1 : 0;   // 1: Jacobi hit the sweep cap
        if (tid == 0) { out[MARG_OUT_X + MARG_MAX_X] = sweeps1; out[MARG_OUT_X + MARG_MAX_X + 1] = sweeps2; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------
struct MargWindow {
    MargHdr hdr;
    std::vector<int> keep_block;      // marg-problem block index of every kept block (in mloc order)
    std::vector<int> keep_size, keep_idx, keep_goff;
    std::vector<double *> keep_addr;
};
struct MargState {
    std::vector<MargWindow> win;
    MargHdr *d_hdr = nullptr;
    int *d_ipool = nullptr, *d_status = nullptr;
    double *d_dpool = nullptr, *d_out = nullptr, *d_scratch = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    size_t lds_bytes = 0;
    int grid = 0;
    bool ran = false;
};

static void marg_free(tcv_batch *b) {
    MargState *s = (MargState *)b->marg;
    if (!s) return;
    (void)hipFree(s->d_hdr); (void)hipFree(s->d_ipool); (void)hipFree(s->d_status); (void)hipFree(s->d_dpool);
    (void)hipFree(s->d_out); (void)hipFree(s->d_scratch);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
    b->marg = nullptr;
}

static int pack_marg(const tcv_problem &p, double *const *drop, int ndrop, const tcv_problem *solve_p, const Packed *solve_pk,
                     MargWindow &mw, std::vector<int> &I, std::vector<double> &D) {
    const int nb = (int)p.blocks.size();
    if (!p.line.empty()) { set_error("line factors are not marginalised (estimator.cpp:1992 `if (0)`)"); return TCV_ERR_UNSUPPORTED; }
    if (p.prior.size() > 1) { set_error("more than one marginalisation factor"); return TCV_ERR_UNSUPPORTED; }
    std::vector<char> touched(nb, 0), dropped(nb, 0);
    for (auto &f : p.imu) for (int k = 0; k < 4; k++) touched[f.b[k]] = 1;
    for (auto &f : p.proj) for (int k = 0; k < 4; k++) touched[f.b[k]] = 1;
    for (auto &f : p.prior) for (int b : f.b) touched[b] = 1;
    for (int k = 0; k < ndrop; k++) {
        auto it = p.index.find(drop[k]);
        if (it == p.index.end()) { set_error("marginalize: dropped block is not part of the problem"); return TCV_ERR_INVALID; }
        if (touched[it->second]) dropped[it->second] = 1;
    }
    // ambient offsets and [m | n] tangent order
    std::vector<int> id_of(nb, -1), gsize, goff, mloc, kind, xsrc, orig;
    int nx = 0;
    for (int b = 0; b < nb; b++) {
        if (!touched[b]) continue;
        id_of[b] = (int)gsize.size();
        gsize.push_back(p.blocks[b].size); goff.push_back(nx); kind.push_back(p.blocks[b].kind); mloc.push_back(-1); xsrc.push_back(-1);
        orig.push_back(b);
        nx += p.blocks[b].size;
    }
    const int nblk = (int)gsize.size();
    int pos = 0;
    for (int c = 0; c < nblk; c++) {
        const ParamBlock &pb = p.blocks[orig[c]];
        if (dropped[orig[c]] && !pb.constant) { mloc[c] = pos; pos += pb.kind == KIND_POSE ? 6 : pb.size; }
    }
    const int m = pos;
    mw.keep_block.clear(); mw.keep_size.clear(); mw.keep_idx.clear(); mw.keep_addr.clear(); mw.keep_goff.clear();
    for (int c = 0; c < nblk; c++) {
        const ParamBlock &pb = p.blocks[orig[c]];
        if (dropped[orig[c]]) continue;
        if (pb.constant) continue;   // constant blocks are not part of the linearised prior
        mloc[c] = pos;
        mw.keep_block.push_back(c); mw.keep_size.push_back(pb.size); mw.keep_idx.push_back(pos); mw.keep_addr.push_back(pb.addr);
        mw.keep_goff.push_back(goff[c]);
        pos += pb.kind == KIND_POSE ? 6 : pb.size;
    }
    const int n = pos - m;
    if (m < 1 || n < 1) { set_error("marginalize: nothing to drop or nothing to keep"); return TCV_ERR_INVALID; }
    if (m > MARG_MAX_M || n > MARG_MAX_N || nx > MARG_MAX_X || p.imu.size() > 16) {
        set_error("marginalisation too large for the LDS-resident kernel (m <= 64, n <= 80)");
        return TCV_ERR_TOO_LARGE;
    }
    // where the current value of each block lives in the solve's state vector
    if (solve_p && solve_pk) {
        std::unordered_map<double *, int> off;
        int o = 0;
        for (int blkid : solve_pk->cam_block) { off[solve_p->blocks[blkid].addr] = o; o += solve_p->blocks[blkid].size; }
        for (int blkid : solve_pk->lm_block) { off[solve_p->blocks[blkid].addr] = o; o += 1; }
        for (int c = 0; c < nblk; c++) { auto it = off.find(p.blocks[orig[c]].addr); if (it != off.end()) xsrc[c] = it->second; }
    }
    MargHdr &H = mw.hdr;
    std::memset(&H, 0, sizeof H);
    H.nblk = nblk; H.pos = pos; H.m = m; H.n = n; H.nx = nx;
    H.n_imu = (int)p.imu.size(); H.n_proj = (int)p.proj.size();
    H.ibase = (long long)I.size(); H.dbase = (long long)D.size();
    const size_t i0 = I.size(), d0 = D.size();
    auto imark = [&]() { return (int)(I.size() - i0); };
    auto dmark = [&]() { return (int)(D.size() - d0); };
    H.o_blk = imark();
    for (int c = 0; c < nblk; c++) { I.push_back(gsize[c]); I.push_back(goff[c]); I.push_back(mloc[c]); I.push_back(kind[c]); I.push_back(xsrc[c]); }
    H.o_imu = imark();
    for (auto &f : p.imu) for (int k = 0; k < 4; k++) I.push_back(id_of[f.b[k]]);
    H.o_proj = imark();
    for (auto &f : p.proj) for (int k = 0; k < 4; k++) I.push_back(id_of[f.b[k]]);
    H.o_prior = imark();
    std::vector<int> pcol;
    const tcv_prior *pr = p.prior.empty() ? nullptr : p.prior[0].prior;
    if (pr) {
        if (pr->n > 128) { set_error("prior with more than 128 rows"); return TCV_ERR_TOO_LARGE; }
        H.prior_n = pr->n; H.prior_nblk = (int)pr->size.size(); H.prior_xsize = (int)pr->x0.size();
        pcol.assign(pr->n, -1);
        for (int k = 0; k < H.prior_nblk; k++) {
            const int c = id_of[p.prior[0].b[k]];
            I.push_back(c); I.push_back(pr->idx[k]); I.push_back(pr->size[k]); I.push_back(pr->xoff[k]);
            const int local = pr->size[k] == 7 ? 6 : pr->size[k];
            for (int j = 0; j < local; j++) if (pr->idx[k] + j < pr->n) pcol[pr->idx[k] + j] = mloc[c] < 0 ? -1 : mloc[c] + j;
        }
    }
    H.o_pcol = imark();
    for (int v : pcol) I.push_back(v);
    H.d_x = dmark();
    for (int c = 0; c < nblk; c++) { const ParamBlock &pb = p.blocks[orig[c]]; D.insert(D.end(), pb.addr, pb.addr + pb.size); }
    H.d_imu = dmark();
    for (auto &f : p.imu) {
        const tcv_imu_preintegration &q = f.pre;
        D.insert(D.end(), q.delta_p, q.delta_p + 3); D.insert(D.end(), q.delta_q, q.delta_q + 4);
        D.insert(D.end(), q.delta_v, q.delta_v + 3); D.insert(D.end(), q.linearized_ba, q.linearized_ba + 3);
        D.insert(D.end(), q.linearized_bg, q.linearized_bg + 3); D.push_back(q.sum_dt);
        const int rc[5][2] = {{0, 9}, {0, 12}, {3, 12}, {6, 9}, {6, 12}};
        for (auto &b : rc) for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) D.push_back(q.jacobian[(b[0] + i) * 15 + b[1] + j]);
        D.insert(D.end(), q.covariance, q.covariance + 225);
    }
    H.d_proj = dmark();
    double psi = 0, pla = 0;
    for (size_t k = 0; k < p.proj.size(); k++) {
        const ProjFac &f = p.proj[k];
        if (k == 0) { psi = f.sqrt_info; pla = f.loss_a; }
        else if (f.sqrt_info != psi || f.loss_a != pla) { set_error("projection factors must share sqrt_info and loss"); return TCV_ERR_UNSUPPORTED; }
        if (p.blocks[f.b[3]].size != 1) { set_error("projection factor: 4th block must be an inverse depth"); return TCV_ERR_UNSUPPORTED; }
        D.insert(D.end(), f.pts, f.pts + 6);
    }
    H.d_prior = dmark();
    if (pr) { D.insert(D.end(), pr->J0.begin(), pr->J0.end()); D.insert(D.end(), pr->r0.begin(), pr->r0.end()); D.insert(D.end(), pr->x0.begin(), pr->x0.end()); }
    H.d_misc = dmark();
    D.insert(D.end(), p.G, p.G + 3); D.push_back(psi); D.push_back(pla); D.push_back(0.0);
    if (D.size() & 1) D.push_back(0.0);
    return TCV_OK;
}

}  // namespace tcv
using namespace tcv;

int tcv_marg_attach(tcv_batch *b, tcv_problem *const *marg_problems, double *const *const *marg_drop, const int *marg_num_drop) {
    MargState *s = new MargState();
    b->marg = s;
    b->marg_free = marg_free;
    s->win.resize(b->n);
    std::vector<int> I;
    std::vector<double> D;
    std::vector<MargHdr> hdrs(b->n);
    size_t lds = 0;
    for (int w = 0; w < b->n; w++) {
        if (!marg_problems[w] || !marg_drop || !marg_drop[w]) { set_error("marginalisation problem / drop list missing"); return TCV_ERR_INVALID; }
        const bool same = marg_problems[w] == b->problems[w];
        const int rc = pack_marg(*marg_problems[w], marg_drop[w], marg_num_drop[w], b->problems[w], &b->packed[w], s->win[w], I, D);
        (void)same;
        if (rc != TCV_OK) return rc;
        s->win[w].hdr.solve_window = w;
        hdrs[w] = s->win[w].hdr;
        const int pos = hdrs[w].pos, m = hdrs[w].m, n = hdrs[w].n;
        const int me = m + (m & 1), ne = n + (n & 1);
        const int r1 = std::max(pos * (pos + 1) / 2, ne * (ne + 1)), r2 = std::max(2 * me * (me + 1), ne * (ne + 1));
        const size_t need = (size_t)(((r1 + 1) & ~1) + ((r2 + 1) & ~1) + MARG_MAX_POS + MARG_MAX_X + 160 + MARG_MAX_N + 64 * PROJ_REC) * 8;
        if (need > (size_t)LDS_DOUBLES * 8) { set_error("marginalisation does not fit LDS"); return TCV_ERR_TOO_LARGE; }
        lds = std::max(lds, need);
    }
    s->lds_bytes = lds;
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return TCV_ERR_HIP;
    s->grid = std::min(b->n, prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
#define MUP(dst, src, T, cnt)                                                                    \
    do {                                                                                         \
        hipError_t e_ = hipMalloc((void **)&dst, sizeof(T) * std::max<size_t>(1, (cnt)));        \
        if (e_ != hipSuccess) return hip_fail(e_, "hipMalloc");                                  \
        if (src) {                                                                               \
            e_ = hipMemcpy(dst, src, sizeof(T) * (cnt), hipMemcpyHostToDevice);                  \
            if (e_ != hipSuccess) return hip_fail(e_, "hipMemcpy H2D");                          \
        }                                                                                        \
    } while (0)
    MUP(s->d_hdr, hdrs.data(), MargHdr, hdrs.size());
    MUP(s->d_ipool, I.data(), int, I.size());
    MUP(s->d_dpool, D.data(), double, D.size());
    MUP(s->d_out, (double *)nullptr, double, (size_t)b->n * MARG_OUT_STRIDE);
    MUP(s->d_status, (int *)nullptr, int, (size_t)b->n);
    MUP(s->d_scratch, (double *)nullptr, double, (size_t)s->grid * MARG_SCR_STRIDE);
#undef MUP
    (void)hipMemset(s->d_status, 0xff, sizeof(int) * b->n);
    if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) return TCV_ERR_HIP;
    b->input_bytes += 0;   // marginalisation reads the same resident inputs
    return TCV_OK;
}

int tcv_marg_run(tcv_batch *b, void *stream) {
    MargState *s = (MargState *)b->marg;
    if (!s) { set_error("batch was created without marginalisation problems"); return TCV_ERR_INVALID; }
    MargArgs a;
    std::memset(&a, 0, sizeof a);
    a.hdr = s->d_hdr; a.ipool = s->d_ipool; a.dpool = s->d_dpool; a.solve_state = b->d_state; a.out = s->d_out;
    a.out_status = s->d_status; a.scratch = s->d_scratch; a.nwin = b->n; a.state_stride = b->state_stride;
    a.use_solved_state = b->solved ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipFuncSetAttribute((const void *)marg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(marg)");
    if ((e = hipEventRecord(s->ev0, st)) != hipSuccess) return hip_fail(e, "hipEventRecord");
    hipLaunchKernelGGL(marg_kernel, dim3(s->grid), dim3(MARG_NT), s->lds_bytes, st, a);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "marg kernel launch");
    if ((e = hipEventRecord(s->ev1, st)) != hipSuccess) return hip_fail(e, "hipEventRecord");
    s->ran = true;
    return TCV_OK;
}

// elapsed time of the last marginalisation launch (called from tcv_batch_synchronize)
void tcv_marg_elapsed(tcv_batch *b) {
    MargState *s = (MargState *)b->marg;
    if (s && s->ran) (void)hipEventElapsedTime(&b->marg_ms, s->ev0, s->ev1);
}

int tcv_marg_get_prior(tcv_batch *b, int window, tcv_prior **out) {
    MargState *s = (MargState *)b->marg;
    if (!s || !s->ran || window < 0 || window >= b->n) { set_error("no marginalisation result for this window"); return TCV_ERR_INVALID; }
    const MargWindow &mw = s->win[window];
    const int n = mw.hdr.n, m = mw.hdr.m;
    std::vector<double> o(MARG_OUT_STRIDE);
    int status = -1;
    hipError_t e = hipMemcpy(o.data(), s->d_out + (size_t)window * MARG_OUT_STRIDE, sizeof(double) * MARG_OUT_STRIDE, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy D2H");
    e = hipMemcpy(&status, s->d_status + window, sizeof(int), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy D2H");
    if (status < 0) { set_error("marginalisation kernel did not complete for this window"); return TCV_ERR_NUMERIC; }
    tcv_prior *pr = new tcv_prior();
    pr->m = m; pr->n = n;
    int xo = 0;
    for (size_t k = 0; k < mw.keep_block.size(); k++) {
        pr->size.push_back(mw.keep_size[k]);
        pr->idx.push_back(mw.keep_idx[k] - m);
        pr->xoff.push_back(xo);
        for (int i = 0; i < mw.keep_size[k]; i++) pr->x0.push_back(o[MARG_OUT_X + mw.keep_goff[k] + i]);
        xo += mw.keep_size[k];
        pr->addr.push_back(mw.keep_addr[k]);
    }
    pr->J0.assign(o.begin() + MARG_OUT_J0, o.begin() + MARG_OUT_J0 + (size_t)n * n);
    pr->r0.assign(o.begin() + MARG_OUT_R0, o.begin() + MARG_OUT_R0 + n);
    pr->As.assign(o.begin() + MARG_OUT_AS, o.begin() + MARG_OUT_AS + (size_t)n * n);
    pr->bs.assign(o.begin() + MARG_OUT_BS, o.begin() + MARG_OUT_BS + n);
    for (double v : pr->J0) if (!(v == v)) { delete pr; set_error("NaN in marginalisation result"); return TCV_ERR_NUMERIC; }
    *out = pr;
    return TCV_OK;
}
