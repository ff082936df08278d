// Fused sliding-window solve for gfx950 (MI355X): one workgroup owns one window at a time and
// runs the whole trust-region (dogleg) loop of ceres::Solve as the reference configures it
// (vins_estimator/src/estimator.cpp:1888-1900: SPARSE_SCHUR + DOGLEG, Ceres defaults otherwise)
// without ever writing a Jacobian or the normal equations to HBM:
//
//   linearise   residual blocks are evaluated one lane per block into LDS staging records
//               (imu_factor.h:19-181, projection_factor.cpp:21-124, line_projection_factor.cpp:19-120,
//               marginalization_factor.cpp:335-384 with the CauchyLoss corrector of :37-68) and
//               gathered, destination-driven and in a fixed order, into the reduced camera
//               system that lives in LDS as 16x16 FP64 tiles (lower triangle, XOR-swizzled);
//   Schur       the inverse-depth blocks are 1x1, so their elimination is a scaled rank-1 update
//               per landmark, again gathered per destination entry;
//   factorise   right-looking tiled Cholesky in LDS; the trailing update runs on the matrix cores
//               (v_mfma_f64_16x16x4_f64); the right-hand side rides along as an extra matrix row;
//   step        dogleg step selection, PoseLocalParameterization::Plus
//               (pose_local_parameterization.cpp:3-19), candidate evaluation fused with the next
//               linearisation.
//
// Trust-region semantics restate upstream Ceres 2.x (not in the reference tree; SURVEY.md
// Appendix C).  All arithmetic is FP64.
#include <hip/hip_runtime.h>
#include "tcv_factors.h"
#include "tcv_packed.h"
#include "tcv_dev.h"
#include "tcv_gauge.h"

namespace tcv {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// ---- LDS tile addressing -------------------------------------------------------------------------
// element (r, c) of a 16x16 tile sits at r*16 + (c ^ (r & 14)): conflict-free for the MFMA operand
// reads (lane -> row l&15, column 4kk + (l>>4)), for the C-layout loads/stores and for row-per-lane
// column walks.

// optional per-phase cycle accounting (build with -DTCV_PROFILE): lane 0 adds s_memtime deltas to 32-bit counters in LDS (the
// spare half of the reduction scratch; a global read-modify-write per mark used to stall wave 0 for ~1 K cycles and inflated
// densely marked phases); the counters are flushed to the global table once per window
#ifdef TCV_PROFILE
#define TCV_MARK(C, id)                                                                        \
    do {                                                                                       \
        const long long t_ = clock64();                                                        \
        if ((C).tid == 0) {      /* the previous mark's time stamp lives in LDS (red + 56), wherever that mark was taken */ \
            typedef __attribute__((address_space(3))) long long lds_ll;                        \
            lds_ll *last_ = (lds_ll *)((C).red + 56);                                          \
            ((lds_u *)((C).red + 40))[id] += (unsigned)(t_ - *last_);                          \
            *last_ = t_;                                                                       \
        }                                                                                      \
    } while (0)
#else
#define TCV_MARK(C, id) do { } while (0)
#endif
// subtractive profiling (build with -DTCV_ABLATE, developer tool): bit k of SolveArgs::skip removes phase k from the kernel (results
// are garbage, every step is forced to be accepted so that the control flow stays the benchmark's); the time difference to skip = 0
// is what the phase really costs under the real overlap conditions -- no instrumentation in the timed code
// (-DTCV_ABLATE_CONST=mask, round 6: the same mask as a COMPILE-TIME constant -- the removed phases' code and registers are really gone, which is
// what a phase-split pipeline's kernels would look like to the register allocator: tools/r06_phase_split_bound.sh)
#if defined(TCV_ABLATE_CONST)
#define TCV_ABLATE 1
#define ABL(C, bit) (!((unsigned)(TCV_ABLATE_CONST) & (1u << (bit))))
#define ABL_FORCE(C) ((((unsigned)(TCV_ABLATE_CONST)) >> 31) != 0)
#define ABL_ACCEPT(C) (((((unsigned)(TCV_ABLATE_CONST)) >> 30) & 1u) != 0)
#elif defined(TCV_ABLATE)
#define ABL(C, bit) (!((C).skip & (1u << (bit))))
#define ABL_FORCE(C) (((C).skip >> 31) != 0)      // bit 31: linear-solver failures are ignored
#define ABL_ACCEPT(C) ((((C).skip >> 30) & 1u) != 0)      // bit 30: every step is accepted (fixed control flow)
#else
#define ABL(C, bit) true
#endif
enum { AB_VIS_EVAL = 0, AB_VIS_GATHER, AB_LM, AB_SCHUR, AB_PRIOR_A, AB_IMU_RAW, AB_IMU_WHITEN, AB_IMU_GATHER, AB_PRIOR_B, AB_FIN_SCALE, AB_FIN_PASS,
       AB_CHAIN_FWD, AB_CHOL, AB_BACK, AB_CHAIN_BWD, AB_LM_BACK, AB_DOGLEG, AB_PLUS, AB_NORMS, AB_SETUP, AB_CH_T, AB_CH_W, AB_CH_MFMA, AB_CH_FETCH,
       AB_COPY_PROG /* only together with the gathers */, AB_ZERO,
       AB_CH_HALF /* NOT a phase: the chain runs over every second step record only (blocks 0, 2, 4, ... of the elimination order: 6 of 11 steps): what a
                     cyclic-reduction order could save at most on the sequential part, before its first level is paid for */ };
enum { PH_SETUP = 0, PH_VIS_EVAL, PH_VIS_GATHER, PH_LM, PH_SCHUR, PH_ZERO, PH_IMU_RAW, PH_IMU_WHITEN, PH_IMU_GATHER, PH_PRIOR,
       PH_COST_RED, PH_FIN_SCALE, PH_FIN_CAUCHY, PH_FIN_PASS, PH_CHOL_DIAG, PH_CHOL_TRSM, PH_CHOL_UPD, PH_BACK, PH_LM_BACK,
       PH_DOGLEG, PH_PLUS, PH_NORMS, PH_OTHER, PH_CHAIN_FWD, PH_CHAIN_BWD, PH_CH_A, PH_CH_B, PH_CH_C, PH_CH_D, PH_COUNT = 32 };

// (-DTCV_CAMW_CONST: the camera-vector width as a literal -- the chain and cooperative translation units, build.py; windows with a relocalisation
// pose go to the ProjectionTdFactor instance, which reads the width from the plan)
#ifdef TCV_CAMW_CONST
#define TCV_CAMW(P) ((int)CAM_W)
#else
#define TCV_CAMW(P) ((P).camw)
#endif
template <int NT>
struct Ctx {
    gbl_d *prof;
    long long t_last;
    // plan / data
    cst_plan *P;
    cst_i *ip;       // plan ints
    cst_d *dp;       // window doubles
    cst_win *W;
    // LDS
    lds_d *tiles, *stage, *xs, *xc, *sc, *ycam, *invdiag, *red, *area;
    lds_d *gcam;     // camera part of the gradient J'r; shares the slot of invdiag (g is dead once the rhs row is written)
    lds_d *rc, *sd;  // Schur corrections of the rhs and of the diagonal (pose part only, < 88 entries each)
    lds_i *flag;
    lds_d *hd;       // chain mode: diagonal of J'J for the Euclidean camera blocks (index t - npp)
    gbl_d *g_imublk, *g_spill;   // chain mode: per-factor J'J | J'r blocks (32 x 32) and the factored fronts
    int nd, ntd;     // dimension and tile rows of the dense system held in the tiles (dense mode: nc / nt, chain mode: npp / nt_c)
    // global scratch (per workgroup)
    gbl_d *v_s, *v_g, *v_D, *v_ghat, *v_y, *v_p, *v_rc, *v_sd, *l_hll, *l_gl, *l_invk, *g_hcl, *g_hp, *g_pr, *g_pdx, *g_sqrt;
    int ntiles, stage_cap;
    int tid;
    unsigned skip;   // TCV_ABLATE builds
    // cooperative mode (tcv_packed.h COOP_*): the group's control block, state hand-off and chunk exports; helpers per group
    gbl_i *cx_ctl;
    gbl_d *cx_x, *cx_exp;
    int cx_h, cx_exp_stride, cx_seq;
    long long cx_timeout;
};

// Every pointer and size in Ctx is the same for all lanes of the workgroup, but the phase functions are not inlined (their register
// budgets would add up) and take the Ctx by reference: the compiler then reloads each field from the kernel's stack frame with FLAT
// loads into VGPRs, cannot tell that the values are uniform, and ends up with chains scratch load -> wait -> global load of a plan
// field -> wait -> use all over the phases (and with 64-bit pointers in VGPRs that spill).  uniform_ctx() copies the Ctx once per
// call through v_readfirstlane: pointers and sizes live in SGPRs from then on, loads through the constant-address-space pointers
// (plan header, window header, uniform plan ints) become scalar loads, and the per-lane ones take an SGPR base.
template <class T> __device__ __forceinline__ T *uni_ptr(T *p) {      // 64-bit (generic / global / constant) pointers
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ gbl_d *uni_ptr(gbl_d *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (gbl_d *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ cst_plan *uni_ptr(cst_plan *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (cst_plan *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ cst_win *uni_ptr(cst_win *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (cst_win *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ cst_i *uni_ptr(cst_i *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (cst_i *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ cst_d *uni_ptr(cst_d *p) {
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return (cst_d *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ lds_d *uni_ptr(lds_d *p) { return (lds_d *)(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p); }      // LDS pointers are 32-bit
__device__ __forceinline__ lds_i *uni_ptr(lds_i *p) { return (lds_i *)(unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)p); }
// a value every lane holds identically (the result of a block reduction, a trust-region scalar): moved to a scalar register pair so that
// it is kept across the calls of the phase functions by v_writelane / v_readlane instead of a scratch spill
__device__ __forceinline__ double uni_d(double v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u2 u = __builtin_bit_cast(u2, v);
    u2 r;
    r.x = __builtin_amdgcn_readfirstlane(u.x);
    r.y = __builtin_amdgcn_readfirstlane(u.y);
    return __builtin_bit_cast(double, r);
}
template <int NT>
__device__ __forceinline__ Ctx<NT> uniform_ctx(const Ctx<NT> &R) {
    Ctx<NT> C;
    C.prof = uni_ptr(R.prof); C.t_last = R.t_last;
    C.P = uni_ptr(R.P); C.ip = uni_ptr(R.ip); C.dp = uni_ptr(R.dp); C.W = uni_ptr(R.W);
    C.tiles = uni_ptr(R.tiles); C.stage = uni_ptr(R.stage); C.xs = uni_ptr(R.xs); C.xc = uni_ptr(R.xc); C.sc = uni_ptr(R.sc);
    C.ycam = uni_ptr(R.ycam); C.invdiag = uni_ptr(R.invdiag); C.red = uni_ptr(R.red); C.area = uni_ptr(R.area); C.gcam = uni_ptr(R.gcam);
    C.rc = uni_ptr(R.rc); C.sd = uni_ptr(R.sd); C.flag = uni_ptr(R.flag); C.hd = uni_ptr(R.hd);
    C.g_imublk = uni_ptr(R.g_imublk); C.g_spill = uni_ptr(R.g_spill);
    C.nd = __builtin_amdgcn_readfirstlane(R.nd); C.ntd = __builtin_amdgcn_readfirstlane(R.ntd);
    C.v_s = uni_ptr(R.v_s); C.v_g = uni_ptr(R.v_g); C.v_D = uni_ptr(R.v_D); C.v_ghat = uni_ptr(R.v_ghat); C.v_y = uni_ptr(R.v_y); C.v_p = uni_ptr(R.v_p);
    C.v_rc = uni_ptr(R.v_rc); C.v_sd = uni_ptr(R.v_sd); C.l_hll = uni_ptr(R.l_hll); C.l_gl = uni_ptr(R.l_gl); C.l_invk = uni_ptr(R.l_invk);
    C.g_hcl = uni_ptr(R.g_hcl); C.g_hp = uni_ptr(R.g_hp); C.g_pr = uni_ptr(R.g_pr); C.g_pdx = uni_ptr(R.g_pdx); C.g_sqrt = uni_ptr(R.g_sqrt);
    C.ntiles = __builtin_amdgcn_readfirstlane(R.ntiles); C.stage_cap = __builtin_amdgcn_readfirstlane(R.stage_cap);
    C.tid = R.tid;
    C.skip = __builtin_amdgcn_readfirstlane(R.skip);
    C.cx_ctl = (gbl_i *)uni_ptr((gbl_d *)R.cx_ctl); C.cx_x = uni_ptr(R.cx_x); C.cx_exp = uni_ptr(R.cx_exp);
    C.cx_h = __builtin_amdgcn_readfirstlane(R.cx_h); C.cx_exp_stride = __builtin_amdgcn_readfirstlane(R.cx_exp_stride);
    C.cx_seq = __builtin_amdgcn_readfirstlane(R.cx_seq); C.cx_timeout = R.cx_timeout;
    return C;
}
enum { SCR_HP = 8256, SCR_SQ = 16 * 225, SCR_LM = 1024 };      // the landmark/camera coupling store comes last: its size is per batch
// The phase functions of the single-workgroup kernels take the context IN REGISTERS: seventeen scalars (25 dwords) as ordinary arguments,
// from which everything else follows (the scratch vectors sit at fixed offsets behind v_s, the LDS vectors behind xs).  Passed by reference
// the 79-dword Ctx was re-read from the stack frame at every entry -- 79 per-lane scratch loads per wavefront and call, ~60 calls per window
// solve: as much L1 / L2 read traffic as everything else the kernel fetches, and a full memory wait in front of every phase.
#define TCV_CTX_PARAMS cst_plan *aP_, cst_i *aip_, cst_d *adp_, cst_win *aW_, gbl_d *ascr_, gbl_d *aimu_, gbl_d *aspill_, gbl_d *aprof_, \
                       lds_d *atiles_, lds_d *astage_, lds_d *axs_, int and_, int antd_, int antiles_, int astagecap_, int atid_, unsigned askip_
#define TCV_CTX_ARGS(K) (K).P, (K).ip, (K).dp, (K).W, (K).v_s, (K).g_imublk, (K).g_spill, (K).prof, (K).tiles, (K).stage, (K).xs, (K).nd, (K).ntd, \
                        (K).ntiles, (K).stage_cap, (K).tid, (K).skip
#define TCV_CTX_FORWARD aP_, aip_, adp_, aW_, ascr_, aimu_, aspill_, aprof_, atiles_, astage_, axs_, and_, antd_, antiles_, astagecap_, atid_, askip_
// global scratch behind v_s, LDS vectors behind xs: the ONE place that knows the layout (the kernels call these too)
template <int NT>
__device__ __forceinline__ void ctx_scratch_layout(Ctx<NT> &C, gbl_d *scr) {
    C.v_s = scr; C.v_g = scr + SCR_NL; C.v_D = scr + 2 * SCR_NL; C.v_ghat = scr + 3 * SCR_NL; C.v_y = scr + 4 * SCR_NL;
    C.v_p = scr + 5 * SCR_NL; C.v_rc = scr + 6 * SCR_NL; C.v_sd = scr + 7 * SCR_NL;
    C.l_hll = scr + 8 * SCR_NL; C.l_gl = C.l_hll + SCR_LM; C.l_invk = C.l_gl + SCR_LM;
    C.g_hp = C.l_invk + SCR_LM; C.g_pr = C.g_hp + SCR_HP; C.g_pdx = C.g_pr + 128;
    C.g_sqrt = C.g_pdx + 128;
    C.g_hcl = C.g_sqrt + SCR_SQ;
}
template <int NT>
__device__ __forceinline__ void ctx_lds_layout(Ctx<NT> &C, lds_d *xs, int nxl, bool chain, int c_stage_cap, int camw) {
    // camw (PlanHdr::camw): width of the vectors over the camera tangent space -- 176 for every window of OptimizationWithLine, 184 when a 12th
    // pose block (the relocalisation pose, estimator.cpp:1854-1886) takes the camera side to 177 dims; rc | sd cover the pose part only (< 88 each)
    lds_d *p = xs;
    C.xs = p; p += nxl;
    C.xc = p; p += nxl;
    C.sc = p; p += camw;
    C.rc = p; C.sd = p + 88; p += 176;
    C.ycam = p; p += camw;
    C.invdiag = p; C.gcam = p; p += camw;
    C.red = p; p += 64;
    C.flag = (lds_i *)(C.red + 62);   // red[] uses at most 5 * NT/64 = 40 doubles; 40..55 hold the profile build's counters, 56 its last time stamp
    C.hd = p;                         // chain mode only (112 doubles)
    C.area = chain ? C.stage + c_stage_cap : p;
}
template <int NT>
__device__ __forceinline__ Ctx<NT> ctx_from_args(TCV_CTX_PARAMS) {
    Ctx<NT> C;
    C.P = uni_ptr(aP_); C.ip = uni_ptr(aip_); C.dp = uni_ptr(adp_); C.W = uni_ptr(aW_);
    C.g_imublk = uni_ptr(aimu_); C.g_spill = uni_ptr(aspill_); C.prof = uni_ptr(aprof_); C.t_last = 0;
    C.tiles = uni_ptr(atiles_); C.stage = uni_ptr(astage_);
    C.nd = __builtin_amdgcn_readfirstlane(and_); C.ntd = __builtin_amdgcn_readfirstlane(antd_);
    C.ntiles = __builtin_amdgcn_readfirstlane(antiles_); C.stage_cap = __builtin_amdgcn_readfirstlane(astagecap_);
    C.tid = atid_;
    C.skip = __builtin_amdgcn_readfirstlane(askip_);
    ctx_scratch_layout<NT>(C, uni_ptr(ascr_));
    cst_plan &P = *C.P;
    ctx_lds_layout<NT>(C, uni_ptr(axs_), (P.nx + P.nland + 1) & ~1, P.chain != 0, P.c_stage_cap, TCV_CAMW(P));
    C.cx_ctl = nullptr; C.cx_x = nullptr; C.cx_exp = nullptr; C.cx_h = 0; C.cx_exp_stride = 0; C.cx_seq = 0; C.cx_timeout = 0;
    return C;
}

enum {
    SCR_TOTAL = 8 * SCR_NL + 3 * SCR_LM + SCR_HP + 2 * 128 + SCR_SQ      // + hcl capacity (tcv_batch_create)
};

template <int NT>
__device__ __forceinline__ double block_sum(double v, lds_d *red, int tid) {
    v = wave_sum_down(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; w++) t += red[w];
    return t;
}

template <int NT, int N>
__device__ __forceinline__ void block_sum_n(double (&v)[N], lds_d *red, int tid) {
#pragma unroll
    for (int k = 0; k < N; k++)
        v[k] = wave_sum_down(v[k]);
    __syncthreads();
    if ((tid & 63) == 0)
#pragma unroll
        for (int k = 0; k < N; k++) red[(tid >> 6) * N + k] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) t += red[w * N + k];
        v[k] = t;
    }
}

// ---- row-unit gathers (tcv_packed.h): acc[e] += sum_rows rec[row][colA + ea] * rec[row][colB + e] ------------------
__device__ __forceinline__ void vis_item1(const lds_d *stage, unsigned it, int ea, double (&acc)[6], int pstride) {
    const int type = it & 1, cb = (it >> 1) & 31, ca = (it >> 6) & 31, base = it >> 11;
    const int ld = type ? LINE_STRIDE : pstride;
    const lds_d *rec = stage + base;
    const double a0 = rec[ca + ea], a1 = rec[ld + ca + ea];
    double b0[6], b1[6];
#pragma unroll
    for (int e = 0; e < 6; e++) { b0[e] = rec[cb + e]; b1[e] = rec[ld + cb + e]; }
#pragma unroll
    for (int e = 0; e < 6; e++) acc[e] += a0 * b0[e] + a1 * b1[e];
}
// Four items per trip: all item words, then all operands, are in flight before the first FMA (one wave per SIMD
// cannot hide LDS latency by switching waves, so the loads are batched by hand).
__device__ __forceinline__ void vis_items(const lds_d *stage, const lds_i *items, int k0, int k1, int kstep, int ea, double (&acc)[6], int pstride) {
    int k = k0;
    for (; k + 3 * kstep < k1; k += 4 * kstep) {
        unsigned it[4];
#pragma unroll
        for (int j = 0; j < 4; j++) it[j] = (unsigned)items[k + j * kstep];
        double a0[4], a1[4], b0[4][6], b1[4][6];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int type = it[j] & 1, cb = (it[j] >> 1) & 31, ca = (it[j] >> 6) & 31, base = it[j] >> 11;
            const int ld = type ? LINE_STRIDE : pstride;
            const lds_d *rec = stage + base;
            a0[j] = rec[ca + ea]; a1[j] = rec[ld + ca + ea];
#pragma unroll
            for (int e = 0; e < 6; e++) { b0[j][e] = rec[cb + e]; b1[j][e] = rec[ld + cb + e]; }
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int e = 0; e < 6; e++) acc[e] += a0[j] * b0[j][e] + a1[j] * b1[j][e];
    }
    for (; k < k1; k += kstep) vis_item1(stage, (unsigned)items[k], ea, acc, pstride);
}

// Schur item: hoff << 18 | nslot << 12 | slotA << 6 | slotB   (slotA = 63: the landmark's gl / kappa instead of Hcl[slotA])
__device__ __forceinline__ void schur_items(const lds_d *hcl, const lds_i *items, int k0, int k1, int kstep, int ea, double (&acc)[6]) {
    int k = k0;
    for (; k + 3 * kstep < k1; k += 4 * kstep) {
        unsigned v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = (unsigned)items[k + j * kstep];
        double w0[4], w1[4], hb[4][6];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int sb = v[j] & 63, sa = (v[j] >> 6) & 63, ns = (v[j] >> 12) & 63, hoff = v[j] >> 18;
            const lds_d *h = hcl + hoff;
            w0[j] = (sa == 63) ? 1.0 : h[6 * sa + ea];
            w1[j] = (sa == 63) ? h[6 * ns + 1] : h[6 * ns];
#pragma unroll
            for (int e = 0; e < 6; e++) hb[j][e] = h[6 * sb + e];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const double w = w0[j] * w1[j];
#pragma unroll
            for (int e = 0; e < 6; e++) acc[e] += w * hb[j][e];
        }
    }
    for (; k < k1; k += kstep) {
        const unsigned v = (unsigned)items[k];
        const int sb = v & 63, sa = (v >> 6) & 63, ns = (v >> 12) & 63, hoff = v >> 18;
        const lds_d *h = hcl + hoff;
        const double w = ((sa == 63) ? 1.0 : h[6 * sa + ea]) * ((sa == 63) ? h[6 * ns + 1] : h[6 * ns]);
#pragma unroll
        for (int e = 0; e < 6; e++) acc[e] += w * h[6 * sb + e];
    }
}

// window data (read-only, HBM/L2) -> LDS, four independent 8-byte loads per thread in flight
template <int NT>
__device__ __forceinline__ void copy_doubles(lds_d *dst, cst_d *src, int n, int tid) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) v2d cst_v2d;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    const int n2 = n >> 1;                      // src and dst are 16-byte aligned (even double offsets, tcv_pack.cpp)
    cst_v2d *s2 = (cst_v2d *)src;
    lds_v2d *d2 = (lds_v2d *)dst;
    for (int i = tid; i < n2; i += 4 * NT) {
        v2d v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + k * NT < n2) v[k] = s2[i + k * NT];
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + k * NT < n2) d2[i + k * NT] = v[k];
    }
    if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

// deep version for data that comes from HBM rather than L2 (the prior's J0, 45 KB per window and linearisation): ten 16-byte
// loads per thread in flight, so that a 39 KB piece costs ONE memory round trip instead of five
template <int NT>
__device__ __forceinline__ void copy_doubles_deep(lds_d *dst, cst_d *src, int n, int tid) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) v2d cst_v2d;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    const int n2 = n >> 1;                      // src and dst 16-byte aligned
    cst_v2d *s2 = (cst_v2d *)src;
    lds_v2d *d2 = (lds_v2d *)dst;
    for (int i = tid; i < n2; i += 10 * NT) {
        v2d v[10];
#pragma unroll
        for (int k = 0; k < 10; k++) v[k] = s2[min(i + k * NT, n2 - 1)];
#pragma unroll
        for (int k = 0; k < 10; k++) if (i + k * NT < n2) d2[i + k * NT] = v[k];
    }
    if ((n & 1) && tid == 0) dst[n - 1] = src[n - 1];
}

// zero n doubles (n even, 16-byte aligned) with 16-byte LDS stores
template <int NT>
__device__ __forceinline__ void zero_lds(lds_d *dst, int n, int tid) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    lds_v2d *d2 = (lds_v2d *)dst;
    const v2d z = {0.0, 0.0};
    for (int i = tid; i < (n >> 1); i += NT) d2[i] = z;
}

template <int N>
__device__ __forceinline__ void wave_sum(double (&acc)[N]) {
#pragma unroll
    for (int e = 0; e < N; e++)
        acc[e] = wave_sum_down(acc[e]);
}

// global (plan, L2-resident) -> LDS copy of a gather program: 16-byte loads, four in flight per thread
template <int NT>
__device__ __forceinline__ void copy_prog(lds_i *dst, cst_i *src, int n, int tid) {
    const int n4 = n >> 2;
    cst_v4i *s4 = (cst_v4i *)src;      // plans and every program inside them start 16-byte aligned (tcv_pack.cpp)
    lds_v4i *d4 = (lds_v4i *)dst;
    for (int i = tid; i < n4; i += 4 * NT) {
        v4i v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + k * NT < n4) v[k] = s4[i + k * NT];
#pragma unroll
        for (int k = 0; k < 4; k++) if (i + k * NT < n4) d4[i + k * NT] = v[k];
    }
    for (int i = (n4 << 2) + tid; i < n; i += NT) dst[i] = src[i];
}


// ---- one visual chunk (point + line factors whose records fit the LDS staging area together), first half: evaluation into the staging
// records and the destination-driven gather of J'J / J'r / the landmark couplings.  cost_pt / cost_ln take the chunk's robustified
// costs of this thread's point and line factor (the single-workgroup kernel passes its one accumulator for both).
template <int NT, bool CHAIN, bool TD = !CHAIN>
__device__ __forceinline__ void vis_part1(Ctx<NT> &C, const lds_d *x, int ch, bool assemble, double &cost_pt, double &cost_ln) {
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    cst_d *dp = C.dp;
    const int nx = P.nx;
    cst_d *misc = dp + C.W->d_misc;
    const double proj_sqrt = misc[3], proj_loss = misc[4], line_loss = misc[5];
    const bool line_exact = misc[8] != 0.0;
    const bool with_td = TD && (P.flags & 1);
    const int prec = with_td ? (int)PROJ_TD_REC : (int)PROJ_REC, pstr = with_td ? (int)PROJ_TD_STRIDE : (int)PROJ_STRIDE;
    cst_i *blk = ip + P.o_blk;
    {
        cst_i *vc = ip + P.o_vchunk + ch * 16;
        const int pb = vc[0], pn = vc[1], lb = vc[2], ln = vc[3], voff = vc[4], vnu = vc[5], vnw = vc[6], vni = vc[7];
        const int lmb = vc[8], lmn = vc[9], ebase = vc[10], esize = vc[11], soff = vc[12], snu = vc[13], snw = vc[14], sni = vc[15];
        lds_d *hcl = C.area, *hll = C.area + esize, *gl = hll + lmn;
        lds_i *lprog = (lds_i *)(C.stage + ((pn * prec + ln * LINE_REC + 1) & ~1));   // gather program behind the records
        if (assemble) {
            for (int i = tid; i < esize + 2 * lmn; i += NT) C.area[i] = 0.0;
            if (ABL(C, AB_COPY_PROG)) copy_prog<NT>(lprog, ip + P.o_vdest + voff, 3 * vnu + vni, tid);
        }
        __syncthreads();
        if (ABL(C, AB_VIS_EVAL)) {
        for (int f = tid; f < pn; f += NT) {
            cst_i *pf = ip + P.o_proj + (pb + f) * 4;
            const lds_d *xi = x + blk[pf[0] * 4 + 1], *xj = x + blk[pf[1] * 4 + 1], *xe = x + blk[pf[2] * 4 + 1];
            const double lam = x[nx + pf[3]];
            lds_d *rec = C.stage + f * prec;
            double r[2];
            if (with_td) {      // ProjectionTdFactor (projection_td_factor.cpp:34-140): 20 Jacobian columns, Td last
                const double tdv = x[blk[P.td_cam * 4 + 1]];
                double pts[6], aux[8], Jl[40];
#pragma unroll
                for (int i = 0; i < 6; i++) pts[i] = dp[C.W->d_proj + (pb + f) * 14 + i];
#pragma unroll
                for (int i = 0; i < 8; i++) aux[i] = dp[C.W->d_proj + (pb + f) * 14 + 6 + i];
                proj_td_eval(CGEN(xi), CGEN(xj), CGEN(xe), lam, tdv, pts, aux, proj_sqrt, misc[6], misc[7], r, assemble ? Jl : nullptr, 20);
                cost_pt += loss_correct2(r, assemble ? Jl : nullptr, 20, 20, proj_loss);
                if (assemble) {
#pragma unroll
                    for (int row = 0; row < 2; row++) {
#pragma unroll
                        for (int c2 = 0; c2 < 19; c2++) rec[row * PROJ_TD_STRIDE + c2] = Jl[row * 20 + c2];
                        rec[row * PROJ_TD_STRIDE + 19] = r[row];
                        rec[row * PROJ_TD_STRIDE + 20] = Jl[row * 20 + 19];
#pragma unroll
                        for (int c2 = 21; c2 < 26; c2++) rec[row * PROJ_TD_STRIDE + c2] = 0.0;
                    }
                }
                continue;
            }
            double *J = assemble ? GEN(rec) : nullptr;
            double pts[6];
#pragma unroll
            for (int i = 0; i < 6; i++) pts[i] = dp[C.W->d_proj + (pb + f) * 6 + i];
            proj_eval(CGEN(xi), CGEN(xj), CGEN(xe), lam, pts, proj_sqrt, r, J, PROJ_STRIDE);
            cost_pt += loss_correct2(r, J, 19, PROJ_STRIDE, proj_loss);
            if (assemble) { rec[19] = r[0]; rec[PROJ_STRIDE + 19] = r[1]; }
        }
        // the line factors of the chunk run on the third wavefront while the first two evaluate the point factors; their costs go through
        // the (idle) reduction scratch back to the threads that used to evaluate them, so that the block sum adds the same numbers in the
        // same order
        const bool lines_aside = NT >= 192 && ln <= 40 && pn <= 128;
        const int lt0 = lines_aside ? 128 : 0;
        for (int f = tid - lt0; f >= 0 && f < ln; f += NT) {
            const int b = ip[P.o_line + lb + f];
            const lds_d *xp = x + blk[b * 4 + 1];
            lds_d *rec = C.stage + pn * prec + f * LINE_REC;
            double r[2];
            double *J = assemble ? GEN(rec) : nullptr;
            double ld9[9], lc[21];
#pragma unroll
            for (int i = 0; i < 9; i++) ld9[i] = dp[C.W->d_line + (lb + f) * 9 + i];
#pragma unroll
            for (int i = 0; i < 21; i++) lc[i] = dp[C.W->d_linec + i];
            line_eval(CGEN(xp), ld9, lc, lc + 9, lc + 18, r, J, LINE_STRIDE, line_exact);
            const double lcost = loss_correct2(r, J, 6, LINE_STRIDE, line_loss);
            if (lines_aside) C.red[f] = lcost; else cost_ln += lcost;
            if (assemble) { rec[6] = r[0]; rec[LINE_STRIDE + 6] = r[1]; }
        }
        }
        if (!assemble) {
            if (ABL(C, AB_VIS_EVAL) && NT >= 192 && ln <= 40 && pn <= 128 && ln > 0) {
                __syncthreads();
                if (tid < ln) cost_ln += C.red[tid];
                __syncthreads();
            }
            TCV_MARK(C, PH_VIS_EVAL); return;
        }
        __syncthreads();
        if (ABL(C, AB_VIS_EVAL) && NT >= 192 && ln <= 40 && pn <= 128 && tid < ln) cost_ln += C.red[tid];
        TCV_MARK(C, PH_VIS_EVAL);
        // gather J'J / J'r / landmark couplings: wave units (long item lists) first, then one unit per thread
        if (ABL(C, AB_VIS_GATHER)) {
            const lds_i *items = lprog + 3 * vnu;
            const int lane = tid & 63, wave = tid >> 6;
            const int nloop = vnw + ((vnu - vnw + NT - 1) / NT) * 1;   // (only for clarity; loops below are separate)
            (void)nloop;
            // wave units (long lists) go to the wavefronts other than the first, which holds the 64 longest thread units of the sorted
            // list and is the critical one: every unit has its own destination, so who accumulates it does not change a bit
            constexpr int NWU = NT / 64 > 1 ? NT / 64 - 1 : 1;
            for (int uu = 0; uu < 2; uu++) {
                const bool wv = (uu == 0);
                for (int u = wv ? (NT / 64 > 1 ? (wave == 0 ? vnw : NT / 64 - 1 - wave) : 0) : vnw + tid; u < (wv ? vnw : vnu); u += (wv ? NWU : NT)) {
                    const unsigned u0 = (unsigned)lprog[3 * u], u1 = (unsigned)lprog[3 * u + 1];
                    const int ib = lprog[3 * u + 2];
                    const int kind = u0 >> 28, ncols = (u0 >> 24) & 15, ea = (u0 >> 20) & 15, n = u0 & 0xfffff;
                    const int o0 = u1 >> 16, o1 = u1 & 0xffff;
                    double acc[6] = {0, 0, 0, 0, 0, 0};
                    if (wv) { vis_items(C.stage, items, ib + lane, ib + n, 64, ea, acc, pstr); wave_sum<6>(acc); if (lane != 0) continue; }
                    else vis_items(C.stage, items, ib, ib + n, 1, ea, acc, pstr);
                    if (kind == DK_TILE) {
#pragma unroll
                        for (int e = 0; e < 6; e++) if (e < ncols) C.tiles[tix(o0 + ea, o1 + e)] += acc[e];
                    } else if (kind == DK_G) {
#pragma unroll
                        for (int e = 0; e < 6; e++) if (e < ncols) C.gcam[o0 + e] += acc[e];
                    } else if (kind == DK_HCL) {
#pragma unroll
                        for (int e = 0; e < 6; e++) hcl[o0 - ebase + e] += acc[e];
                    } else { hll[o0 - lmb] += acc[0]; gl[o0 - lmb] += acc[1]; }
                }
            }
        }
        __syncthreads();
        TCV_MARK(C, PH_VIS_GATHER);
    }
}

// ---- second half: the chunk's landmark pivots and their Schur complement on the pose tiles (tiles -= ..., sd / rc += ...)
template <int NT, bool CHAIN>
__device__ __forceinline__ void vis_part2(Ctx<NT> &C, int ch, bool first, double mu) {
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    const int nc = P.nc;
    {
        cst_i *vc = ip + P.o_vchunk + ch * 16;
        const int lmb = vc[8], lmn = vc[9], ebase = vc[10], esize = vc[11], soff = vc[12], snu = vc[13], snw = vc[14], sni = vc[15];
        lds_d *hcl = C.area, *hll = C.area + esize, *gl = hll + lmn;
        // landmark pivots of this chunk; keep copies for the Cauchy point and the back-substitution.  The staging area
        // is dead now: the Schur program moves in.
        {
            cst_i *lm = ip + P.o_lm;
            for (int l = tid; l < (ABL(C, AB_LM) ? lmn : 0); l += NT) {
                const double h = hll[l];
                double sl;
                if (first) { sl = 1.0 / (1.0 + sqrt(h)); C.v_s[nc + lmb + l] = sl; }
                else sl = C.v_s[nc + lmb + l];
                const double d2 = fmin(fmax(sl * sl * h, 1e-6), 1e32);
                const double kappa = h + mu * d2 / (sl * sl);
                const double ik = 1.0 / kappa;
                const int eo = lm[2 * (lmb + l)] - ebase, ns = lm[2 * (lmb + l) + 1];
                hcl[eo + 6 * ns] = ik;
                hcl[eo + 6 * ns + 1] = gl[l] * ik;
                C.l_hll[lmb + l] = h; C.l_gl[lmb + l] = gl[l]; C.l_invk[lmb + l] = ik;
                C.v_g[nc + lmb + l] = gl[l];
            }
            if (ABL(C, AB_COPY_PROG)) copy_prog<NT>((lds_i *)C.stage, ip + P.o_sdest + soff, 3 * snu + sni, tid);
        }
        __syncthreads();
        for (int i = tid; i < esize; i += NT) C.g_hcl[ebase + i] = hcl[i];
        TCV_MARK(C, PH_LM);
        // Schur complement of the chunk's landmarks
        if (ABL(C, AB_SCHUR)) {
            const lds_i *sprog = (const lds_i *)C.stage;
            const lds_i *items = sprog + 3 * snu;
            const int lane = tid & 63, wave = tid >> 6;
            constexpr int NWU = NT / 64 > 1 ? NT / 64 - 1 : 1;      // wave units off the first wavefront, as in the visual gather
            for (int uu = 0; uu < 2; uu++) {
                const bool wv = (uu == 0);
                for (int u = wv ? (NT / 64 > 1 ? (wave == 0 ? snw : NT / 64 - 1 - wave) : 0) : snw + tid; u < (wv ? snw : snu); u += (wv ? NWU : NT)) {
                    const unsigned u0 = (unsigned)sprog[3 * u], u1 = (unsigned)sprog[3 * u + 1];
                    const int ib = sprog[3 * u + 2];
                    const int kind = u0 >> 28, ncols = (u0 >> 24) & 15, ea = (u0 >> 20) & 15, n = u0 & 0xfffff;
                    const int o0 = u1 >> 16, o1 = u1 & 0xffff;
                    double acc[6] = {0, 0, 0, 0, 0, 0};
                    if (wv) { schur_items(hcl, items, ib + lane, ib + n, 64, ea, acc); wave_sum<6>(acc); if (lane != 0) continue; }
                    else schur_items(hcl, items, ib, ib + n, 1, ea, acc);
                    if (kind == DK_TILE) {
#pragma unroll
                        for (int e = 0; e < 6; e++)
                            if (e < ncols) {
                                C.tiles[tix(o0 + ea, o1 + e)] -= acc[e];
                                if (o0 + ea == o1 + e) C.sd[o0 + ea] += acc[e];
                            }
                    } else {
#pragma unroll
                        for (int e = 0; e < 6; e++) C.rc[o0 + e] += acc[e];
                    }
                }
            }
        }
        __syncthreads();
        TCV_MARK(C, PH_SCHUR);
    }
}

// ---- IMU factors of one chunk, first half: raw residual / Jacobian (lanes = factors, waves 0..3 = the four parts of imu_raw_part), whitening
// J = sqrt_info * J_raw on the matrix cores (one wavefront per factor), cost of this thread's factor
template <int NT, bool CHAIN>
__device__ __forceinline__ void imu_part1(Ctx<NT> &C, const lds_d *x, int ch, bool assemble, double &cost_imu) {
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    cst_d *dp = C.dp;
    cst_i *blk = ip + P.o_blk;
    (void)dp; (void)blk;
    {
        cst_i *ic = ip + P.o_ichunk + ch * 4;
        const int fb = ic[0], fn = ic[1], ncolor = ic[2];
        const unsigned colorbits = (unsigned)ic[3];
        lds_d *recs = CHAIN ? C.stage : C.area;      // chain mode: the whole LDS pool holds the records
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int NW = NT / 64;
        (void)ncolor; (void)colorbits; (void)NW; (void)lane; (void)wave; (void)fb;
        cst_d *misc = dp + C.W->d_misc;
        const double G3[3] = {misc[0], misc[1], misc[2]};
        if (wave < 4 && lane < fn && ABL(C, AB_IMU_RAW)) {
            cst_i *b = ip + P.o_imu + (fb + lane) * 4;
            lds_d *rec = recs + lane * IMU_REC;
            double cst[62];
#pragma unroll
            for (int i = 0; i < 62; i++) cst[i] = dp[C.W->d_imu + (fb + lane) * IMU_CONST + i];
            imu_raw_part(wave, CGEN(x + blk[b[0] * 4 + 1]), CGEN(x + blk[b[1] * 4 + 1]), CGEN(x + blk[b[2] * 4 + 1]),
                         CGEN(x + blk[b[3] * 4 + 1]), cst, G3, GEN(rec), IMU_STRIDE_J, assemble);
        }
        __syncthreads();
        TCV_MARK(C, PH_IMU_RAW);
        if (ABL(C, AB_IMU_WHITEN)) {   // whiten: T = S * [J_raw | r_raw]  (S upper triangular 15 x 15, zero padded to 16 x 16)
            const int i16 = lane & 15, k4 = lane >> 4;
            for (int f = wave; f < fn; f += NW) {
                lds_d *rec = recs + f * IMU_REC;
                const gbl_d *S = C.g_sqrt + (fb + f) * 225;
                double sa[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) { const int k = 4 * kk + k4; const double t = S[min(i16, 14) * 15 + min(k, 14)]; sa[kk] = (i16 < 15 && k < 15) ? t : 0.0; }
                for (int ct = assemble ? 0 : 1; ct < 2; ct++) {
                    const int col = 16 * ct + i16;
                    double bb[4];
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) { const int k = 4 * kk + k4; const double t = rec[min(k, 14) * IMU_STRIDE_J + min(col, 30)]; bb[kk] = (k < 15 && col < 31) ? t : 0.0; }
                    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[kk], bb[kk], acc, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; i++) { const int row = k4 + 4 * i; if (row < 15 && col < 31) rec[row * IMU_STRIDE_J + col] = acc[i]; }
                }
            }
        }
        __syncthreads();
        TCV_MARK(C, PH_IMU_WHITEN);
        if (tid < fn) {
            const lds_d *rec = recs + tid * IMU_REC + 30;
            double s = 0;
#pragma unroll
            for (int r = 0; r < 15; r++) s += rec[r * IMU_STRIDE_J] * rec[r * IMU_STRIDE_J];
            cost_imu += 0.5 * s;
        }
    }
}

// ---- second half: J'J / J'r per factor on the matrix cores, scattered into the reduced camera system colour by colour (factors of one
// colour share no block)
template <int NT, bool CHAIN>
__device__ __forceinline__ void imu_part2(Ctx<NT> &C, int ch, bool assemble) {
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    cst_d *dp = C.dp;
    cst_i *blk = ip + P.o_blk;
    (void)dp; (void)blk;
    {
        cst_i *ic = ip + P.o_ichunk + ch * 4;
        const int fb = ic[0], fn = ic[1], ncolor = ic[2];
        const unsigned colorbits = (unsigned)ic[3];
        lds_d *recs = CHAIN ? C.stage : C.area;      // chain mode: the whole LDS pool holds the records
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int NW = NT / 64;
        (void)ncolor; (void)colorbits; (void)NW; (void)lane; (void)wave; (void)fb;
        if (assemble && ABL(C, AB_IMU_GATHER)) {
            const int i16 = lane & 15, k4 = lane >> 4;
            for (int color = 0; color < ncolor; color++) {
                // factors of this colour: whole factors go round-robin to the wavefronts; the ones left over when their number is not a
                // multiple of the wavefront count are split by J'J tile (the four tiles of a factor have disjoint destinations), one tile per
                // wavefront -- five factors on four wavefronts take 1.25 factor times instead of 2.  Same values, same additions.
                int ncf = 0;
                for (int f = 0; f < fn; f++) ncf += ((int)((colorbits >> (2 * f)) & 3u) == color) ? 1 : 0;
                const int nfull = (ncf / NW) * NW;
                int slot = 0;
                for (int f = 0; f < fn; f++) {
                    if ((int)((colorbits >> (2 * f)) & 3u) != color) continue;
                    const int sl = slot++;
                    const lds_d *rec = recs + f * IMU_REC;
                    if (sl >= nfull) {      // split factor: this wavefront's tile
                        if (wave >= 4) continue;
                        const int tile = (wave + (sl - nfull)) & 3;
                        const int I = (tile == 1 || tile == 2) ? 1 : 0, J = (tile >= 2) ? 1 : 0;
                        const v4i scq = ((cst_v4i *)(ip + P.o_iitem + (fb + f) * 1024) + lane * 4)[tile];
                        const int sce[4] = {scq.x, scq.y, scq.z, scq.w};
                        double oa[4], ob[4];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int k = 4 * kk + k4, ca = 16 * I + i16, cb = 16 * J + i16;
                            const double ta = rec[min(k, 14) * IMU_STRIDE_J + min(ca, 30)], tb = rec[min(k, 14) * IMU_STRIDE_J + min(cb, 30)];
                            oa[kk] = (k < 15 && ca < 31) ? ta : 0.0; ob[kk] = (k < 15 && cb < 31) ? tb : 0.0;
                        }
                        v4f64 acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(oa[kk], ob[kk], acc1, 0, 0, 0);
                        const int bl = 16 * J + i16;
                        int d4[4];
                        double v4[4];
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int al = 16 * I + k4 + 4 * i;
                            if (CHAIN && (sce[i] & IMU_SC_STORE)) C.g_imublk[(fb + f) * IMU_BLK + al * 32 + bl] = acc1[i];
                            d4[i] = (sce[i] & 0xffff) - IMU_SC_BIAS;
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const double tv = C.tiles[max(d4[i], 0)], gv = C.gcam[min(max(-2 - d4[i], 0), CAM_MAX - 1)];
                            v4[i] = d4[i] >= 0 ? tv : gv;
                            if (CHAIN && d4[i] <= -1000) v4[i] = C.hd[-1000 - d4[i]];
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const double v = v4[i] + acc1[i];
                            if (d4[i] >= 0) C.tiles[d4[i]] = v;
                            else if (CHAIN && d4[i] <= -1000) C.hd[-1000 - d4[i]] = v;
                            else if (d4[i] <= -2) C.gcam[-2 - d4[i]] = v;
                        }
                        continue;
                    }
                    if ((sl % NW) != wave) continue;
                    // destinations of this lane's 16 accumulator registers: precomputed by the packer (IMU scatter table), four 16-byte loads
                    cst_v4i *sct = (cst_v4i *)(ip + P.o_iitem + (fb + f) * 1024) + lane * 4;
                    const v4i sc0 = sct[0], sc1 = sct[1], sc2 = sct[2], sc3 = sct[3];
                    const int scv[16] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w, sc2.x, sc2.y, sc2.z, sc2.w, sc3.x, sc3.y, sc3.z, sc3.w};
                    double op[2][4];      // operand values of column tiles 0 and 1 (A and B operands coincide: J' J)
#pragma unroll
                    for (int t = 0; t < 2; t++)
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int k = 4 * kk + k4, col = 16 * t + i16;
                            const double tv = rec[min(k, 14) * IMU_STRIDE_J + min(col, 30)];      // unconditional load, masked after
                            op[t][kk] = (k < 15 && col < 31) ? tv : 0.0;
                        }
                    v4f64 acc[4];
#pragma unroll
                    for (int tile = 0; tile < 4; tile++) {     // (I, J): (0,0) (1,0) (1,1) (0,1)
                        const int I = (tile == 1 || tile == 2) ? 1 : 0, J = (tile >= 2) ? 1 : 0;
                        acc[tile] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) acc[tile] = __builtin_amdgcn_mfma_f64_16x16x4f64(op[I][kk], op[J][kk], acc[tile], 0, 0, 0);
                    }
                    // scatter: all destination reads in flight, then the adds, then the writes (distinct addresses per lane).  Chain mode: the
                    // factor's lower triangle is also parked in HBM/L2 for the chain elimination (whole rows, so that the stores fill their
                    // 64-byte granules; no other factor writes there); entries of a Euclidean block's row / column go nowhere else, their
                    // diagonal feeds the Jacobi scaling.
                    int didx[16];
                    double dval[16];
#pragma unroll
                    for (int tile = 0; tile < 4; tile++) {
                        const int I = (tile == 1 || tile == 2) ? 1 : 0, J = (tile >= 2) ? 1 : 0;
                        const int bl = 16 * J + i16;            // local column of C held by this lane
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int al = 16 * I + k4 + 4 * i;  // local row
                            const int e = scv[tile * 4 + i];
                            if (CHAIN && (e & IMU_SC_STORE)) C.g_imublk[(fb + f) * IMU_BLK + al * 32 + bl] = acc[tile][i];
                            didx[tile * 4 + i] = (e & 0xffff) - IMU_SC_BIAS;      // >= 0: tile element, -2 - ta: gradient entry, -1: nothing
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) {      // both candidate loads unconditional, selected afterwards
                        const double tv = C.tiles[max(didx[q], 0)], gv = C.gcam[min(max(-2 - didx[q], 0), CAM_MAX - 1)];
                        dval[q] = didx[q] >= 0 ? tv : gv;
                        if (CHAIN && didx[q] <= -1000) dval[q] = C.hd[-1000 - didx[q]];
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const double v = dval[q] + acc[q >> 2][q & 3];
                        if (didx[q] >= 0) C.tiles[didx[q]] = v;
                        else if (CHAIN && didx[q] <= -1000) C.hd[-1000 - didx[q]] = v;
                        else if (didx[q] <= -2) C.gcam[-2 - didx[q]] = v;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
        TCV_MARK(C, PH_IMU_GATHER);
    }
}

// ---- marginalisation prior, part B: the constant J0' J0 (cached per solve) joins the tiles
template <int NT, bool CHAIN>
__device__ __forceinline__ void prior_part_b(Ctx<NT> &C, bool assemble) {
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    if (P.prior_n > 0 && assemble && ABL(C, AB_PRIOR_B)) {
        const int n = P.prior_n, npk = n * (n + 1) / 2;
        cst_i *pcol = ip + P.o_pcol;
        cst_i *pdest = ip + P.o_pdest;      // destination of every packed entry, precomputed by the packer
        for (int e0 = tid; e0 < npk; e0 += 4 * NT) {
            double hv[4];
            int di[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = min(e0 + k * NT, npk - 1);
                hv[k] = C.g_hp[e];
                di[k] = (e0 + k * NT < npk) ? pdest[e] : -1;
            }
            double tv[4];
#pragma unroll
            for (int k = 0; k < 4; k++) tv[k] = (CHAIN && di[k] <= -2) ? C.hd[min(-2 - di[k], 111)] : C.tiles[max(di[k], 0)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (di[k] >= 0) C.tiles[di[k]] = tv[k] + hv[k];
                else if (CHAIN && di[k] <= -2) C.hd[-2 - di[k]] = tv[k] + hv[k];
            }
        }
    }
}

// ---- the prior's two products out of LDS (J0 without its leading k0 zero rows: nr x n, column-major) -------------------------------
// r = r0 + J0 dx: thread k0 + row owns a row (thread = ORIGINAL row index, so that the per-thread costs and their block sum are the ones of
// the full matrix); two accumulators by column parity, eight columns' loads in flight.  Returns 0.5 r^2 of the thread's row.
__device__ __forceinline__ double prior_row_lds(const lds_d *J0, cst_d *r0, const lds_d *pdx, lds_d *pr, int n, int nr, int k0, int tid) {
    if (tid < k0 || tid >= n) return 0.0;
    const int row = tid - k0;
    double r = r0[row], r2 = 0.0;
    int j = 0;
    for (; j + 7 < n; j += 8) {
        double a8[8], d8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a8[u] = J0[row + nr * (j + u)]; d8[u] = pdx[j + u]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { r += a8[u] * d8[u]; r2 += a8[u + 1] * d8[u + 1]; }
    }
    for (; j + 1 < n; j += 2) { r += J0[row + nr * j] * pdx[j]; r2 += J0[row + nr * (j + 1)] * pdx[j + 1]; }
    if (n & 1) r += J0[row + nr * (n - 1)] * pdx[n - 1];
    r += r2;
    pr[row] = r;
    return 0.5 * r * r;
}
// (J0' r)[col]: two accumulators by row parity (see the note at Hp = J0'J0: the dropped zero rows do not change the bits)
__device__ __forceinline__ double prior_col_lds(const lds_d *J0, const lds_d *pr, int nr, int col) {
    const lds_d *c = J0 + nr * col;
    double s0 = 0, s1 = 0;
    int i = 0;
    for (; i + 7 < nr; i += 8) {
        double a8[8], r8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { a8[u] = c[i + u]; r8[u] = pr[i + u]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { s0 += a8[u] * r8[u]; s1 += a8[u + 1] * r8[u + 1]; }
    }
    for (; i + 1 < nr; i += 2) { s0 += c[i] * pr[i]; s1 += c[i + 1] * pr[i + 1]; }
    if (nr & 1) s0 += c[nr - 1] * pr[nr - 1];
    return s0 + s1;
}

// ---- linearise at x: cost, and (if assemble) S~ = Hcc - sum_l Hcl Hcl'/kappa_l in the tiles --------
// kappa_l = hll + mu * clamp(s_l^2 hll) / s_l^2 is the landmark pivot of the Jacobi-scaled,
// mu-regularised system expressed in unscaled units (DoglegStrategy + SchurEliminator restated).
template <int NT, bool CHAIN, bool TD = !CHAIN>
__device__ __noinline__ double linearize(TCV_CTX_PARAMS, const lds_d *x, bool first, bool assemble, double mu) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    cst_d *dp = C.dp;
    const int nc = P.nc, L = P.nland, nx = P.nx;
    cst_d *misc = dp + C.W->d_misc;
    const double G3[3] = {misc[0], misc[1], misc[2]};
    const double proj_sqrt = misc[3], proj_loss = misc[4], line_loss = misc[5];
    const bool line_exact = misc[8] != 0.0;
    const int pp_elems = CHAIN ? (C.ntiles << 8) : ((P.ntp * (P.ntp + 1) / 2) << 8);
    // ProjectionTdFactor windows (the dense kernels and the TD instance of the chain kernel): wider point records, see tcv_packed.h
    const bool with_td = TD && (P.flags & 1);
    const int prec = with_td ? (int)PROJ_TD_REC : (int)PROJ_REC, pstr = with_td ? (int)PROJ_TD_STRIDE : (int)PROJ_STRIDE;
    double cost_acc = 0.0;

    if (assemble && ABL(C, AB_ZERO)) {
        zero_lds<NT>(C.tiles, pp_elems, tid);
        for (int i = tid; i < nc; i += NT) C.gcam[i] = 0.0;
        for (int i = tid; i < 176; i += NT) C.rc[i] = 0.0;      // rc | sd
        if (CHAIN) for (int i = tid; i < 112; i += NT) C.hd[i] = 0.0;
    }
    cst_i *blk = ip + P.o_blk;
    TCV_MARK(C, PH_ZERO);
    // ---------------- point + line factors, chunk by chunk -------------------------------------------
    for (int ch = 0; ch < P.n_vis_chunk; ch++) {
        vis_part1<NT, CHAIN, TD>(C, x, ch, assemble, cost_acc, cost_acc);
        if (!assemble) continue;
        vis_part2<NT, CHAIN>(C, ch, first, mu);
    }
    // ---------------- marginalisation prior, part A (marginalization_factor.cpp:335-384): r = r0 + J0 dx, J0' r ------
    // The staging area is free between the Schur phase and the IMU chunks: J0 (n x n, column-major) is staged there
    // once per linearisation so that both products run out of LDS.
    if (P.prior_n > 0 && ABL(C, AB_PRIOR_A)) {
        const int n = P.prior_n, k0 = C.W->prior_k0, nr = n - k0;      // J0 | r0 without their leading zero rows (tcv_packed.h)
        cst_d *J0g = dp + C.W->d_prior, *r0 = J0g + nr * n, *x0 = r0 + nr;
        const bool in_lds = ((nr * n + 1) & ~1) + 2 * n <= C.stage_cap;
        // chain mode: a J0 that does not fit goes through the pool in two column pieces (priors too tall for two pieces take the HBM/L2 path below)
        const int pcap = CHAIN ? (P.c_pool - 2 * n - 2) / nr - 1 : 0;
        const bool staged = CHAIN && !in_lds && n - pcap <= pcap;
        lds_d *J0 = C.stage, *pdx = staged ? C.stage + ((P.c_pool - 2 * n) & ~1) : C.stage + ((nr * n + 1) & ~1), *pr = pdx + n;
        if (tid < P.prior_nblk) {      // dx of one kept block (marginalization_factor.cpp:348-364); fixed-size, fully unrolled
            cst_i *pb = ip + P.o_prior + tid * 4;
            const int gs = pb[2], xo = blk[pb[0] * 4 + 1], x0o = pb[3], ls = gs == 7 ? 6 : gs;
            double d15[15];
#pragma unroll
            for (int i = 0; i < 15; i++) d15[i] = (i < gs) ? x[xo + (i < gs ? i : 0)] - x0[x0o + (i < gs ? i : 0)] : 0.0;
            if (gs == 7) {
                const Quat q0(x0[x0o + 3], x0[x0o + 4], x0[x0o + 5], x0[x0o + 6]), q(x[xo + 3], x[xo + 4], x[xo + 5], x[xo + 6]);
                const Quat dq = inverse(q0) * q;
                const double sg = (dq.w >= 0) ? 2.0 : -2.0;
                d15[3] = sg * dq.x; d15[4] = sg * dq.y; d15[5] = sg * dq.z;
            }
#pragma unroll
            for (int i = 0; i < 15; i++) if (i < ls) { if (in_lds || staged) pdx[pb[1] + i] = d15[i]; else C.g_pdx[pb[1] + i] = d15[i]; }
        }
        if (staged) {
            // J0 does not fit the pool in one piece: columns [0, ns) and [ns, n) are staged one after the other.  The big
            // piece comes second and stays resident for the J0' r pass, so only the small one is read twice.
            int nb = min(n, pcap + 1);
            if ((nr & 1) && ((n - nb) & 1)) nb--;      // keep the second piece 16-byte aligned: nr * (n - nb) even
            const int ns = n - nb;
            lds_d *Jp = C.stage;
            lds_d *pdx2 = pdx, *pr2 = pr;
            const bool rown = tid >= k0 && tid < n;      // thread = original row index
            const int row = rown ? tid - k0 : 0;
            double r = rown ? r0[row] : 0.0, r2 = 0.0;
            for (int piece = 0; piece < 2; piece++) {
                const int c0 = piece == 0 ? 0 : ns, cn = piece == 0 ? ns : nb;
                if (cn == 0) continue;
                copy_doubles_deep<NT>(Jp, J0g + nr * c0, nr * cn, tid);
                __syncthreads();
                if (rown) {
                    int j = 0;
                    for (; j + 7 < cn; j += 8) {      // eight columns' loads in flight, the two accumulators updated in the original order
                        double a8[8], d8[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { a8[u] = Jp[row + nr * (j + u)]; d8[u] = pdx2[c0 + j + u]; }
#pragma unroll
                        for (int u = 0; u < 8; u += 2) { r += a8[u] * d8[u]; r2 += a8[u + 1] * d8[u + 1]; }
                    }
                    for (; j + 1 < cn; j += 2) { r += Jp[row + nr * j] * pdx2[c0 + j]; r2 += Jp[row + nr * (j + 1)] * pdx2[c0 + j + 1]; }
                    if (cn & 1) r += Jp[row + nr * (cn - 1)] * pdx2[c0 + cn - 1];
                }
                if (piece == 0) __syncthreads();
            }
            if (rown) { r += r2; pr2[row] = r; cost_acc += 0.5 * r * r; }
            __syncthreads();
            if (assemble) {
                cst_i *pcol = ip + P.o_pcol;
                for (int piece = 1; piece >= 0; piece--) {
                    const int c0 = piece == 0 ? 0 : ns, cn = piece == 0 ? ns : nb;
                    if (cn == 0) continue;
                    if (piece == 0) { __syncthreads(); copy_doubles_deep<NT>(Jp, J0g, nr * cn, tid); __syncthreads(); }
                    if (tid < cn) {
                        const int t = pcol[c0 + tid];
                        if (t >= 0) C.gcam[t] += prior_col_lds(Jp, pr2, nr, tid);
                    }
                }
            }
            __syncthreads();
        } else if (in_lds) {
            copy_doubles_deep<NT>(J0, J0g, nr * n, tid);      // ONE memory round trip
            __syncthreads();
            cost_acc += prior_row_lds(J0, r0, pdx, pr, n, nr, k0, tid);
            __syncthreads();
            if (assemble) {
                cst_i *pcol = ip + P.o_pcol;
                if (tid < n) {
                    const int t = pcol[tid];
                    if (t >= 0) C.gcam[t] += prior_col_lds(J0, pr, nr, tid);
                }
            }
            __syncthreads();
        } else {
            __syncthreads();
            if (tid >= k0 && tid < n) {
                double r = r0[tid - k0];
                for (int j = 0; j < n; j++) r += J0g[tid - k0 + nr * j] * C.g_pdx[j];
                C.g_pr[tid - k0] = r;
                cost_acc += 0.5 * r * r;
            }
            __syncthreads();
            if (assemble) {
                cst_i *pcol = ip + P.o_pcol;
                if (tid < n && pcol[tid] >= 0) {
                    double s2 = 0;
                    for (int i = 0; i < nr; i++) s2 += J0g[i + nr * tid] * C.g_pr[i];
                    C.gcam[pcol[tid]] += s2;
                }
            }
            __syncthreads();
        }
    }
    TCV_MARK(C, PH_PRIOR);
    if (assemble && !CHAIN) {
        const int all_elems = C.ntiles << 8;
        zero_lds<NT>(C.tiles + pp_elems, all_elems - pp_elems, tid);
    }
    __syncthreads();
    TCV_MARK(C, PH_ZERO);
    // ---------------- IMU factors, chunk by chunk ---------------------------------------------------
    // raw residual/Jacobian: lanes = factors, waves 0..3 = the four parts of imu_raw_part; whitening J = sqrt_info * J_raw
    // and J'J / J'r per factor on the matrix cores (one wavefront per factor); the 30 x 31 result is scattered into the
    // reduced camera system colour by colour (factors of one colour share no block).
    for (int ch = 0; ch < P.n_imu_chunk; ch++) {
        imu_part1<NT, CHAIN>(C, x, ch, assemble, cost_acc);
        imu_part2<NT, CHAIN>(C, ch, assemble);
    }
    // ---------------- marginalisation prior, part B: the constant J0' J0 (cached per solve) joins the tiles --------------
    prior_part_b<NT, CHAIN>(C, assemble);
    TCV_MARK(C, PH_PRIOR);
    const double cost = block_sum<NT>(cost_acc, C.red, tid);
    __syncthreads();
    TCV_MARK(C, PH_COST_RED);
    return cost;
}

// ---- cooperative mode (tcv_packed.h COOP_*): a window on 1 + H workgroups ----------------------------------------------------------
// Hand-off flags live in HBM / L2 and are written with release, read with acquire semantics at agent scope (the workgroups of a group
// normally share an XCD and its L2 -- see the launch mapping in solve_kernel -- but nothing depends on it).  Every wait is bounded:
// a partner that never answers (a group that was not co-resident, a fault) ends the window with status -9 instead of hanging the queue.
// polling loads are RELAXED (a coherent load, no cache invalidation per trip: an acquire in the loop floods the L2 the partner is trying to
// write its exports through); the waiting workgroup issues one acquire fence once the flag has flipped
__device__ __forceinline__ int coop_load(gbl_i *p) { return __hip_atomic_load((int *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coop_store(gbl_i *p, int v) { __hip_atomic_store((int *)p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
// progress marks of a group in its control block (ints 16 .. 31: master, 32 + 4 h + w: wavefront w of helper h), read by tcv_batch_debug_coop when a launch
// has to be diagnosed; relaxed stores of one lane
#define COOP_MARKW(C, h, v) do { if (((C).tid & 63) == 0) __hip_atomic_store((int *)((C).cx_ctl + 32 + 4 * (h) + ((C).tid >> 6)), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#define COOP_MARK(C, slot, v) do { if ((C).tid == 0) __hip_atomic_store((int *)((C).cx_ctl + (slot)), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)

// master: wait until every helper has served request `seq`; false on timeout / abort (uniform over the workgroup)
template <int NT>
__device__ __forceinline__ bool coop_wait_helpers(const Ctx<NT> &C, int seq) {
    int ok = 1;
    if (C.tid < C.cx_h) {
        gbl_i *f = C.cx_ctl + COOP_CTL_DONE + C.tid;
        const long long t0 = (long long)wall_clock64();
        while (coop_load(f) != seq) {
            if (coop_load(C.cx_ctl + COOP_CTL_ABORT) != 0 || (long long)wall_clock64() - t0 > C.cx_timeout) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    // (all waiting lanes sit in the first wavefront; __syncthreads_and would add static LDS to a kernel that owns all 160 KiB dynamically)
    lds_i *slot = (lds_i *)(C.red + 61);
    if (C.tid < 64) { const bool all = __ballot(ok == 0) == 0ull; if (C.tid == 0) *slot = all ? 1 : 0; }
    __syncthreads();
    ok = *slot;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every wavefront: nothing stale of the helpers' exports in this CU's caches
    return ok != 0;
}

// master: publish a request (state, mu, command) to the helpers of the group
template <int NT>
__device__ __forceinline__ void coop_publish(const Ctx<NT> &C, const lds_d *x, int n, double mu, int cmd, int win, int seq) {
    if (x) for (int i = C.tid; i < n; i += NT) C.cx_x[i] = x[i];
    if (C.tid == 0) { C.cx_x[SCR_NL] = mu; C.cx_ctl[COOP_CTL_CMD] = cmd; C.cx_ctl[COOP_CTL_WIN] = win; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // every wavefront's part of x (the release of lane 0 below covers its own wavefront only)
    __syncthreads();
    if (C.tid == 0) coop_store(C.cx_ctl + COOP_CTL_SEQ, seq);
}

// Linearisation by the master of a group: same result as linearize<NT, true>, bit for bit when the plan's chunks hold at most NT point
// factors each (the cooperative packer's rule), because every sum is formed by the same additions in the same order:
//   tiles / gradient / rhs and diagonal corrections: zero, then per chunk c the gathered value a_c (one addition per destination and
//   chunk), then minus the Schur value s_c (the helper exports -s_c = 0 - s_c exactly), then the prior's J0'r, the IMU blocks colour by
//   colour, the cached J0'J0;  cost: per thread its point and line costs chunk by chunk, then its prior row, then its IMU factor.
// Returns the cost and the sequence number of the request; cost NaN and seq = -1 if the helpers did not answer.  The context travels in
// registers like in the other phase functions (TCV_CTX_PARAMS), the group's hand-off block as seven more arguments.
struct CoopLin { double cost; int seq; };
template <int NT>
__device__ __noinline__ CoopLin linearize_coop(TCV_CTX_PARAMS, gbl_i *cx_ctl_, gbl_d *cx_x_, gbl_d *cx_exp_, int cx_h_, int cx_exp_stride_, int cx_seq_, long long cx_timeout_,
                                               const lds_d *x, bool first, bool assemble, double mu, int win) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    C.cx_ctl = (gbl_i *)uni_ptr((gbl_d *)cx_ctl_); C.cx_x = uni_ptr(cx_x_); C.cx_exp = uni_ptr(cx_exp_);
    C.cx_h = __builtin_amdgcn_readfirstlane(cx_h_); C.cx_exp_stride = __builtin_amdgcn_readfirstlane(cx_exp_stride_);
    C.cx_seq = __builtin_amdgcn_readfirstlane(cx_seq_); C.cx_timeout = cx_timeout_;
    cst_plan &P = *C.P;
    const int tid = C.tid;
    cst_i *ip = C.ip;
    cst_d *dp = C.dp;
    const int L = P.nland, nx = P.nx;
    const int seq = C.cx_seq + 1;
    COOP_MARK(C, 16, 1); COOP_MARK(C, 17, seq);
    coop_publish<NT>(C, x, nx + L, mu, (first ? COOP_CMD_FIRST : 0) | (assemble ? COOP_CMD_ASSEMBLE : 0), win, seq);
    COOP_MARK(C, 16, 2);
    cst_i *blk = ip + P.o_blk;
    if (assemble) for (int i = tid; i < 112; i += NT) C.hd[i] = 0.0;
    TCV_MARK(C, PH_ZERO);
    // ---- own work while the helpers evaluate the visual chunks: prior part A (marginalization_factor.cpp:335-384) with J0 staged in the
    // (otherwise idle) pool -- its J0'r is kept in a register until the chunks are folded in, the single-workgroup order --, IMU factors
    double cost_prior = 0.0, cost_imu = 0.0, pg = 0.0;
    int pt = -1;
    if (P.prior_n > 0) {
        const int n = P.prior_n, k0 = C.W->prior_k0, nr = n - k0;      // J0 | r0 without their leading zero rows (tcv_packed.h)
        cst_d *J0g = dp + C.W->d_prior, *r0 = J0g + nr * n, *x0 = r0 + nr;
        lds_d *J0 = C.stage, *pdx = C.stage + ((nr * n + 1) & ~1), *pr = pdx + n;      // n n + 2 n <= stage_cap: checked by tcv_batch_create
        if (tid < P.prior_nblk) {
            cst_i *pb = ip + P.o_prior + tid * 4;
            const int gs = pb[2], xo = blk[pb[0] * 4 + 1], x0o = pb[3], ls = gs == 7 ? 6 : gs;
            double d15[15];
#pragma unroll
            for (int i = 0; i < 15; i++) d15[i] = (i < gs) ? x[xo + (i < gs ? i : 0)] - x0[x0o + (i < gs ? i : 0)] : 0.0;
            if (gs == 7) {
                const Quat q0(x0[x0o + 3], x0[x0o + 4], x0[x0o + 5], x0[x0o + 6]), q(x[xo + 3], x[xo + 4], x[xo + 5], x[xo + 6]);
                const Quat dq = inverse(q0) * q;
                const double sg = (dq.w >= 0) ? 2.0 : -2.0;
                d15[3] = sg * dq.x; d15[4] = sg * dq.y; d15[5] = sg * dq.z;
            }
#pragma unroll
            for (int i = 0; i < 15; i++) if (i < ls) pdx[pb[1] + i] = d15[i];
        }
        copy_doubles_deep<NT>(J0, J0g, nr * n, tid);
        __syncthreads();
        cost_prior = prior_row_lds(J0, r0, pdx, pr, n, nr, k0, tid);      // the in_lds branch of linearize()
        __syncthreads();
        if (assemble && tid < n) {
            const int t = (ip + P.o_pcol)[tid];
            if (t >= 0) { pg = prior_col_lds(J0, pr, nr, tid); pt = t; }
        }
        __syncthreads();
    }
    TCV_MARK(C, PH_PRIOR);
    COOP_MARK(C, 16, 3);
    if (P.n_imu_chunk > 0) imu_part1<NT, true>(C, x, 0, assemble, cost_imu);      // (one chunk: checked by tcv_batch_create)
    COOP_MARK(C, 16, 4);
    // ---- the helpers' chunks
    if (!coop_wait_helpers<NT>(C, seq)) {
        if (tid == 0) coop_store(C.cx_ctl + COOP_CTL_ABORT, 1);
        COOP_MARK(C, 16, 9);
        CoopLin bad; bad.cost = __builtin_nan(""); bad.seq = -1;
        return bad;
    }
    TCV_MARK(C, PH_VIS_GATHER);
    COOP_MARK(C, 16, 5);
    const int nch = P.n_vis_chunk, te = C.ntiles << 8;
    const gbl_d *E = C.cx_exp;
    const int es = C.cx_exp_stride;
    double cost_acc = 0.0;
    for (int c = 0; c < nch; c++) { cost_acc += E[(size_t)c * es + 2 * te + 352 + tid]; cost_acc += E[(size_t)c * es + 2 * te + 352 + 256 + tid]; }
    cost_acc += cost_prior;
    cost_acc += cost_imu;
    if (assemble) {
        typedef double v2d __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) v2d gbl_v2d;
        typedef __attribute__((address_space(3))) v2d lds_v2d;
        // tiles: element pairs; per element t = 0 + a_1 + (-s_1) + a_2 + (-s_2) ...  (all loads of a trip in flight, the additions in order)
        for (int i = tid; i < (te >> 1); i += NT) {
            v2d t = {0.0, 0.0};
            int c = 0;
            for (; c + 1 < nch; c += 2) {
                const gbl_v2d *e0 = (const gbl_v2d *)(E + (size_t)c * es), *e1 = (const gbl_v2d *)(E + (size_t)(c + 1) * es);
                const v2d a0 = e0[i], s0 = e0[(te >> 1) + i], a1 = e1[i], s1 = e1[(te >> 1) + i];
                t += a0; t += s0; t += a1; t += s1;
            }
            if (c < nch) { const gbl_v2d *e0 = (const gbl_v2d *)(E + (size_t)c * es); const v2d a0 = e0[i], s0 = e0[(te >> 1) + i]; t += a0; t += s0; }
            ((lds_v2d *)C.tiles)[i] = t;
        }
        // gradient (176) | rc, sd (176)
        for (int i = tid; i < 352; i += NT) {
            double t = 0.0;
            for (int c = 0; c < nch; c++) t += E[(size_t)c * es + 2 * te + i];
            if (i < 176) C.gcam[i] = t; else C.rc[i - 176] = t;
        }
        __syncthreads();
        if (pt >= 0) C.gcam[pt] += pg;
        __syncthreads();
        TCV_MARK(C, PH_SCHUR);
        if (P.n_imu_chunk > 0) imu_part2<NT, true>(C, 0, assemble);
        prior_part_b<NT, true>(C, assemble);
    }
    TCV_MARK(C, PH_PRIOR);
    const double cost = block_sum<NT>(cost_acc, C.red, tid);
    __syncthreads();
    COOP_MARK(C, 16, 6);
    TCV_MARK(C, PH_COST_RED);
    CoopLin out; out.cost = cost; out.seq = seq;
    return out;
}

// A helper workgroup of group g: serves the master's linearisation requests until it is told to leave.  Its LDS is carved like the
// master's (pose tiles | pool | vectors) plus a second tile set for the Schur values; its scratch pointers are the MASTER's (landmark
// pivots, couplings and scales are written where the master's back-substitution reads them).
template <int NT, bool TD = false>
__device__ __noinline__ void coop_helper(const SolveArgs &A, lds_d *lds, int g_, int h_) {
    const int tid = threadIdx.x;
    // arguments of a non-inlined function arrive in vector registers: made uniform by hand, or every loop over them becomes an exec-mask loop
    const int g = __builtin_amdgcn_readfirstlane(g_), h = __builtin_amdgcn_readfirstlane(h_);
    Ctx<NT> C;
    C.tid = tid;
    gbl_d *scr = (gbl_d *)A.scratch + (size_t)g * A.scratch_stride;
    C.v_s = scr; C.v_g = scr + SCR_NL; C.v_D = scr + 2 * SCR_NL; C.v_ghat = scr + 3 * SCR_NL; C.v_y = scr + 4 * SCR_NL;
    C.v_p = scr + 5 * SCR_NL; C.v_rc = scr + 6 * SCR_NL; C.v_sd = scr + 7 * SCR_NL;
    C.l_hll = scr + 8 * SCR_NL; C.l_gl = C.l_hll + SCR_LM; C.l_invk = C.l_gl + SCR_LM;
    C.g_hp = C.l_invk + SCR_LM; C.g_pr = C.g_hp + SCR_HP; C.g_pdx = C.g_pr + 128;
    C.g_sqrt = C.g_pdx + 128;
    C.g_hcl = C.g_sqrt + SCR_SQ;
    C.prof = nullptr; C.t_last = 0; C.skip = 0;
    C.g_imublk = nullptr; C.g_spill = nullptr;
    C.cx_ctl = (gbl_i *)A.coop_ctl + (size_t)g * COOP_CTL_INTS;
    C.cx_x = (gbl_d *)A.coop_x + (size_t)g * COOP_X_DOUBLES;
    C.cx_exp = (gbl_d *)A.coop_exp + (size_t)g * A.coop_exp_chunks * A.coop_exp_stride;
    C.cx_h = A.coop_h; C.cx_exp_stride = A.coop_exp_stride; C.cx_seq = 0; C.cx_timeout = A.coop_timeout;
    lds_i *bcast = (lds_i *)(lds + LDS_DOUBLES - 2);      // the last 16 bytes of the workgroup's LDS: outside every window's carve
    int seen = 0;
    for (;;) {
        COOP_MARKW(C, h, 1 + (seen << 4));
        if (tid == 0) {
            const long long t0 = (long long)wall_clock64();
            int s;
            while ((s = coop_load(C.cx_ctl + COOP_CTL_SEQ)) == seen) {
                if (coop_load(C.cx_ctl + COOP_CTL_ABORT) != 0 || (long long)wall_clock64() - t0 > C.cx_timeout) { s = -1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            *bcast = s;
        }
        __syncthreads();
        const int seq = __builtin_amdgcn_readfirstlane(*bcast);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
        if (seq < 0) { if (tid == 0) coop_store(C.cx_ctl + COOP_CTL_ABORT, 1); return; }
        const int cmd = __builtin_amdgcn_readfirstlane(C.cx_ctl[COOP_CTL_CMD]), win = __builtin_amdgcn_readfirstlane(C.cx_ctl[COOP_CTL_WIN]);
        COOP_MARKW(C, h, 2 + (seq << 4));
        if (cmd & COOP_CMD_EXIT) return;
        cst_win *W = (cst_win *)A.win + win;
        cst_plan &P = ((cst_plan *)A.plans)[W->plan];
        C.P = &P; C.W = W;
        C.ip = (cst_i *)A.ipool + A.plan_base[W->plan];
        C.dp = (cst_d *)A.dpool + W->dbase;
        const int L = P.nland, nxl = (P.nx + L + 1) & ~1;
        C.ntd = P.nt_c; C.nd = P.npp;
        C.ntiles = C.ntd * (C.ntd + 1) / 2;
        const int te = C.ntiles << 8;
        C.tiles = lds;
        C.stage = lds + te;
        C.stage_cap = P.c_stage_cap;
        lds_d *p = C.stage + P.c_pool;
        C.xs = p; p += nxl;
        C.xc = p; p += nxl;
        C.sc = p; p += 176;
        C.rc = p; C.sd = p + 88; p += 176;
        C.ycam = p; p += 176;
        C.invdiag = p; C.gcam = p; p += 176;
        C.red = p; p += 64;
        C.flag = (lds_i *)(C.red + 62);
        C.hd = p; p += 112;
        C.area = C.stage + P.c_stage_cap;
        lds_d *tiles2 = p;                                   // the cooperative packer leaves room for it (chain_lds = LDS_DOUBLES - te)
        const Ctx<NT> K = uniform_ctx<NT>(C);
        for (int i = tid; i < P.nx + L; i += NT) K.xs[i] = K.cx_x[i];
        const double mu = K.cx_x[SCR_NL];
        const bool first = (cmd & COOP_CMD_FIRST) != 0, assemble = (cmd & COOP_CMD_ASSEMBLE) != 0;
        Ctx<NT> K1 = K, K2 = K;
        K2.tiles = uni_ptr(tiles2);
        for (int ch = h; ch < P.n_vis_chunk; ch += K.cx_h) {
            gbl_d *E = K.cx_exp + (size_t)ch * K.cx_exp_stride;
            if (assemble) {
                zero_lds<NT>(K.tiles, te, tid);
                zero_lds<NT>(tiles2, te, tid);
                for (int i = tid; i < 176; i += NT) { K.gcam[i] = 0.0; K.rc[i] = 0.0; }      // rc | sd
            }
            double cost_pt = 0.0, cost_ln = 0.0;
            COOP_MARKW(C, h, 3 + (seq << 4));
            vis_part1<NT, true, TD>(K1, K.xs, ch, assemble, cost_pt, cost_ln);      // (starts with a barrier: the zeroing above is ordered before the gathers)
            COOP_MARKW(C, h, 4 + (seq << 4));
            if (assemble) {
                vis_part2<NT, true>(K2, ch, first, mu);
                COOP_MARKW(C, h, 5 + (seq << 4));
                typedef double v2d __attribute__((ext_vector_type(2)));
                typedef __attribute__((address_space(1))) v2d gbl_v2d;
                typedef __attribute__((address_space(3))) v2d lds_v2d;
                for (int i = tid; i < (te >> 1); i += NT) { ((gbl_v2d *)E)[i] = ((lds_v2d *)K.tiles)[i]; ((gbl_v2d *)E)[(te >> 1) + i] = ((lds_v2d *)tiles2)[i]; }
                for (int i = tid; i < 176; i += NT) { E[2 * te + i] = K.gcam[i]; E[2 * te + 176 + i] = K.rc[i]; }
            }
            E[2 * te + 352 + tid] = cost_pt;
            E[2 * te + 352 + 256 + tid] = cost_ln;
            COOP_MARKW(C, h, 6 + (seq << 4));
            __syncthreads();
        }
        // every wavefront's exports must have left for the L2 / HBM before lane 0 signals: the release of one lane waits for its own
        // wavefront's stores only
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        COOP_MARKW(C, h, 7 + (seq << 4));
        if (tid == 0) coop_store(K.cx_ctl + COOP_CTL_DONE + h, seq);
        COOP_MARKW(C, h, 8 + (seq << 4));
        seen = seq;
    }
}

// ---- tiled Cholesky of the augmented system in LDS -----------------------------------------------
// sqrt(d) and 1/sqrt(d) from v_rsq_f64 + Goldschmidt/Newton refinement (same scheme LLVM uses for f64 sqrt)
__device__ __forceinline__ void sqrt_rsqrt(double d, double &l, double &inv) {
    // v_rsq_f64 carries ~26 bits; one Goldschmidt step squares the error, one Newton step polishes sqrt(d).
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double e = fma(-g, g, d);
    l = fma(e, h, g);
    inv = h + h;
}

// the two halves go through a 2 x 32-bit vector so that the compiler forms the scalar register PAIR directly (shifting and or-ing
// them together as a 64-bit integer costs two extra SALU instructions per broadcast, and the factorisations broadcast by the hundred)
typedef unsigned tcv_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double readlane_f64(double v, int srclane) {
    const tcv_u2 u = __builtin_bit_cast(tcv_u2, v);
    tcv_u2 r;
    r.x = __builtin_amdgcn_readlane(u.x, srclane);
    r.y = __builtin_amdgcn_readlane(u.y, srclane);
    return __builtin_bit_cast(double, r);
}

// Cholesky of one 16x16 diagonal tile by ONE wavefront, one matrix row per lane held in registers
// (lane & 15 = row; the 16-lane groups 0, 2, 3 compute redundantly, so there is no divergence).  The pivot and the
// pivot column are broadcast with v_readlane (SGPR operands of the FMAs): nothing on the pivot chain touches LDS.
// Only the first cmax columns are pivots (the rest: rhs row / padding).
// Group 1 (lanes 16..31) carries the rows of the IDENTITY through the same column operations (x <- x L^-T, the trick of the chain's T
// step): they end as the rows of L^-T, whose strictly upper part goes to the unused upper triangle of the tile (its diagonal is 1 / l_kk =
// invd) -- for nothing on the pivot chain -- and turns the panel solve below the tile into a product on the matrix cores.
__device__ __forceinline__ bool diag_tile_wave(lds_d *TK, int cmax, lds_d *invd, int lane) {
    const int r = lane & 15;
    const bool idrow = (lane >> 4) == 1;
    double t[16];
#pragma unroll
    for (int c = 0; c < 16; c++) { const double tv = TK[sw(r, c)]; t[c] = idrow ? ((r == c) ? 1.0 : 0.0) : tv; }
    bool ok = true;
    double d = readlane_f64(t[0], 0);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (k < cmax && ok) {
            if (!(d > 0.0) || !(d < 1e300)) ok = false;
            else {
                double l, y;
                sqrt_rsqrt(d, l, y);
                const double lk = (r == k && !idrow) ? l : t[k] * y;
                t[k] = lk;
                if (k + 1 < 16) {       // next pivot first: its rsqrt chain overlaps the rest of this column's updates
                    t[k + 1] -= lk * readlane_f64(lk, k + 1);
                    d = readlane_f64(t[k + 1], k + 1);
                }
#pragma unroll
                for (int c = k + 2; c < 16; c++) t[c] -= lk * readlane_f64(lk, c);
                if (lane == 0) invd[k] = y;
            }
        }
    }
    if (lane < 32) {      // group 0: L (lower triangle and diagonal); group 1: L^-T above the diagonal
#pragma unroll
        for (int c = 0; c < 16; c++) if (idrow ? (c > r) : (c <= r)) TK[sw(r, c)] = t[c];
    }
    return ok;
}

// panel below a factored diagonal tile on the matrix cores: X L_KK' = A_IK, with the L_KK^-T the diagonal factorisation left behind in the
// tile's upper triangle and invd (diag_tile_wave).  A product with an explicit inverse is not backward stable: its residual A - X L' is
// cond(L_KK) eps |A|, and a camera system that is rank deficient beyond the gauge (a pose held by one observation and no IMU factor, a two-frame
// window: only the trust region's mu D^2 holds those directions, cond(L_KK) ~ 1e4) then loses that factor in the solution -- first steps off by
// 1e-3 where substitution gives 1e-7 (tests/dev/fuzz_solve.py).  One refinement step on the matrix cores restores the residual to eps |A|:
//   Y1 = Linv A',  R = A' - L Y1,  Y = Y1 + Linv R,  X = Y'
// in the transposed form because the accumulator layout of v_mfma_f64_16x16x4 IS its B-operand layout (a product can be multiplied from the
// LEFT straight out of the registers): 12 MFMAs per tile instead of 4, in a phase that is 1 % of the kernel.
__device__ __forceinline__ void panel_tile_mfma(lds_d *T, const lds_d *TK, const lds_d *invd, int lane) {
    const int row0 = lane >> 4, col = lane & 15;
    double li[4], ll[4], at[4];      // A operands: Linv[m = col][k], L[m = col][k] (both lower triangular); B operand: A'[k][n = col] = A[col][k]
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int k = 4 * kk + row0;
        const double up = TK[sw(min(k, col), max(k, col))], lo = TK[sw(max(k, col), min(k, col))], dg = invd[col];
        li[kk] = (k < col) ? up : ((k == col) ? dg : 0.0);           // Linv[col][k] = L^-T[k][col]
        ll[kk] = (k <= col) ? -lo : 0.0;                             // -L[col][k]
        at[kk] = T[sw(col, k)];
    }
    v4f64 y1 = {0.0, 0.0, 0.0, 0.0}, r;
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = at[i];                        // A' in the accumulator layout (rows row0 + 4 i of column col) IS the B operand just loaded
#pragma unroll
    for (int kk = 0; kk < 4; kk++) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(li[kk], at[kk], y1, 0, 0, 0);
#ifndef TCV_PANEL_NOREFINE      // (A/B build `build.py --norefine`: the product with the explicit inverse alone, as up to round 5)
#pragma unroll
    for (int kk = 0; kk < 4; kk++) r = __builtin_amdgcn_mfma_f64_16x16x4f64(ll[kk], y1[kk], r, 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(li[kk], r[kk], y1, 0, 0, 0);
#endif
#pragma unroll
    for (int i = 0; i < 4; i++) T[sw(col, row0 + 4 * i)] = y1[i];    // X = Y'
}

template <bool MFMA>
__device__ __forceinline__ void update_tile(lds_d *tiles, int I, int J, int K, int lane) {
    const int row0 = lane >> 4, col = lane & 15;
    lds_d *Ct = tiles + tbase(I, J);
    const lds_d *A = tiles + tbase(I, K), *B = tiles + tbase(J, K);
    if (MFMA) {
        v4f64 acc;
        double av[4], bv[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) { av[kk] = -A[sw(col, 4 * kk + row0)]; bv[kk] = B[sw(col, 4 * kk + row0)]; }
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = Ct[sw(row0 + 4 * i, col)];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++) Ct[sw(row0 + 4 * i, col)] = acc[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int row = row0 + 4 * i;
            double a = Ct[sw(row, col)];
#pragma unroll
            for (int k = 0; k < 16; k++) a -= A[sw(row, k)] * B[sw(col, k)];
            Ct[sw(row, col)] = a;
        }
    }
}
// two independent tiles at once: their dependent MFMA chains interleave on the matrix core and all LDS operands of
// both are in flight before the first MFMA issues
template <bool MFMA>
__device__ __forceinline__ void update_tile2(lds_d *tiles, int I0, int J0, int I1, int J1, int K, int lane) {
    if (!MFMA) { update_tile<false>(tiles, I0, J0, K, lane); update_tile<false>(tiles, I1, J1, K, lane); return; }
    const int row0 = lane >> 4, col = lane & 15;
    lds_d *C0 = tiles + tbase(I0, J0), *C1 = tiles + tbase(I1, J1);
    const lds_d *A0 = tiles + tbase(I0, K), *B0 = tiles + tbase(J0, K), *A1 = tiles + tbase(I1, K), *B1 = tiles + tbase(J1, K);
    v4f64 acc0, acc1;
    double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        a0[kk] = -A0[sw(col, 4 * kk + row0)]; b0[kk] = B0[sw(col, 4 * kk + row0)];
        a1[kk] = -A1[sw(col, 4 * kk + row0)]; b1[kk] = B1[sw(col, 4 * kk + row0)];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { acc0[i] = C0[sw(row0 + 4 * i, col)]; acc1[i] = C1[sw(row0 + 4 * i, col)]; }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b0[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { C0[sw(row0 + 4 * i, col)] = acc0[i]; C1[sw(row0 + 4 * i, col)] = acc1[i]; }
}

// Right-looking tiled Cholesky with look-ahead: while the other waves run the trailing update of step K,
// wave 0 updates tile (K+1, K+1) first and factorises it, so the serial pivot chain of the next diagonal
// tile hides behind the matrix-core work.  Two barriers per tile column.
template <int NT, bool MFMA>
__device__ __noinline__ bool chol_tiles(TCV_CTX_PARAMS, int nt, int nc) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    const int tid = C.tid, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = NT / 64, NG = NT / 16;
    lds_d *tiles = C.tiles;
    if (wave == 0) {
        const int cmax0 = min(16, nc);
        if (!diag_tile_wave(tiles + tbase(0, 0), cmax0, C.invdiag, lane) && lane == 0) *C.flag = 1;
    }
    __syncthreads();
    TCV_MARK(C, PH_CHOL_DIAG);
#ifdef TCV_ABLATE
    if (!ABL_FORCE(C))
#endif
    if (*C.flag) return false;
    for (int K = 0; K + 1 < nt; K++) {
        const int cmax = min(16, nc - 16 * K);
        if (cmax < 16) break;   // partial pivot tile is the last tile row: nothing below / right of it
        const lds_d *TK = tiles + tbase(K, K);
        // panel: X * L_KK^T = A_IK -- on the matrix cores with the L_KK^-T the diagonal factorisation left behind (one tile per wavefront and
        // trip), or by forward substitution, one matrix row per lane
        if (MFMA) {
            for (int I = K + 1 + wave; I < nt; I += NW) panel_tile_mfma(tiles + tbase(I, K), TK, C.invdiag + 16 * K, lane);
        } else {
            const int g = tid >> 4, r = tid & 15;
            for (int I = K + 1 + g; I < nt; I += NG) {
                lds_d *T = tiles + tbase(I, K);
                double xr[16];
#pragma unroll
                for (int c = 0; c < 16; c++) xr[c] = T[sw(r, c)];
#pragma unroll
                for (int c = 0; c < 16; c++) {      // two partial sums halve the dependent FMA chain
                    double a = xr[c], a2 = 0.0;
#pragma unroll
                    for (int c1 = 0; c1 + 1 < c; c1 += 2) { a -= xr[c1] * TK[sw(c, c1)]; a2 -= xr[c1 + 1] * TK[sw(c, c1 + 1)]; }
                    if (c & 1) a -= xr[c - 1] * TK[sw(c, c - 1)];
                    xr[c] = (a + a2) * C.invdiag[16 * K + c];
                }
#pragma unroll
                for (int c = 0; c < 16; c++) T[sw(r, c)] = xr[c];
            }
        }
        __syncthreads();
        TCV_MARK(C, PH_CHOL_TRSM);
        // trailing update A_IJ -= L_IK L_JK^T; wave 0 takes (K+1, K+1) + the next diagonal factorisation, which is
        // priced at DIAG_COST tile updates when the remaining tiles are dealt to the least-loaded wave
        {
            constexpr int DIAG_COST = 12;
            const int cnext = min(16, nc - 16 * (K + 1));
            if (wave == 0) {
                update_tile<MFMA>(tiles, K + 1, K + 1, K, lane);
                if (cnext > 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    if (!diag_tile_wave(tiles + tbase(K + 1, K + 1), cnext, C.invdiag + 16 * (K + 1), lane) && lane == 0) *C.flag = 1;
                }
            }
            // the other tiles: the first DIAG_COST * (NW - 1) go round-robin to waves 1.., the rest to all waves; every
            // wave runs its tiles two at a time
            constexpr int HEAD = (NW > 1) ? DIAG_COST * (NW - 1) : 0;
            int idx = 0, pI = -1, pJ = -1;
            for (int I = K + 1; I < nt; I++)
                for (int J = K + 1; J <= I; J++) {
                    if (I == K + 1 && J == K + 1) continue;
                    const int owner = (idx < HEAD && cnext > 0) ? 1 + idx % (NW - 1) : (idx - ((cnext > 0) ? HEAD : 0)) % NW;
                    idx++;
                    if (owner != wave) continue;
                    if (pI < 0) { pI = I; pJ = J; }
                    else { update_tile2<MFMA>(tiles, pI, pJ, I, J, K, lane); pI = -1; }
                }
            if (pI >= 0) update_tile<MFMA>(tiles, pI, pJ, K, lane);
        }
        __syncthreads();
        TCV_MARK(C, PH_CHOL_UPD);
#ifdef TCV_ABLATE
        if (!ABL_FORCE(C))
#endif
        if (*C.flag) return false;
    }
    return true;
}

// ---- back substitution L^T y = z (z = the factored rhs row) -----------------------------------------
// Column sweep: thread c keeps t_c = z_c - sum_{r > tile} L[r][c] y_r in a register.  Per tile row K (descending): the
// 16 lanes that own its columns solve the 16 x 16 triangular system among themselves (v_readlane broadcasts), publish
// y_K, and after ONE barrier every thread folds tile row K into its own t_c.
template <int NT>
__device__ __noinline__ void back_subst(TCV_CTX_PARAMS, int nc) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    const int tid = C.tid, lane = tid & 63;
    lds_d *tiles = C.tiles, *y = C.ycam;
    const int c = tid, Kc = c >> 4, cc = c & 15;
    double t = tiles[tix(nc, min(c, nc - 1))];
    if (c >= nc) t = 0.0;
    for (int K = (nc - 1) >> 4; K >= 0; K--) {
        const int cmax = min(16, nc - 16 * K);
        if (Kc == K) {     // the 16 lanes holding columns 16K .. 16K+15 (one aligned 16-lane group of one wave)
            const lds_d *TK = tiles + tbase(K, K);
            const int g0 = lane & ~15;
            double lrow[16];
#pragma unroll
            for (int j = 0; j < 16; j++) { const double tv = TK[sw(j, cc)]; lrow[j] = (cc <= j) ? tv : 0.0; }      // L[j][cc]
            const double inv = C.invdiag[c < nc ? c : 0];
#pragma unroll
            for (int j = 15; j >= 0; j--) {
                if (j < cmax) {
                    const double yj = readlane_f64(t * inv, g0 + j);     // lane j of the group: its t is final
                    if (cc == j) t = yj;
                    else if (cc < j) t -= lrow[j] * yj;
                }
            }
            if (cc < cmax) y[c] = t;
        }
        __syncthreads();
        if (Kc < K && c < nc) {
            const lds_d *T = tiles + tbase(K, Kc);
            double lv[16], yv[16];
#pragma unroll
            for (int i = 0; i < 16; i++) { lv[i] = T[sw(i, cc)]; yv[i] = y[16 * K + i]; }
            double s0 = 0, s1 = 0;
#pragma unroll
            for (int i = 0; i < 16; i += 2) { if (i < cmax) s0 += lv[i] * yv[i]; if (i + 1 < cmax) s1 += lv[i + 1] * yv[i + 1]; }
            t -= s0 + s1;
        }
    }
    __syncthreads();
}

// ---- chain mode: block elimination of the Euclidean camera blocks before the dense pose system ------------------------
// The camera system is [E B; B' S]: E the Euclidean (speed-bias) blocks in elimination order -- block tridiagonal, block e_s only
// meets e_{s+1} -- S the pose system in the LDS tiles (rhs row included), B their coupling plus the rhs of the Euclidean rows as
// column npp.  With E = L_E L_E' (L_E block lower-bidiagonal) and W = L_E^-1 B the poses see S - W'W.  Three pipelines run side by
// side, one barrier per step:
//   T-wave (wave 3)       step s: D_s = E_ss - L_s,s-1 L_s,s-1', L_ss = chol(D_s), L_s+1,s = E_s+1,s L_ss^-T  (serial, 18 x 9 panel in
//                         registers, one row per lane, pivots broadcast with v_readlane); one step ahead of the others;
//   column owners         thread c < npp + 1 owns column c of W for the whole elimination: w_s = L_ss^-1 (b_s - L_s,s-1 w_s-1) with
//   (waves 0, 1)          w_s-1 in registers -- no cross-lane traffic, L blocks read from LDS as broadcasts; w_s goes to an LDS buffer
//                         for the matrix cores and to the spill area for the back-substitution;
//   matrix cores          S -= W_s-1' W_s-1 on v_mfma_f64_16x16x4 over the tile pairs that hold coupled columns (wave 2, which has no
//                         other duty).
// Every quantity is the one the textbook right-looking block Cholesky in this order produces; only the schedule differs.
// q accumulates u' H u over the original entries of E and B (Cauchy point).
// LDS pool during the chain: [W buffer 0 | W buffer 1 | per step L_ss, 1/diag, L_next | T workspace | step records]
struct ChainLds { lds_d *wbuf, *ltab, *ta; lds_i *tab; int wld; };
template <int NT>
__device__ __forceinline__ ChainLds chain_lds(const Ctx<NT> &C) {
    ChainLds L;
    L.wld = 16 * C.ntd;
    L.wbuf = C.stage;
    L.ltab = C.stage + 2 * CH_W * L.wld;
    L.ta = L.ltab + C.P->n_e * CH_LT;
    L.tab = (lds_i *)(L.ta + CH_TA);
    return L;
}

// original (unscaled) value of front row r, column i of step h: sum of <= 2 IMU factor blocks (parked in HBM/L2 by the
// linearisation) and the cached J0'J0 of the prior.  The loads are UNCONDITIONAL (offset 0 when a source is absent) and masked
// afterwards: a conditional load becomes an exec-mask branch with its own s_waitcnt vmcnt(0), which serialises the loads of a
// fetch at full memory latency (27 round trips per column owner and step instead of one).
struct ChainSrc { const gbl_d *imu, *hp; };
template <int NT>
__device__ __forceinline__ void chain_entry_fetch(const Ctx<NT> &C, const ChainSrc &G, const lds_i *h, int r, int i, bool valid, double (&v)[3]) {
    const int nsrc = h[CH_NSRC], f0 = h[CH_F0], lc0 = h[CH_LC0] + i, f1 = h[CH_F1], lc1 = h[CH_LC1] + i, pc = h[CH_PC0] + i;
    const unsigned w0 = (unsigned)h[CH_INTS + 2 * r];
    const int pr = h[CH_INTS + 2 * r + 1];
    const int l0 = (w0 >> 16) & 255, l1 = w0 >> 24;
    bool u0 = valid && nsrc > 0 && l0 != 255, u1 = valid && nsrc > 1 && l1 != 255, u2 = valid && h[CH_PC0] >= 0 && pr >= 0;
#ifdef TCV_ABLATE
    if (!ABL(C, AB_CH_FETCH)) u0 = u1 = u2 = false;
#endif
    const int pa = max(pr, pc), pb = min(pr, pc);
    const int o0 = u0 ? f0 * IMU_BLK + max(l0, lc0) * 32 + min(l0, lc0) : 0;
    const int o1 = u1 ? f1 * IMU_BLK + max(l1, lc1) * 32 + min(l1, lc1) : 0;
    const int o2 = u2 ? pa * (pa + 1) / 2 + pb : 0;
    const double x0 = G.imu[o0], x1 = G.imu[o1], x2 = G.hp[o2];
    v[0] = u0 ? x0 : 0.0; v[1] = u1 ? x1 : 0.0; v[2] = u2 ? x2 : 0.0;
}

// T-wave: original entries of the 18 x 9 panel [E_ss; E_s+1,s] of step s, three per lane, fetched one step ahead of their use
template <int NT>
__device__ __forceinline__ void chain_t_fetch(const Ctx<NT> &C, const ChainSrc &G, const ChainLds &L, int s, int lane, double (&v)[3][3]) {
    const lds_i *h = L.tab + s * CH_STRIDE;
    const int nrow = h[CH_NEXT] ? 2 * CH_W : CH_W;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int e = min(lane + 64 * k, 2 * CH_W * CH_W - 1), r = e / CH_W;
        chain_entry_fetch<NT>(C, G, h, r, e - r * CH_W, lane + 64 * k < nrow * CH_W, v[k]);
    }
}

// T-wave, step s: lanes 0..8 hold the rows of D_s, lanes 9..17 the rows of E_s+1,s, lanes 18..26 the rows of the identity; one panel
// factorisation (x <- x L_ss^-T for every row x below the block) gives L_ss, L_s+1,s = E_s+1,s L_ss^-T and L_ss^-T, whose transpose
// the W waves and the back-substitution multiply with.  Returns false (uniform within the wave) on a non-positive pivot.
template <int NT>
__device__ __forceinline__ bool chain_t_step(const Ctx<NT> &C, const ChainLds &L, int s, double mu, double &q, int lane, const double (&v)[3][3]) {
    const lds_i *h = L.tab + s * CH_STRIDE;
    const int t0 = h[CH_T0], npp = C.P->npp, has_next = h[CH_NEXT];
    const int nrow = has_next ? 2 * CH_W : CH_W;
    lds_d *ta = L.ta;
    // (1) original entries, Jacobi-scaled, + mu D^2 on the diagonal (D exactly as finalize computes it); the entries of the diagonal
    // block take the fill of the previous step, - L_s,s-1 L_s,s-1' (rows = this block, columns = the previous one), here, one 9-term
    // product per lane out of LDS, instead of on the row-per-lane layout of the factorisation
    const bool prev = s > 0 && (L.tab + (s - 1) * CH_STRIDE)[CH_NEXT] != 0;
    const lds_d *N = L.ltab + (max(s, 1) - 1) * CH_LT + CH_LN;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int e = lane + 64 * k;
        if (e < nrow * CH_W) {
            const int r = e / CH_W, c = e - r * CH_W, tc = t0 + c;
            const int tr = h[CH_INTS + 2 * r] & 255;
            const double vv = (v[k][0] + v[k][1]) + v[k][2];
            const double scc = C.sc[tc];
            const double w = (r < CH_W) ? ((r == c) ? 1.0 : ((r > c) ? 2.0 : 0.0)) : 2.0;
            q += w * C.ycam[tr] * C.ycam[tc] * vv;
            double add = C.sc[tr] * scc * vv;
            if (r == c) add += mu * fmin(fmax(scc * scc * C.hd[tc - npp], 1e-6), 1e32);
            if (k < 2 && prev && r < CH_W) {
                const int rr = min(r, CH_W - 1);
                double a0 = 0.0, a1 = 0.0;
#pragma unroll
                for (int j = 0; j + 1 < CH_W; j += 2) { a0 += N[rr * CH_W + j] * N[c * CH_W + j]; a1 += N[rr * CH_W + j + 1] * N[c * CH_W + j + 1]; }
                a0 += N[rr * CH_W + CH_W - 1] * N[c * CH_W + CH_W - 1];
                add -= a0 + a1;
            }
            ta[e] = add;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // (2) one row per lane
    const int r = min(lane, nrow - 1);
    double t[CH_W];
#pragma unroll
    for (int c = 0; c < CH_W; c++) { const double tv = ta[r * CH_W + c]; t[c] = (lane >= 2 * CH_W) ? ((lane - 2 * CH_W == c) ? 1.0 : 0.0) : tv; }
    // (3) panel Cholesky: pivot k lives in lane k; column k of every row is scaled, the trailing columns of every row updated
    bool ok = true;
#pragma unroll
    for (int k = 0; k < CH_W; k++) {
        const double d = readlane_f64(t[k], k);
        if (!(d > 0.0) || !(d < 1e300)) ok = false;
        double l, y;
        sqrt_rsqrt(ok ? d : 1.0, l, y);
        const double lk = (lane == k) ? l : t[k] * y;
        t[k] = lk;
#pragma unroll
        for (int c = k + 1; c < CH_W; c++) t[c] -= lk * readlane_f64(lk, c);
    }
    lds_d *out = L.ltab + s * CH_LT;
    if (lane >= 2 * CH_W && lane < 3 * CH_W) {      // row i of L^-T = column i of L^-1
#pragma unroll
        for (int c = 0; c < CH_W; c++) out[c * CH_W + (lane - 2 * CH_W)] = t[c];
    } else if (lane >= CH_W && lane < nrow) {
#pragma unroll
        for (int c = 0; c < CH_W; c++) out[CH_LN + (lane - CH_W) * CH_W + c] = t[c];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return ok;
}

// W waves: original entries of B_s in the accumulator layout of a 16-column tile -- lane (row0 = lane >> 4, col = lane & 15) holds
// rows row0, row0 + 4, row0 + 8 of column 16 J + col -- one step ahead of their use.  A pose row never lies inside the local range of
// the Euclidean block in a factor or in the prior, so each source is base + m * stride (IMU block: stride 32 below the block's
// columns, 1 above; prior triangle: consecutive entries of row pr above, column entries (pc + m, pr) below).
template <int NT>
__device__ __forceinline__ void chain_tile_fetch(const Ctx<NT> &C, const ChainSrc &G, const lds_i *h, int rc, int row0, double (&v)[3][3]) {
    const bool valid = rc != 255;
    const int r = min(rc, CH_MAXROWS - 1);
    const int nsrc = h[CH_NSRC], lc0 = h[CH_LC0], lc1 = h[CH_LC1], pc = h[CH_PC0];
    const unsigned w0 = (unsigned)h[CH_INTS + 2 * r];
    const int pr = h[CH_INTS + 2 * r + 1];
    const int l0 = (w0 >> 16) & 255, l1 = w0 >> 24;
    bool u0 = valid && nsrc > 0 && l0 != 255, u1 = valid && nsrc > 1 && l1 != 255, u2 = valid && pc >= 0 && pr >= 0;
#ifdef TCV_ABLATE
    if (!ABL(C, AB_CH_FETCH)) u0 = u1 = u2 = false;
#endif
    const int b0 = u0 ? h[CH_F0] * IMU_BLK + (l0 < lc0 ? lc0 * 32 + l0 : l0 * 32 + lc0) : 0, s0 = (u0 && l0 < lc0) ? 32 : (u0 ? 1 : 0);
    const int b1 = u1 ? h[CH_F1] * IMU_BLK + (l1 < lc1 ? lc1 * 32 + l1 : l1 * 32 + lc1) : 0, s1 = (u1 && l1 < lc1) ? 32 : (u1 ? 1 : 0);
    const bool below = pr < pc;
    double x0[3], x1[3], x2[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int m = min(row0 + 4 * i, CH_W - 1), pm = pc + m;
        const int o2 = u2 ? (below ? pm * (pm + 1) / 2 + pr : pr * (pr + 1) / 2 + pm) : 0;
        x0[i] = G.imu[b0 + m * s0]; x1[i] = G.imu[b1 + m * s1]; x2[i] = G.hp[o2];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) { v[i][0] = u0 ? x0[i] : 0.0; v[i][1] = u1 ? x1[i] : 0.0; v[i][2] = u2 ? x2[i] : 0.0; }
}

// S -= W' W for the tile pairs (I >= J) whose tile rows hold coupled columns.  A wave runs its pairs two at a time so that the
// operand loads, the dependent MFMA chains and the read-modify-writes of both overlap.  Everything that depends on the lane only
// (operand rows, k < 9 masks, swizzled C offsets) is computed once per elimination (ChainMfmaLane): a trip is 12 operand loads,
// 8 + 8 tile accesses, 6 MFMAs and a handful of address additions.
struct ChainMfmaLane { int ko[3], co[4]; bool kin[3]; int col; };
__device__ __forceinline__ ChainMfmaLane chain_mfma_lane(int lane, int wld) {
    ChainMfmaLane M;
    const int row0 = lane >> 4;
    M.col = lane & 15;
#pragma unroll
    for (int kk = 0; kk < 3; kk++) { const int k = 4 * kk + row0; M.kin[kk] = k < CH_W; M.ko[kk] = min(k, CH_W - 1) * wld + M.col; }
#pragma unroll
    for (int i = 0; i < 4; i++) M.co[i] = sw(row0 + 4 * i, M.col);
    return M;
}
__device__ __forceinline__ void chain_mfma_pair2(lds_d *tiles, const lds_d *W, const ChainMfmaLane &M, int npp, int I0, int J0, int I1, int J1, bool two) {
    lds_d *C0 = tiles + tbase(I0, J0), *C1 = tiles + tbase(I1, J1);
    const lds_d *WI0 = W + 16 * I0, *WJ0 = W + 16 * J0, *WI1 = W + 16 * I1, *WJ1 = W + 16 * J1;
    const bool rhs0 = 16 * J0 + M.col == npp, rhs1 = 16 * J1 + M.col == npp;      // the rhs is a row of the tiles, never a column
    double a0[3], b0[3], a1[3], b1[3];
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
        const double ta0 = WI0[M.ko[kk]], tb0 = WJ0[M.ko[kk]], ta1 = WI1[M.ko[kk]], tb1 = WJ1[M.ko[kk]];
        a0[kk] = M.kin[kk] ? -ta0 : 0.0;
        b0[kk] = (M.kin[kk] && !rhs0) ? tb0 : 0.0;
        a1[kk] = M.kin[kk] ? -ta1 : 0.0;
        b1[kk] = (M.kin[kk] && !rhs1) ? tb1 : 0.0;
    }
    v4f64 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 4; i++) { acc0[i] = C0[M.co[i]]; acc1[i] = C1[M.co[i]]; }
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[kk], b0[kk], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[kk], b1[kk], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { C0[M.co[i]] = acc0[i]; if (two) C1[M.co[i]] = acc1[i]; }
}
// every pair (a, b), a >= b, over the ascending list of active tile rows, two at a time
template <int NT>
__device__ __forceinline__ void chain_mfma_update_all(const Ctx<NT> &C, const lds_d *W, const ChainMfmaLane &M, int tmask, int npp) {
    int act = 0, na = 0;      // 4 bits per entry
    for (int I = 0; I < C.ntd; I++) if ((tmask >> I) & 1) { act |= I << (4 * na); na++; }
    const int npair = na * (na + 1) / 2;
    int a = 0, b = 0, pI = -1, pJ = -1;
    for (int p = 0; p < npair; p++) {
        const int I = (act >> (4 * a)) & 15, J = (act >> (4 * b)) & 15;
        if (pI < 0) { pI = I; pJ = J; }
        else { chain_mfma_pair2(C.tiles, W, M, npp, pI, pJ, I, J, true); pI = -1; }
        if (b == a) { a++; b = 0; } else b++;
    }
    if (pI >= 0) chain_mfma_pair2(C.tiles, W, M, npp, pI, pJ, pI, pJ, false);
}

#ifdef TCV_PROFILE
#define CH_TIC() const long long t_role = clock64()
#define CH_TOC(slot, cond) do { if (cond) ((lds_u *)(C.red + 40))[slot] += (unsigned)(clock64() - t_role); } while (0)
#else
#define CH_TIC() do { } while (0)
#define CH_TOC(slot, cond) do { } while (0)
#endif
struct ChainOut { double q; bool ok; };
template <int NT>
__device__ __noinline__ ChainOut chain_forward(TCV_CTX_PARAMS, double mu) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    static_assert(NT >= 256, "chain layout: four wavefronts (column owners on waves 0-1, matrix cores on 0-2, T pipeline on 3)");
    cst_plan &P = *C.P;
    const int tid = C.tid, lane = tid & 63, wave = tid >> 6, npp = P.npp;
#ifdef TCV_ABLATE
    int ne = P.n_e;
#else
    const int ne = P.n_e;
#endif
    const ChainLds L = chain_lds<NT>(C);
    const int wld = L.wld;
    const ChainSrc G = {C.g_imublk, C.g_hp};
    const ChainMfmaLane ML = chain_mfma_lane(lane, wld);
    double q = 0.0;
    copy_prog<NT>(L.tab, C.ip + P.o_chain, ne * CH_STRIDE, tid);
    for (int i = tid; i < 2 * CH_W * wld; i += NT) L.wbuf[i] = 0.0;
    if (tid == 0) *C.flag = 0;
    __syncthreads();
#ifdef TCV_ABLATE
    if (!ABL(C, AB_CH_HALF)) {      // timing experiment (results are garbage): records 0, 2, 4, ... compacted to the front, the pipelines run over those only
        const int nh = (ne + 1) / 2;
        for (int k = 1; k < nh; k++) {
            int v[(CH_STRIDE + NT - 1) / NT];
            for (int q = 0, i = tid; i < CH_STRIDE; i += NT, q++) v[q] = L.tab[2 * k * CH_STRIDE + i];
            __syncthreads();
            for (int q = 0, i = tid; i < CH_STRIDE; i += NT, q++) L.tab[k * CH_STRIDE + i] = v[q];
            __syncthreads();
        }
        if (tid == 0) (L.tab + (nh - 1) * CH_STRIDE)[CH_NEXT] = 0;
        __syncthreads();
        ne = nh;
    }
#endif
    // W-wave state: wave w < 2 owns the 16-column tiles w, w + 2, w + 4 of W; per tile the previous W rows (accumulator layout: lane
    // (row0, col) holds rows row0 + 4 i of column 16 J + col) and the prefetched entries of the next B rows
    constexpr int WT = 3;
    const int row0 = lane >> 4, col = lane & 15;
    double wp[WT][4], vb[WT][3][3];
    int rcs[WT];
#pragma unroll
    for (int jt = 0; jt < WT; jt++) {
#pragma unroll
        for (int i = 0; i < 4; i++) wp[jt][i] = 0.0;
        rcs[jt] = 255;
        if (wave < 2) {
            const int cc = 16 * (wave + 2 * jt) + col;
            if (wave + 2 * jt < C.ntd && cc <= npp) rcs[jt] = ((const __attribute__((address_space(3))) unsigned char *)(L.tab + CH_COLROW))[cc];
            chain_tile_fetch<NT>(C, G, L.tab, rcs[jt], row0, vb[jt]);
        }
    }
    double vt[3][3], vtn[3][3];      // T-wave: the panel entries of the step being factored / of the one after it
    if (wave == 3) {
        chain_t_fetch<NT>(C, G, L, 0, lane, vt);
        if (ne > 1) chain_t_fetch<NT>(C, G, L, 1, lane, vtn);
        if (!chain_t_step<NT>(C, L, 0, mu, q, lane, vt) && lane == 0) *C.flag = 1;
    }
    __syncthreads();
    for (int s = 0; s <= ne; s++) {
#ifdef TCV_ABLATE
        if (*C.flag && !ABL_FORCE(C)) { ChainOut bad; bad.q = 0.0; bad.ok = false; return bad; }
#else
        if (*C.flag) { ChainOut bad; bad.q = 0.0; bad.ok = false; return bad; }      // uniform: read after a barrier, written before it
#endif
        // ---- T pipeline: one step ahead
        CH_TIC();
        if (wave == 3 && s + 1 < ne && ABL(C, AB_CH_T)) {
#pragma unroll
            for (int k = 0; k < 3; k++) { vt[k][0] = vtn[k][0]; vt[k][1] = vtn[k][1]; vt[k][2] = vtn[k][2]; }
            if (s + 2 < ne) chain_t_fetch<NT>(C, G, L, s + 2, lane, vtn);
            if (!chain_t_step<NT>(C, L, s + 1, mu, q, lane, vt) && lane == 0) *C.flag = 1;
            CH_TOC(PH_CH_A, tid == 192);
        }
        // ---- W waves: W_s = L_ss^-1 (B_s - L_s,s-1 W_s-1), a 16-column tile at a time on the matrix cores.  The output layout of
        // v_mfma_f64_16x16x4 (lane holds rows row0 + 4 i of column col) IS its B-operand layout (rows 4 kk + row0), so W_s-1 and the
        // intermediate product feed the next MFMA straight from the accumulator registers.
        if (s < ne && wave < 2 && ABL(C, AB_CH_W)) {
            const lds_i *h = L.tab + s * CH_STRIDE;
            const int t0 = h[CH_T0], tmask = h[CH_TMASK];
            const bool prev = s > 0 && (h - CH_STRIDE)[CH_NEXT] != 0;
            const lds_d *Li = L.ltab + s * CH_LT, *Np = L.ltab + (max(s, 1) - 1) * CH_LT + CH_LN;
            double aL[3], aN[3];      // A operands: lane holds A[m = col][k = 4 kk + row0]
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                const int k = 4 * kk + row0, o = min(col, CH_W - 1) * CH_W + min(k, CH_W - 1);
                const double tl = Li[o], tn = Np[o];
                const bool in = col < CH_W && k < CH_W;
                aL[kk] = in ? tl : 0.0;
                aN[kk] = (in && prev) ? -tn : 0.0;
            }
            double scm[3], um[3], gm[3];
#pragma unroll
            for (int i = 0; i < 3; i++) { const int m = min(row0 + 4 * i, CH_W - 1); scm[i] = C.sc[t0 + m]; um[i] = C.ycam[t0 + m]; gm[i] = C.gcam[t0 + m]; }
            lds_d *Wb = L.wbuf + (s & 1) * CH_W * wld;
#pragma unroll
            for (int jt = 0; jt < WT; jt++) {
                const int J = wave + 2 * jt;
                if (J >= C.ntd) continue;
                const int cc = 16 * J + col;
                if ((tmask >> J) & 1) {      // else: no coupled column in this tile yet, W stays zero
                    const bool act = rcs[jt] != 255, rhs = cc == npp;
                    const double scc = (act && !rhs) ? C.sc[min(cc, CAM_MAX - 1)] : 0.0, uc = (act && !rhs) ? C.ycam[min(cc, CAM_MAX - 1)] : 0.0;
                    v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < 3; i++) {      // branch-free: selects only (rows 9..11 of the accumulator stay zero)
                        const bool in = act && row0 + 4 * i < CH_W;
                        const double vv = (vb[jt][i][0] + vb[jt][i][1]) + vb[jt][i][2];
                        q += in ? 2.0 * uc * um[i] * vv : 0.0;
                        const double bv = rhs ? scm[i] * gm[i] : scc * scm[i] * vv;
                        acc[i] = in ? bv : 0.0;
                    }
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aN[kk], wp[jt][kk], acc, 0, 0, 0);      // aN = 0 without a previous step
                    v4f64 w4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int kk = 0; kk < 3; kk++) w4 = __builtin_amdgcn_mfma_f64_16x16x4f64(aL[kk], acc[kk], w4, 0, 0, 0);
                    // rows row0, row0 + 4 always exist; row0 + 8 only for row0 = 0.  The spill area holds 16 ntd columns per step.
                    gbl_d *sp = C.g_spill + h[CH_SPILL] + cc * CH_W + row0;
                    lds_d *wo = Wb + row0 * wld + cc;
                    wo[0] = w4[0]; wo[4 * wld] = w4[1]; sp[0] = w4[0]; sp[4] = w4[1];
                    if (row0 == 0) { wo[8 * wld] = w4[2]; sp[8] = w4[2]; }
#pragma unroll
                    for (int i = 0; i < 4; i++) wp[jt][i] = w4[i];
                } else {
                    // W_s of a tile without a coupled column IS zero -- and must be in the registers the next step multiplies with: along
                    // one chain the active tiles only grow, so an inactive tile had never been written, but a window whose IMU chain is
                    // broken (a pre-integration over 10 s is left out, estimator.cpp:1726) starts a second chain with fewer active tiles,
                    // and the block the prior couples to every pose (para_SpeedBias[0]) then picked up the FIRST chain's rows here
#pragma unroll
                    for (int i = 0; i < 4; i++) wp[jt][i] = 0.0;
                }
                if (s + 1 < ne) {      // the next step's entries, whether or not the tile is active yet
                    rcs[jt] = cc <= npp ? ((const __attribute__((address_space(3))) unsigned char *)(h + CH_STRIDE + CH_COLROW))[cc] : 255;
                    chain_tile_fetch<NT>(C, G, h + CH_STRIDE, rcs[jt], row0, vb[jt]);
                }
            }
            CH_TOC(PH_CH_B, tid == 0);
        }
        // ---- matrix cores: S -= W_s-1' W_s-1
        if (s > 0 && wave == 2 && ABL(C, AB_CH_MFMA)) {
            // wave 2 (no other duty) takes every pair.  Which wavefront a step waits for changes along the chain (timers of the -DTCV_PROFILE
            // build at the step's barrier): the early steps, with two or three active tile rows, wait for the T pipeline, the late ones for
            // the W tiles; sharing the pairs out to the W wavefronts (thirds, sevenths, ... measured) never shortened a step
            const lds_i *hp = L.tab + (s - 1) * CH_STRIDE;
            const lds_d *Wp = L.wbuf + ((s - 1) & 1) * CH_W * wld;
            if (wave == 2) chain_mfma_update_all<NT>(C, Wp, ML, hp[CH_TMASK], npp);
            CH_TOC(PH_CH_C, tid == 128);
        }
#ifdef TCV_PROFILE
        const long long t_arrive = clock64();
#endif
        __syncthreads();
#ifdef TCV_PROFILE
        // how long each wavefront waits at the step's barrier: the one that waits least sets the step (slots: chain_bwd's -- unused here -- and 29..31)
        if (lane == 0) ((lds_u *)(C.red + 40))[wave == 0 ? (int)PH_CHAIN_BWD : 28 + wave] += (unsigned)(clock64() - t_arrive);
#endif
        CH_TOC(PH_CH_D, tid == 64);      // the whole interval as wave 1 sees it
    }
    ChainOut o;
    o.q = block_sum<NT>(q, C.red, tid);
    o.ok = true;
    __syncthreads();
    return o;
}

// back-substitution through the chain: y_s = L_ss^-T (z_s - W_s y_p - L_s+1,s' y_s+1), blocks in reverse elimination order; y_p
// (poses) is in ycam, z_s is column npp of W_s.  chain_products (all threads): the W_s y_p products of every step at once, two
// threads per row, W streamed from the spill area with every load in flight together.  chain_backward (ONE wavefront, the other
// waves back-substitute the landmarks meanwhile): the serial recursion, one lane per row of the block, L_ss^-1 and L_s+1,s still in LDS.
template <int NT>
__device__ __forceinline__ void chain_products(Ctx<NT> &C) {
    cst_plan &P = *C.P;
    const int npp = P.npp, ne = P.n_e;
    const ChainLds L = chain_lds<NT>(C);
    lds_d *pp = L.wbuf;           // partial products, 2 per row (the W buffers are dead)
    for (int idx = C.tid; idx < 2 * ne * CH_W; idx += NT) {
        const int e = idx >> 1, part = idx & 1, s = e / CH_W, i = e - s * CH_W;
        const gbl_d *sp = C.g_spill + (L.tab + s * CH_STRIDE)[CH_SPILL] + i;
        const int tm = (L.tab + s * CH_STRIDE)[CH_TMASK];      // tiles without a coupled column were never written: their W is zero
        double a0 = 0.0, a1 = 0.0;
        int cc = part;
#pragma unroll 8
        for (; cc + 2 < npp; cc += 4) {
            const double w0 = sp[cc * CH_W], w1 = sp[(cc + 2) * CH_W];
            a0 += (((tm >> (cc >> 4)) & 1) ? w0 : 0.0) * C.ycam[cc]; a1 += (((tm >> ((cc + 2) >> 4)) & 1) ? w1 : 0.0) * C.ycam[cc + 2];
        }
        for (; cc < npp; cc += 2) { const double w0 = sp[cc * CH_W]; a0 += (((tm >> (cc >> 4)) & 1) ? w0 : 0.0) * C.ycam[cc]; }
        pp[idx] = (part == 0 ? sp[npp * CH_W] : 0.0) - (a0 + a1);
    }
    __syncthreads();
}
template <int NT>
__device__ __noinline__ bool chain_backward(TCV_CTX_PARAMS) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    cst_plan &P = *C.P;
    const int lane = C.tid & 63, ne = P.n_e;
    const ChainLds L = chain_lds<NT>(C);
    const lds_d *pp = L.wbuf;
    bool bad = false;
    const int i = min(lane, CH_W - 1);
    double ynext = 0.0;      // lane j: y_s+1[j]
    for (int s = ne - 1; s >= 0; s--) {
        const lds_i *h = L.tab + s * CH_STRIDE;
        const lds_d *Ls = L.ltab + s * CH_LT;
        // column i of L_ss^-1 and of L_s+1,s, up front
        double lc[CH_W], nc[CH_W];
#pragma unroll
        for (int k = 0; k < CH_W; k++) { lc[k] = Ls[k * CH_W + i]; nc[k] = Ls[CH_LN + k * CH_W + i]; }
        double r = pp[2 * (s * CH_W + i)] + pp[2 * (s * CH_W + i) + 1];
        if (h[CH_NEXT]) {      // - L_s+1,s' y_s+1
#pragma unroll
            for (int j = 0; j < CH_W; j++) r -= nc[j] * readlane_f64(ynext, j);
        }
        double y0 = 0.0, y1 = 0.0;      // y = L_ss^-T r
#pragma unroll
        for (int k = 0; k + 1 < CH_W; k += 2) { y0 += lc[k] * readlane_f64(r, k); y1 += lc[k + 1] * readlane_f64(r, k + 1); }
        y0 += lc[CH_W - 1] * readlane_f64(r, CH_W - 1);
        const double y = y0 + y1;
        if (lane < CH_W) {
            C.ycam[h[CH_T0] + lane] = y;
            C.v_y[h[CH_T0] + lane] = y;
            if (!(fabs(y) < 1e300)) bad = true;
        }
        ynext = y;
    }
    return __ballot(bad) == 0ull;
}

// ---- scale, regularise, factorise and solve (J'J + mu D^2) y = J'r ---------------------------------
// On return (true): v_y = y (scaled space, camera then landmarks), v_D, v_ghat set, scal = {gg, q}.
struct FinOut { double gg, q; bool ok; };
template <int NT, bool MFMA, bool CHAIN>
// (disable_tail_calls, here and on the kernels: a call whose arguments are all values gets the IR `tail` marker, and LLVM's inter-procedural
// register allocation then makes every callee save the callee-saved VGPRs it touches -- 147 scratch stores and 134 loads per call of
// linearize, +1.2 GB of scratch traffic per launch; without the marker the callees save nothing and the caller keeps what it needs)
__device__ __noinline__ __attribute__((disable_tail_calls)) FinOut finalize_and_solve(TCV_CTX_PARAMS, bool first, double mu) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    cst_plan &P = *C.P;
    const int tid = C.tid, nc = P.nc, L = P.nland;
    const int nd = C.nd;      // dimension of the dense system in the tiles: nc, or npp in chain mode
    cst_i *ip = C.ip;
    if (ABL(C, AB_FIN_SCALE)) {
    for (int a = tid; a < nc; a += NT) {
        const double sdv = C.sd[min(a, 87)];
        const double hdv = CHAIN ? C.hd[min(max(a - P.npp, 0), 111)] : 0.0;
        const double tdv = C.tiles[tix(min(a, nd - 1), min(a, nd - 1))];
        const double dH = (CHAIN && a >= P.npp) ? hdv : tdv + (a < P.npp ? sdv : 0.0);
        double s;
        if (first) { s = 1.0 / (1.0 + sqrt(dH)); C.v_s[a] = s; }
        else s = C.v_s[a];
        C.sc[a] = s;
        const double d2 = fmin(fmax(s * s * dH, 1e-6), 1e32);
        const double D = sqrt(d2);
        C.v_D[a] = D;
        const double gh = s * C.gcam[a] / D;
        C.v_ghat[a] = gh;
        C.ycam[a] = s * gh / D;  // u = s * (ghat / D): Cauchy direction in unscaled tangent units
    }
    for (int a = nc + tid; a < TCV_CAMW(*C.P); a += NT) { C.ycam[a] = 0.0; C.sc[a] = 0.0; }
    for (int l = tid; l < L; l += NT) {
        const double s = C.v_s[nc + l], h = C.l_hll[l];
        const double d2 = fmin(fmax(s * s * h, 1e-6), 1e32), D = sqrt(d2);
        C.v_D[nc + l] = D;
        C.v_ghat[nc + l] = s * C.l_gl[l] / D;
    }
    }
    __syncthreads();
    TCV_MARK(C, PH_FIN_SCALE);
    // Cauchy point: gg = |ghat|^2, q = |J (ghat / D)|^2 = u' H u with H = [S~ + sum Hcl Hcl'/kappa, Hcl; Hcl', hll].
    // The camera part u_c' S~ u_c is accumulated in the same sweep over the tiles that applies the Jacobi scaling,
    // adds mu D^2, writes the rhs row and the identity padding.
    double acc[2] = {0.0, 0.0};
    for (int a = tid; a < nc; a += NT) { const double gh = C.v_ghat[a]; acc[0] += gh * gh; }
    if (ABL(C, AB_FIN_SCALE)) {
        cst_i *lm = ip + P.o_lm, *sp = ip + P.o_lmslotptr, *so = ip + P.o_lmslot;
        for (int l = tid; l < L; l += NT) {
            const gbl_d *h = C.g_hcl + lm[2 * l];
            double t = 0;
            const int s0 = sp[l], s1 = sp[l + 1];
            for (int s = s0; s < s1; s++) {
                const int off = so[s];
#pragma unroll
                for (int e = 0; e < 6; e++) t += h[(s - s0) * 6 + e] * C.ycam[off + e];
            }
            const double gh = C.v_ghat[nc + l];
            const double ul = C.v_s[nc + l] * gh / C.v_D[nc + l];
            acc[1] += t * t * C.l_invk[l] + 2.0 * t * ul + C.l_hll[l] * ul * ul;
            acc[0] += gh * gh;
        }
    }
    TCV_MARK(C, PH_FIN_CAUCHY);
    if (ABL(C, AB_FIN_PASS)) {
        constexpr int NW = NT / 64;
        const int lane = tid & 63, wave = tid >> 6;
        const int r = lane >> 2, c0 = (lane & 3) << 2;
        int t = 0;
        for (int I = 0; I < C.ntd; I++)
            for (int J = 0; J <= I; J++, t++) {
                if ((t % NW) != wave) continue;
                lds_d *T = C.tiles + (t << 8);
                const int a = 16 * I + r;
                double tv[4];
#pragma unroll
                for (int i = 0; i < 4; i++) tv[i] = T[sw(r, c0 + i)];
                if (a < nd) {
                    const double sa = C.sc[a], ua = C.ycam[a], sda = C.sd[min(a, 87)];
                    double sb[4], ub[4];
#pragma unroll
                    for (int i = 0; i < 4; i++) { sb[i] = C.sc[16 * J + c0 + i]; ub[i] = C.ycam[16 * J + c0 + i]; }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int b = 16 * J + c0 + i;
                        double v = 0.0;
                        if (b <= a) {
                            acc[1] += ((a == b) ? 1.0 : 2.0) * tv[i] * ua * ub[i];
                            v = sa * sb[i] * tv[i];
                            if (a == b) v += mu * fmin(fmax(sa * sa * (tv[i] + (a < P.npp ? sda : 0.0)), 1e-6), 1e32);   // mu D_a^2
                        }
                        T[sw(r, c0 + i)] = v;
                    }
                } else if (a == nd) {
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const int b = 16 * J + c0 + i;
                        const double scb = C.sc[b], gb = C.gcam[b], rcb = C.rc[min(b, 87)];      // sc/gcam hold 176 entries
                        T[sw(r, c0 + i)] = (b < nd) ? scb * (gb - (b < P.npp ? rcb : 0.0)) : ((b == nd) ? 1.0 : 0.0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 4; i++) T[sw(r, c0 + i)] = (16 * J + c0 + i == a) ? 1.0 : 0.0;
                }
            }
    }
    block_sum_n<NT, 2>(acc, C.red, tid);
    FinOut out;
    out.gg = acc[0]; out.q = acc[1]; out.ok = false;
    if (tid == 0) *C.flag = 0;
    __syncthreads();
    TCV_MARK(C, PH_FIN_PASS);
    if (CHAIN) {
        if (ABL(C, AB_CHAIN_FWD)) {
            const ChainOut co = chain_forward<NT>(TCV_CTX_ARGS(C), mu);
            if (!co.ok) return out;
            out.q += co.q;
        }
        TCV_MARK(C, PH_CHAIN_FWD);
    }
    if (ABL(C, AB_CHOL) && !chol_tiles<NT, MFMA>(TCV_CTX_ARGS(C), C.ntd, nd)) return out;
    if (ABL(C, AB_BACK)) back_subst<NT>(TCV_CTX_ARGS(C), nd);
    TCV_MARK(C, PH_BACK);
    // landmarks: y_l = (gl - Hcl' (s o y_c)) / (s_l kappa_l)
    bool bad = false;
    for (int a = tid; a < nd; a += NT) {
        const double y = C.ycam[a];
        C.v_y[a] = y;
        if (!(fabs(y) < 1e300)) bad = true;
    }
    // chain mode: wave 0 walks the chain backwards (Euclidean blocks in reverse elimination order) while the other
    // waves back-substitute the landmarks, which only meet pose-kind blocks
    if (CHAIN && ABL(C, AB_CHAIN_BWD) && ABL(C, AB_CHAIN_FWD)) chain_products<NT>(C);
    const int l_first = CHAIN ? tid - 64 : tid, l_step = CHAIN ? NT - 64 : NT;
    if (CHAIN && tid < 64 && ABL(C, AB_CHAIN_BWD) && ABL(C, AB_CHAIN_FWD)) { if (!chain_backward<NT>(TCV_CTX_ARGS(C))) bad = true; }
    if ((!CHAIN || tid >= 64) && ABL(C, AB_LM_BACK)) {
        cst_i *lm = ip + P.o_lm, *sp = ip + P.o_lmslotptr, *so = ip + P.o_lmslot;
        for (int l = l_first; l < L; l += l_step) {
            const gbl_d *h = C.g_hcl + lm[2 * l];
            double t = 0;
            const int s0 = sp[l], s1 = sp[l + 1];
            for (int s = s0; s < s1; s++) {
                const int off = so[s];
#pragma unroll
                for (int e = 0; e < 6; e++) t += h[(s - s0) * 6 + e] * (C.sc[off + e] * C.ycam[off + e]);
            }
            const double y = (C.l_gl[l] - t) * C.l_invk[l] / C.v_s[nc + l];
            C.v_y[nc + l] = y;
            if (!(fabs(y) < 1e300)) bad = true;
        }
    }
#ifdef TCV_ABLATE
    if (ABL_FORCE(C)) bad = false;
#endif
    if (bad) *C.flag = 2;
    __syncthreads();
    int anybad = *C.flag;
    __syncthreads();
    TCV_MARK(C, PH_LM_BACK);
#ifdef TCV_ABLATE
    if (ABL_FORCE(C)) anybad = 0;
#endif
    out.ok = anybad == 0;
    return out;
}

// ---- ambient-space helpers ---------------------------------------------------------------------------
template <int NT>
__device__ __noinline__ void apply_plus(TCV_CTX_PARAMS, const lds_d *x, const gbl_d *delta_scaled, const gbl_d *s, lds_d *xo) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    // delta = step o scale; per block Plus (pose_local_parameterization.cpp:3-19) or x + delta
    cst_plan &P = *C.P;
    cst_i *blk = C.ip + P.o_blk;
    for (int b = C.tid; b < P.nblk; b += NT) {
        const int gs = blk[b * 4], go = blk[b * 4 + 1], lo = blk[b * 4 + 2], kind = blk[b * 4 + 3];
        if (lo < 0) {
            for (int i = 0; i < gs; i++) xo[go + i] = x[go + i];
        } else if (kind == KIND_POSE) {
            double d[6], xv[7], ov[7];
#pragma unroll
            for (int i = 0; i < 6; i++) d[i] = delta_scaled[lo + i] * s[lo + i];
#pragma unroll
            for (int i = 0; i < 7; i++) xv[i] = x[go + i];
            pose_plus(xv, d, ov);
#pragma unroll
            for (int i = 0; i < 7; i++) xo[go + i] = ov[i];
        } else {
            for (int i = 0; i < gs; i++) xo[go + i] = x[go + i] + delta_scaled[lo + i] * s[lo + i];
        }
    }
    for (int l = C.tid; l < P.nland; l += NT) xo[P.nx + l] = x[P.nx + l] + delta_scaled[P.nc + l] * s[P.nc + l];
}

struct Norms2 { double xn2, dn2; };      // returned in registers: reference outputs of a non-inlined function live in scratch memory
template <int NT>
__device__ __noinline__ Norms2 ambient_norms(TCV_CTX_PARAMS, const lds_d *x, const lds_d *xo) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    cst_plan &P = *C.P;
    cst_i *blk = C.ip + P.o_blk;
    double acc[2] = {0.0, 0.0};
    for (int b = C.tid; b < P.nblk; b += NT) {
        const int gs = blk[b * 4], go = blk[b * 4 + 1], lo = blk[b * 4 + 2];
        if (lo < 0) continue;
        for (int i = 0; i < gs; i++) {
            const double a = x[go + i], d = a - xo[go + i];
            acc[0] += a * a;
            acc[1] += d * d;
        }
    }
    for (int l = C.tid; l < P.nland; l += NT) {
        const double a = x[P.nx + l], d = a - xo[P.nx + l];
        acc[0] += a * a;
        acc[1] += d * d;
    }
    block_sum_n<NT, 2>(acc, C.red, C.tid);
    Norms2 o;
    o.xn2 = acc[0]; o.dn2 = acc[1];
    return o;
}

template <int NT>
__device__ __noinline__ double grad_max(TCV_CTX_PARAMS) {
    Ctx<NT> C = ctx_from_args<NT>(TCV_CTX_FORWARD);
    double m = 0;
    for (int i = C.tid; i < C.P->nc; i += NT) m = fmax(m, fabs(C.gcam[i]));
    for (int i = C.tid; i < C.P->nland; i += NT) m = fmax(m, fabs(C.v_g[C.P->nc + i]));
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o));
    __syncthreads();
    if ((C.tid & 63) == 0) C.red[C.tid >> 6] = m;
    __syncthreads();
    double t = 0;
    for (int w = 0; w < NT / 64; w++) t = fmax(t, C.red[w]);
    __syncthreads();
    return t;
}

// ---- the kernel ---------------------------------------------------------------------------------------
// COOP (cooperative mode, chain layout, tcv_packed.h): the grid holds 1 + coop_h workgroups per window GROUP, in blocks of eight groups.
// Inside a block workgroup r is member r / 8 of group r % 8: workgroups are dealt to the eight XCDs round-robin by index, so the master
// and the helpers of a group land on the same XCD and hand their exports over through its L2, and the members of a group are close
// together in dispatch order (a group is resident as a whole or waits as a whole; tcv_batch_solve additionally keeps the cooperative
// launches in flight within the CU count and runs the same plan with one workgroup per window otherwise).
// sqrt_info = LLT(cov^-1).matrixL()^T of every IMU factor of a window (imu_factor.h:64), one 16-lane group per factor
template <int NT>
__device__ __noinline__ void sqrt_info_all(cst_d *imu0_, int n_imu_, gbl_d *g_sqrt_, lds_d *lds_, int tid) {
    cst_d *imu0 = uni_ptr(imu0_);
    gbl_d *g_sqrt = uni_ptr(g_sqrt_);
    lds_d *ws = uni_ptr(lds_);
    const int n_imu = __builtin_amdgcn_readfirstlane(n_imu_);
    for (int f = tid >> 4; f < n_imu; f += NT / 16)
        (void)imu_sqrt_info_group(imu0 + f * IMU_CONST + IMU_COV, g_sqrt + f * 225, ws + f * 450, ws + f * 450 + 225, tid & 15);
}

// double2vector() (estimator.cpp:1537-1581) on the solved states while they are still in LDS (SolveArgs::gauge_fix): thread i < n_frames takes frame i, every
// thread recomputes rot_diff from the un-fixed pose 0, which is overwritten behind the barrier -- the arithmetic of tcv::gauge_batch_kernel (tcv_gauge.hip),
// the same bits.  A function of its own: atan2 / sin / cos in FP64 would otherwise sit in the kernel's register allocation.
template <int NT>
__device__ __noinline__ void gauge_epilogue(lds_d *xs_, cst_d *x_init_, cst_i *ft_, int n_frames_, int tid) {
    lds_d *xs = uni_ptr(xs_);
    cst_d *x_init = uni_ptr(x_init_);
    cst_i *ft = uni_ptr(ft_);
    const int n_frames = __builtin_amdgcn_readfirstlane(n_frames_);
    const bool act = tid < n_frames && n_frames > 0 && ft[0] >= 0 && ft[2 * min(tid, max(n_frames - 1, 0))] >= 0;
    M3 R; V3 Pn, V;
    int go = 0, so = -1;
    if (act) {
        const int g0 = ft[0];
        go = ft[2 * tid]; so = ft[2 * tid + 1];
        double p0i[7], p0[7], pi[7], vi[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < 7; k++) { p0i[k] = x_init[g0 + k]; p0[k] = xs[g0 + k]; pi[k] = xs[go + k]; }
        if (so >= 0) { vi[0] = xs[so]; vi[1] = xs[so + 1]; vi[2] = xs[so + 2]; }
        const M3 R0 = to_matrix(Quat(p0i + 3));      // Rs[0] before the solve: what vector2double() turned into para_Pose[0]
        const M3 rot = gauge_rot_diff(R0, p0);
        gauge_frame(rot, p0i, p0, pi, so >= 0 ? vi : nullptr, R, Pn, V);
    }
    __syncthreads();
    if (act) {
        const Quat q = r2q(R);
        xs[go] = Pn.x; xs[go + 1] = Pn.y; xs[go + 2] = Pn.z; xs[go + 3] = q.x; xs[go + 4] = q.y; xs[go + 5] = q.z; xs[go + 6] = q.w;
        if (so >= 0) { xs[so] = V.x; xs[so + 1] = V.y; xs[so + 2] = V.z; }
    }
    __syncthreads();
}

template <int NT, bool MFMA, bool CHAIN, bool COOP = false, bool TD = !CHAIN>
// (-DTCV_CHAIN_OCC1, developer build libtcv_hip_occ1.so: the chain kernel compiled for ONE wavefront per SIMD -- 512 registers, no spills -- to
// measure what the 156 spilled registers of the production kernel cost at equal occupancy, profiles/r03_spill_ab.txt)
// (-DTCV_CHAIN_OCC3, libtcv_hip_occ3.so: THREE wavefronts per SIMD -- 168 registers -- for the occupancy experiment of tools/dev_occupancy3.py)
#if defined(TCV_CHAIN_OCC)      // (round 6: any occupancy, -DTCV_CHAIN_OCC=4 -> 128 registers)
#define TCV_CHAIN_WAVES TCV_CHAIN_OCC
#elif defined(TCV_CHAIN_OCC1)
#define TCV_CHAIN_WAVES 1
#elif defined(TCV_CHAIN_OCC3)
#define TCV_CHAIN_WAVES 3
#else
#define TCV_CHAIN_WAVES 2
#endif
__global__ void __launch_bounds__(NT) __attribute__((disable_tail_calls)) __attribute__((amdgpu_waves_per_eu((CHAIN && !COOP) ? TCV_CHAIN_WAVES : 1, COOP ? 1 : (CHAIN ? TCV_CHAIN_WAVES : 8)))) solve_kernel(SolveArgs A) {
    extern __shared__ __attribute__((aligned(16))) double lds_raw[];
    lds_d *lds = (lds_d *)lds_raw;
    int tid = threadIdx.x;
    int slot = blockIdx.x, wstride = gridDim.x;      // scratch slot and first window of this workgroup; window stride
    if (CHAIN && !COOP && NT == 256 && A.role_mode != 0) {
        // Role placement experiment.  The phases give the wavefronts different roles (wave 3: T pipeline, waves 0-1: column owners and the
        // point factors, ...); two workgroups that share a CU and run in step put the same role on the same SIMD.  HW_ID: wave slot
        // [3:0], SIMD [5:4], CU [11:8].
        const unsigned hw = (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        const int simd = (int)(hw >> 4) & 3, wslot = (int)hw & 15, pw = tid >> 6;
        lds_i *ex = (lds_i *)lds;
        if ((tid & 63) == 0) { ex[pw] = simd; ex[4 + pw] = wslot; }
        __syncthreads();
        const int s0 = ex[0], s1 = ex[1], s2 = ex[2], s3 = ex[3], w0 = ex[4];
        __syncthreads();
        const bool perm = ((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15;
        const int mode = A.role_mode & 15;
        int lw = pw;
        if (mode == 1) lw = (pw + 2 * (w0 & 1)) & 3;
        else if (mode == 2) lw = (pw + 2 * (((int)blockIdx.x >> 8) & 1)) & 3;
        else if (mode == 3 && perm) lw = (simd + 2 * (w0 & 1)) & 3;
        else if (mode == 4 && perm) lw = simd;
        else if (mode == 5) lw = (pw + (w0 & 1)) & 3;
        else if (mode == 6 && perm) lw = (simd + (w0 & 3)) & 3;
        if ((A.role_mode & 16) && A.prof && (tid & 63) == 0) {
            gbl_d *pr = (gbl_d *)A.prof + (size_t)blockIdx.x * 32;
            pr[pw * 4 + simd] += 1.0;
            if (pw == 0) { pr[16 + (w0 & 7)] += 1.0; pr[24 + (perm ? 1 : 0)] += 1.0; }
        }
        tid = (tid & 63) | (__builtin_amdgcn_readfirstlane(lw) << 6);      // the wave index stays provably uniform
        // de-phasing experiment (role_mode bit 5, count in bits 8..): the second workgroup of a CU starts late, so that the two do not walk
        // through the same phases (and memory bursts) at the same time
        if ((A.role_mode & 32) && (w0 & 1)) { const int nsl = __builtin_amdgcn_readfirstlane(A.role_mode >> 8); for (int k = 0; k < nsl; k++) __builtin_amdgcn_s_sleep(127); }
    }
    if (COOP) {
        const int G8 = 8 * (1 + A.coop_h);      // workgroups of a block of eight groups: contiguous in dispatch order, one group per XCD
        const int blk8 = (int)blockIdx.x / G8, r8 = (int)blockIdx.x - blk8 * G8;
        const int member = r8 >> 3, g = blk8 * 8 + ((r8 - A.coop_rot) & 7);      // (workgroup b runs on XCD b % 8: the host picks the rotation, tcv_capi.hip coop_admit)
        if (g >= A.coop_groups) return;
        if (member > 0) { coop_helper<NT, TD>(A, lds, g, member - 1); return; }
        slot = g; wstride = A.coop_groups;
    }
    Ctx<NT> C;
    C.tid = tid;
    C.cx_ctl = nullptr; C.cx_x = nullptr; C.cx_exp = nullptr; C.cx_h = 0; C.cx_exp_stride = 0; C.cx_seq = 0; C.cx_timeout = 0;
    if (COOP) {
        C.cx_ctl = (gbl_i *)A.coop_ctl + (size_t)slot * COOP_CTL_INTS;
        C.cx_x = (gbl_d *)A.coop_x + (size_t)slot * COOP_X_DOUBLES;
        C.cx_exp = (gbl_d *)A.coop_exp + (size_t)slot * A.coop_exp_chunks * A.coop_exp_stride;
        C.cx_h = A.coop_h; C.cx_exp_stride = A.coop_exp_stride; C.cx_timeout = A.coop_timeout;
    }
    gbl_d *scr = (gbl_d *)A.scratch + (size_t)slot * A.scratch_stride;
    ctx_scratch_layout<NT>(C, scr);
    C.prof = A.prof ? (gbl_d *)A.prof + (size_t)slot * 32 : nullptr;
    C.t_last = 0;
    C.skip = (unsigned)A.pad2;
    C.g_imublk = nullptr; C.g_spill = nullptr;

    for (int win = slot; win < A.nwin; win += wstride) {
        cst_win *W = (cst_win *)A.win + win;
        cst_plan &P = ((cst_plan *)A.plans)[W->plan];
        C.P = &P; C.W = W;
        C.ip = (cst_i *)A.ipool + A.plan_base[W->plan];
        C.dp = (cst_d *)A.dpool + W->dbase;
        const int nc = P.nc, L = P.nland, nxl = (P.nx + L + 1) & ~1, nl = nc + L;
        C.ntd = CHAIN ? P.nt_c : P.nt;
        C.nd = CHAIN ? P.npp : nc;
        C.ntiles = C.ntd * (C.ntd + 1) / 2;
        const int pp_tiles = P.ntp * (P.ntp + 1) / 2;
        C.tiles = lds;
        lds_d *p;
        if (CHAIN) {      // [pose tiles | pool: staging + area, IMU records, fronts | vectors]
            C.stage = lds + (C.ntiles << 8);
            C.stage_cap = P.c_stage_cap;
            p = C.stage + P.c_pool;
            C.g_imublk = (gbl_d *)A.imublk + (size_t)slot * 16 * IMU_BLK;
            C.g_spill = (gbl_d *)A.spill + (size_t)slot * A.spill_stride;
        } else {
            C.stage = lds + (pp_tiles << 8);
            C.stage_cap = (C.ntiles - pp_tiles) << 8;
            p = lds + (C.ntiles << 8);
        }
        ctx_lds_layout<NT>(C, p, nxl, CHAIN, P.c_stage_cap, TCV_CAMW(P));      // xs, xc, sc, rc | sd, ycam, invdiag = gcam, red, flag, hd, area
#ifdef TCV_PROFILE
        if (tid < PH_COUNT) ((lds_u *)(C.red + 40))[tid] = 0u;
        if (tid == 0) { typedef __attribute__((address_space(3))) long long lds_ll; *(lds_ll *)(C.red + 56) = clock64(); }
        __syncthreads();
#endif
        typedef __attribute__((address_space(1))) DevSummary gbl_sum;
        gbl_sum *S = (gbl_sum *)A.summary + win;
        // C lives in the stack frame (the phase functions take it by reference); the kernel's own accesses go through a copy whose
        // address never escapes, so that they stay in registers (SGPRs: every field is uniform) across the calls
        const Ctx<NT> K = uniform_ctx<NT>(C);

        for (int i = tid; i < P.nx + L; i += NT) K.xs[i] = K.dp[W->d_x + i];
        // sqrt_info = LLT(cov^-1).matrixL()^T once per solve (imu_factor.h:64 recomputes it per Evaluate)
        if (W->d_sqrt >= 0) {
            for (int i = tid; i < P.n_imu * 225; i += NT) C.g_sqrt[i] = K.dp[W->d_sqrt + i];
        } else {
            // one 16-lane group per factor, workspace in the (still unused) tile region; in a function of its own (registers of its own: inlined
            // into the kernel its loops reloaded spilled values from scratch memory in every step)
            sqrt_info_all<NT>(K.dp + W->d_imu, P.n_imu, C.g_sqrt, lds, tid);
        }
        __syncthreads();
        if (A.sqrt_out && W->d_sqrt < 0 && W->sqrt_export >= 0 && tid < 225) A.sqrt_out[(size_t)win * 225 + tid] = C.g_sqrt[W->sqrt_export * 225 + tid];
        // constant part of the prior: Hp = J0' J0 (packed lower), marginalization_factor.cpp:366,371-380.  The tile
        // region is still unused: J0 is staged there (columns contiguous) and every thread forms 1 x 2 entry pairs.
        if (P.prior_n > 0 && ABL(C, AB_SETUP)) {
            const int n = P.prior_n, nr = n - W->prior_k0;      // J0 without its leading zero rows: nr x n, column-major (tcv_packed.h)
            cst_d *J0g = C.dp + W->d_prior;
            const bool in_lds = nr * n <= (C.ntiles << 8) + (CHAIN ? P.c_pool : 0);
            if (in_lds) copy_doubles<NT>(lds, J0g, nr * n, tid);
            __syncthreads();
            for (int e = tid; e < n * (n + 1) / 2; e += NT) {      // packed index e = a (a + 1) / 2 + b, b <= a: every lane has an entry
                int a = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
                while ((a + 1) * (a + 2) / 2 <= e) a++;
                while (a * (a + 1) / 2 > e) a--;
                const int b = e - a * (a + 1) / 2;
                double s0 = 0, s1 = 0;
                if (in_lds) {
                    // (two accumulators by row parity, rows ascending: dropping the zero rows leaves each chain's non-zero terms in their
                    // order -- an odd number of dropped rows only swaps the names of the two chains -- so s0 + s1 keeps its bits)
                    const lds_d *ca = lds + nr * a, *cb = lds + nr * b;
                    int i = 0;
                    for (; i + 7 < nr; i += 8) {      // eight rows' loads in flight, the two accumulators updated in the original order
                        double a8[8], b8[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) { a8[u] = ca[i + u]; b8[u] = cb[i + u]; }
#pragma unroll
                        for (int u = 0; u < 8; u += 2) { s0 += a8[u] * b8[u]; s1 += a8[u + 1] * b8[u + 1]; }
                    }
                    for (; i + 1 < nr; i += 2) { s0 += ca[i] * cb[i]; s1 += ca[i + 1] * cb[i + 1]; }
                    if (nr & 1) s0 += ca[nr - 1] * cb[nr - 1];
                } else {
                    for (int i = 0; i < nr; i++) s0 += J0g[i + nr * a] * J0g[i + nr * b];
                }
                C.g_hp[a * (a + 1) / 2 + b] = s0 + s1;
            }
        }
        __syncthreads();
        TCV_MARK(C, PH_SETUP);

        // Solver::Options::max_num_iterations (estimator.cpp:1890; sensor.yaml ships up to 100): honoured as given.  The per-iteration
        // trace of DevSummary holds the first MAX_TRACE entries; num_iterations keeps counting beyond it.
        const int max_it = A.max_iterations;
        // Solver::Options::max_solver_time_in_seconds (estimator.cpp:1892-1897): Ceres checks the wall clock at the start of every
        // iteration and stops with NO_CONVERGENCE; here the budget runs on the device's constant-rate clock from the moment the
        // workgroup picks the window up.  Like in the reference this makes the iteration count timing dependent; 0 = no limit.
        const long long t_window = (A.max_ticks > 0) ? (long long)wall_clock64() : 0;
        const bool fixed = A.fixed_iterations != 0;
        double radius = 1e4, mu = 1e-8, lin_mu = 1e-8;
        bool reuse = false, tiles_valid = true;
        int invalid = 0, termination = 0, nrec = 1, status = 0;
#define TCV_COOP_LIN(X, FIRST, ASM, MU) coop_lin(linearize_coop<NT>(TCV_CTX_ARGS(K), C.cx_ctl, C.cx_x, C.cx_exp, C.cx_h, C.cx_exp_stride, C.cx_seq, C.cx_timeout, X, FIRST, ASM, MU, win))
        auto coop_lin = [&](const CoopLin &r) -> double { C.cx_seq = __builtin_amdgcn_readfirstlane(r.seq); return r.cost; };      // the master's sequence number lives in the kernel
        double cost = uni_d(COOP ? TCV_COOP_LIN(K.xs, true, true, mu) : linearize<NT, CHAIN, TD>(TCV_CTX_ARGS(K), K.xs, true, true, mu));
        bool first = true;
        const double initial_cost = cost;
        if (tid == 0) { S->cost[0] = cost; S->step_ok[0] = 1; S->dogleg_case[0] = 0; S->radius[0] = radius; S->mu[0] = mu; }
        double gg = 0, q = 0, alpha = 0, yg = 0, gdy = 0, dy2 = 0;
        // |x| and |x+ - x| in ambient space feed the parameter-tolerance test only: with the convergence tests off (fixed iterations) nothing
        // reads them, and the three reductions per iteration are not run
        double xn2 = 0.0, dn2 = 0.0;
        if (!fixed) { const Norms2 nn = ambient_norms<NT>(TCV_CTX_ARGS(K), K.xs, K.xs); xn2 = uni_d(nn.xn2); dn2 = uni_d(nn.dn2); }
        double x_norm = uni_d(sqrt(xn2));
        bool done = false;
        if (COOP && C.cx_seq < 0) { done = true; status = -9; termination = 5; }      // the helpers did not answer (timeout)
        if (!fixed && !done) {
            const double gm = grad_max<NT>(TCV_CTX_ARGS(K));
            if (gm <= 1e-10) { termination = 1; done = true; }
        }
        int it = 0;
        while (!done) {
            if (it >= max_it) break;
            if (A.max_ticks > 0) {      // uniform decision: lane 0 of wave 0 reads the clock, everybody follows
                if (tid == 0) *K.flag = ((long long)wall_clock64() - t_window > A.max_ticks) ? 7 : 0;
                __syncthreads();
                const int over = *K.flag;
                __syncthreads();
                if (tid == 0) *K.flag = 0;
                if (over) break;
            }
            it++;
            bool ls_ok = true;
            if (!reuse) {
                reuse = true;
                ls_ok = false;
                while (mu < 1.0) {
                    if (!tiles_valid || lin_mu != mu) {
                        if (COOP) (void)TCV_COOP_LIN(K.xs, false, true, mu); else (void)linearize<NT, CHAIN, TD>(TCV_CTX_ARGS(K), K.xs, false, true, mu);
                        if (COOP && C.cx_seq < 0) break;
                        lin_mu = mu;
                    }
                    const FinOut fo = finalize_and_solve<NT, MFMA, CHAIN>(TCV_CTX_ARGS(K), first, mu);
                    const bool ok = fo.ok;
                    gg = uni_d(fo.gg); q = uni_d(fo.q);
                    first = false;
                    tiles_valid = false;
                    if (ok) { ls_ok = true; break; }
                    mu = uni_d(mu * 10.0);
                }
                if (COOP && C.cx_seq < 0) { status = -9; termination = 5; break; }
                if (ls_ok && ABL(C, AB_DOGLEG)) {
                    alpha = uni_d(gg / q);
                    // dot products for the dogleg interpolation and the model decrease
                    double acc[3] = {0.0, 0.0, 0.0};
                    for (int i = tid; i < nl; i += NT) {
                        // (the scratch vectors by their fixed offsets behind v_s: one base pointer live across the calls instead of five)
                        const double y = (K.v_s + 4 * SCR_NL)[i], D = (K.v_s + 2 * SCR_NL)[i], gh = (K.v_s + 3 * SCR_NL)[i];
                        acc[0] += y * (gh * D);   // y' g_s
                        acc[1] += gh * D * y;     // ghat . (D y) = -ghat . gn
                        acc[2] += (D * y) * (D * y);
                    }
                    block_sum_n<NT, 3>(acc, K.red, tid);
                    yg = uni_d(acc[0]); gdy = uni_d(acc[1]); dy2 = uni_d(acc[2]);
                }
            }
            TCV_MARK(C, PH_OTHER);
            double model_cost_change = 0, step_norm = 0, ca = 0, cb = 0;
            int dcase = 0;
            bool step_valid = false;
            if (ls_ok) {
                // traditional dogleg in D-space: gn = -D y, Cauchy = -alpha ghat
                const double gnorm = sqrt(gg), gn_norm = sqrt(dy2);
                if (gn_norm <= radius) { ca = 0.0; cb = -1.0; step_norm = gn_norm; dcase = 1; }
                else if (gnorm * alpha >= radius) { ca = -(radius / gnorm); cb = 0.0; step_norm = radius; dcase = 2; }
                else {
                    const double b_dot_a = alpha * gdy;          // (-alpha ghat) . (-D y)
                    const double a_sq = (alpha * gnorm) * (alpha * gnorm);
                    const double bma_sq = a_sq - 2.0 * b_dot_a + gn_norm * gn_norm;
                    const double c = b_dot_a - a_sq;
                    const double d = sqrt(c * c + bma_sq * (radius * radius - a_sq));
                    const double beta = (c <= 0) ? (d - c) / bma_sq : (radius * radius - a_sq) / (d + c);
                    ca = -alpha * (1.0 - beta); cb = -beta; dcase = 3;
                    // |ca ghat - cb' gn|: step = ca*ghat + beta*gn with gn = -D y
                    step_norm = sqrt(ca * ca * gg + 2.0 * ca * cb * gdy + cb * cb * dy2);
                }
                // step (scaled-J space) p = ca * (ghat / D) + cb * y ; model decrease via H_s y = g_s - mu D^2 y
                const double pg = ca * gg + cb * yg;
                const double vHy = gg - mu * gdy, yHy = yg - mu * dy2;
                const double pHp = ca * ca * q + 2.0 * ca * cb * vHy + cb * cb * yHy;
                model_cost_change = uni_d(-pg - 0.5 * pHp);
                ca = uni_d(ca); cb = uni_d(cb); step_norm = uni_d(step_norm);
                // sign convention of the minimiser: step = -(...) so that model_cost_change > 0 for descent
                step_valid = model_cost_change > 0.0;
            }
#ifdef TCV_ABLATE
            if (ABL_ACCEPT(C)) step_valid = true;
#endif
            if (!step_valid) {
                invalid++;
                if (tid == 0 && nrec < MAX_TRACE) {
                    S->cost[nrec] = cost; S->step_ok[nrec] = 0; S->dogleg_case[nrec] = -1; S->mu[nrec] = mu;
                    S->radius[nrec] = radius; S->rho[nrec] = 0; S->model_cost_change[nrec] = model_cost_change;
                    S->cost_candidate[nrec] = 0; S->step_norm[nrec] = 0;
                }
                nrec++;
                if (invalid >= 5) { termination = 5; break; }
                mu = uni_d(mu * 10.0);
                reuse = false;
                continue;
            }
            invalid = 0;
            TCV_MARK(C, PH_DOGLEG);
            for (int i = tid; i < nl; i += NT) (K.v_s + 5 * SCR_NL)[i] = ca * ((K.v_s + 3 * SCR_NL)[i] / (K.v_s + 2 * SCR_NL)[i]) + cb * (K.v_s + 4 * SCR_NL)[i];
            __syncthreads();
            if (ABL(C, AB_PLUS)) apply_plus<NT>(TCV_CTX_ARGS(K), K.xs, K.v_s + 5 * SCR_NL, K.v_s, K.xc);
            if (A.first_delta && it == 1)
                for (int i = tid; i < nl; i += NT) A.first_delta[(size_t)win * A.delta_stride + i] = (K.v_s + 5 * SCR_NL)[i] * K.v_s[i];
            __syncthreads();
            TCV_MARK(C, PH_PLUS);
            const double mu_next = uni_d(fmax(1e-8, 2.0 * mu / 10.0));
            const bool want_asm = (it < max_it) || !fixed;
            const double cost_c = uni_d(COOP ? TCV_COOP_LIN(K.xc, false, want_asm, mu_next) : linearize<NT, CHAIN, TD>(TCV_CTX_ARGS(K), K.xc, false, want_asm, mu_next));
            if (COOP && C.cx_seq < 0) { status = -9; termination = 5; break; }
            tiles_valid = want_asm;
            lin_mu = mu_next;
            if (!fixed && ABL(C, AB_NORMS)) { const Norms2 nn = ambient_norms<NT>(TCV_CTX_ARGS(K), K.xs, K.xc); xn2 = uni_d(nn.xn2); dn2 = uni_d(nn.dn2); }
            TCV_MARK(C, PH_NORMS);
#ifdef TCV_ABLATE
            const double rho = ABL_ACCEPT(C) ? 1.0 : (cost - cost_c) / model_cost_change;
#else
            const double rho = (cost - cost_c) / model_cost_change;
#endif
            if (tid == 0 && nrec < MAX_TRACE) {
                S->model_cost_change[nrec] = model_cost_change; S->cost_candidate[nrec] = cost_c;
                S->radius[nrec] = radius; S->mu[nrec] = mu; S->rho[nrec] = rho; S->step_norm[nrec] = step_norm;
                S->dogleg_case[nrec] = dcase;
            }
            if (!fixed && sqrt(dn2) <= 1e-8 * (x_norm + 1e-8)) {
                if (tid == 0 && nrec < MAX_TRACE) { S->cost[nrec] = cost; S->step_ok[nrec] = 0; }
                nrec++; termination = 2; break;
            }
            const double cost_change = cost - cost_c;
            if (!fixed && fabs(cost_change) <= 1e-6 * cost) {
                if (tid == 0 && nrec < MAX_TRACE) { S->cost[nrec] = cost; S->step_ok[nrec] = 0; }
                nrec++; termination = 3; break;
            }
            if (rho > 1e-3) {
                for (int i = tid; i < P.nx + L; i += NT) K.xs[i] = K.xc[i];
                if (!fixed && ABL(C, AB_NORMS)) { const Norms2 nn = ambient_norms<NT>(TCV_CTX_ARGS(K), K.xc, K.xc); xn2 = uni_d(nn.xn2); dn2 = uni_d(nn.dn2); }
                x_norm = uni_d(sqrt(xn2));
                cost = cost_c;
                if (rho < 0.25) radius = uni_d(radius * 0.5);
                if (rho > 0.75) radius = uni_d(fmax(radius, 3.0 * step_norm));
                mu = mu_next;
                reuse = false;
                if (tid == 0 && nrec < MAX_TRACE) { S->cost[nrec] = cost; S->step_ok[nrec] = 1; }
                nrec++;
                if (!fixed) {
                    const double gm = grad_max<NT>(TCV_CTX_ARGS(K));
                    if (gm <= 1e-10) { termination = 1; break; }
                }
            } else {
                radius = uni_d(radius * 0.5);
                reuse = true;
                tiles_valid = false;
                if (tid == 0 && nrec < MAX_TRACE) { S->cost[nrec] = cost; S->step_ok[nrec] = 0; }
                nrec++;
            }
            if (radius < 1e-32) { termination = 4; break; }
        }
        __syncthreads();
        if (A.gauge_fix) gauge_epilogue<NT>(K.xs, K.dp + W->d_x, K.ip + P.o_frames, P.n_frames, tid);
        for (int i = tid; i < P.nx + L; i += NT) A.state_out[(size_t)win * A.state_stride + i] = K.xs[i];
        if (tid == 0) {
            S->num_iterations = nrec;
            S->termination = termination;
            S->status = status;
            S->initial_cost = initial_cost;
            S->final_cost = cost;
        }
        __syncthreads();
        TCV_MARK(C, PH_OTHER);
#ifdef TCV_PROFILE
        if (tid == 0 && C.prof)
            for (int i = 0; i < PH_COUNT; i++) { lds_u *lp = (lds_u *)(K.red + 40); C.prof[i] += (double)lp[i]; lp[i] = 0u; }
        __syncthreads();
#endif
    }
    if (COOP && C.cx_seq >= 0) {      // the helpers of this group leave with the master
        __syncthreads();
        coop_publish<NT>(C, nullptr, 0, 0.0, COOP_CMD_EXIT, 0, C.cx_seq + 1);
    }
}

}  // namespace tcv

// The chain kernel is built in its own translation unit (-DTCV_SOLVE_CHAIN_TU, every device function inlined so that the
// 2-waves-per-SIMD register budget covers the whole call tree: the occupancy attribute does not reach non-inlined callees).
#ifdef TCV_SOLVE_COOP_TU
// cooperative chain kernel (own translation unit, like the chain kernel): grid = (1 + coop_h) x (groups rounded up to a multiple of 8)
extern "C" int tcv_launch_solve_coop(const tcv::SolveArgs *args, int grid, size_t lds_bytes, void *stream) {
    using namespace tcv;
    hipStream_t st = (hipStream_t)stream;
    // (one instance for plain and ESTIMATE_TD windows: the helpers' factor evaluation is the only place that tells them apart)
    hipError_t e = hipFuncSetAttribute((const void *)solve_kernel<256, true, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((solve_kernel<256, true, true, true, true>), dim3(grid), dim3(256), lds_bytes, st, *args);
    return (int)hipGetLastError();
}
#elif defined(TCV_SOLVE_CHAIN_TD_TU)
// chain kernel with ProjectionTdFactor (ESTIMATE_TD windows): its own instantiation and translation unit, so that the factor's code and
// registers stay out of the kernel every shipped configuration runs
extern "C" int tcv_launch_solve_chain_td(const tcv::SolveArgs *args, int grid, size_t lds_bytes, void *stream) {
    using namespace tcv;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipFuncSetAttribute((const void *)solve_kernel<256, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((solve_kernel<256, true, true, false, true>), dim3(grid), dim3(256), lds_bytes, st, *args);
    return (int)hipGetLastError();
}
#elif defined(TCV_SOLVE_CHAIN_TU)
extern "C" int tcv_launch_solve_chain(const tcv::SolveArgs *args, int grid, size_t lds_bytes, void *stream) {
    using namespace tcv;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipFuncSetAttribute((const void *)solve_kernel<256, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((solve_kernel<256, true, true>), dim3(grid), dim3(256), lds_bytes, st, *args);
    return (int)hipGetLastError();
}
#else
extern "C" int tcv_launch_solve_chain(const tcv::SolveArgs *args, int grid, size_t lds_bytes, void *stream);
extern "C" int tcv_launch_solve_coop(const tcv::SolveArgs *args, int grid, size_t lds_bytes, void *stream);
extern "C" int tcv_launch_solve_chain_td(const tcv::SolveArgs *args, int grid, size_t lds_bytes, void *stream);
extern "C" int tcv_launch_solve(const tcv::SolveArgs *args, int grid, int nthreads, size_t lds_bytes, void *stream) {
    using namespace tcv;
    if (args->chain && args->coop_h > 0) return tcv_launch_solve_coop(args, grid, lds_bytes, stream);
    if (args->chain && args->chain_td) return tcv_launch_solve_chain_td(args, grid, lds_bytes, stream);
    if (args->chain) return tcv_launch_solve_chain(args, grid, lds_bytes, stream);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e;
#define TCV_LAUNCH(NTV, MF)                                                                                                \
    do {                                                                                                                   \
        e = hipFuncSetAttribute((const void *)solve_kernel<NTV, MF, false>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                                (int)lds_bytes);                                                                           \
        if (e != hipSuccess) return (int)e;                                                                                \
        hipLaunchKernelGGL((solve_kernel<NTV, MF, false>), dim3(grid), dim3(NTV), lds_bytes, st, *args);                   \
    } while (0)
    if (nthreads == 512) { if (args->use_mfma) TCV_LAUNCH(512, true); else TCV_LAUNCH(512, false); }
    else { if (args->use_mfma) TCV_LAUNCH(256, true); else TCV_LAUNCH(256, false); }
#undef TCV_LAUNCH
    return (int)hipGetLastError();
}

extern "C" int tcv_solve_scratch_doubles(void) { return tcv::SCR_TOTAL; }
#endif
