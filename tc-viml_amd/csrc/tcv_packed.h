// Device-resident layout of one packed sliding window ("what OptimizationWithLine hands to the
// solver", reference vins_estimator/src/estimator.cpp:1677-1900) and of the static assembly plan
// that goes with its graph structure.
//
// A window is split into
//   * DATA  (doubles, one copy per window):  initial states, IMU pre-integration constants,
//     point / line observations, the marginalisation prior.  These are the "algorithmic bytes".
//   * PLAN  (ints, one copy per distinct graph STRUCTURE, shared by every window with the same
//     structure):  block tables and the destination-driven gather lists that turn per-factor
//     Jacobian blocks into the block-sparse normal equations without atomics, in a fixed order.
//
// Unknown ordering inside the fused solver (tangent space):
//   [ pose-kind camera blocks (6 each: the 11 poses + the extrinsic) | Euclidean camera blocks
//   (speed-bias 9 each) ] followed, outside the dense system, by the landmarks (inverse depths,
//   eliminated by the Schur complement: they only ever meet {pose_i, pose_j, ex}, estimator.cpp:1767).
#pragma once

namespace tcv {

enum { KIND_EUCLID = 0, KIND_POSE = 1 };
enum { CAM_W = 176, CAM_MAX = 184 };      // camera tangent dims the fused solver's vectors hold: every window of OptimizationWithLine / with the relocalisation pose
enum { TILE = 16, TILE_ELEMS = 256 };
enum { LDS_DOUBLES = 20480 };  // 160 KiB per workgroup on gfx950
int chain_lds_doubles();       // LDS doubles of a chain-mode workgroup (two per CU); tcv_pack.cpp
// Host threads a batch-level operation may start: min(want, cores the process is GRANTED / batch-level operations running right now).
// The grant is the cgroup CPU quota (cpu.max) or the affinity mask, not the machine's thread count: four callers packing 512 windows each on
// sixteen threads under a 16-core quota used to run 64 threads into the scheduler's throttling.  tcv_pack.cpp
struct HostOp { HostOp(); ~HostOp(); int threads(int want) const; };
int host_threads(int want);      // the same share for code that runs inside somebody's HostOp (does not count as an operation of its own)
}  // namespace tcv
#include <functional>
namespace tcv {
// fn(t) for t in [0, nth): index claiming by the calling thread and by persistent worker threads (created once, tcv_pack.cpp); returns
// when all have finished.  nth <= 1: plain call.
void parallel_run(int nth, const std::function<void(int)> &fn);
// items 0 .. n - 1 claimed ONE AT A TIME by up to nth threads (the caller among them): fn(item, slot), slot < nth unique per thread.  A
// worker that wakes up late finds fewer items instead of a fixed share nobody else may touch (a strided split of 64 windows over 8 threads
// waited a whole share -- 0.36 ms -- for the last two workers).
void parallel_items(int n, int nth, const std::function<void(int, int)> &fn);
void async_run(std::function<void()> fn);      // fn() on a worker thread, some time later; nobody waits (here and now if the process has no worker)
bool prior_keep_zero_rows();   // developer A/B switch TCV_PRIOR_FULL (re-read by every tcv_batch_create / tcv_solve); tcv_pack.cpp
void prior_refresh_switch();
enum { MAX_TRACE = 64 };

// staging record strides (doubles per residual row): Jacobian columns followed by the residual
enum {
    PROJ_STRIDE = 20,  // [pose_i 6 | pose_j 6 | ex 6 | inv depth 1 | r]
    PROJ_REC = 41,     // 2 rows x 20 + 1 pad: an odd record stride spreads same-column accesses of different factors over all LDS banks
    PROJ_TD_STRIDE = 26, PROJ_TD_REC = 53,   // ProjectionTdFactor: [.. | r | td | 0 0 0 0 0]
    LINE_STRIDE = 7,   // [pose 6 | r]
    LINE_REC = 14,
    IMU_STRIDE_J = 31, // [pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9 | r]
    IMU_REC = 465,
    IMU_CONST = 287    // doubles of pre-integration constants per IMU factor
};

// 16 x 16 tiles of the reduced camera system in LDS: lower-triangular tile order, XOR-swizzled columns (conflict-free row AND column
// walks); shared by the kernels and by the packer (which precomputes scatter destinations)
__host__ __device__ inline int sw(int r, int c) { return (r << 4) + (c ^ (r & 14)); }
__host__ __device__ inline int tbase(int I, int J) { return ((I * (I + 1) / 2) + J) << 8; }
__host__ __device__ inline int tix(int a, int b) { return tbase(a >> 4, b >> 4) + sw(a & 15, b & 15); }
// IMU scatter table (PlanHdr::o_iitem, n_imu x 1024 ints, [factor][lane][16]): per lane and accumulator register q = 4 tile + i of the
// factor's four 16 x 16 J'J tiles the destination of the value: low 16 bits = d + IMU_SC_BIAS with d >= 0 a tile element, d = -1 nothing,
// d <= -2 the gradient entry -2 - d, d <= -1000 the diagonal store of a Euclidean block (chain layout); bit 16: the value is also
// parked in the per-factor J'J block in HBM (chain layout)
enum { IMU_SC_BIAS = 4096, IMU_SC_STORE = 1 << 16 };

// gather destination kinds
enum { DK_TILE = 0, DK_G = 1, DK_HCL = 2, DK_HLL = 3, DK_GL = 4, DK_RC = 5 };

// The gathers are destination driven: the host turns the graph into ROW UNITS, one per row of a destination block,
//   acc[e] += sum_rows rec[row][colA + ea] * rec[row][colB + e]   (e < ncols <= 6, IMU: <= 9)   over the unit's items,
// so every staged Jacobian value is read once per row and the summation order is fixed (bit-wise deterministic).
//   unit : u0 = kind << 28 | ncols << 24 | ea << 20 | nitems ; u1 = o0 << 16 | o1 ; u2 = item_begin  (IMU: u2, u3 = items)
//     DK_TILE o0/o1 = first tangent row/col of the block pair (row >= col); DK_G o0 = first tangent index;
//     DK_HCL  o0 = offset in the landmark/camera coupling store; DK_HLL o0 = landmark (acc[0] = hll, acc[1] = gl);
//     DK_RC   o0 = first tangent index (Schur program: rhs correction)
//   visual item: rec_base << 11 | colA << 6 | colB << 1 | type   (type 0: point record stride 20, 1: line stride 7)
//   imu item   : fac_local << 10 | colA << 5 | colB
//   schur item : landmark << 16 | offA << 8 | offB   (offsets in the landmark's Hcl slice; offA = 255: gl instead)
inline unsigned pack_unit(int dest, int ea, int eb) { return ((unsigned)dest << 8) | ((unsigned)ea << 4) | (unsigned)eb; }

struct PlanHdr {
    // sizes
    int nblk, nland;    // camera blocks (constant ones included), landmarks
    int nc, nx;         // camera tangent dim (variable blocks), camera ambient dim
    int npp;            // tangent dims of the pose-kind blocks (they come first)
    int nt, ntp;        // 16x16 tile rows of the augmented system (nc + 1 rhs row), tile rows covering npp
    int n_imu, n_proj, n_line;
    int prior_n, prior_nblk, prior_xsize;
    int hcl_total;      // doubles of the landmark/camera coupling store
    int n_imu_chunk;    // IMU factors are staged through LDS in chunks
    int n_vis_chunk;    // point/line factors likewise (1 for the BASELINE configs)
    int lds_area;       // doubles of the time-shared LDS area
    int flags;          // bit 0: the point factors are ProjectionTdFactors (chain layout: the kernel instance with TD = true)
    int td_cam;         // camera block index of para_Td (-1 none); point records then carry 26 columns per row:
                        // [.. 19 as below | r | td | 5 zeros] so that Td rides through the 6-wide gather machinery
    int camw;           // width of the LDS vectors over the camera tangent space (sc, ycam, invdiag / gcam): CAM_W for nc <= CAM_W, else CAM_MAX
                        // (a 12th pose block -- the relocalisation pose, estimator.cpp:1854-1886 -- takes the camera side to 177 dims)
    // int-pool offsets (relative to the plan base)
    int o_blk;      // nblk x 4 : gsize, goff (ambient), loff (tangent, -1 constant), kind
    int o_imu;      // n_imu x 4 block ids
    int o_proj;     // n_proj x 4 : blk_i, blk_j, blk_ex, landmark
    int o_line;     // n_line
    int o_prior;    // prior_nblk x 4 : blk id, idx (first J0 column), gsize, x0 offset
    int o_pcol;     // prior_n : tangent index of each J0 column (-1 constant)
    int o_pdest;    // prior_n (prior_n + 1) / 2: where entry e = a (a + 1) / 2 + b of the packed Hp = J0'J0 goes: >= 0 tile element, -1 nowhere,
                    // <= -2: diagonal store -2 - d of a Euclidean block (chain layout, whose off-diagonal Euclidean entries are read from Hp directly)
    int o_lm;       // nland x 2 : e_off (offset of the landmark's slice in the Hcl store), nslot
    int o_lmslot;   // sum nslot : tangent offset of every slot's block (ordered by landmark)
    int o_lmslotptr;// nland + 1
    int o_vchunk;   // n_vis_chunk x 16: proj_begin, proj_count, line_begin, line_count, vprog offset (rel. o_vdest), units,
                    //   wave units, items, lm_begin, lm_count, hcl_begin, hcl_size, sprog offset (rel. o_sdest), units, wave units, items
    int o_vdest, o_vunit, o_vitem;   // o_vdest: visual gather programs of all chunks (units then items per chunk)
    int n_vdest, n_vunit, n_vitem;
    int o_sdest, o_sunit, o_sitem;   // Schur plan
    int n_sdest, n_sunit, n_sitem;
    int o_ichunk;   // n_imu_chunk x 4 : fac_begin, fac_count, number of colours, 0
    int o_idest, o_iunit, o_iitem;   // o_idest: n_imu x 32 tangent index of each local column (-1 constant; host-side source of the scatter table); o_iunit: n_imu colours; o_iitem: scatter table
    int n_idest, n_iunit, n_iitem;
    // chain mode (tcv_solve.hip, CHAIN = true): the Euclidean camera blocks (speed-biases) are eliminated one by one in a
    // fixed order BEFORE the dense pose system; only the pose part (npp + 1 rhs row) lives in LDS tiles
    int chain;          // 1: this plan was laid out for the chain kernel
    int n_e;            // number of chain steps (= free Euclidean camera blocks, all 9 wide)
    int nt_c;           // tile rows of the pose system (npp + 1 rhs row)
    int c_stage_cap, c_area_cap;   // split of the LDS pool in the visual phase (doubles)
    int c_pool;         // doubles of the LDS pool (staging | area | IMU records | fronts)
    int c_spill;        // doubles of the L-column spill per workgroup
    int o_chain;        // n_e x CH_STRIDE ints (16-byte aligned)
    int n_frames, o_frames;   // n_frames x 2 : ambient offset of para_Pose[i], para_SpeedBias[i] (-1: not in the problem)
    int plan_ints;  // total ints of this plan (header excluded)
};

struct WinHdr {
    int plan;           // index of the plan
    int sqrt_export;    // >= 0: the solve writes this IMU factor's sqrt_info to SolveArgs::sqrt_out (the factor the batch's marginalisation needs)
    long long dbase;    // element offset of this window's doubles in the batch data pool
    // double-pool offsets (relative to dbase)
    int d_x;        // nx + nland initial state (camera blocks in block order, then landmarks)
    int d_imu;      // n_imu x 287
    int d_proj;     // n_proj x 6 (ProjectionTdFactor: x 14 = pts_i, pts_j, aux 8)
    int d_line;     // n_line x 9
    int d_linec;    // 21 : K, Ric, Tic (row-major)
    int d_prior;    // J0 rows prior_k0 .. n-1 ((n - prior_k0) x n column-major), r0 (n - prior_k0), x0 (prior_xsize)
    int d_misc;     // G(3), proj sqrt_info, proj loss a, line loss a, TR, ROW, line Jacobian mode (0 reference, 1 exact)
    int d_sqrt;     // optional host-provided sqrt_info, n_imu x 225 (-1: computed on device)
    int n_doubles;  // doubles of this window
    int prior_k0;   // leading rows of the prior's J0 | r0 that are exact zeros -- the eigenvalues of A' the marginalisation thresholded
                    // (marginalization_factor.cpp:284-293; about half of the 75 on the benchmark windows): neither stored, fetched nor multiplied
};
// number of leading rows i of the prior with J0[i][:] == 0 and r0[i] == 0 (J0 n x n column-major)
inline int prior_zero_rows(const double *J0, const double *r0, int n) {
    int k0 = 0;
    for (; k0 < n; k0++) {
        if (r0[k0] != 0.0) break;
        bool z = true;
        for (int j = 0; j < n && z; j++) z = J0[k0 + (size_t)n * j] == 0.0;
        if (!z) break;
    }
    return k0 < n ? k0 : (n > 0 ? n - 1 : 0);      // keep one row: the kernels never see an empty prior
}

// per-workgroup global scratch layout (doubles)
enum { SCR_NL = 1280 };  // capacity of an nl-sized vector (nc + nland)

// per-window result block written by the solver
struct DevSummary {
    int num_iterations, termination;
    int status, pad;
    double initial_cost, final_cost;
    double cost[MAX_TRACE], cost_candidate[MAX_TRACE], model_cost_change[MAX_TRACE];
    double radius[MAX_TRACE], mu[MAX_TRACE], rho[MAX_TRACE], step_norm[MAX_TRACE];
    int step_ok[MAX_TRACE], dogleg_case[MAX_TRACE];
};

struct SolveArgs {
    const WinHdr *win;
    const PlanHdr *plans;
    const long long *plan_base;   // element offset of every plan in the int pool
    const int *ipool;
    const double *dpool;
    double *state_out;            // per window: nx + nland (stride state_stride)
    DevSummary *summary;
    double *first_delta;          // optional, per window: nc + nland tangent step of iteration 1 (stride delta_stride)
    double *scratch;              // per workgroup
    double *prof;                 // TCV_PROFILE builds: 32 cycle accumulators (lane 0 of every workgroup adds), else unused
    int nwin, state_stride, delta_stride, scratch_stride;
    int max_iterations, fixed_iterations, use_mfma, chain;
    double *imublk;               // chain mode, per workgroup: 16 x IMU_BLK doubles (per-factor J'J | J'r blocks, 32 x 32 row-major)
    double *spill;                // chain mode, per workgroup: factored fronts (spill_stride doubles)
    int spill_stride, pad2;       // pad2: phase skip mask of the -DTCV_ABLATE developer build (0 otherwise)
    long long max_ticks;          // max_solver_time_in_seconds in ticks of the constant-rate device clock (wall_clock64); 0: no limit
    double *sqrt_out;             // optional, per window: 225 doubles -- the sqrt_info of IMU factor WinHdr::sqrt_export as computed by this solve
                                  // (the marginalisation of the same batch reads it instead of factorising the covariance again)
    // cooperative (small-batch) mode, see COOP_* below: 0 helpers = off
    int coop_h, coop_groups;      // helper workgroups per window group, number of groups (persistent: group g owns windows g, g + groups, ...)
    int coop_exp_chunks, coop_exp_stride;   // per group: chunk slots of the export area and doubles per slot
    int *coop_ctl;                // per group COOP_CTL_INTS ints
    double *coop_x;               // per group COOP_X_DOUBLES: the state the master hands to its helpers (+ mu)
    double *coop_exp;             // per group coop_exp_chunks x coop_exp_stride doubles
    long long coop_timeout;       // ticks of the constant-rate device clock a workgroup waits for its partners before it gives up (status -9)
    int role_mode, chain_td;          // chain kernel: placement of the wavefront roles on the SIMDs (tcv_solve.hip, solve_kernel), developer switch TCV_ROLE_MODE
    int coop_rot, gauge_fix;          // cooperative mode: group g runs on the workgroups b with b % 8 == (g + coop_rot) % 8, i.e. on XCD (g + coop_rot) % 8;
                                      // gauge_fix: double2vector() in the kernel's epilogue (tcv_batch_set_fused_gauge_fix)
};

// ---- cooperative mode (tcv_solve.hip, solve_kernel<.., COOP = true>) ---------------------------------------------------------------
// A batch that leaves most of the chip idle (a per-sequence replay, a single estimator: BASELINE configs[3] / [4]) gives every window a
// GROUP of 1 + H workgroups on 1 + H CUs.  The MASTER runs the trust-region loop, the chain elimination, the Cholesky factorisation
// and the dogleg exactly as the single-workgroup kernel does; per linearisation it publishes the state, and HELPER h evaluates the
// point / line factors of visual chunks h, h + H, ... , gathers their J'J / J'r, eliminates the chunk's landmarks and exports, per
// chunk, [gathered pose tiles | minus the Schur update of the pose tiles | gradient | rhs / diagonal corrections | per-thread costs]
// to HBM / L2, while the master evaluates the prior and the IMU factors.  The master then folds the chunk exports into its tiles in
// chunk order -- the additions the single-workgroup kernel performs, in its order: bit-identical results for the same chunking --
// scatters the IMU blocks and goes on.  Hand-offs are release / acquire flags at agent scope; every wait is bounded by a timeout.
enum {
    COOP_MAX_H = 7,
    COOP_CTL_INTS = 64,           // [0] sequence number of the master's request, [1] command, [2] window, [3] abort; [8 + h] sequence number helper h has served
    COOP_CTL_SEQ = 0, COOP_CTL_CMD = 1, COOP_CTL_WIN = 2, COOP_CTL_ABORT = 3, COOP_CTL_DONE = 8,
    COOP_CMD_FIRST = 1, COOP_CMD_ASSEMBLE = 2, COOP_CMD_EXIT = 4,
    COOP_X_DOUBLES = SCR_NL + 8,  // x (nx + nland) | mu at SCR_NL
    COOP_EXP_VEC = 176 + 176 + 256 + 256      // behind the two tile sets of a chunk export: gradient | rc, sd | point costs | line costs (one per thread)
};

// chain step record (CH_STRIDE ints per step, copied into LDS by the kernel): a header, two ints per "front row" and a byte map from
// pose columns to front rows.  The front rows of Euclidean block e_s: rows 0..8 the block itself, then the later-eliminated
// Euclidean block (0 or 9 rows), the coupled pose rows (ascending tangent index, fill of earlier steps included), last the rhs.
enum {
    CH_T0 = 0,      // tangent offset of the block
    CH_R,           // rows between the diagonal block and the rhs row
    CH_NEXT,        // 1: rows 9..17 are the next step's block (coupled through an IMU factor / the prior)
    CH_NSRC,        // number of IMU factors touching the block (<= 2)
    CH_F0, CH_LC0,  // factor index and local column of the block in it (6 or 21)
    CH_F1, CH_LC1,
    CH_PC0,         // prior column of the block's first tangent column (-1: not in the prior)
    CH_SPILL,       // offset of this step's W rows (9 x (npp + 1), column-major 9-vectors) in the workgroup's spill area
    CH_TMASK,       // bit I: pose tile row I (16 tangent columns) holds a column coupled to this block (the rhs column counts)
    CH_INTS = 16,   // row r: int 2r   = tangent index (rhs: 255) | row in the next front << 8 (255 none)
                    //                   | local index in factor 0 << 16 (255 none) | local index in factor 1 << 24
                    //        int 2r+1 = prior column of the row (-1 none)
};
// CH_COLROW: byte c = front row of pose tangent column c (column npp = the rhs), 255: not coupled
enum { CH_W = 9, CH_MAXROWS = 96, CH_COLROW = CH_INTS + 2 * CH_MAXROWS, CH_STRIDE = CH_COLROW + CH_MAXROWS / 4, IMU_BLK = 1024 };   // per-factor J'J block 32 x 32
// LDS pool of the chain phase (doubles): two W buffers (9 x 16 nt_c each), per step L_ee^-1 (9 x 9, offset 0) | L_next,e (9 x 9, offset
// CH_LN), the T-wave's 18 x 9 workspace, then the step records
enum { CH_LT = 164, CH_LN = 82, CH_TA = 164 };
inline int chain_pool_doubles(int n_e, int nt_c) { return 2 * CH_W * 16 * nt_c + n_e * CH_LT + CH_TA + (n_e * CH_STRIDE + 1) / 2 + 8; }


}  // namespace tcv
