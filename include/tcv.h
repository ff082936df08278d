/* tcv.h -- C-ABI of the MI355X-native sliding-window factor-graph solver that replaces the
 * Ceres-based back end of TC-VIML's vins_estimator (Estimator::OptimizationWithLine,
 * reference vins_estimator/src/estimator.cpp:1677-2119).
 *
 * The reference has no FFI; its hot path talks to Ceres through C++ virtual classes.  Every
 * entry point below names the reference call it stands in for (paths relative to
 * /root/reference/vins_estimator/src/).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions (identical to the reference):
 *   pose block      double[7] = px py pz qx qy qz qw      (estimator.h:166, estimator.cpp:1496-1503)
 *   speed-bias      double[9] = v(3) ba(3) bg(3)          (estimator.h:167)
 *   inverse depth   double[1]                             (estimator.h:168)
 *   Jacobians       row-major num_residuals x global_size (ceres::CostFunction::Evaluate)
 *   all arithmetic  FP64
 *
 * Error convention: every function returns 0 (TCV_OK) or a negative tcv_status; nothing aborts.
 * The reference ignores Ceres failures (estimator.cpp:1900-1903); here they are reported.
 * A handle is single-threaded; distinct handles are independent.
 */
#ifndef TCV_H
#define TCV_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    TCV_OK = 0,
    TCV_ERR_INVALID = -1,     /* bad argument / unknown parameter block / inconsistent sizes     */
    TCV_ERR_NO_DEVICE = -2,   /* no HIP device: the product never falls back to a CPU path        */
    TCV_ERR_TOO_LARGE = -3,   /* window exceeds the on-chip (LDS) budget of the fused solver       */
    TCV_ERR_HIP = -4,         /* HIP runtime error, see tcv_last_error()                           */
    TCV_ERR_UNSUPPORTED = -5, /* structure the fused kernels do not implement                      */
    TCV_ERR_NUMERIC = -6      /* NaN/Inf in inputs or solver failure; caller state left untouched  */
} tcv_status;

typedef enum { TCV_PARAM_EUCLIDEAN = 0, TCV_PARAM_POSE = 1 } tcv_parameterization; /* pose_local_parameterization.h:7-13 */
typedef enum { TCV_LOSS_NONE = 0, TCV_LOSS_CAUCHY = 1 } tcv_loss;                  /* estimator.cpp:1682 CauchyLoss(1.0) */

typedef struct tcv_problem tcv_problem; /* ceres::Problem           estimator.cpp:1679 */
typedef struct tcv_prior tcv_prior;     /* MarginalizationInfo      marginalization_factor.h:46-72 */
typedef struct tcv_batch tcv_batch;     /* device-resident batch of independent windows (throughput mode) */
typedef struct tcv_preint tcv_preint;   /* IntegrationBase whose numbers stay on the device (tcv_preintegrate_device) */

/* IntegrationBase fields read by IMUFactor (imu_factor.h:16,61-79; integration_base.h:188-203). */
typedef struct {
    double delta_p[3];
    double delta_q[4]; /* x y z w */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double sum_dt;
    double jacobian[225];   /* 15x15 row-major, state order P,R,V,BA,BG (parameters.h:66-73) */
    double covariance[225]; /* 15x15 row-major */
} tcv_imu_preintegration;

/* ceres::Solver::Options fields the reference sets (estimator.cpp:1888-1897) + determinism switch. */
typedef struct {
    int max_num_iterations;            /* NUM_ITERATIONS                                             */
    double max_solver_time_in_seconds; /* <= 0 or fixed_iterations: ignore the clock (deterministic); else
                                          checked at the start of every iteration like Ceres, on the device
                                          clock, per window */
    int fixed_iterations;              /* 1: run exactly max_num_iterations, convergence tests off   */
    int workgroups_per_window;         /* 0 (default): a batch planned for the cooperative small-batch mode (tcv_set_cooperative) runs each
                                          window on 1 + H workgroups; 1: one workgroup per window whatever the plan (same chunks, same
                                          additions: bit-identical results -- the A/B switch of the parity tests).
                                          ABI note: until round 2 this slot was `compute_sqrt_info_on_device` (a reserved slot nobody read, default 1);
                                          a caller that still writes 1 here by hand switches the cooperative mode off -- fill the struct
                                          with tcv_solver_options_default() and change only what you mean to */
    int use_mfma;                      /* dense layout only: 1 (default) panel solve and trailing Cholesky update on v_mfma_f64_16x16x4_f64
                                          (the panel is a product with the diagonal tile's inverse plus one refinement step: residual
                                          eps |A|, like substitution), 0 FP64 VALU with forward substitution (debug).  The chain layout
                                          always uses the matrix cores. */
    int threads_per_window;            /* dense layout only: 256 (default) or 512 threads per workgroup; the chain layout is
                                          built for 256 */
    int record_first_step;             /* 1: keep the tangent step of iteration 1 (parity tests)      */
} tcv_solver_options;

#define TCV_MAX_TRACE 64
/* ceres::Solver::Summary subset (estimator.cpp:1899-1902 reads iterations.size()) + parity trace.
 * max_num_iterations is honoured whatever its size (sensor.yaml ships 100); the per-iteration arrays below keep the
 * FIRST TCV_MAX_TRACE entries of the trace, num_iterations keeps counting beyond them. */
typedef struct {
    int num_iterations;   /* = summary.iterations.size(): iteration 0 + accepted + rejected steps (may exceed TCV_MAX_TRACE) */
    int termination;      /* 0 NO_CONVERGENCE 1 gradient 2 parameter 3 function 4 radius 5 FAILURE */
    double initial_cost, final_cost;
    double cost[TCV_MAX_TRACE], cost_candidate[TCV_MAX_TRACE], model_cost_change[TCV_MAX_TRACE];
    double radius[TCV_MAX_TRACE], mu[TCV_MAX_TRACE], rho[TCV_MAX_TRACE], step_norm[TCV_MAX_TRACE];
    int step_ok[TCV_MAX_TRACE], dogleg_case[TCV_MAX_TRACE];
} tcv_solver_summary;

/* Frame-indexed description of one window: the arrays Estimator owns (estimator.h:166-172) plus
 * the factor lists OptimizationWithLine walks.  tcv_problem_from_window() performs the graph
 * construction of estimator.cpp:1683-1846 on it. */
typedef struct {
    int n_frames, n_landmarks, n_imu, n_proj, n_line;
    int estimate_extrinsic;    /* 0: SetParameterBlockConstant(para_Ex_Pose) estimator.cpp:1694-1698 */
    double *para_pose;         /* n_frames x 7, updated in place by solve */
    double *para_speedbias;    /* n_frames x 9 */
    double *para_ex_pose;      /* 7 */
    double *para_feature;      /* n_landmarks */
    const int *imu_frame_i, *imu_frame_j;
    const tcv_imu_preintegration *imu; /* n_imu */
    const int *proj_frame_i, *proj_frame_j, *proj_feature;
    const double *proj_pts;    /* n_proj x 6: pts_i xyz, pts_j xyz */
    double proj_sqrt_info;     /* ProjectionFactor::sqrt_info(0,0) = FOCAL_LENGTH/1.5, estimator.cpp:48 */
    double proj_loss_a;        /* CauchyLoss scale, <= 0: no loss */
    const int *line_frame;
    const double *line_data;   /* n_line x 9: pts_start xyz, pts_end xyz, A B C (line_projection_factor.cpp:6-17) */
    double line_K[9], line_Ric[9], line_Tic[3]; /* row-major; captured constants estimator.cpp:1777-1781,1834 */
    double line_loss_a;
    double gravity[3];         /* global G, parameters.cpp:11,91 */
    const tcv_prior *prior;    /* last_marginalization_info or NULL, estimator.cpp:1714-1720 */
    /* which blocks the prior is attached to, in the prior's keep-block order: kind 0 pose,1 speedbias,2 ex */
    const int *prior_block_kind, *prior_block_index;
    /* ESTIMATE_TD (estimator.cpp:1703-1707, :1757-1763): para_td != NULL turns every point factor into a ProjectionTdFactor
     * on the extra 1-dim block para_Td[0].  proj_td_aux: n_proj x 8 = velocity_i xy, velocity_j xy, td_i, td_j, row_i, row_j
     * (the constructor arguments of projection_td_factor.cpp:6-19); td_TR / td_ROW: the globals TR and ROW (parameters.cpp).
     * A prior block of kind 3 is para_Td. */
    double *para_td;
    const double *proj_td_aux;
    double td_TR, td_ROW;
    int line_exact_jacobian;   /* 0: the reference's line Jacobian; 1: tcv_problem_set_line_jacobian(p, 1) */
    int pad_;
    /* optional (NULL: `imu` above): n_imu device-resident pre-integrations (tcv_preintegrate_device); entry k replaces imu[k].  A window
     * whose IMU factors are all device-resident uploads none of the 287 doubles per factor. */
    const tcv_preint *const *imu_device;
} tcv_window_desc;

/* ---- library ------------------------------------------------------------------------------ */
const char *tcv_version(void);
const char *tcv_last_error(void);
int tcv_device_count(void);            /* 0 when no HIP device is visible */
int tcv_set_device(int device);
/* Device memory this library holds (all devices): bytes in buffers that live objects own (batches, device-resident priors and
 * pre-integrations, ...), bytes kept in its free list for reuse (at most 3 GiB, released under memory pressure), number of live
 * buffers.  Every tcv_*_destroy returns its buffers: after destroying everything `live_bytes` is what it was before (leak checks). */
int tcv_device_memory_stats(unsigned long long *live_bytes, unsigned long long *cached_bytes, int *live_buffers);
/* 0 (default): batches whose speed-bias blocks form chains (every window OptimizationWithLine builds) use the chain layout
 * of the fused solver (speed-biases eliminated block by block before the dense pose system; two windows per CU when the
 * batch has more windows than the device has CUs, otherwise one window per CU with the whole LDS);
 * 1: always the dense 171-dim layout (one window per CU; cross-check / arbitrary graphs).  Read at tcv_batch_create. */
int tcv_set_solver_variant(int variant);
/* Cooperative small-batch mode of the chain layout.  The reference runs ONE estimator at 10 Hz (estimator_node.cpp:282-466); a batch
 * of a handful of windows leaves 250 of 256 CUs idle, so such a batch gives every window a group of 1 + H workgroups: H helpers
 * evaluate the point / line factors chunk by chunk, gather J'J and eliminate the landmarks while the master evaluates the prior and
 * the IMU factors; the master folds their exports into the reduced camera system in a fixed order and runs the chain elimination,
 * the Cholesky factorisation and the dogleg as before.  helpers = -1 (default): automatic (batches with n (1 + H) <= CUs, H from the
 * factor count of the largest window); 0: off; 1..7: that many helpers when the batch allows it.  Read at tcv_batch_create. */
int tcv_set_cooperative(int helpers);

/* ---- ceres::Problem surface (estimator.cpp:1679-1886) ---------------------------------------- */
int tcv_problem_create(tcv_problem **out);
void tcv_problem_destroy(tcv_problem *p);
/* Problem::AddParameterBlock(double*, int[, LocalParameterization*])   estimator.cpp:1686-1687,1693,1838 */
int tcv_problem_add_parameter_block(tcv_problem *p, double *values, int size, int parameterization);
/* Problem::SetParameterBlockConstant   estimator.cpp:1697 */
int tcv_problem_set_parameter_block_constant(tcv_problem *p, double *values);
int tcv_problem_set_gravity(tcv_problem *p, const double G[3]);
/* AddResidualBlock(new IMUFactor(pre), NULL, P_i, SB_i, P_j, SB_j)   estimator.cpp:1728-1731 */
int tcv_problem_add_imu_factor(tcv_problem *p, const tcv_imu_preintegration *pre, double *pose_i,
                               double *speedbias_i, double *pose_j, double *speedbias_j);
/* AddResidualBlock(new ProjectionFactor(pts_i, pts_j), loss, P_i, P_j, Ex, Feature)   estimator.cpp:1766-1767 */
int tcv_problem_add_projection_factor(tcv_problem *p, const double pts_i[3], const double pts_j[3],
                                      double sqrt_info, double loss_a, double *pose_i, double *pose_j,
                                      double *ex_pose, double *inv_depth);
/* AddResidualBlock(new ProjectionTdFactor(pts_i, pts_j, velocity_i, velocity_j, td_i, td_j, row_i, row_j), loss, P_i, P_j, Ex,
 * Feature, Td)   estimator.cpp:1757-1763 (ESTIMATE_TD).  A problem holds either ProjectionFactors or ProjectionTdFactors, all on
 * the same Td block (chain layout like every other window: Td is the last column of the pose part; solver variant 1 = dense
 * layout).  TR / ROW (rolling-shutter read-out time and image rows, globals in the reference) are set once per problem with
 * tcv_problem_set_rolling_shutter (default TR = 0, ROW = 1). */
int tcv_problem_add_projection_td_factor(tcv_problem *p, const double pts_i[3], const double pts_j[3], const double velocity_i[2],
                                         const double velocity_j[2], double td_i, double td_j, double row_i, double row_j,
                                         double sqrt_info, double loss_a, double *pose_i, double *pose_j, double *ex_pose,
                                         double *inv_depth, double *td);
int tcv_problem_set_rolling_shutter(tcv_problem *p, double TR, double ROW);
/* Opt-in extension, NOT the reference's behaviour.  LineProjectionFactor::Evaluate (line_projection_factor.cpp:73-116) fills its
 * Jacobian with the derivative of the squared point-line distance chained through [I | skew(p_cam)] -- a camera-frame perturbation --
 * while the parameter block is the world-frame body pose; exact = 0 (default) reproduces that as written.  exact = 1 uses the
 * derivative of the same residual with respect to PoseLocalParameterization's (delta p, delta theta); the residual is unchanged. */
int tcv_problem_set_line_jacobian(tcv_problem *p, int exact);
/* AddResidualBlock(new LineProjectionFactor(ps, pe, abc, K, Ric, Tic), loss, P_f)   estimator.cpp:1834-1840 */
int tcv_problem_add_line_factor(tcv_problem *p, const double pts_start[3], const double pts_end[3],
                                const double line_abc[3], const double K[9], const double b_c_R[9],
                                const double b_c_T[3], double loss_a, double *pose);
/* AddResidualBlock(new MarginalizationFactor(info), NULL, last_marginalization_parameter_blocks)  :1717-1719 */
/* The problem keeps a pointer to `prior` (like MarginalizationFactor keeps its MarginalizationInfo*): the prior must outlive
 * the problem and every batch created from it. */
int tcv_problem_add_marginalization_factor(tcv_problem *p, const tcv_prior *prior, double *const *blocks,
                                           int num_blocks);
/* graph construction of estimator.cpp:1683-1846 from frame-indexed arrays */
int tcv_problem_from_window(const tcv_window_desc *w, tcv_problem **out);
/* Names the blocks that are the window's frames (para_Pose[i], para_SpeedBias[i], estimator.h:166-167); only
 * tcv_batch_gauge_fix needs it.  tcv_problem_from_window() does this implicitly. */
int tcv_problem_set_frames(tcv_problem *p, int n_frames, double *const *pose, double *const *speedbias);
int tcv_problem_num_parameter_blocks(const tcv_problem *p);
int tcv_problem_num_residual_blocks(const tcv_problem *p);
int tcv_problem_num_residuals(const tcv_problem *p);
/* host-only (no device needed): packs the problem into the device plan/data layout and reports
 * out[16] = nc, nx, npp, nland, tile rows, visual chunks, imu chunks, visual units/items, Schur units/items,
 * imu units/items, plan ints, window doubles, LDS bytes.  Error codes as tcv_batch_create. */
int tcv_problem_plan_stats(const tcv_problem *p, int *out16);
/* Packer diagnostics (host only, no reference counterpart).  tcv_problem_plan_ints: the plan the device would get for this problem --
 * its header (as ints) followed by the int pool; *len = number of ints (out may be NULL or too small: only *len is set then).
 * tcv_set_packer_reference(1): plans are built by the generic gather-program builder without any cache -- the reference the fast
 * builder is compared with int by int (tests/test_pack_cpu.py).  tcv_plan_cache_stats: out4 = { whole-plan cache hits, misses,
 * camera-half cache hits, misses } since the process started. */
int tcv_problem_plan_ints(const tcv_problem *p, int *out, int cap, int *len);
int tcv_set_packer_reference(int on);
int tcv_plan_cache_stats(long long *out4);
/* the packing pass of tcv_batch_create (plans and data sizes of n problems on `threads` of the library's host worker threads, chunked
 * for `coop_chunks` helper workgroups, 0 = single-workgroup plans) without a device; *seconds = wall time.  tools/dev_pack_bench.py */
int tcv_problems_pack_bench(tcv_problem *const *problems, int n, int threads, int coop_chunks, double *seconds);

/* ceres::Solve(options, &problem, &summary)   estimator.cpp:1900.  Updates the caller's blocks in place. */
void tcv_solver_options_default(tcv_solver_options *o);
int tcv_solve(const tcv_solver_options *o, tcv_problem *p, tcv_solver_summary *summary);

/* ---- MarginalizationInfo surface (marginalization_factor.cpp:89-321, estimator.cpp:1913-2044) -- */
/* `p` holds exactly the factors the reference would wrap in ResidualBlockInfo; `drop` lists the
 * parameter blocks of the drop_sets.  Equivalent to addResidualBlockInfo* + preMarginalize +
 * marginalize.  The new prior keeps the current values of the kept blocks as linearisation point.
 * A marginalisation that keeps nothing (every block its factors touch is dropped) returns a prior with n = 0 and no blocks, like the
 * reference's empty MarginalizationInfo (marginalization_factor.cpp:174-194); tcv_problem_add_marginalization_factor accepts it as a
 * factor without residuals (estimator.cpp:1714-1720), tcv_prior_create builds one from n = 0, num_blocks = 0. */
int tcv_marginalize(tcv_problem *p, double *const *drop, int num_drop, tcv_prior **out);
/* MarginalizationInfo fields: m, n, keep_block_size/idx/data, linearized_jacobians (n x n,
 * column-major like Eigen::MatrixXd), linearized_residuals (marginalization_factor.h:57-70).
 * keep_block_idx counts from the start of the [m | n] ordering exactly as in the reference (>= m). */
int tcv_prior_create(tcv_prior **out, int m, int n, int num_blocks, const int *keep_block_size,
                     const int *keep_block_idx, const double *keep_block_data_concat,
                     const double *linearized_jacobians, const double *linearized_residuals);
int tcv_prior_dims(const tcv_prior *pr, int *m, int *n, int *num_blocks, int *sum_block_size);
int tcv_prior_export(const tcv_prior *pr, int *keep_block_size, int *keep_block_idx,
                     double *keep_block_data_concat, double *linearized_jacobians, double *linearized_residuals);
/* parity/debug: Schur system A' (n x n row-major), b' the factors were taken from (marginalization_factor.cpp:281-282) */
int tcv_prior_export_schur(const tcv_prior *pr, double *A_schur, double *b_schur);
/* getParameterBlocks(): addresses (un-shifted) of the kept blocks, caller applies addr_shift  :301-321 */
int tcv_prior_keep_block_addresses(const tcv_prior *pr, double **addresses);
void tcv_prior_destroy(tcv_prior *pr);

/* ---- throughput mode: many independent windows resident in HBM ------------------------------ */
/* marg_problems[i] (optional, may be NULL array) shares parameter-block addresses with problems[i]
 * and holds the marginalisation factor set; drop lists as in tcv_marginalize.  Single entries may be NULL too: window i is then only
 * solved (a MARGIN_SECOND_NEW frame whose prior does not hold para_Pose[WINDOW_SIZE - 1], estimator.cpp:2049-2050) -- the windows of a
 * lock-step frame form ONE batch whether or not they marginalise; tcv_batch_get_prior fails for such a window, the get_priors calls leave
 * its entry NULL. */
int tcv_batch_create(tcv_batch **out, tcv_problem *const *problems, tcv_problem *const *marg_problems,
                     double *const *const *marg_drop, const int *marg_num_drop, int n);
/* The marginalisation problems of a batch created with marg_problems == NULL, attached afterwards (same arguments as tcv_batch_create):
 * their packing and upload may run while tcv_batch_solve of the same batch is on the device -- the native estimator overlaps them
 * with the solve.  (The solve of such a batch does not hand its IMU factor's sqrt_info to the marginalisation, which forms its own:
 * same bits.)  Once per batch, before tcv_batch_marginalize. */
int tcv_batch_attach_marginalization(tcv_batch *b, tcv_problem *const *marg_problems, double *const *const *marg_drop, const int *marg_num_drop);
void tcv_batch_destroy(tcv_batch *b);
/* one pass of the hot path over the batch: solve every window from its uploaded initial state
 * (and, if marg problems were given, marginalise at the solution).  Asynchronous on `hip_stream`
 * (a hipStream_t cast to void*, NULL = default stream, TCV_STREAM_THREAD = the calling thread's own stream, on which this library
 * issues its uploads, splices and downloads anyway: host threads that each drive their own batches then use one stream each -- the
 * runtime maps streams onto a handful of hardware queues, fewer streams collide less); inputs and outputs stay in HBM. */
#define TCV_STREAM_THREAD ((void *)(~(size_t)0))
/* Which of the calling thread's two library streams "its own stream" means from now on (0, the default, or 1; created at first use): a host
 * thread that drives two independent groups of work -- one's host side against the other's kernels -- gives each group a stream of its own
 * by selecting the slot before every call it makes for that group.  Returns the previous slot. */
int tcv_thread_stream_slot(int slot);
int tcv_batch_solve(tcv_batch *b, const tcv_solver_options *o, void *hip_stream);
int tcv_batch_marginalize(tcv_batch *b, void *hip_stream);
/* Estimator::double2vector() gauge fix (estimator.cpp:1537-1581) followed by vector2double() (:1492-1512), in place
 * on the solved states in HBM, so that the marginalisation linearises at the gauge-fixed states exactly as the
 * reference does (double2vector :1905, vector2double :1915).  The origin (Rs[0], Ps[0]) is the uploaded initial
 * pose of frame 0.  Needs the frame table (tcv_problem_set_frames / tcv_problem_from_window). */
int tcv_batch_gauge_fix(tcv_batch *b, void *hip_stream);
/* The same fix in the solve kernel's epilogue, where the solved states are still on chip: from the next tcv_batch_solve on the batch's states come out
 * gauge-fixed (one kernel launch and its gap less per frame) and tcv_batch_gauge_fix behind such a solve is a no-op.  Same arithmetic, same bits
 * (tests/test_gpu_gauge.py).  Needs the frame tables, like tcv_batch_gauge_fix. */
int tcv_batch_set_fused_gauge_fix(tcv_batch *b, int on);
int tcv_batch_synchronize(tcv_batch *b);
/* copy results back: states into the callers' parameter blocks, summaries, priors */
int tcv_batch_download_states(tcv_batch *b);
int tcv_batch_get_summaries(tcv_batch *b, tcv_solver_summary *out, int n);
/* tcv_batch_download_states plus, per window, the three summary numbers a per-frame caller reads -- Summary::iterations.size() (the one the
 * reference reads, estimator.cpp:1902), termination type, final cost -- in ONE device round trip (any of the arrays may be NULL; n = batch size) */
int tcv_batch_download_states_brief(tcv_batch *b, int *num_iterations, int *termination, double *final_cost);
/* the same in two halves, for callers that enqueue more work behind the solve: _begin puts the copies on `hip_stream` (behind the batch's
 * work in flight) and returns; _end waits for the copies only -- not for what the caller enqueued behind them, e.g. tcv_batch_marginalize --
 * and writes the states into the callers' blocks.  Between the two calls the parameter blocks must not be read. */
int tcv_batch_download_states_begin(tcv_batch *b, void *hip_stream);
int tcv_batch_download_states_end(tcv_batch *b, int *num_iterations, int *termination, double *final_cost);
int tcv_batch_get_prior(tcv_batch *b, int window, tcv_prior **out);
/* every window's prior in one call (n = the batch size): out[k] as tcv_batch_get_prior(b, k, &out[k]) would return it; the host copies are
 * made by several host threads.  On an error nothing is returned (out[] is all NULL). */
int tcv_batch_get_priors(tcv_batch *b, tcv_prior **out, int n);
/* optional: ONE device-to-host copy of every window's marginalisation result; later tcv_batch_get_prior calls are served from it
 * (until the next tcv_batch_marginalize) */
int tcv_batch_download_priors(tcv_batch *b);
/* the same without A', b' (the parity / debug half of the result: tcv_prior_export_schur fails on priors obtained this way): what an
 * estimator needs per frame -- J0, r0 and the linearisation point, 62 KB instead of 113 KB per window, into pinned host memory */
int tcv_batch_download_priors_compact(tcv_batch *b);
/* DEVICE-RESIDENT frame-to-frame state: what `last_marginalization_info` is between two frames of the reference (estimator.h:176-177,
 * estimator.cpp:2027-2044).  out[k] is the prior of window k with its LAYOUT on the host (m, n, keep_block_size / idx, the un-shifted block
 * addresses: everything tcv_prior_dims / tcv_prior_keep_block_addresses and the graph construction of the next window need) and its
 * NUMBERS -- linearized_jacobians, linearized_residuals, keep_block_data -- left where the marginalisation kernel wrote them, in HBM (the
 * handle keeps that buffer alive after tcv_batch_destroy).  A problem that holds such a prior (tcv_problem_add_marginalization_factor /
 * tcv_window_desc::prior as usual) packs and uploads nothing of it: tcv_batch_create copies J0 | r0 | x0 device-to-device into the new
 * batch's pool, without the rows the marginalisation thresholded, exactly as the host path lays them out -- the next solve reads the same
 * bits.  Per window only two ints come down (status, number of thresholded rows).  tcv_prior_export materialises the numbers on the host
 * on demand (export / checkpoint); tcv_prior_export_schur is not available.  n = the batch size; on an error out[] is all NULL. */
int tcv_batch_get_priors_device(tcv_batch *b, tcv_prior **out, int n);
/* The same without waiting for the marginalisation (tcv_batch_marginalize may still be running): takes the marginalisation off the host's
 * critical path -- estimator.cpp publishes the frame's pose after it (:2027-2044 run inside optimization()), here the next frame's host work
 * overlaps it.  The handles are ordered behind the producing kernel on the device (an event the next tcv_batch_create waits for on its
 * stream); the number of thresholded rows, which sizes the host layout of a prior, is read on the device.  What the host does not know
 * yet is whether the marginalisation succeeded: ask tcv_batch_marg_status (it waits) before trusting anything computed on such a prior --
 * a failed one (status != 0, or a NaN) poisons the consumer's window, it does not crash it.  Keep the batch alive until then. */
int tcv_batch_get_priors_device_async(tcv_batch *b, tcv_prior **out, int n);
/* 1 if the prior's numbers live on the device (tcv_batch_get_priors_device) and have not been materialised on the host, else 0 */
int tcv_prior_is_device_resident(const tcv_prior *pr);
/* replaces the prior of the problem's marginalisation factor by one with the SAME layout (n, keep_block_size / idx), keeping the
 * factor's parameter blocks: a caller whose graph does not change from frame to frame (a benchmark loop; a window in steady state)
 * re-uses its problem object and only hands over the new last_marginalization_info */
int tcv_problem_set_marginalization_prior(tcv_problem *p, const tcv_prior *prior);
/* the same for n problems at once (problems[k] takes priors[k]), and tcv_prior_destroy for n handles (NULL entries are skipped): a
 * throughput loop hands a whole batch's priors on with two calls instead of 2 n */
int tcv_problems_set_marginalization_prior(tcv_problem *const *problems, tcv_prior *const *priors, int n);
void tcv_priors_destroy(tcv_prior *const *priors, int n);
/* per-window status of the last marginalisation: 0 ok, 1 an eigen-solver hit its sweep cap, 2 result produced by the
 * cyclic-Jacobi safety net (the tridiagonal eigen-solver failed its orthogonality / trace self-check), -1 not run, -2 a NaN in the
 * result (a prior made from it is unusable).  Waits for the batch's work in flight. */
int tcv_batch_marg_status(tcv_batch *b, int *out, int n);
/* tangent step of iteration 1 (needs record_first_step): free camera blocks in the order they were
 * added (local size each), then the inverse depths in order of first use.  Parity/debug surface. */
int tcv_batch_get_first_step(tcv_batch *b, int window, double *out, int cap, int *len);
/* number of distinct graph structures (plans) in the batch, their bytes, launch grid and LDS bytes */
int tcv_batch_plan_stats(tcv_batch *b, int *num_plans, double *plan_bytes, int *grid, int *lds_bytes);
/* cooperative mode of this batch (tcv_set_cooperative): helper workgroups per window (0: one workgroup per window), window groups
 * resident at a time, visual chunks of the largest plan, and the workgroups per window the LAST tcv_batch_solve actually used: 1 when the
 * option asked for it or when the cooperative launches already in flight on the device left no room (all cooperative grids in flight
 * must fit the chip together; the fall-back runs the same plan and returns the same bits) */
int tcv_batch_cooperative(const tcv_batch *b, int *helpers, int *groups, int *chunks, int *last_solve_workgroups);
/* layout of the fused solver chosen for this batch: 0 chain (speed-biases eliminated block by block), 1 dense; < 0: error */
int tcv_batch_layout(const tcv_batch *b);
/* bytes of window input resident in HBM and elapsed milliseconds of the last solve / marginalise
 * kernels measured with HIP events on the launch stream */
int tcv_batch_stats(tcv_batch *b, double *input_bytes, double *solve_ms, double *marg_ms);
int tcv_batch_size(const tcv_batch *b);

/* ---- Estimator::double2vector() gauge fix (estimator.cpp:1537-1581) ------------------------ */
/* The yaw of frame 0 and its position are unobservable: after the solve the window is rotated back about the
 * vertical by the yaw drift of frame 0 (Utility::R2ypr / ypr2R, utility.h:70-112; full rotation near the Euler
 * singularity, :1556-1563) and re-anchored at the original Ps[0].  Host pointers; evaluated on the GPU.
 *   origin_R0 (3x3 row-major) / origin_P0 : Rs[0], Ps[0] before the solve (or last_R0 / last_P0 after a failure, :1542-1547)
 *   para_pose n x 7, para_speedbias n x 9  : the solved parameter blocks (inputs, untouched)
 *   Rs n x 9 row-major, Ps n x 3, Vs n x 3 : outputs (:1565-1581);  pose_out n x 7 (optional) = what the next
 *   vector2double() writes from them (:1494-1503, Eigen's matrix -> quaternion conversion) */
int tcv_gauge_fix(int n_frames, const double origin_R0[9], const double origin_P0[3], const double *para_pose,
                  const double *para_speedbias, double *Rs, double *Ps, double *Vs, double *pose_out);

/* ---- 2D-3D line association (the step that feeds the line factors) --------------------------------- */
/* Estimator::UpdateLinesInFoV (estimator.cpp:385-447) for every frame of the window and Estimator::LineCorrespondenceInFrame
 * (:671-885, with CalAngleDist :602-613, CalEulerDist :615-669, Line2D feature_manager.cpp:4-73) for every detected 2D line.
 * Host pointers; evaluated on the GPU.
 *   poses n_frames x 7, ex_pose 7 (para_Pose / para_Ex_Pose); Rbw (3x3 row-major), Tbw: prior-map -> VIO world (sensor.yaml:40-57);
 *   K 3x3 row-major; width / height in pixels; window_size = WINDOW_SIZE (FoV margin, :405-408);
 *   lines3d n_map x 6: end points of the prior map lines (line_3d.txt rows);
 *   det_frame / det_lines n_det x 4: frame index and pixel end points xs ys xe ye of each detected line;
 *   angle_th [rad], overlap_th (sensor.yaml:119-122).
 * fov_given != 0: in_fov is an INPUT (the reference freezes WorldLinesInFOV[i] when frame i enters the window, :328, and re-matches
 * against it with the current poses, updateLinePairInWindow :449-481).  fov_given = 2 + f: the same, except that row f is computed
 * first (UpdateLinesInFoV(f) for the frame that has just entered) and written back to in_fov: one call per image instead of two.
 * Outputs (any may be NULL): in_fov n_frames x n_map (WorldLinesInFOV membership), match_index n_det (row of lines3d, -1: no
 * credible line), err n_det x 3 = errA, errD, overlap as the reference's Eigen::Vector3f (-1 -1 -1: none), projected n_det x 4
 * (pixel end points of the chosen projected line; the detected line itself when there is no match). */
int tcv_match_lines(int n_frames, const double *poses, const double *ex_pose, const double *Rbw, const double *Tbw, const double *K,
                    int width, int height, int window_size, int n_map, const double *lines3d, int n_det, const int *det_frame,
                    const double *det_lines, double angle_th, double overlap_th, int fov_given, unsigned char *in_fov, int *match_index,
                    float *err, double *projected);

/* the same for n independent calls (one per sequence of a lock-step frame: every sequence has its own map, poses and detections) with ONE
 * upload, ONE download and ONE synchronisation for all of them; call k behaves exactly like tcv_match_lines(args[k]...) */
typedef struct tcv_match_lines_args {
    int n_frames; const double *poses, *ex_pose, *Rbw, *Tbw, *K; int width, height, window_size, n_map; const double *lines3d;
    int n_det; const int *det_frame; const double *det_lines; double angle_th, overlap_th; int fov_given;
    unsigned char *in_fov; int *match_index; float *err; double *projected;
    const struct tcv_line_map *map_device;      /* NULL, or the same n_map lines resident on the device (tcv_line_map_create): nothing of the map is uploaded */
} tcv_match_lines_args;
int tcv_match_lines_batch(int n, const tcv_match_lines_args *args);
/* The 3D line map of a sequence (`lines_3d`, n x 6: what setParameters reads from line_3d.txt once, estimator.cpp:54-124) resident in HBM for
 * the life of the handle: the per-frame association then uploads poses and detections only (the map is 43 KB per call otherwise). */
typedef struct tcv_line_map tcv_line_map;
int tcv_line_map_create(tcv_line_map **out, int n_map, const double *lines3d);
void tcv_line_map_destroy(tcv_line_map *m);

/* ---- IMU pre-integration (the producer of the IMU factor's constants) ------- */
/* Batched `IntegrationBase(acc_0, gyr_0, linearized_ba, linearized_bg)` followed by `push_back(dt, acc, gyr)` for every
 * buffered sample (integration_base.h:13-36, propagate :130-158, midPointIntegration :54-128).  `repropagate(ba, bg)`
 * (:38-52) is the same call with the new linearisation biases.  Host pointers.
 *   first/count[i]   : sample rows of pre-integration i in `samples7`
 *   samples7         : num_samples x 7 = dt, acc xyz, gyr xyz   (dt_buf / acc_buf / gyr_buf)
 *   acc0_gyr0_ba_bg  : n x 12
 *   noise            : ACC_N, GYR_N, ACC_W, GYR_W (parameters.h; noise matrix integration_base.h:21-27) */
int tcv_preintegrate(int n, const int *first, const int *count, const double *samples7, int num_samples,
                     const double *acc0_gyr0_ba_bg, const double noise[4], tcv_imu_preintegration *out);

/* The same with the results LEFT ON THE DEVICE -- what `pre_integrations[]` are between two frames of the reference (estimator.h:
 * pre_integrations[WINDOW_SIZE + 1], estimator.cpp:200-206): out[i] is a handle on pre-integration i in HBM.  A problem takes it through
 * tcv_problem_add_imu_factor_device (or tcv_window_desc::imu_device); tcv_batch_create copies the factor's constants device-to-device.
 * Only sum_dt -- the one number the window management reads (estimator.cpp:1726 `sum_dt > 10.0`) -- is kept on the host
 * (tcv_preint_sum_dt: the sum of the dt column, added in the order the kernel adds it); tcv_preint_export materialises everything. */
int tcv_preintegrate_device(int n, const int *first, const int *count, const double *samples7, int num_samples,
                            const double *acc0_gyr0_ba_bg, const double noise[4], tcv_preint **out);
double tcv_preint_sum_dt(const tcv_preint *pre);
int tcv_preint_export(const tcv_preint *pre, tcv_imu_preintegration *out);
void tcv_preint_destroy(tcv_preint *pre);
/* IMUFactor(pre_integrations[j]) on a device-resident pre-integration (estimator.cpp:1723-1732); the handle must outlive the
 * tcv_batch_create / tcv_solve / tcv_marginalize calls that use the problem */
int tcv_problem_add_imu_factor_device(tcv_problem *p, const tcv_preint *pre, double *pose_i, double *speedbias_i, double *pose_j,
                                      double *speedbias_j);

/* ---- batched factor evaluation (parity / debug surface; CostFunction::Evaluate layout) ------- */
/* All pointers are HOST pointers; the call uploads, evaluates on the GPU, downloads.
 * jacobians may be NULL.  Layouts follow the reference: row-major, global block width. */
/* IMUFactor::Evaluate imu_factor.h:19-181: params n x (7+9+7+9), residuals n x 15, jacobians n x (15*7+15*9+15*7+15*9).
 * sqrt_info_io: n x 225; if use_given_sqrt_info == 0 it is computed on the device and written back. */
int tcv_eval_imu_factors(int n, const tcv_imu_preintegration *pre, const double *params, const double G[3],
                         int use_given_sqrt_info, double *sqrt_info_io, double *residuals, double *jacobians);
/* ProjectionFactor::Evaluate projection_factor.cpp:21-124: params n x (7+7+7+1), pts n x 6, residuals n x 2, jacobians n x (14+14+14+2) */
int tcv_eval_projection_factors(int n, const double *pts, const double *params, double sqrt_info,
                                double *residuals, double *jacobians);
/* ProjectionTdFactor::Evaluate projection_td_factor.cpp:34-140 (camera-IMU time offset + rolling shutter; selected by ESTIMATE_TD,
 * estimator.cpp:1757, which is 0 in every shipped configuration; the fused solver takes it through tcv_problem_add_projection_td_factor /
 * tcv_window::para_td above -- this entry point is the per-factor parity check):
 * params n x (7+7+7+1+1) = pose_i, pose_j, ex_pose, inverse depth, td; pts n x 6; aux n x 8 = velocity_i xy, velocity_j xy, td_i,
 * td_j, row_i, row_j (constructor arguments, :6-18); TR / ROW = rolling-shutter read-out time / image height (parameters.cpp:92,145);
 * residuals n x 2, jacobians n x (14+14+14+2+2) */
int tcv_eval_projection_td_factors(int n, const double *pts, const double *aux, const double *params, double sqrt_info,
                                   double TR, double ROW, double *residuals, double *jacobians);
/* LineProjectionFactor::Evaluate line_projection_factor.cpp:19-120: params n x 7, line n x 9, residuals n x 2, jacobians n x 14 */
int tcv_eval_line_factors(int n, const double *line_data, const double K[9], const double b_c_R[9],
                          const double b_c_T[3], const double *params, double *residuals, double *jacobians);
/* PoseLocalParameterization::Plus pose_local_parameterization.cpp:3-19: x n x 7, delta n x 6 */
int tcv_pose_plus(int n, const double *x, const double *delta, double *x_plus_delta);

/* Measurement aid (no reference counterpart; SURVEY.md 8(d) "FP64 peak ... microbench it"): the FP64 rate of the current device,
 * measured with dependence-free chains of v_fma_f64 and of v_mfma_f64_16x16x4_f64 on every CU.
 * out4 = { vector FMA TFLOP/s, MFMA TFLOP/s, compute units, shader clock reported by the runtime [MHz] }. */
int tcv_microbench_fp64(double *out4);

#ifdef __cplusplus
}
#endif
#endif /* TCV_H */
