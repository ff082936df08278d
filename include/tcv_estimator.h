/* tcv_estimator.h -- native (C++) mirror of the per-frame window management around the hot path, behind a C-ABI.
 *
 * What Estimator / FeatureManager do between two calls of OptimizationWithLine (reference vins_estimator/src):
 *   processIMU (estimator.cpp:191-228), processImagewithLine (:230-383: addFeaturesCheckParallax feature_manager.cpp:260-334,
 *   UpdateLinesInFoV / updateLinePairInWindow :385-481, removeLineOutlier feature_manager.cpp:494-534), solveOdometry
 *   (:1476-1490: triangulate feature_manager.cpp:440-492 + OptimizationWithLine), double2vector / setDepth / removeFailures,
 *   failureDetection (:1629-1675), slideWindowWithLinesFoV (:2121-2259, removeBackShiftDepth feature_manager.cpp:559-616,
 *   removeFront :655-696) and the chaining of the marginalisation prior (:2027-2044, :2083-2113).
 * The solver itself is the C-ABI of tcv.h (device pre-integration, fused solve, gauge fix, marginalisation, line association).
 * Out of scope as in the rest of this library: image / line front end (the caller delivers tracked points and line tracks),
 * initialisation (the window is filled from caller-provided states), relocalisation, ESTIMATE_TD.
 *
 * Several estimators (one per sequence) are advanced in lock step by tcv_estimators_optimize(): all full windows of a frame form
 * ONE device batch (BASELINE configs[4]: per-sequence replay, many sequences per GPU).
 */
#ifndef TCV_ESTIMATOR_H
#define TCV_ESTIMATOR_H
#include "tcv.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct tcv_estimator tcv_estimator;

typedef struct {
    double focal_length;        /* FOCAL_LENGTH (parameters.h:19): ProjectionFactor::sqrt_info = focal_length / 1.5 */
    double min_parallax;        /* MIN_PARALLAX = keyframe_parallax / FOCAL_LENGTH (parameters.cpp:73-74) */
    double init_depth;          /* INIT_DEPTH (parameters.cpp:131) */
    double acc_n, gyr_n, acc_w, gyr_w;   /* sensor.yaml:90-93 */
    double gravity[3];          /* G */
    double imu_dt;              /* sample period of the IMU stream */
    double K[9];                /* camera matrix, row-major */
    int width, height;          /* image size in pixels */
    double tic[3], ric[9];      /* camera-IMU extrinsics (row-major rotation) */
    int estimate_extrinsic;
    double angle_th, overlap_th, dist_th;   /* line association gates (sensor.yaml:119-122, threshold) */
    int num_iterations;         /* NUM_ITERATIONS */
    int fixed_iterations;       /* 1: exactly num_iterations trust-region iterations (deterministic) */
    int line_exact_jacobian;    /* 0: the reference's line Jacobian (default); 1: tcv_problem_set_line_jacobian(p, 1), see tcv.h */
    int pad_;
    double solver_time;         /* SOLVER_TIME (sensor.yaml:85 max_solver_time, 0.04 s shipped): the wall budget of a window's solve,
                                   options.max_solver_time_in_seconds of estimator.cpp:1894-1897 -- x 4/5 when the frame marginalises the oldest
                                   frame.  A lock-step frame is ONE batch with ONE budget: as soon as any of its windows is MARGIN_OLD the
                                   whole frame runs on the tighter 4/5 budget.  <= 0 (default) or fixed_iterations: no clock (deterministic) */
} tcv_estimator_config;

/* per-frame statistics of the last optimised window */
typedef struct {
    int marg_flag;              /* 0 MARGIN_OLD, 1 MARGIN_SECOND_NEW */
    int n_landmarks, n_proj, n_line, n_line_obs;
    int iterations, prior_n;    /* iterations = summary.iterations.size() of the window's solve (what estimator.cpp:1902 logs) */
    int termination;            /* tcv_solver_summary::termination of that solve: 0 NO_CONVERGENCE (iteration cap or the solver_time budget), 1 gradient,
                                   2 parameter, 3 function tolerance, 4 radius, 5 FAILURE (this slot was padding until round 5: same size and offsets) */
    double final_cost;
} tcv_estimator_stats;

int tcv_estimator_create(tcv_estimator **out, const tcv_estimator_config *cfg);
void tcv_estimator_destroy(tcv_estimator *e);
/* biases every slot of the window starts from (the reference's initialisation calibrates the gyroscope bias) */
int tcv_estimator_set_biases(tcv_estimator *e, const double ba[3], const double bg[3]);
/* prior 3D line map (n x 6, map frame) and the map -> world transform: switches the estimator to association mode, where
 * lines arrive as (track id, pixel end points) and the 2D-3D association runs every frame */
int tcv_estimator_set_line_map(tcv_estimator *e, int n, const double *lines3d, const double Rbw[9], const double Tbw[3]);

/* processIMU for the samples since the previous frame + the first half of processImagewithLine.
 *   acc / gyr: (n_imu + 1) x 3, row 0 = the previous frame's last sample (n_imu = 0 for the very first frame);
 *   point_ids / points: n_points tracked features, (x, y, 1) on the normalised plane;
 *   lines: association mode: n_lines x 4 pixel end points with line_ids; otherwise n_lines x 9 = 3D start, 3D end (world frame),
 *          A B C of the detected line (line_ids ignored);
 *   truth: 15 doubles P(3) R(9 row-major) V(3) or NULL -- states of the newest slot while the window fills (initialisation is out
 *          of scope);
 *   *ready = 1 when the window is full and must be optimised (tcv_estimators_optimize) before tcv_estimator_finish_frame. */
int tcv_estimator_begin_frame(tcv_estimator *e, int n_imu, const double *acc, const double *gyr, int n_points, const int *point_ids,
                              const double *points, int n_lines, const int *line_ids, const double *lines, const double *truth, int *ready);
/* tcv_estimator_begin_frame for the estimators of a lock-step frame (one input record each), on the library's host worker threads.
 * ready[i] as above; returns TCV_OK or the first failure (rc[i], when given, holds every estimator's own status). */
typedef struct tcv_frame_input {
    int n_imu; const double *acc, *gyr;
    int n_points; const int *point_ids; const double *points;
    int n_lines; const int *line_ids; const double *lines;
    const double *truth;
} tcv_frame_input;
int tcv_estimators_begin_frames(tcv_estimator *const *e, int n, const tcv_frame_input *in, int *ready, int *rc);
/* solveOdometry + double2vector + marginalisation for every estimator in the list (all must be ready): one device batch.
 * The solver options (num_iterations, fixed_iterations) and the IMU noise are those of the first estimator of the list.
 * A window whose own marginalisation fails numerically (eigen-solver sweep cap) does not fail the batch: the other estimators are
 * applied, the call returns TCV_OK, and tcv_estimator_finish_frame of THAT estimator returns TCV_ERR_NUMERIC (reset it, like after
 * failureDetection).  Any other error (HIP, invalid input) fails the whole call and applies nothing.
 * The marginalisation is off the caller's critical path: the call returns when the solve and the gauge fix are done and the new states
 * are applied; the marginalisation (whose result, the next prior, never leaves the device) is launched behind them and runs while the
 * caller finishes this frame and starts the next.  Its status is read at the estimator's NEXT tcv_estimators_optimize: a window that was
 * solved on the prior of a failed marginalisation is not applied and reports TCV_ERR_NUMERIC from tcv_estimator_finish_frame -- one frame
 * later than a failure of the solve itself.  (TCV_EST_MARG_WAIT=1 in the environment: wait for it inside the call, as until round 4.)
 * Thread safety: estimators are independent objects; different host threads may drive different estimator lists, on the same or on
 * different devices (every call issues its copies and kernels on the calling thread's own stream: the threads overlap on the device). */
int tcv_estimators_optimize(tcv_estimator *const *e, int n);
/* The same frame in two calls, for a host thread that overlaps the host side of one group of estimators with the device side of another:
 * _begin returns when the frame's last command is on the device (association, windows, upload, solve, gauge fix, the copy of the states and --
 * frames of few windows -- the marginalisation), _end waits for the states, applies them and consumes the ticket (also on failure).
 * tcv_estimators_optimize = _begin + _end.  Between the two calls the estimators of the ticket must not be touched.  A thread that keeps two
 * tickets in flight puts them on different library streams (tcv_thread_stream_slot(0 / 1) around each group's calls, include/tcv.h):
 * on one stream the second group's association round trip would queue behind the first group's solve. */
typedef struct tcv_opt_ticket tcv_opt_ticket;
int tcv_estimators_optimize_begin(tcv_estimator *const *e, int n, tcv_opt_ticket **ticket);
int tcv_estimators_optimize_end(tcv_opt_ticket *ticket);
/* failureDetection, the published state (Ps / Rs / Vs[WINDOW_SIZE], quaternion x y z w) and slideWindow.
 * TCV_ERR_NUMERIC: failure detection fired (the reference would reset the estimator). */
int tcv_estimator_finish_frame(tcv_estimator *e, double P[3], double q_xyzw[4], double V[3]);
/* tcv_estimator_finish_frame (and tcv_estimator_get_stats, taken before the window slides) for every estimator of a lock-step list, on
 * the library's host worker threads: P 3 n, q_xyzw 4 n, V 3 n doubles, rc n ints (per estimator: what its own finish_frame returns),
 * stats n records or NULL.  Returns TCV_OK when every estimator finished, else the first failure (its text in tcv_last_error). */
int tcv_estimators_finish_frames(tcv_estimator *const *e, int n, double *P, double *q_xyzw, double *V, int *rc, tcv_estimator_stats *stats);
/* Estimator::clearState() + setParameter() (estimator.cpp:126-189, :39-52; estimator_node.cpp:437-446 after a failure): drops the
 * window, the IMU buffers, every feature / line track, the marginalisation prior and the biases; configuration and line map stay.
 * "Reset it" above means this call (or destroy + create). */
int tcv_estimator_reset(tcv_estimator *e);
int tcv_estimator_get_stats(const tcv_estimator *e, tcv_estimator_stats *out);
/* Window tap (test / checkpoint aid, no reference counterpart): with the tap on, every tcv_estimators_optimize keeps a snapshot of the
 * window this estimator handed to the solver -- what OptimizationWithLine sees after vector2double and the graph construction
 * (estimator.cpp:1492-1535, :1683-1846): parameter arrays before and after the solve, factor lists, pre-integrations, the incoming prior.
 * tests/test_gpu_teacher.py re-solves every tapped window of a native replay with the CPU oracle.  The arrays of a snapshot belong to the
 * estimator and stay valid until its next tcv_estimators_optimize / reset / destroy.  The tap waits for device-resident inputs (prior,
 * pre-integrations) to copy them: it costs a frame its overlap, never its results. */
typedef struct tcv_window_snapshot {
    int n_frames, n_landmarks, n_imu, n_proj, n_line, marg_flag, estimate_extrinsic, line_exact_jacobian;
    const double *pose_in, *speedbias_in, *ex_pose_in, *feature_in;        /* before the solve: n_frames x 7, n_frames x 9, 7, n_landmarks */
    const double *pose_out, *speedbias_out, *ex_pose_out, *feature_out;    /* after the solve and the gauge fix (zeros when the window failed) */
    const tcv_imu_preintegration *imu; const int *imu_frame_i, *imu_frame_j;
    const int *proj_frame_i, *proj_frame_j, *proj_feature; const double *proj_pts;      /* n_proj x 6 */
    const int *line_frame; const double *line_data;                                      /* n_line x 9 */
    double line_K[9], line_Ric[9], line_Tic[3], gravity[3], proj_sqrt_info;
    int prior_m, prior_n, prior_nblk;                                                    /* prior_n = 0: no prior */
    const int *prior_block_kind, *prior_block_index, *prior_block_size, *prior_block_idx;
    const double *prior_x0, *prior_J0, *prior_r0;                                        /* concatenated blocks, n x n column-major, n */
    int iterations, applied; double final_cost;
} tcv_window_snapshot;
int tcv_estimator_set_window_tap(tcv_estimator *e, int on);
int tcv_estimator_get_window_snapshot(const tcv_estimator *e, tcv_window_snapshot *out);
/* host-side time accounting of tcv_estimators_optimize since the last call (seconds): out8 = pre-integration, association +
 * triangulation + window, problem construction, batch_create (pack + H2D), kernels (launch to sync), downloads, apply / prior
 * chaining, number of calls.  Development aid (tools/replay_euroc.py --profile). */
int tcv_estimators_profile(double *out8);
/* kernel-side accounting of the lock-step frames since the last call: out8 = { solve-kernel ms (HIP events around every launch, summed),
 * solve launches, marginalisation-kernel ms, marginalisation launches whose batch has been retired, windows solved, sum over those windows
 * of their ALGORITHMIC bytes per linearisation (SURVEY.md 8(d) formula on the window's own factor counts) x linearisations, linearisations,
 * windows marginalised by the retired launches }.  bench.py --mode replay prices the solve kernel against the HBM roofline with it. */
int tcv_estimators_kernel_profile(double *out8);

#ifdef __cplusplus
}
#endif
#endif
