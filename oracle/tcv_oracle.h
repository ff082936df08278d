/* CPU oracle for the TC-VIML back-end hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference ships no tests or golden vectors for this path and neither
 * Ceres nor Eigen can be built in the authoring container.  The pin is two
 * independent restatements (this file and oracle/np_oracle.py) agreeing on committed fixtures.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (libtcv_hip.so) never does.
 *
 * Paths cited below are relative to /root/reference/vins_estimator/src/.
 */
#ifndef TCV_ORACLE_H
#define TCV_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_IMU_STRIDE 287 /* dp3 dq4(xyzw) dv3 ba3 bg3 sum_dt1 dp_dba9 dp_dbg9 dq_dbg9 dv_dba9 dv_dbg9 cov225 */

enum { ORC_BLK_POSE = 0, ORC_BLK_SB = 1, ORC_BLK_EX = 2 };

typedef struct {
    int n_frames, n_landmarks, n_imu, n_proj, n_line;
    int ex_constant;
    double *pose;      /* F x 7  (p, q xyzw)              estimator.h:166 para_Pose      */
    double *speedbias; /* F x 9  (v, ba, bg)              estimator.h:167 para_SpeedBias */
    double *ex_pose;   /* 7                               estimator.h:169 para_Ex_Pose   */
    double *lam;       /* L inverse depths                estimator.h:168 para_Feature   */
    const int *imu_i, *imu_j;
    const double *imu_c;    /* n_imu x ORC_IMU_STRIDE */
    const double *imu_sqrt; /* optional n_imu x 225 row-major upper-triangular sqrt_info; NULL = compute */
    const int *proj_i, *proj_j, *proj_l;
    const double *proj_pts; /* n_proj x 6: pts_i xyz, pts_j xyz */
    double proj_sqrt_info, proj_loss_a;
    const int *line_f;
    const double *line_c; /* n_line x 9: start xyz, end xyz, A B C */
    double K[9], Ric[9], Tic[3], line_loss_a;
    double G[3];
    int prior_n, prior_nblk;
    const int *prior_kind, *prior_index, *prior_size, *prior_idx;
    const double *prior_x0; /* concatenated global-size blocks */
    const double *prior_J0; /* n x n column-major (Eigen MatrixXd, marginalization_factor.h:68) */
    const double *prior_r0; /* n */
} orc_window;

typedef struct {
    int num_iterations;      /* entries used in the per-iteration arrays (iteration 0 included) */
    int termination;         /* 0 no-convergence(max it) 1 gradient 2 parameter 3 function 4 radius 5 failure */
    double initial_cost, final_cost;
    double cost[128], cost_candidate[128], model_cost_change[128], radius[128], mu[128], rho[128],
        step_norm[128];
    int step_ok[128], dogleg_case[128];
    double first_delta[2048]; /* local step (unscaled) of iteration 1, camera dims then landmarks */
    int n_local, n_cam;
} orc_summary;

/* I0 integration_base.h:13-158. acc/gyr: (S+1) x 3, sample 0 = constructor (acc_0, gyr_0). */
void orc_preintegrate(const double *acc, const double *gyr, int n_samples, double dt, const double *lin_ba,
                      const double *lin_bg, double acc_n, double gyr_n, double acc_w, double gyr_w,
                      double *imu_c_out /* ORC_IMU_STRIDE */, double *jac_out /* 225, may be NULL */);

/* imu_factor.h:64 */
int orc_imu_sqrt_info(const double *cov, double *sqrt_info /* 225 row-major */);

/* I1 imu_factor.h:19-181.  jac[k] may be NULL; layouts row-major 15x7, 15x9, 15x7, 15x9. */
void orc_imu_evaluate(const double *pose_i, const double *sb_i, const double *pose_j, const double *sb_j,
                      const double *imu_c, const double *G, const double *sqrt_info, double *r, double **jac);
/* P1 projection_factor.cpp:21-124.  row-major 2x7,2x7,2x7,2x1 */
void orc_proj_evaluate(const double *pose_i, const double *pose_j, const double *ex, double lam,
                       const double *pts_i, const double *pts_j, double sqrt_info, double *r, double **jac);
/* T1 projection_td_factor.cpp:34-140.  aux = velocity_i xy, velocity_j xy, td_i, td_j, row_i, row_j; jac[k] row-major 2x7,2x7,2x7,2x1,2x1 */
void orc_proj_td_evaluate(const double *pose_i, const double *pose_j, const double *ex, double lam, double td,
                          const double *pts_i, const double *pts_j, const double *aux, double sqrt_info, double TR, double ROW,
                          double *r, double **jac);
/* L1 line_projection_factor.cpp:19-120.  row-major 2x7 */
void orc_line_evaluate(const double *pose, const double *line_c, const double *K, const double *Ric,
                       const double *Tic, double *r, double *jac);
/* S2 pose_local_parameterization.cpp:3-19 */
void orc_pose_plus(const double *x, const double *delta, double *out);
/* C1 marginalization_factor.cpp:37-68 (CauchyLoss per upstream Ceres). Js: nblk blocks row-major nr x cols[k].
 * Returns block cost. loss_a <= 0 means no loss. */
double orc_loss_correct(int nr, double *r, int nblk, double **Js, const int *cols, double loss_a);
/* M0 marginalization_factor.cpp:335-384; r has n entries; J (optional) n x n_local_cols in prior column order */
void orc_prior_residual(const orc_window *w, double *r);

/* whole solve (G1/G2 + Ceres semantics). Updates the window's state arrays in place. */
int orc_solve(orc_window *w, int max_num_iterations, int fixed_iterations, orc_summary *out);

/* debug: dense H = J'J and g = J'r (unscaled local J, loss-corrected), nlocal = nc + L, row-major */
int orc_linearize_dense(const orc_window *w, double *H, double *g, double *cost, int *nlocal, int *nc);

/* M1-M4, MARGIN_OLD (estimator.cpp:1911-2046).  Outputs the new prior with blocks already shifted.
 * Arrays must hold: kind/index/size/idx >= 2F+1 ints, x0 >= 16F+7, J0 >= n_max^2, r0 >= n_max.
 * A_out/b_out (optional, (m+n)^2 / (m+n)) receive the pre-Schur system, As_out/bs_out the Schur system. */
int orc_marginalize_old(const orc_window *w, int *m_out, int *n_out, int *nblk_out, int *kind, int *index,
                        int *size, int *idx, double *x0, double *J0, double *r0, double *As_out, double *bs_out);

/* G3 Estimator::double2vector() gauge fix, estimator.cpp:1537-1581 (Utility::R2ypr / ypr2R utility.h:70-112), followed by the
 * matrix -> quaternion conversion of the next vector2double() (:1494-1503, Eigen `Quaterniond q{R}`).
 * R0 3x3 row-major, pose n x 7, sb n x 9 -> Rs n x 9 row-major, Ps n x 3, Vs n x 3, pose_out n x 7 (may be NULL) */
void orc_gauge_fix(int n, const double *R0, const double *P0, const double *pose, const double *sb, double *Rs,
                   double *Ps, double *Vs, double *pose_out);

/* symmetric eigen-decomposition (cyclic Jacobi), ascending eigenvalues; V column-major n x n */
void orc_eig_sym(int n, double *A /* n x n, destroyed */, double *evals, double *V);

#ifdef __cplusplus
}
#endif
#endif
