// Minimal C++ caller of the C-ABI (include/tcv.h), shaped like the calls a retargeted vins_estimator/src/estimator.cpp makes in
// Estimator::OptimizationWithLine() (reference :1677-2119): build the problem on the estimator's own arrays, solve in place,
// gauge-fix, marginalise the oldest frame, hand the prior to the next window.  INTEGRATION.md walks through the same calls.
//
//   g++ -std=c++14 -Iinclude examples/estimator_shim.cpp -Ltc-viml_amd -ltcv_hip -Wl,-rpath,$PWD/tc-viml_amd -o shim
//
// It runs on a toy window (two frames, one landmark seen from both, one IMU factor with an identity-like pre-integration) and only
// demonstrates the call sequence and the error convention; the parity tests live in tests/.
#include <cstdio>
#include <cstring>
#include <vector>

#include "tcv.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != TCV_OK) { std::printf("%s -> %d (%s)\n", #call, rc_, tcv_last_error()); return rc_ == TCV_ERR_NO_DEVICE ? 0 : 1; } \
    } while (0)

int main() {
    std::printf("%s, %d HIP device(s)\n", tcv_version(), tcv_device_count());
    // the estimator's state arrays (estimator.h:166-172)
    double para_Pose[2][7] = {{0, 0, 0, 0, 0, 0, 1}, {0.1, 0, 0, 0, 0, 0, 1}};
    double para_SpeedBias[2][9] = {{1, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 0, 0, 0, 0, 0, 0, 0, 0}};
    double para_Ex_Pose[1][7] = {{0, 0, 0, 0, 0, 0, 1}};
    double para_Feature[1][1] = {{0.2}};
    const double G[3] = {0, 0, 9.81007};

    tcv_problem *problem = nullptr;
    CHECK(tcv_problem_create(&problem));
    CHECK(tcv_problem_set_gravity(problem, G));
    for (int i = 0; i < 2; i++) {
        CHECK(tcv_problem_add_parameter_block(problem, para_Pose[i], 7, TCV_PARAM_POSE));
        CHECK(tcv_problem_add_parameter_block(problem, para_SpeedBias[i], 9, TCV_PARAM_EUCLIDEAN));
    }
    CHECK(tcv_problem_add_parameter_block(problem, para_Ex_Pose[0], 7, TCV_PARAM_POSE));
    CHECK(tcv_problem_set_parameter_block_constant(problem, para_Ex_Pose[0]));      // ESTIMATE_EXTRINSIC == 0

    tcv_imu_preintegration pre;
    std::memset(&pre, 0, sizeof pre);
    pre.delta_q[3] = 1.0; pre.sum_dt = 0.1;
    pre.delta_p[0] = 0.1; pre.delta_p[2] = 0.5 * 9.81007 * 0.01; pre.delta_v[2] = 9.81007 * 0.1;      // free fall compensated: standing still + moving 1 m/s in x
    for (int i = 0; i < 15; i++) { pre.jacobian[16 * i] = 1.0; pre.covariance[16 * i] = 1e-4; }
    CHECK(tcv_problem_add_imu_factor(problem, &pre, para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1]));

    const double pts_i[3] = {0.0, 0.0, 1.0}, pts_j[3] = {-0.02, 0.0, 1.0};        // a point 5 m ahead, camera moved 0.1 m in x
    CHECK(tcv_problem_add_projection_factor(problem, pts_i, pts_j, 460.0 / 1.5, 1.0, para_Pose[0], para_Pose[1], para_Ex_Pose[0], para_Feature[0]));
    double *pose_frames[2] = {para_Pose[0], para_Pose[1]}, *sb_frames[2] = {para_SpeedBias[0], para_SpeedBias[1]};
    CHECK(tcv_problem_set_frames(problem, 2, pose_frames, sb_frames));
    std::printf("problem: %d parameter blocks, %d residual blocks, %d residuals\n", tcv_problem_num_parameter_blocks(problem),
                tcv_problem_num_residual_blocks(problem), tcv_problem_num_residuals(problem));

    tcv_solver_options opt;
    tcv_solver_options_default(&opt);
    opt.max_num_iterations = 8;
    tcv_solver_summary summary;
    CHECK(tcv_solve(&opt, problem, &summary));                                       // ceres::Solve: blocks updated in place
    std::printf("solve: %d iterations, cost %.6g -> %.6g, inverse depth %.6f\n", summary.num_iterations, summary.initial_cost, summary.final_cost,
                para_Feature[0][0]);

    // marginalise frame 0 (MARGIN_OLD): same factors, drop pose 0 / speed-bias 0 / the landmark anchored there
    double *drop[3] = {para_Pose[0], para_SpeedBias[0], para_Feature[0]};
    tcv_prior *prior = nullptr;
    CHECK(tcv_marginalize(problem, drop, 3, &prior));
    int m, n, nb, xs;
    CHECK(tcv_prior_dims(prior, &m, &n, &nb, &xs));
    std::vector<double *> keep(nb);
    CHECK(tcv_prior_keep_block_addresses(prior, keep.data()));                       // the caller applies addr_shift (estimator.cpp:2027-2039)
    std::printf("prior: m = %d, n = %d, %d kept blocks\n", m, n, nb);
    tcv_prior_destroy(prior);

    // The same frame through the batched entry points, in the order a per-frame loop uses them with the marginalisation OFF the caller's
    // path: the solve problems alone make the batch; the marginalisation problems are attached while the solve runs;
    // the states come back as soon as solve + gauge fix are done; the marginalisation is launched behind them and its prior is handed on
    // as a device-resident handle without a wait; its status is asked for later.
    tcv_batch *batch = nullptr;
    tcv_problem *solve_problems[1] = {problem}, *marg_problems[1] = {problem};      // (this toy marginalises with every factor of the window)
    double *const *drops[1] = {drop};
    const int ndrops[1] = {3};
    CHECK(tcv_batch_create(&batch, solve_problems, nullptr, nullptr, nullptr, 1));
    CHECK(tcv_batch_solve(batch, &opt, nullptr));
    CHECK(tcv_batch_gauge_fix(batch, nullptr));
    CHECK(tcv_batch_attach_marginalization(batch, marg_problems, drops, ndrops));
    CHECK(tcv_batch_download_states(batch));                                         // the frame's result: publish the pose here
    CHECK(tcv_batch_marginalize(batch, nullptr));
    tcv_prior *next_prior[1] = {nullptr};
    CHECK(tcv_batch_get_priors_device_async(batch, next_prior, 1));                  // no wait; the next tcv_batch_create is ordered behind the kernel
    int status[1] = {-1};
    CHECK(tcv_batch_marg_status(batch, status, 1));                                  // (waits) 0: the prior is good
    CHECK(tcv_prior_dims(next_prior[0], &m, &n, &nb, &xs));
    std::printf("batched frame: marginalisation status %d, prior m = %d, n = %d, device-resident %d\n", status[0], m, n, tcv_prior_is_device_resident(next_prior[0]));
    tcv_prior_destroy(next_prior[0]);
    tcv_batch_destroy(batch);
    tcv_problem_destroy(problem);
    return 0;
}
