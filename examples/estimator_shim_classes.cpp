// The same toy window as estimator_shim.cpp, written against include/tcv_ceres_shim.hpp: the statements have the shape of
// Estimator::OptimizationWithLine() (reference vins_estimator/src/estimator.cpp:1679-2046) with `ceres::` / the factor classes
// replaced by `tcvshim::`.
//
//   g++ -std=c++14 -Iinclude examples/estimator_shim_classes.cpp -Ltc-viml_amd -ltcv_hip -Wl,-rpath,$PWD/tc-viml_amd -o shim2
#include <cstdio>
#include <cstring>
#include <unordered_map>

#include "tcv_ceres_shim.hpp"

using namespace tcvshim;

int main() {
    if (tcv_device_count() < 1) { std::printf("no HIP device: nothing to run (the library has no CPU path)\n"); return 0; }
    double para_Pose[2][7] = {{0, 0, 0, 0, 0, 0, 1}, {0.1, 0, 0, 0, 0, 0, 1}};
    double para_SpeedBias[2][9] = {{1, 0, 0, 0, 0, 0, 0, 0, 0}, {1, 0, 0, 0, 0, 0, 0, 0, 0}};
    double para_Ex_Pose[1][7] = {{0, 0, 0, 0, 0, 0, 1}};
    double para_Feature[1][1] = {{0.2}};
    const double G[3] = {0, 0, 9.81007};
    tcv_imu_preintegration pre;
    std::memset(&pre, 0, sizeof pre);
    pre.delta_q[3] = 1.0; pre.sum_dt = 0.1;
    pre.delta_p[0] = 0.1; pre.delta_p[2] = 0.5 * 9.81007 * 0.01; pre.delta_v[2] = 9.81007 * 0.1;
    for (int i = 0; i < 15; i++) { pre.jacobian[16 * i] = 1.0; pre.covariance[16 * i] = 1e-4; }
    const double pts_i[3] = {0.0, 0.0, 1.0}, pts_j[3] = {-0.02, 0.0, 1.0};
    try {
        Solver::Summary summary;
        {
            Problem problem;                                                       // estimator.cpp:1679
            problem.SetGravity(G);
            LossFunction *loss_function = new CauchyLoss(1.0);                     // :1682
            for (int i = 0; i < 2; i++) {                                          // :1683-1688
                problem.AddParameterBlock(para_Pose[i], 7, new PoseLocalParameterization());
                problem.AddParameterBlock(para_SpeedBias[i], 9);
            }
            problem.AddParameterBlock(para_Ex_Pose[0], 7, new PoseLocalParameterization());      // :1689-1701
            problem.SetParameterBlockConstant(para_Ex_Pose[0]);
            problem.AddResidualBlock(new IMUFactor(pre), nullptr, para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1]);      // :1728-1731
            problem.AddResidualBlock(new ProjectionFactor(pts_i, pts_j), loss_function, para_Pose[0], para_Pose[1], para_Ex_Pose[0], para_Feature[0]);      // :1766
            Solver::Options options;                                               // :1888-1897
            options.linear_solver_type = SPARSE_SCHUR;
            options.trust_region_strategy_type = DOGLEG;
            options.max_num_iterations = 8;
            Solve(options, &problem, &summary);                                    // :1900
        }                                                                          // problem (and its factors) destroyed here, like :2119
        std::printf("solve: %d iterations, cost %.6g -> %.6g, inverse depth %.6f\n", (int)summary.iterations.size(), summary.initial_cost, summary.final_cost,
                    para_Feature[0][0]);

        // MARGIN_OLD, estimator.cpp:1911-2046
        CauchyLoss loss(1.0);
        MarginalizationInfo *marginalization_info = new MarginalizationInfo();
        marginalization_info->SetGravity(G);
        marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(new IMUFactor(pre), nullptr,
            std::vector<double *>{para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1]}, std::vector<int>{0, 1}));                 // :1936-1942
        marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(new ProjectionFactor(pts_i, pts_j), &loss,
            std::vector<double *>{para_Pose[0], para_Pose[1], para_Ex_Pose[0], para_Feature[0]}, std::vector<int>{0, 3}));                     // :1980-1986
        marginalization_info->preMarginalize();
        marginalization_info->marginalize();
        std::unordered_map<long, double *> addr_shift;                             // :2027-2039
        addr_shift[reinterpret_cast<long>(para_Pose[1])] = para_Pose[0];
        addr_shift[reinterpret_cast<long>(para_SpeedBias[1])] = para_SpeedBias[0];
        addr_shift[reinterpret_cast<long>(para_Ex_Pose[0])] = para_Ex_Pose[0];
        std::vector<double *> parameter_blocks = marginalization_info->getParameterBlocks(addr_shift);
        std::printf("prior: m = %d, n = %d, %d kept blocks, first -> para_Pose[0]: %s\n", marginalization_info->m, marginalization_info->n,
                    (int)parameter_blocks.size(), parameter_blocks[0] == para_Pose[0] ? "yes" : "no");

        // next window: the prior enters as a MarginalizationFactor (:1714-1720)
        {
            Problem problem;
            problem.SetGravity(G);
            problem.AddParameterBlock(para_Pose[0], 7, new PoseLocalParameterization());
            problem.AddParameterBlock(para_SpeedBias[0], 9);
            problem.AddResidualBlock(new MarginalizationFactor(marginalization_info), nullptr, parameter_blocks);
            Solver::Options options;
            options.max_num_iterations = 4;
            Solve(options, &problem, &summary);
            std::printf("prior-only window: cost %.6g -> %.6g\n", summary.initial_cost, summary.final_cost);
        }
        delete marginalization_info;
    } catch (const std::exception &e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
