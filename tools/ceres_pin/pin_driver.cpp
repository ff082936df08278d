// Pin driver: the ceres::Problem of Estimator::OptimizationWithLine (vins_estimator/src/estimator.cpp:1679-1900) rebuilt from the golden
// window of this repository (tools/ceres_pin/dump_window.py) and solved by the REAL Ceres with the reference's own factor classes.
// Built by tools/ceres_pin/CMakeLists.txt against a checkout of the reference; cannot be built in this repository's image (no Eigen / Ceres /
// ROS there) and is not part of the product.  Usage: pin_driver golden_window.txt ceres_pin.txt [max_num_iterations = 100]
#include <ceres/ceres.h>
#include <Eigen/Dense>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "parameters.h"
#include "factor/imu_factor.h"
#include "factor/integration_base.h"
#include "factor/line_projection_factor.h"
#include "factor/marginalization_factor.h"
#include "factor/pose_local_parameterization.h"
#include "factor/projection_factor.h"

static std::vector<double> nums(std::istringstream &is) { std::vector<double> v; double x; while (is >> x) v.push_back(x); return v; }

int main(int argc, char **argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: pin_driver golden_window.txt out.txt [max_num_iterations]\n"); return 2; }
    const int max_it = argc > 3 ? std::atoi(argv[3]) : 100;
    std::ifstream in(argv[1]);
    if (!in) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    int nf = 0, L = 0, n_imu = 0, n_proj = 0, n_line = 0, has_prior = 0;
    std::vector<std::vector<double>> pose, sb, imu, proj, line, pblock, j0rows;
    std::vector<double> ex, lam, linecam, r0, noise, g;
    std::vector<int> imu_i, imu_j, proj_i, proj_j, proj_l, line_f;
    int pm = 0, pn = 0, pnb = 0;
    double proj_sqrt = 0;
    for (std::string ln; std::getline(in, ln);) {
        std::istringstream is(ln);
        std::string tag; is >> tag;
        if (tag == "WINDOW") is >> nf >> L >> n_imu >> n_proj >> n_line >> has_prior;
        else if (tag == "G") g = nums(is);
        else if (tag == "NOISE") noise = nums(is);
        else if (tag == "PROJ_SQRT_INFO") is >> proj_sqrt;
        else if (tag == "POSE") pose.push_back(nums(is));
        else if (tag == "SPEEDBIAS") sb.push_back(nums(is));
        else if (tag == "EX") ex = nums(is);
        else if (tag == "LAM") lam = nums(is);
        else if (tag == "IMU") { int i, j; is >> i >> j; imu_i.push_back(i); imu_j.push_back(j); imu.push_back(nums(is)); }
        else if (tag == "PROJ") { int i, j, l; is >> i >> j >> l; proj_i.push_back(i); proj_j.push_back(j); proj_l.push_back(l); proj.push_back(nums(is)); }
        else if (tag == "LINECAM") linecam = nums(is);
        else if (tag == "LINE") { int f; is >> f; line_f.push_back(f); line.push_back(nums(is)); }
        else if (tag == "PRIOR") is >> pm >> pn >> pnb;
        else if (tag == "PBLOCK") pblock.push_back(nums(is));      // kind idx size col x0...
        else if (tag == "J0ROW") j0rows.push_back(nums(is));
        else if (tag == "R0") r0 = nums(is);
    }
    // globals parameters.cpp would read from the yaml (parameters.cpp:11, :88-93; projection_factor.cpp:3)
    G = Eigen::Vector3d(g[0], g[1], g[2]);
    ACC_N = noise[0]; GYR_N = noise[1]; ACC_W = noise[2]; GYR_W = noise[3];
    ESTIMATE_TD = 0;
    ProjectionFactor::sqrt_info = proj_sqrt * Eigen::Matrix2d::Identity();

    // para_Pose / para_SpeedBias / para_Ex_Pose / para_Feature (estimator.h:166-172)
    std::vector<std::array<double, 7>> P(nf);
    std::vector<std::array<double, 9>> S(nf);
    std::array<double, 7> E;
    std::vector<std::array<double, 1>> F(L);
    for (int i = 0; i < nf; i++) { for (int k = 0; k < 7; k++) P[i][k] = pose[i][k]; for (int k = 0; k < 9; k++) S[i][k] = sb[i][k]; }
    for (int k = 0; k < 7; k++) E[k] = ex[k];
    for (int l = 0; l < L; l++) F[l][0] = lam[l];

    ceres::Problem problem;
    ceres::LossFunction *loss_function = new ceres::CauchyLoss(1.0);
    for (int i = 0; i < nf; i++) {                                                   // :1683-1688
        problem.AddParameterBlock(P[i].data(), 7, new PoseLocalParameterization());
        problem.AddParameterBlock(S[i].data(), 9);
    }
    problem.AddParameterBlock(E.data(), 7, new PoseLocalParameterization());        // :1690-1701 (ESTIMATE_EXTRINSIC: block stays free)
    std::vector<std::pair<ceres::CostFunction *, std::vector<double *>>> blocks;   // in the order they are added: the output's factor order

    // prior (:1714-1720): MarginalizationInfo filled field by field, as MarginalizationInfo::getParameterBlocks leaves it (:301-321)
    auto *mi = new MarginalizationInfo();
    std::vector<std::vector<double>> keep_data;
    std::vector<double *> prior_blocks;
    if (has_prior) {
        mi->m = pm; mi->n = pn;
        keep_data.resize(pnb);
        for (int b = 0; b < pnb; b++) {
            const int kind = (int)pblock[b][0], idx = (int)pblock[b][1], size = (int)pblock[b][2], col = (int)pblock[b][3];
            keep_data[b].assign(pblock[b].begin() + 4, pblock[b].begin() + 4 + size);
            mi->keep_block_size.push_back(size); mi->keep_block_idx.push_back(col + pm); mi->keep_block_data.push_back(keep_data[b].data());
            prior_blocks.push_back(kind == 0 ? P[idx].data() : (kind == 1 ? S[idx].data() : E.data()));
        }
        mi->linearized_jacobians = Eigen::MatrixXd(pn, pn);
        for (int r = 0; r < pn; r++) for (int c = 0; c < pn; c++) mi->linearized_jacobians(r, c) = j0rows[r][c];
        mi->linearized_residuals = Eigen::VectorXd(pn);
        for (int r = 0; r < pn; r++) mi->linearized_residuals(r) = r0[r];
        auto *mf = new MarginalizationFactor(mi);
        problem.AddResidualBlock(mf, NULL, prior_blocks);
        blocks.push_back({mf, prior_blocks});
    }
    // IMU factors (:1723-1732): IntegrationBase with the golden pre-integration's fields
    std::vector<std::unique_ptr<IntegrationBase>> pre;
    for (int k = 0; k < n_imu; k++) {
        const std::vector<double> &v = imu[k];      // sum_dt, delta_p 3, delta_q 4 (xyzw), delta_v 3, ba 3, bg 3, jacobian 225, covariance 225
        auto ib = std::make_unique<IntegrationBase>(Eigen::Vector3d::Zero(), Eigen::Vector3d::Zero(), Eigen::Vector3d(v[11], v[12], v[13]), Eigen::Vector3d(v[14], v[15], v[16]));
        ib->sum_dt = v[0];
        ib->delta_p = Eigen::Vector3d(v[1], v[2], v[3]);
        ib->delta_q = Eigen::Quaterniond(v[7], v[4], v[5], v[6]);
        ib->delta_v = Eigen::Vector3d(v[8], v[9], v[10]);
        for (int r = 0; r < 15; r++) for (int c = 0; c < 15; c++) { ib->jacobian(r, c) = v[17 + 15 * r + c]; ib->covariance(r, c) = v[242 + 15 * r + c]; }
        if (ib->sum_dt > 10.0) { pre.push_back(std::move(ib)); continue; }
        auto *f = new IMUFactor(ib.get());
        std::vector<double *> pb{P[imu_i[k]].data(), S[imu_i[k]].data(), P[imu_j[k]].data(), S[imu_j[k]].data()};
        problem.AddResidualBlock(f, NULL, pb);
        blocks.push_back({f, pb});
        pre.push_back(std::move(ib));
    }
    // point factors (:1734-1774)
    for (int k = 0; k < n_proj; k++) {
        const std::vector<double> &v = proj[k];
        auto *f = new ProjectionFactor(Eigen::Vector3d(v[0], v[1], v[2]), Eigen::Vector3d(v[3], v[4], v[5]));
        std::vector<double *> pb{P[proj_i[k]].data(), P[proj_j[k]].data(), E.data(), F[proj_l[k]].data()};
        problem.AddResidualBlock(f, loss_function, pb);
        blocks.push_back({f, pb});
    }
    // line factors (:1776-1848)
    if (n_line > 0) {
        Eigen::Matrix3d K, Ric; Eigen::Vector3d Tic;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { K(r, c) = linecam[3 * r + c]; Ric(r, c) = linecam[9 + 3 * r + c]; }
        Tic = Eigen::Vector3d(linecam[18], linecam[19], linecam[20]);
        for (int k = 0; k < n_line; k++) {
            const std::vector<double> &v = line[k];
            auto *f = new LineProjectionFactor(Eigen::Vector3d(v[0], v[1], v[2]), Eigen::Vector3d(v[3], v[4], v[5]), Eigen::Vector3d(v[6], v[7], v[8]), K, Ric, Tic);
            problem.AddParameterBlock(P[line_f[k]].data(), 7, new PoseLocalParameterization());      // :1837-1838 (re-added with a new parameterisation, as the reference does)
            std::vector<double *> pb{P[line_f[k]].data()};
            problem.AddResidualBlock(f, loss_function, pb);
            blocks.push_back({f, pb});
        }
    }

    std::ofstream out(argv[2]);
    out << std::setprecision(17);
    out << "CERES_VERSION " << CERES_VERSION_STRING << "\n";
    // per-factor residuals and Jacobians at the initial states, straight from CostFunction::Evaluate (no loss, global Jacobians, row-major)
    for (size_t b = 0; b < blocks.size(); b++) {
        ceres::CostFunction *cf = blocks[b].first;
        const int nr = cf->num_residuals();
        const std::vector<int> &sizes = cf->parameter_block_sizes();
        std::vector<double> r(nr);
        std::vector<std::vector<double>> J(sizes.size());
        std::vector<double *> Jp(sizes.size());
        for (size_t k = 0; k < sizes.size(); k++) { J[k].assign((size_t)nr * sizes[k], 0.0); Jp[k] = J[k].data(); }
        cf->Evaluate(blocks[b].second.data(), r.data(), Jp.data());
        out << "FACTOR " << b << " " << nr << " " << sizes.size();
        for (int s : sizes) out << " " << s;
        out << "\nR";
        for (double x : r) out << " " << x;
        out << "\n";
        for (size_t k = 0; k < sizes.size(); k++) { out << "J" << k; for (double x : J[k]) out << " " << x; out << "\n"; }
    }
    ceres::Solver::Options options;                                                  // :1888-1897
    options.linear_solver_type = ceres::SPARSE_SCHUR;
    options.trust_region_strategy_type = ceres::DOGLEG;
    options.max_num_iterations = max_it;
    options.max_solver_time_in_seconds = 1e9;                                        // deterministic: the wall clock never binds
    ceres::Solver::Summary summary;
    ceres::Solve(options, &problem, &summary);
    out << "SUMMARY " << summary.iterations.size() << " " << summary.initial_cost << " " << summary.final_cost << " " << (int)summary.termination_type << "\n";
    for (const auto &it : summary.iterations)
        out << "ITER " << it.iteration << " " << it.cost << " " << it.cost_change << " " << it.step_norm << " " << it.trust_region_radius << " " << it.relative_decrease << " "
            << (int)it.step_is_successful << " " << (int)it.step_is_valid << " " << it.gradient_max_norm << "\n";
    for (int i = 0; i < nf; i++) { out << "POSE"; for (double x : P[i]) out << " " << x; out << "\n"; }
    for (int i = 0; i < nf; i++) { out << "SPEEDBIAS"; for (double x : S[i]) out << " " << x; out << "\n"; }
    out << "EX"; for (double x : E) out << " " << x; out << "\n";
    out << "LAM"; for (int l = 0; l < L; l++) out << " " << F[l][0]; out << "\nEND\n";
    std::cout << summary.BriefReport() << "\n";
    return 0;
}
