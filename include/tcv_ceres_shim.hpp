// tcv_ceres_shim.hpp -- header-only C++ veneer over the C-ABI of tcv.h with the class shapes that
// vins_estimator/src/estimator.cpp uses in Estimator::OptimizationWithLine() (reference :1677-2119), so that the function can be
// retargeted by swapping includes and a namespace instead of rewriting its body:
//
//     ceres::Problem / Solver::Options / Solver::Summary / Solve / CauchyLoss        ->  tcvshim::...
//     IMUFactor, ProjectionFactor, LineProjectionFactor, MarginalizationFactor        ->  tcvshim::... (value holders, no Evaluate)
//     PoseLocalParameterization                                                       ->  tcvshim::PoseLocalParameterization (a tag)
//     MarginalizationInfo, ResidualBlockInfo                                          ->  tcvshim::...
//
// The factor set is closed (the solver's kernels implement exactly these factors), so `AddResidualBlock` is overloaded per factor
// type instead of taking a generic CostFunction*.  Vector / matrix arguments are accepted from anything with `.data()` (Eigen
// vectors / matrices in the reference) or from raw pointers; 3x3 matrices are expected ROW-major in memory unless the
// `from_column_major` helper is used (Eigen's default is column-major).  Ownership follows Ceres: the Problem deletes the factors,
// losses and parameterisations handed to it (estimator.cpp:2119), MarginalizationInfo deletes its ResidualBlockInfos and their
// factors but not the loss (marginalization_factor.cpp:71-87).
#ifndef TCV_CERES_SHIM_HPP
#define TCV_CERES_SHIM_HPP

#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "tcv.h"

namespace tcvshim {

namespace detail {
template <class T> auto ptr(const T &v) -> decltype(v.data()) { return v.data(); }
inline const double *ptr(const double *p) { return p; }
inline void check(int rc, const char *what) {
    if (rc != TCV_OK) throw std::runtime_error(std::string(what) + ": " + tcv_last_error());
}
}  // namespace detail

// Eigen::Matrix3d is column-major: transposes a 3x3 into the row-major layout the C-ABI expects
inline void from_column_major(const double *cm, double rm[9]) {
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) rm[3 * r + c] = cm[3 * c + r];
}

struct LossFunction { virtual ~LossFunction() {} virtual double scale() const = 0; };
struct CauchyLoss : LossFunction {                                        // ceres::CauchyLoss(1.0), estimator.cpp:1682
    explicit CauchyLoss(double a) : a_(a) {}
    double scale() const override { return a_; }
    double a_;
};
struct LocalParameterization { virtual ~LocalParameterization() {} };
struct PoseLocalParameterization : LocalParameterization {};              // pose_local_parameterization.h:7-13

struct CostFunction { virtual ~CostFunction() {} };
struct IMUFactor : CostFunction {                                         // imu_factor.h:13-16: holds the IntegrationBase fields it reads
    explicit IMUFactor(const tcv_imu_preintegration &p) : pre(p) {}
    tcv_imu_preintegration pre;
};
struct ProjectionFactor : CostFunction {                                  // projection_factor.h:10-23
    template <class A, class B> ProjectionFactor(const A &pts_i_, const B &pts_j_) {
        std::memcpy(pts_i, detail::ptr(pts_i_), 24); std::memcpy(pts_j, detail::ptr(pts_j_), 24);
    }
    double pts_i[3], pts_j[3];
    static double &sqrt_info() { static double s = 460.0 / 1.5; return s; }      // ProjectionFactor::sqrt_info = FOCAL_LENGTH / 1.5 * I (estimator.cpp:48)
};
struct ProjectionTdFactor : CostFunction {                                // projection_td_factor.h:11-30 (ESTIMATE_TD)
    template <class A, class B, class Cc, class D>
    ProjectionTdFactor(const A &pts_i_, const B &pts_j_, const Cc &velocity_i_, const D &velocity_j_, double td_i_, double td_j_, double row_i_, double row_j_)
        : td_i(td_i_), td_j(td_j_), row_i(row_i_), row_j(row_j_) {
        std::memcpy(pts_i, detail::ptr(pts_i_), 24); std::memcpy(pts_j, detail::ptr(pts_j_), 24);
        std::memcpy(velocity_i, detail::ptr(velocity_i_), 16); std::memcpy(velocity_j, detail::ptr(velocity_j_), 16);
    }
    double pts_i[3], pts_j[3], velocity_i[2], velocity_j[2], td_i, td_j, row_i, row_j;
    static double &TR() { static double v = 0.0; return v; }                     // the globals TR and ROW (parameters.cpp)
    static double &ROW() { static double v = 1.0; return v; }
};
struct LineProjectionFactor : CostFunction {                              // line_projection_factor.cpp:6-17; K, b_c_R row-major
    template <class A, class B, class Cc, class D, class E, class F>
    LineProjectionFactor(const A &ps, const B &pe, const Cc &abc, const D &K_, const E &R_, const F &T_) {
        std::memcpy(pts_start, detail::ptr(ps), 24); std::memcpy(pts_end, detail::ptr(pe), 24); std::memcpy(line, detail::ptr(abc), 24);
        std::memcpy(K, detail::ptr(K_), 72); std::memcpy(R, detail::ptr(R_), 72); std::memcpy(T, detail::ptr(T_), 24);
    }
    double pts_start[3], pts_end[3], line[3], K[9], R[9], T[3];
};

class MarginalizationInfo;
struct MarginalizationFactor : CostFunction {                             // marginalization_factor.h:75-82
    explicit MarginalizationFactor(MarginalizationInfo *info) : marginalization_info(info) {}
    MarginalizationInfo *marginalization_info;
};

class Problem {                                                           // ceres::Problem as used in estimator.cpp:1679-1886
  public:
    Problem() { detail::check(tcv_problem_create(&p_), "tcv_problem_create"); }
    ~Problem() {
        for (auto *c : owned_cost_) delete c;
        for (auto *l : owned_param_) delete l;
        for (auto *l : owned_loss_) delete l;
        tcv_problem_destroy(p_);
    }
    Problem(const Problem &) = delete;
    Problem &operator=(const Problem &) = delete;
    void SetGravity(const double G[3]) { detail::check(tcv_problem_set_gravity(p_, G), "set_gravity"); }      // the reference's global G (parameters.cpp:11)
    void AddParameterBlock(double *values, int size, LocalParameterization *lp = nullptr) {
        detail::check(tcv_problem_add_parameter_block(p_, values, size, lp ? TCV_PARAM_POSE : TCV_PARAM_EUCLIDEAN), "AddParameterBlock");
        if (lp) own(owned_param_, lp);
    }
    void SetParameterBlockConstant(double *values) { detail::check(tcv_problem_set_parameter_block_constant(p_, values), "SetParameterBlockConstant"); }
    void AddResidualBlock(IMUFactor *f, LossFunction *, double *pose_i, double *sb_i, double *pose_j, double *sb_j) {
        detail::check(tcv_problem_add_imu_factor(p_, &f->pre, pose_i, sb_i, pose_j, sb_j), "AddResidualBlock(IMUFactor)");
        own(owned_cost_, f);
    }
    void AddResidualBlock(ProjectionFactor *f, LossFunction *loss, double *pose_i, double *pose_j, double *ex, double *inv_depth) {
        detail::check(tcv_problem_add_projection_factor(p_, f->pts_i, f->pts_j, ProjectionFactor::sqrt_info(), loss ? loss->scale() : 0.0, pose_i, pose_j, ex,
                                                        inv_depth), "AddResidualBlock(ProjectionFactor)");
        own(owned_cost_, f); if (loss) own(owned_loss_, loss);
    }
    void AddResidualBlock(ProjectionTdFactor *f, LossFunction *loss, double *pose_i, double *pose_j, double *ex, double *inv_depth, double *td) {
        detail::check(tcv_problem_set_rolling_shutter(p_, ProjectionTdFactor::TR(), ProjectionTdFactor::ROW()), "tcv_problem_set_rolling_shutter");
        detail::check(tcv_problem_add_projection_td_factor(p_, f->pts_i, f->pts_j, f->velocity_i, f->velocity_j, f->td_i, f->td_j, f->row_i, f->row_j,
                                                           ProjectionFactor::sqrt_info(), loss ? loss->scale() : 0.0, pose_i, pose_j, ex, inv_depth, td),
                      "AddResidualBlock(ProjectionTdFactor)");
        own(owned_cost_, f); if (loss) own(owned_loss_, loss);
    }
    void AddResidualBlock(LineProjectionFactor *f, LossFunction *loss, double *pose) {
        detail::check(tcv_problem_add_line_factor(p_, f->pts_start, f->pts_end, f->line, f->K, f->R, f->T, loss ? loss->scale() : 0.0, pose),
                      "AddResidualBlock(LineProjectionFactor)");
        own(owned_cost_, f); if (loss) own(owned_loss_, loss);
    }
    inline void AddResidualBlock(MarginalizationFactor *f, LossFunction *, const std::vector<double *> &blocks);
    tcv_problem *handle() { return p_; }

  private:
    template <class T, class U> static void own(std::vector<T *> &v, U *x) { T *b = x; for (auto *y : v) if (y == b) return; v.push_back(b); }
    tcv_problem *p_ = nullptr;
    std::vector<CostFunction *> owned_cost_;
    std::vector<LocalParameterization *> owned_param_;
    std::vector<LossFunction *> owned_loss_;
};

enum LinearSolverType { DENSE_SCHUR, SPARSE_SCHUR };
enum TrustRegionStrategyType { LEVENBERG_MARQUARDT, DOGLEG };
struct Solver {
    struct Options {                                                       // the fields estimator.cpp:1888-1897 sets
        LinearSolverType linear_solver_type = SPARSE_SCHUR;
        TrustRegionStrategyType trust_region_strategy_type = DOGLEG;
        int max_num_iterations = 8;
        double max_solver_time_in_seconds = 0.0;
    };
    struct Summary {
        struct IterationSummary { double cost; };
        std::vector<IterationSummary> iterations;                          // estimator.cpp:1902 reads iterations.size()
        double initial_cost = 0, final_cost = 0;
        int termination_type = 0;
        tcv_solver_summary raw;
    };
};
inline void Solve(const Solver::Options &o, Problem *problem, Solver::Summary *summary) {
    if (o.trust_region_strategy_type != DOGLEG) throw std::runtime_error("tcvshim::Solve: only the DOGLEG strategy of the reference is implemented");
    tcv_solver_options opt;
    tcv_solver_options_default(&opt);
    opt.max_num_iterations = o.max_num_iterations;
    opt.max_solver_time_in_seconds = o.max_solver_time_in_seconds;
    tcv_solver_summary s;
    const int rc = tcv_solve(&opt, problem->handle(), &s);
    if (rc != TCV_OK && rc != TCV_ERR_NUMERIC) detail::check(rc, "tcv_solve");      // Ceres failures are silent in the reference (:1900-1903)
    if (summary) {
        summary->raw = s; summary->initial_cost = s.initial_cost; summary->final_cost = s.final_cost; summary->termination_type = s.termination;
        summary->iterations.clear();
        for (int i = 0; i < s.num_iterations && i < TCV_MAX_TRACE; i++) summary->iterations.push_back({s.cost[i]});
    }
}

struct ResidualBlockInfo {                                                 // marginalization_factor.h:15-35
    ResidualBlockInfo(CostFunction *c, LossFunction *l, std::vector<double *> blocks, std::vector<int> drop)
        : cost_function(c), loss_function(l), parameter_blocks(std::move(blocks)), drop_set(std::move(drop)) {}
    CostFunction *cost_function;
    LossFunction *loss_function;
    std::vector<double *> parameter_blocks;
    std::vector<int> drop_set;
};

class MarginalizationInfo {                                                // marginalization_factor.h:46-72
  public:
    MarginalizationInfo() {}
    ~MarginalizationInfo() {
        for (auto *r : factors) { delete r->cost_function; delete r; }
        if (prior_) tcv_prior_destroy(prior_);
    }
    MarginalizationInfo(const MarginalizationInfo &) = delete;
    MarginalizationInfo &operator=(const MarginalizationInfo &) = delete;
    void SetGravity(const double G[3]) { std::memcpy(G_, G, 24); }
    void addResidualBlockInfo(ResidualBlockInfo *r) { factors.push_back(r); }      // :89-108
    void preMarginalize() {}                                                       // :110-129: evaluation happens on the device in marginalize()
    inline void marginalize();                                                     // :174-299
    std::vector<double *> getParameterBlocks(std::unordered_map<long, double *> &addr_shift) {      // :301-321
        int m_, n_, nb, xs;
        detail::check(tcv_prior_dims(prior_, &m_, &n_, &nb, &xs), "tcv_prior_dims");
        std::vector<double *> keep(nb);
        detail::check(tcv_prior_keep_block_addresses(prior_, keep.data()), "tcv_prior_keep_block_addresses");
        for (auto &a : keep) a = addr_shift[reinterpret_cast<long>(a)];
        return keep;
    }
    const tcv_prior *prior() const { return prior_; }
    int m = 0, n = 0;
    std::vector<ResidualBlockInfo *> factors;

  private:
    tcv_prior *prior_ = nullptr;
    double G_[3] = {0, 0, 9.8};
};

inline void Problem::AddResidualBlock(MarginalizationFactor *f, LossFunction *, const std::vector<double *> &blocks) {
    detail::check(tcv_problem_add_marginalization_factor(p_, f->marginalization_info->prior(), blocks.data(), (int)blocks.size()),
                  "AddResidualBlock(MarginalizationFactor)");
    own(owned_cost_, f);
}

inline void MarginalizationInfo::marginalize() {
    Problem mp;
    mp.SetGravity(G_);
    std::vector<double *> drop;
    auto add_drop = [&](double *a) { for (auto *d : drop) if (d == a) return; drop.push_back(a); };
    for (auto *r : factors) {
        const auto &b = r->parameter_blocks;
        if (auto *imu = dynamic_cast<IMUFactor *>(r->cost_function)) {
            detail::check(tcv_problem_add_parameter_block(mp.handle(), b[0], 7, TCV_PARAM_POSE), "marg pose");
            detail::check(tcv_problem_add_parameter_block(mp.handle(), b[2], 7, TCV_PARAM_POSE), "marg pose");
            detail::check(tcv_problem_add_imu_factor(mp.handle(), &imu->pre, b[0], b[1], b[2], b[3]), "marg imu");
        } else if (auto *pf = dynamic_cast<ProjectionFactor *>(r->cost_function)) {
            for (int k = 0; k < 3; k++) detail::check(tcv_problem_add_parameter_block(mp.handle(), b[k], 7, TCV_PARAM_POSE), "marg pose");
            detail::check(tcv_problem_add_projection_factor(mp.handle(), pf->pts_i, pf->pts_j, ProjectionFactor::sqrt_info(),
                                                            r->loss_function ? r->loss_function->scale() : 0.0, b[0], b[1], b[2], b[3]), "marg projection");
        } else if (auto *tf = dynamic_cast<ProjectionTdFactor *>(r->cost_function)) {
            for (int k = 0; k < 3; k++) detail::check(tcv_problem_add_parameter_block(mp.handle(), b[k], 7, TCV_PARAM_POSE), "marg pose");
            detail::check(tcv_problem_set_rolling_shutter(mp.handle(), ProjectionTdFactor::TR(), ProjectionTdFactor::ROW()), "marg rolling shutter");
            detail::check(tcv_problem_add_projection_td_factor(mp.handle(), tf->pts_i, tf->pts_j, tf->velocity_i, tf->velocity_j, tf->td_i, tf->td_j, tf->row_i,
                                                               tf->row_j, ProjectionFactor::sqrt_info(), r->loss_function ? r->loss_function->scale() : 0.0,
                                                               b[0], b[1], b[2], b[3], b[4]), "marg projection td");
        } else if (auto *mf = dynamic_cast<MarginalizationFactor *>(r->cost_function)) {
            const tcv_prior *pr = mf->marginalization_info->prior();
            int m_, n_, nb, xs;
            detail::check(tcv_prior_dims(pr, &m_, &n_, &nb, &xs), "tcv_prior_dims");
            std::vector<int> size(nb), idx(nb);
            detail::check(tcv_prior_export(pr, size.data(), idx.data(), nullptr, nullptr, nullptr), "tcv_prior_export");
            for (int k = 0; k < nb; k++)
                detail::check(tcv_problem_add_parameter_block(mp.handle(), b[k], size[k], size[k] == 7 ? TCV_PARAM_POSE : TCV_PARAM_EUCLIDEAN), "marg prior block");
            detail::check(tcv_problem_add_marginalization_factor(mp.handle(), pr, b.data(), nb), "marg prior");
        } else throw std::runtime_error("MarginalizationInfo: unsupported factor type (line factors are not marginalised, estimator.cpp:1992)");
        for (int d : r->drop_set) add_drop(b[d]);
    }
    if (prior_) { tcv_prior_destroy(prior_); prior_ = nullptr; }
    detail::check(tcv_marginalize(mp.handle(), drop.data(), (int)drop.size(), &prior_), "tcv_marginalize");
    int nb, xs;
    detail::check(tcv_prior_dims(prior_, &m, &n, &nb, &xs), "tcv_prior_dims");
}

}  // namespace tcvshim
#endif
