// Host-side data model behind the C-ABI of include/tcv.h: the ceres::Problem-shaped graph
// (reference vins_estimator/src/estimator.cpp:1679-1886), the MarginalizationInfo-shaped prior
// (factor/marginalization_factor.h:46-72) and the packer that turns a problem into the
// device-resident plan + data of tcv_packed.h.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tcv.h"
#include "tcv_packed.h"

struct tcv_prior {
    int m = 0, n = 0;
    std::vector<int> size, idx;       // keep_block_size / keep_block_idx (idx relative to m, i.e. column of J0)
    std::vector<int> xoff;            // offset of every block in x0
    std::vector<double> x0;           // keep_block_data, concatenated
    std::vector<double> J0, r0;       // linearized_jacobians (n x n column-major), linearized_residuals
    std::vector<double *> addr;       // addresses of the kept blocks at marginalisation time (un-shifted)
};

namespace tcv {

struct ParamBlock {
    double *addr;
    int size, kind;
    bool constant;
};
struct ImuFac { tcv_imu_preintegration pre; int b[4]; };
struct ProjFac { double pts[6]; double sqrt_info, loss_a; int b[4]; };
struct LineFac { double d[9]; double K[9], R[9], T[3]; double loss_a; int b; };
struct PriorFac { const tcv_prior *prior; std::vector<int> b; };

struct Packed {
    PlanHdr hdr;
    std::vector<int> ints;
    std::vector<double> doubles;
    WinHdr win;
    // host-side maps for download
    std::vector<int> cam_block;      // problem block index of every camera block
    std::vector<int> lm_block;       // problem block index of every landmark
    std::vector<int> proj_order;     // sorted position -> original projection factor index
};

}  // namespace tcv

struct tcv_problem {
    std::vector<tcv::ParamBlock> blocks;
    std::unordered_map<double *, int> index;
    std::vector<tcv::ImuFac> imu;
    std::vector<tcv::ProjFac> proj;
    std::vector<tcv::LineFac> line;
    std::vector<tcv::PriorFac> prior;
    double G[3] = {0, 0, 9.8};
};

namespace tcv {
void set_error(const std::string &s);
// returns TCV_OK or a negative status; fills out.  imu_sqrt: optional host-provided sqrt_info (n_imu x 225).
int pack_problem(const tcv_problem &p, Packed &out, const double *imu_sqrt);
}  // namespace tcv
