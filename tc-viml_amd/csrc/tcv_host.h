// Host-side data model behind the C-ABI of include/tcv.h: the ceres::Problem-shaped graph
// (reference vins_estimator/src/estimator.cpp:1679-1886), the MarginalizationInfo-shaped prior
// (factor/marginalization_factor.h:46-72) and the packer that turns a problem into the
// device-resident plan + data of tcv_packed.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/tcv.h"
#include "tcv_packed.h"

namespace tcv {
// a device allocation shared between its producer (a batch) and the handles that still read it (device-resident priors)
struct DevBlob {
    void *p = nullptr;
    int dev = 0;
    // non-null: recorded behind the commands that fill the blob, by a producer that did not wait for them (tcv_preintegrate_device): a
    // consumer on another stream waits for it (wait_ready), a host reader synchronises it
    hipEvent_t ready = nullptr;
    int wait_ready(hipStream_t consumer) const;      // TCV_OK or TCV_ERR_HIP
    int sync_ready() const;
    ~DevBlob();
};
}  // namespace tcv

struct tcv_prior {
    int m = 0, n = 0;
    std::vector<int> size, idx;       // keep_block_size / keep_block_idx (idx relative to m, i.e. column of J0)
    std::vector<int> xoff;            // offset of every block in x0
    int xsize = 0;                    // doubles of x0 (= keep_block_data, concatenated), whether or not it is on the host
    mutable std::vector<double> x0;   // keep_block_data, concatenated
    mutable std::vector<double> J0, r0;   // linearized_jacobians (n x n column-major), linearized_residuals
    std::vector<double *> addr;       // addresses of the kept blocks at marginalisation time (un-shifted)
    std::vector<double> As, bs;       // Schur system A', b' the factors were taken from (parity/debug, may be empty)
    // device-resident form (tcv_batch_get_priors_device): J0 | r0 | the marginalisation problem's state vector stay in the producing
    // batch's result buffer; `host` says whether x0 / J0 / r0 above have been materialised (tcv_prior_host)
    std::shared_ptr<tcv::DevBlob> dev;
    const double *d_block = nullptr;  // this window's result block in that buffer (layout: tcv_marg.hip MARG_OUT_*)
    int k0 = 0;                       // leading rows of J0 | r0 that are exact zeros (tcv_packed.h prior_zero_rows, computed on the device)
                                      // -1: not known on the host (tcv_batch_get_priors_device_async: the marginalisation may still be running) --
    const int *d_k0 = nullptr;        //     the consumer reads it on the device, behind the producer's event (dev->ready): [status | k0] of this window
    const int *d_status = nullptr;
    std::vector<int> x_goff;          // per kept block: offset of its values in the result block's state region
    mutable bool host = true;
    mutable std::mutex mu;            // materialisation (several packing threads may ask for it at once)
};
// IntegrationBase on the device (tcv_preintegrate_device): the kernel's output record [delta_p 3 | delta_q 4 | delta_v 3 | ba 3 | bg 3 |
// sum_dt | jacobian 225 | covariance 225] stays in HBM
struct tcv_preint {
    std::shared_ptr<tcv::DevBlob> dev;
    const double *d_out = nullptr;
    double sum_dt = 0.0;
    mutable bool host = false;
    mutable tcv_imu_preintegration pod;
    mutable std::mutex mu;
};
int tcv_preint_host(const tcv_preint *pre);      // materialises `pod` (once); TCV_OK or an error
// the prior's numbers on the host: no-op for a host prior, one device-to-host copy (once) for a device-resident one; TCV_OK or an error
int tcv_prior_host(const tcv_prior *pr);

namespace tcv {

struct ParamBlock {
    double *addr;
    int size, kind;
    bool constant;
};
struct ImuFac { tcv_imu_preintegration pre; int b[4]; const tcv_preint *dev = nullptr; };   // dev: the constants live on the device, `pre` holds sum_dt only
struct ProjFac { double pts[6]; double sqrt_info, loss_a; int b[4]; double aux[8]; int btd; };   // btd >= 0: ProjectionTdFactor on block btd
struct LineFac { double d[9]; double K[9], R[9], T[3]; double loss_a; int b; };
struct PriorFac { const tcv_prior *prior; std::vector<int> b; };

// Int pools of the plans (~170 KB per window) come from a process-wide free list of power-of-two blocks: a malloc of that size is a fresh
// mmap whose pages fault in on first touch (~100 us per plan, as much as building it); a live estimator makes and drops one per frame.
void *plan_block_alloc(size_t bytes);
void plan_block_free(void *p, size_t bytes);
template <class T> struct PlanAlloc {
    typedef T value_type;
    PlanAlloc() = default;
    template <class U> PlanAlloc(const PlanAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(plan_block_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { plan_block_free(p, n * sizeof(T)); }
    template <class U> bool operator==(const PlanAlloc<U> &) const { return true; }
    template <class U> bool operator!=(const PlanAlloc<U> &) const { return false; }
};
typedef std::vector<int, PlanAlloc<int>> PlanInts;

// structural half of a packed window (plan header + int pool + host-side maps): a function of the graph STRUCTURE only, shared by every
// window with that structure (tcv_pack.cpp keeps a process-wide cache keyed by the structure)
struct PlanTemplate {
    PlanHdr hdr;
    PlanInts ints;
    unsigned long long hash = 0;      // plan_content_hash(hdr, ints), computed once by the packing thread that made the template
    std::vector<int> cam_block, cam_loff, lm_block, proj_order;
};

struct Packed {
    PlanHdr hdr;
    std::shared_ptr<const PlanTemplate> tmpl;   // set when the plan came from / went into the cache: `ints` is then empty and tmpl->ints holds the plan
    PlanInts ints;
    std::vector<double> doubles;
    WinHdr win;
    unsigned long long plan_hash = 0;   // of hdr + plan ints (tcv_batch_create: structure de-duplication without copying 200 KB keys)
    bool key_hashed = false;            // plan_hash was set by the packer from the structure key (a plan built for this window alone: tcv_pack.cpp)
    int dev_imu_doubles = 0;         // > 0: every IMU factor of the window is device-resident: n_imu x 287 doubles in the device-only tail (WinHdr::d_imu points there)
    bool prior_k0_deferred = false;  // the prior's zero-row count is read on the device (tcv_prior::k0 < 0): WinHdr::prior_k0 is patched by the splice kernel
    int batch_dev = -1;              // device the batch is created on (tcv_batch_create; -1: no device-resident input is spliced): a prior / pre-integration
                                     // resident on ANOTHER device goes through the host (peer access is never enabled)
    int dev_prior_doubles = 0;       // > 0: the window's prior is device-resident: doubles of its J0 | r0 | x0 region, which lives in the batch's
                                     // device-only tail (not in the uploaded slice) and is filled by the splice kernel of tcv_batch_create
    // host-side maps for download
    std::vector<int> cam_block;      // problem block index of every camera block
    std::vector<int> cam_loff;       // tangent offset of every camera block (-1 constant)
    std::vector<int> lm_block;       // problem block index of every landmark
    std::vector<int> proj_order;     // sorted position -> original projection factor index
};

}  // namespace tcv

namespace tcv {
// address -> block index of a problem: open addressing in one flat array (a node-based map cost a live estimator 200 small allocations and
// frees per window and frame)
struct AddrIndex {
    std::vector<std::pair<double *, int>> tab;      // power-of-two size; first == nullptr: empty
    size_t used = 0;
    static size_t h(const double *a) { size_t x = (size_t)a >> 3; x *= 0x9E3779B97F4A7C15ull; return x >> 17; }
    int find(double *a) const {
        if (!a || tab.empty()) return -1;      // nullptr marks an empty slot: a NULL key must never match one
        const size_t m = tab.size() - 1;
        for (size_t i = h(a) & m;; i = (i + 1) & m) { if (tab[i].first == a) return tab[i].second; if (!tab[i].first) return -1; }
    }
    void put(double *a, int v) {
        if (!a) return;                          // callers reject NULL addresses before they get here (tcv_problem_add_parameter_block)
        if (2 * (used + 1) > tab.size()) grow();
        const size_t m = tab.size() - 1;
        for (size_t i = h(a) & m;; i = (i + 1) & m) { if (tab[i].first == a) { tab[i].second = v; return; } if (!tab[i].first) { tab[i] = {a, v}; used++; return; } }
    }
    void grow() {
        std::vector<std::pair<double *, int>> old;
        old.swap(tab);
        tab.assign(old.empty() ? 512 : 2 * old.size(), {nullptr, 0});
        used = 0;
        for (auto &kv : old) if (kv.first) put(kv.first, kv.second);
    }
};
}  // namespace tcv

struct tcv_problem {
    std::vector<tcv::ParamBlock> blocks;
    tcv::AddrIndex index;
    std::vector<tcv::ImuFac> imu;
    std::vector<tcv::ProjFac> proj;
    std::vector<tcv::LineFac> line;
    std::vector<tcv::PriorFac> prior;
    double G[3] = {0, 0, 9.8};
    double td_TR = 0.0, td_ROW = 1.0;        // rolling-shutter globals of ProjectionTdFactor
    int line_exact = 0;                      // 1: line factors use the exact derivative of their residual (opt-in extension)
    std::vector<int> frame_pose, frame_sb;   // block ids of para_Pose[i] / para_SpeedBias[i] (gauge fix), may be empty
};

struct tcv_batch {
    int n = 0;
    std::vector<tcv_problem *> problems;
    std::vector<tcv::Packed> packed;         // host copies of per-window maps (ints/doubles released after upload)
    std::vector<tcv::PlanHdr> plans;
    std::vector<long long> plan_base;
    std::vector<tcv::WinHdr> wins;
    int state_stride = 0, delta_stride = 0;
    double input_bytes = 0, plan_bytes = 0;
    // device
    void *d_input = nullptr;                 // one allocation: [data pool | window headers | plan headers | plan offsets | plan ints | splice jobs | device-only tail: the regions of device-resident priors]
    tcv::WinHdr *d_win = nullptr;
    tcv::PlanHdr *d_plans = nullptr;
    long long *d_plan_base = nullptr;
    int *d_ipool = nullptr;
    void *d_zero = nullptr;                  // one allocation, zeroed by one memset: [d_delta | d_scratch | d_summary | d_prof]
    double *d_dpool = nullptr, *d_state = nullptr, *d_delta = nullptr, *d_scratch = nullptr;
    tcv::DevSummary *d_summary = nullptr;
    double *d_prof = nullptr;
    int grid = 0, nthreads = 256;
    size_t lds_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    float solve_ms = 0, marg_ms = 0;
    bool solved = false;
    std::vector<double> h_state;
    bool gauge_fixed = false;
    bool fuse_gauge = false, gauge_in_solve = false;      // tcv_batch_set_fused_gauge_fix: the fix in the solve kernel's epilogue; whether the LAST solve applied it
    hipStream_t last_stream = nullptr;     // stream of the last asynchronous call
    std::vector<hipStream_t> streams;      // every stream with work of this batch in flight (tcv_batch_synchronize / batch_free wait for all of them; null: the device)
    hipEvent_t ev_dl = nullptr;            // tcv_batch_download_states_begin: recorded behind the enqueued copy of the states
    void *dl_staging = nullptr;            // its pinned buffer, until tcv_batch_download_states_end
    hipEvent_t ev_order = nullptr;         // orders a call on a new stream behind the pending work of the previous one (tcv_batch_enter_stream)
    hipEvent_t ev_inflight = nullptr;      // tcv_batch_get_priors_device_async: the work in flight is tracked by this event from then on, not by the streams it runs
    bool wait_inflight = false;            // on (the batch outlives the call, a stream may go with its host thread); waited for by synchronize / destroy / the next call
    bool pending = false;                  // asynchronous work issued since the last synchronize (batch_free waits before it recycles the buffers)
    bool chain = false;                   // all plans use the chain layout (2 workgroups per CU)
    int chain_lds = 0;                    // LDS doubles per chain-layout workgroup of this batch
    double *d_imublk = nullptr, *d_spill = nullptr;   // chain mode: per-workgroup IMU J'J blocks and factored fronts
    double *d_sqrt_out = nullptr;         // per window 225 doubles: the solve's sqrt_info of the IMU factor the marginalisation needs
    bool sqrt_out_valid = false;          // the last solve wrote it (device-computed sqrt_info)
    int spill_stride = 0;
    int hcl_cap = 0;                      // doubles of the landmark/camera coupling store per workgroup (largest window of the batch)
    // cooperative mode (tcv_packed.h COOP_*): helpers per window group (0: off), groups, scratch slots (= grid without it)
    int coop_h = 0, coop_groups = 0, slots = 0;
    int last_wg = 0;                              // workgroups per window of the last solve (1: single-workgroup kernel)
    int n_cu = 0, coop_claim = 0, coop_dev = 0;   // CUs of the device; CUs this batch's cooperative launch in flight has claimed (tcv_batch_solve)
    int coop_claim_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, coop_rot = 0;   // the claim per XCD; XCD of the launch's first group (tcv_capi.hip coop_admit)
    int coop_exp_chunks = 0, coop_exp_stride = 0;
    int *d_coop_ctl = nullptr;
    double *d_coop_x = nullptr, *d_coop_exp = nullptr;
    // marginalisation
    void *marg = nullptr;                 // tcv_marg.hip state
    void (*marg_free)(tcv_batch *) = nullptr;
};


// every asynchronous entry point of a batch passes through here: a call on another stream than the previous one waits (on the device) for
// the batch's pending work there -- the gauge fix and the marginalisation read what the solve wrote -- and the stream joins the set
// tcv_batch_synchronize waits for
int tcv_batch_enter_stream(tcv_batch *b, void *hip_stream);
int tcv_marg_sqrt_source(const tcv_batch *b, int window);   // index of that factor among the solve problem's IMU factors, -1 none
int tcv_marg_attach(tcv_batch *b, tcv_problem *const *marg_problems, double *const *const *marg_drop, const int *marg_num_drop);
int tcv_marg_run(tcv_batch *b, void *stream);
int tcv_marg_get_prior(tcv_batch *b, int window, tcv_prior **out);
void tcv_marg_elapsed(tcv_batch *b);
int tcv_marg_download(tcv_batch *b, int compact);
int tcv_marg_get_priors_device(tcv_batch *b, tcv_prior **out, int n, bool nowait);
int tcv_marg_status_prefetch(tcv_batch *b, void *stream);
bool tcv_marg_has_problem(const tcv_batch *b, int window);      // false: marg_problems[window] was NULL
int tcv_marg_layout_n(const tcv_batch *b, int window);          // n of the prior this window's marginalisation makes (known from the attached problem, before the kernel runs); -1: none
// copies device-resident priors into a batch's data pool (one job per window that holds one): launched by tcv_batch_create on the stream
// of its upload, behind it
namespace tcv {
// kind 0: a prior (src = the result block of its window, tcv_marg.hip MARG_OUT_*); kind 1: an IMU factor's constants (src = the
// pre-integration kernel's output record; n, k0, nblk unused)
struct PriorSplice {
    const double *src; long long dst; int n, k0, nblk, kind; int goff[32], size[32];
    const int *k0_src;      // non-null: k0 is read here, on the device ([status | k0] of the producing marginalisation), and written to the window header `win`
    int win, pad;
};
enum { PRIOR_SPLICE_MAX_BLOCKS = 32 };
int launch_prior_splice(const PriorSplice *d_jobs, int njobs, double *d_dpool, void *d_win_headers, hipStream_t st);
}  // namespace tcv

namespace tcv {
// device allocations go through a per-process free list (size-bucketed, per device): a per-frame estimator creates and destroys a
// batch every frame, and hipMalloc / hipFree (each a device synchronisation) were a quarter of its host time
hipError_t dev_malloc(void **p, size_t bytes);
hipError_t dev_free(void *p);
int hip_fail(hipError_t e, const char *what);
int device_ready();
void set_error(const std::string &s);
// returns TCV_OK or a negative status; fills out.  imu_sqrt: optional host-provided sqrt_info (n_imu x 225).
// mode: 0 = chain layout when the graph allows it (speed-bias blocks form chains), else dense; 1 = dense layout
// chain_lds: LDS doubles of a chain-layout workgroup (0: chain_lds_doubles(), half a CU's LDS so that two workgroups share a CU)
// plan_only: plan + the size of the data half (out.win.n_doubles); the data is then written by pack_problem_data into a buffer of the
// caller (one upload buffer per batch)
// coop_chunks > 0: plan for the cooperative (small-batch) kernel with at least that many visual chunks (tcv_packed.h COOP_*)
int pack_problem(const tcv_problem &p, Packed &out, const double *imu_sqrt, int mode = 0, int chain_lds = 0, bool plan_only = false, int coop_chunks = 0);
int pack_problem_data(const tcv_problem &p, Packed &out, const double *imu_sqrt, double *dst);
// pinned host staging buffers for uploads / downloads, recycled through a small per-process pool (hipHostMalloc costs milliseconds)
void *host_staging_acquire(size_t bytes);
void host_staging_release(void *p);
// buffers of commands left in flight on the calling thread's stream `st`: released at the thread's next wait on it (tcv_capi.hip)
void defer_release(void *host_staging, void *dev_buf, hipStream_t st);
void flush_deferred(hipStream_t st);
// a non-blocking stream of the calling thread on its current device, created on first use and kept: the one-shot entry points
// (pre-integration, line association, gauge fix) launch there and wait for THAT stream, never for the device -- another host thread's
// batches keep running (several estimator groups per GPU, bench.py --mode replay)
hipStream_t util_stream();
// plan-cache statistics (hits, misses, entries); tcv_pack.cpp
void plan_cache_stats(long long *hits, long long *misses, long long *entries);
void set_pack_reference(int on);
void pack_laps_print();      // TCV_DEBUG_PACK2=1: per-phase times of the packer since the last call, on stderr
bool pack_reference();
void cam_cache_stats(long long *hits, long long *misses);      // the camera halves of the plans (tcv_pack.cpp)
unsigned long long plan_content_hash(const PlanHdr &hdr, const PlanInts &ints);
}  // namespace tcv
