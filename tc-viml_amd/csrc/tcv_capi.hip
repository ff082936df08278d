// C-ABI of include/tcv.h: ceres::Problem-shaped graph building, ceres::Solve, MarginalizationInfo,
// the device-resident batch mode and the batched factor-evaluation (parity) surface.
// There is no CPU fallback anywhere in this file: without a HIP device every compute entry point
// returns TCV_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#if defined(__linux__)
#include <sys/syscall.h>
#include <unistd.h>
#endif

#include "tcv_factors.h"
#include "tcv_host.h"

extern "C" int tcv_launch_solve(const tcv::SolveArgs *args, int grid, int nthreads, size_t lds_bytes, void *stream);
extern "C" int tcv_solve_scratch_doubles(void);
static int g_solver_variant = 0;   // 0: chain layout when the graph allows it, 1: always the dense 171-dim layout
// CUs claimed by cooperative launches in flight, per device AND per XCD: a cooperative kernel spins on its partners, so all cooperative
// grids in flight together must be resident (two half-resident cooperative kernels could wait for each other's CUs until their
// timeouts).  The members of a group share an XCD -- workgroup b of a launch is dispatched to XCD b % 8 and solve_kernel makes the
// members of group g the workgroups with b % 8 == (g + rot) % 8 -- so the budget that counts is the XCD's (n_cu / 8 CUs), not the
// chip's: five one-window batches of eight workgroups each, all starting at XCD 0, would pass a chip-wide count (40 of 256) and
// sit half-resident on that XCD's 32 CUs.  A launch therefore (1) rotates its groups so that its first group lands on the XCD with the
// fewest claimed CUs (SolveArgs::coop_rot) and (2) is admitted only if every XCD keeps its claims within its CUs.
static std::mutex g_coop_mu;
static int g_coop_claimed[64][8];
static void coop_release(tcv_batch *b) {
    if (b->coop_claim > 0) {
        std::lock_guard<std::mutex> g(g_coop_mu);
        for (int x = 0; x < 8; x++) g_coop_claimed[b->coop_dev & 63][x] -= b->coop_claim_xcd[x];
        b->coop_claim = 0;
    }
}
// admission of a cooperative launch of `groups` groups of `wg` workgroups each: fills b->coop_claim_xcd / coop_rot and returns true, or
// leaves the table untouched and returns false (the caller then runs the same plan on one workgroup per window: same bits)
static bool coop_admit(tcv_batch *b, int groups, int wg) {
    std::lock_guard<std::mutex> g(g_coop_mu);
    int *cl = g_coop_claimed[b->coop_dev & 63];
    const int per_xcd = std::max(1, b->n_cu / 8);
    int rot = 0;
    for (int x = 1; x < 8; x++) if (cl[x] < cl[rot]) rot = x;
    int need[8];
    for (int x = 0; x < 8; x++) need[x] = 0;
    for (int q = 0; q < 8; q++) need[(q + rot) & 7] = wg * ((groups + 7 - q) / 8);      // groups g = q, q + 8, ... land on XCD (q + rot) % 8
    for (int x = 0; x < 8; x++) if (cl[x] + need[x] > per_xcd) return false;
    int total = 0;
    for (int x = 0; x < 8; x++) { cl[x] += need[x]; b->coop_claim_xcd[x] = need[x]; total += need[x]; }
    b->coop_claim = total; b->coop_rot = rot;
    return true;
}
static int g_coop_helpers = -1;    // cooperative mode of small batches: -1 automatic, 0 off, h >= 1: h helper workgroups per window (when the batch allows it)
extern "C" int tcv_launch_marg(const void *args, int grid, size_t lds_bytes, void *stream);

namespace tcv {
static thread_local std::string g_err;
void set_error(const std::string &s) { g_err = s; }
int hip_fail(hipError_t e, const char *what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return TCV_ERR_HIP;
}
#define HIPCHK(x)                                         \
    do {                                                  \
        hipError_t e_ = (x);                              \
        if (e_ != hipSuccess) return hip_fail(e_, #x);    \
    } while (0)

// ---- device memory free list -------------------------------------------------------------------------------------------------
namespace {
struct DevPool {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> free_list;      // (device, bucket bytes) -> buffer
    std::map<void *, std::pair<int, size_t>> live;               // buffer -> (device, bucket bytes)
    size_t cached = 0;
};
DevPool &pool() { static DevPool *p = new DevPool(); return *p; }      // leaked on purpose: no hipFree at process exit after the runtime is gone
size_t bucket_of(size_t n) {
    if (n <= 256) return 256;
    if (n > (size_t(2) << 20)) return (n + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1);      // 2 MiB granules above 2 MiB
    size_t b = 256;
    while (b < n) b <<= 1;
    return b;
}
constexpr size_t POOL_MAX_CACHED = size_t(3) << 30;
}  // namespace
static std::atomic<long long> g_dev_misses{0};
long long dev_pool_misses() { return g_dev_misses.load(); }
hipError_t dev_malloc(void **p, size_t bytes) {
    static const bool off = getenv("TCV_NO_DEV_POOL") != nullptr;
    if (off) return ::hipMalloc(p, bytes);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t b = bucket_of(bytes);
    DevPool &P = pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_list.find({dev, b});
        if (it != P.free_list.end()) { *p = it->second; P.free_list.erase(it); P.cached -= b; P.live[*p] = {dev, b}; return hipSuccess; }
    }
    g_dev_misses++;
    hipError_t e = ::hipMalloc(p, b);
    if (e != hipSuccess) {      // out of memory: give the cached buffers back and retry once
        std::vector<void *> drop;
        { std::lock_guard<std::mutex> g(P.mu); for (auto &kv : P.free_list) drop.push_back(kv.second); P.free_list.clear(); P.cached = 0; }
        for (void *q : drop) (void)::hipFree(q);
        e = ::hipMalloc(p, b);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.live[*p] = {dev, b};
    return hipSuccess;
}
hipError_t dev_free(void *p) {
    if (!p) return hipSuccess;
    DevPool &P = pool();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            const auto key = it->second;
            P.live.erase(it);
            if (P.cached + key.second <= POOL_MAX_CACHED) { P.free_list.insert({key, p}); P.cached += key.second; return hipSuccess; }
        }
    }
    return ::hipFree(p);
}

void dev_pool_stats(unsigned long long *live_bytes, unsigned long long *cached_bytes, int *live_buffers) {
    DevPool &P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    unsigned long long lb = 0;
    for (auto &kv : P.live) lb += kv.second.second;
    if (live_bytes) *live_bytes = lb;
    if (cached_bytes) *cached_bytes = P.cached;
    if (live_buffers) *live_buffers = (int)P.live.size();
}

// The calling thread's own stream per device.  A host thread that ends gives its streams back: the runtime maps streams onto a handful of
// hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and a stream that is never destroyed keeps its share of a queue -- after a few
// generations of short-lived host threads two LIVE threads end up on one queue and the small commands of one (a pre-integration, an
// upload) wait behind the other's 2 ms solve kernel (lock-step replay on 2 host threads: 1 500 against 2 200 windows/s).  The main thread's
// streams are left to the runtime's own teardown at process exit.
namespace {
struct Deferred { void *h, *d; hipStream_t st; };
struct ThreadStreams {
    std::map<int, hipStream_t> m;
    std::map<int, hipStream_t> aux;      // a second stream per device for work that runs beside the thread's main sequence (aux_stream)
    std::vector<Deferred> deferred;      // buffers of commands still in flight on one of these streams (defer_release)
    bool main_thread = false;
    ~ThreadStreams();
};
ThreadStreams &thread_streams() { thread_local ThreadStreams mine; return mine; }
}  // namespace
// A call that leaves its commands in flight on the calling thread's stream (tcv_preintegrate_device: nobody on the host needs the result)
// parks the pinned staging buffer and the device input blob here instead of waiting for the stream; they go back to their pools at the
// thread's next wait on that stream (every tcv_batch_create ends with one).
// The list is bounded: a thread that pre-integrates and never creates a batch (or whose frames keep failing before tcv_batch_create) would
// otherwise pin host and device memory until it ends -- at DEFER_MAX parked commands the stream is drained and its buffers released.
enum { DEFER_MAX = 32 };
void flush_deferred(hipStream_t st);
void defer_release(void *host_staging, void *dev_buf, hipStream_t st) {
    std::vector<Deferred> &v = thread_streams().deferred;
    if (v.size() >= DEFER_MAX) { (void)(st ? hipStreamSynchronize(st) : hipDeviceSynchronize()); flush_deferred(st); }
    v.push_back(Deferred{host_staging, dev_buf, st});
}
void flush_deferred(hipStream_t st) {
    std::vector<Deferred> &v = thread_streams().deferred;
    size_t k = 0;
    for (size_t i = 0; i < v.size(); i++) {
        if (v[i].st == st) { host_staging_release(v[i].h); (void)dev_free(v[i].d); }
        else v[k++] = v[i];
    }
    v.resize(k);
}
namespace {
// Streams of host threads that have ended, per device.  A thread's stream is drained and parked here instead of being destroyed: a batch
// that ran on it (TCV_STREAM_THREAD, the native estimator) may still hold the handle in tcv_batch::streams / last_stream, and a
// hipStreamSynchronize / hipEventRecord on a destroyed stream is an error (round-4 advisor finding).  The next new thread takes a parked
// stream over, so the process never holds more streams than it had live threads at once -- which is what keeps two LIVE threads off one
// hardware queue (see above).  Leaked on purpose, like the device pool.
// `orphans`: the buffers of commands a thread left in flight when it ended (defer_release), per stream; whoever takes the stream over waits
// for it once and releases them.  The destructor below makes NO HIP call: it runs among the thread's TLS destructors,
// where a profiler's own thread state may be gone already (rocprofv3 aborted in hipStreamSynchronize there -- "must be non nullptr" -- as soon
// as a replay or the stream mode ran on more than one host thread; round 5).
struct StreamPark { std::mutex mu; std::multimap<int, hipStream_t> idle; std::map<hipStream_t, std::vector<Deferred>> orphans; };
StreamPark &stream_park() { static StreamPark *p = new StreamPark(); return *p; }
hipStream_t take_parked_stream(int dev) {
    StreamPark &P = stream_park();
    hipStream_t st = nullptr;
    std::vector<std::pair<hipStream_t, std::vector<Deferred>>> drain;      // waited for and released OUTSIDE the lock: every ending thread's destructor needs it
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.idle.find(dev);
        if (it != P.idle.end()) {
            st = it->second;
            P.idle.erase(it);
            auto oi = P.orphans.find(st);
            if (oi != P.orphans.end()) { drain.emplace_back(st, std::move(oi->second)); P.orphans.erase(oi); }
        }
        // orphans nobody will ever adopt -- commands a thread left on the NULL stream (its stream creation had failed), or on a stream that is not
        // parked (the ending thread's aux stream was its main stream) -- would keep their pinned staging and device memory for the life of the
        // process: whoever comes by for a stream takes them along
        for (auto oi = P.orphans.begin(); oi != P.orphans.end();) {
            bool parked = false;
            for (auto &kv : P.idle) if (kv.second == oi->first) { parked = true; break; }
            if (!parked) { drain.emplace_back(oi->first, std::move(oi->second)); oi = P.orphans.erase(oi); } else ++oi;
        }
    }
    for (auto &d : drain) {      // what a stream's previous thread left in flight has long finished, as a rule: wait (here a HIP call is fine) and release
        (void)(d.first ? hipStreamSynchronize(d.first) : hipDeviceSynchronize());
        for (auto &x : d.second) { host_staging_release(x.h); (void)dev_free(x.d); }
    }
    return st;
}
ThreadStreams::~ThreadStreams() {
    if (main_thread) return;      // (the main thread's streams and whatever it left in flight go with the runtime's own teardown at process exit)
    StreamPark &P = stream_park();
    std::lock_guard<std::mutex> g(P.mu);
    for (auto &x : deferred) P.orphans[x.st].push_back(x);      // still in flight, possibly: released by the stream's next owner after its first wait
    deferred.clear();
    for (auto &kv : m) if (kv.second) P.idle.emplace(kv.first, kv.second);
    for (auto &kv : aux) if (kv.second) { bool own = true; for (auto &k2 : m) if (k2.second == kv.second) own = false; if (own) P.idle.emplace(kv.first, kv.second); }
}
}  // namespace
// the calling thread's second stream (created at first use): the native estimator runs a frame's marginalisation there, beside the next
// frame's association and upload on the thread's main stream
hipStream_t aux_stream() {
    ThreadStreams &mine = thread_streams();
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto it = mine.aux.find(dev);
    if (it != mine.aux.end()) return it->second;
    (void)util_stream();      // (sets main_thread)
    hipStream_t st = take_parked_stream(dev);
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = util_stream();
    mine.aux.emplace(dev, st);
    return st;
}
namespace { thread_local int g_stream_slot = 0; }
hipStream_t aux_stream();
hipStream_t util_stream() {
    if (g_stream_slot == 1) { const int keep = g_stream_slot; g_stream_slot = 0; hipStream_t a = aux_stream(); g_stream_slot = keep; return a; }      // (aux_stream asks for the main one once, to learn whether this is the main thread)
    ThreadStreams &mine = thread_streams();
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto it = mine.m.find(dev);
    if (it != mine.m.end()) return it->second;
#if defined(__linux__)
    mine.main_thread = (long)syscall(SYS_gettid) == (long)getpid();
#else
    mine.main_thread = true;
#endif
    hipStream_t st = take_parked_stream(dev);
    if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;      // the default stream: slower, still correct
    mine.m.emplace(dev, st);
    return st;
}

int device_ready() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        g_err = "no HIP device visible: libtcv_hip has no CPU fallback";
        return TCV_ERR_NO_DEVICE;
    }
    return TCV_OK;
}
}  // namespace tcv
using namespace tcv;

// =====================================================================================================
// library
// =====================================================================================================
extern "C" const char *tcv_version(void) { return "tcv-hip 0.1 (gfx950)"; }
extern "C" const char *tcv_last_error(void) { return g_err.c_str(); }
extern "C" int tcv_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int tcv_set_solver_variant(int variant) {
    if (variant != 0 && variant != 1) return TCV_ERR_INVALID;
    g_solver_variant = variant;
    return TCV_OK;
}
extern "C" int tcv_set_cooperative(int helpers) {
    if (helpers < -1 || helpers > COOP_MAX_H) { set_error("set_cooperative: -1 (automatic), 0 (off) or 1..7 helper workgroups per window"); return TCV_ERR_INVALID; }
    g_coop_helpers = helpers;
    return TCV_OK;
}
extern "C" int tcv_device_memory_stats(unsigned long long *live_bytes, unsigned long long *cached_bytes, int *live_buffers) {
    tcv::dev_pool_stats(live_bytes, cached_bytes, live_buffers);
    return TCV_OK;
}
extern "C" int tcv_thread_stream_slot(int slot) {
    const int prev = g_stream_slot;
    g_stream_slot = slot == 1 ? 1 : 0;
    return prev;
}
extern "C" int tcv_set_device(int device) {
    if (int rc = device_ready()) return rc;
    HIPCHK(hipSetDevice(device));
    return TCV_OK;
}

// =====================================================================================================
// ceres::Problem surface
// =====================================================================================================
extern "C" int tcv_problem_create(tcv_problem **out) {
    if (!out) return TCV_ERR_INVALID;
    *out = new tcv_problem();
    return TCV_OK;
}
extern "C" void tcv_problem_destroy(tcv_problem *p) { delete p; }

static int block_of(tcv_problem *p, double *addr) {
    return p->index.find(addr);
}

extern "C" int tcv_problem_add_parameter_block(tcv_problem *p, double *values, int size, int parameterization) {
    if (!p || !values || size <= 0) { set_error("add_parameter_block: bad argument"); return TCV_ERR_INVALID; }
    if (parameterization == TCV_PARAM_POSE && size != 7) { set_error("pose parameterisation needs a size-7 block"); return TCV_ERR_INVALID; }
    const int b = block_of(p, values);
    if (b >= 0) {  // Ceres 2.x accepts re-adding a block (estimator.cpp:1837-1838); size must agree
        if (p->blocks[b].size != size) { set_error("parameter block re-added with a different size"); return TCV_ERR_INVALID; }
        p->blocks[b].kind = parameterization == TCV_PARAM_POSE ? KIND_POSE : p->blocks[b].kind;
        return TCV_OK;
    }
    p->index.put(values, (int)p->blocks.size());
    p->blocks.push_back(ParamBlock{values, size, parameterization == TCV_PARAM_POSE ? KIND_POSE : KIND_EUCLID, false});
    return TCV_OK;
}
extern "C" int tcv_problem_set_parameter_block_constant(tcv_problem *p, double *values) {
    const int b = p ? block_of(p, values) : -1;
    if (b < 0) { set_error("set_parameter_block_constant: unknown block"); return TCV_ERR_INVALID; }
    p->blocks[b].constant = true;
    return TCV_OK;
}
extern "C" int tcv_problem_set_gravity(tcv_problem *p, const double G[3]) {
    if (!p || !G) return TCV_ERR_INVALID;
    for (int i = 0; i < 3; i++) p->G[i] = G[i];
    return TCV_OK;
}
// implicit AddParameterBlock like ceres::Problem::AddResidualBlock does for unknown pointers
static int ensure_block(tcv_problem *p, double *addr, int size) {
    int b = block_of(p, addr);
    if (b >= 0) return p->blocks[b].size == size ? b : -1;
    if (tcv_problem_add_parameter_block(p, addr, size, TCV_PARAM_EUCLIDEAN) != TCV_OK) return -1;
    return block_of(p, addr);
}
extern "C" int tcv_problem_add_imu_factor(tcv_problem *p, const tcv_imu_preintegration *pre, double *pose_i, double *sb_i,
                                          double *pose_j, double *sb_j) {
    if (!p || !pre) return TCV_ERR_INVALID;
    ImuFac f;
    f.pre = *pre;
    double *a[4] = {pose_i, sb_i, pose_j, sb_j};
    const int sz[4] = {7, 9, 7, 9};
    for (int k = 0; k < 4; k++) {
        f.b[k] = a[k] ? ensure_block(p, a[k], sz[k]) : -1;
        if (f.b[k] < 0) { set_error("add_imu_factor: bad parameter block"); return TCV_ERR_INVALID; }
    }
    p->imu.push_back(f);
    return TCV_OK;
}
extern "C" int tcv_problem_add_imu_factor_device(tcv_problem *p, const tcv_preint *pre, double *pose_i, double *sb_i, double *pose_j, double *sb_j) {
    if (!p || !pre) return TCV_ERR_INVALID;
    ImuFac f;
    std::memset(&f.pre, 0, sizeof f.pre);
    f.pre.sum_dt = pre->sum_dt;
    f.dev = pre;
    double *a[4] = {pose_i, sb_i, pose_j, sb_j};
    const int sz[4] = {7, 9, 7, 9};
    for (int k = 0; k < 4; k++) {
        f.b[k] = a[k] ? ensure_block(p, a[k], sz[k]) : -1;
        if (f.b[k] < 0) { set_error("add_imu_factor_device: bad parameter block"); return TCV_ERR_INVALID; }
    }
    p->imu.push_back(f);
    return TCV_OK;
}
extern "C" int tcv_problem_add_projection_factor(tcv_problem *p, const double pts_i[3], const double pts_j[3], double sqrt_info,
                                                 double loss_a, double *pose_i, double *pose_j, double *ex_pose,
                                                 double *inv_depth) {
    if (!p || !pts_i || !pts_j) return TCV_ERR_INVALID;
    ProjFac f;
    for (int i = 0; i < 3; i++) { f.pts[i] = pts_i[i]; f.pts[3 + i] = pts_j[i]; }
    f.sqrt_info = sqrt_info; f.loss_a = loss_a;
    double *a[4] = {pose_i, pose_j, ex_pose, inv_depth};
    const int sz[4] = {7, 7, 7, 1};
    for (int k = 0; k < 4; k++) {
        f.b[k] = a[k] ? ensure_block(p, a[k], sz[k]) : -1;
        if (f.b[k] < 0) { set_error("add_projection_factor: bad parameter block"); return TCV_ERR_INVALID; }
    }
    for (int i = 0; i < 8; i++) f.aux[i] = 0.0;
    f.btd = -1;
    p->proj.push_back(f);
    return TCV_OK;
}
extern "C" int tcv_problem_add_projection_td_factor(tcv_problem *p, const double pts_i[3], const double pts_j[3], const double vel_i[2],
                                                    const double vel_j[2], double td_i, double td_j, double row_i, double row_j, double sqrt_info,
                                                    double loss_a, double *pose_i, double *pose_j, double *ex_pose, double *inv_depth, double *td) {
    if (!p || !vel_i || !vel_j || !td) { set_error("add_projection_td_factor: bad argument"); return TCV_ERR_INVALID; }
    const int btd = ensure_block(p, td, 1);
    if (btd < 0) { set_error("add_projection_td_factor: bad td block"); return TCV_ERR_INVALID; }
    const int rc = tcv_problem_add_projection_factor(p, pts_i, pts_j, sqrt_info, loss_a, pose_i, pose_j, ex_pose, inv_depth);
    if (rc != TCV_OK) return rc;
    ProjFac &f = p->proj.back();
    f.aux[0] = vel_i[0]; f.aux[1] = vel_i[1]; f.aux[2] = vel_j[0]; f.aux[3] = vel_j[1];
    f.aux[4] = td_i; f.aux[5] = td_j; f.aux[6] = row_i; f.aux[7] = row_j;
    f.btd = btd;
    return TCV_OK;
}
extern "C" int tcv_problem_set_line_jacobian(tcv_problem *p, int exact) {
    if (!p || (exact != 0 && exact != 1)) { set_error("set_line_jacobian: 0 (reference) or 1 (exact)"); return TCV_ERR_INVALID; }
    p->line_exact = exact;
    return TCV_OK;
}
extern "C" int tcv_problem_set_rolling_shutter(tcv_problem *p, double TR, double ROW) {
    if (!p || !(ROW > 0.0) || !(TR == TR)) { set_error("set_rolling_shutter: ROW must be positive"); return TCV_ERR_INVALID; }
    p->td_TR = TR; p->td_ROW = ROW;
    return TCV_OK;
}
extern "C" int tcv_problem_add_line_factor(tcv_problem *p, const double ps[3], const double pe[3], const double abc[3],
                                           const double K[9], const double R[9], const double T[3], double loss_a, double *pose) {
    if (!p || !ps || !pe || !abc || !K || !R || !T || !pose) return TCV_ERR_INVALID;
    LineFac f;
    for (int i = 0; i < 3; i++) { f.d[i] = ps[i]; f.d[3 + i] = pe[i]; f.d[6 + i] = abc[i]; f.T[i] = T[i]; }
    for (int i = 0; i < 9; i++) { f.K[i] = K[i]; f.R[i] = R[i]; }
    f.loss_a = loss_a;
    f.b = ensure_block(p, pose, 7);
    if (f.b < 0) { set_error("add_line_factor: bad parameter block"); return TCV_ERR_INVALID; }
    p->line.push_back(f);
    return TCV_OK;
}
extern "C" int tcv_problem_add_marginalization_factor(tcv_problem *p, const tcv_prior *prior, double *const *blocks, int n) {
    if (!p || !prior || n != (int)prior->size.size() || (n > 0 && !blocks)) { set_error("add_marginalization_factor: block count mismatch"); return TCV_ERR_INVALID; }
    // the prior of a marginalisation that kept nothing (n = 0): the reference adds a MarginalizationFactor with no residuals over no blocks
    // (estimator.cpp:1714-1720 on the empty MarginalizationInfo of marginalization_factor.cpp:174-194) -- accepted, contributes nothing
    if (prior->n == 0) return TCV_OK;
    PriorFac f;
    f.prior = prior;
    for (int k = 0; k < n; k++) {
        const int b = ensure_block(p, blocks[k], prior->size[k]);
        if (b < 0) { set_error("add_marginalization_factor: bad parameter block"); return TCV_ERR_INVALID; }
        f.b.push_back(b);
    }
    p->prior.push_back(f);
    return TCV_OK;
}
extern "C" int tcv_problem_set_frames(tcv_problem *p, int n_frames, double *const *pose, double *const *speedbias) {
    if (!p || n_frames <= 0 || !pose) { set_error("set_frames: bad argument"); return TCV_ERR_INVALID; }
    std::vector<int> fp(n_frames, -1), fs(n_frames, -1);
    for (int i = 0; i < n_frames; i++) {
        fp[i] = block_of(p, pose[i]);
        if (fp[i] < 0 || p->blocks[fp[i]].size != 7) { set_error("set_frames: unknown pose block"); return TCV_ERR_INVALID; }
        if (speedbias && speedbias[i]) {
            fs[i] = block_of(p, speedbias[i]);
            if (fs[i] < 0 || p->blocks[fs[i]].size != 9) { set_error("set_frames: unknown speed-bias block"); return TCV_ERR_INVALID; }
        }
    }
    p->frame_pose = fp; p->frame_sb = fs;
    return TCV_OK;
}
extern "C" int tcv_problem_num_parameter_blocks(const tcv_problem *p) { return p ? (int)p->blocks.size() : 0; }
extern "C" int tcv_problem_num_residual_blocks(const tcv_problem *p) {
    return p ? (int)(p->imu.size() + p->proj.size() + p->line.size() + p->prior.size()) : 0;
}
extern "C" int tcv_problem_num_residuals(const tcv_problem *p) {
    if (!p) return 0;
    int n = 15 * (int)p->imu.size() + 2 * (int)p->proj.size() + 2 * (int)p->line.size();
    for (auto &f : p->prior) n += f.prior->n;
    return n;
}

// host-only: packs the problem (no device needed) and reports the sizes of its plan and data.
// out[0..15] = nc, nx, npp, nland, nt, n_vis_chunk, n_imu_chunk, n_vunit, n_vitem, n_sunit, n_sitem, n_iunit, n_iitem,
//              plan ints, window doubles, LDS bytes
extern "C" int tcv_problem_plan_stats(const tcv_problem *p, int *out) {
    if (!p || !out) return TCV_ERR_INVALID;
    Packed pk;
    const char *ec = getenv("TCV_PLAN_COOP");      // developer / test switch: the plan the cooperative mode would use with that many helpers
    const int rc = pack_problem(*p, pk, nullptr, g_solver_variant, 0, false, ec ? atoi(ec) : 0);
    if (rc != TCV_OK) return rc;
    const PlanHdr &H = pk.hdr;
    const int v[16] = {H.nc, H.nx, H.npp, H.nland, H.chain ? H.nt_c : H.nt, H.n_vis_chunk, H.n_imu_chunk, H.n_vunit, H.n_vitem, H.n_sunit, H.n_sitem,
                       H.chain ? H.n_e : H.n_iunit, H.n_iitem, H.plan_ints, pk.win.n_doubles,
                       H.chain ? chain_lds_doubles() * 8
                               : (H.nt * (H.nt + 1) / 2 * 256 + 2 * ((H.nx + H.nland + 1) & ~1) + (3 * H.camw + 176) + 64 + H.lds_area) * 8};
    std::memcpy(out, v, sizeof v);
    return TCV_OK;
}

// diagnostics of the packer (tests/test_pack_cpu.py): the plan of a problem as the device would get it -- header (as ints) followed by the
// int pool -- and the switch between the fast gather-program builder with its caches and the generic reference builder
extern "C" int tcv_problem_plan_ints(const tcv_problem *p, int *out, int cap, int *len) {
    if (!p || !len) return TCV_ERR_INVALID;
    Packed pk;
    const char *ec = getenv("TCV_PLAN_COOP");
    const int rc = pack_problem(*p, pk, nullptr, g_solver_variant, 0, true, ec ? atoi(ec) : 0);
    if (rc != TCV_OK) return rc;
    const PlanInts &pints = pk.tmpl ? pk.tmpl->ints : pk.ints;
    const int nh = (int)(sizeof(PlanHdr) / sizeof(int));
    *len = nh + (int)pints.size();
    if (out && cap >= *len) { std::memcpy(out, &pk.hdr, sizeof(PlanHdr)); std::memcpy(out + nh, pints.data(), sizeof(int) * pints.size()); }
    return TCV_OK;
}
extern "C" int tcv_set_packer_reference(int on) { tcv::set_pack_reference(on); return TCV_OK; }
// the packing pass of tcv_batch_create (plans + data sizes of n problems on `threads` host threads of the library's worker pool) without
// a device: seconds of wall time in *seconds
extern "C" int tcv_problems_pack_bench(tcv_problem *const *problems, int n, int threads, int coop_chunks, double *seconds) {
    if (!problems || n <= 0 || !seconds) return TCV_ERR_INVALID;
    // TCV_PACK_BENCH_FRAME = f: the problems are packed f at a time and every group's plans are released before the next group starts --
    // the life cycle of a lock-step frame's batch (the int pools come back from the block pool) instead of n plans alive at once
    int frame = n;
    if (const char *e = getenv("TCV_PACK_BENCH_FRAME")) { const int v = atoi(e); if (v > 0) frame = std::min(n, v); }
    std::vector<int> rcs(n, TCV_OK);
    std::vector<std::string> msgs(n);
    const auto t0 = std::chrono::steady_clock::now();
    for (int b = 0; b < n; b += frame) {
        const int m = std::min(frame, n - b), nth = std::max(1, std::min(threads, m));
        std::vector<Packed> packed(m);
        auto one = [&](int w) {
            rcs[b + w] = pack_problem(*problems[b + w], packed[w], nullptr, g_solver_variant, coop_chunks > 0 ? (int)LDS_DOUBLES : 0, true, coop_chunks);
            if (rcs[b + w] != TCV_OK) msgs[b + w] = tcv_last_error();      // (the text is per thread: a worker's is carried over to the caller below)
        };
        if (getenv("TCV_PACK_BENCH_STRIDED")) tcv::parallel_run(nth, [&](int t) { for (int w = t; w < m; w += nth) one(w); });      // (a fixed share per thread: up to round 5)
        else tcv::parallel_items(m, nth, [&](int w, int) { one(w); });
    }
    *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("TCV_DEBUG_PACK2")) tcv::pack_laps_print();
    for (int w = 0; w < n; w++) if (rcs[w] != TCV_OK) { set_error(msgs[w]); return rcs[w]; }
    return TCV_OK;
}
extern "C" int tcv_plan_cache_stats(long long *out4) {
    if (!out4) return TCV_ERR_INVALID;
    long long e = 0;
    tcv::plan_cache_stats(&out4[0], &out4[1], &e);
    tcv::cam_cache_stats(&out4[2], &out4[3]);
    return TCV_OK;
}

// graph construction of estimator.cpp:1683-1846 from frame-indexed arrays
extern "C" int tcv_problem_from_window(const tcv_window_desc *w, tcv_problem **out) {
    if (!w || !out || w->n_frames <= 0 || !w->para_pose || !w->para_speedbias || !w->para_ex_pose || (w->n_imu > 0 && (!w->imu || !w->imu_frame_i || !w->imu_frame_j)) ||
        (w->n_proj > 0 && (!w->proj_pts || !w->proj_frame_i || !w->proj_frame_j || !w->proj_feature || !w->para_feature)) ||
        (w->n_line > 0 && (!w->line_data || !w->line_frame)) ||
        (w->prior && w->prior->n > 0 && (!w->prior_block_kind || !w->prior_block_index)) || w->n_imu < 0 || w->n_proj < 0 || w->n_line < 0 ||
        (w->para_td && w->n_proj > 0 && !w->proj_td_aux) || (w->para_td && !(w->td_ROW > 0.0))) {
        set_error("problem_from_window: missing array in the window description");
        return TCV_ERR_INVALID;
    }
    auto frame_ok = [&](int f) { return f >= 0 && f < w->n_frames; };
    for (int k = 0; k < w->n_imu; k++) if (!frame_ok(w->imu_frame_i[k]) || !frame_ok(w->imu_frame_j[k])) { set_error("problem_from_window: IMU frame index out of range"); return TCV_ERR_INVALID; }
    for (int k = 0; k < w->n_proj; k++)
        if (!frame_ok(w->proj_frame_i[k]) || !frame_ok(w->proj_frame_j[k]) || w->proj_feature[k] < 0 || w->proj_feature[k] >= w->n_landmarks) { set_error("problem_from_window: projection frame / feature index out of range"); return TCV_ERR_INVALID; }
    for (int k = 0; k < w->n_line; k++) if (!frame_ok(w->line_frame[k])) { set_error("problem_from_window: line frame index out of range"); return TCV_ERR_INVALID; }
    tcv_problem *p = new tcv_problem();
    int rc = TCV_OK;
    auto chk = [&](int r) { if (rc == TCV_OK && r != TCV_OK) rc = r; };
    for (int i = 0; i < w->n_frames; i++) {   // :1683-1688
        chk(tcv_problem_add_parameter_block(p, w->para_pose + 7 * i, 7, TCV_PARAM_POSE));
        chk(tcv_problem_add_parameter_block(p, w->para_speedbias + 9 * i, 9, TCV_PARAM_EUCLIDEAN));
    }
    chk(tcv_problem_add_parameter_block(p, w->para_ex_pose, 7, TCV_PARAM_POSE));   // :1689-1701
    if (!w->estimate_extrinsic) chk(tcv_problem_set_parameter_block_constant(p, w->para_ex_pose));
    chk(tcv_problem_set_gravity(p, w->gravity));
    if (w->line_exact_jacobian) chk(tcv_problem_set_line_jacobian(p, 1));
    if (w->para_td) {   // :1703-1707
        chk(tcv_problem_add_parameter_block(p, w->para_td, 1, TCV_PARAM_EUCLIDEAN));
        chk(tcv_problem_set_rolling_shutter(p, w->td_TR, w->td_ROW));
    }
    if (w->prior) {   // :1714-1720
        int m, n, nb, xs;
        tcv_prior_dims(w->prior, &m, &n, &nb, &xs);
        std::vector<double *> blocks(nb);
        for (int k = 0; k < nb; k++) {
            const int kind = w->prior_block_kind[k], idx = w->prior_block_index[k];
            if (kind < 0 || kind > 3 || (kind < 2 && !frame_ok(idx)) || (kind == 3 && !w->para_td)) { set_error("problem_from_window: prior block kind / index out of range"); rc = TCV_ERR_INVALID; break; }
            blocks[k] = kind == 0 ? w->para_pose + 7 * idx : (kind == 1 ? w->para_speedbias + 9 * idx : (kind == 2 ? w->para_ex_pose : w->para_td));
        }
        if (rc == TCV_OK) chk(tcv_problem_add_marginalization_factor(p, w->prior, blocks.data(), nb));
    }
    for (int k = 0; k < w->n_imu; k++) {   // :1723-1732
        if (w->imu[k].sum_dt > 10.0) continue;
        const int i = w->imu_frame_i[k], j = w->imu_frame_j[k];
        if (w->imu_device && w->imu_device[k])
            chk(tcv_problem_add_imu_factor_device(p, w->imu_device[k], w->para_pose + 7 * i, w->para_speedbias + 9 * i, w->para_pose + 7 * j,
                                                  w->para_speedbias + 9 * j));
        else
        chk(tcv_problem_add_imu_factor(p, w->imu + k, w->para_pose + 7 * i, w->para_speedbias + 9 * i, w->para_pose + 7 * j,
                                       w->para_speedbias + 9 * j));
    }
    // The point and line factors: same result as tcv_problem_add_projection[_td]_factor / tcv_problem_add_line_factor per factor (blocks the
    // factors introduce are added in order of first appearance, like ceres::Problem::AddResidualBlock does for unknown pointers), but the
    // block indices of the frame-indexed arrays are known here -- no address lookup per pointer (four per point factor: a third of a
    // lock-step frame's problem construction)
    if (rc == TCV_OK) {
        const int b_ex = block_of(p, w->para_ex_pose), b_td = w->para_td ? block_of(p, w->para_td) : -1;
        std::vector<int> b_pose(w->n_frames), b_feat(std::max(1, w->n_landmarks), -1);
        for (int i = 0; i < w->n_frames; i++) b_pose[i] = block_of(p, w->para_pose + 7 * i);
        p->proj.reserve((size_t)w->n_proj); p->line.reserve((size_t)w->n_line);
        p->blocks.reserve(p->blocks.size() + (size_t)w->n_landmarks);
        for (int k = 0; k < w->n_proj && rc == TCV_OK; k++) {   // :1737-1771
            const int l = w->proj_feature[k];
            if (b_feat[l] < 0) {
                double *addr = w->para_feature + l;
                b_feat[l] = ensure_block(p, addr, 1);      // (an address that is a block already must be a size-1 block)
                if (b_feat[l] < 0) { set_error("add_projection_factor: bad parameter block"); rc = TCV_ERR_INVALID; break; }
            }
            p->proj.emplace_back();
            ProjFac &f = p->proj.back();
            const double *pt = w->proj_pts + 6 * k;
            for (int i = 0; i < 6; i++) f.pts[i] = pt[i];
            f.sqrt_info = w->proj_sqrt_info; f.loss_a = w->proj_loss_a;
            f.b[0] = b_pose[w->proj_frame_i[k]]; f.b[1] = b_pose[w->proj_frame_j[k]]; f.b[2] = b_ex; f.b[3] = b_feat[l];
            if (w->para_td) { const double *a = w->proj_td_aux + 8 * k; for (int i = 0; i < 8; i++) f.aux[i] = a[i]; f.btd = b_td; }
            else { for (int i = 0; i < 8; i++) f.aux[i] = 0.0; f.btd = -1; }
        }
        for (int k = 0; k < w->n_line && rc == TCV_OK; k++) {   // :1786-1846
            p->line.emplace_back();
            LineFac &f = p->line.back();
            const double *d = w->line_data + 9 * k;
            for (int i = 0; i < 9; i++) { f.d[i] = d[i]; f.K[i] = w->line_K[i]; f.R[i] = w->line_Ric[i]; }
            for (int i = 0; i < 3; i++) f.T[i] = w->line_Tic[i];
            f.loss_a = w->line_loss_a;
            f.b = b_pose[w->line_frame[k]];
        }
    }
    {
        std::vector<double *> fp(w->n_frames), fs(w->n_frames);
        for (int i = 0; i < w->n_frames; i++) { fp[i] = w->para_pose + 7 * i; fs[i] = w->para_speedbias + 9 * i; }
        chk(tcv_problem_set_frames(p, w->n_frames, fp.data(), fs.data()));
    }
    if (rc != TCV_OK) { delete p; return rc; }
    *out = p;
    return TCV_OK;
}

// =====================================================================================================
// prior (MarginalizationInfo layout, marginalization_factor.h:57-70)
// =====================================================================================================
extern "C" int tcv_prior_create(tcv_prior **out, int m, int n, int nb, const int *size, const int *idx, const double *x0,
                                const double *J0, const double *r0) {
    if (out && n == 0 && nb == 0 && m >= 0) {      // the empty MarginalizationInfo of a marginalisation that kept nothing (tcv_marginalize can return one)
        tcv_prior *pr = new tcv_prior();
        pr->m = m;
        *out = pr;
        return TCV_OK;
    }
    if (!out || n <= 0 || nb <= 0 || m < 0 || !size || !idx || !x0 || !J0 || !r0) { set_error("prior_create: bad argument"); return TCV_ERR_INVALID; }
    // the kernels index fixed-size (128-entry) dx / residual buffers by keep_block_idx: a malformed layout must not reach the device
    if (n > 128) { set_error("prior_create: more than 128 rows"); return TCV_ERR_TOO_LARGE; }
    {
        std::vector<char> used(n, 0);
        int sum_local = 0;
        for (int k = 0; k < nb; k++) {
            if (size[k] <= 0) { set_error("prior_create: keep_block_size must be positive"); return TCV_ERR_INVALID; }
            const int local = size[k] == 7 ? 6 : size[k];      // MarginalizationInfo::localSize, marginalization_factor.cpp:100-103
            if (idx[k] < m || idx[k] - m + local > n) { set_error("prior_create: keep_block_idx outside [m, m + n)"); return TCV_ERR_INVALID; }
            for (int j = 0; j < local; j++) {
                if (used[idx[k] - m + j]) { set_error("prior_create: kept blocks overlap"); return TCV_ERR_INVALID; }
                used[idx[k] - m + j] = 1;
            }
            sum_local += local;
        }
        if (sum_local != n) { set_error("prior_create: local sizes of the kept blocks do not sum to n"); return TCV_ERR_INVALID; }
    }
    tcv_prior *pr = new tcv_prior();
    pr->m = m; pr->n = n;
    int xs = 0;
    for (int k = 0; k < nb; k++) {   // keep_block_idx counts from the start of the [m | n] ordering (marginalization_factor.cpp:312, :347)
        pr->size.push_back(size[k]); pr->idx.push_back(idx[k] - m); pr->xoff.push_back(xs); xs += size[k];
    }
    pr->x0.assign(x0, x0 + xs);
    pr->xsize = xs;
    pr->J0.assign(J0, J0 + (size_t)n * n);
    pr->r0.assign(r0, r0 + n);
    pr->addr.assign(nb, nullptr);
    *out = pr;
    return TCV_OK;
}
extern "C" int tcv_prior_dims(const tcv_prior *pr, int *m, int *n, int *nb, int *xs) {
    if (!pr) return TCV_ERR_INVALID;
    if (m) *m = pr->m;
    if (n) *n = pr->n;
    if (nb) *nb = (int)pr->size.size();
    if (xs) *xs = pr->xsize;
    return TCV_OK;
}
extern "C" int tcv_prior_is_device_resident(const tcv_prior *pr) {
    if (!pr) return 0;
    std::lock_guard<std::mutex> g(pr->mu);
    return pr->host ? 0 : 1;
}
extern "C" int tcv_prior_export(const tcv_prior *pr, int *size, int *idx, double *x0, double *J0, double *r0) {
    if (!pr) return TCV_ERR_INVALID;
    if (x0 || J0 || r0) if (int rc = tcv_prior_host(pr)) return rc;      // a device-resident prior is materialised on demand
    if (size) std::copy(pr->size.begin(), pr->size.end(), size);
    if (idx) for (size_t k = 0; k < pr->idx.size(); k++) idx[k] = pr->idx[k] + pr->m;
    if (x0) std::copy(pr->x0.begin(), pr->x0.end(), x0);
    if (J0) std::copy(pr->J0.begin(), pr->J0.end(), J0);
    if (r0) std::copy(pr->r0.begin(), pr->r0.end(), r0);
    return TCV_OK;
}
// parity/debug: the Schur system (A' n x n row-major, b') the prior was factored from; TCV_ERR_INVALID if not recorded
extern "C" int tcv_prior_export_schur(const tcv_prior *pr, double *As, double *bs) {
    if (pr && pr->n == 0) return TCV_OK;      // (the empty prior of a marginalisation that kept nothing: nothing to copy)
    if (!pr || pr->As.empty()) { set_error("prior carries no Schur system"); return TCV_ERR_INVALID; }
    if (As) std::copy(pr->As.begin(), pr->As.end(), As);
    if (bs) std::copy(pr->bs.begin(), pr->bs.end(), bs);
    return TCV_OK;
}
extern "C" int tcv_prior_keep_block_addresses(const tcv_prior *pr, double **addresses) {
    if (!pr || !addresses) return TCV_ERR_INVALID;
    std::copy(pr->addr.begin(), pr->addr.end(), addresses);
    return TCV_OK;
}
extern "C" void tcv_prior_destroy(tcv_prior *pr) { delete pr; }

// =====================================================================================================
// batch: many independent windows resident in HBM
// =====================================================================================================
static void batch_free(tcv_batch *b) {
    if (!b) return;
    if (b->wait_inflight) (void)hipEventSynchronize(b->ev_inflight);
    if (b->pending)      // the buffers go back to the free list: nothing of this batch may still be running on any stream it used
        for (hipStream_t st : b->streams) { if (st) (void)hipStreamSynchronize(st); else (void)hipDeviceSynchronize(); }
    if (b->dl_staging) { if (b->ev_dl) (void)hipEventSynchronize(b->ev_dl); tcv::host_staging_release(b->dl_staging); b->dl_staging = nullptr; }
    if (b->ev_dl) (void)hipEventDestroy(b->ev_dl);
    if (b->ev_order) (void)hipEventDestroy(b->ev_order);
    if (b->ev_inflight) (void)hipEventDestroy(b->ev_inflight);
    coop_release(b);
    tcv::dev_free(b->d_input);      // (d_dpool, d_win, d_plans, d_plan_base, d_ipool point into it)
    tcv::dev_free(b->d_imublk); tcv::dev_free(b->d_spill); tcv::dev_free(b->d_sqrt_out);
    tcv::dev_free(b->d_coop_ctl); tcv::dev_free(b->d_coop_x); tcv::dev_free(b->d_coop_exp);
    tcv::dev_free(b->d_zero); tcv::dev_free(b->d_state);      // (d_delta, d_scratch, d_summary, d_prof live inside d_zero)
    b->d_zero = nullptr; b->d_prof = nullptr; b->d_delta = nullptr; b->d_scratch = nullptr; b->d_summary = nullptr;
    if (b->ev0) hipEventDestroy(b->ev0);
    if (b->ev1) hipEventDestroy(b->ev1);
    if (b->marg_free) b->marg_free(b);
    delete b;
}


// ---- pinned host staging pool ----------------------------------------------------------------------------------------------
namespace tcv {
namespace {
struct HostBuf { void *p; size_t cap; bool busy; };
std::mutex g_host_mu;
std::vector<HostBuf> g_host_bufs;
enum { HOST_POOL_MAX = 24 };      // idle pinned buffers kept (a host thread that keeps two lock-step groups in flight holds four to six at a time; releasing one to the runtime costs a device-wide wait)
}  // namespace
static std::atomic<long long> g_host_allocs{0}, g_host_frees{0};
extern "C" long long tcv_debug_host_pool_allocs(void) { return g_host_allocs.load() * 1000000 + g_host_frees.load(); }
extern "C" long long tcv_debug_dev_pool_misses(void) { return tcv::dev_pool_misses(); }
void *host_staging_acquire(size_t bytes) {
    size_t cap = (size_t)1 << 20;
    while (cap < bytes) cap <<= 1;
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        HostBuf *best = nullptr;
        for (auto &h : g_host_bufs) if (!h.busy && h.cap >= bytes && (!best || h.cap < best->cap)) best = &h;
        if (best) { best->busy = true; return best->p; }
    }
    void *p = nullptr;
    g_host_allocs++;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> g(g_host_mu);
    g_host_bufs.push_back(HostBuf{p, cap, true});
    return p;
}
void host_staging_release(void *p) {
    if (!p) return;
    void *to_free = nullptr;
    {
        std::lock_guard<std::mutex> g(g_host_mu);
        size_t idle = 0;
        for (auto &h : g_host_bufs) if (!h.busy) idle++;
        for (size_t i = 0; i < g_host_bufs.size(); i++)
            if (g_host_bufs[i].p == p) {
                if (idle >= HOST_POOL_MAX) { to_free = p; g_host_bufs.erase(g_host_bufs.begin() + i); }
                else g_host_bufs[i].busy = false;
                break;
            }
    }
    if (to_free) { g_host_frees++; (void)hipHostFree(to_free); }
}
}  // namespace tcv

extern "C" int tcv_batch_create(tcv_batch **out, tcv_problem *const *problems, tcv_problem *const *marg_problems,
                                double *const *const *marg_drop, const int *marg_num_drop, int n) {
    if (!out || !problems || n <= 0) { set_error("batch_create: bad argument"); return TCV_ERR_INVALID; }
    for (int w = 0; w < n; w++)
        if (!problems[w] || (marg_problems && (!marg_drop || !marg_num_drop))) { set_error("batch_create: null problem in the batch"); return TCV_ERR_INVALID; }
    if (int rc = device_ready()) return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    tcv::prior_refresh_switch();
    const tcv::HostOp host_op;      // host threads of this call: the granted cores shared with the batch-level calls running beside it
    tcv_batch *b = new tcv_batch();
    b->n = n;
    b->problems.assign(problems, problems + n);
    b->packed.resize(n);
    { int cur = -1; if (hipGetDevice(&cur) != hipSuccess) cur = -1; for (auto &pk : b->packed) pk.batch_dev = cur; }
    std::unordered_map<unsigned long long, std::vector<int>> plan_by_hash;   // structure de-duplication
    std::vector<std::pair<const int *, size_t>> plan_src;                    // per device plan: its ints on the host
    size_t ipool_size = 0;
    int max_state = 0, max_nl = 0;
    size_t max_lds = 0;
    int dev = 0, n_cu = 0;
    hipError_t e0 = hipGetDevice(&dev);
    if (e0 == hipSuccess) e0 = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e0 != hipSuccess) { batch_free(b); return hip_fail(e0, "hipDeviceGetAttribute"); }
    if (n_cu <= 0) n_cu = 256;
    // chain layout: two workgroups share a CU's LDS when the batch is larger than the chip; a batch that leaves CUs idle anyway gives
    // every workgroup the whole 160 KiB (fewer chunk passes over the visual factors, the prior's J0 staged in one piece)
    int chain_lds = (n <= n_cu && !getenv("TCV_CHAIN_LDS_DOUBLES")) ? (int)LDS_DOUBLES : chain_lds_doubles();
    b->chain_lds = chain_lds;
    int mode = g_solver_variant;
    // the chain layout is used only if every window of the batch allows it: the packing pass below starts over with the dense layout
    // at the first window that does not
    b->chain = (mode == 0);
    // packing (symbolic elimination, gather programs, data layout) is independent per window: host threads share the work
    // Cooperative mode (tcv_packed.h COOP_*): a chain-layout batch that leaves most of the chip idle gives every window 1 + H workgroups.
    // H: one helper per ~96 point / line factors of the largest window, as many as the CUs allow.  TCV_COOP_H overrides (0: off).
    int coop_h = 0;
    if (mode == 0) {
        int want = g_coop_helpers;
        if (const char *eh = getenv("TCV_COOP_H")) want = atoi(eh);
        size_t fmax = 0;
        for (int w = 0; w < n; w++) fmax = std::max(fmax, problems[w]->proj.size() + problems[w]->line.size());
        if (want < 0) want = fmax >= 96 ? std::min<int>(COOP_MAX_H, std::max<int>(2, (int)((fmax + 95) / 96))) : 0;
        want = std::min(want, (int)COOP_MAX_H);
        while (want > 0 && (long long)((n + 7) / 8) * (1 + want) > n_cu / 8) want--;      // the groups of a launch are dealt round-robin to the XCDs: per XCD ceil(n / 8) groups on n_cu / 8 CUs
        // A large batch seldom has the device to itself -- a lock-step replay drives it from two host threads, and a launch that fills the chip with
        // helpers makes the other thread's launch queue behind it (64 streams: 10.8 K windows/s with seven helpers per window against 12.1 K with
        // two; 128 streams: 15.5 K with three against 17.7 K with two; profiles/r05_replay_coop_helpers.txt).  From 24 windows on a launch takes
        // half the chip -- three quarters if that is what two helpers per window need; TCV_COOP_H / tcv_set_cooperative still decide otherwise.
        if (n >= 24 && g_coop_helpers < 0 && !getenv("TCV_COOP_H")) {
            int w2 = want;
            while (w2 > 0 && (long long)n * (1 + w2) > n_cu / 2) w2--;
            if (w2 < 2 && want >= 2 && (long long)n * 3 <= (long long)n_cu * 3 / 4) w2 = 2;
            want = std::min(want, w2);
        }
        if (want == 1 && g_coop_helpers < 0 && !getenv("TCV_COOP_H")) want = 0;      // a single helper is not worth the hand-offs
        coop_h = want;
    }
    // (a plan out of the cache carries the hash of its template, computed once by the thread that built it)
    auto plan_hash_of = [](const Packed &pk) -> unsigned long long { return pk.tmpl ? pk.tmpl->hash : tcv::plan_content_hash(pk.hdr, pk.ints); };
    auto pack_all = [&](int md, std::string &msg) -> int {
        const int nth = host_op.threads(std::min(n, 16));
        std::vector<int> rcs(n, TCV_OK);
        std::vector<std::string> msgs(nth);
        std::vector<int> who(n, 0);
        tcv::parallel_items(n, nth, [&](int w, int t) {
            rcs[w] = pack_problem(*problems[w], b->packed[w], nullptr, md, chain_lds, true, md == 0 ? coop_h : 0);      // plan + data size
            who[w] = t;
            if (rcs[w] != TCV_OK && msgs[t].empty()) msgs[t] = tcv_last_error();      // the message is thread-local
            // hash of the plan for the structure de-duplication below (a plan out of the cache is de-duplicated by its template: hashed
            // there, once per template, not once per window)
            if (rcs[w] == TCV_OK && !b->packed[w].tmpl && !b->packed[w].key_hashed) b->packed[w].plan_hash = plan_hash_of(b->packed[w]);
        });
        for (int w = 0; w < n; w++) if (rcs[w] != TCV_OK) { msg = msgs[who[w]]; return rcs[w]; }
        return TCV_OK;
    };
    {
        std::string msg;
        int rc = pack_all(mode, msg);
        // a window that does not fit half a CU's LDS (more than ~280 landmarks: its vectors over the unknowns and the chain's working set leave no
        // pool for the visual chunks) gives the WHOLE batch one workgroup per CU with all 160 KiB -- up to 1024 landmarks / 4096 point factors --
        // instead of failing; such a batch runs at 1 / 1.7 of the two-per-CU rate
        if (rc == TCV_ERR_TOO_LARGE && mode == 0 && chain_lds < (int)LDS_DOUBLES && !getenv("TCV_CHAIN_LDS_DOUBLES")) {
            chain_lds = (int)LDS_DOUBLES; b->chain_lds = chain_lds;
            msg.clear();
            rc = pack_all(mode, msg);
        }
        if (rc == TCV_OK && mode == 0 && coop_h > 0)
            for (int w = 0; w < n; w++) {      // what the cooperative master assumes: the prior staged in one piece, one IMU chunk
                const PlanHdr &H = b->packed[w].hdr;
                if (!H.chain || H.n_imu_chunk > 1 || H.camw != (int)CAM_W || (H.prior_n > 0 && H.prior_n * H.prior_n + 2 * H.prior_n > H.c_stage_cap)) {      // (the export layout of the helpers holds CAM_W-wide vectors)
                    coop_h = 0;
                    rc = pack_all(mode, msg);
                    break;
                }
            }
        if (rc == TCV_OK && mode == 0)
            for (int w = 0; w < n; w++)
                if (!b->packed[w].hdr.chain) {      // a window is not chain-eligible: the whole batch uses the dense layout
                    mode = 1; b->chain = false; coop_h = 0;
                    rc = pack_all(mode, msg);
                    break;
                }
        if (rc != TCV_OK) { batch_free(b); if (!msg.empty()) set_error(msg); return rc; }
    }
    const auto t_plans = std::chrono::steady_clock::now();
    std::map<const PlanTemplate *, int> plan_of_tmpl;   // windows that share a cached plan template share the device plan
    size_t dtotal = 0;
    for (int w = 0; w < n; w++) {
        Packed &pk = b->packed[w];
        const PlanInts &pints = pk.tmpl ? pk.tmpl->ints : pk.ints;
        int pid = -1;
        if (pk.tmpl) { auto it = plan_of_tmpl.find(pk.tmpl.get()); if (it != plan_of_tmpl.end()) pid = it->second; }
        if (pid < 0) {
            // equal plans share one device copy: candidates by hash (computed with the packing, in parallel), confirmed by comparison -- the
            // replay's windows are all different (200 KB of plan each), the benchmark's all equal
            if (pk.tmpl) pk.plan_hash = plan_hash_of(pk);
            std::vector<int> &cands = plan_by_hash[pk.plan_hash];
            for (int c : cands)
                if (plan_src[c].second == pints.size() && std::memcmp(&b->plans[c], &pk.hdr, sizeof(PlanHdr)) == 0 &&
                    std::memcmp(plan_src[c].first, pints.data(), sizeof(int) * pints.size()) == 0) { pid = c; break; }
            if (pid < 0) {
                pid = (int)b->plans.size();
                cands.push_back(pid);
                b->plans.push_back(pk.hdr);
                b->plan_base.push_back((long long)ipool_size);
                plan_src.push_back({pints.data(), pints.size()});      // (stays valid: pk.ints / the template live until the copy into the staging buffer)
                ipool_size += pints.size();
            }
            if (pk.tmpl) plan_of_tmpl[pk.tmpl.get()] = pid;
        }
        pk.win.plan = pid;
        pk.win.dbase = (long long)dtotal;
        dtotal += (size_t)pk.win.n_doubles;
        max_state = std::max(max_state, pk.hdr.nx + pk.hdr.nland);
        max_nl = std::max(max_nl, pk.hdr.nc + pk.hdr.nland);
        const int nt = pk.hdr.nt;
        const size_t lds = b->chain ? (size_t)chain_lds * 8
                                    : (size_t)(nt * (nt + 1) / 2 * 256 + 2 * ((pk.hdr.nx + pk.hdr.nland + 1) & ~1) + (3 * pk.hdr.camw + 176) + 64 + pk.hdr.lds_area) * 8;
        max_lds = std::max(max_lds, lds);
        b->spill_stride = std::max(b->spill_stride, pk.hdr.c_spill);
        b->hcl_cap = std::max(b->hcl_cap, (pk.hdr.hcl_total + 63) & ~63);
        b->input_bytes += 8.0 * pk.win.n_doubles;
    }
    // data half: every window written straight into one pinned upload buffer, in parallel
    // ONE pinned staging buffer and ONE device blob for everything the kernels read: [data pool | window headers | plan headers | plan
    // offsets | plan ints], one asynchronous copy on the calling thread's own stream (five synchronous copies through the default stream
    // used to cost a lock-step frame more than its packing)
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_win = up16(sizeof(double) * std::max<size_t>(1, dtotal)), o_plans = up16(o_win + sizeof(WinHdr) * (size_t)n);
    const size_t o_pbase = up16(o_plans + sizeof(PlanHdr) * b->plans.size()), o_ipool = up16(o_pbase + sizeof(long long) * b->plan_base.size());
    // device-resident priors (tcv_batch_get_priors_device): nothing of them is packed or uploaded.  Their J0 | r0 | x0 regions live in a
    // device-only tail behind the uploaded blob (WinHdr::d_prior is relative to the window's slice and simply points there), filled by one
    // splice job per window on the upload's stream
    std::vector<int> splice_win, splice_imu_win;
    std::vector<long long> tail_off(n, -1), tail_imu(n, -1);
    size_t tail_doubles = 0, n_jobs = 0;
    for (int w = 0; w < n; w++) {
        if (b->packed[w].dev_prior_doubles > 0) { splice_win.push_back(w); tail_off[w] = (long long)tail_doubles; tail_doubles += ((size_t)b->packed[w].dev_prior_doubles + 1) & ~(size_t)1; n_jobs++; }
        if (b->packed[w].dev_imu_doubles > 0) { splice_imu_win.push_back(w); tail_imu[w] = (long long)tail_doubles; tail_doubles += ((size_t)b->packed[w].dev_imu_doubles + 1) & ~(size_t)1; n_jobs += problems[w]->imu.size(); }
    }
    const size_t o_jobs = up16(o_ipool + sizeof(int) * std::max<size_t>(1, ipool_size));
    const size_t in_bytes = up16(o_jobs + sizeof(PriorSplice) * n_jobs);
    const size_t dev_bytes = in_bytes + sizeof(double) * tail_doubles;
    if ((in_bytes + sizeof(double) * tail_doubles) / sizeof(double) >= ((size_t)1 << 31)) { batch_free(b); set_error("batch too large (data pool offsets are 32-bit)"); return TCV_ERR_TOO_LARGE; }
    double *h_dpool = (double *)host_staging_acquire(in_bytes);
    if (!h_dpool) { batch_free(b); set_error("hipHostMalloc (upload staging) failed"); return TCV_ERR_HIP; }
    const auto t_dedup = std::chrono::steady_clock::now();
    {
        const int nth = host_op.threads(std::min(n, 16));
        std::vector<int> rcs(n, TCV_OK);
        std::vector<std::string> msgs(nth);
        std::vector<int> who(n, 0);
        const int n_plans = (int)plan_src.size();
        tcv::parallel_items(n + n_plans, nth, [&](int i, int t) {
            if (i < n_plans) {      // the plans straight into the upload buffer (the larger items first)
                std::memcpy((char *)h_dpool + o_ipool + sizeof(int) * (size_t)b->plan_base[i], plan_src[i].first, sizeof(int) * plan_src[i].second);
                return;
            }
            const int w = i - n_plans;
            rcs[w] = pack_problem_data(*problems[w], b->packed[w], nullptr, h_dpool + b->packed[w].win.dbase);
            who[w] = t;
            if (rcs[w] != TCV_OK && msgs[t].empty()) msgs[t] = tcv_last_error();
        });
        for (int w = 0; w < n; w++) { b->packed[w].ints.clear(); b->packed[w].ints.shrink_to_fit(); }
        for (int w = 0; w < n; w++) if (rcs[w] != TCV_OK) { host_staging_release(h_dpool); batch_free(b); if (!msgs[who[w]].empty()) set_error(msgs[who[w]]); return rcs[w]; }
        for (int w : splice_win) {      // the prior region of the window: in the tail, addressed relative to the window's own slice
            Packed &pk = b->packed[w];
            pk.win.d_prior = (int)((long long)(in_bytes / sizeof(double)) + tail_off[w] - pk.win.dbase);
        }
        for (int w : splice_imu_win) {  // likewise the constants of its (device-resident) IMU factors
            Packed &pk = b->packed[w];
            pk.win.d_imu = (int)((long long)(in_bytes / sizeof(double)) + tail_imu[w] - pk.win.dbase);
        }
        for (int w = 0; w < n; w++) b->wins.push_back(b->packed[w].win);
    }
    const auto t_packed = std::chrono::steady_clock::now();
    b->plan_bytes = 4.0 * ipool_size;
    b->state_stride = (max_state + 1) & ~1;
    b->delta_stride = (max_nl + 1) & ~1;
    b->lds_bytes = max_lds;
    b->grid = std::min(n, n_cu * ((b->chain && chain_lds < (int)LDS_DOUBLES) ? 2 : 1));      // (two workgroups per CU only when each takes half its LDS)
    if (const char *eg = getenv("TCV_GRID")) { const int g = atoi(eg); if (g > 0) b->grid = std::min(n, g); }      // tuning experiments
    b->slots = b->grid;
    b->n_cu = n_cu; b->coop_dev = dev;
    if (b->chain && coop_h > 0) {
        b->coop_h = coop_h;
        b->coop_groups = std::min(n, 8 * std::max(1, (n_cu / 8) / (1 + coop_h)));      // whole groups per XCD
        b->slots = b->coop_groups;
        b->grid = (1 + coop_h) * ((b->coop_groups + 7) & ~7);
        b->lds_bytes = (size_t)LDS_DOUBLES * 8;
        int te_max = 0;
        for (auto &H : b->plans) { b->coop_exp_chunks = std::max(b->coop_exp_chunks, H.n_vis_chunk); te_max = std::max(te_max, (H.nt_c * (H.nt_c + 1) / 2) << 8); }
        b->coop_exp_stride = 2 * te_max + COOP_EXP_VEC;
    }
    const int scr = tcv_solve_scratch_doubles() + b->hcl_cap;
    hipStream_t ust = tcv::util_stream();
    // error exits from here on: the asynchronous upload below may still be reading the pinned staging buffer and writing the device blob --
    // both go back to pools another host thread takes from -- so the stream is drained before anything is released
    auto bail = [&]() { (void)(ust ? hipStreamSynchronize(ust) : hipDeviceSynchronize()); host_staging_release(h_dpool); h_dpool = nullptr; batch_free(b); };
#define UP(dst, src, T, cnt)                                                                          \
    do {                                                                                              \
        hipError_t e_ = tcv::dev_malloc((void **)&dst, sizeof(T) * std::max<size_t>(1, (cnt)));             \
        if (e_ != hipSuccess) { bail(); return hip_fail(e_, "hipMalloc"); }                    \
        if (src) {                                                                                    \
            e_ = hipMemcpy(dst, src, sizeof(T) * (cnt), hipMemcpyHostToDevice);                       \
            if (e_ != hipSuccess) { bail(); return hip_fail(e_, "hipMemcpy H2D"); }            \
        }                                                                                             \
    } while (0)
    auto t_marg = t_packed, t_issue = t_packed, t_alloc = t_packed;
    if (marg_problems) {      // the marginalisation problems first: which IMU factor's sqrt_info the solve exports is part of the window headers
        hipError_t e_ = tcv::dev_malloc((void **)&b->d_sqrt_out, sizeof(double) * (size_t)n * 225);
        if (e_ != hipSuccess) { bail(); return hip_fail(e_, "hipMalloc"); }
        const int rc = tcv_marg_attach(b, marg_problems, marg_drop, marg_num_drop);
        if (rc != TCV_OK) { bail(); return rc; }
        for (int w = 0; w < n; w++) b->wins[w].sqrt_export = tcv_marg_sqrt_source(b, w);
    }
    t_marg = std::chrono::steady_clock::now();
    {
        char *hb = (char *)h_dpool;
        std::memcpy(hb + o_win, b->wins.data(), sizeof(WinHdr) * (size_t)n);
        std::memcpy(hb + o_plans, b->plans.data(), sizeof(PlanHdr) * b->plans.size());
        std::memcpy(hb + o_pbase, b->plan_base.data(), sizeof(long long) * b->plan_base.size());
        PriorSplice *hj = (PriorSplice *)(hb + o_jobs);
        for (size_t q = 0; q < splice_win.size(); q++) {
            const int w = splice_win[q];
            const tcv_prior *pr = problems[w]->prior[0].prior;
            PriorSplice &J = hj[q];
            std::memset(&J, 0, sizeof J);
            J.src = pr->d_block; J.dst = b->packed[w].win.dbase + b->packed[w].win.d_prior;
            J.n = pr->n; J.k0 = b->packed[w].win.prior_k0; J.nblk = (int)pr->size.size();
            J.k0_src = b->packed[w].prior_k0_deferred ? pr->d_k0 : nullptr; J.win = w;
            for (int k = 0; k < J.nblk; k++) { J.goff[k] = pr->x_goff[k]; J.size[k] = pr->size[k]; }
        }
        {
            size_t q = splice_win.size();
            for (int w : splice_imu_win)
                for (size_t f = 0; f < problems[w]->imu.size(); f++, q++) {
                    PriorSplice &J = hj[q];
                    std::memset(&J, 0, sizeof J);
                    J.kind = 1; J.src = problems[w]->imu[f].dev->d_out;
                    J.dst = b->packed[w].win.dbase + b->packed[w].win.d_imu + (long long)f * IMU_CONST;
                }
        }
        hipError_t e_ = tcv::dev_malloc(&b->d_input, dev_bytes);
        if (e_ == hipSuccess) e_ = hipMemcpyAsync(b->d_input, hb, in_bytes, hipMemcpyHostToDevice, ust);
        if (e_ != hipSuccess) { bail(); return hip_fail(e_, "upload of the batch"); }
        char *db = (char *)b->d_input;
        b->d_dpool = (double *)db; b->d_win = (WinHdr *)(db + o_win); b->d_plans = (PlanHdr *)(db + o_plans);
        b->d_plan_base = (long long *)(db + o_pbase); b->d_ipool = (int *)(db + o_ipool);
        if (n_jobs > 0) {      // behind the upload on its stream; a source whose producer did not wait for its kernel carries an event
            for (int w : splice_imu_win)
                for (size_t f = 0; f < problems[w]->imu.size(); f++)
                    if (const int rcw = problems[w]->imu[f].dev->dev->wait_ready(ust)) { bail(); return rcw; }
            for (int w : splice_win)
                if (const int rcw = problems[w]->prior[0].prior->dev->wait_ready(ust)) { bail(); return rcw; }
            const int rcs = tcv::launch_prior_splice((const PriorSplice *)(db + o_jobs), (int)n_jobs, b->d_dpool, (void *)b->d_win, ust);
            if (rcs != TCV_OK) { bail(); return rcs; }
        }
    }
    t_issue = std::chrono::steady_clock::now();
    UP(b->d_state, (double *)nullptr, double, (size_t)n * b->state_stride);
    // the four buffers that start as zeros share ONE allocation and one memset: four fill kernels of 4 - 5 us with their launch gaps sat between
    // the upload and the frame's solve (a lock-step frame's GPU timeline, tools/gpu_calls.md#r05_gpu_z43: ~50 us of a 2.1 ms frame)
    size_t zero_bytes = 0;
    {
        auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t o_delta = 0, o_scr = up256(o_delta + sizeof(double) * std::max<size_t>(1, (size_t)n * b->delta_stride));
        const size_t o_sum = up256(o_scr + sizeof(double) * std::max<size_t>(1, (size_t)b->slots * scr));
        const size_t o_prof = up256(o_sum + sizeof(DevSummary) * std::max<size_t>(1, (size_t)n));
        zero_bytes = up256(o_prof + sizeof(double) * std::max<size_t>(1, (size_t)32 * b->slots));
        hipError_t e_ = tcv::dev_malloc(&b->d_zero, zero_bytes);
        if (e_ != hipSuccess) { bail(); return hip_fail(e_, "hipMalloc"); }
        char *z = (char *)b->d_zero;
        b->d_delta = (double *)(z + o_delta); b->d_scratch = (double *)(z + o_scr); b->d_summary = (DevSummary *)(z + o_sum); b->d_prof = (double *)(z + o_prof);
    }
    if (b->chain) {
        UP(b->d_imublk, (double *)nullptr, double, (size_t)b->slots * 16 * IMU_BLK);
        UP(b->d_spill, (double *)nullptr, double, (size_t)b->slots * b->spill_stride);
    }
    if (b->coop_h > 0) {
        UP(b->d_coop_ctl, (int *)nullptr, int, (size_t)b->coop_groups * COOP_CTL_INTS);
        UP(b->d_coop_x, (double *)nullptr, double, (size_t)b->coop_groups * COOP_X_DOUBLES);
        UP(b->d_coop_exp, (double *)nullptr, double, (size_t)b->coop_groups * b->coop_exp_chunks * b->coop_exp_stride);
    }
#undef UP
    t_alloc = std::chrono::steady_clock::now();
    e0 = hipMemsetAsync(b->d_zero, 0, zero_bytes, ust);
    {      // the batch is complete on the device before any stream uses it (and the upload is drained before its staging buffer is released, whatever the memsets returned)
        const hipError_t es = ust ? hipStreamSynchronize(ust) : hipDeviceSynchronize();
        if (e0 == hipSuccess) e0 = es;
    }
    host_staging_release(h_dpool);
    h_dpool = nullptr;
    tcv::flush_deferred(ust);      // (this thread just waited for its stream)
    if (e0 != hipSuccess) { batch_free(b); return hip_fail(e0, "upload of the batch"); }
    e0 = hipEventCreate(&b->ev0);
    if (e0 == hipSuccess) e0 = hipEventCreate(&b->ev1);
    if (e0 != hipSuccess) { batch_free(b); return hip_fail(e0, "hipEventCreate"); }
    const auto t_up = std::chrono::steady_clock::now();
    if (getenv("TCV_DEBUG_PACK")) {
        const auto t_end = std::chrono::steady_clock::now();
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
        fprintf(stderr, "[batch_create] n %d: plans %.3f ms, plan pool + staging %.3f ms, data %.3f ms\n", n, ms(t_begin, t_plans), ms(t_plans, t_dedup), ms(t_dedup, t_packed));
        fprintf(stderr, "[batch_create] n %d: pack %.3f ms, marg attach %.3f ms, blob + upload issue + splice %.3f ms, allocations %.3f ms, memsets + sync %.3f ms (%d splice jobs, %.1f KB up)\n", n,
                ms(t_begin, t_packed), ms(t_packed, t_marg), ms(t_marg, t_issue), ms(t_issue, t_alloc), ms(t_alloc, t_up), (int)n_jobs, in_bytes / 1024.0);
        (void)t_end;
    }
    *out = b;
    return TCV_OK;
}
// The marginalisation problems of a batch that was created without them: their packing and upload can then run while the batch's solve
// is on the device (the native estimator does: ~0.3 ms of a lock-step frame).  The solve of such a batch does not export an IMU factor's
// sqrt_info (which factor is part of the window headers, uploaded before): the marginalisation kernel forms its own -- same function, same bits.
extern "C" int tcv_batch_attach_marginalization(tcv_batch *b, tcv_problem *const *marg_problems, double *const *const *marg_drop, const int *marg_num_drop) {
    if (!b || !marg_problems || !marg_drop || !marg_num_drop) { set_error("batch_attach_marginalization: bad argument"); return TCV_ERR_INVALID; }
    if (b->marg) { set_error("batch_attach_marginalization: the batch has marginalisation problems already"); return TCV_ERR_INVALID; }
    const tcv::HostOp host_op;
    const int rc = tcv_marg_attach(b, marg_problems, marg_drop, marg_num_drop);
    if (rc != TCV_OK && b->marg && b->marg_free) b->marg_free(b);      // (nothing half-attached stays behind)
    return rc;
}
extern "C" void tcv_batch_destroy(tcv_batch *b) { batch_free(b); }
extern "C" int tcv_batch_size(const tcv_batch *b) { return b ? b->n : 0; }

extern "C" void tcv_solver_options_default(tcv_solver_options *o) {
    if (!o) return;
    o->max_num_iterations = 8;
    o->max_solver_time_in_seconds = 0.0;
    o->fixed_iterations = 0;
    o->workgroups_per_window = 0;
    o->use_mfma = 1;
    o->threads_per_window = 256;
    o->record_first_step = 0;
}

int tcv_batch_enter_stream(tcv_batch *b, void *hip_stream) {
    hipStream_t st = (hipStream_t)hip_stream;
    if (b->wait_inflight) HIPCHK(hipStreamWaitEvent(st, b->ev_inflight, 0));      // (the stream that work ran on may be gone: its event orders the new call)
    else if (b->pending && b->last_stream != st) {
        if (!b->ev_order) HIPCHK(hipEventCreateWithFlags(&b->ev_order, hipEventDisableTiming));
        HIPCHK(hipEventRecord(b->ev_order, b->last_stream));
        HIPCHK(hipStreamWaitEvent(st, b->ev_order, 0));
    }
    if (std::find(b->streams.begin(), b->streams.end(), st) == b->streams.end()) b->streams.push_back(st);
    b->last_stream = st; b->pending = true;
    return TCV_OK;
}

extern "C" int tcv_batch_solve(tcv_batch *b, const tcv_solver_options *o, void *hip_stream) {
    if (!b || !o) return TCV_ERR_INVALID;
    if (hip_stream == TCV_STREAM_THREAD) hip_stream = (void *)tcv::util_stream();
    SolveArgs a;
    std::memset(&a, 0, sizeof a);
    a.win = b->d_win; a.plans = b->d_plans; a.plan_base = b->d_plan_base; a.ipool = b->d_ipool; a.dpool = b->d_dpool;
    a.state_out = b->d_state; a.summary = b->d_summary; a.first_delta = o->record_first_step ? b->d_delta : nullptr;
    a.scratch = b->d_scratch;
    a.prof = b->d_prof;
    a.nwin = b->n; a.state_stride = b->state_stride; a.delta_stride = b->delta_stride; a.scratch_stride = tcv_solve_scratch_doubles() + b->hcl_cap;
    a.max_iterations = o->max_num_iterations; a.fixed_iterations = o->fixed_iterations; a.use_mfma = o->use_mfma;
    a.chain = b->chain ? 1 : 0; a.chain_td = 0;
    // ESTIMATE_TD windows: the kernel instance with ProjectionTdFactor.  It also takes the windows with a relocalisation pose (camera vectors 184
    // wide, PlanHdr::camw): the width is a literal in the instance every shipped configuration runs (build.py: -DTCV_CAMW_CONST for that
    // translation unit; read from the plan it cost the benchmark 0.3 %), a plan field in this one
    if (b->chain) for (int w = 0; w < b->n; w++) if ((b->packed[w].hdr.flags & 1) || b->packed[w].hdr.camw != (int)CAM_W) { a.chain_td = 1; break; }
    a.imublk = b->d_imublk; a.spill = b->d_spill; a.spill_stride = b->spill_stride;
    a.max_ticks = 0;
    a.sqrt_out = b->d_sqrt_out;
    a.gauge_fix = b->fuse_gauge ? 1 : 0;
    // cooperative plans also run on the single-workgroup kernel (workgroups_per_window = 1): same chunks, same additions, same bits
    bool coop = b->coop_h > 0 && o->workgroups_per_window != 1;
    if (coop && b->coop_claim == 0)      // (a claim still held: the previous cooperative solve of this batch, same stream order, same rotation)
        if (!coop_admit(b, b->coop_groups, 1 + b->coop_h)) coop = false;      // the XCDs are taken by other cooperative launches: the same plan on one workgroup per window, the same bits
    int grid = b->grid;
    b->last_wg = coop ? 1 + b->coop_h : 1;
    if (coop) {
        a.coop_h = b->coop_h; a.coop_groups = b->coop_groups; a.coop_exp_chunks = b->coop_exp_chunks; a.coop_exp_stride = b->coop_exp_stride;
        a.coop_ctl = b->d_coop_ctl; a.coop_x = b->d_coop_x; a.coop_exp = b->d_coop_exp; a.coop_rot = b->coop_rot;
        int dev = 0, khz = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
        a.coop_timeout = (long long)khz * 2000;      // 2 s
    } else if (b->coop_h > 0) grid = b->slots;
    b->sqrt_out_valid = b->d_sqrt_out != nullptr;
    if (const char *rm = getenv("TCV_ROLE_MODE")) a.role_mode = atoi(rm);
    if (const char *sk = getenv("TCV_ABLATE_SKIP")) a.pad2 = (int)(unsigned)strtoul(sk, nullptr, 0);      // -DTCV_ABLATE builds only read it
    if (o->max_solver_time_in_seconds > 0.0 && !o->fixed_iterations) {
        int dev = 0, khz = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;      // 100 MHz on CDNA
        a.max_ticks = (long long)(o->max_solver_time_in_seconds * 1e3 * (double)khz);
        if (a.max_ticks < 1) a.max_ticks = 1;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    if (int rc = tcv_batch_enter_stream(b, hip_stream)) return rc;
    if (coop) HIPCHK(hipMemsetAsync(b->d_coop_ctl, 0, sizeof(int) * (size_t)b->coop_groups * COOP_CTL_INTS, st));
    HIPCHK(hipEventRecord(b->ev0, st));
    const int rc = tcv_launch_solve(&a, grid, (!b->chain && o->threads_per_window == 512) ? 512 : 256, b->lds_bytes, hip_stream);
    if (rc != 0) return hip_fail((hipError_t)rc, "solve kernel launch");
    HIPCHK(hipEventRecord(b->ev1, st));
    b->solved = true;
    b->gauge_in_solve = a.gauge_fix != 0;
    if (a.gauge_fix) b->gauge_fixed = true;
    return TCV_OK;
}
extern "C" int tcv_batch_marginalize(tcv_batch *b, void *hip_stream) {
    if (!b) return TCV_ERR_INVALID;
    if (hip_stream == TCV_STREAM_THREAD) hip_stream = (void *)tcv::util_stream();
    if (int rc = tcv_batch_enter_stream(b, hip_stream)) return rc;
    return tcv_marg_run(b, hip_stream);
}
// waits for every stream this batch has work in flight on (the whole device when one of them is the default stream), not for the
// device: batches driven from different host threads on different streams overlap (bench.py --mode stream)
extern "C" int tcv_batch_synchronize(tcv_batch *b) {
    if (!b) return TCV_ERR_INVALID;
    if (b->wait_inflight) { HIPCHK(hipEventSynchronize(b->ev_inflight)); b->wait_inflight = false; }
    for (hipStream_t st : b->streams) { if (st) HIPCHK(hipStreamSynchronize(st)); else HIPCHK(hipDeviceSynchronize()); }
    b->streams.clear();
    b->pending = false;
    coop_release(b);
    if (b->solved) hipEventElapsedTime(&b->solve_ms, b->ev0, b->ev1);
    tcv_marg_elapsed(b);
    return TCV_OK;
}
extern "C" int tcv_batch_download_states(tcv_batch *b) {
    if (!b || !b->solved) return TCV_ERR_INVALID;
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    b->h_state.resize((size_t)b->n * b->state_stride);
    {      // through pinned staging on the calling thread's own stream (the default stream serialises the host threads of a process)
        const size_t bytes = sizeof(double) * b->h_state.size();
        void *hs = host_staging_acquire(bytes);
        if (!hs) { set_error("hipHostMalloc (download staging) failed"); return TCV_ERR_HIP; }
        hipStream_t ust = tcv::util_stream();
        hipError_t e_ = hipMemcpyAsync(hs, b->d_state, bytes, hipMemcpyDeviceToHost, ust);
        if (e_ == hipSuccess) e_ = ust ? hipStreamSynchronize(ust) : hipDeviceSynchronize();
        if (e_ == hipSuccess) std::memcpy(b->h_state.data(), hs, bytes);
        host_staging_release(hs);
        if (e_ != hipSuccess) return hip_fail(e_, "download of the states");
    }
    for (int w = 0; w < b->n; w++) {
        const Packed &pk = b->packed[w];
        const tcv_problem &p = *b->problems[w];
        const double *x = b->h_state.data() + (size_t)w * b->state_stride;
        int off = 0;
        for (int blk : pk.cam_block) { std::memcpy(p.blocks[blk].addr, x + off, sizeof(double) * p.blocks[blk].size); off += p.blocks[blk].size; }
        for (int blk : pk.lm_block) p.blocks[blk].addr[0] = x[off++];
    }
    return TCV_OK;
}
// Split form for callers that enqueue more work behind the solve: _begin puts the copy of the states (and of the summary heads) on `hip_stream`
// behind whatever the batch has in flight there and returns; the caller may launch the marginalisation right behind it; _end waits for the
// COPY only (an event), not for the stream, and hands the results out.  The native estimator does: the marginalisation starts the moment the
// states have left, not a host round trip later.
extern "C" int tcv_batch_download_states_begin(tcv_batch *b, void *hip_stream) {
    if (!b || !b->solved) return TCV_ERR_INVALID;
    if (b->dl_staging) { set_error("batch_download_states_begin: a download is in flight already"); return TCV_ERR_INVALID; }
    if (hip_stream == TCV_STREAM_THREAD) hip_stream = (void *)tcv::util_stream();
    hipStream_t st = (hipStream_t)hip_stream;
    if (int rc = tcv_batch_enter_stream(b, hip_stream)) return rc;
    b->h_state.resize((size_t)b->n * b->state_stride);
    const size_t sbytes = sizeof(double) * b->h_state.size(), hbytes = (size_t)32 * b->n;
    static_assert(offsetof(DevSummary, final_cost) == 24 && offsetof(DevSummary, num_iterations) == 0 && offsetof(DevSummary, termination) == 4, "head of DevSummary");
    char *hs = (char *)host_staging_acquire(sbytes + hbytes);
    if (!hs) { set_error("hipHostMalloc (download staging) failed"); return TCV_ERR_HIP; }
    hipError_t e_ = hipSuccess;
    if (!b->ev_dl) e_ = hipEventCreateWithFlags(&b->ev_dl, hipEventDisableTiming);
    if (e_ == hipSuccess) e_ = hipMemcpyAsync(hs, b->d_state, sbytes, hipMemcpyDeviceToHost, st);
    if (e_ == hipSuccess) e_ = hipMemcpy2DAsync(hs + sbytes, 32, b->d_summary, sizeof(DevSummary), 32, (size_t)b->n, hipMemcpyDeviceToHost, st);
    if (e_ == hipSuccess) e_ = hipEventRecord(b->ev_dl, st);
    if (e_ != hipSuccess) { (void)(st ? hipStreamSynchronize(st) : hipDeviceSynchronize()); host_staging_release(hs); return hip_fail(e_, "download of the states"); }
    b->dl_staging = hs;
    return TCV_OK;
}
extern "C" int tcv_batch_download_states_end(tcv_batch *b, int *num_iterations, int *termination, double *final_cost) {
    if (!b || !b->dl_staging || !b->ev_dl) { set_error("batch_download_states_end: no download in flight"); return TCV_ERR_INVALID; }
    char *hs = (char *)b->dl_staging;
    const size_t sbytes = sizeof(double) * b->h_state.size();
    const hipError_t e_ = hipEventSynchronize(b->ev_dl);
    if (e_ == hipSuccess) {
        std::memcpy(b->h_state.data(), hs, sbytes);
        for (int w = 0; w < b->n; w++) {
            const char *q = hs + sbytes + (size_t)32 * w;
            int it, tm; double fc;
            std::memcpy(&it, q, 4); std::memcpy(&tm, q + 4, 4); std::memcpy(&fc, q + 24, 8);
            if (num_iterations) num_iterations[w] = it;
            if (termination) termination[w] = tm;
            if (final_cost) final_cost[w] = fc;
        }
    }
    host_staging_release(hs);
    b->dl_staging = nullptr;
    if (e_ != hipSuccess) return hip_fail(e_, "download of the states");
    // the solve (and the gauge fix) lie behind the copy on the stream: they have finished -- its duration and the CUs its cooperative launch held
    coop_release(b);
    if (b->solved) (void)hipEventElapsedTime(&b->solve_ms, b->ev0, b->ev1);
    for (int w = 0; w < b->n; w++) {
        const Packed &pk = b->packed[w];
        const tcv_problem &p = *b->problems[w];
        const double *x = b->h_state.data() + (size_t)w * b->state_stride;
        int off = 0;
        for (int blk : pk.cam_block) { std::memcpy(p.blocks[blk].addr, x + off, sizeof(double) * p.blocks[blk].size); off += p.blocks[blk].size; }
        for (int blk : pk.lm_block) p.blocks[blk].addr[0] = x[off++];
    }
    return TCV_OK;
}
// States into the callers' blocks AND the three numbers of the summary a per-frame caller reads (the reference reads one:
// summary.iterations.size(), estimator.cpp:1902) in ONE device round trip: the states and the heads of the summary records (32 of their
// 4 136 bytes, a pitched copy) on the calling thread's stream, one wait.
extern "C" int tcv_batch_download_states_brief(tcv_batch *b, int *num_iterations, int *termination, double *final_cost) {
    if (!b || !b->solved) return TCV_ERR_INVALID;
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    if (int rc = tcv_batch_download_states_begin(b, (void *)tcv::util_stream())) return rc;
    const int rc = tcv_batch_download_states_end(b, num_iterations, termination, final_cost);
    if (rc == TCV_OK) { if (int rcs = tcv_batch_synchronize(b)) return rcs; }      // (nothing of this call stays in flight)
    return rc;
}
static void summary_to_public(const DevSummary &s, tcv_solver_summary *o) {
    std::memset(o, 0, sizeof *o);
    o->num_iterations = s.num_iterations; o->termination = s.termination;
    o->initial_cost = s.initial_cost; o->final_cost = s.final_cost;
    for (int i = 0; i < TCV_MAX_TRACE; i++) {
        o->cost[i] = s.cost[i]; o->cost_candidate[i] = s.cost_candidate[i]; o->model_cost_change[i] = s.model_cost_change[i];
        o->radius[i] = s.radius[i]; o->mu[i] = s.mu[i]; o->rho[i] = s.rho[i]; o->step_norm[i] = s.step_norm[i];
        o->step_ok[i] = s.step_ok[i]; o->dogleg_case[i] = s.dogleg_case[i];
    }
}
extern "C" int tcv_batch_get_summaries(tcv_batch *b, tcv_solver_summary *out, int n) {
    if (!b || !out || n < 0 || n > b->n) { set_error("batch_get_summaries: n exceeds the batch size"); return TCV_ERR_INVALID; }
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    const size_t bytes = sizeof(DevSummary) * (size_t)n;
    DevSummary *h = (DevSummary *)host_staging_acquire(std::max<size_t>(bytes, 16));
    if (!h) { set_error("hipHostMalloc (download staging) failed"); return TCV_ERR_HIP; }
    hipStream_t ust = tcv::util_stream();
    hipError_t e_ = n ? hipMemcpyAsync(h, b->d_summary, bytes, hipMemcpyDeviceToHost, ust) : hipSuccess;
    if (e_ == hipSuccess) e_ = ust ? hipStreamSynchronize(ust) : hipDeviceSynchronize();
    if (e_ == hipSuccess) for (int i = 0; i < n; i++) summary_to_public(h[i], out + i);
    host_staging_release(h);
    if (e_ != hipSuccess) return hip_fail(e_, "download of the summaries");
    return TCV_OK;
}
// tangent step of iteration 1 in problem order: free camera blocks in the order they were added
// (local size each), then the inverse depths in landmark order.  Returns the length in *len.
extern "C" int tcv_batch_get_first_step(tcv_batch *b, int window, double *out, int cap, int *len) {
    if (!b || window < 0 || window >= b->n || !out) return TCV_ERR_INVALID;
    const PlanHdr &H = b->plans[b->wins[window].plan];
    std::vector<double> d(b->delta_stride);
    HIPCHK(hipMemcpy(d.data(), b->d_delta + (size_t)window * b->delta_stride, sizeof(double) * b->delta_stride, hipMemcpyDeviceToHost));
    const tcv_problem &p = *b->problems[window];
    const Packed &pk = b->packed[window];
    int k = 0;
    const std::vector<int> &loff = pk.cam_loff;
    for (size_t c = 0; c < pk.cam_block.size(); c++) {
        if (loff[c] < 0) continue;
        const ParamBlock &pb = p.blocks[pk.cam_block[c]];
        const int ls = pb.kind == KIND_POSE ? 6 : pb.size;
        for (int j = 0; j < ls; j++) { if (k >= cap) return TCV_ERR_INVALID; out[k++] = d[loff[c] + j]; }
    }
    for (int l = 0; l < H.nland; l++) { if (k >= cap) return TCV_ERR_INVALID; out[k++] = d[H.nc + l]; }
    if (len) *len = k;
    return TCV_OK;
}
extern "C" int tcv_batch_get_prior(tcv_batch *b, int window, tcv_prior **out) {
    if (!b || !out) return TCV_ERR_INVALID;
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    return tcv_marg_get_prior(b, window, out);
}
extern "C" int tcv_batch_get_priors(tcv_batch *b, tcv_prior **out, int n) {
    if (!b || !out || n != b->n) { set_error("batch_get_priors: n must be the batch size"); return TCV_ERR_INVALID; }
    for (int k = 0; k < n; k++) out[k] = nullptr;
    // (the per-window fallback of tcv_marg_get_prior copies on the null stream, which is not ordered behind a marginalisation launched on a
    // non-blocking stream: wait for the batch's own streams first)
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    const tcv::HostOp host_op;
    const int nth = host_op.threads(std::max(1, std::min(n / 16, 8)));
    std::vector<int> rcs(nth, TCV_OK);
    std::vector<std::string> msgs(nth);
    auto work = [&](int t) {
        // a worker thread starts on device 0: the batch's buffers live on its own device.  The CALLER runs this lambda too (parallel_run):
        // whatever thread it is, its current device is put back
        int prev = -1;
        if (nth > 1 && hipGetDevice(&prev) == hipSuccess && prev != b->coop_dev) (void)hipSetDevice(b->coop_dev); else prev = -1;
        for (int k = t; k < n; k += nth) {
            if (!tcv_marg_has_problem(b, k)) continue;      // not marginalised: out[k] stays NULL
            const int rc = tcv_marg_get_prior(b, k, &out[k]);
            if (rc != TCV_OK) { rcs[t] = rc; msgs[t] = tcv_last_error(); break; }
        }
        if (prev >= 0) (void)hipSetDevice(prev);
    };
    tcv::parallel_run(nth, work);
    for (int t = 0; t < nth; t++)
        if (rcs[t] != TCV_OK) {
            for (int k = 0; k < n; k++) if (out[k]) { tcv_prior_destroy(out[k]); out[k] = nullptr; }
            set_error(msgs[t]);
            return rcs[t];
        }
    return TCV_OK;
}
extern "C" int tcv_batch_get_priors_device_async(tcv_batch *b, tcv_prior **out, int n) {
    if (!b || !out || n != b->n) { set_error("batch_get_priors_device_async: n must be the batch size"); return TCV_ERR_INVALID; }
    for (int k = 0; k < n; k++) out[k] = nullptr;
    const int rc = tcv_marg_get_priors_device(b, out, n, true);      // (no wait: the marginalisation may still be running)
    if (rc == TCV_OK && b->pending && !b->streams.empty()) {
        // the batch outlives this call with work in flight: from here on that work is an event, not the streams it runs on -- the calling
        // thread may end (its stream goes with it) before somebody asks for the status or destroys the batch
        (void)tcv_marg_status_prefetch(b, (void *)b->last_stream);
        hipError_t e = hipSuccess;
        if (!b->ev_inflight) e = hipEventCreateWithFlags(&b->ev_inflight, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(b->ev_inflight, b->last_stream);
        if (e != hipSuccess) {
            // no event to track the work by: wait for it here (the streams are still this batch's) -- the handles stay valid, the call merely
            // was not asynchronous; if even that fails the handles are destroyed: "on an error out[] is all NULL" (include/tcv.h)
            const int rcs = tcv_batch_synchronize(b);
            if (rcs != TCV_OK) { for (int k = 0; k < n; k++) if (out[k]) { tcv_prior_destroy(out[k]); out[k] = nullptr; } return rcs; }
            return TCV_OK;
        }
        b->streams.clear();
        b->wait_inflight = true;
    }
    return rc;
}
extern "C" int tcv_batch_get_priors_device(tcv_batch *b, tcv_prior **out, int n) {
    if (!b || !out || n != b->n) { set_error("batch_get_priors_device: n must be the batch size"); return TCV_ERR_INVALID; }
    for (int k = 0; k < n; k++) out[k] = nullptr;
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    return tcv_marg_get_priors_device(b, out, n, false);
}
extern "C" int tcv_problem_set_marginalization_prior(tcv_problem *p, const tcv_prior *prior) {
    if (!p || !prior || p->prior.size() != 1) { set_error("set_marginalization_prior: the problem must hold exactly one marginalisation factor"); return TCV_ERR_INVALID; }
    const tcv_prior *old = p->prior[0].prior;
    if (old->n != prior->n || old->size != prior->size || old->idx != prior->idx) { set_error("set_marginalization_prior: the new prior has another layout"); return TCV_ERR_INVALID; }
    p->prior[0].prior = prior;
    return TCV_OK;
}
extern "C" int tcv_problems_set_marginalization_prior(tcv_problem *const *problems, tcv_prior *const *priors, int n) {
    if (!problems || !priors || n < 0) return TCV_ERR_INVALID;
    for (int k = 0; k < n; k++) if (int rc = tcv_problem_set_marginalization_prior(problems[k], priors[k])) return rc;
    return TCV_OK;
}
extern "C" void tcv_priors_destroy(tcv_prior *const *priors, int n) {
    if (priors) for (int k = 0; k < n; k++) delete priors[k];
}
extern "C" int tcv_batch_download_priors(tcv_batch *b) {
    if (!b) return TCV_ERR_INVALID;
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    return tcv_marg_download(b, 0);
}
extern "C" int tcv_batch_download_priors_compact(tcv_batch *b) {
    if (!b) return TCV_ERR_INVALID;
    if (b->pending) if (int rc = tcv_batch_synchronize(b)) return rc;
    return tcv_marg_download(b, 1);
}
extern "C" int tcv_batch_stats(tcv_batch *b, double *input_bytes, double *solve_ms, double *marg_ms) {
    if (!b) return TCV_ERR_INVALID;
    if (input_bytes) *input_bytes = b->input_bytes;
    if (solve_ms) *solve_ms = b->solve_ms;
    if (marg_ms) *marg_ms = b->marg_ms;
    return TCV_OK;
}
// TCV_PROFILE builds only: copies (and clears) the 32 per-phase cycle accumulators of the solve kernel
extern "C" int tcv_batch_profile(tcv_batch *b, double *out32) {
    if (!b || !out32) return TCV_ERR_INVALID;
    std::vector<double> h((size_t)32 * b->slots);
    HIPCHK(hipMemcpy(h.data(), b->d_prof, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; i++) { out32[i] = 0; for (int g = 0; g < b->slots; g++) out32[i] += h[(size_t)g * 32 + i]; }
    HIPCHK(hipMemset(b->d_prof, 0, sizeof(double) * h.size()));
    return TCV_OK;
}
extern "C" int tcv_batch_cooperative(const tcv_batch *b, int *helpers, int *groups, int *chunks, int *last_solve_workgroups) {
    if (!b) return TCV_ERR_INVALID;
    if (last_solve_workgroups) *last_solve_workgroups = b->last_wg;
    if (helpers) *helpers = b->coop_h;
    if (groups) *groups = b->coop_h > 0 ? b->coop_groups : 0;
    if (chunks) { int c = 0; for (auto &H : b->plans) c = std::max(c, H.n_vis_chunk); *chunks = c; }
    return TCV_OK;
}
// developer diagnostics: the control block of group `group` (COOP_CTL_INTS ints: request / served sequence numbers, abort flag, progress
// marks), copied on a private stream so that it can be read while a launch is still running
extern "C" int tcv_batch_debug_coop(tcv_batch *b, int group, int *out64) {
    if (!b || !out64 || b->coop_h <= 0 || group < 0 || group >= b->coop_groups) { set_error("debug_coop: no cooperative plan / bad group"); return TCV_ERR_INVALID; }
    hipStream_t st = nullptr;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipError_t e = hipMemcpyAsync(out64, b->d_coop_ctl + (size_t)group * COOP_CTL_INTS, sizeof(int) * COOP_CTL_INTS, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipStreamDestroy(st);
    if (e != hipSuccess) return hip_fail(e, "debug_coop copy");
    return TCV_OK;
}
extern "C" int tcv_batch_layout(const tcv_batch *b) { return b ? (b->chain ? 0 : 1) : TCV_ERR_INVALID; }
extern "C" int tcv_batch_plan_stats(tcv_batch *b, int *num_plans, double *plan_bytes, int *grid, int *lds_bytes) {
    if (!b) return TCV_ERR_INVALID;
    if (num_plans) *num_plans = (int)b->plans.size();
    if (plan_bytes) *plan_bytes = b->plan_bytes;
    if (grid) *grid = b->grid;
    if (lds_bytes) *lds_bytes = (int)b->lds_bytes;
    return TCV_OK;
}

// ceres::Solve(options, &problem, &summary)   estimator.cpp:1900
extern "C" int tcv_solve(const tcv_solver_options *o, tcv_problem *p, tcv_solver_summary *summary) {
    if (!o || !p) return TCV_ERR_INVALID;
    tcv_batch *b = nullptr;
    tcv_problem *arr[1] = {p};
    int rc = tcv_batch_create(&b, arr, nullptr, nullptr, nullptr, 1);
    if (rc != TCV_OK) return rc;
    rc = tcv_batch_solve(b, o, nullptr);
    if (rc == TCV_OK) rc = tcv_batch_synchronize(b);
    tcv_solver_summary s;
    if (rc == TCV_OK) rc = tcv_batch_get_summaries(b, &s, 1);
    if (rc == TCV_OK) {
        if (summary) *summary = s;
        if (!(s.final_cost == s.final_cost) || s.termination == 5) {
            set_error("solver failure (NaN cost or no valid step); parameter blocks left untouched");
            rc = TCV_ERR_NUMERIC;
        } else rc = tcv_batch_download_states(b);
    }
    tcv_batch_destroy(b);
    return rc;
}

// single-window MarginalizationInfo path (marginalization_factor.cpp:89-321): a batch of one
extern "C" int tcv_marginalize(tcv_problem *p, double *const *drop, int num_drop, tcv_prior **out) {
    if (!p || !drop || num_drop <= 0 || !out) return TCV_ERR_INVALID;
    tcv_batch *b = nullptr;
    tcv_problem *arr[1] = {p};
    double *const *dr[1] = {drop};
    int nd[1] = {num_drop};
    // the marginalisation problem doubles as the (unused) solve problem of the batch
    int rc = tcv_batch_create(&b, arr, arr, dr, nd, 1);
    if (rc != TCV_OK) return rc;
    rc = tcv_batch_marginalize(b, nullptr);
    if (rc == TCV_OK) rc = tcv_batch_synchronize(b);
    if (rc == TCV_OK) rc = tcv_batch_get_prior(b, 0, out);
    tcv_batch_destroy(b);
    return rc;
}

// =====================================================================================================
// batched factor evaluation (CostFunction::Evaluate layout: row-major, global block width, 7th column zero)
// =====================================================================================================
namespace tcv {
__global__ void eval_imu_kernel(int n, const double *consts, const double *params, double G0, double G1, double G2, int use_given,
                                double *sqrt_io, double *res, double *jac, double *work) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double G[3] = {G0, G1, G2};
    const double *pm = params + (size_t)i * 32, *c = consts + (size_t)i * IMU_CONST;
    double *S = sqrt_io + (size_t)i * 225;
    if (!use_given) imu_sqrt_info(c + IMU_COV, S, work + (size_t)i * 450);
    double *w = work + (size_t)i * 450;   // reuse as raw J (15 x 30)
    double rr[15];
    imu_raw(pm, pm + 7, pm + 16, pm + 23, c, G, rr, 1, jac ? w : nullptr, 30);
    for (int r = 0; r < 15; r++) {
        double a = 0;
        for (int s = r; s < 15; s++) a += S[r * 15 + s] * rr[s];
        res[(size_t)i * 15 + r] = a;
    }
    if (!jac) return;
    double *J = jac + (size_t)i * 480;
    const int gofs[4] = {0, 105, 240, 345}, gw[4] = {7, 9, 7, 9}, lc[4] = {0, 6, 15, 21}, lw[4] = {6, 9, 6, 9};
    for (int b = 0; b < 4; b++)
        for (int r = 0; r < 15; r++) {
            for (int cc = 0; cc < lw[b]; cc++) {
                double a = 0;
                for (int s = r; s < 15; s++) a += S[r * 15 + s] * w[s * 30 + lc[b] + cc];
                J[gofs[b] + r * gw[b] + cc] = a;
            }
            if (gw[b] == 7) J[gofs[b] + r * 7 + 6] = 0.0;
        }
}
__global__ void eval_proj_kernel(int n, const double *pts, const double *params, double sqrt_info, double *res, double *jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *pm = params + (size_t)i * 22;
    double r[2], J[38];
    proj_eval(pm, pm + 7, pm + 14, pm[21], pts + (size_t)i * 6, sqrt_info, r, jac ? J : nullptr, 19);
    res[2 * i] = r[0]; res[2 * i + 1] = r[1];
    if (!jac) return;
    double *o = jac + (size_t)i * 44;   // 2x7 | 2x7 | 2x7 | 2x1
    for (int b = 0; b < 3; b++)
        for (int row = 0; row < 2; row++) {
            for (int c = 0; c < 6; c++) o[b * 14 + row * 7 + c] = J[row * 19 + 6 * b + c];
            o[b * 14 + row * 7 + 6] = 0.0;
        }
    o[42] = J[18]; o[43] = J[19 + 18];
}
__global__ void eval_proj_td_kernel(int n, const double *pts, const double *aux, const double *params, double sqrt_info, double TR, double ROW,
                                    double *res, double *jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *pm = params + (size_t)i * 23;
    double r[2], J[40];
    proj_td_eval(pm, pm + 7, pm + 14, pm[21], pm[22], pts + (size_t)i * 6, aux + (size_t)i * 8, sqrt_info, TR, ROW, r, jac ? J : nullptr, 20);
    res[2 * i] = r[0]; res[2 * i + 1] = r[1];
    if (!jac) return;
    double *o = jac + (size_t)i * 46;   // 2x7 | 2x7 | 2x7 | 2x1 | 2x1
    for (int b = 0; b < 3; b++)
        for (int row = 0; row < 2; row++) {
            for (int c = 0; c < 6; c++) o[b * 14 + row * 7 + c] = J[row * 20 + 6 * b + c];
            o[b * 14 + row * 7 + 6] = 0.0;
        }
    o[42] = J[18]; o[43] = J[20 + 18]; o[44] = J[19]; o[45] = J[20 + 19];
}
__global__ void eval_line_kernel(int n, const double *line, const double *consts21, const double *params, double *res, double *jac) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r[2], J[12];
    line_eval(params + (size_t)i * 7, line + (size_t)i * 9, consts21, consts21 + 9, consts21 + 18, r, jac ? J : nullptr, 6);
    res[2 * i] = r[0]; res[2 * i + 1] = r[1];
    if (!jac) return;
    double *o = jac + (size_t)i * 14;
    for (int row = 0; row < 2; row++) { for (int c = 0; c < 6; c++) o[row * 7 + c] = J[row * 6 + c]; o[row * 7 + 6] = 0.0; }
}
__global__ void pose_plus_kernel(int n, const double *x, const double *d, double *o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pose_plus(x + (size_t)i * 7, d + (size_t)i * 6, o + (size_t)i * 7);
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) tcv::dev_free(p); }
    int alloc(size_t bytes) { return tcv::dev_malloc(&p, std::max<size_t>(bytes, 8)) == hipSuccess ? 0 : -1; }
    template <class T> T *as() { return (T *)p; }
};
}  // namespace tcv

#define EVAL_ALLOC(buf, bytes)                                                         \
    if (buf.alloc(bytes)) { set_error("hipMalloc failed in eval"); return TCV_ERR_HIP; }

extern "C" int tcv_eval_imu_factors(int n, const tcv_imu_preintegration *pre, const double *params, const double G[3],
                                    int use_given, double *sqrt_io, double *res, double *jac) {
    if (n <= 0 || !pre || !params || !G || !sqrt_io || !res) return TCV_ERR_INVALID;
    if (int rc = device_ready()) return rc;
    tcv_problem tmp;   // reuse the packer's constant layout
    std::vector<double> consts((size_t)n * IMU_CONST);
    for (int i = 0; i < n; i++) {
        double *D = consts.data() + (size_t)i * IMU_CONST;
        const tcv_imu_preintegration &q = pre[i];
        std::memcpy(D, q.delta_p, 24); std::memcpy(D + 3, q.delta_q, 32); std::memcpy(D + 7, q.delta_v, 24);
        std::memcpy(D + 10, q.linearized_ba, 24); std::memcpy(D + 13, q.linearized_bg, 24); D[16] = q.sum_dt;
        const int rc[5][2] = {{0, 9}, {0, 12}, {3, 12}, {6, 9}, {6, 12}};
        int o = 17;
        for (auto &b : rc) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) D[o++] = q.jacobian[(b[0] + r) * 15 + b[1] + c];
        std::memcpy(D + 62, q.covariance, 225 * 8);
    }
    DevBuf dc, dp, ds, dr, dj, dw;
    EVAL_ALLOC(dc, consts.size() * 8); EVAL_ALLOC(dp, (size_t)n * 32 * 8); EVAL_ALLOC(ds, (size_t)n * 225 * 8);
    EVAL_ALLOC(dr, (size_t)n * 15 * 8); EVAL_ALLOC(dj, (size_t)n * 480 * 8); EVAL_ALLOC(dw, (size_t)n * 450 * 8);
    HIPCHK(hipMemcpy(dc.p, consts.data(), consts.size() * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dp.p, params, (size_t)n * 32 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ds.p, sqrt_io, (size_t)n * 225 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_imu_kernel, dim3((n + 63) / 64), dim3(64), 0, 0, n, dc.as<double>(), dp.as<double>(), G[0], G[1], G[2],
                       use_given, ds.as<double>(), dr.as<double>(), jac ? dj.as<double>() : nullptr, dw.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(sqrt_io, ds.p, (size_t)n * 225 * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(res, dr.p, (size_t)n * 15 * 8, hipMemcpyDeviceToHost));
    if (jac) HIPCHK(hipMemcpy(jac, dj.p, (size_t)n * 480 * 8, hipMemcpyDeviceToHost));
    return TCV_OK;
}
extern "C" int tcv_eval_projection_factors(int n, const double *pts, const double *params, double sqrt_info, double *res, double *jac) {
    if (n <= 0 || !pts || !params || !res) return TCV_ERR_INVALID;
    if (int rc = device_ready()) return rc;
    DevBuf d1, d2, dr, dj;
    EVAL_ALLOC(d1, (size_t)n * 6 * 8); EVAL_ALLOC(d2, (size_t)n * 22 * 8); EVAL_ALLOC(dr, (size_t)n * 2 * 8); EVAL_ALLOC(dj, (size_t)n * 44 * 8);
    HIPCHK(hipMemcpy(d1.p, pts, (size_t)n * 6 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, params, (size_t)n * 22 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_proj_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d1.as<double>(), d2.as<double>(), sqrt_info,
                       dr.as<double>(), jac ? dj.as<double>() : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(res, dr.p, (size_t)n * 2 * 8, hipMemcpyDeviceToHost));
    if (jac) HIPCHK(hipMemcpy(jac, dj.p, (size_t)n * 44 * 8, hipMemcpyDeviceToHost));
    return TCV_OK;
}
extern "C" int tcv_eval_projection_td_factors(int n, const double *pts, const double *aux, const double *params, double sqrt_info,
                                              double TR, double ROW, double *res, double *jac) {
    if (n <= 0 || !pts || !aux || !params || !res || !(ROW > 0)) return TCV_ERR_INVALID;
    if (int rc = device_ready()) return rc;
    DevBuf d1, da, d2, dr, dj;
    EVAL_ALLOC(d1, (size_t)n * 6 * 8); EVAL_ALLOC(da, (size_t)n * 8 * 8); EVAL_ALLOC(d2, (size_t)n * 23 * 8); EVAL_ALLOC(dr, (size_t)n * 2 * 8);
    EVAL_ALLOC(dj, (size_t)n * 46 * 8);
    HIPCHK(hipMemcpy(d1.p, pts, (size_t)n * 6 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(da.p, aux, (size_t)n * 8 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, params, (size_t)n * 23 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_proj_td_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d1.as<double>(), da.as<double>(), d2.as<double>(), sqrt_info,
                       TR, ROW, dr.as<double>(), jac ? dj.as<double>() : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(res, dr.p, (size_t)n * 2 * 8, hipMemcpyDeviceToHost));
    if (jac) HIPCHK(hipMemcpy(jac, dj.p, (size_t)n * 46 * 8, hipMemcpyDeviceToHost));
    return TCV_OK;
}
extern "C" int tcv_eval_line_factors(int n, const double *line, const double K[9], const double R[9], const double T[3],
                                     const double *params, double *res, double *jac) {
    if (n <= 0 || !line || !K || !R || !T || !params || !res) return TCV_ERR_INVALID;
    if (int rc = device_ready()) return rc;
    double c21[21];
    std::memcpy(c21, K, 72); std::memcpy(c21 + 9, R, 72); std::memcpy(c21 + 18, T, 24);
    DevBuf d1, dc, d2, dr, dj;
    EVAL_ALLOC(d1, (size_t)n * 9 * 8); EVAL_ALLOC(dc, 21 * 8); EVAL_ALLOC(d2, (size_t)n * 7 * 8); EVAL_ALLOC(dr, (size_t)n * 2 * 8); EVAL_ALLOC(dj, (size_t)n * 14 * 8);
    HIPCHK(hipMemcpy(d1.p, line, (size_t)n * 9 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dc.p, c21, 21 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, params, (size_t)n * 7 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(eval_line_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d1.as<double>(), dc.as<double>(), d2.as<double>(),
                       dr.as<double>(), jac ? dj.as<double>() : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(res, dr.p, (size_t)n * 2 * 8, hipMemcpyDeviceToHost));
    if (jac) HIPCHK(hipMemcpy(jac, dj.p, (size_t)n * 14 * 8, hipMemcpyDeviceToHost));
    return TCV_OK;
}
extern "C" int tcv_pose_plus(int n, const double *x, const double *delta, double *out) {
    if (n <= 0 || !x || !delta || !out) return TCV_ERR_INVALID;
    if (int rc = device_ready()) return rc;
    DevBuf d1, d2, d3;
    EVAL_ALLOC(d1, (size_t)n * 7 * 8); EVAL_ALLOC(d2, (size_t)n * 6 * 8); EVAL_ALLOC(d3, (size_t)n * 7 * 8);
    HIPCHK(hipMemcpy(d1.p, x, (size_t)n * 7 * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d2.p, delta, (size_t)n * 6 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(pose_plus_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d1.as<double>(), d2.as<double>(), d3.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d3.p, (size_t)n * 7 * 8, hipMemcpyDeviceToHost));
    return TCV_OK;
}

