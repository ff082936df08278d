// Packer: ceres::Problem-shaped graph (estimator.cpp:1679-1886) -> device plan + window data
// (tcv_packed.h).  Everything structural that the reference redoes per frame through
// AddParameterBlock / AddResidualBlock pointer chasing is resolved here, once, into flat gather lists.
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <functional>
#include <deque>
#include <condition_variable>
#include <thread>
#include <tuple>
#include <unordered_map>

#include "tcv_host.h"

namespace tcv {

// ---- host thread budget (HostOp, tcv_packed.h)
static int host_core_grant_probe() {
    int g = (int)std::thread::hardware_concurrency();
    if (g <= 0) g = 1;
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) g = std::min(g, c); }
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {      // cgroup v2: "<quota> <period>" or "max <period>"
        char q[64]; long long per = 0;
        if (fscanf(f, "%63s %lld", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0) { const long long quota = atoll(q); if (quota > 0) g = std::min<long long>(g, std::max<long long>(1, (quota + per - 1) / per)); }
        fclose(f);
    } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {      // cgroup v1
        long long quota = -1, per = 0;
        if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
        fclose(f1);
        if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &per) != 1) per = 0; fclose(f2); }
        if (quota > 0 && per > 0) g = std::min<long long>(g, std::max<long long>(1, (quota + per - 1) / per));
    }
#endif
    if (const char *e = getenv("TCV_HOST_THREADS")) { const int v = atoi(e); if (v > 0) g = v; }
    return std::max(1, g);
}
static int host_core_grant() {
    static const int grant = host_core_grant_probe();      // (once: several host threads ask at the same time)
    return grant;
}
static std::atomic<int> g_host_ops{0};
HostOp::HostOp() { g_host_ops.fetch_add(1, std::memory_order_relaxed); }
HostOp::~HostOp() { g_host_ops.fetch_sub(1, std::memory_order_relaxed); }
int host_threads(int want) {
    const int active = std::max(1, g_host_ops.load(std::memory_order_relaxed));
    return std::max(1, std::min(want, std::max(1, host_core_grant() / active)));
}
int HostOp::threads(int want) const { return host_threads(want); }

// ---- persistent worker threads (parallel_run, tcv_packed.h) ------------------------------------------------------------------
// A batch-level call has three or four short parallel sections (plans, data, marginalisation problems, copies): 16 std::thread
// creations and joins per section were ~2 ms of a 512-window tcv_batch_create.  The workers are created once (up to the core grant
// minus the caller), sleep on a condition variable and claim task indices of the posted calls; the CALLER claims indices too, so a call
// makes progress whatever the workers are busy with, and returns when every index has finished.  The pool object is never destroyed
// (the detached workers may outlive static destruction).
namespace {
struct ParCall {
    const std::function<void(int)> *fn;
    std::function<void(int)> own;      // async_run: the call owns its function (nobody waits for it)
    int n;
    std::atomic<int> next{0}, done{0};
    std::mutex mu;
    std::condition_variable cv;
};
struct WorkerPool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<ParCall>> q;
    int nworkers = 0;
    std::atomic<unsigned> posted{0};      // calls pushed so far: what a worker polls before it goes to sleep
    std::atomic<int> sleepers{0};         // workers inside cv.wait: a post with nobody asleep skips the futex
};
WorkerPool &worker_pool() { static WorkerPool *P = new WorkerPool(); return *P; }
// TCV_WORKER_SPIN_US = t > 0: a worker that has just finished a section polls for the next one for t microseconds before it sleeps on the
// condition variable, and the caller polls for its last task the same way (a lock-step frame is a burst of short parallel sections a few
// microseconds apart).  Measured on the 2 x 64-core host at 8 / 64 / 128 replay streams with t = 40 and 150: the same windows/s within the
// run-to-run spread and 15 - 40 % more CPU time (profiles/r05_replay_host_workers.txt) -- so the default is 0: sleep at once.
int worker_spin_us() {
    static const int us = [] { const char *e = getenv("TCV_WORKER_SPIN_US"); const int v = e ? atoi(e) : 0; return std::max(0, std::min(v, 2000)); }();
    return us;
}
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}
void run_call(ParCall &c) {
    for (;;) {
        const int t = c.next.fetch_add(1, std::memory_order_relaxed);
        if (t >= c.n) return;
        (*c.fn)(t);
        if (c.done.fetch_add(1, std::memory_order_acq_rel) + 1 == c.n) { std::lock_guard<std::mutex> g(c.mu); c.cv.notify_all(); }
    }
}
void worker_main() {
    WorkerPool &P = worker_pool();
    unsigned seen = P.posted.load(std::memory_order_acquire);
    for (;;) {
        std::shared_ptr<ParCall> c;
        {
            std::unique_lock<std::mutex> g(P.mu);
            for (;;) {
                while (!P.q.empty() && P.q.front()->next.load(std::memory_order_relaxed) >= P.q.front()->n) P.q.pop_front();      // fully claimed
                if (!P.q.empty()) { c = P.q.front(); break; }
                seen = P.posted.load(std::memory_order_acquire);
                const int spin = worker_spin_us();
                if (spin > 0) {      // poll outside the lock, then look again
                    g.unlock();
                    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(spin);
                    bool news = false;
                    for (int it = 0;; it++) {
                        if (P.posted.load(std::memory_order_acquire) != seen) { news = true; break; }
                        cpu_relax();
                        if ((it & 63) == 63 && std::chrono::steady_clock::now() >= t_end) break;
                    }
                    g.lock();
                    if (news) continue;
                    if (P.posted.load(std::memory_order_acquire) != seen) continue;
                }
                P.sleepers.fetch_add(1, std::memory_order_relaxed);
                P.cv.wait(g);
                P.sleepers.fetch_sub(1, std::memory_order_relaxed);
            }
        }
        run_call(*c);
    }
}
void wait_call(ParCall &c) {
    const int spin = worker_spin_us();
    if (spin > 0) {
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(spin);
        for (int it = 0;; it++) {
            if (c.done.load(std::memory_order_acquire) >= c.n) return;
            cpu_relax();
            if ((it & 63) == 63 && std::chrono::steady_clock::now() >= t_end) break;
        }
    }
    std::unique_lock<std::mutex> g(c.mu);
    c.cv.wait(g, [&] { return c.done.load(std::memory_order_acquire) >= c.n; });
}
}  // namespace
void parallel_run(int nth, const std::function<void(int)> &fn) {
    if (nth <= 1) { fn(0); return; }
    static const bool off = getenv("TCV_NO_WORKER_POOL") != nullptr;      // A/B: a thread per task, as before
    if (off) {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; t++) th.emplace_back(fn, t);
        for (auto &x : th) x.join();
        return;
    }
    WorkerPool &P = worker_pool();
    auto c = std::make_shared<ParCall>();
    c->fn = &fn; c->n = nth;
    bool wake;
    {
        std::lock_guard<std::mutex> g(P.mu);
        const int want = std::min(31, std::max(1, host_core_grant() - 1));
        while (P.nworkers < std::min(want, nth - 1)) { std::thread(worker_main).detach(); P.nworkers++; }
        P.q.push_back(c);
        P.posted.fetch_add(1, std::memory_order_release);
        wake = P.sleepers.load(std::memory_order_relaxed) > 0;
    }
    if (wake) P.cv.notify_all();
    run_call(*c);
    wait_call(*c);
}

void parallel_items(int n, int nth, const std::function<void(int, int)> &fn) {
    if (n <= 0) return;
    nth = std::min(nth, n);
    if (nth <= 1) { for (int i = 0; i < n; i++) fn(i, 0); return; }
    std::atomic<int> next{0};
    parallel_run(nth, [&](int t) { for (;;) { const int i = next.fetch_add(1, std::memory_order_relaxed); if (i >= n) return; fn(i, t); } });
}

// ---- blocks of the plans' int pools (PlanAlloc, tcv_host.h): power-of-two size classes from 16 KB, a bounded free list per class
namespace {
struct BlockPool { std::mutex mu; std::vector<void *> idle[12]; };      // 16 KB .. 32 MB
BlockPool &block_pool() { static BlockPool *p = new BlockPool(); return *p; }
inline int block_class(size_t bytes, size_t &cap) { int c = 0; cap = (size_t)16 << 10; while (cap < bytes) { cap <<= 1; c++; } return c; }
}  // namespace
void *plan_block_alloc(size_t bytes) {
    if (bytes < ((size_t)16 << 10)) return ::operator new(bytes);
    size_t cap;
    const int c = block_class(bytes, cap);
    if (c < 12) {
        BlockPool &P = block_pool();
        std::lock_guard<std::mutex> g(P.mu);
        if (!P.idle[c].empty()) { void *p = P.idle[c].back(); P.idle[c].pop_back(); return p; }
    }
    return ::operator new(cap);
}
void plan_block_free(void *p, size_t bytes) {
    if (!p) return;
    if (bytes < ((size_t)16 << 10)) { ::operator delete(p); return; }
    size_t cap;
    const int c = block_class(bytes, cap);
    if (c < 12) {
        BlockPool &P = block_pool();
        std::lock_guard<std::mutex> g(P.mu);
        if (P.idle[c].size() < (size_t)(c <= 5 ? 192 : 8)) { P.idle[c].push_back(p); return; }      // (<= 512 KB: the plans of a lock-step frame; larger blocks: a handful)
    }
    ::operator delete(p);
}

// fire and forget on the worker pool (the retired problems of a lock-step frame are destroyed this way: 0.3 ms of the caller's frame
// at 64 windows).  Without workers -- a one-core grant -- the function runs here.
void async_run(std::function<void()> fn) {
    WorkerPool &P = worker_pool();
    bool have_worker;
    {
        std::lock_guard<std::mutex> g(P.mu);
        if (P.nworkers == 0 && host_core_grant() > 1) { std::thread(worker_main).detach(); P.nworkers++; }
        have_worker = P.nworkers > 0;
    }
    if (!have_worker) { fn(); return; }
    auto c = std::make_shared<ParCall>();
    auto sp = std::make_shared<std::function<void()>>(std::move(fn));
    c->own = [sp](int) { (*sp)(); };
    c->fn = &c->own; c->n = 1;
    { std::lock_guard<std::mutex> g(P.mu); P.q.push_back(c); P.posted.fetch_add(1, std::memory_order_release); }
    P.cv.notify_one();
}

// TCV_PRIOR_FULL: keep the exact-zero rows of the prior (A/B partner of the default).  The environment is read once per batch
// (prior_refresh_switch), not once per window.
static std::atomic<int> g_prior_full{-1};
void prior_refresh_switch() { g_prior_full.store(getenv("TCV_PRIOR_FULL") ? 1 : 0, std::memory_order_relaxed); }
bool prior_keep_zero_rows() {
    int v = g_prior_full.load(std::memory_order_relaxed);
    if (v < 0) { prior_refresh_switch(); v = g_prior_full.load(std::memory_order_relaxed); }
    return v != 0;
}

// Two chain-mode workgroups share one CU's 160 KiB of LDS.  TCV_CHAIN_LDS_DOUBLES overrides the per-workgroup size (tuning).
int chain_lds_doubles() {
    static const int v = [] {      // (initialised once, by whichever packer thread comes first: a plain static written by all of them was a data race, found by the ThreadSanitizer build)
        const char *e = getenv("TCV_CHAIN_LDS_DOUBLES");
        int x = e ? atoi(e) : LDS_DOUBLES / 2;
        if (x < 6144 || x > LDS_DOUBLES) x = LDS_DOUBLES / 2;
        return x & ~1;
    }();
    return v;
}

namespace {

struct DestKey {
    int kind, o0, o1;
    bool operator<(const DestKey &o) const { return std::tie(kind, o0, o1) < std::tie(o.kind, o.o0, o.o1); }
};
// Destinations of one gather program with their item lists, in insertion order (deterministic).  Flat storage: the packer builds two of
// these per visual chunk for every cold window, and a hash map plus one vector per destination was most of its time.  Destination ids come
// from a direct-index table over (kind, o0, o1) -- validated by a generation stamp instead of being cleared --, the items are appended as
// (id, item) pairs and bucketed by a stable counting sort in finish().  Every live list owns a table of its own (two lists of one chunk
// are alive together: with a shared table an add() to the first after the second had stamped the same slot would silently open a duplicate
// destination); the tables (0.8 MB each) come from a process-wide pool, so the packing threads a tcv_batch_create starts do not allocate
// and zero them again.
struct DestList {
    std::vector<DestKey> keys;                 // insertion order
    std::vector<std::pair<int, int>> shape;    // (la, lb); lb = 0: triangle of la
    std::vector<int> pair_id, pair_item;       // every add(), in order
    std::vector<int> start, items;             // after finish(): items of destination id = items[start[id] .. start[id + 1])
    enum { T_TILE = 0, T_G = CAM_MAX * CAM_MAX, T_RC = T_G + 256, T_HLL = T_RC + 256, T_HCL = T_HLL + 2048, T_SIZE = T_HCL + 65536 };
    struct Table { std::vector<unsigned> stamp; std::vector<int> val; unsigned gen = 0; Table() : stamp(T_SIZE, 0u), val(T_SIZE, 0) {} };
    struct Pool { std::mutex mu; std::vector<Table *> idle; ~Pool() { for (Table *t : idle) delete t; } };
    static Pool &pool() { static Pool p; return p; }
    Table *tab;
    unsigned gen;
    bool overflow = false;                     // a key outside the table (reported by the caller as a field overflow)
    DestList() : tab(nullptr) {
        {
            Pool &P = pool();
            std::lock_guard<std::mutex> g(P.mu);
            if (!P.idle.empty()) { tab = P.idle.back(); P.idle.pop_back(); }
        }
        if (!tab) tab = new Table();
        Table &t = *tab;
        if (++t.gen == 0) { std::fill(t.stamp.begin(), t.stamp.end(), 0u); t.gen = 1; }
        gen = t.gen;
        keys.reserve(2048); shape.reserve(2048); pair_id.reserve(16384); pair_item.reserve(16384);
    }
    ~DestList() {
        Pool &P = pool();
        std::lock_guard<std::mutex> g(P.mu);
        if (P.idle.size() < 64) { P.idle.push_back(tab); tab = nullptr; }
        delete tab;
    }
    DestList(const DestList &) = delete;
    DestList &operator=(const DestList &) = delete;
    Table &table() { return *tab; }
    void add(int kind, int o0, int o1, int la, int lb, int item) {
        int slot = -1;
        if (kind == DK_TILE) { if (o0 >= 0 && o0 < CAM_MAX && o1 >= 0 && o1 < CAM_MAX) slot = T_TILE + o0 * CAM_MAX + o1; }
        else if (kind == DK_G) { if (o0 >= 0 && o0 < 256) slot = T_G + o0; }
        else if (kind == DK_RC) { if (o0 >= 0 && o0 < 256) slot = T_RC + o0; }
        else if (kind == DK_HLL) { if (o0 >= 0 && o0 < 2048) slot = T_HLL + o0; }
        else if (kind == DK_HCL) { if (o0 >= 0 && o0 < 65536) slot = T_HCL + o0; }
        if (slot < 0) { overflow = true; return; }
        Table &t = table();
        int id;
        if (t.stamp[slot] != gen) {
            id = (int)keys.size();
            t.stamp[slot] = gen; t.val[slot] = id;
            keys.push_back(DestKey{kind, o0, o1});
            shape.emplace_back(la, lb);
        } else id = t.val[slot];
        pair_id.push_back(id); pair_item.push_back(item);
    }
    void finish() {
        const int nd = (int)keys.size();
        start.assign(nd + 1, 0);
        for (int id : pair_id) start[id + 1]++;
        for (int d = 0; d < nd; d++) start[d + 1] += start[d];
        items.resize(pair_id.size());
        std::vector<int> pos(start.begin(), start.end() - 1);
        for (size_t k = 0; k < pair_id.size(); k++) items[pos[pair_id[k]]++] = pair_item[k];
    }
    int count(int id) const { return start[id + 1] - start[id]; }
    const int *first(int id) const { return items.data() + start[id]; }
};

// A ROW UNIT is one row of a destination block: up to `ncols` register accumulators
//   acc[e] += sum_rows rec[row][colA + ea] * rec[row][colB + e]      over the destination's items.
// unit record (UNIT_INTS ints): u0 = kind << 28 | ncols << 24 | ea << 20 | nitems,  u1 = o0 << 16 | o1,  u2 = item_begin
// (IMU units carry their <= 2 items inline instead: u2 = item0, u3 = item1).  Units are sorted by descending item
// count; those with more than WAVE_UNIT_ITEMS items come first and are processed by a whole wavefront each.
// A thread unit costs its wavefront ~40 instructions per item while the other lanes idle, a wave unit ~150 instructions whatever its
// length: the longest lists (above WAVE_UNIT_ITEMS items) go to whole wavefronts, but no more than WAVE_UNIT_MAX of them -- a chunk that
// holds every factor of a large window has dozens of lists of that length, and turning them all into wave units serialises them.
enum { WAVE_UNIT_ITEMS = 32, WAVE_UNIT_MAX = 12 };
struct RowProg {
    std::vector<int> units, items;
    int n_units = 0, n_wave_units = 0;
};
static bool emit_rows(const DestList &dl, RowProg &out, bool inline_items, int wave_items = WAVE_UNIT_ITEMS, int wave_max = WAVE_UNIT_MAX) {
    struct U { int u0, u1, u2, u3, n; };
    std::vector<U> us;
    us.reserve(dl.keys.size() * 4);
    if (!inline_items) out.items.reserve(out.items.size() + dl.items.size());
    if (dl.overflow) return false;
    for (size_t id = 0; id < dl.keys.size(); id++) {
        const DestKey &k = dl.keys[id];
        const int n = dl.count((int)id);
        const int la = dl.shape[id].first, lb = dl.shape[id].second;
        int ib = 0, i0 = 0, i1 = 0;
        if (inline_items) {
            if (n > 2) return false;
            i0 = dl.first((int)id)[0]; i1 = n > 1 ? dl.first((int)id)[1] : 0;
        } else {
            ib = (int)out.items.size();
            out.items.insert(out.items.end(), dl.first((int)id), dl.first((int)id) + n);
            if (ib + n >= (1 << 16) * 16) return false;
        }
        if (k.o0 >= (1 << 16) || k.o1 >= (1 << 16) || n >= (1 << 20)) return false;
        const int nrow = (k.kind == DK_TILE) ? la : 1;
        for (int ea = 0; ea < nrow; ea++) {
            const int ncols = (k.kind == DK_TILE) ? (lb == 0 ? ea + 1 : lb) : la;   // non-tile dests: `la` accumulators
            U u;
            u.u0 = (int)(((unsigned)k.kind << 28) | ((unsigned)ncols << 24) | ((unsigned)ea << 20) | (unsigned)n);
            u.u1 = (int)(((unsigned)k.o0 << 16) | (unsigned)k.o1);
            u.u2 = inline_items ? i0 : ib; u.u3 = i1; u.n = n;
            us.push_back(u);
        }
    }
    std::stable_sort(us.begin(), us.end(), [](const U &a, const U &b) { return a.n > b.n; });
    for (auto &u : us) {
        out.units.push_back(u.u0); out.units.push_back(u.u1); out.units.push_back(u.u2);
        if (inline_items) out.units.push_back(u.u3);
        if (!inline_items && u.n > wave_items && out.n_wave_units < wave_max) out.n_wave_units++;
    }
    out.n_units = (int)us.size();
    return true;
}

// add all pairwise products of one factor's column groups.
// cols: (tangent offset or -1 if constant, column in the record, width)
struct Col { int t, c, w; };
template <class MakeItem>
void add_pairs(DestList &dl, const std::vector<Col> &cols, MakeItem mk, int rcol) {
    for (size_t a = 0; a < cols.size(); a++) {
        if (cols[a].t < 0) continue;
        dl.add(DK_TILE, cols[a].t, cols[a].t, cols[a].w, 0, mk(cols[a].c, cols[a].c));
        dl.add(DK_G, cols[a].t, 0, cols[a].w, 1, mk(rcol, cols[a].c));          // g[t + e] += sum r * J[:, c + e]
        for (size_t b = 0; b < a; b++) {
            if (cols[b].t < 0) continue;
            if (cols[a].t > cols[b].t) dl.add(DK_TILE, cols[a].t, cols[b].t, cols[a].w, cols[b].w, mk(cols[a].c, cols[b].c));
            else dl.add(DK_TILE, cols[b].t, cols[a].t, cols[b].w, cols[a].w, mk(cols[b].c, cols[a].c));
        }
    }
}

}  // namespace

// ---- camera half of a plan -------------------------------------------------------------------------------------------------------
// Everything of a plan that is a function of the CAMERA-side structure alone -- block table, IMU factor tables (colours, tangent maps,
// the 40 KB scatter table of the J'J tiles), the prior's column and destination tables, the chain elimination's step records, the
// frame table: a live estimator keeps that structure from frame to frame (11 poses + 11 speed-biases + extrinsic, ten IMU factors, the
// previous frame's prior layout), while its visual half -- which landmark is seen from which frames -- changes with every frame.
// The camera half is the PREFIX of the plan's int pool (so its offsets do not depend on the visual half) and is cached process-wide
// by its structure; a window whose camera structure is known copies 65 KB instead of rebuilding it (symbolic elimination included).
namespace {
struct ChainStep { int cam, t0; std::vector<int> prow_t; bool next; int nsrc, f[2], lc[2]; };
struct CamPlan {
    bool eligible = true;        // want_chain: the speed-bias blocks form a chain (false: the caller asks again for the dense layout)
    PlanHdr hdr;                 // the camera fields: nblk, nc, nx, npp, nt, ntp, n_imu, prior_*, n_imu_chunk, flags, td_cam, camw, chain, n_e, nt_c, c_spill,
                                 // n_frames and the o_* / n_* of the tables below
    std::vector<int> ints;       // [blk | imu | prior | pcol | idest | iunit | iitem | ichunk | pdest | chain | frames], a multiple of 4 ints
    std::vector<int> loff;       // tangent offset of every camera block (-1 constant)
    std::vector<int> goff;
};
struct CamKey {
    std::vector<int> k;
    size_t h = 0;
    void seal() { unsigned long long x = 1469598103934665603ull; for (int v : k) { x ^= (unsigned)v; x *= 1099511628211ull; } h = (size_t)x; }
    bool operator==(const CamKey &o) const { return h == o.h && k == o.k; }
};
struct CamKeyHash { size_t operator()(const CamKey &k) const { return k.h; } };
std::mutex g_cam_mu;
std::unordered_map<CamKey, std::shared_ptr<const CamPlan>, CamKeyHash> g_cam_cache;
long long g_cam_hits = 0, g_cam_misses = 0;
enum { CAM_CACHE_MAX = 512 };

// what the camera half is built from (filled by pack_plan's classification)
struct CamIn {
    const tcv_problem *p;
    const std::vector<int> *cam_block, *cam_of;
    int td_blk;
    bool want_chain;
    int per_imu;                 // IMU factors per staging chunk (a function of the LDS pool the visual half leaves: part of the key)
};

int build_cam(const CamIn &in, CamPlan &out) {
    const tcv_problem &p = *in.p;
    const std::vector<int> &cam_block = *in.cam_block, &cam_of = *in.cam_of;
    const int nblk = (int)cam_block.size();
    const int td_blk = in.td_blk;
    std::vector<int> gsize(nblk), goff(nblk), loff(nblk, -1), kind(nblk);
    int nx = 0, nc = 0;
    for (int c = 0; c < nblk; c++) {
        const ParamBlock &pb = p.blocks[cam_block[c]];
        gsize[c] = pb.size; kind[c] = pb.kind; goff[c] = nx; nx += pb.size;
        if (pb.kind == KIND_EUCLID && pb.size > 15) { set_error("Euclidean block wider than 15"); return TCV_ERR_UNSUPPORTED; }
    }
    for (int c = 0; c < nblk; c++)
        if (kind[c] == KIND_POSE && !p.blocks[cam_block[c]].constant) { loff[c] = nc; nc += 6; }
    // Td comes right behind the poses: its column rides through the 6-wide gather machinery as a pseudo block whose other five
    // columns are structural zeros, so five more tangent rows must follow it.  It counts as part of the "pose part" npp: the
    // leading tangent dims the visual factors touch (landmark Schur corrections of the diagonal and the right-hand side).
    const int td_cam = td_blk >= 0 ? cam_of[td_blk] : -1;
    if (td_cam >= 0 && !p.blocks[td_blk].constant) { loff[td_cam] = nc; nc += 1; }
    const int npp = nc;
    for (int c = 0; c < nblk; c++)
        if (kind[c] != KIND_POSE && c != td_cam && !p.blocks[cam_block[c]].constant) { loff[c] = nc; nc += gsize[c]; }
    if (nc < 1) { set_error("no free camera-side parameter block"); return TCV_ERR_INVALID; }
    const int nt = (nc + 1 + 15) / 16, ntp = (npp + 15) / 16;
    if (td_cam >= 0 && loff[td_cam] >= 0 && loff[td_cam] + 6 > nt * 16) { set_error("Td block: no room for its gather slot"); return TCV_ERR_UNSUPPORTED; }
    // camera tangent dims: 171 for the 11 frames + extrinsic of OptimizationWithLine (172 with Td), 177 with the relocalisation pose (:1854-1886)
    if (nc > CAM_MAX - 1 || npp > 88) { set_error("window too large for the fused solver (camera tangent dim > 183)"); return TCV_ERR_TOO_LARGE; }
    const int camw = nc <= CAM_W - 1 ? (int)CAM_W : (int)CAM_MAX;
    out.loff = loff; out.goff = goff;

    // ---- chain layout: the free Euclidean camera blocks (speed-biases, 9 wide) only meet their IMU neighbours and the
    // prior, so they are eliminated one after the other BEFORE the dense pose system (block-sparse Cholesky with the poses
    // ordered last, what SPARSE_SCHUR's reduced-camera factorisation exploits too).  Symbolic elimination at block level:
    // eligible iff every Euclidean block has at most one later-eliminated Euclidean neighbour and that one is next in order.
    std::vector<ChainStep> chain;
    bool use_chain = in.want_chain;
    std::vector<int> eorder;
    if (use_chain) {
        std::vector<char> in_prior(nblk, 0);
        if (!p.prior.empty()) for (int b : p.prior[0].b) if (cam_of[b] >= 0) in_prior[cam_of[b]] = 1;
        // (Td is no chain block: it belongs to the pose part, one column wide -- every point factor and so every pose meets it)
        for (int c = nblk - 1; c >= 0; c--) if (kind[c] != KIND_POSE && c != td_cam && loff[c] >= 0 && !in_prior[c]) eorder.push_back(c);
        for (int c = nblk - 1; c >= 0; c--) if (kind[c] != KIND_POSE && c != td_cam && loff[c] >= 0 && in_prior[c]) eorder.push_back(c);
        for (int c : eorder) if (gsize[c] != CH_W) use_chain = false;
        if (eorder.empty() || eorder.size() > 16 || npp < 1) use_chain = false;
    }
    if (use_chain) {
        const int ne = (int)eorder.size();
        std::vector<int> pos(nblk, -1);
        for (int s2 = 0; s2 < ne; s2++) pos[eorder[s2]] = s2;
        std::vector<char> adj((size_t)nblk * nblk, 0);
        auto link = [&](const int *bs, int nb2) { for (int i = 0; i < nb2; i++) for (int j = 0; j < nb2; j++) { const int a2 = bs[i], b2 = bs[j]; if (a2 != b2 && a2 >= 0 && b2 >= 0 && loff[a2] >= 0 && loff[b2] >= 0) adj[(size_t)a2 * nblk + b2] = 1; } };
        for (auto &f : p.imu) { const int bs[4] = {cam_of[f.b[0]], cam_of[f.b[1]], cam_of[f.b[2]], cam_of[f.b[3]]}; link(bs, 4); }
        if (!p.prior.empty()) { std::vector<int> bs; for (int b : p.prior[0].b) bs.push_back(cam_of[b]); link(bs.data(), (int)bs.size()); }
        for (int s2 = 0; s2 < ne && use_chain; s2++) {
            const int e = eorder[s2];
            ChainStep st;
            st.cam = e; st.t0 = loff[e]; st.next = false; st.nsrc = 0; st.f[0] = st.f[1] = 0; st.lc[0] = st.lc[1] = 0;
            std::vector<int> later;
            for (int x = 0; x < nblk; x++) if (adj[(size_t)e * nblk + x] && pos[x] > s2) later.push_back(x);
            if (later.size() > 1 || (later.size() == 1 && pos[later[0]] != s2 + 1)) { use_chain = false; break; }
            st.next = !later.empty();
            std::vector<int> prow;
            for (int x = 0; x < nblk; x++) if (adj[(size_t)e * nblk + x] && (kind[x] == KIND_POSE || x == td_cam)) prow.push_back(x);
            std::sort(prow.begin(), prow.end(), [&](int a2, int b2) { return loff[a2] < loff[b2]; });
            for (int x : prow) for (int j = 0; j < (x == td_cam ? 1 : 6); j++) st.prow_t.push_back(loff[x] + j);
            // fill: the eliminated block's neighbours become a clique (pose-pose is dense anyway)
            for (int x : later) for (int y : prow) { adj[(size_t)x * nblk + y] = 1; adj[(size_t)y * nblk + x] = 1; }
            for (size_t k = 0; k < p.imu.size(); k++)
                for (int sl = 1; sl < 4; sl += 2)
                    if (cam_of[p.imu[k].b[sl]] == e) {
                        if (st.nsrc >= 2) { use_chain = false; break; }
                        st.f[st.nsrc] = (int)k; st.lc[st.nsrc] = sl == 1 ? 6 : 21; st.nsrc++;
                    }
            if (CH_W + (st.next ? CH_W : 0) + (int)st.prow_t.size() + 1 > CH_MAXROWS) use_chain = false;
            chain.push_back(st);
        }
    }
    if (in.want_chain && !use_chain) { out.eligible = false; return TCV_OK; }
    // (Td's gather slot is six columns wide, tcv_packed.h: the pose tiles have to cover the five structural zeros behind its column)
    const int nt_c = (std::max(npp + 1, (td_cam >= 0 && loff[td_cam] >= 0) ? loff[td_cam] + 6 : 0) + 15) / 16;

    PlanHdr &H = out.hdr;
    std::memset(&H, 0, sizeof(H));
    H.nblk = nblk; H.nc = nc; H.nx = nx; H.npp = npp; H.nt = nt; H.ntp = ntp;
    H.n_imu = (int)p.imu.size();
    H.flags = td_blk >= 0 ? 1 : 0; H.td_cam = td_cam; H.camw = camw;
    H.chain = use_chain ? 1 : 0; H.n_e = (int)chain.size(); H.nt_c = nt_c;
    std::vector<int> &I = out.ints;
    I.clear();
    auto mark = [&]() { return (int)I.size(); };
    H.o_blk = mark();
    for (int c = 0; c < nblk; c++) { I.push_back(gsize[c]); I.push_back(goff[c]); I.push_back(loff[c]); I.push_back(kind[c]); }
    H.o_imu = mark();
    for (auto &f : p.imu) for (int k = 0; k < 4; k++) I.push_back(cam_of[f.b[k]]);

    // ---- prior
    if (p.prior.size() > 1) { set_error("more than one marginalisation factor"); return TCV_ERR_UNSUPPORTED; }
    const tcv_prior *pr = p.prior.empty() ? nullptr : p.prior[0].prior;
    H.o_prior = mark();
    std::vector<int> pcol;
    if (pr) {
        if (pr->n > 128) { set_error("prior with more than 128 rows"); return TCV_ERR_TOO_LARGE; }
        H.prior_n = pr->n; H.prior_nblk = (int)pr->size.size(); H.prior_xsize = pr->xsize;
        pcol.assign(pr->n, -1);
        for (int k = 0; k < H.prior_nblk; k++) {
            const int b = p.prior[0].b[k];
            if (cam_of[b] < 0) { set_error("prior attached to a landmark block"); return TCV_ERR_UNSUPPORTED; }
            const ParamBlock &pb = p.blocks[b];
            if (pb.size != pr->size[k]) { set_error("prior block size mismatch"); return TCV_ERR_INVALID; }
            I.push_back(cam_of[b]); I.push_back(pr->idx[k]); I.push_back(pr->size[k]); I.push_back(pr->xoff[k]);
            const int local = pr->size[k] == 7 ? 6 : pr->size[k];
            const int t = loff[cam_of[b]];
            for (int j = 0; j < local; j++)
                if (pr->idx[k] + j < pr->n) pcol[pr->idx[k] + j] = t < 0 ? -1 : t + j;
        }
    }
    H.o_pcol = mark();
    for (int v : pcol) I.push_back(v);

    std::vector<int> imap;
    // ---- IMU chunks.  Per factor: the tangent index of each of its 30 local Jacobian columns (-1 for a constant
    // block) and a colour; factors of one colour share no parameter block, so their J'J tiles can be scattered into the
    // reduced camera system concurrently (the frame chain needs two colours).
    {
        const int per = std::max(1, std::min(H.n_imu, in.per_imu));
        if (H.n_imu > 16) { set_error("more than 16 IMU factors"); return TCV_ERR_TOO_LARGE; }
        std::vector<int> icolor(H.n_imu, 0), ichunk;
        for (int fb = 0; fb < H.n_imu; fb += per) {
            const int fn = std::min(per, H.n_imu - fb);
            int ncol = 0;
            for (int k = 0; k < fn; k++) {       // greedy colouring inside the chunk
                const ImuFac &f = p.imu[fb + k];
                for (int a2 = 0; a2 < 4; a2++)
                    for (int b2 = 0; b2 < a2; b2++)
                        if (f.b[a2] == f.b[b2]) { set_error("IMU factor uses one block twice"); return TCV_ERR_UNSUPPORTED; }
                int col = 0;
                for (;; col++) {
                    bool clash = false;
                    for (int j = 0; j < k && !clash; j++) {
                        if (icolor[fb + j] != col) continue;
                        for (int a2 = 0; a2 < 4; a2++) for (int b2 = 0; b2 < 4; b2++) if (p.imu[fb + j].b[a2] == f.b[b2]) clash = true;
                    }
                    if (!clash) break;
                }
                icolor[fb + k] = col;
                ncol = std::max(ncol, col + 1);
            }
            if (ncol > 4) { set_error("IMU factors of one chunk need more than 4 colours"); return TCV_ERR_UNSUPPORTED; }
            unsigned bits = 0;
            for (int k = 0; k < fn; k++) bits |= (unsigned)icolor[fb + k] << (2 * k);      // 2 bits per factor
            ichunk.push_back(fb); ichunk.push_back(fn); ichunk.push_back(ncol); ichunk.push_back((int)bits);
        }
        for (auto &f : p.imu) {
            const int colc[4] = {0, 6, 15, 21}, colw[4] = {6, 9, 6, 9};
            int m[32];
            for (int i = 0; i < 32; i++) m[i] = -1;
            for (int s2 = 0; s2 < 4; s2++) {
                const int t = loff[cam_of[f.b[s2]]];
                for (int j = 0; j < colw[s2]; j++) m[colc[s2] + j] = t < 0 ? -1 : t + j;
            }
            imap.insert(imap.end(), m, m + 32);
        }
        H.n_imu_chunk = (int)ichunk.size() / 4;
        H.o_idest = mark(); I.insert(I.end(), imap.begin(), imap.end()); H.n_idest = (int)imap.size();
        H.o_iunit = mark(); I.insert(I.end(), icolor.begin(), icolor.end()); H.n_iunit = (int)icolor.size();
        // scatter table of the J'J tiles (see IMU_SC_BIAS): what the kernel would derive from the tangent map for every lane and register
        while ((I.size() & 3) != 0) I.push_back(0);      // read with 16-byte loads
        H.o_iitem = mark(); H.n_iitem = H.n_imu * 1024;
        I.resize(I.size() + (size_t)H.n_imu * 1024, 0);
        int *sct = I.data() + H.o_iitem;
        for (int f = 0; f < H.n_imu; f++) {
            const int *tm = imap.data() + (size_t)f * 32;
            for (int lane = 0; lane < 64; lane++) {
                const int i16 = lane & 15, k4 = lane >> 4;
                for (int tile = 0; tile < 4; tile++) {     // (I, J): (0,0) (1,0) (1,1) (0,1)
                    const int Ir = (tile == 1 || tile == 2) ? 1 : 0, Jc = (tile >= 2) ? 1 : 0;
                    const int bl = 16 * Jc + i16;
                    for (int i = 0; i < 4; i++) {
                        const int al = 16 * Ir + k4 + 4 * i;
                        const int ta = tm[al], tb = tm[bl];
                        int d = -1, store = 0;
                        if (ta >= 0) {
                            if (bl == 30) d = -2 - ta;
                            else if (tile != 3 && bl < 30 && al >= bl && tb >= 0) {
                                store = use_chain ? 1 : 0;
                                if (use_chain && (ta >= npp || tb >= npp)) d = (al == bl) ? -1000 - (ta - npp) : -1;
                                else d = ta >= tb ? tix(ta, tb) : tix(tb, ta);
                            }
                        }
                        if (d + IMU_SC_BIAS < 0 || d + IMU_SC_BIAS >= (1 << 16)) { set_error("IMU scatter destination out of range"); return TCV_ERR_TOO_LARGE; }
                        sct[(size_t)f * 1024 + lane * 16 + tile * 4 + i] = (d + IMU_SC_BIAS) | (store ? IMU_SC_STORE : 0);
                    }
                }
            }
        }
        H.o_ichunk = mark(); I.insert(I.end(), ichunk.begin(), ichunk.end());
    }
    {
        // destinations of the constant part of the prior (Hp, packed lower triangle): same derivation as the kernel's former inline one
        H.o_pdest = mark();
        const int pn = (int)pcol.size();
        I.reserve(I.size() + (size_t)pn * (pn + 1) / 2 + 4096);
        for (int a2 = 0; a2 < pn; a2++)
            for (int b2 = 0; b2 <= a2; b2++) {
                const int ta = pcol[a2], tb = pcol[b2];
                int d = -1;
                if (ta >= 0 && tb >= 0) {
                    if (use_chain && (ta >= npp || tb >= npp)) { if (ta == tb) d = -2 - (ta - npp); }
                    else d = ta >= tb ? tix(ta, tb) : tix(tb, ta);
                }
                I.push_back(d);
            }
    }
    // ---- chain step tables
    while ((I.size() & 3) != 0) I.push_back(0);
    H.o_chain = mark();
    if (use_chain) {
        const int ne = (int)chain.size();
        std::vector<int> tab((size_t)ne * CH_STRIDE, 0);
        std::vector<int> pinv(nc + 1, -1);
        for (size_t j = 0; j < pcol.size(); j++) if (pcol[j] >= 0) pinv[pcol[j]] = (int)j;
        const int wstride = 16 * nt_c * CH_W;      // W rows of one step in the spill area: 16 nt_c columns (whole tiles) x 9
        std::vector<std::vector<int>> rowt(ne);
        for (int s2 = 0; s2 < ne; s2++) {
            const ChainStep &st = chain[s2];
            std::vector<int> &rt = rowt[s2];
            for (int j = 0; j < CH_W; j++) rt.push_back(st.t0 + j);
            if (st.next) for (int j = 0; j < CH_W; j++) rt.push_back(chain[s2 + 1].t0 + j);
            for (int t : st.prow_t) rt.push_back(t);
            rt.push_back(-2);
        }
        for (int s2 = 0; s2 < ne; s2++) {
            const ChainStep &st = chain[s2];
            const std::vector<int> &rt = rowt[s2];
            const int nr = (int)rt.size();
            int *h = tab.data() + (size_t)s2 * CH_STRIDE;
            h[CH_T0] = st.t0; h[CH_R] = nr - CH_W - 1; h[CH_NEXT] = st.next ? 1 : 0; h[CH_NSRC] = st.nsrc;
            h[CH_F0] = st.f[0]; h[CH_LC0] = st.lc[0]; h[CH_F1] = st.f[1]; h[CH_LC1] = st.lc[1];
            h[CH_PC0] = pinv[st.t0]; h[CH_SPILL] = s2 * wstride;
            unsigned char *colrow = reinterpret_cast<unsigned char *>(h + CH_COLROW);
            for (int c = 0; c < CH_MAXROWS; c++) colrow[c] = 255;
            int tmask = 0;
            for (int r = 0; r < nr; r++) {
                int vn = 255;
                if (st.next) {
                    const std::vector<int> &nx2 = rowt[s2 + 1];
                    for (size_t q = 0; q < nx2.size(); q++) if (nx2[q] == rt[r]) vn = (int)q;
                    if (r >= CH_W && vn == 255) { set_error("chain: fill row missing in the next front"); return TCV_ERR_INVALID; }
                }
                int l01[2] = {255, 255};
                for (int src = 0; src < st.nsrc; src++)
                    if (rt[r] >= 0)
                        for (int l = 0; l < 30; l++) if (imap[(size_t)st.f[src] * 32 + l] == rt[r]) l01[src] = l;
                const int tr = rt[r] >= 0 ? rt[r] : 255;
                if (tr > 254 && rt[r] >= 0) { set_error("chain: tangent index overflow"); return TCV_ERR_TOO_LARGE; }
                h[CH_INTS + 2 * r] = tr | (vn << 8) | (l01[0] << 16) | (l01[1] << 24);
                h[CH_INTS + 2 * r + 1] = rt[r] >= 0 ? pinv[rt[r]] : -1;
                // pose rows (tangent index < npp) and the rhs row are the columns of this step's W
                const int col = rt[r] == -2 ? npp : ((rt[r] >= 0 && rt[r] < npp) ? rt[r] : -1);
                if (col >= 0) {
                    if (col >= CH_MAXROWS) { set_error("chain: pose column overflow"); return TCV_ERR_TOO_LARGE; }
                    colrow[col] = (unsigned char)r;
                    tmask |= 1 << (col >> 4);
                }
            }
            h[CH_TMASK] = tmask;
        }
        const int spill = ne * wstride;
        H.c_spill = (spill + 1) & ~1;
        I.insert(I.end(), tab.begin(), tab.end());
    }
    // ---- frame table (gauge fix, estimator.cpp:1537-1581)
    H.n_frames = (int)p.frame_pose.size();
    H.o_frames = mark();
    for (int i = 0; i < H.n_frames; i++) {
        const int bp = p.frame_pose[i], bs = i < (int)p.frame_sb.size() ? p.frame_sb[i] : -1;
        I.push_back(bp >= 0 && cam_of[bp] >= 0 ? goff[cam_of[bp]] : -1);
        I.push_back(bs >= 0 && cam_of[bs] >= 0 ? goff[cam_of[bs]] : -1);
    }
    while ((I.size() & 3) != 0) I.push_back(0);
    return TCV_OK;
}

void cam_key(const CamIn &in, CamKey &key) {
    const tcv_problem &p = *in.p;
    const std::vector<int> &cam_block = *in.cam_block, &cam_of = *in.cam_of;
    std::vector<int> &k = key.k;
    k.clear();
    k.reserve(64 + cam_block.size() + 4 * p.imu.size());
    k.push_back(in.want_chain ? 1 : 0); k.push_back(in.per_imu); k.push_back(in.td_blk >= 0 ? cam_of[in.td_blk] : -1); k.push_back((int)cam_block.size());
    for (int b : cam_block) { const ParamBlock &pb = p.blocks[b]; k.push_back(pb.size | (pb.kind << 8) | ((int)pb.constant << 16)); }
    k.push_back((int)p.imu.size());
    for (auto &f : p.imu) for (int j = 0; j < 4; j++) k.push_back(cam_of[f.b[j]]);
    k.push_back((int)p.prior.size());
    for (auto &f : p.prior) {
        const tcv_prior *pr = f.prior;
        k.push_back(pr->n); k.push_back((int)pr->size.size()); k.push_back(pr->xsize);
        for (size_t j = 0; j < f.b.size(); j++) { k.push_back(cam_of[f.b[j]]); k.push_back(pr->size[j]); k.push_back(pr->idx[j]); k.push_back(pr->xoff[j]); }
    }
    k.push_back((int)p.frame_pose.size());
    for (int v : p.frame_pose) k.push_back(v >= 0 ? cam_of[v] : -1);
    k.push_back((int)p.frame_sb.size());
    for (int v : p.frame_sb) k.push_back(v >= 0 ? cam_of[v] : -1);
    key.seal();
}

// the camera half for this structure: out of the cache, or built and put there
int get_cam(const CamIn &in, std::shared_ptr<const CamPlan> &out, bool use_cache) {
    CamKey key;
    if (use_cache) {
        cam_key(in, key);
        std::lock_guard<std::mutex> g(g_cam_mu);
        auto it = g_cam_cache.find(key);
        if (it != g_cam_cache.end()) { out = it->second; g_cam_hits++; return TCV_OK; }
        g_cam_misses++;
    }
    auto N = std::make_shared<CamPlan>();
    const int rc = build_cam(in, *N);
    if (rc != TCV_OK) return rc;
    out = N;
    if (use_cache) {
        std::lock_guard<std::mutex> g(g_cam_mu);
        if (g_cam_cache.size() >= CAM_CACHE_MAX) g_cam_cache.clear();
        g_cam_cache.emplace(std::move(key), N);
    }
    return TCV_OK;
}

// ---- gather program of one visual chunk, fast path -----------------------------------------------------------------------------------
// Same program as DestList + emit_rows above produce (the reference path, TCV_PACK_REF=1; tests/test_pack_cpu.py compares the two int by
// int): destinations in order of first appearance, the items of a destination in factor order, units sorted by descending item count
// (stable).  Two passes over the factors with a direct-index table of a few hundred slots (block ORDINALS instead of tangent offsets:
// every pose-kind block is 6 wide) -- a count pass that also records the order of first appearance, and a fill pass --, then a
// counting sort of the units.  ~8x faster than the generic path on a replay window (500 point factors, 90 landmarks).
struct ProgScratch { std::vector<int> cnt, start, order, ukey; };
ProgScratch &prog_scratch() { thread_local ProgScratch s; return s; }

// slots of the table: [TILE 16 x 16 ordinals | G / RC 16 | per landmark (HLL) | per landmark slot (HCL)]
enum { PS_TILE = 0, PS_G = 256, PS_LM = 272 };

// writes [units (3 ints each) | items] behind `prog` (which the caller has aligned); n_units / n_wave_units / n_items describe them
struct ProgOut { int n_units = 0, n_wave_units = 0, n_items = 0; };
template <class Walk, class DestOf>
bool fast_prog(int nslot, Walk &&walk, DestOf &&dest_of, std::vector<int> &prog, ProgOut &po, int wave_items, int wave_max) {
    ProgScratch &S = prog_scratch();
    S.cnt.assign(nslot, 0); S.order.clear();
    // pass A: counts + order of first appearance
    walk([&](int slot, int) { if (S.cnt[slot]++ == 0) S.order.push_back(slot); });
    const int nd = (int)S.order.size();
    S.start.resize(nslot);
    int total = 0, nu = 0, nmax = 0;
    struct D { int kind, o0, o1, la, lb; };
    static thread_local std::vector<D> ds;
    ds.resize(nd);
    for (int d = 0; d < nd; d++) {
        const int sl = S.order[d];
        S.start[sl] = total; total += S.cnt[sl];
        D &q = ds[d];
        dest_of(sl, q.kind, q.o0, q.o1, q.la, q.lb);
        if (q.o0 >= (1 << 16) || q.o1 >= (1 << 16) || S.cnt[sl] >= (1 << 20)) return false;
        nu += (q.kind == DK_TILE) ? q.la : 1;
        nmax = std::max(nmax, S.cnt[sl]);
    }
    if ((size_t)total >= (size_t)(1 << 16) * 16) return false;
    const size_t off = prog.size();
    prog.resize(off + 3 * (size_t)nu + total);
    int *un = prog.data() + off, *items = un + 3 * (size_t)nu;
    // pass B: fill (the count doubles as the cursor: start[slot] advances, the unit records below use start - cnt)
    walk([&](int slot, int item) { items[S.start[slot]++] = item; });
    // units in destination order, placed by a stable counting sort on descending item count
    std::vector<int> &bk = S.ukey;
    bk.assign(nmax + 2, 0);
    for (int d = 0; d < nd; d++) bk[nmax - S.cnt[S.order[d]] + 1] += (ds[d].kind == DK_TILE) ? ds[d].la : 1;
    for (int b = 0; b <= nmax; b++) bk[b + 1] += bk[b];
    int nw = 0;
    for (int d = 0; d < nd; d++) {
        const int sl = S.order[d], n = S.cnt[sl], ib = S.start[sl] - n;
        const D &q = ds[d];
        const int nrow = (q.kind == DK_TILE) ? q.la : 1;
        const int u1 = (int)(((unsigned)q.o0 << 16) | (unsigned)q.o1);
        int pos = bk[nmax - n];
        bk[nmax - n] += nrow;
        for (int ea = 0; ea < nrow; ea++, pos++) {
            const int ncols = (q.kind == DK_TILE) ? (q.lb == 0 ? ea + 1 : q.lb) : q.la;
            un[3 * pos] = (int)(((unsigned)q.kind << 28) | ((unsigned)ncols << 24) | ((unsigned)ea << 20) | (unsigned)n);
            un[3 * pos + 1] = u1; un[3 * pos + 2] = ib;
        }
        if (n > wave_items) nw += nrow;
    }
    po.n_units = nu; po.n_wave_units = std::min(nw, wave_max); po.n_items = total;
    return true;
}

}  // namespace

void cam_cache_stats(long long *hits, long long *misses) {
    std::lock_guard<std::mutex> g(g_cam_mu);
    if (hits) *hits = g_cam_hits;
    if (misses) *misses = g_cam_misses;
}

// developer profile of the packer (TCV_DEBUG_PACK2=1): time per phase, summed over all threads; printed and cleared by pack_laps_print()
namespace {
std::mutex g_lap_mu;
std::map<std::string, std::pair<double, long long>> g_laps;
void pack_lap_add(const char *what, double us) { std::lock_guard<std::mutex> g(g_lap_mu); auto &e = g_laps[what]; e.first += us; e.second++; }
}  // namespace
void pack_laps_print() {
    std::lock_guard<std::mutex> g(g_lap_mu);
    for (auto &kv : g_laps) fprintf(stderr, "[pack] %-32s %8.1f us avg over %lld\n", kv.first.c_str(), kv.second.first / std::max<long long>(1, kv.second.second), kv.second.second);
    g_laps.clear();
}

// tcv_set_packer_reference (diagnostics): 1 = every plan is built by the generic path and nothing is cached
static std::atomic<int> g_pack_reference{getenv("TCV_PACK_REF") ? 1 : 0};
void set_pack_reference(int on) { g_pack_reference.store(on ? 1 : 0, std::memory_order_relaxed); }
bool pack_reference() { return g_pack_reference.load(std::memory_order_relaxed) != 0; }

// structural half: plan header, int pool and the host-side maps (everything that does not depend on the VALUES of the window)
// coop_chunks > 0: plan for the cooperative kernel (tcv_packed.h COOP_*): at least that many visual chunks of at most 256 point factors
// each (one helper workgroup per chunk, one lane per factor), and an LDS budget that leaves room for a helper's second tile set
static int pack_plan(const tcv_problem &p, Packed &out, int mode, int chain_lds, int coop_chunks) {
    const int nb = (int)p.blocks.size();
    static const bool dbg_t = getenv("TCV_DEBUG_PACK2") != nullptr;      // developer: where the symbolic packing spends its time
    const bool ref_path = g_pack_reference.load(std::memory_order_relaxed) != 0;      // the generic gather-program builder (DestList + emit_rows) and no camera-half cache: the reference of tests/test_pack_cpu.py
    static const bool no_cache = getenv("TCV_NO_CAM_CACHE") != nullptr;      // (TCV_NO_PLAN_CACHE switches the whole-plan cache off, this one the camera halves')
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg_t) { const auto t = std::chrono::steady_clock::now(); pack_lap_add(what, std::chrono::duration<double, std::micro>(t - t_prev).count()); t_prev = std::chrono::steady_clock::now(); } };
    // ---- classify blocks: landmarks = size-1 Euclidean blocks used only as 4th block of projection factors
    std::vector<int> use_other(nb, 0);
    for (auto &f : p.proj) { for (int k = 0; k < 3; k++) use_other[f.b[k]]++; if (f.btd >= 0) use_other[f.btd]++; }
    // ProjectionTdFactor (ESTIMATE_TD): all point factors or none, one shared 1-dim Td block
    int td_blk = -1;
    for (size_t k = 0; k < p.proj.size(); k++) {
        const ProjFac &f = p.proj[k];
        if (k == 0) td_blk = f.btd;
        else if (f.btd != td_blk) { set_error("projection factors must all be ProjectionTdFactors on one Td block, or none"); return TCV_ERR_UNSUPPORTED; }
    }
    if (td_blk >= 0) {
        const ParamBlock &pb = p.blocks[td_blk];
        if (pb.size != 1 || pb.kind != KIND_EUCLID) { set_error("Td must be a size-1 Euclidean block"); return TCV_ERR_UNSUPPORTED; }
    }
    for (auto &f : p.imu) for (int k = 0; k < 4; k++) use_other[f.b[k]]++;
    for (auto &f : p.line) use_other[f.b]++;
    for (auto &f : p.prior) for (int b : f.b) use_other[b]++;
    std::vector<int> lm_of(nb, -1), cam_of(nb, -1);
    out.cam_block.clear(); out.lm_block.clear();
    // landmark numbering = order of first appearance in the projection factors (feature_index, estimator.cpp:1743)
    for (auto &f : p.proj) {
        const int b = f.b[3];
        const ParamBlock &pb = p.blocks[b];
        if (pb.size != 1 || pb.kind != KIND_EUCLID || use_other[b] || pb.constant) {
            set_error("projection factor: 4th block must be a free size-1 inverse-depth block used by projection factors only");
            return TCV_ERR_UNSUPPORTED;
        }
        if (lm_of[b] < 0) { lm_of[b] = (int)out.lm_block.size(); out.lm_block.push_back(b); }
    }
    for (int b = 0; b < nb; b++)
        if (lm_of[b] < 0) { cam_of[b] = (int)out.cam_block.size(); out.cam_block.push_back(b); }
    const int nblk = (int)out.cam_block.size(), L = (int)out.lm_block.size();
    for (auto &f : p.proj)
        for (int k = 0; k < 3; k++)
            if (p.blocks[f.b[k]].kind != KIND_POSE || p.blocks[f.b[k]].size != 7) {
                set_error("projection factor: first three blocks must be pose blocks");
                return TCV_ERR_UNSUPPORTED;
            }
    for (auto &f : p.imu)
        for (int k = 0; k < 4; k++) {
            const ParamBlock &pb = p.blocks[f.b[k]];
            const bool ok = (k % 2 == 0) ? (pb.kind == KIND_POSE && pb.size == 7) : (pb.kind == KIND_EUCLID && pb.size == 9);
            if (!ok) { set_error("IMU factor: blocks must be pose(7), speed-bias(9), pose(7), speed-bias(9)"); return TCV_ERR_UNSUPPORTED; }
        }
    for (auto &f : p.line)
        if (p.blocks[f.b].kind != KIND_POSE) { set_error("line factor: block must be a pose"); return TCV_ERR_UNSUPPORTED; }

    // ---- sizes of the camera side (the tables themselves come from the camera half below)
    int nx = 0, nc = 0, npose_free = 0;
    for (int c = 0; c < nblk; c++) {
        const ParamBlock &pb = p.blocks[out.cam_block[c]];
        nx += pb.size;
        if (pb.kind == KIND_POSE && !pb.constant) npose_free++;
    }
    const int td_cam = td_blk >= 0 ? cam_of[td_blk] : -1;
    const int td_free = (td_cam >= 0 && !p.blocks[td_blk].constant) ? 1 : 0;
    const int npp = 6 * npose_free + td_free;
    nc = npp;
    for (int c = 0; c < nblk; c++) { const ParamBlock &pb = p.blocks[out.cam_block[c]]; if (pb.kind != KIND_POSE && c != td_cam && !pb.constant) nc += pb.size; }
    if (nc < 1) { set_error("no free camera-side parameter block"); return TCV_ERR_INVALID; }
    const int nt = (nc + 1 + 15) / 16, ntp = (npp + 15) / 16;
    const int ntiles = nt * (nt + 1) / 2, pp_tiles = ntp * (ntp + 1) / 2;
    if (nc > CAM_MAX - 1 || npp > 88 || nc + L > SCR_NL || L > 1024) { set_error("window too large for the fused solver (camera tangent dim > 183)"); return TCV_ERR_TOO_LARGE; }
    const int camw = nc <= CAM_W - 1 ? (int)CAM_W : (int)CAM_MAX;
    const int nxl = (nx + L + 1) & ~1;
    int area_cap = LDS_DOUBLES - ntiles * 256 - 2 * nxl - (3 * camw + 176) - 64;
    int stage_cap = (ntiles - pp_tiles) * 256;
    // (the dense layout holds the whole camera system in LDS tiles: 12 tile rows -- a window with the relocalisation pose -- do not fit; such a
    // window needs the chain layout, decided below)
    const bool dense_fits = area_cap >= 512;
    if (!dense_fits && mode != 0) { set_error("window too large for the fused solver (LDS)"); return TCV_ERR_TOO_LARGE; }
    const int td_t_pre = td_free ? 6 * npose_free : -1;
    const int nt_c = (std::max(npp + 1, td_t_pre >= 0 ? td_t_pre + 6 : 0) + 15) / 16, ctiles = nt_c * (nt_c + 1) / 2;
    const int c_vec = 2 * nxl + (3 * camw + 176) + 64 + 112;
    const int c_lds = coop_chunks > 0 ? (LDS_DOUBLES - ctiles * 256 - 8) : ((chain_lds >= 6144 && chain_lds <= LDS_DOUBLES) ? (chain_lds & ~1) : chain_lds_doubles());
    const int c_pool = c_lds - ctiles * 256 - c_vec;
    const int n_imu = (int)p.imu.size();

    // ---- the camera half: chain layout if the graph allows it and its LDS pool holds the chain's working set
    CamIn cin;
    cin.p = &p; cin.cam_block = &out.cam_block; cin.cam_of = &cam_of; cin.td_blk = td_blk;
    std::shared_ptr<const CamPlan> cam;
    bool use_chain = (mode == 0);
    auto cam_for = [&](bool chain) -> int {
        const int imu_cap = chain ? c_pool : area_cap;       // chain mode: the whole pool holds IMU records
        if (n_imu > 0 && imu_cap < IMU_REC) { if (chain) return 1; set_error("no LDS room for IMU staging"); return TCV_ERR_TOO_LARGE; }
        cin.want_chain = chain; cin.per_imu = std::max(1, std::min(n_imu, imu_cap / IMU_REC));
        return get_cam(cin, cam, !ref_path && !no_cache);
    };
    if (use_chain) {
        const int rc = cam_for(true);
        if (rc < 0) return rc;
        if (rc == 1 || !cam->eligible || c_pool < chain_pool_doubles(cam->hdr.n_e, nt_c) || c_pool < IMU_REC) use_chain = false;
    }
    if (!use_chain) {
        if (!dense_fits) { set_error("window too large for the fused solver (LDS; its speed-bias blocks do not form a chain either)"); return TCV_ERR_TOO_LARGE; }
        const int rc = cam_for(false);
        if (rc != TCV_OK) return rc;
    }
    const std::vector<int> *loffp = &cam->loff;
    lap("classification + camera half");

    const int prec = td_blk >= 0 ? (int)PROJ_TD_REC : (int)PROJ_REC;      // doubles per staged point record
    const int td_t = td_cam >= 0 ? (*loffp)[td_cam] : -1;

    // ---- projection factors sorted by landmark (stable), landmark slots
    const int nproj = (int)p.proj.size(), nline = (int)p.line.size();
    std::vector<int> &order = out.proj_order;
    order.resize(nproj);
    std::vector<int> lmptr(L + 1, 0);
    for (auto &f : p.proj) lmptr[lm_of[f.b[3]] + 1]++;
    for (int l = 0; l < L; l++) lmptr[l + 1] += lmptr[l];
    {      // stable counting sort by landmark
        std::vector<int> pos(lmptr.begin(), lmptr.end() - 1);
        for (int i = 0; i < nproj; i++) order[pos[lm_of[p.proj[i].b[3]]]++] = i;
    }
    // slots of a landmark = the distinct free pose-kind blocks (and Td) it is seen from, in order of first appearance: flat arrays, and per
    // sorted factor the slot of each of its columns (-1: constant block)
    const int ncol_f = td_t >= 0 ? 4 : 3;
    std::vector<int> slotptr(L + 1, 0), slot_t;
    std::vector<signed char> fslot((size_t)nproj * 4, -1);
    std::vector<int> ft((size_t)nproj * 4, -1);      // per sorted factor: tangent offset of each column group (-1 constant)
    slot_t.reserve((size_t)L * 8);
    std::vector<int> e_off(L + 1, 0);
    for (int l = 0; l < L; l++) {
        const int s0 = (int)slot_t.size();
        slotptr[l] = s0;
        for (int k = lmptr[l]; k < lmptr[l + 1]; k++) {
            const ProjFac &f = p.proj[order[k]];
            for (int s = 0; s < ncol_f; s++) {
                const int t = s < 3 ? (*loffp)[cam_of[f.b[s]]] : td_t;
                ft[(size_t)k * 4 + s] = t;
                if (t < 0) continue;
                int q = s0;
                const int s1 = (int)slot_t.size();
                while (q < s1 && slot_t[q] != t) q++;
                if (q == s1) slot_t.push_back(t);
                fslot[(size_t)k * 4 + s] = (signed char)(q - s0);
            }
        }
        const int ns = (int)slot_t.size() - s0;
        if (ns > 40) { set_error("landmark observed from more than 40 blocks"); return TCV_ERR_TOO_LARGE; }
        if (e_off[l] >= (1 << 16) - 256) { set_error("landmark coupling store too large"); return TCV_ERR_TOO_LARGE; }
        e_off[l + 1] = e_off[l] + 6 * ns + 2;   // + 1/kappa and gl/kappa behind the slice
    }
    slotptr[L] = (int)slot_t.size();
    // (the reference order of the slots: Td is appended behind the three pose slots of every factor, as above)
    if (e_off[L] > (1 << 16) - 256) { set_error("landmark coupling store too large"); return TCV_ERR_TOO_LARGE; }
    for (int k = 0; k < nproj; k++) {
        const int *t4 = ft.data() + (size_t)k * 4;
        for (int a2 = 0; a2 < ncol_f; a2++) for (int b2 = 0; b2 < a2; b2++)
            if (t4[a2] >= 0 && t4[a2] == t4[b2]) { set_error("projection factor uses one block twice"); return TCV_ERR_UNSUPPORTED; }
    }
    std::vector<int> line_t(nline);
    for (int k = 0; k < nline; k++) line_t[k] = (*loffp)[cam_of[p.line[k].b]];

    // ---- visual chunks (whole landmarks; lines ride in chunk 0).  Staging holds the factor records AND the chunk's
    // gather program (units + items), so both count against the capacity.
    struct VChunk { int pb, pn, lb, ln, lmb, lmn; };
    std::vector<VChunk> vch;
    auto build_chunks = [&](int stage_cap, int area_cap, std::vector<VChunk> &vch) -> int {
        vch.clear();
        const int npose = npp / 6;
        const int npb = npose + (td_t >= 0 ? 1 : 0);
        const int base_prog = 3 * (npb * (npb + 1) / 2 * 6 + npb);      // upper bound on tile + gradient units
        auto need = [&](int recs, int nf, int nl_, int slots, int nln) {
            const int ints = base_prog + 3 * (slots + nl_) + (td_t >= 0 ? 19 : 13) * nf + 2 * nln;      // items per point factor: 6 (10) block pairs + 3 (4) gradients + 3 (4) landmark couplings + 1
            return ((recs + 1) & ~1) + (ints + 1) / 2 + 8;
        };
        if (need(nline * LINE_REC, 0, 0, 0, nline) > stage_cap) { set_error("too many line factors for LDS staging"); return TCV_ERR_TOO_LARGE; }
        // whole landmarks per chunk, greedily; the line factors ride in the LAST chunk (the least full one) if they fit
        VChunk cur{0, 0, 0, 0, 0, 0};
        int recs = 0, hcl = 0, nf_c = 0, slots_c = 0;
        for (int l = 0; l < L; l++) {
            const int nf = lmptr[l + 1] - lmptr[l], ns = slotptr[l + 1] - slotptr[l], nh = 6 * ns + 2;
            if (need(nf * prec, nf, 1, ns, 0) > stage_cap || nh + 3 > area_cap) { set_error("landmark track too long for LDS staging"); return TCV_ERR_TOO_LARGE; }
            if (need(recs + nf * prec, nf_c + nf, cur.lmn + 1, slots_c + ns, 0) > stage_cap || hcl + nh + 3 * (cur.lmn + 1) > area_cap) {
                vch.push_back(cur);
                cur = VChunk{lmptr[l], 0, 0, 0, l, 0};
                recs = 0; hcl = 0; nf_c = 0; slots_c = 0;
            }
            cur.pn += nf; cur.lmn += 1; recs += nf * prec; hcl += nh; nf_c += nf; slots_c += ns;
        }
        if (nline > 0 && need(recs + nline * LINE_REC, nf_c, cur.lmn, slots_c, nline) > stage_cap) {
            vch.push_back(cur);
            cur = VChunk{lmptr[L], 0, 0, 0, L, 0};
        }
        cur.ln = nline;
        vch.push_back(cur);
        return TCV_OK;
    };
    static const int wu_v = getenv("TCV_WU_V") ? atoi(getenv("TCV_WU_V")) : (int)WAVE_UNIT_ITEMS, wu_s = getenv("TCV_WU_S") ? atoi(getenv("TCV_WU_S")) : (int)WAVE_UNIT_ITEMS,
                     wu_max = getenv("TCV_WU_MAX") ? atoi(getenv("TCV_WU_MAX")) : (int)WAVE_UNIT_MAX;      // tuning experiments
    // the two gather programs of one chunk: the generic builder (reference) ...
    auto chunk_progs_ref = [&](const VChunk &c, RowProg &vp, RowProg &sp) -> int {
        DestList dl, sl;
        std::vector<Col> cols;
        cols.reserve(4);
        for (int k = 0; k < c.pn; k++) {
            const ProjFac &f = p.proj[order[c.pb + k]];
            const int l = lm_of[f.b[3]];
            const int base = k * prec;
            if (base >= (1 << 21)) { set_error("staging offset overflow"); return TCV_ERR_TOO_LARGE; }
            auto mk = [&](int ca, int cb) { return (int)(((unsigned)base << 11) | ((unsigned)ca << 6) | ((unsigned)cb << 1)); };
            cols.clear();
            for (int s2 = 0; s2 < 3; s2++) cols.push_back(Col{(*loffp)[cam_of[f.b[s2]]], 6 * s2, 6});
            if (td_t >= 0) cols.push_back(Col{td_t, 20, 6});      // [td | 5 zero columns]
            add_pairs(dl, cols, mk, 19);
            for (auto &cc : cols) {
                if (cc.t < 0) continue;
                const int *sb = slot_t.data() + slotptr[l], *se = slot_t.data() + slotptr[l + 1];
                const int slot = (int)(std::find(sb, se, cc.t) - sb);
                dl.add(DK_HCL, e_off[l] + 6 * slot, 0, 6, 1, mk(18, cc.c));      // Hcl[e] += sum J[:,18] * J[:, c + e]
            }
            dl.add(DK_HLL, l, 0, 2, 1, mk(18, 18));                              // acc[0] = hll, acc[1] = gl (columns 18, 19)
        }
        for (int k = 0; k < c.ln; k++) {
            const LineFac &f = p.line[c.lb + k];
            const int base = c.pn * prec + k * LINE_REC;
            if (base >= (1 << 21)) { set_error("staging offset overflow"); return TCV_ERR_TOO_LARGE; }
            auto mk = [&](int ca, int cb) { return (int)(((unsigned)base << 11) | ((unsigned)ca << 6) | ((unsigned)cb << 1) | 1u); };
            cols.clear();
            cols.push_back(Col{(*loffp)[cam_of[f.b]], 0, 6});
            add_pairs(dl, cols, mk, 6);
        }
        for (int l = c.lmb; l < c.lmb + c.lmn; l++) {
            const int *sl_t = slot_t.data() + slotptr[l];
            const int nsl = slotptr[l + 1] - slotptr[l];
            for (int a2 = 0; a2 < nsl; a2++) {
                const unsigned hoff = (unsigned)(e_off[l] - e_off[c.lmb]), ns = (unsigned)nsl;
                auto mk = [&](int sa, int sb) { return (int)((hoff << 18) | (ns << 12) | ((unsigned)sa << 6) | (unsigned)sb); };
                sl.add(DK_TILE, sl_t[a2], sl_t[a2], 6, 0, mk(a2, a2));
                sl.add(DK_RC, sl_t[a2], 0, 6, 1, mk(63, a2));                // rc[t + e] += gl/kappa * Hcl[slot][e]
                for (int b2 = 0; b2 < a2; b2++) {
                    if (sl_t[a2] > sl_t[b2]) sl.add(DK_TILE, sl_t[a2], sl_t[b2], 6, 6, mk(a2, b2));
                    else sl.add(DK_TILE, sl_t[b2], sl_t[a2], 6, 6, mk(b2, a2));
                }
            }
        }
        dl.finish(); sl.finish();
        if (!emit_rows(dl, vp, false, wu_v, wu_max) || !emit_rows(sl, sp, false, wu_s, wu_max)) { set_error("gather program field overflow"); return TCV_ERR_TOO_LARGE; }
        return TCV_OK;
    };
    // ... and the fast one (same programs, int by int)
    auto chunk_progs_fast = [&](const VChunk &c, std::vector<int> &vprog, std::vector<int> &sprog, ProgOut &vp, ProgOut &sp) -> int {
        if ((c.pn > 0 && (long long)(c.pn - 1) * prec >= (1 << 21)) || (c.ln > 0 && (long long)c.pn * prec + (long long)(c.ln - 1) * LINE_REC >= (1 << 21))) { set_error("staging offset overflow"); return TCV_ERR_TOO_LARGE; }
        const int sp0 = slotptr[c.lmb];
        const int PS_HCL = PS_LM + c.lmn, nslot = PS_HCL + (slotptr[c.lmb + c.lmn] - sp0);
        const int colc[4] = {0, 6, 12, 20};
        auto vis_walk = [&](auto add) {
            for (int k = 0; k < c.pn; k++) {
                const int kk = c.pb + k;
                const int l = lm_of[p.proj[order[kk]].b[3]];
                const int *t4 = ft.data() + (size_t)kk * 4;
                const unsigned base = (unsigned)(k * prec) << 11;
                for (int a2 = 0; a2 < ncol_f; a2++) {
                    if (t4[a2] < 0) continue;
                    const int oa = t4[a2] / 6, ca = colc[a2];
                    add(PS_TILE + oa * 16 + oa, (int)(base | ((unsigned)ca << 6) | ((unsigned)ca << 1)));
                    add(PS_G + oa, (int)(base | (19u << 6) | ((unsigned)ca << 1)));
                    for (int b2 = 0; b2 < a2; b2++) {
                        if (t4[b2] < 0) continue;
                        const int ob = t4[b2] / 6, cb = colc[b2];
                        if (t4[a2] > t4[b2]) add(PS_TILE + oa * 16 + ob, (int)(base | ((unsigned)ca << 6) | ((unsigned)cb << 1)));
                        else add(PS_TILE + ob * 16 + oa, (int)(base | ((unsigned)cb << 6) | ((unsigned)ca << 1)));
                    }
                }
                for (int a2 = 0; a2 < ncol_f; a2++) {
                    if (t4[a2] < 0) continue;
                    add(PS_HCL + (slotptr[l] - sp0) + fslot[(size_t)kk * 4 + a2], (int)(base | (18u << 6) | ((unsigned)colc[a2] << 1)));
                }
                add(PS_LM + (l - c.lmb), (int)(base | (18u << 6) | (18u << 1)));
            }
            for (int k = 0; k < c.ln; k++) {
                const int t = line_t[c.lb + k];
                if (t < 0) continue;
                const unsigned base = (unsigned)(c.pn * prec + k * LINE_REC) << 11;
                const int o = t / 6;
                add(PS_TILE + o * 16 + o, (int)(base | 1u));
                add(PS_G + o, (int)(base | (6u << 6) | 1u));
            }
        };
        auto vis_dest = [&](int sl, int &kind, int &o0, int &o1, int &la, int &lb) {
            if (sl < PS_G) { const int oa = sl >> 4, ob = sl & 15; kind = DK_TILE; o0 = 6 * oa; o1 = 6 * ob; la = 6; lb = oa == ob ? 0 : 6; }
            else if (sl < PS_LM) { kind = DK_G; o0 = 6 * (sl - PS_G); o1 = 0; la = 6; lb = 1; }
            else if (sl < PS_HCL) { kind = DK_HLL; o0 = c.lmb + (sl - PS_LM); o1 = 0; la = 2; lb = 1; }
            else {
                const int q = sp0 + (sl - PS_HCL);      // global slot index: its landmark by binary search in slotptr
                const int l = (int)(std::upper_bound(slotptr.begin() + c.lmb, slotptr.begin() + c.lmb + c.lmn + 1, q) - slotptr.begin()) - 1;
                kind = DK_HCL; o0 = e_off[l] + 6 * (q - slotptr[l]); o1 = 0; la = 6; lb = 1;
            }
        };
        if (!fast_prog(nslot, vis_walk, vis_dest, vprog, vp, wu_v, wu_max)) { set_error("gather program field overflow"); return TCV_ERR_TOO_LARGE; }
        auto sch_walk = [&](auto add) {
            for (int l = c.lmb; l < c.lmb + c.lmn; l++) {
                const int *sl_t = slot_t.data() + slotptr[l];
                const int nsl = slotptr[l + 1] - slotptr[l];
                const unsigned hdr = ((unsigned)(e_off[l] - e_off[c.lmb]) << 18) | ((unsigned)nsl << 12);
                for (int a2 = 0; a2 < nsl; a2++) {
                    const int oa = sl_t[a2] / 6;
                    add(PS_TILE + oa * 16 + oa, (int)(hdr | ((unsigned)a2 << 6) | (unsigned)a2));
                    add(PS_G + oa, (int)(hdr | (63u << 6) | (unsigned)a2));
                    for (int b2 = 0; b2 < a2; b2++) {
                        const int ob = sl_t[b2] / 6;
                        if (sl_t[a2] > sl_t[b2]) add(PS_TILE + oa * 16 + ob, (int)(hdr | ((unsigned)a2 << 6) | (unsigned)b2));
                        else add(PS_TILE + ob * 16 + oa, (int)(hdr | ((unsigned)b2 << 6) | (unsigned)a2));
                    }
                }
            }
        };
        auto sch_dest = [&](int sl, int &kind, int &o0, int &o1, int &la, int &lb) {
            if (sl < PS_G) { const int oa = sl >> 4, ob = sl & 15; kind = DK_TILE; o0 = 6 * oa; o1 = 6 * ob; la = 6; lb = oa == ob ? 0 : 6; }
            else { kind = DK_RC; o0 = 6 * (sl - PS_G); o1 = 0; la = 6; lb = 1; }
        };
        if (!fast_prog(PS_LM, sch_walk, sch_dest, sprog, sp, wu_s, wu_max)) { set_error("gather program field overflow"); return TCV_ERR_TOO_LARGE; }
        return TCV_OK;
    };
    // the fast path indexes blocks by ordinal = tangent offset / 6: every slot of the pose part is 6 wide (Td: its 6-wide gather slot) and
    // there are at most 16 of them (npp <= 88); anything else takes the generic path
    const bool fast_ok = !ref_path && npp <= 90;
    // gather programs of a chunk list; measures the staging / area doubles the largest chunk needs
    auto emit_all = [&](const std::vector<VChunk> &vch, int stage_chk, int area_chk, std::vector<int> &vprog, std::vector<int> &sprog,
                        std::vector<int> &vchunk_tab, int &max_stage, int &max_area) -> int {
        vprog.clear(); sprog.clear(); vchunk_tab.clear(); max_stage = 0; max_area = 0;
        for (auto &c : vch) {
            while (vprog.size() & 3) vprog.push_back(0);        // every program starts 16-byte aligned (vector loads on the device)
            while (sprog.size() & 3) sprog.push_back(0);
            const int voff = (int)vprog.size(), soff = (int)sprog.size();
            ProgOut vp, sp;
            if (fast_ok) { const int rcp = chunk_progs_fast(c, vprog, sprog, vp, sp); if (rcp != TCV_OK) return rcp; }
            else {
                RowProg rv, rs;
                const int rcp = chunk_progs_ref(c, rv, rs);
                if (rcp != TCV_OK) return rcp;
                vprog.insert(vprog.end(), rv.units.begin(), rv.units.end()); vprog.insert(vprog.end(), rv.items.begin(), rv.items.end());
                sprog.insert(sprog.end(), rs.units.begin(), rs.units.end()); sprog.insert(sprog.end(), rs.items.begin(), rs.items.end());
                vp.n_units = rv.n_units; vp.n_wave_units = rv.n_wave_units; vp.n_items = (int)rv.items.size();
                sp.n_units = rs.n_units; sp.n_wave_units = rs.n_wave_units; sp.n_items = (int)rs.items.size();
            }
            if (getenv("TCV_DEBUG_UNITS")) {
                auto dump = [](const char *nm, const ProgOut &rp, const int *un) {
                    fprintf(stderr, "[pack] %s: %d units (%d wave units), items per unit (descending):", nm, rp.n_units, rp.n_wave_units);
                    for (int u = 0; u < rp.n_units; u += (u < 16 ? 1 : 16)) fprintf(stderr, " %d", un[3 * u] & 0xfffff);
                    fprintf(stderr, "\n");
                };
                dump("visual", vp, vprog.data() + voff); dump("schur", sp, sprog.data() + soff);
            }
            const int recs = (c.pn * prec + c.ln * LINE_REC + 1) & ~1;
            const int st_need = std::max(recs + (3 * vp.n_units + vp.n_items + 1) / 2 + 2, (3 * sp.n_units + sp.n_items + 1) / 2 + 2);
            const int ar_need = (e_off[c.lmb + c.lmn] - e_off[c.lmb]) + 3 * c.lmn + 8;
            max_stage = std::max(max_stage, st_need); max_area = std::max(max_area, ar_need);
            if (stage_chk >= 0 && (st_need > stage_chk || ar_need > area_chk)) {
                set_error("gather program does not fit the LDS staging area"); return TCV_ERR_TOO_LARGE;
            }
            const int tab[16] = {c.pb, c.pn, c.lb, c.ln, voff, vp.n_units, vp.n_wave_units, vp.n_items,
                                 c.lmb, c.lmn, e_off[c.lmb], e_off[c.lmb + c.lmn] - e_off[c.lmb],
                                 soff, sp.n_units, sp.n_wave_units, sp.n_items};
            vchunk_tab.insert(vchunk_tab.end(), tab, tab + 16);
        }
        return TCV_OK;
    };
    lap("landmark slots");
    static thread_local std::vector<int> vprog, sprog, vchunk_tab;      // (reused: a plan per window per frame)
    if (use_chain) {
        // chain layout: the LDS pool (staging | landmark coupling area) is small, so the landmarks are dealt evenly to the
        // smallest number k of chunks whose EXACT programs fit; the line factors are spread over the chunks
        bool found = false;
        // lower bound on k from the records alone, then jump by the measured overshoot: two exact trials instead of k
        const int kmax = std::max(1, std::min(L, 48));
        int k = std::max(1, (nproj * prec + nline * LINE_REC + c_pool - 1) / std::max(1, c_pool));
        if (const char *ek = getenv("TCV_VIS_CHUNKS")) k = std::max(k, atoi(ek));      // tuning experiments
        k = std::max(k, std::min(coop_chunks, kmax));
        for (; k <= kmax && !found;) {
            // point factors: an even split, except that a boundary a few factors above a multiple of 64 is pulled down to it -- one lane
            // evaluates one factor, so 64 + 68 + 68 factors cost five wavefront passes of the evaluation and 64 + 64 + 72 cost four.  Line
            // factors: dealt so that the chunks' record volumes even out (the chunk with more point factors gets fewer lines).
            std::vector<VChunk> cand;
            int l = 0, lines_left = nline, lb_next = 0;
            const long long rec_even = ((long long)nproj * prec + (long long)nline * LINE_REC) / k;
            for (int c = 0; c < k; c++) {
                VChunk cur{lmptr[l], 0, lb_next, 0, l, 0};
                int target = (int)((long long)(c + 1) * nproj / k);
                if (target % 64 <= 8 && target >= 64) target -= target % 64;
                while (l < L && (c == k - 1 || lmptr[l + 1] <= target || cur.lmn == 0)) { cur.pn += lmptr[l + 1] - lmptr[l]; cur.lmn++; l++; }
                int ln_c = c == k - 1 ? lines_left : (int)std::max<long long>(0, std::min<long long>(lines_left, (rec_even - (long long)cur.pn * prec + LINE_REC / 2) / LINE_REC));
                cur.ln = ln_c; lb_next += ln_c; lines_left -= ln_c;
                cand.push_back(cur);
            }
            int ms = 0, ma = 0;
            if (coop_chunks > 0 && k < kmax) {      // one lane per point factor: a cooperative chunk holds at most one pass of a 256-thread helper
                bool wide = false;
                for (auto &c : cand) wide = wide || c.pn > 256;
                if (wide) { k++; continue; }
            }
            if (emit_all(cand, -1, -1, vprog, sprog, vchunk_tab, ms, ma) != TCV_OK) { k++; continue; }
            ma = (ma + 1) & ~1;
            if (getenv("TCV_DEBUG_PACK")) fprintf(stderr, "[pack] k %d ms %d ma %d pool %d\n", k, ms, ma, c_pool);
            if (ms + ma <= c_pool) { found = true; vch = cand; area_cap = ma; stage_cap = c_pool - ma; }
            else k = std::max(k + 1, std::min(kmax, (int)(((long long)k * (ms + ma) + c_pool - 1) / c_pool)));      // need(k) ~ a / k + b, b > 0: never overshoots the smallest k
        }
        if (!found) {      // the visual half does not fit the chain layout's pool: the dense layout (its camera half, its slot offsets are the same)
            use_chain = false;
            if (!dense_fits) { set_error("window too large for the fused solver (LDS; its speed-bias blocks do not form a chain either)"); return TCV_ERR_TOO_LARGE; }
            const int rc = cam_for(false);
            if (rc != TCV_OK) return rc;
            area_cap = LDS_DOUBLES - ntiles * 256 - 2 * nxl - (3 * camw + 176) - 64;
            stage_cap = (ntiles - pp_tiles) * 256;
        }
    }
    if (!use_chain) {
        // build_chunks sizes the chunks by an estimate of the VISUAL program; the exact programs (emit_all) may need more -- the Schur program of
        // ragged tracks (many camera-block pairs per landmark) does: the chunks are then cut for a smaller budget until the exact programs fit
        int cap_try = stage_cap, rc = TCV_OK;
        std::string first_msg;
        for (int attempt = 0; attempt < 12; attempt++) {
            const int rcb = build_chunks(cap_try, area_cap, vch);
            if (rcb != TCV_OK) { if (attempt == 0) return rcb; rc = TCV_ERR_TOO_LARGE; break; }      // (a single landmark no longer fits the reduced budget)
            int ms = 0, ma = 0;
            rc = emit_all(vch, stage_cap, area_cap, vprog, sprog, vchunk_tab, ms, ma);
            if (rc != TCV_ERR_TOO_LARGE) break;
            if (first_msg.empty()) first_msg = tcv_last_error();
            cap_try = cap_try * 4 / 5;
        }
        if (rc != TCV_OK) { if (!first_msg.empty()) set_error(first_msg); return rc; }
    }
    lap("chunks + gather programs");
    set_error("");

    // ---- the plan: header = the camera half's fields + the window's own, int pool = [camera half | visual half]
    PlanHdr &H = out.hdr;
    H = cam->hdr;
    H.nland = L; H.n_proj = nproj; H.n_line = nline;
    H.hcl_total = e_off[L];
    H.lds_area = area_cap;
    H.c_stage_cap = stage_cap; H.c_area_cap = area_cap; H.c_pool = c_pool;
    H.n_vis_chunk = (int)vch.size();
    PlanInts &I = out.ints;
    I.clear();
    I.reserve(cam->ints.size() + 4 * (size_t)nproj + nline + 3 * (size_t)L + slot_t.size() + vchunk_tab.size() + vprog.size() + sprog.size() + 32);
    I.insert(I.end(), cam->ints.begin(), cam->ints.end());
    auto mark = [&]() { return (int)I.size(); };
    H.o_proj = mark();
    for (int k = 0; k < nproj; k++) {
        const ProjFac &f = p.proj[order[k]];
        for (int s = 0; s < 3; s++) I.push_back(cam_of[f.b[s]]);
        I.push_back(lm_of[f.b[3]]);
    }
    H.o_line = mark();
    for (auto &f : p.line) I.push_back(cam_of[f.b]);
    H.o_lm = mark();
    for (int l = 0; l < L; l++) { I.push_back(e_off[l]); I.push_back(slotptr[l + 1] - slotptr[l]); }
    H.o_lmslotptr = mark();
    for (int l = 0; l <= L; l++) I.push_back(slotptr[l]);
    H.o_lmslot = mark();
    I.insert(I.end(), slot_t.begin(), slot_t.end());
    H.o_vchunk = mark(); I.insert(I.end(), vchunk_tab.begin(), vchunk_tab.end());
    while ((I.size() & 3) != 0) I.push_back(0);
    H.o_vdest = mark(); I.insert(I.end(), vprog.begin(), vprog.end()); H.n_vdest = 0;
    H.o_vunit = H.o_vdest; H.n_vunit = (int)vprog.size(); H.o_vitem = H.o_vdest; H.n_vitem = 0;
    while ((I.size() & 3) != 0) I.push_back(0);
    H.o_sdest = mark(); I.insert(I.end(), sprog.begin(), sprog.end()); H.n_sdest = 0;
    H.o_sunit = H.o_sdest; H.n_sunit = (int)sprog.size(); H.o_sitem = H.o_sdest; H.n_sitem = 0;
    while ((I.size() & 3) != 0) I.push_back(0);          // plans are concatenated: keep every plan 16-byte aligned
    H.plan_ints = (int)I.size();
    out.cam_loff = cam->loff;
    lap("plan assembly");
    return TCV_OK;
}

// data half: the window's doubles in the layout the plan expects (offsets depend on the counts only).  The sink either counts
// (dst == nullptr) or writes: a batch first sizes every window, then all windows are written straight into one upload buffer.
namespace {
struct Sink {
    double *dst;
    size_t n = 0;
    explicit Sink(double *d) : dst(d) {}
    void put(const double *s, size_t k) { if (dst) std::memcpy(dst + n, s, k * sizeof(double)); n += k; }
    void put1(double v) { if (dst) dst[n] = v; n++; }
    void zeros(size_t k) { if (dst) std::memset(dst + n, 0, k * sizeof(double)); n += k; }
};
}  // namespace

static int pack_data_to(const tcv_problem &p, Packed &out, const double *imu_sqrt, Sink &D) {
    const PlanHdr &H = out.hdr;
    const int nblk = (int)out.cam_block.size(), L = (int)out.lm_block.size();
    const std::vector<int> &order = out.proj_order;
    const bool with_td = (H.flags & 1) != 0;
    const tcv_prior *pr = p.prior.empty() ? nullptr : p.prior[0].prior;
    WinHdr &W = out.win;
    const int plan_id = W.plan;
    const long long dbase = W.dbase;
    std::memset(&W, 0, sizeof(W));
    W.plan = plan_id; W.dbase = dbase; W.sqrt_export = -1;
    W.d_x = (int)D.n;
    for (int c = 0; c < nblk; c++) { const ParamBlock &pb = p.blocks[out.cam_block[c]]; D.put(pb.addr, pb.size); }
    for (int l = 0; l < L; l++) D.put1(p.blocks[out.lm_block[l]].addr[0]);
    W.d_imu = (int)D.n;
    // device-resident pre-integrations (tcv_preintegrate_device): when EVERY IMU factor of the window is one, nothing is written here --
    // tcv_batch_create places the n_imu x 287 region in the batch's device-only tail (it patches d_imu) and a splice job per factor fills it.
    // Decided by the sizing pass, like the prior's (a mixed window materialises its device-resident ones on the host).
    bool imu_on_device = out.dev_imu_doubles > 0;
    if (!D.dst) {
        imu_on_device = !p.imu.empty();
        for (auto &f : p.imu) if (!f.dev || !f.dev->dev || f.dev->dev->dev != out.batch_dev) imu_on_device = false;      // (a blob of another device: host path)
    }
    out.dev_imu_doubles = imu_on_device ? (int)p.imu.size() * IMU_CONST : 0;
    for (auto &f : p.imu) {
        if (imu_on_device) break;
        if (f.dev) { if (int rc = tcv_preint_host(f.dev)) return rc; }
        const tcv_imu_preintegration &q = f.dev ? f.dev->pod : f.pre;
        D.put(q.delta_p, 3); D.put(q.delta_q, 4); D.put(q.delta_v, 3); D.put(q.linearized_ba, 3); D.put(q.linearized_bg, 3); D.put1(q.sum_dt);
        const int rc[5][2] = {{0, 9}, {0, 12}, {3, 12}, {6, 9}, {6, 12}};   // dp_dba dp_dbg dq_dbg dv_dba dv_dbg (imu_factor.h:61-79)
        for (auto &b : rc) for (int i = 0; i < 3; i++) D.put(q.jacobian + (b[0] + i) * 15 + b[1], 3);
        D.put(q.covariance, 225);
    }
    W.d_proj = (int)D.n;
    double psi = 0, pla = 0;
    for (size_t k = 0; k < order.size(); k++) {
        const ProjFac &f = p.proj[order[k]];
        if (k == 0) { psi = f.sqrt_info; pla = f.loss_a; }
        else if (f.sqrt_info != psi || f.loss_a != pla) { set_error("projection factors must share sqrt_info and loss"); return TCV_ERR_UNSUPPORTED; }
        D.put(f.pts, 6);
        if (with_td) D.put(f.aux, 8);
    }
    W.d_line = (int)D.n;
    double lla = 0;
    for (size_t k = 0; k < p.line.size(); k++) {
        const LineFac &f = p.line[k];
        if (k == 0) lla = f.loss_a;
        else {
            const LineFac &g = p.line[0];
            if (f.loss_a != lla || std::memcmp(f.K, g.K, sizeof f.K) || std::memcmp(f.R, g.R, sizeof f.R) || std::memcmp(f.T, g.T, sizeof f.T)) {
                set_error("line factors must share K, b_c_R, b_c_T and loss"); return TCV_ERR_UNSUPPORTED;
            }
        }
        D.put(f.d, 9);
    }
    W.d_linec = (int)D.n;
    if (!p.line.empty()) { const LineFac &g = p.line[0]; D.put(g.K, 9); D.put(g.R, 9); D.put(g.T, 3); }
    else D.zeros(21);
    if (D.n & 1) D.put1(0.0);        // J0 is copied with 16-byte loads
    W.d_prior = (int)D.n;
    // (decided by the sizing pass and kept for the writing pass: another thread may materialise the prior on the host in between)
    bool on_device = out.dev_prior_doubles > 0;
    if (!D.dst) {
        on_device = false;
        if (pr) { std::lock_guard<std::mutex> g(pr->mu); on_device = !pr->host && pr->dev && pr->dev->dev == out.batch_dev && (int)pr->size.size() <= PRIOR_SPLICE_MAX_BLOCKS; }
    }
    out.dev_prior_doubles = 0;
    out.prior_k0_deferred = false;
    if (pr && on_device) {
        // device-resident (tcv_batch_get_priors_device): nothing is written here; tcv_batch_create gives the region a place in the batch's
        // device-only tail (it patches d_prior) and the splice kernel fills it with the same layout as below
        // (k0 not known on the host yet -- tcv_batch_get_priors_device_async --: room for every row, the splice kernel compacts by the
        // count it reads on the device and writes it into the window header)
        out.prior_k0_deferred = pr->k0 < 0 && !prior_keep_zero_rows();
        const int n = pr->n, k0 = (prior_keep_zero_rows() || pr->k0 < 0) ? 0 : pr->k0, nr = n - k0;
        W.prior_k0 = k0;
        out.dev_prior_doubles = nr * n + nr + pr->xsize;
    } else if (pr) {      // the rows of the thresholded eigenvalues (exact zeros in J0 and r0) are dropped, tcv_packed.h
        if (int rc = tcv_prior_host(pr)) return rc;
        const int n = pr->n, k0 = prior_keep_zero_rows() ? 0 : prior_zero_rows(pr->J0.data(), pr->r0.data(), n), nr = n - k0;
        W.prior_k0 = k0;
        for (int j = 0; j < n; j++) D.put(pr->J0.data() + (size_t)n * j + k0, nr);
        D.put(pr->r0.data() + k0, nr);
        D.put(pr->x0.data(), pr->x0.size());
    }
    W.d_misc = (int)D.n;
    D.put(p.G, 3); D.put1(psi); D.put1(pla); D.put1(lla); D.put1(p.td_TR); D.put1(p.td_ROW); D.put1(p.line_exact ? 1.0 : 0.0);
    W.d_sqrt = -1;
    if (imu_sqrt && H.n_imu) { W.d_sqrt = (int)D.n; D.put(imu_sqrt, (size_t)225 * H.n_imu); }
    if (D.n & 1) D.put1(0.0);
    W.n_doubles = (int)D.n;
    if (D.dst) for (size_t i = 0; i < D.n; i++) { const double v = D.dst[i]; if (!(v == v) || v > 1e300 || v < -1e300) { set_error("NaN/Inf in window data"); return TCV_ERR_NUMERIC; } }
    return TCV_OK;
}

static int pack_data(const tcv_problem &p, Packed &out, const double *imu_sqrt) {
    Sink cnt(nullptr);
    int rc = pack_data_to(p, out, imu_sqrt, cnt);
    if (rc != TCV_OK) return rc;
    out.doubles.resize(cnt.n);
    Sink w(out.doubles.data());
    return pack_data_to(p, out, imu_sqrt, w);
}

int pack_problem_data(const tcv_problem &p, Packed &out, const double *imu_sqrt, double *dst) {
    Sink w(dst);
    return pack_data_to(p, out, imu_sqrt, w);
}


// ---- plan cache ---------------------------------------------------------------------------------------------------------------
// The plan (symbolic elimination, chunking, gather programs: ~1 ms of host time for a cfg-3 window) is a function of the graph
// structure alone.  Windows of one batch, the frames of a bench / streaming loop and re-solves of the same window share it: the
// structure is serialised into an int key (block sizes / kinds / constness, the block indices of every factor, the prior's block
// layout, the frame table, layout mode and LDS budget), and a process-wide table maps key -> PlanTemplate.
namespace {
// key = the serialised structure with its hash computed once, outside the cache lock
struct PlanKey {
    std::vector<int> k;
    size_t h = 0;
    void seal() {      // four independent multiply chains over 8-byte words (one chain per int was a third of a cache lookup)
        unsigned long long x[4] = {1469598103934665603ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
        const size_t n = k.size();
        size_t i = 0;
        for (; i + 8 <= n; i += 8)
            for (int c = 0; c < 4; c++) { unsigned long long v; std::memcpy(&v, k.data() + i + 2 * c, 8); x[c] = (x[c] ^ v) * 0xFF51AFD7ED558CCDull; x[c] ^= x[c] >> 32; }
        for (; i < n; i++) { x[0] = (x[0] ^ (unsigned)k[i]) * 1099511628211ull; }
        unsigned long long r = x[0];
        for (int c = 1; c < 4; c++) { r = (r ^ x[c]) * 0xFF51AFD7ED558CCDull; r ^= r >> 29; }
        h = (size_t)(r ^ n);
    }
    bool operator==(const PlanKey &o) const { return h == o.h && k == o.k; }
};
struct KeyHash { size_t operator()(const PlanKey &k) const { return k.h; } };
// LRU: a live estimator / replay produces a new structure almost every frame (tracks start and end), so most structures are seen once.
//  * The table is cut into 16 shards by the key's hash, each with its own lock, list and 24 entries (384 plans at most): 8 - 16 packer
//    threads looked up and inserted under ONE lock before, 20 + 12 us per window at 16 threads against 3.5 + 1 us alone.
//  * A structure enters the cache at its SECOND appearance: every shard remembers the hashes of its last 64 misses; a miss whose hash is
//    not among them builds its plan for the caller alone (no template, no eviction, the int pool goes back to the block pool with the
//    batch, still warm), a miss that is builds it again and keeps it.  A bench / streaming loop pays one extra build per structure, a live
//    estimator none of the bookkeeping.  TCV_PLAN_CACHE_EAGER=1: insert at first sight (the behaviour up to round 5).
struct CacheEntry { PlanKey key; std::shared_ptr<const PlanTemplate> tmpl; };
enum { CACHE_SHARDS = 16, CACHE_SHARD_ENTRIES = 24, CACHE_RECENT = 64 };
struct CacheShard {
    std::mutex mu;
    std::list<CacheEntry> lru;
    std::unordered_map<PlanKey, std::list<CacheEntry>::iterator, KeyHash> map;
    size_t recent[CACHE_RECENT] = {0};
    unsigned recent_next = 0;
    char pad[64];
};
CacheShard *g_shards() { static CacheShard *s = new CacheShard[CACHE_SHARDS]; return s; }
inline CacheShard &shard_of(const PlanKey &k) { return g_shards()[(k.h >> 7) & (CACHE_SHARDS - 1)]; }
std::atomic<long long> g_hits{0}, g_misses{0};

void structure_key(const tcv_problem &p, int mode, int chain_lds, int coop_chunks, PlanKey &key) {
    std::vector<int> &k = key.k;
    size_t n = 16 + p.blocks.size() + 4 * p.imu.size() + 5 * p.proj.size() + p.line.size() + p.frame_pose.size() + p.frame_sb.size();
    for (auto &f : p.prior) n += 3 + 4 * f.b.size();
    k.resize(n);
    int *q = k.data();
    *q++ = mode; *q++ = chain_lds; *q++ = coop_chunks; *q++ = (int)p.blocks.size();
    for (auto &b : p.blocks) *q++ = b.size | (b.kind << 8) | ((int)b.constant << 16);
    *q++ = (int)p.imu.size();
    for (auto &f : p.imu) { q[0] = f.b[0]; q[1] = f.b[1]; q[2] = f.b[2]; q[3] = f.b[3]; q += 4; }
    *q++ = (int)p.proj.size();
    for (auto &f : p.proj) { q[0] = f.b[0]; q[1] = f.b[1]; q[2] = f.b[2]; q[3] = f.b[3]; q[4] = f.btd; q += 5; }
    *q++ = (int)p.line.size();
    for (auto &f : p.line) *q++ = f.b;
    *q++ = (int)p.prior.size();
    for (auto &f : p.prior) {
        const tcv_prior *pr = f.prior;
        *q++ = pr->n; *q++ = (int)pr->size.size(); *q++ = pr->xsize;
        for (size_t j = 0; j < f.b.size(); j++) { *q++ = f.b[j]; *q++ = pr->size[j]; *q++ = pr->idx[j]; *q++ = pr->xoff[j]; }
    }
    *q++ = (int)p.frame_pose.size();
    for (int v : p.frame_pose) *q++ = v;
    for (int v : p.frame_sb) *q++ = v;
    k.resize((size_t)(q - k.data()));
    key.seal();
}
}  // namespace

// content hash of a plan (header + int pool): tcv_batch_create gives equal plans of a batch one device copy
unsigned long long plan_content_hash(const PlanHdr &hdr, const PlanInts &pints) {
    unsigned long long h = 0x9E3779B97F4A7C15ull ^ pints.size();
    auto mix = [&](const int *p, size_t cnt) {
        size_t i = 0;
        for (; i + 1 < cnt; i += 2) { unsigned long long v; std::memcpy(&v, p + i, 8); h = (h ^ v) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
        if (i < cnt) { h = (h ^ (unsigned)p[i]) * 0xFF51AFD7ED558CCDull; h ^= h >> 29; }
    };
    mix(pints.data(), pints.size());
    mix(reinterpret_cast<const int *>(&hdr), sizeof(PlanHdr) / sizeof(int));
    return h;
}

void plan_cache_stats(long long *hits, long long *misses, long long *entries) {
    if (hits) *hits = g_hits.load();
    if (misses) *misses = g_misses.load();
    if (entries) {
        long long e = 0;
        for (int i = 0; i < CACHE_SHARDS; i++) { CacheShard &S = g_shards()[i]; std::lock_guard<std::mutex> g(S.mu); e += (long long)S.map.size(); }
        *entries = e;
    }
}

int pack_problem(const tcv_problem &p, Packed &out, const double *imu_sqrt, int mode, int chain_lds, bool plan_only, int coop_chunks) {
    static const bool no_cache_env = getenv("TCV_NO_PLAN_CACHE") != nullptr;
    const bool no_cache = no_cache_env || pack_reference();
    const int c_lds = (chain_lds >= 6144 && chain_lds <= LDS_DOUBLES) ? (chain_lds & ~1) : chain_lds_doubles();
    PlanKey key;
    std::shared_ptr<const PlanTemplate> T;
    std::memset(&out.win, 0, sizeof out.win);
    static const bool dbg_t = getenv("TCV_DEBUG_PACK2") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg_t) { const auto t = std::chrono::steady_clock::now(); pack_lap_add(what, std::chrono::duration<double, std::micro>(t - t_prev).count()); t_prev = std::chrono::steady_clock::now(); } };
    bool keep = false;      // the structure has been seen before (or TCV_PLAN_CACHE_EAGER): its plan goes into the cache
    if (!no_cache) {
        static const bool eager = getenv("TCV_PLAN_CACHE_EAGER") != nullptr;
        structure_key(p, mode, c_lds, coop_chunks, key);
        CacheShard &S = shard_of(key);
        std::lock_guard<std::mutex> g(S.mu);
        auto it = S.map.find(key);
        if (it != S.map.end()) { S.lru.splice(S.lru.begin(), S.lru, it->second); T = it->second->tmpl; g_hits.fetch_add(1, std::memory_order_relaxed); }
        else {
            g_misses.fetch_add(1, std::memory_order_relaxed);
            keep = eager;
            for (int i = 0; i < CACHE_RECENT && !keep; i++) keep = S.recent[i] == key.h;
            if (!keep) { S.recent[S.recent_next] = key.h; S.recent_next = (S.recent_next + 1) % CACHE_RECENT; }
        }
    }
    lap("key + lookup");
    out.key_hashed = false;
    if (T) {
        out.hdr = T->hdr; out.tmpl = T; out.ints.clear();
        out.cam_block = T->cam_block; out.cam_loff = T->cam_loff; out.lm_block = T->lm_block; out.proj_order = T->proj_order;
    } else {
        out.tmpl.reset();
        const int rc = pack_plan(p, out, mode, c_lds, coop_chunks);
        if (rc != TCV_OK) return rc;
        lap("pack_plan");
        // identity for the de-duplication of equal plans inside a batch (tcv_batch_create compares the contents of candidates): equal
        // structure keys give equal plans, so the key's hash serves -- hashing the 170 KB of content cost as much as a third of the build
        const unsigned long long id = (unsigned long long)key.h * 0x9E3779B97F4A7C15ull + (unsigned long long)out.ints.size();
        if (!no_cache && !keep) { out.plan_hash = id; out.key_hashed = true; }
        if (!no_cache && keep) {
            auto N = std::make_shared<PlanTemplate>();
            N->hdr = out.hdr; N->ints.swap(out.ints);
            N->cam_block = out.cam_block; N->cam_loff = out.cam_loff; N->lm_block = out.lm_block; N->proj_order = out.proj_order;
            N->hash = id;
            out.tmpl = N;
            std::shared_ptr<const PlanTemplate> victim;      // (released outside the lock)
            CacheShard &S = shard_of(key);
            std::lock_guard<std::mutex> g(S.mu);
            if (S.map.find(key) == S.map.end()) {      // (another thread may have packed the same structure meanwhile)
                if (S.map.size() >= CACHE_SHARD_ENTRIES) { victim = S.lru.back().tmpl; S.map.erase(S.lru.back().key); S.lru.pop_back(); }
                S.lru.push_front(CacheEntry{key, N});
                S.map.emplace(std::move(key), S.lru.begin());
            }
        }
    }
    lap("template + cache insert");
    if (plan_only) {      // size only: the caller writes the data with pack_problem_data once every window of its batch has an offset
        Sink cnt(nullptr);
        const int rcd = pack_data_to(p, out, imu_sqrt, cnt);
        lap("data size");
        return rcd;
    }
    return pack_data(p, out, imu_sqrt);
}

}  // namespace tcv
