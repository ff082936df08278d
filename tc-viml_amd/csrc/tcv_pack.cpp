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
static int host_core_grant() {
    static int grant = 0;
    if (grant > 0) return grant;
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
    grant = std::max(1, g);
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
};
WorkerPool &worker_pool() { static WorkerPool *P = new WorkerPool(); return *P; }
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
    for (;;) {
        std::shared_ptr<ParCall> c;
        {
            std::unique_lock<std::mutex> g(P.mu);
            for (;;) {
                while (!P.q.empty() && P.q.front()->next.load(std::memory_order_relaxed) >= P.q.front()->n) P.q.pop_front();      // fully claimed
                if (!P.q.empty()) { c = P.q.front(); break; }
                P.cv.wait(g);
            }
        }
        run_call(*c);
    }
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
    {
        std::lock_guard<std::mutex> g(P.mu);
        const int want = std::min(31, std::max(1, host_core_grant() - 1));
        while (P.nworkers < std::min(want, nth - 1)) { std::thread(worker_main).detach(); P.nworkers++; }
        P.q.push_back(c);
    }
    P.cv.notify_all();
    run_call(*c);
    std::unique_lock<std::mutex> g(c->mu);
    c->cv.wait(g, [&] { return c->done.load(std::memory_order_acquire) >= c->n; });
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
    static int v = 0;
    if (v == 0) {
        const char *e = getenv("TCV_CHAIN_LDS_DOUBLES");
        v = e ? atoi(e) : LDS_DOUBLES / 2;
        if (v < 6144 || v > LDS_DOUBLES) v = LDS_DOUBLES / 2;
        v &= ~1;
    }
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

// structural half: plan header, int pool and the host-side maps (everything that does not depend on the VALUES of the window)
// coop_chunks > 0: plan for the cooperative kernel (tcv_packed.h COOP_*): at least that many visual chunks of at most 256 point factors
// each (one helper workgroup per chunk, one lane per factor), and an LDS budget that leaves room for a helper's second tile set
static int pack_plan(const tcv_problem &p, Packed &out, int mode, int chain_lds, int coop_chunks) {
    const int nb = (int)p.blocks.size();
    static const bool dbg_t = getenv("TCV_DEBUG_PACK2") != nullptr;      // developer: where the symbolic packing spends its time
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (dbg_t) { const auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[pack_plan] %-28s %7.1f us\n", what, std::chrono::duration<double, std::micro>(t - t_prev).count()); t_prev = t; } };
    // ---- classify blocks: landmarks = size-1 Euclidean blocks used only as 4th block of projection factors
    std::vector<int> use_lm(nb, 0), use_other(nb, 0);
    for (auto &f : p.proj) { use_lm[f.b[3]]++; for (int k = 0; k < 3; k++) use_other[f.b[k]]++; if (f.btd >= 0) use_other[f.btd]++; }
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

    // ---- ambient / tangent offsets: pose-kind blocks first in tangent space
    std::vector<int> gsize(nblk), goff(nblk), loff(nblk, -1), kind(nblk);
    int nx = 0, nc = 0;
    for (int c = 0; c < nblk; c++) {
        const ParamBlock &pb = p.blocks[out.cam_block[c]];
        gsize[c] = pb.size; kind[c] = pb.kind; goff[c] = nx; nx += pb.size;
        if (pb.kind == KIND_EUCLID && pb.size > 15) { set_error("Euclidean block wider than 15"); return TCV_ERR_UNSUPPORTED; }
    }
    for (int c = 0; c < nblk; c++)
        if (kind[c] == KIND_POSE && !p.blocks[out.cam_block[c]].constant) { loff[c] = nc; nc += 6; }
    // Td comes right behind the poses: its column rides through the 6-wide gather machinery as a pseudo block whose other five
    // columns are structural zeros, so five more tangent rows must follow it.  It counts as part of the "pose part" npp: the
    // leading tangent dims the visual factors touch (landmark Schur corrections of the diagonal and the right-hand side).
    const int td_cam = td_blk >= 0 ? cam_of[td_blk] : -1;
    if (td_cam >= 0 && !p.blocks[td_blk].constant) { loff[td_cam] = nc; nc += 1; }
    const int npp = nc;
    for (int c = 0; c < nblk; c++)
        if (kind[c] != KIND_POSE && c != td_cam && !p.blocks[out.cam_block[c]].constant) { loff[c] = nc; nc += gsize[c]; }
    if (nc < 1) { set_error("no free camera-side parameter block"); return TCV_ERR_INVALID; }
    out.cam_loff = loff;
    const int nt = (nc + 1 + 15) / 16, ntp = (npp + 15) / 16;
    const int ntiles = nt * (nt + 1) / 2, pp_tiles = ntp * (ntp + 1) / 2;
    if (td_cam >= 0 && loff[td_cam] >= 0 && loff[td_cam] + 6 > nt * 16) { set_error("Td block: no room for its gather slot"); return TCV_ERR_UNSUPPORTED; }
    // camera tangent dims: 171 for the 11 frames + extrinsic of OptimizationWithLine (172 with Td), 177 with the relocalisation pose (:1854-1886)
    if (nc > CAM_MAX - 1 || npp > 88 || nc + L > SCR_NL || L > 1024) { set_error("window too large for the fused solver (camera tangent dim > 183)"); return TCV_ERR_TOO_LARGE; }
    const int camw = nc <= CAM_W - 1 ? (int)CAM_W : (int)CAM_MAX;
    const int nxl = (nx + L + 1) & ~1;
    int area_cap = LDS_DOUBLES - ntiles * 256 - 2 * nxl - (3 * camw + 176) - 64;
    int stage_cap = (ntiles - pp_tiles) * 256;
    // (the dense layout holds the whole camera system in LDS tiles: 12 tile rows -- a window with the relocalisation pose -- do not fit; such a
    // window needs the chain layout, decided below)
    const bool dense_fits = area_cap >= 512;
    if (!dense_fits && mode != 0) { set_error("window too large for the fused solver (LDS)"); return TCV_ERR_TOO_LARGE; }

    // ---- chain layout: the free Euclidean camera blocks (speed-biases, 9 wide) only meet their IMU neighbours and the
    // prior, so they are eliminated one after the other BEFORE the dense pose system (block-sparse Cholesky with the poses
    // ordered last, what SPARSE_SCHUR's reduced-camera factorisation exploits too).  Symbolic elimination at block level:
    // eligible iff every Euclidean block has at most one later-eliminated Euclidean neighbour and that one is next in order.
    struct ChainStep { int cam, t0; std::vector<int> prow_t; bool next; int nsrc, f[2], lc[2]; };
    std::vector<ChainStep> chain;
    bool use_chain = (mode == 0);
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
        std::vector<std::vector<char>> adj(nblk, std::vector<char>(nblk, 0));
        auto link = [&](const std::vector<int> &bs) { for (int a2 : bs) for (int b2 : bs) if (a2 != b2 && loff[a2] >= 0 && loff[b2] >= 0) adj[a2][b2] = 1; };
        for (auto &f : p.imu) link({cam_of[f.b[0]], cam_of[f.b[1]], cam_of[f.b[2]], cam_of[f.b[3]]});
        if (!p.prior.empty()) { std::vector<int> bs; for (int b : p.prior[0].b) bs.push_back(cam_of[b]); link(bs); }
        for (int s2 = 0; s2 < ne && use_chain; s2++) {
            const int e = eorder[s2];
            ChainStep st;
            st.cam = e; st.t0 = loff[e]; st.next = false; st.nsrc = 0; st.f[0] = st.f[1] = 0; st.lc[0] = st.lc[1] = 0;
            std::vector<int> later;
            for (int x = 0; x < nblk; x++) if (adj[e][x] && pos[x] > s2) later.push_back(x);
            if (later.size() > 1 || (later.size() == 1 && pos[later[0]] != s2 + 1)) { use_chain = false; break; }
            st.next = !later.empty();
            std::vector<int> prow;
            for (int x = 0; x < nblk; x++) if (adj[e][x] && (kind[x] == KIND_POSE || x == td_cam)) prow.push_back(x);
            std::sort(prow.begin(), prow.end(), [&](int a2, int b2) { return loff[a2] < loff[b2]; });
            for (int x : prow) for (int j = 0; j < (x == td_cam ? 1 : 6); j++) st.prow_t.push_back(loff[x] + j);
            // fill: the eliminated block's neighbours become a clique (pose-pose is dense anyway)
            for (int x : later) for (int y : prow) { adj[x][y] = 1; adj[y][x] = 1; }
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
    // (Td's gather slot is six columns wide, tcv_packed.h: the pose tiles have to cover the five structural zeros behind its column)
    const int nt_c = (std::max(npp + 1, (td_cam >= 0 && loff[td_cam] >= 0) ? loff[td_cam] + 6 : 0) + 15) / 16, ctiles = nt_c * (nt_c + 1) / 2;
    const int c_vec = 2 * nxl + (3 * camw + 176) + 64 + 112;
    const int c_lds = coop_chunks > 0 ? (LDS_DOUBLES - ctiles * 256 - 8) : ((chain_lds >= 6144 && chain_lds <= LDS_DOUBLES) ? (chain_lds & ~1) : chain_lds_doubles());
    const int c_pool = c_lds - ctiles * 256 - c_vec;
    if (use_chain && (c_pool < chain_pool_doubles((int)chain.size(), nt_c) || c_pool < IMU_REC)) use_chain = false;

    lap("classification + chain steps");
    PlanHdr &H = out.hdr;
    std::memset(&H, 0, sizeof(H));
    H.nblk = nblk; H.nland = L; H.nc = nc; H.nx = nx; H.npp = npp; H.nt = nt; H.ntp = ntp;
    H.n_imu = (int)p.imu.size(); H.n_proj = (int)p.proj.size(); H.n_line = (int)p.line.size();
    H.lds_area = area_cap;
    H.flags = td_blk >= 0 ? 1 : 0; H.td_cam = td_cam; H.camw = camw;
    const int prec = td_blk >= 0 ? (int)PROJ_TD_REC : (int)PROJ_REC;      // doubles per staged point record
    const int td_t = td_cam >= 0 ? loff[td_cam] : -1;
    std::vector<int> &I = out.ints;
    I.clear();
    auto mark = [&]() { return (int)I.size(); };

    H.o_blk = mark();
    for (int c = 0; c < nblk; c++) { I.push_back(gsize[c]); I.push_back(goff[c]); I.push_back(loff[c]); I.push_back(kind[c]); }
    H.o_imu = mark();
    for (auto &f : p.imu) for (int k = 0; k < 4; k++) I.push_back(cam_of[f.b[k]]);

    // ---- projection factors sorted by landmark (stable), landmark slots
    std::vector<int> &order = out.proj_order;
    order.resize(p.proj.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lm_of[p.proj[a].b[3]] < lm_of[p.proj[b].b[3]]; });
    std::vector<int> lmptr(L + 1, 0);
    for (auto &f : p.proj) lmptr[lm_of[f.b[3]] + 1]++;
    for (int l = 0; l < L; l++) lmptr[l + 1] += lmptr[l];
    std::vector<std::vector<int>> lm_slots(L);   // tangent offsets of the distinct free pose blocks
    std::vector<int> e_off(L + 1, 0);
    for (int l = 0; l < L; l++) {
        for (int k = lmptr[l]; k < lmptr[l + 1]; k++) {
            const ProjFac &f = p.proj[order[k]];
            for (int s = 0; s < 3; s++) {
                const int t = loff[cam_of[f.b[s]]];
                if (t < 0) continue;
                if (std::find(lm_slots[l].begin(), lm_slots[l].end(), t) == lm_slots[l].end()) lm_slots[l].push_back(t);
            }
            if (td_t >= 0 && std::find(lm_slots[l].begin(), lm_slots[l].end(), td_t) == lm_slots[l].end()) lm_slots[l].push_back(td_t);
        }
        if (lm_slots[l].size() > 40) { set_error("landmark observed from more than 40 blocks"); return TCV_ERR_TOO_LARGE; }
        if (e_off[l] >= (1 << 16) - 256) { set_error("landmark coupling store too large"); return TCV_ERR_TOO_LARGE; }
        e_off[l + 1] = e_off[l] + 6 * (int)lm_slots[l].size() + 2;   // + 1/kappa and gl/kappa behind the slice
    }
    H.hcl_total = e_off[L];
    if (H.hcl_total > (1 << 16) - 256) { set_error("landmark coupling store too large"); return TCV_ERR_TOO_LARGE; }
    H.o_proj = mark();
    for (size_t k = 0; k < order.size(); k++) {
        const ProjFac &f = p.proj[order[k]];
        for (int s = 0; s < 3; s++) I.push_back(cam_of[f.b[s]]);
        I.push_back(lm_of[f.b[3]]);
    }
    H.o_line = mark();
    for (auto &f : p.line) I.push_back(cam_of[f.b]);

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

    H.o_lm = mark();
    for (int l = 0; l < L; l++) { I.push_back(e_off[l]); I.push_back((int)lm_slots[l].size()); }
    H.o_lmslotptr = mark();
    { int acc = 0; for (int l = 0; l < L; l++) { I.push_back(acc); acc += (int)lm_slots[l].size(); } I.push_back(acc); }
    H.o_lmslot = mark();
    for (int l = 0; l < L; l++) for (int t : lm_slots[l]) I.push_back(t);

    // ---- visual chunks (whole landmarks; lines ride in chunk 0).  Staging holds the factor records AND the chunk's
    // gather program (units + items), so both count against the capacity.
    struct VChunk { int pb, pn, lb, ln, lmb, lmn; };
    std::vector<VChunk> vch;
    auto build_chunks = [&](int stage_cap, int area_cap, std::vector<VChunk> &vch) -> int {
        vch.clear();
        const int nline = (int)p.line.size();
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
            const int nf = lmptr[l + 1] - lmptr[l], ns = (int)lm_slots[l].size(), nh = 6 * ns + 2;
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
    // gather programs of a chunk list; measures the staging / area doubles the largest chunk needs
    auto emit_all = [&](const std::vector<VChunk> &vch, int stage_chk, int area_chk, std::vector<int> &vprog, std::vector<int> &sprog,
                        std::vector<int> &vchunk_tab, int &max_stage, int &max_area) -> int {
        vprog.clear(); sprog.clear(); vchunk_tab.clear(); max_stage = 0; max_area = 0;
        for (auto &c : vch) {
        DestList dl, sl;
        std::vector<Col> cols;      // (one allocation per chunk instead of one per factor)
        cols.reserve(4);
        for (int k = 0; k < c.pn; k++) {
            const ProjFac &f = p.proj[order[c.pb + k]];
            const int l = lm_of[f.b[3]];
            const int base = k * prec;
            if (base >= (1 << 21)) { set_error("staging offset overflow"); return TCV_ERR_TOO_LARGE; }
            auto mk = [&](int ca, int cb) { return (int)(((unsigned)base << 11) | ((unsigned)ca << 6) | ((unsigned)cb << 1)); };
            cols.clear();
            for (int s2 = 0; s2 < 3; s2++) cols.push_back(Col{loff[cam_of[f.b[s2]]], 6 * s2, 6});
            if (td_t >= 0) cols.push_back(Col{td_t, 20, 6});      // [td | 5 zero columns]
            for (size_t a2 = 0; a2 < cols.size(); a2++)
                for (size_t b2 = 0; b2 < a2; b2++)
                    if (cols[a2].t >= 0 && cols[a2].t == cols[b2].t) { set_error("projection factor uses one block twice"); return TCV_ERR_UNSUPPORTED; }
            add_pairs(dl, cols, mk, 19);
            for (auto &cc : cols) {
                if (cc.t < 0) continue;
                const int slot = (int)(std::find(lm_slots[l].begin(), lm_slots[l].end(), cc.t) - lm_slots[l].begin());
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
            cols.push_back(Col{loff[cam_of[f.b]], 0, 6});
            add_pairs(dl, cols, mk, 6);
        }
        for (int l = c.lmb; l < c.lmb + c.lmn; l++) {
            const auto &sl_t = lm_slots[l];
            for (size_t a2 = 0; a2 < sl_t.size(); a2++) {
                const unsigned hoff = (unsigned)(e_off[l] - e_off[c.lmb]), ns = (unsigned)sl_t.size();
                auto mk = [&](int sa, int sb) { return (int)((hoff << 18) | (ns << 12) | ((unsigned)sa << 6) | (unsigned)sb); };
                sl.add(DK_TILE, sl_t[a2], sl_t[a2], 6, 0, mk((int)a2, (int)a2));
                sl.add(DK_RC, sl_t[a2], 0, 6, 1, mk(63, (int)a2));                // rc[t + e] += gl/kappa * Hcl[slot][e]
                for (size_t b2 = 0; b2 < a2; b2++) {
                    if (sl_t[a2] > sl_t[b2]) sl.add(DK_TILE, sl_t[a2], sl_t[b2], 6, 6, mk((int)a2, (int)b2));
                    else sl.add(DK_TILE, sl_t[b2], sl_t[a2], 6, 6, mk((int)b2, (int)a2));
                }
            }
        }
        dl.finish(); sl.finish();
        RowProg vp, sp;
        static const int wu_v = getenv("TCV_WU_V") ? atoi(getenv("TCV_WU_V")) : (int)WAVE_UNIT_ITEMS, wu_s = getenv("TCV_WU_S") ? atoi(getenv("TCV_WU_S")) : (int)WAVE_UNIT_ITEMS,
                         wu_max = getenv("TCV_WU_MAX") ? atoi(getenv("TCV_WU_MAX")) : (int)WAVE_UNIT_MAX;      // tuning experiments
        if (!emit_rows(dl, vp, false, wu_v, wu_max) || !emit_rows(sl, sp, false, wu_s, wu_max)) { set_error("gather program field overflow"); return TCV_ERR_TOO_LARGE; }
        if (getenv("TCV_DEBUG_UNITS")) {
            auto dump = [](const char *nm, const RowProg &rp) {
                fprintf(stderr, "[pack] %s: %d units (%d wave units), items per unit (descending):", nm, rp.n_units, rp.n_wave_units);
                for (int u = 0; u < rp.n_units; u += (u < 16 ? 1 : 16)) fprintf(stderr, " %d", rp.units[3 * u] & 0xfffff);
                fprintf(stderr, "\n");
            };
            dump("visual", vp); dump("schur", sp);
        }
        const int recs = (c.pn * prec + c.ln * LINE_REC + 1) & ~1;
        const int st_need = std::max(recs + ((int)(vp.units.size() + vp.items.size()) + 1) / 2 + 2, ((int)(sp.units.size() + sp.items.size()) + 1) / 2 + 2);
        const int ar_need = (e_off[c.lmb + c.lmn] - e_off[c.lmb]) + 3 * c.lmn + 8;
        max_stage = std::max(max_stage, st_need); max_area = std::max(max_area, ar_need);
        if (stage_chk >= 0 && (st_need > stage_chk || ar_need > area_chk)) {
            set_error("gather program does not fit the LDS staging area"); return TCV_ERR_TOO_LARGE;
        }
        while (vprog.size() & 3) vprog.push_back(0);        // every program starts 16-byte aligned (vector loads on the device)
        while (sprog.size() & 3) sprog.push_back(0);
        const int voff = (int)vprog.size(), soff = (int)sprog.size();
        vprog.insert(vprog.end(), vp.units.begin(), vp.units.end()); vprog.insert(vprog.end(), vp.items.begin(), vp.items.end());
        sprog.insert(sprog.end(), sp.units.begin(), sp.units.end()); sprog.insert(sprog.end(), sp.items.begin(), sp.items.end());
        const int tab[16] = {c.pb, c.pn, c.lb, c.ln, voff, vp.n_units, vp.n_wave_units, (int)vp.items.size(),
                             c.lmb, c.lmn, e_off[c.lmb], e_off[c.lmb + c.lmn] - e_off[c.lmb],
                             soff, sp.n_units, sp.n_wave_units, (int)sp.items.size()};
        vchunk_tab.insert(vchunk_tab.end(), tab, tab + 16);
            }
        return TCV_OK;
    };
    lap("tables, landmark slots");
    std::vector<int> vprog, sprog, vchunk_tab;
    if (use_chain) {
        // chain layout: the LDS pool (staging | landmark coupling area) is small, so the landmarks are dealt evenly to the
        // smallest number k of chunks whose EXACT programs fit; the line factors are spread over the chunks
        bool found = false;
        const int nline = (int)p.line.size(), nproj = (int)p.proj.size();
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
        if (!found) { use_chain = false; chain.clear(); }
    }
    if (!use_chain && !dense_fits) { set_error("window too large for the fused solver (LDS; its speed-bias blocks do not form a chain either)"); return TCV_ERR_TOO_LARGE; }
    if (!use_chain) {
        chain.clear();
        { const int rc = build_chunks(stage_cap, area_cap, vch); if (rc != TCV_OK) return rc; }
        int ms = 0, ma = 0;
        const int rc = emit_all(vch, stage_cap, area_cap, vprog, sprog, vchunk_tab, ms, ma);
        if (rc != TCV_OK) return rc;
    }
    lap("chunks + gather programs");
    set_error("");
    H.chain = use_chain ? 1 : 0; H.n_e = (int)chain.size(); H.nt_c = nt_c; H.c_stage_cap = stage_cap; H.c_area_cap = area_cap; H.c_pool = c_pool;
    if (use_chain) H.lds_area = area_cap;
    H.n_vis_chunk = (int)vch.size();
    H.o_vchunk = mark(); I.insert(I.end(), vchunk_tab.begin(), vchunk_tab.end());
    while ((I.size() & 3) != 0) I.push_back(0);
    H.o_vdest = mark(); I.insert(I.end(), vprog.begin(), vprog.end()); H.n_vdest = 0;
    H.o_vunit = H.o_vdest; H.n_vunit = (int)vprog.size(); H.o_vitem = H.o_vdest; H.n_vitem = 0;
    while ((I.size() & 3) != 0) I.push_back(0);
    H.o_sdest = mark(); I.insert(I.end(), sprog.begin(), sprog.end()); H.n_sdest = 0;
    H.o_sunit = H.o_sdest; H.n_sunit = (int)sprog.size(); H.o_sitem = H.o_sdest; H.n_sitem = 0;

    std::vector<int> imap_all;
    // ---- IMU chunks.  Per factor: the tangent index of each of its 30 local Jacobian columns (-1 for a constant
    // block) and a colour; factors of one colour share no parameter block, so their J'J tiles can be scattered into the
    // reduced camera system concurrently (the frame chain needs two colours).
    {
        const int imu_cap = use_chain ? c_pool : area_cap;       // chain mode: the whole pool holds IMU records
        const int per = std::max(1, std::min(H.n_imu, imu_cap / IMU_REC));
        if (H.n_imu > 0 && imu_cap < IMU_REC) { set_error("no LDS room for IMU staging"); return TCV_ERR_TOO_LARGE; }
        if (H.n_imu > 16) { set_error("more than 16 IMU factors"); return TCV_ERR_TOO_LARGE; }
        std::vector<int> &imap = imap_all;
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
        std::vector<int> sct((size_t)H.n_imu * 1024, 0);
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
        while ((I.size() & 3) != 0) I.push_back(0);      // read with 16-byte loads
        H.o_iitem = mark(); I.insert(I.end(), sct.begin(), sct.end()); H.n_iitem = (int)sct.size();
        H.o_ichunk = mark(); I.insert(I.end(), ichunk.begin(), ichunk.end());
    }
    {
        // destinations of the constant part of the prior (Hp, packed lower triangle): same derivation as the kernel's former inline one
        H.o_pdest = mark();
        const int pn = (int)pcol.size();
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
    lap("program copy, IMU tables");
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
                        for (int l = 0; l < 30; l++) if (imap_all[(size_t)st.f[src] * 32 + l] == rt[r]) l01[src] = l;
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
    while ((I.size() & 3) != 0) I.push_back(0);          // plans are concatenated: keep every plan 16-byte aligned
    H.plan_ints = (int)I.size();
    lap("chain tables, frames");

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
    void seal() {
        unsigned long long x = 1469598103934665603ull;
        for (int v : k) { x ^= (unsigned)v; x *= 1099511628211ull; }
        h = (size_t)x;
    }
    bool operator==(const PlanKey &o) const { return h == o.h && k == o.k; }
};
struct KeyHash { size_t operator()(const PlanKey &k) const { return k.h; } };
// LRU: a live estimator / replay produces a new structure almost every frame (tracks start and end), so most entries never hit again;
// 256 entries hold the structures a bench / streaming loop cycles through (~27 MB of plans at most) and the cold ones fall off the end
// one by one instead of the whole table being cleared.
struct CacheEntry { PlanKey key; std::shared_ptr<const PlanTemplate> tmpl; };
std::mutex g_cache_mu;
std::list<CacheEntry> g_lru;
std::unordered_map<PlanKey, std::list<CacheEntry>::iterator, KeyHash> g_cache;
long long g_hits = 0, g_misses = 0;
enum { CACHE_MAX_ENTRIES = 256 };

void structure_key(const tcv_problem &p, int mode, int chain_lds, int coop_chunks, PlanKey &key) {
    std::vector<int> &k = key.k;
    k.clear();
    k.reserve(16 + 3 * p.blocks.size() + 4 * p.imu.size() + 5 * p.proj.size() + p.line.size() + 64);
    k.push_back(mode); k.push_back(chain_lds); k.push_back(coop_chunks); k.push_back((int)p.blocks.size());
    for (auto &b : p.blocks) k.push_back(b.size | (b.kind << 8) | ((int)b.constant << 16));
    k.push_back((int)p.imu.size());
    for (auto &f : p.imu) for (int j = 0; j < 4; j++) k.push_back(f.b[j]);
    k.push_back((int)p.proj.size());
    for (auto &f : p.proj) { for (int j = 0; j < 4; j++) k.push_back(f.b[j]); k.push_back(f.btd); }
    k.push_back((int)p.line.size());
    for (auto &f : p.line) k.push_back(f.b);
    k.push_back((int)p.prior.size());
    for (auto &f : p.prior) {
        const tcv_prior *pr = f.prior;
        k.push_back(pr->n); k.push_back((int)pr->size.size()); k.push_back(pr->xsize);
        for (size_t j = 0; j < f.b.size(); j++) { k.push_back(f.b[j]); k.push_back(pr->size[j]); k.push_back(pr->idx[j]); k.push_back(pr->xoff[j]); }
    }
    k.push_back((int)p.frame_pose.size());
    for (int v : p.frame_pose) k.push_back(v);
    for (int v : p.frame_sb) k.push_back(v);
    key.seal();
}
}  // namespace

void plan_cache_stats(long long *hits, long long *misses, long long *entries) {
    std::lock_guard<std::mutex> g(g_cache_mu);
    if (hits) *hits = g_hits;
    if (misses) *misses = g_misses;
    if (entries) *entries = (long long)g_cache.size();
}

int pack_problem(const tcv_problem &p, Packed &out, const double *imu_sqrt, int mode, int chain_lds, bool plan_only, int coop_chunks) {
    static const bool no_cache = getenv("TCV_NO_PLAN_CACHE") != nullptr;
    const int c_lds = (chain_lds >= 6144 && chain_lds <= LDS_DOUBLES) ? (chain_lds & ~1) : chain_lds_doubles();
    PlanKey key;
    std::shared_ptr<const PlanTemplate> T;
    std::memset(&out.win, 0, sizeof out.win);
    if (!no_cache) {
        structure_key(p, mode, c_lds, coop_chunks, key);
        std::lock_guard<std::mutex> g(g_cache_mu);
        auto it = g_cache.find(key);
        if (it != g_cache.end()) { g_lru.splice(g_lru.begin(), g_lru, it->second); T = it->second->tmpl; g_hits++; } else g_misses++;
    }
    if (T) {
        out.hdr = T->hdr; out.tmpl = T; out.ints.clear();
        out.cam_block = T->cam_block; out.cam_loff = T->cam_loff; out.lm_block = T->lm_block; out.proj_order = T->proj_order;
    } else {
        out.tmpl.reset();
        const int rc = pack_plan(p, out, mode, c_lds, coop_chunks);
        if (rc != TCV_OK) return rc;
        if (!no_cache) {
            auto N = std::make_shared<PlanTemplate>();
            N->hdr = out.hdr; N->ints.swap(out.ints);
            N->cam_block = out.cam_block; N->cam_loff = out.cam_loff; N->lm_block = out.lm_block; N->proj_order = out.proj_order;
            out.tmpl = N;
            std::lock_guard<std::mutex> g(g_cache_mu);
            if (g_cache.find(key) == g_cache.end()) {      // (another thread may have packed the same structure meanwhile)
                while (g_cache.size() >= CACHE_MAX_ENTRIES) { g_cache.erase(g_lru.back().key); g_lru.pop_back(); }
                g_lru.push_front(CacheEntry{key, N});
                g_cache.emplace(std::move(key), g_lru.begin());
            }
        }
    }
    if (plan_only) {      // size only: the caller writes the data with pack_problem_data once every window of its batch has an offset
        Sink cnt(nullptr);
        return pack_data_to(p, out, imu_sqrt, cnt);
    }
    return pack_data(p, out, imu_sqrt);
}

}  // namespace tcv
