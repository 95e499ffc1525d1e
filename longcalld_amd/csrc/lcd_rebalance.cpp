// lcd_rebalance.cpp -- cross-rank rebalancing of region queues (north_star: "RCCL over xGMI used only to rebalance region queues").
//
// The reference balances its chunk workers inside ONE process by work stealing (kt_for, src/kthread.c:24-64; called at src/call_var_main.c:773).  With one
// process per GPU there is nothing to steal from, so the ranks run one EPOCH before the hot path starts:
//   1. every rank packs its region jobs per chunk into one buffer per chunk (lcd_region_jobs_pack: the inputs of lcd_batch_add_region, byte for byte) and
//      prices it (lcd_region_job_cost: the same DP-cell estimate the in-process dispatcher orders by);
//   2. ncclAllGather of the queue depths, then of the (cost, bytes) tables;
//   3. the same deterministic plan on every rank (lcd_rebalance_plan: from the most loaded rank to the least loaded one, the job that brings the pair closest to
//      equal; a job moves at most once);
//   4. ncclSend / ncclRecv of WHOLE packed buffers inside one ncclGroupStart / End -- 15 KB to a few MB each, so neither ring bandwidth nor the per-link
//      153 GB/s of xGMI matter; what matters is that every rank agrees on the plan and that a job lands exactly once.
// No data-path collective exists: chunks are independent until stitch_var_main (src/collect_var.c:2983).
// librccl is opened lazily (dlopen) by lcd_comm_create: the library has no load-time dependency on it, and a single-GPU caller never touches it.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/lcd_hotpath.h"

namespace {
thread_local std::string g_rb_err;
int rb_err(int code, const std::string &m) { g_rb_err = m; return code; }
constexpr int64_t LCD_PACK_MAGIC = 0x4C434452; // 'LCDR' (longcalld_amd/rebalance.py MAGIC)

// ---- the slice of the NCCL API the epoch needs (rccl.h: same names and types) ----
typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_INT32 = 2, NCCL_INT64 = 4, NCCL_FLOAT64 = 8 };
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(nccl_uid_t *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    // optional (absent from a very old librccl: the epoch works without them)
    int (*CommAbort)(nccl_comm_t) = nullptr;
    int (*CommCount)(nccl_comm_t, int *) = nullptr;
    int (*CommUserRank)(nccl_comm_t, int *) = nullptr;
    int (*CommCuDevice)(nccl_comm_t, int *) = nullptr;
};
Rccl g_rccl; std::once_flag g_rccl_once; std::string g_rccl_why;
bool rccl_open() {
    std::call_once(g_rccl_once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { g_rccl.h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (g_rccl.h) break; }
        if (!g_rccl.h) { g_rccl_why = std::string("librccl not found: ") + (dlerror() ? dlerror() : "?"); return; }
#define LCD_SYM(field, sym) do { *(void **)(&g_rccl.field) = dlsym(g_rccl.h, sym); if (!g_rccl.field) { g_rccl_why = std::string("librccl lacks ") + sym; dlclose(g_rccl.h); g_rccl.h = nullptr; return; } } while (0)
        LCD_SYM(GetUniqueId, "ncclGetUniqueId"); LCD_SYM(CommInitRank, "ncclCommInitRank"); LCD_SYM(CommDestroy, "ncclCommDestroy"); LCD_SYM(AllGather, "ncclAllGather");
        LCD_SYM(Send, "ncclSend"); LCD_SYM(Recv, "ncclRecv"); LCD_SYM(GroupStart, "ncclGroupStart"); LCD_SYM(GroupEnd, "ncclGroupEnd"); LCD_SYM(GetErrorString, "ncclGetErrorString");
#undef LCD_SYM
        *(void **)(&g_rccl.CommAbort) = dlsym(g_rccl.h, "ncclCommAbort"); *(void **)(&g_rccl.CommCount) = dlsym(g_rccl.h, "ncclCommCount");
        *(void **)(&g_rccl.CommUserRank) = dlsym(g_rccl.h, "ncclCommUserRank"); *(void **)(&g_rccl.CommCuDevice) = dlsym(g_rccl.h, "ncclCommCuDevice");
    });
    return g_rccl.h != nullptr;
}
} // namespace

struct lcd_comm_s { nccl_comm_t comm = nullptr; int world = 1, rank = 0, device = 0; hipStream_t st = nullptr; bool dead = false; /* aborted by a failed epoch: only lcd_comm_destroy is left */ };

extern "C" {
const char *lcd_rebalance_last_error(void) { return g_rb_err.c_str(); }

// DP-cell estimate of one region job (longcalld_amd/rebalance.py region_cost; the shape of lcd_batch_cost without running the host planner): a region with phased
// reads runs banded K1 chains (bases x band), an unphased one the unbanded K2 chain over its full-cover reads (bases x length)
double lcd_region_job_cost(const lcd_region_job_t *j) {
    if (j->n_reads <= 0) return 0.0;
    double sum = 0, full = 0, maxl = 0; bool phased = false;
    for (int i = 0; i < j->n_reads; ++i) {
        const double l = j->lens[i];
        sum += l; maxl = std::max(maxl, l);
        if (j->haps[i] > 0) phased = true;
        if (j->fully_covers[i] == 12) full += l;
    }
    if (phased) return sum * std::min(maxl + 1, 2 * (10 + maxl / 100) + 1 + 64);
    return full * (maxl + 1);
}

// The wire format of a chunk's region jobs (== longcalld_amd/rebalance.py pack_regions): int64 {magic, n}, n x int64 {reg_len, n_reads, ref_len}, then per region
// read_ids | covers | haps (int32 each) | phase_sets (int64) | lens (int32) | ref bases | the reads' bases | the reads' qualities (zeros where the job has none).
// buf == NULL: the size only.
uint64_t lcd_region_jobs_pack(int n, const lcd_region_job_t *jobs, uint8_t *buf) {
    uint64_t o = 16 + 24ull * (uint64_t)std::max(n, 0);
    if (buf) { const int64_t h[2] = {LCD_PACK_MAGIC, n}; memcpy(buf, h, 16); }
    for (int r = 0; r < n; ++r) {
        const lcd_region_job_t &j = jobs[r];
        if (buf) { const int64_t s[3] = {j.reg_len, j.n_reads, j.ref_seq_len}; memcpy(buf + 16 + 24ull * r, s, 24); }
        const size_t m = (size_t)std::max(j.n_reads, 0);
        auto put = [&](const void *p, size_t nb) { if (buf && nb) memcpy(buf + o, p, nb); o += nb; };
        put(j.read_ids, 4 * m); put(j.fully_covers, 4 * m); put(j.haps, 4 * m); put(j.phase_sets, 8 * m); put(j.lens, 4 * m); put(j.ref_seq, (size_t)std::max(j.ref_seq_len, 0));
        for (size_t i = 0; i < m; ++i) put(j.seqs[i], (size_t)std::max(j.lens[i], 0));
        for (size_t i = 0; i < m; ++i) {
            const size_t l = (size_t)std::max(j.lens[i], 0);
            if (j.quals && j.quals[i]) put(j.quals[i], l); else { if (buf && l) memset(buf + o, 0, l); o += l; }
        }
    }
    return o;
}

// every region of a packed buffer into a batch (lcd_batch_add_region with pointers into the buffer); returns the number of regions added, < 0 on a malformed buffer
int lcd_batch_add_packed(lcd_batch_t *b, const uint8_t *buf, uint64_t nbytes) {
    if (nbytes < 16) return rb_err(-50, "packed region buffer: shorter than its header");
    int64_t h[2]; memcpy(h, buf, 16);
    if (h[0] != LCD_PACK_MAGIC || h[1] < 0 || h[1] > INT32_MAX || 16 + 24ull * (uint64_t)h[1] > nbytes) return rb_err(-50, "packed region buffer: bad magic / region count");
    const int n = (int)h[1];
    uint64_t o = 16 + 24ull * (uint64_t)n;
    std::vector<int> ids, cov, haps, lens; std::vector<int64_t> ps; std::vector<const uint8_t *> sp, qp;
    for (int r = 0; r < n; ++r) {
        int64_t s[3]; memcpy(s, buf + 16 + 24ull * r, 24);
        // (bytes from another rank: nothing is cast before it is bounded -- a region is at most LONGCALLD max_noisy_reg_len = 50 kb + flanks, a read slice far below 2^30)
        if (s[0] < 0 || s[0] > INT32_MAX || s[1] < 0 || s[2] < 0 || s[1] > (1 << 24) || s[2] > INT32_MAX) return rb_err(-50, "packed region buffer: bad region header");
        const size_t m = (size_t)s[1];
        if (o + 24 * m + (uint64_t)s[2] > nbytes) return rb_err(-50, "packed region buffer: truncated");
        ids.resize(m); cov.resize(m); haps.resize(m); lens.resize(m); ps.resize(m); sp.resize(m); qp.resize(m);
        memcpy(ids.data(), buf + o, 4 * m); o += 4 * m; memcpy(cov.data(), buf + o, 4 * m); o += 4 * m; memcpy(haps.data(), buf + o, 4 * m); o += 4 * m;
        memcpy(ps.data(), buf + o, 8 * m); o += 8 * m; memcpy(lens.data(), buf + o, 4 * m); o += 4 * m;
        const uint8_t *ref = buf + o; o += (uint64_t)s[2];
        uint64_t tot = 0; for (size_t i = 0; i < m; ++i) { if (lens[i] < 0 || lens[i] > (1 << 30)) return rb_err(-50, "packed region buffer: bad read length"); tot += (uint64_t)lens[i]; }
        if (o + 2 * tot > nbytes) return rb_err(-50, "packed region buffer: truncated");
        for (size_t i = 0; i < m; ++i) { sp[i] = buf + o; o += (uint64_t)lens[i]; }
        for (size_t i = 0; i < m; ++i) { qp[i] = buf + o; o += (uint64_t)lens[i]; }
        if (lcd_batch_add_region(b, s[0], (int)m, ids.data(), lens.data(), sp.data(), qp.data(), cov.data(), haps.data(), ps.data(), ref, (int)s[2]) < 0) return rb_err(-51, lcd_last_error());
    }
    if (o != nbytes) return rb_err(-50, "packed region buffer: trailing bytes");
    return n;
}

// The common plan: repeatedly move, from the most loaded rank to the least loaded one, the job that brings the pair closest to equal (a job moves at most once), until
// the most loaded rank is within `tol` of the mean.  costs: the ranks' job costs one rank after the other (n_jobs[r] each).  moves: room for sum(n_jobs) entries;
// index = position in the source rank's queue.  Deterministic: identical inputs give identical plans on every rank.  Returns the number of moves.
int lcd_rebalance_plan(int world, const int *n_jobs, const double *costs, double tol, int max_moves, lcd_move_t *moves, double *load_before, double *load_after) {
    std::vector<size_t> first((size_t)world + 1, 0);
    for (int r = 0; r < world; ++r) first[r + 1] = first[r] + (size_t)std::max(n_jobs[r], 0);
    std::vector<double> load((size_t)world, 0.0);
    for (int r = 0; r < world; ++r) { double s = 0; for (size_t i = first[r]; i < first[r + 1]; ++i) s += costs[i]; load[r] = s; } // (left to right, as Python's sum())
    if (load_before) for (int r = 0; r < world; ++r) load_before[r] = load[r];
    std::vector<char> moved(first[world], 0);
    double tot = 0; for (double l : load) tot += l;
    const double mean = world > 0 ? tot / world : 0.0;
    int nm = 0;
    while (mean > 0 && (max_moves < 0 || nm < max_moves)) {
        int src = 0, dst = 0;
        for (int r = 1; r < world; ++r) { if (load[r] > load[src]) src = r; if (load[r] < load[dst]) dst = r; } // (ties: the lowest rank, both ways)
        if (!((load[src] - mean) / mean > tol)) break;
        const double gap = load[src] - load[dst];
        long best = -1; double best_c = 0.0;
        for (size_t i = first[src]; i < first[src + 1]; ++i) { // the job closest to half the gap; anything >= the gap would only swap the roles
            const double c = costs[i];
            if (moved[i] || c <= 0 || c >= gap) continue;
            if (best < 0 || std::fabs(c - gap / 2) < std::fabs(best_c - gap / 2)) { best = (long)i; best_c = c; }
        }
        if (best < 0) {
            // nothing of src's is smaller than the gap (SV-heavy chunks: a few jobs as large as the whole imbalance).  A SWAP still levels the pair: src's job a for
            // dst's job b with 0 < a - b < gap, the difference closest to half the gap -- two moves, each job still moves once.  (the first pair in (a, b) order wins ties)
            long ba = -1, bb = -1; double bd = 0.0;
            if (max_moves < 0 || nm + 2 <= max_moves)
                for (size_t i = first[src]; i < first[src + 1]; ++i) {
                    if (moved[i] || costs[i] <= 0) continue;
                    for (size_t j = first[dst]; j < first[dst + 1]; ++j) {
                        if (moved[j] || costs[j] <= 0) continue;
                        const double d = costs[i] - costs[j];
                        if (d <= 0 || d >= gap) continue;
                        if (ba < 0 || std::fabs(d - gap / 2) < std::fabs(bd - gap / 2)) { ba = (long)i; bb = (long)j; bd = d; }
                    }
                }
            if (ba < 0) break;
            moved[(size_t)ba] = 1; moved[(size_t)bb] = 1;
            moves[nm].src = src; moves[nm].index = (int)((size_t)ba - first[src]); moves[nm].dst = dst; ++nm;
            moves[nm].src = dst; moves[nm].index = (int)((size_t)bb - first[dst]); moves[nm].dst = src; ++nm;
            load[src] -= bd; load[dst] += bd;
            continue;
        }
        moved[(size_t)best] = 1;
        moves[nm].src = src; moves[nm].index = (int)((size_t)best - first[src]); moves[nm].dst = dst; ++nm;
        load[src] -= best_c; load[dst] += best_c;
    }
    if (load_after) for (int r = 0; r < world; ++r) load_after[r] = load[r];
    return nm;
}

// ---- the RCCL side ----
int lcd_rccl_unique_id(uint8_t id[128]) { // rank 0 makes it; the caller hands the 128 bytes to the other ranks (a file, MPI, torch.distributed's store ...)
    if (!rccl_open()) return rb_err(-52, g_rccl_why);
    nccl_uid_t u; const int rc = g_rccl.GetUniqueId(&u);
    if (rc != 0) return rb_err(-53, std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc));
    memcpy(id, u.internal, 128);
    return 0;
}
lcd_comm_t *lcd_comm_create(int world, int rank, const uint8_t id[128], int device) {
    if (!rccl_open()) { rb_err(-52, g_rccl_why); return nullptr; }
    if (hipSetDevice(device) != hipSuccess) { rb_err(-10, "lcd_comm_create: hipSetDevice failed"); return nullptr; }
    lcd_comm_t *c = new lcd_comm_s(); c->world = world; c->rank = rank; c->device = device;
    nccl_uid_t u; memcpy(u.internal, id, 128);
    const int rc = g_rccl.CommInitRank(&c->comm, world, u, rank);
    if (rc != 0) { rb_err(-53, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc)); delete c; return nullptr; }
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess) { rb_err(-10, "lcd_comm_create: no stream"); g_rccl.CommDestroy(c->comm); delete c; return nullptr; }
    return c;
}
void lcd_comm_destroy(lcd_comm_t *c) {
    if (!c) return;
    if (c->st) (void)hipStreamDestroy(c->st);
    if (c->comm) (void)g_rccl.CommDestroy(c->comm); // (NULL after an abort)
    delete c;
}

// One epoch.  In: this rank's queue (cost, bytes, buffer per job).  Out: its new queue -- the jobs it keeps (pointers into the caller's buffers, owned = 0) and the
// jobs it received (malloc()'d, owned = 1: the caller frees those buffers), out arrays malloc()'d -- and the epoch's statistics.
// Failure: every device buffer and every host block this call made is released on every path (Scratch below); a group that was started is ended; and because the
// peers of a rank that stops in the middle of the epoch would wait in their collectives for ever, a failure after the first collective ABORTS the communicator
// (ncclCommAbort: the peers' calls return with an error) and marks it dead -- the only thing left to do with `c` is lcd_comm_destroy.
namespace {
struct Scratch { // what one epoch allocates; anything not handed to the caller dies with it
    std::vector<void *> dev; std::vector<void *> host; bool group_open = false;
    void *dmalloc(size_t n) { void *p = nullptr; if (hipMalloc(&p, n ? n : 1) != hipSuccess) { (void)hipGetLastError(); return nullptr; } dev.push_back(p); return p; }
    void *hmalloc(size_t n) { void *p = malloc(n ? n : 1); if (p) host.push_back(p); return p; }
    void keep_host() { host.clear(); } // (the out arrays and received buffers now belong to the caller)
    ~Scratch() {
        if (group_open) (void)g_rccl.GroupEnd();
        for (void *p : dev) (void)hipFree(p);
        for (void *p : host) free(p);
    }
};
}
int lcd_rebalance_exchange(lcd_comm_t *c, int n_jobs, const double *cost, const uint64_t *nbytes, const uint8_t *const *bufs, double tol,
                           int *n_out, double **cost_out, uint64_t **nbytes_out, uint8_t ***bufs_out, uint8_t **owned_out, lcd_rebalance_stats_t *st) {
    if (!c) return rb_err(-4, "lcd_rebalance_exchange: no communicator");
    if (c->dead) return rb_err(-53, "lcd_rebalance_exchange: the communicator was aborted by an earlier failure");
    if (n_jobs < 0) return rb_err(-4, "lcd_rebalance_exchange: negative queue depth");
    Scratch sc;
    bool in_epoch = false; // (a collective has been issued: the peers are waiting for this rank)
    auto fail = [&](int code, const std::string &m) {
        //  abort FIRST: ending a partly filled group would launch a subset of the planned send / recv pairs and can block in connection setup against
        // peers that posted the full set; ncclCommAbort is legal with a group pending.  The group is ended afterwards (its error is of no interest).  Without
        // ncclCommAbort in the library the communicator is leaked rather than destroyed: ncclCommDestroy on a communicator with operations outstanding can hang.
        if (in_epoch && c->comm) { if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm); c->comm = nullptr; c->dead = true; }
        if (sc.group_open) { (void)g_rccl.GroupEnd(); sc.group_open = false; }
        return rb_err(code, m);
    };
#define RB_HIP(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return fail(-10, "HIP call failed: " #x); } } while (0)
#define RB_NCCL(x) do { const int rc_ = (x); if (rc_ != 0) return fail(-53, std::string(#x ": ") + g_rccl.GetErrorString(rc_)); } while (0)
#define RB_MEM(p) do { if (!(p)) return fail(-3, "lcd_rebalance_exchange: out of memory (" #p ")"); } while (0)
    RB_HIP(hipSetDevice(c->device));
    const int W = c->world, me = c->rank;
    // 1. queue depths
    int *d_cnt = (int *)sc.dmalloc(sizeof(int) * (size_t)(W + 1)); RB_MEM(d_cnt);
    RB_HIP(hipMemcpyAsync(d_cnt + W, &n_jobs, sizeof(int), hipMemcpyHostToDevice, c->st));
    in_epoch = true;
    RB_NCCL(g_rccl.AllGather(d_cnt + W, d_cnt, 1, NCCL_INT32, c->comm, c->st));
    std::vector<int> cnt((size_t)W);
    RB_HIP(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * (size_t)W, hipMemcpyDeviceToHost, c->st));
    RB_HIP(hipStreamSynchronize(c->st));
    int maxn = 0; for (int x : cnt) { if (x < 0) return fail(-50, "lcd_rebalance_exchange: a rank reported a negative queue depth"); maxn = std::max(maxn, x); }
    // 2. the (cost, bytes) tables, padded to the deepest queue
    std::vector<double> tab((size_t)2 * maxn, 0.0), all((size_t)2 * maxn * W, 0.0);
    for (int i = 0; i < n_jobs; ++i) { tab[2 * (size_t)i] = cost[i]; tab[2 * (size_t)i + 1] = (double)nbytes[i]; } // (a job buffer is far below 2^53 bytes)
    if (maxn > 0) {
        double *d_tab = (double *)sc.dmalloc(sizeof(double) * (size_t)2 * maxn * (W + 1)); RB_MEM(d_tab);
        RB_HIP(hipMemcpyAsync(d_tab + (size_t)2 * maxn * W, tab.data(), sizeof(double) * (size_t)2 * maxn, hipMemcpyHostToDevice, c->st));
        RB_NCCL(g_rccl.AllGather(d_tab + (size_t)2 * maxn * W, d_tab, (size_t)2 * maxn, NCCL_FLOAT64, c->comm, c->st));
        RB_HIP(hipMemcpyAsync(all.data(), d_tab, sizeof(double) * (size_t)2 * maxn * W, hipMemcpyDeviceToHost, c->st));
        RB_HIP(hipStreamSynchronize(c->st));
    }
    std::vector<double> costs; std::vector<uint64_t> sizes; std::vector<size_t> first((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) { first[r + 1] = first[r] + (size_t)cnt[r]; for (int i = 0; i < cnt[r]; ++i) { costs.push_back(all[((size_t)r * maxn + i) * 2]); sizes.push_back((uint64_t)all[((size_t)r * maxn + i) * 2 + 1]); } }
    // 3. the plan
    std::vector<lcd_move_t> mv(costs.size() + 1); std::vector<double> lb((size_t)W), la((size_t)W);
    const int nm = lcd_rebalance_plan(W, cnt.data(), costs.data(), tol, -1, mv.data(), lb.data(), la.data());
    // 4. whole buffers, point to point, one group
    std::vector<char> sent((size_t)n_jobs, 0);
    struct Rx { double cost; uint64_t n; uint8_t *d; };
    std::vector<Rx> rx; std::vector<uint8_t *> d_tx; uint64_t moved_bytes = 0;
    for (int k = 0; k < nm; ++k) {
        const uint64_t nb = sizes[first[mv[k].src] + (size_t)mv[k].index]; moved_bytes += nb;
        if (mv[k].src == me) {
            uint8_t *d = (uint8_t *)sc.dmalloc(nb); RB_MEM(d);
            RB_HIP(hipMemcpyAsync(d, bufs[mv[k].index], nb, hipMemcpyHostToDevice, c->st));
            d_tx.push_back(d); sent[(size_t)mv[k].index] = 1;
        } else if (mv[k].dst == me) {
            uint8_t *d = (uint8_t *)sc.dmalloc(nb); RB_MEM(d);
            rx.push_back({costs[first[mv[k].src] + (size_t)mv[k].index], nb, d});
        }
    }
    if (nm > 0) {
        RB_NCCL(g_rccl.GroupStart());
        sc.group_open = true;
        size_t it = 0, ir = 0;
        for (int k = 0; k < nm; ++k) { // (the same order on every rank: the k-th transfer between a pair is the k-th on both sides)
            const uint64_t nb = sizes[first[mv[k].src] + (size_t)mv[k].index];
            if (mv[k].src == me) { RB_NCCL(g_rccl.Send(d_tx[it++], nb, NCCL_UINT8, mv[k].dst, c->comm, c->st)); }
            else if (mv[k].dst == me) { RB_NCCL(g_rccl.Recv(rx[ir++].d, nb, NCCL_UINT8, mv[k].src, c->comm, c->st)); }
        }
        sc.group_open = false;
        RB_NCCL(g_rccl.GroupEnd());
    }
    const int n_keep = (int)std::count(sent.begin(), sent.end(), 0), n_new = n_keep + (int)rx.size();
    double *co = (double *)sc.hmalloc(sizeof(double) * (size_t)(n_new + 1)); RB_MEM(co);
    uint64_t *no = (uint64_t *)sc.hmalloc(sizeof(uint64_t) * (size_t)(n_new + 1)); RB_MEM(no);
    uint8_t **bo = (uint8_t **)sc.hmalloc(sizeof(uint8_t *) * (size_t)(n_new + 1)); RB_MEM(bo);
    uint8_t *ow = (uint8_t *)sc.hmalloc((size_t)(n_new + 1)); RB_MEM(ow);
    int w = 0;
    for (int i = 0; i < n_jobs; ++i) if (!sent[(size_t)i]) { co[w] = cost[i]; no[w] = nbytes[i]; bo[w] = (uint8_t *)bufs[i]; ow[w] = 0; ++w; }
    for (const Rx &r : rx) {
        uint8_t *hbuf = (uint8_t *)sc.hmalloc(r.n); RB_MEM(hbuf);
        RB_HIP(hipMemcpyAsync(hbuf, r.d, r.n, hipMemcpyDeviceToHost, c->st));
        co[w] = r.cost; no[w] = r.n; bo[w] = hbuf; ow[w] = 1; ++w;
    }
    RB_HIP(hipStreamSynchronize(c->st));
    sc.keep_host(); // (from here on nothing fails: the host blocks are the caller's; the device buffers go with `sc`)
    *n_out = n_new; *cost_out = co; *nbytes_out = no; *bufs_out = bo; *owned_out = ow;
    if (st) {
        double mean = 0; for (double l : lb) mean += l; mean = W > 0 ? mean / W : 0.0;
        st->n_moves = nm; st->moved_bytes = moved_bytes; st->world = W;
        st->imbalance_before = mean > 0 ? *std::max_element(lb.begin(), lb.end()) / mean : 1.0;
        st->imbalance_after = mean > 0 ? *std::max_element(la.begin(), la.end()) / mean : 1.0;
        st->load_before_mine = lb[(size_t)me]; st->load_after_mine = la[(size_t)me]; st->jobs_before_mine = n_jobs; st->jobs_after_mine = n_new;
    }
    return 0;
#undef RB_HIP
#undef RB_NCCL
#undef RB_MEM
}

// What RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice): the evidence a scaling run prints per rank.  0, or < 0 if the
// communicator is dead / librccl lacks the calls
int lcd_comm_info(lcd_comm_t *c, int *nccl_world, int *nccl_rank, int *nccl_device) {
    if (!c || !c->comm || c->dead) return rb_err(-4, "lcd_comm_info: no live communicator");
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return rb_err(-52, "librccl lacks ncclCommCount / ncclCommUserRank");
    int n = -1, r = -1, d = -1;
    int rc = g_rccl.CommCount(c->comm, &n); if (rc != 0) return rb_err(-53, std::string("ncclCommCount: ") + g_rccl.GetErrorString(rc));
    rc = g_rccl.CommUserRank(c->comm, &r); if (rc != 0) return rb_err(-53, std::string("ncclCommUserRank: ") + g_rccl.GetErrorString(rc));
    if (g_rccl.CommCuDevice && g_rccl.CommCuDevice(c->comm, &d) != 0) d = -1;
    if (nccl_world) *nccl_world = n; if (nccl_rank) *nccl_rank = r; if (nccl_device) *nccl_device = d;
    return 0;
}
}
