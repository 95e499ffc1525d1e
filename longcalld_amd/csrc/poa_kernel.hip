// poa_kernel.hip -- K1/K2: partial-order alignment chains on gfx950 (CDNA4, wave64).
//
// Replaces what src/align.c:762-857 (abpoa_partial_aln_msa_cons) and :872-943 (abpoa_aln_msa_cons) ask
// of abPOA.  One workgroup owns one chain (= one abpoa_t life: region x haplotype for K1, region for K2)
// and keeps the whole graph build device-resident: align read -> backtrack -> add alignment -> re-sort ->
// next read, then MSA / clustering / consensus.  No host round trip inside a chain.
//
// Workgroup size = window / 4: 64 threads for the banded K1 rows and short K2 rows (256-column window), 128 ... 1024 threads for the
// unbanded K2 rows (up to 4 096 columns); a lane owns four consecutive columns.  Integer DP only (no MFMA):
//   * per read a "row plan" (CSR of the usable predecessors + edge bonus of every row) is built in parallel, so a row costs one
//     dependent load level instead of a linked-list walk; rows take it from a 64-row register window by v_readlane;
//   * H / E1 / E2 values live only in a K-slot LDS row ring (and, for the few rows that a far successor or the end node reads, in a
//     small HBM spill area); what is streamed to HBM is a 1-byte direction code per cell that replays the oracle's backtrack;
//   * the horizontal-gap recurrence is an in-lane prefix + one DPP wave prefix-max pair per 256 cells
//     (F[j] = max_k<j Hpre[k] - o - (j-k)e  ==  prefixmax(Hpre[k]+k e) - o - j e);
//   * banded rows (align_windowed): the adaptive band is pulled from the predecessors' row-max columns (no scatter), the window is
//     addressed column mod WIN so shifted windows need no per-cell bounds test;
//   * unbanded rows (align_unbanded) run systolic across the wavefronts: mailboxes + progress counters, no workgroup barrier per row;
//   * the backtrack (wavefront 0) speculates runs of matches 64 steps at a time;
//   * the graph update is parallel (prefix sums), the re-sort's serial Kahn walk runs on packed 16-bit words in LDS (or HBM for graphs
//     that do not fit the chain's pool);
//   * rows wider than the window fall back to the generic rows of align_to_subgraph (int32 H/E1/E2 planes in HBM, value backtrack).
// Semantics are defined by oracle/poa.c (see its header); this file must match it bit for bit.
#include <hip/hip_runtime.h>
#include <mutex>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {

#ifndef LCD_UPD_U
#define LCD_UPD_U 6 // path entries per thread in flight in the graph update (build switch; 4 / 6 / 8 measured in round 6: 6 and 8 are ~1 % ahead -- a 650-entry path is two chunks of 384 instead of three of 256)
#endif
#ifndef LCD_PLAN_U
#define LCD_PLAN_U 6 // rows per thread in flight in the plan build (build switch; see LCD_UPD_U)
#endif

constexpr int MAXP = 64;   // predecessors staged in LDS per row (more are read from the plan in HBM)
constexpr int MAXW = 16;   // wavefronts per workgroup

struct Smem { // small per-workgroup state of every class (static LDS counts against the workgroups-per-CU budget of the narrow classes)
    int tot1[2][MAXW], tot2[2][MAXW];
    int bh[MAXW], bl[MAXW], br[MAXW];
    int rm[4][5];          // ring slot meta: beg, end, HBM offset, row-max leftmost / rightmost column
    int scan[MAXW];
    int bc[8];
    unsigned long long prof[4];
};
struct SmemWide { // multi-wavefront classes only (never referenced by the 64-thread kernel, so it costs that class no LDS)
    int pb[MAXP], pe[MAXP], bonus[MAXP], pml[MAXP], pmr[MAXP];
    unsigned po[MAXP];
    int ppi[MAXP], pslot[MAXP];
    // mailboxes of the systolic unbanded rows (align_unbanded): progress counter, F-scan carry, boundary H, per wavefront
    int prog[MAXW];
    int carry1[16][MAXW], carry2[16][MAXW], bndH[16][MAXW];
};

#ifdef LCD_X_BYTESTAT
__device__ unsigned long long g_bs_tot[15][19]; // (experiment) bytes by (class, array group), + chains, reads, cells
__device__ unsigned g_bs_done[256]; // finished chains per launch (keyed by the launch's chain table)
#endif
__shared__ SmemWide g_wide;
__shared__ Smem g_smem; // file scope: non-inlined device functions reach it as LDS (a Smem& parameter would be a generic pointer -> flat ops)

template <int NT> struct Cfg;
// One LDS pool per workgroup: [row ring | query cache] during the DP, re-used as 16-bit graph arrays by the re-sort.
// (sizes are per launch: PoaChain.wmax columns per ring slot, PoaChain.lds_words in total; only the slot count is per class)
template <> struct Cfg<64> { static constexpr int K = 2; };
template <> struct Cfg<128> { static constexpr int K = 2; };
template <> struct Cfg<256> { static constexpr int K = 2; };
template <> struct Cfg<512> { static constexpr int K = 2; };
template <> struct Cfg<1024> { static constexpr int K = 2; };

struct Ctx {
    int *H, *E1, *E2;                 // generic rows: values in HBM
    uint8_t *code8; int *ord, *spill;  // windowed rows: direction codes / predecessor ordinals / spilled value rows (same arena bytes)
    int *rbeg, *rend; uint32_t *roff, *ooff, *spoff;
    int *ml, *mr, *idx2node, *node2idx, *remain, *deg, *queue;
    int *out_head, *out_tail, *in_head, *in_tail, *nin, *aligned;
    int *e_from, *e_to, *e_w, *e_next_out, *e_next_in;
    unsigned long long *rid;
    int *cig_node, *cig_qpos, *cig_node0, *cig_qpos0;
    uint8_t *base, *imap;
    int *het, *clu, *nclu; uint8_t *prof;
    int *pl_start, *pl_pidx, *pl_bonus, *pl_rem; uint8_t *pl_base;
    int *e_slot; int plan_valid, plan_bi, plan_ei, plan_rend; // the plan in the arrays above is the one of (bi, ei, remain_end) and matches the graph (see build_plan)
    int *aa_node, *aa_flag, *aa_eid;
    int *tb;                          // column-tile boundaries of the unbanded rows: 4 x node_cap ints (H of the last column, by tile parity; F carries)
    long long alg_adjust;             // cells of the reference's algorithm minus cells computed (certified band: full rows minus the intervals, attempts included)
    int cert_generic, cert_generic_seen, cert_sest, cert_ubtop, cert_bztop; unsigned long long cert_cells0; // the read at hand goes through the generic rows over its intervals (align_certified)
    int *cert; int cert_on, cert_hist; // certified band of a K2 chain (align_certified): 7 x node_cap ints; largest bound-to-score slack of the chain's reads so far
    int wmax, seq_cap, pool_words, spill_x, ring_k, plan_k, solo;
    unsigned long long wd_deadline;   // shader-clock tick after which the chain gives up (LCD_ERR_WATCHDOG): checked once per 64 DP rows, per read, per 256 backtrack steps
    int ring16;                        // the chain's LDS ring holds 16-bit values (align_lean<2, C> only; the generic rows then see half as many int32 columns)
    int topo_mode;                     // test switches of the re-sort (LCD_DBG bits 64 / 128 / 256): 1 = never the compact LDS copy, 2 = the compact copy even where the packed words fit, 4 = its FIFO holds three nodes
    int mm_valid;                      // g.deg / g.queue hold, by topological index, every row's smallest predecessor index / largest successor index (topo_sort_block; subgraph_nodes_wave0)
    uint8_t *cut; int cut_valid;       // cut[i] != 0: the Kahn walk's FIFO was empty right after it popped the node of topological index i (where topo_sort_incremental may restart the walk); same bytes as `prof` (chain_output only)
    int upd_new, upd_newe, upd_moved;  // add_alignment_block's last call: nodes / edges it created, and whether some node's heaviest out-edge is another one now
    int inc_off;                       // LCD_DBG bit 1024: every topology change takes the full re-sort (test switch)
#ifdef LCD_X_PHASESTAT
    unsigned long long pstat[24];      // (experiment) ticks inside the per-read phases, see LCD_PT
#endif
#ifdef LCD_X_BYTESTAT
    unsigned long long bstat[16];      // (experiment) bytes the chain's memory instructions ask for, by array group, see LCD_BS
#endif
#ifdef LCD_X_INCSTAT
    unsigned inc_stat[13];             // (experiment) topo_sort_incremental: refusals by reason 0..9, successes, nodes walked, pieces
#endif
    int n_node, n_edge, node_cap, edge_cap, rid_words;
    unsigned long long cell_cap;
    int status;
    unsigned long long t_dp, t_bt, t_plan, t_poll, t_kahn, t_bp, t_setup;
};

__device__ __forceinline__ int ilog2_32(int v) { return 31 - __clz(v); }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

__device__ __forceinline__ int wave_max(int v) {
    for (int d = 32; d >= 1; d >>= 1) v = imax(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
    for (int d = 32; d >= 1; d >>= 1) v = imin(v, __shfl_xor(v, d));
    return v;
}
// inclusive prefix max over lanes
__device__ __forceinline__ int wave_incl_prefix_max(int v, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(v, d);
        if (lane >= d) v = imax(v, y);
    }
    return v;
}
__device__ __forceinline__ int wave_incl_prefix_add(int v, int lane) {
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(v, d);
        if (lane >= d) v += y;
    }
    return v;
}

// ---- DPP wave scans (gfx9 row_shr / row_bcast forms; all 64 lanes must be active) ----
template <int CTRL, int ROWMASK>
__device__ __forceinline__ int dpp_take(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROWMASK, 0xf, false); }
#define LCD_DPP_SCAN(OP, ID)                                        \
    v = OP(v, dpp_take<0x111, 0xf>(ID, v)); v = OP(v, dpp_take<0x112, 0xf>(ID, v)); \
    v = OP(v, dpp_take<0x114, 0xf>(ID, v)); v = OP(v, dpp_take<0x118, 0xf>(ID, v)); \
    v = OP(v, dpp_take<0x142, 0xa>(ID, v)); v = OP(v, dpp_take<0x143, 0xc>(ID, v));
__device__ __forceinline__ int iadd(int a, int b) { return a + b; }
__device__ __forceinline__ int scan_max(int v) { LCD_DPP_SCAN(imax, LCD_NEG * 2) return v; }
__device__ __forceinline__ int scan_min(int v) { LCD_DPP_SCAN(imin, (1 << 30)) return v; }
__device__ __forceinline__ int scan_add(int v) { LCD_DPP_SCAN(iadd, 0) return v; }
// two inclusive prefix-max scans interleaved: v_max_i32_dpp leaves lanes without a source untouched (no bound_ctrl), which
// is exactly max(v, nothing); the other scan's instruction + s_nop 0 fill the 2 wait states a DPP read needs after a write
__device__ __forceinline__ void scan_max2(int &a, int &b) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b));
}
// three inclusive prefix-max scans interleaved: each chain's next DPP read is two instructions behind its write, which is the wait a DPP operand needs -- no s_nop
// inside.  (The third chain is the adaptive band's row maximum: H = max(Hpre, F) and F[j] <= Hpre[k] - (o + e) for some k < j, so the row's maximum and the columns
// that reach it are those of Hpre -- known before the F scan, not after it.)
__device__ __forceinline__ void scan_max3(int &a, int &b, int &c) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_max_i32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_max_i32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_max_i32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(a), "+v"(b), "+v"(c));
}
__device__ __forceinline__ int shr1(int identity, int v) { return dpp_take<0x138, 0xf>(identity, v); } // wave_shr:1
__device__ __forceinline__ int lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }


// LDS-only workgroup barrier: orders LDS traffic without draining the HBM store queue (vmcnt), which a
// __syncthreads() would do.  Single-wavefront workgroups need no s_barrier at all (LDS ops of one wave are in order).
template <int NT>
__device__ __forceinline__ void lds_barrier() {
    if (NT > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("" ::: "memory");
}
#define LCD_RL(v, t) __builtin_amdgcn_readlane((v), (t))
__device__ __forceinline__ int glb_ld(const int *p);
__device__ __forceinline__ int glb_ld_u8(const uint8_t *p);
__device__ __forceinline__ int usgpr(const int v);

// (experiment, -DLCD_X_BYTESTAT) the HBM account of a chain: bytes its memory instructions ask for (elements x element size, per lane -- not sectors or cache lines),
// by array group, from the trip counts the phases actually ran with.  tools/hbm_account.py sums the chains' lines and sets them against the PMC counters.
//  0 direction codes written   1 row metadata written (rbeg, rend, roff)   2 plan arrays read by the rows   3 read bases staged
//  4 backtrack reads (codes, row metadata, order)   5 path (cigar) written + read   6 graph update (node / edge / read-set arrays, path scratch)
//  7 plan build: graph arrays read   8 plan build: plan arrays written   9 full re-sort   10 incremental re-sort   11 remain by pointer jumping
//  12 chain output (MSA rows, consensus)   13 certified-band node arrays + intervals   14 generic rows (int32 planes)   15 per-row extremes (compute_mm)
#ifdef LCD_X_BYTESTAT
#define LCD_BS(k, v) do { g.bstat[k] += (unsigned long long)(v); } while (0)
#else
#define LCD_BS(k, v) do { } while (0)
#endif
// (experiment, -DLCD_X_PHASESTAT) ticks between two marks of a per-read phase, outstanding memory operations drained at each mark
#ifdef LCD_X_PHASESTAT
#define LCD_PT0() long long pt_ = clock64()
#define LCD_PT(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long t_ = clock64(); g.pstat[k] += (unsigned long long)(t_ - pt_); pt_ = t_; } while (0)
#else
#define LCD_PT0() do { } while (0)
#define LCD_PT(k) do { } while (0)
#endif

// ---------------- graph mutation (thread 0 only) ----------------
__device__ int add_node(Ctx &g, uint8_t b) {
    if (g.n_node >= g.node_cap) { g.status = LCD_ERR_NODES; return g.node_cap - 1; }
    int id = g.n_node++;
    g.base[id] = b; g.out_head[id] = g.out_tail[id] = g.in_head[id] = g.in_tail[id] = -1; g.nin[id] = 0; g.aligned[id] = id;
    return id;
}
__device__ void add_edge(Ctx &g, int from, int to, int check, int read_id) {
    if (check) {
        for (int e = g.out_head[from]; e >= 0; e = g.e_next_out[e])
            if (g.e_to[e] == to) {
                g.e_w[e] += 1;
                g.rid[(size_t)e * g.rid_words + (read_id >> 6)] |= 1ull << (read_id & 63);
                return;
            }
    }
    if (g.n_edge >= g.edge_cap) { g.status = LCD_ERR_EDGES; return; }
    int e = g.n_edge++;
    g.e_from[e] = from; g.e_to[e] = to; g.e_w[e] = 1; g.e_next_out[e] = -1; g.e_next_in[e] = -1;
    for (int k = 0; k < g.rid_words; ++k) g.rid[(size_t)e * g.rid_words + k] = 0;
    g.rid[(size_t)e * g.rid_words + (read_id >> 6)] |= 1ull << (read_id & 63);
    if (g.out_tail[from] < 0) g.out_head[from] = e; else g.e_next_out[g.out_tail[from]] = e;
    g.out_tail[from] = e;
    if (g.in_tail[to] < 0) g.in_head[to] = e; else g.e_next_in[g.in_tail[to]] = e;
    g.in_tail[to] = e; g.nin[to] += 1;
}

// Kahn BFS order + remain (oracle/poa.c topo_sort)
__device__ void topo_sort(Ctx &g) {
    const int n = g.n_node;
    for (int i = 0; i < n; ++i) g.deg[i] = g.nin[i];
    int qh = 0, qt = 0, index = 0;
    g.queue[qt++] = 0;
    while (qh < qt) {
        int cur = g.queue[qh++];
        g.idx2node[index] = cur; g.node2idx[cur] = index++;
        if (cur == 1) break;
        for (int e = g.out_head[cur]; e >= 0; e = g.e_next_out[e]) {
            int out = g.e_to[e];
            int d = g.deg[out] - 1; g.deg[out] = d;
            if (d == 0) {
                bool ok = true;
                for (int a = g.aligned[out]; a != out; a = g.aligned[a]) if (g.deg[a] != 0) { ok = false; break; }
                if (!ok) continue;
                g.queue[qt++] = out;
                for (int a = g.aligned[out]; a != out; a = g.aligned[a]) g.queue[qt++] = a;
            }
        }
    }
    if (index != n) { g.status = LCD_ERR_TOPO; return; }
    g.remain[1] = -1;
    for (int i = n - 2; i >= 0; --i) {
        int v = g.idx2node[i], mw = -1, mid = 1;
        for (int e = g.out_head[v]; e >= 0; e = g.e_next_out[e])
            if (g.e_w[e] > mw) { mw = g.e_w[e]; mid = g.e_to[e]; }
        g.remain[v] = g.remain[mid] + 1;
    }
}

__device__ bool add_alignment(Ctx &g, int beg_node, int end_node, const uint8_t *seq, int len, int n_cig, int read_id) {
    if (g.n_node == 2) {
        int last = 0;
        for (int i = 0; i < len; ++i) { int id = add_node(g, seq[i]); add_edge(g, last, id, 0, read_id); last = id; }
        add_edge(g, last, 1, 0, read_id);
        return true;
    }
    if (n_cig == 0) return false;
    int last = beg_node, last_new = 0;
    for (int i = 0; i < n_cig; ++i) {
        uint8_t b = seq[g.cig_qpos[i]];
        int node = g.cig_node[i];
        if (node >= 0) {
            if (g.base[node] != b) {
                int a = -1;
                for (int x = g.aligned[node]; x != node; x = g.aligned[x]) if (g.base[x] == b) { a = x; break; }
                if (a != -1) { add_edge(g, last, a, 1 - last_new, read_id); last = a; last_new = 0; }
                else {
                    int nid = add_node(g, b);
                    add_edge(g, last, nid, 0, read_id); last = nid; last_new = 1;
                    g.aligned[nid] = g.aligned[node]; g.aligned[node] = nid;
                }
            } else { add_edge(g, last, node, 1 - last_new, read_id); last = node; last_new = 0; }
        } else {
            int nid = add_node(g, b);
            add_edge(g, last, nid, 0, read_id); last = nid; last_new = 1;
        }
    }
    add_edge(g, last, end_node, 1 - last_new, read_id);
    return true;
}

// ---- parallel graph update (oracle/poa.c add_alignment; abpoa_add_subgraph_alignment) ----
// Almost every cigar entry re-uses an existing node along an existing edge, and a read's path visits every node at most
// once, so the entries are independent: resolve each entry's node in parallel, number the new nodes / new edges by prefix
// sums in cigar order (== the ids the serial algorithm hands out), then create and link them in parallel -- every node
// gains at most one out-edge, one in-edge and one aligned sibling per read, so the list appends never collide.
template <int NT>
__device__ int block_excl_scan(int v, Smem &sm, int *total) { // exclusive prefix sum over the NT threads; one call = two barriers
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    const int incl = scan_add(v);
    if constexpr (NW == 1) { *total = lane63(incl); return incl - v; } // (one wavefront: no exchange, and no barrier -- a barrier drains the stores the caller has in flight, ~2 us each time)
    if (lane == 63) sm.scan[wave] = incl;
    lds_barrier<NT>(); // (the partial sums go through LDS only: no need to wait for the caller's HBM stores)
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) { const int t = sm.scan[k]; if (k < wave) woff += t; tot += t; }
    lds_barrier<NT>();
    *total = tot;
    return woff + incl - v;
}

// returns 0: nothing changed, 1: only edge weights / read sets changed and some node's heaviest out-edge is another one now (the topological order stands, `remain`
// does not), 3: only weights / read sets changed and every heaviest out-edge is the same (order and `remain` stand), 2: nodes or edges were added
template <int NT>
__device__ __attribute__((noinline)) int add_alignment_block(Ctx &g, Smem &sm, int beg_node, int end_node, const uint8_t *seq, int len, int n_cig, int read_id) {
    const int tid = threadIdx.x;
    const int rw = read_id >> 6; const unsigned long long rbit = 1ull << (read_id & 63);
    if (g.n_node == 2) { // first read: a chain source -> bases -> sink (abpoa_add_graph_sequence)
        if (len + 2 > g.node_cap || len + 1 > g.edge_cap) { g.status = LCD_ERR_NODES; return 0; }
        for (int i = tid; i < len; i += NT) {
            const int id = 2 + i;
            g.base[id] = seq[i]; g.in_head[id] = g.in_tail[id] = i; g.out_head[id] = g.out_tail[id] = i + 1; g.nin[id] = 1; g.aligned[id] = id;
        }
        for (int e = tid; e <= len; e += NT) {
            g.e_from[e] = e == 0 ? 0 : 1 + e; g.e_to[e] = e == len ? 1 : 2 + e; g.e_w[e] = 1; g.e_next_out[e] = -1; g.e_next_in[e] = -1;
            for (int k = 0; k < g.rid_words; ++k) g.rid[(size_t)e * g.rid_words + k] = k == rw ? rbit : 0ull;
        }
        if (tid == 0) { g.out_head[0] = g.out_tail[0] = 0; g.in_head[1] = g.in_tail[1] = len; g.nin[1] = 1; }
        g.n_node = len + 2; g.n_edge = len + 1;
        g.upd_new = len; g.upd_newe = len + 1; g.upd_moved = 0;
        __syncthreads();
        return 2;
    }
    if (n_cig == 0) return 0;
    // Every phase below is a chain of dependent loads per cigar entry (entry -> node -> its lists -> edge fields), and a workgroup -- one wavefront for most
    // chains -- that walks the entries 64 at a time pays that chain's latency once per 64 entries.  U entries per thread are in flight instead: the loads of all U
    // are issued before anything is stored (the stores are to int arrays the compiler must assume aliased), the prefix sums then run batch by batch in entry order.
    constexpr int U = LCD_UPD_U;
    // phase 1: the node each entry lands on: existing (>= 0), new and aligned to an anchor, or plain new
    int carry = 0;
    LCD_PT0();
    for (int base = 0; base < n_cig; base += U * NT) {
        int isnew[U], flag[U], T[U];
        int qp[U], nd[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + u * NT + tid; qp[u] = 0; nd[u] = -1; if (i < n_cig) { qp[u] = g.cig_qpos[i]; nd[u] = g.cig_node[i]; } }
        int bq[U], bn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = base + u * NT + tid; bq[u] = 0; bn[u] = 0; if (i < n_cig) { bq[u] = seq[qp[u]]; if (nd[u] >= 0) bn[u] = g.base[nd[u]]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * NT + tid;
            isnew[u] = 0; flag[u] = -2; T[u] = 0;
            if (i < n_cig) {
                const int node = nd[u];
                if (node >= 0) {
                    if (bn[u] == bq[u]) T[u] = node;
                    else {
                        int a = -1;
                        for (int x = g.aligned[node]; x != node; x = g.aligned[x]) if (g.base[x] == bq[u]) { a = x; break; }
                        if (a >= 0) T[u] = a; else { isnew[u] = 1; flag[u] = node; }
                    }
                } else { isnew[u] = 1; flag[u] = -1; }
            }
        }
        LCD_PT(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * NT >= n_cig) break; // (uniform)
            const int i = base + u * NT + tid;
            int tot;
            const int rank = block_excl_scan<NT>(isnew[u], sm, &tot);
            if (i < n_cig) { g.aa_node[i] = isnew[u] ? g.n_node + carry + rank : T[u]; g.aa_flag[i] = flag[u]; }
            carry += tot;
        }
        LCD_PT(1);
    }
    const int n_new = carry;
    if (g.n_node + n_new > g.node_cap) { g.status = LCD_ERR_NODES; return 0; }
    __syncthreads();
    LCD_PT(2);
    // phase 2: new nodes
    if (n_new > 0)
    for (int i = tid; i < n_cig; i += NT) {
        const int flag = g.aa_flag[i];
        if (flag == -2) continue;
        const int id = g.aa_node[i];
        g.base[id] = seq[g.cig_qpos[i]]; g.out_head[id] = g.out_tail[id] = g.in_head[id] = g.in_tail[id] = -1; g.nin[id] = 0;
        if (flag >= 0) { g.aligned[id] = g.aligned[flag]; g.aligned[flag] = id; } else g.aligned[id] = id;
    }
    LCD_PT(3);
    // phase 3: edges j = 0..n_cig (from path[j-1] to path[j]); existing ones gain weight + read id, the others are numbered
    carry = 0;
    int heavy_moved = 0;
    for (int base = 0; base <= n_cig; base += U * NT) {
        int need[U], from[U], to[U], oh[U]; bool chk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = base + u * NT + tid;
            need[u] = 0; from[u] = 0; to[u] = 0; chk[u] = false; oh[u] = -1;
            if (j <= n_cig) {
                from[u] = j == 0 ? beg_node : g.aa_node[j - 1]; to[u] = j == n_cig ? end_node : g.aa_node[j];
                const bool from_new = j > 0 && g.aa_flag[j - 1] != -2, to_new = j < n_cig && g.aa_flag[j] != -2;
                need[u] = 1; chk[u] = !from_new && !to_new;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (chk[u]) oh[u] = g.out_head[from[u]];
        // the first out-edge of every entry's node (the usual case: the one edge of a backbone node) in flight together
        int e0w[U], e0t[U], e0n[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { e0w[u] = 0; e0t[u] = -1; e0n[u] = -1; if (oh[u] >= 0) { e0w[u] = g.e_w[oh[u]]; e0t[u] = g.e_to[oh[u]]; e0n[u] = g.e_next_out[oh[u]]; } }
        LCD_PT(4);
        int fnd[U], wnew[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fnd[u] = -1; wnew[u] = 0;
            if (!chk[u]) continue;
            // the edge exists: one more read on it.  `remain` follows every node's HEAVIEST out-edge (first maximum in list order): it only has to be
            // recomputed if this increment changes which edge that is -- reads on the majority path never do
            int found = -1, pos_found = 0, amax = -1, pos_amax = 0, wmax = -1, pos = 0, wfound = 0;
            for (int e = oh[u]; e >= 0; ++pos) {
                int wt, et, en;
                if (pos == 0) { wt = e0w[u]; et = e0t[u]; en = e0n[u]; } else { wt = g.e_w[e]; et = g.e_to[e]; en = g.e_next_out[e]; }
                if (wt > wmax) { wmax = wt; amax = e; pos_amax = pos; }
                if (found < 0 && et == to[u]) { found = e; pos_found = pos; wfound = wt; }
                e = en;
            }
            if (found >= 0) {
                const int wn = wfound + 1;
                fnd[u] = found; wnew[u] = wn; need[u] = 0;
                if (found != amax && (wn > wmax || (wn == wmax && pos_found < pos_amax))) heavy_moved = 1;
            }
        }
        // the read sets and (where a weight reached a power of two: the edge's bonus, ilog2 of its weight, goes up) the edges' plan entries of all U edges are fetched
        // together, then everything is stored: a path visits an edge once, so no two of these touch the same words -- and a load behind each store would be U round
        // trips one after the other
        unsigned long long rv[U]; int sl[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rv[u] = 0; sl[u] = -1;
            if (fnd[u] >= 0) {
                rv[u] = g.rid[(size_t)fnd[u] * g.rid_words + rw];
                if (g.plan_valid && (wnew[u] & (wnew[u] - 1)) == 0) sl[u] = g.e_slot[fnd[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (fnd[u] >= 0) {
                g.e_w[fnd[u]] = wnew[u]; g.rid[(size_t)fnd[u] * g.rid_words + rw] = rv[u] | rbit;
                if (sl[u] >= 0) g.pl_bonus[sl[u]] = ilog2_32(wnew[u]);
            }
        LCD_PT(5);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * NT > n_cig) break; // (uniform)
            const int j = base + u * NT + tid;
            int tot;
            const int rank = block_excl_scan<NT>(need[u], sm, &tot);
            if (j <= n_cig) g.aa_eid[j] = need[u] ? g.n_edge + carry + rank : -1;
            carry += tot;
        }
        LCD_PT(6);
    }
    const int n_newe = carry;
    if (g.n_edge + n_newe > g.edge_cap) { g.status = LCD_ERR_EDGES; return 0; }
    __syncthreads(); // new-node fields (phase 2) and edge numbers are visible
    // phase 4: create + link the new edges
    if (n_newe > 0)
    for (int j = tid; j <= n_cig; j += NT) {
        const int e = g.aa_eid[j];
        if (e < 0) continue;
        const int from = j == 0 ? beg_node : g.aa_node[j - 1], to = j == n_cig ? end_node : g.aa_node[j];
        g.e_from[e] = from; g.e_to[e] = to; g.e_w[e] = 1; g.e_next_out[e] = -1; g.e_next_in[e] = -1;
        for (int k = 0; k < g.rid_words; ++k) g.rid[(size_t)e * g.rid_words + k] = k == rw ? rbit : 0ull;
        const int ot = g.out_tail[from];
        if (ot < 0) g.out_head[from] = e; else g.e_next_out[ot] = e;
        g.out_tail[from] = e;
        const int it = g.in_tail[to];
        if (it < 0) g.in_head[to] = e; else g.e_next_in[it] = e;
        g.in_tail[to] = e; g.nin[to] += 1;
    }
    g.n_node += n_new; g.n_edge += n_newe;
    const int moved = __syncthreads_or(heavy_moved);
    LCD_PT(7);
    g.upd_new = n_new; g.upd_newe = n_newe; g.upd_moved = moved;
    return (n_new || n_newe) ? 2 : moved ? 1 : 3;
}

// A parallel pass over [lo, hi) with U elements per thread IN FLIGHT: `ld` (straight-line loads, no side effects; called with an index clamped into the range)
// runs for all U before `st` runs for any, so the U dependent-load chains overlap instead of following one another -- the per-read graph phases are latency
// chains through L2 / HBM run by one wavefront, and a loop that stores into an int array between two loads is a chain per iteration for the compiler.
#ifndef LCD_BF_U
#define LCD_BF_U 4 // elements per thread in flight in the per-read O(nodes) passes (build switch; 8 measured in round 5: see profiles/NOTES_r05.md)
#endif
template <int U, int NT, typename LoadF, typename StoreF>
__device__ __forceinline__ void batched_for(const int lo, const int hi, LoadF ld, StoreF st) {
    for (int b = lo + (int)threadIdx.x; b < hi; b += U * NT) {
        decltype(ld(0)) r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = b + u * NT; r[u] = ld(i < hi ? i : hi - 1); }
#pragma unroll
        for (int u = 0; u < U; ++u) { const int i = b + u * NT; if (i < hi) st(i, r[u]); }
    }
}

// Kahn BFS order + remain for the whole workgroup: the pointer-chasing part still runs on one lane (the FIFO order is
// inherently serial) but on 16-bit copies of the graph staged in LDS, so each dependent step costs an LDS access (~60 clk)
// instead of an HBM/L2 access (~500 clk); staging in and out is a coalesced parallel copy.  Falls back to HBM when the
// graph does not fit the pool.
// LDS budget: 8 B per node + 4 B per edge (the pool size decides how many single-wavefront chains fit a CU, and those chains are
// most of the work): deg | queue (= the topological order itself) | out_head | aligned ring ; edge next | edge to.  node2idx goes
// straight to HBM (stores do not stall the walk); after the walk `aligned` is overwritten by each node's heaviest successor
// (computed in parallel from the edge weights in HBM) and `deg` by remain + 1.
// The walk on 16-bit packed words: `deg`, `queue` (n halfwords each), node words `nw` (n), edge words `ew` (E) live in LDS when the
// graph fits the workgroup's pool, in HBM scratch (the row-plan arrays, free between two reads) when it does not -- the single-wavefront
// chains are kept to a small LDS pool on purpose (more of them per CU is worth more than a fast re-sort, DESIGN "Submission").
// remain[v] = number of nodes on the heaviest path from v to the sink = depth of v in the tree "v -> heaviest successor" (root: the sink).
// Serial in topological order it is one dependent load chain of length n; by pointer jumping it is ceil(log2 n) rounds over all nodes:
// (P, D) <- (P[P], D + D[P]).  Two (P, D) buffers of 16-bit entries in the row-plan arrays (HBM, free between two reads); the caller has
// filled buffer 0: P = heaviest successor (the sink points to itself), D = 1 (sink: 0).
template <int NT, typename U16P /* unsigned short * in HBM or in LDS (address space 3) */>
__device__ __forceinline__ void remain_by_jumping(Ctx &g, const int n, U16P b0, U16P b1) {
    const int tid = threadIdx.x;
    U16P buf[2] = {b0, b1}; // each: P[n] | D[n]
    const int rounds = n > 2 ? 32 - __clz(n - 1) : 1;
    int src = 0;
    for (int r = 0; r < rounds; ++r) {
        U16P Ps = buf[src], Ds = Ps + n;
        U16P Pd = buf[src ^ 1], Dd = Pd + n;
        struct PD { int pp, d; };
        batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { const int p = Ps[v]; PD r; r.pp = Ps[p]; r.d = Ds[v] + Ds[p]; return r; },
                           [&](const int v, const PD r) { Pd[v] = (unsigned short)r.pp; Dd[v] = (unsigned short)r.d; });
        __syncthreads();
        src ^= 1;
    }
    U16P D = buf[src] + n;
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { return (int)D[v]; }, [&](const int v, const int d) { g.remain[v] = d - 1; });
}
typedef __attribute__((address_space(3))) unsigned short *lcd_lds_u16p;
__device__ __forceinline__ lcd_lds_u16p lds_u16(const void *p) { return (lcd_lds_u16p)(uintptr_t)(unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)p; }

template <int NT, bool INLDS>
__device__ __forceinline__ void topo_sort_arrays(Ctx &g, Smem &sm, unsigned short *deg, unsigned short *queue, unsigned *nw, unsigned *ew) {
    const int tid = threadIdx.x;
    const int n = g.n_node, E = g.n_edge;
    struct NW { int d; unsigned w; };
    batched_for<LCD_BF_U, NT>(0, n, [&](const int i) { NW r; r.d = g.nin[i]; r.w = (unsigned)(g.out_head[i] + 1) | ((unsigned)g.aligned[i] << 16); return r; },
                       [&](const int i, const NW r) { deg[i] = (unsigned short)r.d; nw[i] = r.w; });
    batched_for<LCD_BF_U, NT>(0, E, [&](const int e) { return (unsigned)g.e_to[e] | ((unsigned)(g.e_next_out[e] + 1) << 16); }, [&](const int e, const unsigned w) { ew[e] = w; });
    // Chain links for the walk below: link(v) = w when v's only out-edge goes to w, w has no other in-edge and no aligned ring -- when v is
    // popped with nothing else queued, w is the next node whatever else happens.  POA graphs are mostly such chains (the backbone between
    // bubbles), so the walk takes them 64 nodes at a time: jump tables J1 = link, J4 = link^4, J16 = link^16 (self-loops at chain ends) let
    // lane t of wavefront 0 reach link^t(v) in <= 9 loads.  The tables live in the row-plan arrays (HBM, free between two reads).
    unsigned short *J1 = (unsigned short *)g.pl_start, *J4 = J1 + n, *J16 = (unsigned short *)g.pl_rem;
    __syncthreads();
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) {
        const unsigned e = nw[v] & 0xffffu;
        const unsigned w = ew[e != 0 ? e - 1 : 0];      // (straight-line: the loads of a node without an out-edge are harmless)
        const int to = (int)(w & 0xffffu);
        const bool link = e != 0 && (w >> 16) == 0 && deg[to] == 1 && (int)(nw[to] >> 16) == to;
        return link ? to : v;
    }, [&](const int v, const int l) { J1[v] = (unsigned short)l; });
    __syncthreads();
    // RL[v] = length of the chain that starts at v, capped at 16: a jump costs up to 9 dependent loads, so chains shorter than 4 are walked
    // node by node (graphs of noisy reads are bubbles every few nodes: probing every node for a chain cost more than it saved)
    unsigned char *RL = (unsigned char *)(J16 + n), *R4 = RL + n; // (second half of pl_rem: 2n bytes)
    struct XR { int x, r; };
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { XR o; o.x = v; o.r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int y = J1[o.x]; o.r += y != o.x; o.x = y; }
        return o; }, [&](const int v, const XR o) { J4[v] = (unsigned short)o.x; R4[v] = (unsigned char)o.r; });
    __syncthreads();
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { XR o; o.x = v; o.r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o.r += R4[o.x]; o.x = J4[o.x]; }
        return o; }, [&](const int v, const XR o) { J16[v] = (unsigned short)o.x; RL[v] = (unsigned char)o.r; });
    __syncthreads();
    const long long tk0 = clock64();
    if (tid < 64) { // wavefront 0, every lane with the same scalars (loads broadcast, identical stores coincide); lanes differ only in the chain step
        const int lane = tid;
        int qh = 0, qt = 0, index = 0;
        queue[qt++] = 0;
        int cur = 0; // (the node at queue[qh] is kept in a register whenever it is the one just pushed: linear stretches never re-read the queue)
        bool have = true;
        while (qh < qt) {
            if (!have) cur = queue[qh];
            ++qh; have = false;
            while (qh == qt) { // nothing else queued: follow the chain that starts at cur, 64 nodes per step
                if (RL[cur] < 4) break;
                int x = cur;
                const int c16 = lane >> 4, c4 = (lane >> 2) & 3, c1 = lane & 3;
                for (int i = 0; i < 3; ++i) if (i < c16) x = J16[x];
                for (int i = 0; i < 3; ++i) if (i < c4) x = J4[x];
                for (int i = 0; i < 3; ++i) if (i < c1) x = J1[x];
                const int prev = __shfl_up(x, 1);
                const unsigned long long adv = __ballot(lane == 0 || x != prev); // lanes still advancing: a prefix (a chain end is a self-loop)
                const int L = __popcll(adv);
                if (L <= 1) break;
                // x_0 .. x_{L-2} are complete (their one out-edge leads to the next node, which nothing else waits for); x_{L-1} goes on
                if (lane < L - 1) { g.node2idx[x] = index + lane; g.cut[index + lane] = 1; }
                if (lane >= 1 && lane < L) queue[qh - 1 + lane] = (unsigned short)x;
                index += L - 1; qh += L - 1; qt = qh;
                cur = __shfl(x, L - 1);
                if (L < 64 || index > n) break; // (index > n: the jump tables do not describe chains -- the count below turns it into LCD_ERR_TOPO)
            }
            g.node2idx[cur] = index; g.cut[index] = (uint8_t)(qh == qt); ++index; // (idx2node is the queue itself, copied out below)
            if (cur == 1 || index > n) break;
            for (unsigned e = nw[cur] & 0xffffu; e != 0;) {
                const unsigned w = ew[e - 1];
                const int out = (int)(w & 0xffffu);
                e = w >> 16;
                const int d = deg[out] - 1; deg[out] = (unsigned short)d;
                if (d == 0) {
                    bool ok = true;
                    const int a0 = (int)(nw[out] >> 16);
                    for (int a = a0; a != out; a = (int)(nw[a] >> 16)) if (deg[a] != 0) { ok = false; break; }
                    if (!ok) continue;
                    if (qh == qt) { cur = out; have = true; }
                    queue[qt++] = (unsigned short)out;
                    for (int a = a0; a != out; a = (int)(nw[a] >> 16)) queue[qt++] = (unsigned short)a;
                }
            }
        }
        if (index != n) g.status = LCD_ERR_TOPO;
        if (lane == 0) sm.bc[6] = g.status;
    }
    __syncthreads();
    g.t_kahn += (unsigned long long)(clock64() - tk0);
    g.status = sm.bc[6];
    if (g.status == LCD_OK) {
        // heaviest successor of every node (first maximum in out-edge order; 1 = the sink when there is no out-edge), in parallel; it
        // replaces the ring pointer in the high half of the node's own word (nobody else reads that word in this loop)
        // (P, D) buffers of the pointer jumping: with the walk's arrays in LDS, buffer 0 takes the place of deg | queue (dead after this loop; thread v alone touches
        // slot v) and buffer 1 that of the node words -- every round is then an LDS pass instead of a round trip to HBM; otherwise the row-plan arrays in HBM
        unsigned short *P0 = INLDS ? deg : (unsigned short *)g.pl_start, *D0 = INLDS ? queue : P0 + n; // (the jump tables of the walk are dead now)
        struct HV { int mw, mid, q; unsigned more; };
        batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { // (the first two out-edges straight-line: nearly every node has no more)
            HV r; r.mw = -1; r.mid = 1; r.q = queue[v];
            const unsigned e0 = nw[v] & 0xffffu;
            const unsigned w0 = ew[e0 ? e0 - 1 : 0]; const int t0 = g.e_w[e0 ? e0 - 1 : 0];
            const unsigned e1 = e0 ? w0 >> 16 : 0;
            const unsigned w1 = ew[e1 ? e1 - 1 : 0]; const int t1 = g.e_w[e1 ? e1 - 1 : 0];
            if (e0) { r.mw = t0; r.mid = (int)(w0 & 0xffffu); }
            if (e1 && t1 > r.mw) { r.mw = t1; r.mid = (int)(w1 & 0xffffu); }
            r.more = e1 ? w1 >> 16 : 0;
            return r;
        }, [&](const int v, HV r) {
            for (unsigned e = r.more; e != 0;) { const unsigned w = ew[e - 1]; const int wt = g.e_w[e - 1]; if (wt > r.mw) { r.mw = wt; r.mid = (int)(w & 0xffffu); } e = w >> 16; }
            g.idx2node[v] = r.q;
            P0[v] = (unsigned short)r.mid; D0[v] = (unsigned short)(v == 1 ? 0 : 1);
        });
        __syncthreads();
        if (INLDS) remain_by_jumping<NT, lcd_lds_u16p>(g, n, lds_u16(deg), lds_u16(nw));
        else remain_by_jumping<NT, unsigned short *>(g, n, (unsigned short *)g.pl_start, (unsigned short *)g.pl_rem);
    }
    __syncthreads();
}


// ---- the re-sort of graphs that do not fit the pool as packed words (above): a COMPACT copy -- 5 B per node + 2 B per edge + a 512 B FIFO ----
// Graphs of noisy reads are twice the size of their reads and re-sorted after nearly every read; at 8 B per node + 4 B per edge an 800-node graph misses the 8 KB
// pool of the single-wavefront class and walked its packed words in HBM: ~1 300 ticks per node, 36 % of that class's time on the ONT shape (round 4 chain profile).
// Here: in-degrees as bytes (bit 7: "a chain of >= 4 links starts here", the jump test), out-edges as CSR (u16 offsets + u16 targets, in out-edge list order -- the
// order of the FIFO), the aligned ring as u16, and the FIFO itself a 256-entry ring: the order goes straight to idx2node / node2idx in HBM (stores do not stall the
// walk).  A FIFO that would hold more than 256 nodes, or an in-degree above 127, hands the graph to the HBM walk (returns false; nothing it wrote is kept).
typedef __attribute__((address_space(3))) uint8_t *lcd_lds_u8p;
template <int NT>
__device__ __forceinline__ bool topo_sort_compact(Ctx &g, Smem &sm, int *lds_pool) {
    const int tid = threadIdx.x;
    const int n = g.n_node, E = g.n_edge;
    constexpr int QR = 256;
    const int qcap = (g.topo_mode & 4) ? 3 : QR; // (LCD_DBG bit 256: a FIFO of three nodes -- every bubble hands the graph to the HBM walk: the test of that hand-over)
    const int n1 = (n + 2) & ~1;
    lcd_lds_u16p ring = lds_u16(lds_pool), ostart = ring + QR, al = ostart + n1, eto = al + n;
    lcd_lds_u8p deg = (lcd_lds_u8p)(eto + E);
    // ---- staging: in-degree, aligned ring, out-degree (into ostart[v + 1]) ----
    int bad = 0;
    struct S1 { int d, a, cnt, more; };
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) {
        S1 r; r.d = g.nin[v]; r.a = g.aligned[v];
        const int e0 = g.out_head[v], c0 = e0 >= 0 ? e0 : 0;
        const int e1 = e0 >= 0 ? g.e_next_out[c0] : -1, c1 = e1 >= 0 ? e1 : 0;
        r.more = e1 >= 0 ? g.e_next_out[c1] : -1;
        r.cnt = (e0 >= 0) + (e1 >= 0);
        return r;
    }, [&](const int v, S1 r) {
        for (int e = r.more; e >= 0; e = g.e_next_out[e]) ++r.cnt;
        if (r.d > 127) bad = 1;
        deg[v] = (uint8_t)r.d; al[v] = (unsigned short)r.a; ostart[v + 1] = (unsigned short)r.cnt;
    });
    if (__syncthreads_or(bad)) return false;
    { // exclusive prefix sums of the out-degrees: a contiguous chunk per thread, one block scan of the chunk sums
        const int per = (n + NT - 1) / NT, lo = imin(n, tid * per), hi = imin(n, lo + per);
        int sum = 0;
        for (int v = lo; v < hi; ++v) sum += ostart[v + 1];
        int tot;
        int run = block_excl_scan<NT>(sum, sm, &tot);
        // (in place: ostart[v + 1] <- offset of v + 1; a thread's chunk is its own, ostart[lo] belongs to the chunk before -- written by that thread as ITS last entry)
        for (int v = lo; v < hi; ++v) { run += ostart[v + 1]; ostart[v + 1] = (unsigned short)run; }
        if (tid == 0) ostart[0] = 0;
        if (tot != E) bad = 1; // (every edge is on exactly one out-list)
    }
    if (__syncthreads_or(bad)) return false;
    struct S2 { int t0, t1, more; };
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) {
        S2 r;
        const int e0 = g.out_head[v], c0 = e0 >= 0 ? e0 : 0;
        r.t0 = e0 >= 0 ? g.e_to[c0] : -1;
        const int e1 = e0 >= 0 ? g.e_next_out[c0] : -1, c1 = e1 >= 0 ? e1 : 0;
        r.t1 = e1 >= 0 ? g.e_to[c1] : -1;
        r.more = e1 >= 0 ? g.e_next_out[c1] : -1;
        return r;
    }, [&](const int v, const S2 r) {
        int k = ostart[v];
        if (r.t0 >= 0) eto[k++] = (unsigned short)r.t0;
        if (r.t1 >= 0) eto[k++] = (unsigned short)r.t1;
        for (int e = r.more; e >= 0; e = g.e_next_out[e]) eto[k++] = (unsigned short)g.e_to[e];
    });
    __syncthreads();
    // ---- chain links and jump tables, as in topo_sort_arrays (tables in the row-plan arrays in HBM) ----
    unsigned short *J1 = (unsigned short *)g.pl_start, *J4 = J1 + n, *J16 = (unsigned short *)g.pl_rem;
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) {
        const int s0 = ostart[v], s1 = ostart[v + 1];
        const int to = eto[s0 < E ? s0 : 0];
        const bool link = s1 - s0 == 1 && deg[to] == 1 && (int)al[to] == to;
        return link ? to : v;
    }, [&](const int v, const int l) { J1[v] = (unsigned short)l; });
    __syncthreads();
    unsigned char *R4 = (unsigned char *)(J16 + n) + n;
    struct XR { int x, r; };
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { XR o; o.x = v; o.r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int y = J1[o.x]; o.r += y != o.x; o.x = y; }
        return o; }, [&](const int v, const XR o) { J4[v] = (unsigned short)o.x; R4[v] = (unsigned char)o.r; });
    __syncthreads();
    batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { XR o; o.x = v; o.r = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o.r += R4[o.x]; o.x = J4[o.x]; }
        return o; }, [&](const int v, const XR o) { J16[v] = (unsigned short)o.x; if (o.r >= 4) deg[v] = (uint8_t)(deg[v] | 0x80); });
    __syncthreads();
    const long long tk0 = clock64();
    if (tid < 64) { // wavefront 0, every lane with the same scalars
        const int lane = tid;
        int qh = 0, qt = 1, index = 0, ovf = 0;
        int cur = 0; bool have = true;
        while (qh < qt) {
            if (!have) cur = ring[qh & (QR - 1)];
            ++qh; have = false;
            while (qh == qt) { // nothing else queued: follow the chain that starts at cur, 64 nodes per step
                if (!(deg[cur] & 0x80)) break;
                int x = cur;
                const int c16 = lane >> 4, c4 = (lane >> 2) & 3, c1 = lane & 3;
                for (int i = 0; i < 3; ++i) if (i < c16) x = J16[x];
                for (int i = 0; i < 3; ++i) if (i < c4) x = J4[x];
                for (int i = 0; i < 3; ++i) if (i < c1) x = J1[x];
                const int prev = __shfl_up(x, 1);
                const unsigned long long adv = __ballot(lane == 0 || x != prev);
                const int L = __popcll(adv);
                if (L <= 1) break;
                if (lane < L - 1) { g.node2idx[x] = index + lane; g.idx2node[index + lane] = x; g.cut[index + lane] = 1; }
                index += L - 1; qh += L - 1; qt = qh;
                cur = __shfl(x, L - 1);
                if (L < 64 || index > n) break;
            }
            g.node2idx[cur] = index; g.idx2node[index < n ? index : 0] = cur; g.cut[index < n ? index : 0] = (uint8_t)(qh == qt); ++index;
            if (cur == 1 || index > n) break;
            const int s0 = ostart[cur], s1 = ostart[cur + 1];
            int out_n = eto[s0 < E ? s0 : 0];
            for (int k = s0; k < s1;) {
                const int out = out_n;
                ++k;
                out_n = eto[k < E ? k : 0]; // (the targets do not change during the walk: the next one is on its way while this one is handled)
                const int dv = deg[out], d = (dv & 0x7f) - 1;
                deg[out] = (uint8_t)((dv & 0x80) | d);
                if (d == 0) {
                    bool ok = true;
                    const int a0 = al[out];
                    int na = 0;
                    for (int a = a0; a != out; a = al[a]) { if (deg[a] & 0x7f) { ok = false; break; } ++na; }
                    if (!ok) continue;
                    if (qt - qh + na + 1 > qcap) { ovf = 1; break; }
                    if (qh == qt) { cur = out; have = true; }
                    ring[qt++ & (QR - 1)] = (unsigned short)out;
                    for (int a = a0; a != out; a = al[a]) ring[qt++ & (QR - 1)] = (unsigned short)a;
                }
            }
            if (ovf) break;
        }
        if (!ovf && index != n) g.status = LCD_ERR_TOPO;
        if (lane == 0) { sm.bc[6] = g.status; sm.bc[5] = ovf; }
    }
    __syncthreads();
    g.t_kahn += (unsigned long long)(clock64() - tk0);
    g.status = sm.bc[6];
    const int ovf = sm.bc[5];
    __syncthreads();
    if (ovf) return false;
    if (g.status == LCD_OK) {
        // heaviest successor of every node from the graph's own arrays (parallel: their latency overlaps), remain by pointer jumping in LDS (the walk's arrays are dead)
        lcd_lds_u16p P0 = lds_u16(lds_pool), D0 = P0 + n;
        struct HV { int mw, mid, more; };
        batched_for<LCD_BF_U, NT>(0, n, [&](const int v) {
            HV r; r.mw = -1; r.mid = 1;
            const int e0 = g.out_head[v], c0 = e0 >= 0 ? e0 : 0;
            const int t0 = g.e_w[c0], to0 = g.e_to[c0];
            const int e1 = e0 >= 0 ? g.e_next_out[c0] : -1, c1 = e1 >= 0 ? e1 : 0;
            const int t1 = g.e_w[c1], to1 = g.e_to[c1];
            if (e0 >= 0) { r.mw = t0; r.mid = to0; }
            if (e1 >= 0 && t1 > r.mw) { r.mw = t1; r.mid = to1; }
            r.more = e1 >= 0 ? g.e_next_out[c1] : -1;
            return r;
        }, [&](const int v, HV r) {
            for (int e = r.more; e >= 0; e = g.e_next_out[e]) { const int wt = g.e_w[e]; if (wt > r.mw) { r.mw = wt; r.mid = g.e_to[e]; } }
            P0[v] = (unsigned short)r.mid; D0[v] = (unsigned short)(v == 1 ? 0 : 1);
        });
        __syncthreads();
        remain_by_jumping<NT, lcd_lds_u16p>(g, n, P0, P0 + 2 * n);
    }
    __syncthreads();
    return true;
}

template <int NT>
__device__ __attribute__((noinline)) void topo_sort_block(Ctx &g, Smem &sm, int *lds_pool, const int want_mm) {
    const int tid = threadIdx.x;
    const int n = g.n_node, E = g.n_edge;
    if (n >= 65535 || E >= 65535) { // ids do not fit 16 bits: plain serial walk on the graph arrays
        if (tid == 0) { topo_sort(g); sm.bc[6] = g.status; }
        __syncthreads();
        g.status = sm.bc[6];
        g.mm_valid = 0; g.cut_valid = 0;
        __syncthreads();
        return;
    }
    // node word = out_head + 1 (low 16 bits) | next node of the aligned ring (high 16); edge word = to (low) | next out-edge + 1 (high):
    // one read per node / edge instead of two on the serial walk's dependency chain
    const bool fits_packed = (size_t)8 * n + (size_t)4 * E + 64 <= (size_t)g.pool_words * 4;
    const size_t compact_bytes = (size_t)512 + 2 * (size_t)((n + 2) & ~1) + 2 * (size_t)n + 2 * (size_t)E + (size_t)n + 64;
    const bool fits_compact = !(g.topo_mode & 1) && (compact_bytes > (size_t)8 * n + 64 ? compact_bytes : (size_t)8 * n + 64) <= (size_t)g.pool_words * 4;
    bool done = false;
    if (fits_compact && (!fits_packed || (g.topo_mode & 2))) done = topo_sort_compact<NT>(g, sm, lds_pool);
    if (done) { /* order, node2idx, remain are there */ }
    else if (fits_packed) {
        unsigned short *deg = (unsigned short *)lds_pool, *queue = deg + n;
        unsigned *nw = (unsigned *)(queue + n + (n & 1)), *ew = nw + n; // (4-byte aligned: the pool is, and 2n + (n & 1) halfwords are even)
        topo_sort_arrays<NT, true>(g, sm, deg, queue, nw, ew);
    } else
        topo_sort_arrays<NT, false>(g, sm, (unsigned short *)g.deg, (unsigned short *)g.queue, (unsigned *)g.pl_bonus, (unsigned *)g.pl_pidx);
    // (every row's smallest predecessor index / largest successor index, for the sub-graph sweeps of partial-cover reads: compute_mm, made when such a read comes)
    g.mm_valid = 0; g.cut_valid = g.status == LCD_OK;
    __syncthreads();
}

// every row's smallest predecessor index / largest successor index (g.deg / g.queue, free between two re-sorts), for the sub-graph sweeps of the partial-cover reads
// until the order changes again (subgraph_nodes_wave0)
template <int NT>
__device__ __attribute__((noinline)) void compute_mm(Ctx &g) {
    const int n = g.n_node;
    struct MM { int mn, mx, ein, eout; };
    batched_for<LCD_BF_U, NT>(0, n, [&](const int idx) { // (the first two edges of either list straight-line: nearly every node has no more)
        MM r; r.mn = 1 << 30; r.mx = -1;
        const int v = g.idx2node[idx];
        const int i0 = g.in_head[v], o0 = g.out_head[v], ci0 = i0 >= 0 ? i0 : 0, co0 = o0 >= 0 ? o0 : 0;
        const int fi0 = g.e_from[ci0], i1 = i0 >= 0 ? g.e_next_in[ci0] : -1, to0 = g.e_to[co0], o1 = o0 >= 0 ? g.e_next_out[co0] : -1;
        const int ci1 = i1 >= 0 ? i1 : 0, co1 = o1 >= 0 ? o1 : 0;
        const int fi1 = g.e_from[ci1], to1 = g.e_to[co1];
        const int a0 = g.node2idx[fi0], a1 = g.node2idx[fi1], b0 = g.node2idx[to0], b1 = g.node2idx[to1];
        if (i0 >= 0) r.mn = a0;
        if (i1 >= 0) r.mn = imin(r.mn, a1);
        if (o0 >= 0) r.mx = b0;
        if (o1 >= 0) r.mx = imax(r.mx, b1);
        r.ein = i1 >= 0 ? g.e_next_in[ci1] : -1; r.eout = o1 >= 0 ? g.e_next_out[co1] : -1;
        return r;
    }, [&](const int idx, MM r) {
        for (int e = r.ein; e >= 0; e = g.e_next_in[e]) r.mn = imin(r.mn, g.node2idx[g.e_from[e]]);
        for (int e = r.eout; e >= 0; e = g.e_next_out[e]) r.mx = imax(r.mx, g.node2idx[g.e_to[e]]);
        g.deg[idx] = r.mn; g.queue[idx] = r.mx;
    });
    g.mm_valid = 1;
    __syncthreads();
}

// ---- the re-sort of a graph that changed in a few places: the Kahn walk repeated only where it can differ (round 6) ----
// 45 % of the clean reads add one or two bubbles (a substituted base, a homopolymer one longer or shorter) to a graph of several hundred nodes, and the full re-sort
// above -- staging, jump tables, the serial walk over all nodes, heaviest successors, log2(n) pointer-jumping rounds -- cost as much as half the read's DP.  The
// oracle's order (oracle/poa.c topo_sort: Kahn with a FIFO, aligned groups released together) is reproduced EXACTLY from the old one:
//   * `cut[i]` (recorded by every walk): the FIFO was empty right after the node of index i was popped.  At such a point the walk's state is (popped = the first i + 1
//     nodes, nothing queued), so it can be restarted there with "a node is ready when all its in-edges come from popped nodes" in place of in-degree counters.
//   * the read's path is monotone in the old order, and what it added starts at old nodes (the source of a new edge; the row before an aligned group that gained a
//     member): before the first of them nothing moves.  The walk restarts at the largest cut at or before it and goes on until it pops a node c with nothing queued,
//     c a cut of the OLD order as well, and the old nodes emitted so far exactly the old range [q, idx(c)]: from there on both walks are in the same state (a popped
//     source of an in-edge delays nobody), so the old order stands, shifted by the number of new nodes emitted -- up to the largest cut before the next thing added.
//   * anything unexpected (no cut in reach, too many nodes walked, a new node never emitted, a queue that runs dry) returns false: the caller takes the full re-sort.
// `remain` of a new node is that of its only successor plus one; an old node's heaviest out-edge cannot become a new edge (weight 1, last in its list), and
// add_alignment_block says when a weight increment moved one (upd_moved: the caller then re-makes `remain` as after a weights-only read).
// The walk itself runs on REGISTERS: the rows of a 64-row window of the old order (lane = row: node, cut flag, up to three out-edge targets, up to three in-edge
// sources, next node of the aligned ring) and the new nodes (lane = new node: its one source, its one target, its ring) are fetched by all lanes at once -- six
// dependent trips for the whole window instead of five per node walked -- with every reference already turned into "old index" or "new node t"; the serial part then
// reads them by v_readlane and keeps the popped sets as two 64-bit masks.  A node with a fourth edge, or a piece that outgrows its window twice, hands the graph to the
// full re-sort; the far end of a long deletion edge (an old node beyond the window whose readiness is asked) is looked up in HBM.
constexpr int INC_Q = 32, INC_NEW = 128, INC_NEWREF = 1 << 20;
typedef __attribute__((address_space(3))) int inc_lds_i32; // (the lists live in the workgroup's LDS pool, addressed by byte offset: ds_read / ds_write, not flat operations)
__device__ __forceinline__ int inc_ld(const unsigned o) { return *(const inc_lds_i32 *)(uintptr_t)o; }
__device__ __forceinline__ void inc_st(const unsigned o, const int v) { *(inc_lds_i32 *)(uintptr_t)o = v; }
template <int NT>
__device__ __attribute__((noinline)) bool topo_sort_incremental(Ctx &g, Smem &sm, const unsigned pool_off, const int beg_node, const int end_node, const int n_cig) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = g.n_node, n_new = g.upd_new, n_old = n - n_new;
    // the pool's words (free between two reads) go to the lists below: 1/8 to the places to revisit, 3/8 to the pieces' table (6 words each), 1/2 to the nodes emitted
    const int inc_free = g.pool_words - INC_Q - 8;
    const int INC_ML = imin(1024, inc_free / 8), INC_SEG = imin(512, inc_free * 3 / 8 / 6), INC_EL = imin(4096, inc_free - INC_ML - 6 * INC_SEG);
    if (!g.cut_valid || g.inc_off || n >= 65535 || n_new > INC_NEW || n_old < 3 || INC_ML < 32 || INC_SEG < 16 || INC_EL < 128) {
#ifdef LCD_X_INCSTAT
        g.inc_stat[0] += 1;
#endif
        return false;
    }
    const unsigned ml_o = pool_off;                      // where the walk has to be repeated: old indices, in path order (ascending)
    const unsigned el_o = ml_o + 4u * (unsigned)INC_ML;  // the nodes the walks emitted: id | (FIFO empty after the pop) << 16
    const unsigned seg_o = el_o + 4u * (unsigned)INC_EL; // per repeated piece: q, ic, first entry of el, entries, new nodes emitted before / after
    const unsigned q_o = seg_o + 24u * (unsigned)INC_SEG;
    LCD_PT0();
    if (wave == 0) {
        int fail = 0;
        int n_ml = 0, n_nj = 0;
        // ---- what the read added, in path order: edge j runs from path[j - 1] (beg_node) to path[j] (end_node) ----
        for (int base4 = 0; base4 <= n_cig && !fail; base4 += 256) {
            int eid4[4], fl4[4], flp4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { // (the loads of four 64-entry pieces in flight; an entry of the path array costs one trip)
                const int j = base4 + u * 64 + lane;
                eid4[u] = -1; fl4[u] = -2; flp4[u] = -2;
                if (j <= n_cig) { eid4[u] = g.aa_eid[j]; if (j < n_cig) fl4[u] = g.aa_flag[j]; if (j > 0) flp4[u] = g.aa_flag[j - 1]; }
            }
            int v04[4], v14[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { // (and the old indices the entries refer to, for all four pieces before the first list is written)
                const int j = base4 + u * 64 + lane;
                v04[u] = -1; v14[u] = -1;
                if (eid4[u] >= 0 && flp4[u] == -2) { const int from = j == 0 ? beg_node : g.aa_node[j - 1]; v04[u] = g.node2idx[from]; }
                if (fl4[u] >= 0) v14[u] = g.node2idx[fl4[u]] - 1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int base = base4 + u * 64;
                if (base > n_cig || fail) break; // (uniform)
                const int eid = eid4[u], fl = fl4[u], flp = flp4[u];
                if (!__any(eid >= 0 || fl != -2)) continue; // nothing added on these 64 entries (uniform)
                const int has0 = (eid >= 0 && flp == -2) ? 1 : 0, has1 = fl >= 0 ? 1 : 0;
                const int cnt = has0 + has1, incl = scan_add(cnt), tot = lane63(incl);
                if (n_ml + tot > INC_ML) { fail = 1; break; }
                const int off = n_ml + incl - cnt;
                if (has0) inc_st(ml_o + 4u * (unsigned)off, v04[u]);
                if (has1) inc_st(ml_o + 4u * (unsigned)(off + has0), v14[u]);
                n_ml += tot;
                n_nj += (int)__popcll(__ballot(fl != -2));
            }
        }
        if (!fail && n_nj != n_new) fail = 3;
        LCD_PT(15);
        auto LD = [&](const int *p) { return usgpr(glb_ld(p)); };
        auto ref_of = [&](const int x) { return x < n_old ? g.node2idx[x] : INC_NEWREF + (x - n_old); }; // (per lane: a gather)
        auto largest_cut = [&](const int p) { // largest i <= p with cut[i] (-1: none within 256 indices)
            for (int base = p, it = 0; it < 4 && base >= 0; ++it, base -= 64) {
                const int i = base - lane;
                const int c = i >= 0 ? glb_ld_u8(g.cut + i) : 0;
                const unsigned long long m = __ballot(c != 0);
                if (m) return base - (int)__builtin_ctzll(m);
            }
            return -1;
        };
        // ---- the new nodes' records: lane t (and t + 64) = node n_old + t: one in-edge, one out-edge (the read's path enters and leaves it once), and `remain` of the
        //      old node its out-edge leads to ----
        int Nf[2], Nt[2], Nal[2], Nrm[2];
        {
            int nbad = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Nf[h] = -1; Nt[h] = -1; Nal[h] = -1; Nrm[h] = LCD_NEG; // (LCD_NEG: the out-edge leads to the next new node; remain of the sink is -1)
                const int t = h * 64 + lane;
                if (t < n_new) {
                    const int x = n_old + t;
                    const int ih = g.in_head[x], oh = g.out_head[x], al = g.aligned[x];
                    if (ih < 0 || oh < 0) nbad = 1;
                    else {
                        const int f = g.e_from[ih], to = g.e_to[oh];
                        if (g.e_next_in[ih] >= 0 || g.e_next_out[oh] >= 0) nbad = 1;
                        Nf[h] = ref_of(f); Nt[h] = ref_of(to);
                        if (to < n_old) Nrm[h] = g.remain[to]; else if (to != x + 1) nbad = 1; // (the next new node of the path has the next id)
                    }
                    if (al != x) Nal[h] = ref_of(al);
                }
            }
            if (__any(nbad) && !fail) fail = 3;
        }
        // ---- the window: lane l = old row B + l ----
        int B = 0, Wv = -1, Wcf = 0, Wno = 0, Wt0 = -1, Wt1 = -1, Wt2 = -1, Wox = -1, Wni = 0, Wf0 = -1, Wf1 = -1, Wf2 = -1, Wix = -1, Wal = -1;
        auto gather = [&](const int base) {
#ifdef LCD_X_PHASESTAT
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const long long tg0_ = clock64();
#endif
            B = base;
            const int r = base + lane; const bool val = r < n_old;
            int v = -1, cf = 0, oh = -1, ih = -1, al = -1;
            if (val) { v = g.idx2node[r]; cf = glb_ld_u8(g.cut + r); }
            if (val) { oh = g.out_head[v]; ih = g.in_head[v]; al = g.aligned[v]; }
            int t0 = -1, t1 = -1, t2 = -1, o1 = -1, o2 = -1, o3 = -1, f0 = -1, f1 = -1, f2 = -1, i1 = -1, i2 = -1, i3 = -1;
            if (oh >= 0) { t0 = g.e_to[oh]; o1 = g.e_next_out[oh]; }
            if (ih >= 0) { f0 = g.e_from[ih]; i1 = g.e_next_in[ih]; }
            if (o1 >= 0) { t1 = g.e_to[o1]; o2 = g.e_next_out[o1]; }
            if (i1 >= 0) { f1 = g.e_from[i1]; i2 = g.e_next_in[i1]; }
            if (o2 >= 0) { t2 = g.e_to[o2]; o3 = g.e_next_out[o2]; }
            if (i2 >= 0) { f2 = g.e_from[i2]; i3 = g.e_next_in[i2]; }
            Wv = v; Wcf = cf;
            Wno = (oh >= 0) + (o1 >= 0) + (o2 >= 0) + (o3 >= 0); // (4: more than the record holds -- the fourth edge's id goes along, the list is followed in HBM from there)
            Wni = (ih >= 0) + (i1 >= 0) + (i2 >= 0) + (i3 >= 0);
            Wox = o3; Wix = i3;
            Wt0 = t0 >= 0 ? ref_of(t0) : -1; Wt1 = t1 >= 0 ? ref_of(t1) : -1; Wt2 = t2 >= 0 ? ref_of(t2) : -1;
            Wf0 = f0 >= 0 ? ref_of(f0) : -1; Wf1 = f1 >= 0 ? ref_of(f1) : -1; Wf2 = f2 >= 0 ? ref_of(f2) : -1;
            Wal = (val && al != v) ? ref_of(al) : -1;
#ifdef LCD_X_PHASESTAT
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); g.pstat[23] += (unsigned long long)(clock64() - tg0_);
#endif
        };
        // a record is read field by field, as the walk needs it (a typical node: one out-edge, one or two in-edges, no ring -- five v_readlane's, not thirty)
        auto in_window = [&](const int ref) { return ref >= B && ref < B + 64 && ref < n_old; };
        auto ring_next = [&](const int ref, bool &ok) { // next node of ref's aligned ring (-1: alone); ok = false: an old row outside the window
            ok = true;
            if (ref >= INC_NEWREF) { const int t = ref - INC_NEWREF; return t < 64 ? LCD_RL(Nal[0], t) : LCD_RL(Nal[1], t - 64); }
            if (!in_window(ref)) { ok = false; return -1; }
            return LCD_RL(Wal, ref - B);
        };
        unsigned long long Rdone = 0, Nd0 = 0, Nd1 = 0; // popped: rows of the window (every row before it is), new nodes 0..63 / 64..127
        auto done = [&](const int ref) {
            if (ref >= INC_NEWREF) { const int t = ref - INC_NEWREF; return (((t < 64 ? Nd0 : Nd1) >> (t & 63)) & 1ull) != 0; }
            if (ref < B) return true;
            if (ref >= B + 64) return false;
            return ((Rdone >> (ref - B)) & 1ull) != 0;
        };
        auto upto = [&](const int l) { return l >= 63 ? ~0ull : ((2ull << l) - 1ull); }; // lanes 0..l
        int k = 0, D = 0, n_el = 0, n_seg = 0, ic_prev = -1;
        bool have_window = false;
        while (k < n_ml && !fail) {
            const int p = usgpr(inc_ld(ml_o + 4u * (unsigned)k));
            int q;
            if (have_window && p >= B && p < B + 64) { // (the cut flags of the window's rows are in the lanes)
                const unsigned long long cm = __ballot(Wcf != 0) & upto(p - B);
                q = cm ? B + 63 - (int)__builtin_clzll(cm) : largest_cut(p);
            } else q = largest_cut(p);
            if (q <= ic_prev || n_seg >= INC_SEG || n_el >= INC_EL) { fail = q <= ic_prev ? 4 : 5; break; }
            if (!have_window || q < B || q + 12 > B + 64) { gather(q); have_window = true; Rdone = 0; }
            Rdone |= upto(q - B); // (every row up to q has been popped)
            const int k0 = k, D0 = D, el0 = n_el; const unsigned long long Ns0 = Nd0, Ns1 = Nd1;
            int ic = -1;
            for (int attempt = 0; attempt < 2 && !fail; ++attempt) {
                bool exceed = false;
                int cur = q, m_old = 1, maxe = q, qh = 0, qt = 0;
                inc_st(el_o + 4u * (unsigned)n_el, LCD_RL(Wv, q - B) | (1 << 16)); ++n_el;
                for (;;) {
                    // the out-edges of the node at hand: a new node has one; a window row up to three in its record, the rest followed in HBM from the fourth edge's id
                    int c_no, c_t0 = -1, c_t1 = -1, c_t2 = -1, emore = -1;
                    if (cur >= INC_NEWREF) { const int t = cur - INC_NEWREF; c_no = 1; c_t0 = t < 64 ? LCD_RL(Nt[0], t) : LCD_RL(Nt[1], t - 64); }
                    else {
                        if (!in_window(cur)) { exceed = true; break; }
                        const int l = cur - B;
                        c_no = LCD_RL(Wno, l);
                        if (c_no > 0) c_t0 = LCD_RL(Wt0, l);
                        if (c_no > 1) c_t1 = LCD_RL(Wt1, l);
                        if (c_no > 2) c_t2 = LCD_RL(Wt2, l);
                        if (c_no > 3) emore = LCD_RL(Wox, l);
                    }
                    // the oracle's counters: a node's in-degree reaches 0 when its LAST in-edge is taken, and the out-edges of the node at hand are taken one by one --
                    // an in-edge from a popped node has been taken, one from the node at hand only if its target is among those already visited (`seen`: window rows and
                    // new nodes as masks, the rare target beyond the window by value)
                    unsigned long long seenR = 0, seenN0 = 0, seenN1 = 0; int so0 = -2, so1 = -2;
                    auto ref1 = [&](const int x) { return x < n_old ? LD(g.node2idx + x) : INC_NEWREF + (x - n_old); }; // (uniform)
                    auto is_seen = [&](const int w) {
                        if (w >= INC_NEWREF) { const int t = w - INC_NEWREF; return (((t < 64 ? seenN0 : seenN1) >> (t & 63)) & 1ull) != 0; }
                        if (w >= B && w < B + 64) return ((seenR >> (w - B)) & 1ull) != 0;
                        return w == so0 || w == so1;
                    };
                    auto taken = [&](const int fr, const int w) { return fr == cur ? is_seen(w) : done(fr); }; // has the in-edge fr -> w been taken?
                    auto ready = [&](const int w) {
                        int e = -1;
                        if (w >= INC_NEWREF) { const int t = w - INC_NEWREF; return taken(t < 64 ? LCD_RL(Nf[0], t) : LCD_RL(Nf[1], t - 64), w); }
                        if (in_window(w)) {
                            const int l = w - B, ni = LCD_RL(Wni, l);
                            if (ni > 0 && !taken(LCD_RL(Wf0, l), w)) return false;
                            if (ni > 1 && !taken(LCD_RL(Wf1, l), w)) return false;
                            if (ni > 2 && !taken(LCD_RL(Wf2, l), w)) return false;
                            if (ni <= 3) return true;
                            e = LCD_RL(Wix, l); // (a fourth, fifth ... in-edge: followed in HBM)
                        } else { // an old node beyond the window (the far end of a long deletion edge): its in-edges from HBM
                            if (w < B || w >= n_old) { fail = 7; return false; }
                            e = LD(g.in_head + LD(g.idx2node + w));
                        }
                        for (; e >= 0;) {
                            const int f = LD(g.e_from + e), en = LD(g.e_next_in + e);
                            if (!taken(ref1(f), w)) return false;
                            e = en;
                        }
                        return true;
                    };
                    for (int kk = 0; !fail && !exceed; ++kk) {
                        int w;
                        if (kk < 3) { if (kk >= c_no) break; w = kk == 0 ? c_t0 : kk == 1 ? c_t1 : c_t2; }
                        else { if (emore < 0) break; w = ref1(LD(g.e_to + emore)); emore = LD(g.e_next_out + emore); }
                        if (w >= INC_NEWREF) { const int t = w - INC_NEWREF; if (t < 64) seenN0 |= 1ull << t; else seenN1 |= 1ull << (t - 64); }
                        else if (w >= B && w < B + 64) seenR |= 1ull << (w - B);
                        else if (so0 == -2) so0 = w; else if (so1 == -2) so1 = w; else { fail = 6; break; }
                        if (!ready(w)) continue;
                        // the aligned group goes out together, the node whose in-degree just ran out first, the others in ring order -- if every member is ready
                        bool ok = true, inw; int na = 0;
                        const int al0 = ring_next(w, inw);
                        if (inw) {
                            for (int a = al0; a >= 0 && a != w;) {
                                if (++na > 8) { fail = 6; ok = false; break; }
                                if (!ready(a)) { ok = false; break; }
                                bool ina; const int an = ring_next(a, ina);
                                if (!ina) { exceed = true; ok = false; break; }
                                a = an;
                            }
                        } else { // (beyond the window: alone in its ring, or the piece needs a window further on)
                            const int wn = LD(g.idx2node + w);
                            if (LD(g.aligned + wn) != wn) { exceed = true; ok = false; }
                        }
                        if (!ok || fail) continue;
                        if (qt - qh + na + 1 > INC_Q) { fail = 6; break; }
                        inc_st(q_o + 4u * (unsigned)(qt++ & (INC_Q - 1)), w);
                        if (na > 0) for (int a = al0; a >= 0 && a != w;) { inc_st(q_o + 4u * (unsigned)(qt++ & (INC_Q - 1)), a); bool ina; a = ring_next(a, ina); }
                    }
                    if (fail || exceed) break;
                    if (qh == qt || n_el >= INC_EL) { fail = qh == qt ? 7 : 8; break; } // (a FIFO that runs dry before the walk is back on the old order: not a state the argument above covers)
                    cur = usgpr(inc_ld(q_o + 4u * (unsigned)(qh++ & (INC_Q - 1))));
                    const bool empty = qh == qt;
                    int oi = -1;
                    if (cur >= INC_NEWREF) {
                        const int t = cur - INC_NEWREF;
                        if (t < 64) Nd0 |= 1ull << t; else Nd1 |= 1ull << (t - 64);
                        ++D; inc_st(el_o + 4u * (unsigned)n_el, (n_old + t) | ((int)empty << 16)); ++n_el;
                    } else {
                        if (cur < B || cur >= B + 64) { exceed = true; break; }
                        Rdone |= 1ull << (cur - B);
                        oi = cur; ++m_old; maxe = imax(maxe, oi);
                        inc_st(el_o + 4u * (unsigned)n_el, LCD_RL(Wv, cur - B) | ((int)empty << 16)); ++n_el;
                    }
                    if (empty && oi >= 0 && oi == maxe && maxe == q + m_old - 1 && LCD_RL(Wcf, oi - B) != 0) {
                        while (k < n_ml && usgpr(inc_ld(ml_o + 4u * (unsigned)k)) < oi) ++k;
                        if (k < n_ml) { // the next thing added: does it start right here?
                            const int p2 = usgpr(inc_ld(ml_o + 4u * (unsigned)k));
                            int q2;
                            if (p2 < B + 64) { const unsigned long long cm = __ballot(Wcf != 0) & upto(p2 - B); q2 = cm ? B + 63 - (int)__builtin_clzll(cm) : -1; }
                            else q2 = largest_cut(p2);
                            if (q2 == oi) continue; // the walk goes on
                        }
                        ic = oi;
                        break;
                    }
                }
                if (!exceed || fail) break;
                // the piece left the window: once more with the window at its own start (if it is not there already)
                if (q == B || attempt == 1) { fail = 8; break; }
                gather(q); Rdone = 1ull;
                k = k0; D = D0; n_el = el0; Nd0 = Ns0; Nd1 = Ns1;
            }
            if (fail) break;
            if (ic < 0) { fail = 8; break; }
            const unsigned so = seg_o + 24u * (unsigned)n_seg;
            inc_st(so, q); inc_st(so + 4, ic); inc_st(so + 8, el0); inc_st(so + 12, n_el - el0); inc_st(so + 16, D0); inc_st(so + 20, D);
            ++n_seg; ic_prev = ic;
        }
        if (!fail && D != n_new) fail = 9;
        LCD_PT(16);
        if (!fail) {
            // ---- apply.  One wavefront, no barrier.  First every old row behind a piece moves up by the new nodes emitted before it (top-down, 256 rows at a time: a
            //      chunk's stores land at or above its own rows, which have all been loaded, never on the rows of a later chunk), then the pieces' rows are written ----
            auto piece_of = [&](const int key, const unsigned field) { // last piece whose `field` (0: q, 8: first entry of el) is <= key; -1: none  (per lane)
                int lo = 0, hi = n_seg; // invariant: pieces [0, lo) qualify, [hi, n_seg) do not
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (inc_ld(seg_o + 24u * (unsigned)mid + field) <= key) lo = mid + 1; else hi = mid; }
                return lo - 1;
            };
            if (n_new > 0) {
                int s0 = 0;
                while (s0 < n_seg && usgpr(inc_ld(seg_o + 24u * (unsigned)s0 + 20)) == 0) ++s0; // the first piece with a new node: nothing before its end moves
                const int row_lo = s0 < n_seg ? usgpr(inc_ld(seg_o + 24u * (unsigned)s0 + 4)) + 1 : n_old;
                for (int top = n_old; top > row_lo; top -= 256) {
                    int v[4], c[4], d[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = top - 1 - (u * 64 + lane);
                        v[u] = 0; c[u] = 0; d[u] = 0;
                        if (i >= row_lo) {
                            const int sp = piece_of(i, 0);
                            if (sp >= 0 && i > inc_ld(seg_o + 24u * (unsigned)sp + 4)) d[u] = inc_ld(seg_o + 24u * (unsigned)sp + 20); // (behind piece sp; rows inside a piece are written below)
                            if (d[u] > 0) { v[u] = g.idx2node[i]; c[u] = g.cut[i]; }
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int i = top - 1 - (u * 64 + lane); if (d[u] > 0) { g.idx2node[i + d[u]] = v[u]; g.cut[i + d[u]] = (uint8_t)c[u]; g.node2idx[v[u]] = i + d[u]; } }
                }
            }
            for (int e = lane; e < n_el; e += 64) {
                const int sp = piece_of(e, 8);
                const unsigned so = seg_o + 24u * (unsigned)(sp < 0 ? 0 : sp);
                const int w = inc_ld(el_o + 4u * (unsigned)e), v = w & 0xffff, pos = inc_ld(so) + inc_ld(so + 16) + (e - inc_ld(so + 8));
                g.idx2node[pos] = v; g.node2idx[v] = pos; g.cut[pos] = (uint8_t)(w >> 16);
            }
            LCD_PT(17);
            // ---- remain of the new nodes: that of the node its one out-edge leads to, plus one; runs of new nodes (an inserted stretch) from their last node backwards ----
            if (n_new > 0) {
                int r0 = Nrm[0] > LCD_NEG ? Nrm[0] + 1 : LCD_NEG, r1 = Nrm[1] > LCD_NEG ? Nrm[1] + 1 : LCD_NEG; // (LCD_NEG: the out-edge leads to the next new node)
                if (lane >= n_new) r0 = 0;
                if (64 + lane >= n_new) r1 = 0;
                for (int it = 0; it < INC_NEW && __any(r0 == LCD_NEG || r1 == LCD_NEG); ++it) {
                    const int n0 = __shfl_down(r0, 1), n1 = __shfl_down(r1, 1), b0 = __builtin_amdgcn_readfirstlane(r1);
                    const int nx0 = lane == 63 ? b0 : n0;
                    if (r0 == LCD_NEG && nx0 != LCD_NEG) r0 = nx0 + 1;
                    if (r1 == LCD_NEG && lane < 63 && n1 != LCD_NEG) r1 = n1 + 1;
                }
                if (lane < n_new) g.remain[n_old + lane] = r0;
                if (64 + lane < n_new) g.remain[n_old + 64 + lane] = r1;
            }
            LCD_PT(18);
        }
        if (lane == 0) { sm.bc[5] = fail; sm.bc[4] = n_seg; sm.bc[2] = n_el; }
    }
    __syncthreads();
    const int fail = sm.bc[5];
#ifdef LCD_X_INCSTAT
    if (fail) g.inc_stat[fail < 10 ? fail : 0] += 1; else { g.inc_stat[10] += 1; g.inc_stat[11] += (unsigned)sm.bc[2]; g.inc_stat[12] += (unsigned)sm.bc[4]; }
#endif
    __syncthreads();
    if (fail) return false;
    g.mm_valid = 0;
    return true;
}

// The read changed edge weights only (no new node, no new edge): the topological order stands; `remain` follows the heaviest out-edge
// and may not.  Same LDS sweep as above without the Kahn walk: heaviest successors in parallel, one serial pass over the order.
template <int NT>
__device__ __attribute__((noinline)) void topo_remain_block(Ctx &g, Smem &sm, int *lds_pool) {
    const int tid = threadIdx.x;
    const int n = g.n_node;
    if (n >= 65535) { // ids do not fit 16 bits: serial (same arithmetic)
        if (tid == 0) {
            g.remain[1] = -1;
            for (int i = n - 2; i >= 0; --i) {
                const int v = g.idx2node[i]; int mw = -1, mid = 1;
                for (int e = g.out_head[v]; e >= 0; e = g.e_next_out[e]) if (g.e_w[e] > mw) { mw = g.e_w[e]; mid = g.e_to[e]; }
                g.remain[v] = g.remain[mid] + 1;
            }
        }
        __syncthreads();
        return;
    }
    const bool inlds = (size_t)8 * n + 64 <= (size_t)g.pool_words * 4; // both (P, D) buffers in the workgroup's LDS pool (free between two reads)
    if (inlds) {
        const lcd_lds_u16p P0 = lds_u16(lds_pool), D0 = P0 + n;
        struct HV { int mw, mid, more; };
        batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { // (the first two out-edges straight-line: nearly every node has no more)
            HV r; r.mw = -1; r.mid = 1;
            const int e0 = g.out_head[v], c0 = e0 >= 0 ? e0 : 0;
            const int t0 = g.e_w[c0], to0 = g.e_to[c0], e1 = e0 >= 0 ? g.e_next_out[c0] : -1, c1 = e1 >= 0 ? e1 : 0;
            const int t1 = g.e_w[c1], to1 = g.e_to[c1];
            if (e0 >= 0) { r.mw = t0; r.mid = to0; }
            if (e1 >= 0 && t1 > r.mw) { r.mw = t1; r.mid = to1; }
            r.more = e1 >= 0 ? g.e_next_out[c1] : -1;
            return r;
        }, [&](const int v, HV r) {
            for (int e = r.more; e >= 0; e = g.e_next_out[e]) { const int wt = g.e_w[e]; if (wt > r.mw) { r.mw = wt; r.mid = g.e_to[e]; } }
            P0[v] = (unsigned short)r.mid; D0[v] = (unsigned short)(v == 1 ? 0 : 1);
        });
        __syncthreads();
        remain_by_jumping<NT, lcd_lds_u16p>(g, n, P0, P0 + 2 * n);
    } else {
        unsigned short *P0 = (unsigned short *)g.pl_start, *D0 = P0 + n;
        struct HV { int mw, mid, more; };
        batched_for<LCD_BF_U, NT>(0, n, [&](const int v) { // (the first two out-edges straight-line: nearly every node has no more)
            HV r; r.mw = -1; r.mid = 1;
            const int e0 = g.out_head[v], c0 = e0 >= 0 ? e0 : 0;
            const int t0 = g.e_w[c0], to0 = g.e_to[c0], e1 = e0 >= 0 ? g.e_next_out[c0] : -1, c1 = e1 >= 0 ? e1 : 0;
            const int t1 = g.e_w[c1], to1 = g.e_to[c1];
            if (e0 >= 0) { r.mw = t0; r.mid = to0; }
            if (e1 >= 0 && t1 > r.mw) { r.mw = t1; r.mid = to1; }
            r.more = e1 >= 0 ? g.e_next_out[c1] : -1;
            return r;
        }, [&](const int v, HV r) {
            for (int e = r.more; e >= 0; e = g.e_next_out[e]) { const int wt = g.e_w[e]; if (wt > r.mw) { r.mw = wt; r.mid = g.e_to[e]; } }
            P0[v] = (unsigned short)r.mid; D0[v] = (unsigned short)(v == 1 ? 0 : 1);
        });
        __syncthreads();
        remain_by_jumping<NT, unsigned short *>(g, n, (unsigned short *)g.pl_start, (unsigned short *)g.pl_rem);
    }
    __syncthreads();
}

// sub-graph boundaries (oracle/poa.c subgraph_nodes): min/max sweeps on wavefront 0, result broadcast through LDS
__device__ void subgraph_nodes_wave0(Ctx &g, int lane, int inc_beg, int inc_end, int *exc_beg, int *exc_end) {
    const int bi = g.node2idx[inc_beg], ei = g.node2idx[inc_end];
    int b = bi, e = ei, up, down;
    if (g.mm_valid) {
        // the sweeps over the per-row extremes the last re-sort left behind (reads that only add weight change neither the order nor the edges): one coalesced load
        // per row, four of them in flight, instead of a walk index -> node -> in-edges -> their sources -> those rows' indices per row and per read
        const int *minp = g.deg, *maxs = g.queue;
        auto sweep_min = [&](const int lo, const int hi, const int init) { // min over [lo, hi]
            int m = init;
            for (int i = lo + lane; i <= hi; i += 256) {
                const int a0 = minp[i], a1 = minp[imin(i + 64, hi)], a2 = minp[imin(i + 128, hi)], a3 = minp[imin(i + 192, hi)];
                m = imin(imin(m, a0), imin(imin(a1, a2), a3));
            }
            return wave_min(m);
        };
        auto sweep_max = [&](const int lo, const int hi, const int init) {
            int m = init;
            for (int i = lo + lane; i <= hi; i += 256) {
                const int a0 = maxs[i], a1 = maxs[imin(i + 64, hi)], a2 = maxs[imin(i + 128, hi)], a3 = maxs[imin(i + 192, hi)];
                m = imax(imax(m, a0), imax(imax(a1, a2), a3));
            }
            return wave_max(m);
        };
        for (;;) {
            const int mn = sweep_min(b, e, b);
            // (a row in (mn, b] with a predecessor before mn <=> the minimum over those rows is below mn)
            if (mn + 1 > b || sweep_min(mn + 1, b, mn) >= mn) { up = mn; break; }
            e = b; b = mn;
        }
        b = bi; e = ei;
        for (;;) {
            const int mx = sweep_max(b, e, e);
            if (e > mx - 1 || sweep_max(e, mx - 1, mx) <= mx) { down = mx; break; }
            b = e; e = mx;
        }
        *exc_beg = g.idx2node[up]; *exc_end = g.idx2node[down];
        return;
    }
    for (;;) {
        int mn = b;
        for (int i = b + lane; i <= e; i += 64)
            for (int ed = g.in_head[g.idx2node[i]]; ed >= 0; ed = g.e_next_in[ed]) mn = imin(mn, g.node2idx[g.e_from[ed]]);
        mn = wave_min(mn);
        int bad = 0;
        for (int i = mn + 1 + lane; i <= b; i += 64)
            for (int ed = g.in_head[g.idx2node[i]]; ed >= 0; ed = g.e_next_in[ed]) if (g.node2idx[g.e_from[ed]] < mn) bad = 1;
        if (!__any(bad)) { up = mn; break; }
        e = b; b = mn;
    }
    b = bi; e = ei;
    for (;;) {
        int mx = e;
        for (int i = b + lane; i <= e; i += 64)
            for (int ed = g.out_head[g.idx2node[i]]; ed >= 0; ed = g.e_next_out[ed]) mx = imax(mx, g.node2idx[g.e_to[ed]]);
        mx = wave_max(mx);
        int bad = 0;
        for (int i = e + lane; i < mx; i += 64)
            for (int ed = g.out_head[g.idx2node[i]]; ed >= 0; ed = g.e_next_out[ed]) if (g.node2idx[g.e_to[ed]] > mx) bad = 1;
        if (!__any(bad)) { down = mx; break; }
        b = e; e = mx;
    }
    *exc_beg = g.idx2node[up]; *exc_end = g.idx2node[down];
}



// six per-symbol counters packed 3 x 21 bits into two 64-bit words (register-resident, no dynamic array indexing)
struct Cnt6 {
    unsigned long long a, b;
    __device__ __forceinline__ Cnt6() : a(0), b(0) {}
    __device__ __forceinline__ void add(int sym) { if (sym < 3) a += 1ull << (21 * sym); else b += 1ull << (21 * (sym - 3)); }
    __device__ __forceinline__ int get(int sym) const { return (int)(((sym < 3 ? a >> (21 * sym) : b >> (21 * (sym - 3)))) & 0x1fffff); }
};
constexpr int RMAX = 4; // 64-column chunks per wavefront per sweep: NW*RMAX*64 == WMAX

// ---------------------------------------------------------------------------------------------------------------------
// Direction codes.  The windowed DP below keeps H/E1/E2 VALUES only in the LDS ring (and, for the few rows that a far
// successor or the end node will read, in a small HBM spill area); what it streams to HBM is ONE byte per cell (+ 4 bytes
// per cell on rows with >= 2 usable predecessors) that lets the backtrack replay oracle/poa.c's decisions exactly:
//   bits 0-2  source of H, in the oracle's priority order: 0 match/mismatch, 1 E1, 2 E2, 3 insertion run (gap piece 1),
//             4 insertion run (piece 2), 5 both pieces tie
//   bit 3/4   Y1/Y2: this column is NOT where a piece-1/2 insertion run ending further right would open, i.e.
//             prefixmax_{k<j}(Hpre[k]+k e) > Hpre[j]+j e.  The oracle's "closest k with H[k]-gap(j-k)==H[j]" is the
//             largest k<j with Y==0 (oracle/poa.c:353-361; a k whose H came from F can never satisfy the equality)
//   bit 5/6   O1/O2: E-out of this cell opens here (H-oe >= Ein-e), the oracle's first test in the E state (:366)
//   bit 7     the match predecessor is not the first one of the row's plan (rows with >= 2 predecessors only)
// The predecessor ordinals (first maximum in plan order == first equality the oracle's backtrack finds) of rows with >= 2
// predecessors go to the `ord` plane: bits 0-7 match, 8-15 E1, 16-23 E2.
// ---------------------------------------------------------------------------------------------------------------------

// Explicit address spaces for the two homes of a predecessor row.  With generic pointers hipcc merges "ring slot or HBM"
// into ONE flat_load behind a pointer select, and a flat load makes every row wait on vmcnt(0) -- i.e. on the row stores
// still draining to HBM (~2 us) -- although the common case only touches LDS.
typedef int lcd_v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int lcd_lds_i32;
typedef __attribute__((address_space(3))) lcd_v4i lcd_lds_v4i;
typedef __attribute__((address_space(1))) int lcd_glb_i32;
typedef __attribute__((address_space(1))) lcd_v4i lcd_glb_v4i;
// LDS is addressed by 32-bit byte offset (an integer, not a pointer: a generic->LDS pointer cast inside a non-inlined function
// needs a null check that hipcc 7.2 mis-selects for gfx950)
typedef __attribute__((address_space(3))) uint8_t lcd_lds_u8;
__device__ __forceinline__ int lds_ld(const unsigned o) { return *(const lcd_lds_i32 *)(uintptr_t)o; }
__device__ __forceinline__ int4 lds_ld4(const unsigned o) { const lcd_v4i v = *(const lcd_lds_v4i *)(uintptr_t)o; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void lds_st4(const unsigned o, const int4 v) { lcd_v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(lcd_lds_v4i *)(uintptr_t)o = t; }
__device__ __forceinline__ int lds_ld_u8(const unsigned o) { return *(const lcd_lds_u8 *)(uintptr_t)o; }
__device__ __forceinline__ void lds_st_u8(const unsigned o, const int v) { *(lcd_lds_u8 *)(uintptr_t)o = (uint8_t)v; }
__device__ __forceinline__ unsigned lds_off(const void *p) { return (unsigned)(uintptr_t)(const lcd_lds_u8 *)p; }
__device__ __forceinline__ int glb_ld(const int *p) { return *(const lcd_glb_i32 *)p; }
__device__ __forceinline__ int4 glb_ld4(const int *p) { const lcd_v4i v = *(const lcd_glb_v4i *)p; return make_int4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void glb_st(void *p, const int v) { *(lcd_glb_i32 *)p = v; }
__device__ __forceinline__ int glb_ld_u8(const uint8_t *p) { return *(const __attribute__((address_space(1))) uint8_t *)p; }
__device__ __forceinline__ void glb_st4(int *p, const int4 v) { lcd_v4i t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *(lcd_glb_v4i *)p = t; }
// A value loaded from HBM inside a conditional must be waited for INSIDE that conditional: otherwise the compiler places the
// s_waitcnt vmcnt(0) at the join, where it also waits (in-order vmcnt) for the row stores of every iteration that did not
// take the branch -- a full HBM store round trip (~2 us) per DP row.
#define LCD_PIN(x) asm volatile("" : "+v"(x))
// acc = (acc << 1) | (a OP b): a compare into VCC and an add-with-carry of acc to itself -- two instructions per flag and no VCC -> VALU
// wait states, against v_cmp + s_nop + v_cndmask + (a share of) v_or3 with constants materialised by v_mov in the compiler's version
// (the direction code is ~40 % of a DP row's instructions).
#define LCD_PUSH_GT(acc, a, b) asm("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc")
#define LCD_PUSH_GE(acc, a, b) asm("v_cmp_ge_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc")
constexpr int CB_Y1 = 8, CB_Y2 = 16, CB_O1 = 32, CB_O2 = 64, CB_PM = 128;
constexpr int LCD_GUARD = LCD_NEG * 2; // out-of-band filler: (guard + anything a cell can add) stays below LCD_NEG, so it never wins a max

// reachability map over [bi, ei] + the row plan (per row, by topological index: remain / base / CSR of the usable
// predecessors + edge bonus).  Also marks (imap bit 1) every row whose VALUES must outlive the LDS ring: a predecessor more
// than K rows back, or a predecessor of the end node; and fills pd[] (LDS, optional) with the distance to each row's first
// predecessor for the speculative backtrack.
// Reachability map of a sub-graph [bi, ei] (oracle/poa.c align_to_subgraph: a row counts if a path from the first row reaches it inside the range; the last row always
// counts): reach[x] = OR over x's predecessors p >= bi of reach[p].  It was one lane pushing flags row by row -- four dependent trips to HBM per row, ~8 M ticks for a
// 650-row graph, and with 3 % of the reads partial-cover ones that was 11 % of the K1 chains' time (round 6, -DLCD_X_PHASESTAT).  Here wavefront 0 takes 64 rows at a
// time (lane = row): every lane fetches its row's first two predecessors' indices in parallel, flags of earlier blocks come from a bitmap in LDS, and inside the block
// rows that hang on the row before them (the backbone) are filled a run at a time with shifts of a 64-bit mask; only the other rows are taken one by one.
// `rb`: 8 bytes per 64 rows in LDS (the ring's place: nothing lives there before the rows start).  Called by wavefront 0 only; writes g.imap[bi..ei].
__device__ __attribute__((noinline)) void reach_map_wave0(const Ctx &g, const int bi, const int ei, unsigned long long *rb) {
    const int lane = threadIdx.x & 63;
    for (int base = bi, blk = 0; base <= ei; base += 64, ++blk) {
        const int row = base + lane;
        const bool valid = row <= ei;
        int p0 = -1, p1 = -1, n1 = -1;
        if (valid) {
            const int v = g.idx2node[row];
            const int ih = g.in_head[v];
            if (ih >= 0) {
                const int f0 = g.e_from[ih], n0 = g.e_next_in[ih];
                p0 = g.node2idx[f0];
                if (n0 >= 0) { const int f1 = g.e_from[n0]; n1 = g.e_next_in[n0]; p1 = g.node2idx[f1]; }
            }
        }
        if (p0 < bi) p0 = -1; // (a predecessor before the range's first row does not count)
        if (p1 < bi) p1 = -1;
        auto outside = [&](const int p) { return p >= 0 && p < base && ((rb[(p - bi) >> 6] >> ((p - bi) & 63)) & 1ull) != 0; };
        const bool ext = valid && (row == bi || row == ei || outside(p0) || outside(p1));
        const bool chain = valid && lane > 0 && p0 == row - 1 && p1 < 0 && n1 < 0;
        const bool odd = valid && !chain && (p0 >= base || p1 >= base || n1 >= 0);
        const unsigned long long C = __ballot(chain);
        unsigned long long O = __ballot(odd), R = __ballot(ext);
        auto flood = [&](unsigned long long r) { // set bits run up through consecutive rows of C
            unsigned long long m = C;
            r |= (r << 1) & m; m &= m << 1;
            r |= (r << 2) & m; m &= m << 2;
            r |= (r << 4) & m; m &= m << 4;
            r |= (r << 8) & m; m &= m << 8;
            r |= (r << 16) & m; m &= m << 16;
            r |= (r << 32) & m;
            return r;
        };
        while (O) {
            const int k = (int)__builtin_ctzll(O);
            O &= O - 1;
            if ((R >> k) & 1ull) continue;
            R = flood(R);
            const int q0 = __builtin_amdgcn_readlane(p0, k), q1 = __builtin_amdgcn_readlane(p1, k);
            bool hit = (q0 >= base && ((R >> (q0 - base)) & 1ull)) || (q1 >= base && ((R >> (q1 - base)) & 1ull));
            for (int e = __builtin_amdgcn_readlane(n1, k); e >= 0 && !hit; e = usgpr(glb_ld(g.e_next_in + e))) { // (a third, fourth ... in-edge: rare)
                const int p = usgpr(glb_ld(g.node2idx + usgpr(glb_ld(g.e_from + e))));
                if (p < bi) continue;
                hit = p >= base ? ((R >> (p - base)) & 1ull) != 0 : ((rb[(p - bi) >> 6] >> ((p - bi) & 63)) & 1ull) != 0;
            }
            if (hit) R |= 1ull << k;
        }
        R = flood(R);
        if (lane == 0) rb[blk] = R;
        if (valid) g.imap[row] = (uint8_t)((R >> lane) & 1ull);
    }
}

template <int NT>
__device__ void build_plan(Ctx &g, Smem &sm, const int bi, const int ei, const int remain_end, uint8_t *pd, const int K, int *ring, const int ring_bytes) {
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = g.n_node;
    LCD_PT0();
    if (bi == 0 && ei == n - 1) {
        for (int i = tid; i < n; i += NT) g.imap[i] = 1;
    } else {
        if (!(g.topo_mode & 1) && (size_t)((ei - bi) / 64 + 1) * 8 <= (size_t)ring_bytes) { // (LCD_DBG bit 64, as for the re-sort: the serial form)
            if (tid < 64) reach_map_wave0(g, bi, ei, (unsigned long long *)ring);
        } else {
        for (int i = bi + tid; i <= ei; i += NT) g.imap[i] = 0;
        __syncthreads();
        if (tid == 0) {
            g.imap[bi] = 1; g.imap[ei] = 1;
            for (int i = bi; i < ei; ++i) {
                if (!g.imap[i]) continue;
                for (int e = g.out_head[g.idx2node[i]]; e >= 0; e = g.e_next_out[e]) {
                    int x = g.node2idx[g.e_to[e]];
                    if (x >= bi && x <= ei) g.imap[x] = 1;
                }
            }
        }
        }
    }
    for (int e = tid; e < g.n_edge; e += NT) g.e_slot[e] = -1;
    __syncthreads();
    LCD_PT(8);
    // U rows per thread in flight (see batched_for): a row's chain is index -> node -> first in-edge -> its source -> that row's index -> its reachability, six
    // dependent loads, and nearly every row has one or two in-edges -- those are taken straight-line for all U rows together and kept for the second pass
    constexpr int U = LCD_PLAN_U;
    int carry = 0;
    for (int base = bi; base <= ei; base += U * NT) {
        int v[U], cnt[U], rem[U], vbs[U], ih[U], n0[U], n1[U], p0[U], p1[U], w0[U], w1[U]; bool us0[U], us1[U];
        {
            int vv[U], im[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int idx = base + u * NT + tid, ci = idx <= ei ? idx : ei; im[u] = g.imap[ci]; vv[u] = g.idx2node[ci]; }
            LCD_PT(9);
#pragma unroll
            for (int u = 0; u < U; ++u) { ih[u] = g.in_head[vv[u]]; rem[u] = g.remain[vv[u]]; vbs[u] = g.base[vv[u]]; }
            LCD_PT(10);
            int f0[U], f1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int c = ih[u] >= 0 ? ih[u] : 0; f0[u] = g.e_from[c]; n0[u] = ih[u] >= 0 ? g.e_next_in[c] : -1; w0[u] = g.e_w[c]; }
            LCD_PT(11);
#pragma unroll
            for (int u = 0; u < U; ++u) { const int c = n0[u] >= 0 ? n0[u] : 0; p0[u] = g.node2idx[f0[u]]; f1[u] = g.e_from[c]; n1[u] = n0[u] >= 0 ? g.e_next_in[c] : -1; w1[u] = g.e_w[c]; }
            LCD_PT(12);
            int i0[U], i1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { p1[u] = g.node2idx[f1[u]]; i0[u] = g.imap[p0[u] >= 0 && p0[u] < n ? p0[u] : 0]; }
#pragma unroll
            for (int u = 0; u < U; ++u) i1[u] = g.imap[p1[u] >= 0 && p1[u] < n ? p1[u] : 0];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * NT + tid;
                v[u] = (idx <= ei && im[u]) ? vv[u] : -1;
                us0[u] = v[u] >= 0 && ih[u] >= 0 && p0[u] >= bi && p0[u] < ei && i0[u];
                us1[u] = v[u] >= 0 && n0[u] >= 0 && p1[u] >= bi && p1[u] < ei && i1[u];
                cnt[u] = (us0[u] ? 1 : 0) + (us1[u] ? 1 : 0);
                if (v[u] >= 0) for (int e = n1[u]; e >= 0; e = g.e_next_in[e]) { const int pi = g.node2idx[g.e_from[e]]; cnt[u] += (pi >= bi && pi < ei && g.imap[pi]); }
            }
        }
        LCD_PT(13);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (base + u * NT > ei) break; // (uniform)
            const int idx = base + u * NT + tid;
            const int incl = scan_add(cnt[u]);
            int woff = 0, tot = 0;
            if constexpr (NW == 1) tot = lane63(incl); // (one wavefront: no exchange and no barrier -- a barrier drains the plan stores in flight, ~2 us each time)
            else {
                if (lane == 63) sm.scan[wave] = incl;
                lds_barrier<NT>();
#pragma unroll
                for (int k = 0; k < NW; ++k) { const int t = sm.scan[k]; if (k < wave) woff += t; tot += t; }
            }
            const int start = carry + woff + incl - cnt[u];
            if (idx <= ei) {
                g.pl_start[idx] = start;
                g.pl_rem[idx] = v[u] >= 0 ? rem[u] - remain_end : (1 << 30); // 1<<30: row not reachable from beg
                g.pl_base[idx] = v[u] >= 0 ? vbs[u] : 4;
                int first = 255;
                if (v[u] >= 0) {
                    int k = start;
                    auto put = [&](const int e, const int pi, const int wt) {
                        g.pl_pidx[k] = pi; g.pl_bonus[k] = ilog2_32(wt); g.e_slot[e] = k;
                        if (k == start && idx - pi < 255) first = idx - pi;
                        if (idx == ei || idx - pi > K) g.imap[pi] = 3; // same value from every writer
                        ++k;
                    };
                    if (us0[u]) put(ih[u], p0[u], w0[u]);
                    if (us1[u]) put(n0[u], p1[u], w1[u]);
                    for (int e = n1[u]; e >= 0; e = g.e_next_in[e]) {
                        const int pi = g.node2idx[g.e_from[e]];
                        if (pi >= bi && pi < ei && g.imap[pi]) put(e, pi, g.e_w[e]);
                    }
                }
                if (pd) pd[idx - bi] = (uint8_t)first;
            }
            carry += tot;
            if constexpr (NW > 1) lds_barrier<NT>();
        }
        LCD_PT(14);
    }
    if (tid == 0) { g.pl_start[ei + 1] = carry; g.pl_start[ei + 2] = carry; g.pl_start[ei + 3] = carry; }
    __syncthreads();
}

// ================= windowed DP (banded K1 rows and unbanded K2 rows that fit one sweep of the workgroup) =================
// Every lane owns FOUR consecutive columns of a WIN-column window [beg4, beg4+WIN) that starts at the row's band (beg4 = beg
// rounded down to 4), so a row is ONE sweep: predecessor rows come from the LDS ring as ds_read_b128, the horizontal-gap
// prefix is 3 in-lane max + ONE interleaved DPP scan pair per 256 cells, the row maximum is one DPP scan + a ballot.
// A ring slot is addressed by (column mod WIN) and is rewritten in full by every row (LCD_GUARD outside the band), so rows
// whose windows are shifted against each other need no per-cell bounds test; the few alias cases (a predecessor band that
// reaches a full window away) are detected per row and make the caller fall back to the generic rows.
// Returns the number of cigar entries (written at g.cig_node0/g.cig_qpos0 + *cig_pos), or -1 = not representable here.

// Arguments of a non-inlined device function arrive in VGPRs (and a by-value struct through the stack), so the compiler
// treats them as per-lane values: 64-bit pointers cost VGPR pairs and every "scalar" computation runs on the vector ALU.
// These put workgroup-uniform values back into SGPRs.
__device__ __forceinline__ int usgpr(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned usgpr(const unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long usgpr(const unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
template <typename T> __device__ __forceinline__ T *usgpr(T *p) { return (T *)usgpr((unsigned long long)p); }
__device__ void ctx_to_sgpr(Ctx &g) {
    g.code8 = usgpr(g.code8); g.ord = usgpr(g.ord); g.spill = usgpr(g.spill);
    g.rbeg = usgpr(g.rbeg); g.rend = usgpr(g.rend); g.roff = usgpr(g.roff); g.ooff = usgpr(g.ooff); g.spoff = usgpr(g.spoff);
    g.ml = usgpr(g.ml); g.mr = usgpr(g.mr); g.idx2node = usgpr(g.idx2node);
    g.cig_node0 = usgpr(g.cig_node0); g.cig_qpos0 = usgpr(g.cig_qpos0); g.imap = usgpr(g.imap);
    g.tb = usgpr(g.tb); g.cert = usgpr(g.cert); g.node_cap = usgpr(g.node_cap);
    g.pl_start = usgpr(g.pl_start); g.pl_pidx = usgpr(g.pl_pidx); g.pl_bonus = usgpr(g.pl_bonus); g.pl_rem = usgpr(g.pl_rem); g.pl_base = usgpr(g.pl_base);
    g.wd_deadline = usgpr(g.wd_deadline); g.wmax = usgpr(g.wmax); g.pool_words = usgpr(g.pool_words); g.seq_cap = usgpr(g.seq_cap); g.cell_cap = usgpr(g.cell_cap); g.status = usgpr(g.status); g.spill_x = usgpr(g.spill_x); g.ring_k = usgpr(g.ring_k); g.plan_k = usgpr(g.plan_k);
}
// End node (best predecessor at column qlen; its values are in the spill area) + the code-driven backtrack, on wavefront 0.
// Results through sm.bc[0] = #cigar entries, [1] = status, [4] = first cigar slot.
__device__ __forceinline__ void code_backtrack(const Ctx &g, Smem &sm, const unsigned pd, const int bi, const int ei, const int qlen, const int SLOTW, const int WM,
                                               const int lane, const int amask /* rows start at their band's first column rounded down to the lane's cell group: ~(C - 1) */) {
        int best = LCD_NEG, br = -1;
        {
            const int p0 = glb_ld(g.pl_start + (ei)), np = glb_ld(g.pl_start + (ei + 1)) - p0;
            for (int t = 0; t < np; ++t) {
                const int pi = glb_ld(g.pl_pidx + (p0 + t));
                if (qlen < glb_ld(g.rbeg + (pi)) || qlen > glb_ld(g.rend + (pi))) continue;
                const int c = glb_ld(g.spill + ((size_t)(unsigned)glb_ld((const int *)g.spoff + (pi)) * SLOTW + (qlen & WM))) + glb_ld(g.pl_bonus + (p0 + t));
                if (c > best) { best = c; br = pi; }
            }
        }
        int pos = qlen;
        int status = g.status;
        if (br >= 0 && best > LCD_NEG / 2 && status == LCD_OK) {
            int i = br, j = qlen, st = 0;
            // speculation width: 64 steps of the first-predecessor chain per attempt while the runs of matches are long (clean reads: hundreds
            // of columns); halved down to 8 when they turn out short (noisy reads: a difference every ~20 columns) -- the chain is followed
            // lane-serially through LDS before the codes can be fetched, so a 64-step look-ahead for a 3-step run was most of the backtrack there
            int sw = 64;
            unsigned wd_it = 0; int wd_left = -1; // (a backtrack that does not end: the last four states (row, column, state) go out with the error -- LCD_ERR_WATCHDOG, sm.prof -> PoaChainOut.t_plan ...)
            while (i != bi && j > 0 && status == LCD_OK) {
                if (wd_left < 0 && (++wd_it & 255u) == 0 && (unsigned long long)clock64() > g.wd_deadline) wd_left = 3;
                if (wd_left >= 0) {
                    if (lane == 0) sm.prof[3 - wd_left] = (unsigned long long)(unsigned)i | ((unsigned long long)(unsigned)(j & 0xffffff) << 32) | ((unsigned long long)(unsigned)st << 56);
                    if (wd_left-- == 0) {
                        status = LCD_ERR_WATCHDOG; break;
                    }
                }
                if (st == 0 && pd != 0xffffffffu) {
                    // speculate a run of matches along first predecessors: lane t looks at the cell t steps up the diagonal
                    int my_i = -1, my_nx = -1;
                    // the usual stretch is the backbone: every row's first predecessor is the row before it, so lane t's row is i - t and ONE parallel look at the
                    // distances replaces the lane-serial walk through them (64 dependent LDS reads per 64 cells -- most of a clean read's backtrack)
                    int nbb = 0;
                    {
                        const int ci = i - lane;
                        const bool okl = lane < sw && ci > bi && j - lane > 0;
                        const int d = okl ? lds_ld_u8(pd + (ci - bi)) : 255;
                        const unsigned long long one = __ballot(okl && d == 1);
                        nbb = one == ~0ull ? 64 : (int)__builtin_ctzll(~one); // lanes 0 .. nbb - 1 step along the backbone
                        if (nbb >= 16 || nbb >= sw) {
                            if (lane < nbb) { my_i = ci; my_nx = ci - 1; }
                            else if (lane == nbb && okl) { my_i = ci; my_nx = d != 255 ? ci - d : -1; }
                        } else nbb = -1;
                    }
                    if (nbb < 0) {
                        int cur = i; bool alive = true;
                        for (int t = 0; t < sw; ++t) {
                            const bool ok = alive && cur != bi && j - t > 0;
                            const int d = ok ? lds_ld_u8(pd + (cur - bi)) : 255;
                            if (lane == t) { my_i = ok ? cur : -1; my_nx = d != 255 ? cur - d : -1; }
                            alive = ok && d != 255;
                            if (!alive) break;
                            cur -= d;
                        }
                    }
                    // A full 64-step stretch of backbone: look at the next three stretches of 64 in the same round trips (the loads of all four are issued together;
                    // runs of matches of a clean read are hundreds of cells long, and every round of this loop is a chain metadata -> code -> node of dependent loads)
                    if (nbb == 64 && sw == 64) {
                        int ex = 0; // further steps along the backbone, a multiple of 64 lanes' worth or less
                        int dd[3];
#pragma unroll
                        for (int u = 0; u < 3; ++u) { const int ci = i - 64 * (u + 1) - lane; const bool okl = ci > bi && j - 64 * (u + 1) - lane > 0; dd[u] = okl ? lds_ld_u8(pd + (ci - bi)) : 255; }
                        bool open = true;
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const unsigned long long one = __ballot(dd[u] == 1);
                            const int c = one == ~0ull ? 64 : (int)__builtin_ctzll(~one);
                            if (open) ex += c;
                            open = open && c == 64;
                        }
                        // cells 0 .. 63 + ex: all of them step to the row before (the cell AFTER the last one is reached by row - 1 as well)
                        int rbv[4], rev[4], rov[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int ci = imax(i - 64 * u - lane, bi); rbv[u] = glb_ld(g.rbeg + ci); rev[u] = glb_ld(g.rend + ci); rov[u] = glb_ld((const int *)g.roff + ci); }
                        int cdv[4]; bool inr[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int jj = j - 64 * u - lane;
                            inr[u] = 64 * u + lane < 64 + ex && jj >= rbv[u] && jj <= rev[u];
                            cdv[u] = glb_ld_u8(g.code8 + ((size_t)(unsigned)rov[u] + (inr[u] ? jj - (rbv[u] & amask) : 0)));
                        }
                        int mtot = 0; open = true;
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool good = inr[u] && (cdv[u] & (7 | CB_PM)) == 0;
                            const unsigned long long bad = __ballot(!good);
                            const int c = bad ? __ffsll((long long)bad) - 1 : 64;
                            if (open) mtot += c;
                            open = open && c == 64;
                        }
                        if (mtot > 0) {
                            int nodev[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) nodev[u] = glb_ld(g.idx2node + imax(i - 64 * u - lane, bi));
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int st_ = 64 * u + lane;
                                if (st_ < mtot) { g.cig_node0[pos - 1 - st_] = nodev[u]; g.cig_qpos0[pos - 1 - st_] = j - st_ - 1; }
                            }
                            i -= mtot; pos -= mtot; j -= mtot;
                            continue;
                        }
                    }
                    bool good = false;
                    const int jj = j - lane;
                    if (my_i >= 0 && my_nx >= 0) {
                        const int rb = glb_ld(g.rbeg + (my_i)), re = glb_ld(g.rend + (my_i));
                        if (jj >= rb && jj <= re) good = (glb_ld_u8(g.code8 + ((size_t)(unsigned)glb_ld((const int *)g.roff + (my_i)) + (jj - (rb & amask)))) & (7 | CB_PM)) == 0;
                    }
                    const unsigned long long bad = __ballot(!good);
                    const int m = bad ? __ffsll((long long)bad) - 1 : 64;
                    if (m >= sw) sw = sw < 64 ? sw * 2 : 64; else if (m * 4 < sw && sw > 8) sw >>= 1;
                    if (m > 0) {
                        if (lane < m) { g.cig_node0[pos - 1 - lane] = glb_ld(g.idx2node + (my_i)); g.cig_qpos0[pos - 1 - lane] = jj - 1; }
                        i = LCD_RL(my_nx, __builtin_amdgcn_readfirstlane(m - 1));
                        pos -= m; j -= m;
                        continue;
                    }
                }
                // one step, replaying the oracle's decision from the code
                const int rb = glb_ld(g.rbeg + (i)), rb4 = rb & amask;
                const size_t ro = (unsigned)glb_ld((const int *)g.roff + (i));
                const int c = glb_ld_u8(g.code8 + (ro + (j - rb4)));
                const int p0 = glb_ld(g.pl_start + (i)), np = glb_ld(g.pl_start + (i + 1)) - p0;
                const int ow = np > 1 ? glb_ld(g.ord + ((size_t)(unsigned)glb_ld((const int *)g.ooff + (i)) + (j - rb4))) : 0;
                if (st == 0) {
                    const int hs = c & 7;
                    if (hs == 0) {
                        if (lane == 0) { g.cig_node0[pos - 1] = glb_ld(g.idx2node + (i)); g.cig_qpos0[pos - 1] = j - 1; }
                        --pos; i = glb_ld(g.pl_pidx + (p0 + (ow & 255))); --j;
                    } else if (hs <= 2) {
                        i = glb_ld(g.pl_pidx + (p0 + ((ow >> (8 * hs)) & 255))); st = hs;
                    } else if (hs <= 5) { // insertion run: back to the closest opening column of a matching gap piece
                        int k = -1;
                        if (hs != 4) { int p = j - 1; while (p > rb && (glb_ld_u8(g.code8 + (ro + (p - rb4))) & CB_Y1)) --p; k = p; }
                        if (hs != 3) { int p = j - 1; while (p > rb && (glb_ld_u8(g.code8 + (ro + (p - rb4))) & CB_Y2)) --p; k = imax(k, p); }
                        const int nins = j - k;
                        for (int u = lane; u < nins; u += 64) { g.cig_node0[pos - nins + u] = -1; g.cig_qpos0[pos - nins + u] = k + u; }
                        pos -= nins; j = k;
                    } else status = LCD_ERR_BACKTRACK;
                } else {
                    if (c & (st == 1 ? CB_O1 : CB_O2)) st = 0;
                    else i = glb_ld(g.pl_pidx + (p0 + ((ow >> (8 * st)) & 255)));
                }
            }
            for (int u = lane; u < j; u += 64) { g.cig_node0[pos - j + u] = -1; g.cig_qpos0[pos - j + u] = u; }
            pos -= j;
        }
        if (lane == 0) { sm.bc[0] = qlen - pos; sm.bc[1] = status; sm.bc[4] = pos; sm.bc[5] = best; }
    }

struct WinOut { int status; unsigned long long t_dp, t_bt, cells; int cig_pos; unsigned long long t_plan, t_poll; int score; unsigned long long t_setup;
                int clobber; /* the rows' ring outgrew the pool's layout and took the place of the query cache / first-predecessor distances: whatever runs next for this read must not trust `pd` */ };
// (not inlined, context by value: the row loop then only carries the dozen pointers it uses instead of the chain's whole
//  context -- inlined, hipcc spilled the scalar registers of ~45 pointers into VGPR lanes and re-read them every row)
// C consecutive ints from / to LDS (byte offset) or HBM: one ds_read_b128 / b64 / b32 (global_load_dwordx4 / x2 / dword)
typedef int lcd_v2i __attribute__((ext_vector_type(2)));
template <int C> __device__ __forceinline__ void lds_ldc(const unsigned o, int (&v)[C]) {
    if constexpr (C == 8) { const lcd_v4i t = *(const lcd_lds_v4i *)(uintptr_t)o, u = *(const lcd_lds_v4i *)(uintptr_t)(o + 16); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w; }
    else if constexpr (C == 4) { const lcd_v4i t = *(const lcd_lds_v4i *)(uintptr_t)o; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (C == 2) { const lcd_v2i t = *(const __attribute__((address_space(3))) lcd_v2i *)(uintptr_t)o; v[0] = t.x; v[1] = t.y; }
    else v[0] = *(const lcd_lds_i32 *)(uintptr_t)o;
}
template <int C> __device__ __forceinline__ void lds_stc(const unsigned o, const int (&v)[C]) {
    if constexpr (C == 8) { lcd_v4i t, u; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; u.x = v[4]; u.y = v[5]; u.z = v[6]; u.w = v[7]; *(lcd_lds_v4i *)(uintptr_t)o = t; *(lcd_lds_v4i *)(uintptr_t)(o + 16) = u; }
    else if constexpr (C == 4) { lcd_v4i t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; *(lcd_lds_v4i *)(uintptr_t)o = t; }
    else if constexpr (C == 2) { lcd_v2i t; t.x = v[0]; t.y = v[1]; *(__attribute__((address_space(3))) lcd_v2i *)(uintptr_t)o = t; }
    else *(lcd_lds_i32 *)(uintptr_t)o = v[0];
}
// 16-bit ring values (certified-band K2 chains of the single-wavefront class: H, E1, E2 of reads below 15 000 bases fit int16; the ring is three quarters of such a
// chain's LDS pool, and LDS x time is what a submission runs out of first).  Stores saturate (v_cvt_pk_i16_i32); everything at the floor -- the fillers outside a
// row's interval, unreachable cells -- comes back as LCD_GUARD: below every real value, which is all the rows ask of them
typedef short lcd_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk16(const int a, const int b) { const lcd_v2s t = __builtin_amdgcn_cvt_pk_i16(a, b); return __builtin_bit_cast(unsigned, t); }
__device__ __forceinline__ int un16(const int v) { return v == -32768 ? LCD_GUARD : v; }
__device__ __forceinline__ int lds_ld16(const unsigned o) { return un16((int)*(const __attribute__((address_space(3))) short *)(uintptr_t)o); }
template <int C> __device__ __forceinline__ void lds_ldc16(const unsigned o, int (&v)[C]) {
    if constexpr (C == 1) v[0] = lds_ld16(o);
    else {
        unsigned w[C / 2];
        if constexpr (C == 2) w[0] = (unsigned)*(const lcd_lds_i32 *)(uintptr_t)o;
        else if constexpr (C == 4) { const lcd_v2i t = *(const __attribute__((address_space(3))) lcd_v2i *)(uintptr_t)o; w[0] = (unsigned)t.x; w[1] = (unsigned)t.y; }
        else { const lcd_v4i t = *(const lcd_lds_v4i *)(uintptr_t)o; w[0] = (unsigned)t.x; w[1] = (unsigned)t.y; w[2] = (unsigned)t.z; w[3] = (unsigned)t.w; }
#pragma unroll
        for (int k = 0; k < C / 2; ++k) { v[2 * k] = un16((int)(short)(w[k] & 0xffffu)); v[2 * k + 1] = un16((int)w[k] >> 16); }
    }
}
template <int C> __device__ __forceinline__ void lds_stc16(const unsigned o, const int (&v)[C]) {
    if constexpr (C == 1) *(__attribute__((address_space(3))) short *)(uintptr_t)o = (short)(pk16(v[0], v[0]) & 0xffffu);
    else if constexpr (C == 2) *(lcd_lds_i32 *)(uintptr_t)o = (int)pk16(v[0], v[1]);
    else if constexpr (C == 4) { lcd_v2i t; t.x = (int)pk16(v[0], v[1]); t.y = (int)pk16(v[2], v[3]); *(__attribute__((address_space(3))) lcd_v2i *)(uintptr_t)o = t; }
    else { lcd_v4i t; t.x = (int)pk16(v[0], v[1]); t.y = (int)pk16(v[2], v[3]); t.z = (int)pk16(v[4], v[5]); t.w = (int)pk16(v[6], v[7]); *(lcd_lds_v4i *)(uintptr_t)o = t; }
}
template <int C> __device__ __forceinline__ void glb_ldc(const int *p, int (&v)[C]) {
    if constexpr (C == 8) { const lcd_v4i t = *(const lcd_glb_v4i *)p, u = *(const lcd_glb_v4i *)(p + 4); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; v[4] = u.x; v[5] = u.y; v[6] = u.z; v[7] = u.w; }
    else if constexpr (C == 4) { const lcd_v4i t = *(const lcd_glb_v4i *)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else if constexpr (C == 2) { const lcd_v2i t = *(const __attribute__((address_space(1))) lcd_v2i *)p; v[0] = t.x; v[1] = t.y; }
    else v[0] = *(const lcd_glb_i32 *)p;
}
template <int C> __device__ __forceinline__ void glb_stc(int *p, const int (&v)[C]) {
    if constexpr (C == 8) { lcd_v4i t, u; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; u.x = v[4]; u.y = v[5]; u.z = v[6]; u.w = v[7]; *(lcd_glb_v4i *)p = t; *(lcd_glb_v4i *)(p + 4) = u; }
    else if constexpr (C == 4) { lcd_v4i t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3]; *(lcd_glb_v4i *)p = t; }
    else if constexpr (C == 2) { lcd_v2i t; t.x = v[0]; t.y = v[1]; *(__attribute__((address_space(1))) lcd_v2i *)p = t; }
    else *(lcd_glb_i32 *)p = v[0];
}

// C = cells per lane = WIN / NT.  C = 4 is the general shape; the single-wavefront class also has C = 2 and C = 1 for narrow bands
// (a 40-column HiFi band uses 10 of 64 lanes at four cells per lane, and every lane pays the cell code four times: fewer cells per
// lane means proportionally fewer instructions per row for the same band).
// MODE 0: rows span the whole window (w = qlen); 1: the oracle's adaptive band; 2: the rows' column intervals come from a table (g.cert hull, see
// align_certified: cells outside an interval count as unreachable, exactly like cells outside an adaptive band)
template <bool SOLO> __device__ __forceinline__ void win_sync() {
    if (SOLO) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else __syncthreads();
}
// SOLO: the function is run by wavefront 0 of a WIDER workgroup (NT = 64 here, the others wait at a barrier of their own): its workgroup barriers become waits
// of the one wavefront on its own memory operations
template <int NT, int MODE, int C, bool SOLO = false>
__device__ __attribute__((noinline)) int align_windowed(const Ctx *gp_, const unsigned ring_, const unsigned sq1_, const unsigned pd_ /* 0xffffffff: none */, const LcdScoring sc_, const int w_,
                              const int bi_, const int ei_, const int rem_beg_, const uint8_t *seq_hbm_, const int qlen_,
                              WinOut *wo_) {
    constexpr int NW = NT / 64;
    constexpr int WIN = NT * C, WM = WIN - 1, SLOTW = 3 * WIN, CM = ~(C - 1);
    constexpr bool BANDED = MODE == 1, FIXED = MODE == 2, BND = MODE != 0;
    Smem &sm = g_smem;
    Ctx g = *usgpr(gp_); // (by pointer: a by-value context is 440 B of outgoing-argument stack per call site and per lane)
    ctx_to_sgpr(g);
    // ring slots: per class, except the single-wavefront class whose slot metadata lives in lanes -- there the host gives long chains of noisy reads more
    // (PoaChain.ring_k: a predecessor further back than the ring costs two dependent trips to HBM, metadata then values)
    int K = NT == 64 ? g.ring_k : Cfg<NT>::K;
    const unsigned ring = usgpr(ring_); unsigned sq1 = usgpr(sq1_), pd = usgpr(pd_);
    const int w = usgpr(w_), bi = usgpr(bi_), ei = usgpr(ei_), rem_beg = usgpr(rem_beg_), qlen = usgpr(qlen_);
    const uint8_t *seq_hbm = usgpr(seq_hbm_); WinOut *wo = usgpr(wo_);
    LcdScoring sc; sc.match = usgpr(sc_.match); sc.mismatch = usgpr(sc_.mismatch); sc.o1 = usgpr(sc_.o1); sc.e1 = usgpr(sc_.e1); sc.o2 = usgpr(sc_.o2); sc.e2 = usgpr(sc_.e2); sc.dbg = usgpr(sc_.dbg);
    const int tid = threadIdx.x, lane = tid & 63, wave = usgpr(tid >> 6); // (wave in an SGPR: branches on it are scalar)
    const int o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2, oe1 = o1 + e1, oe2 = o2 + e2;
    const int QB = (qlen + 12 + 15) & ~15;
    if constexpr (NT == 64 && BND) {
        // the pool of a single-wavefront chain is laid out for the window the host expects (PoaChain.wmax columns per ring slot); a wider
        // window moves the query cache up and gives up the first-predecessor distances -- or, if the pool is too small for that, leaves
        // the read to the next wider window / the generic rows
        unsigned ring_bytes = (unsigned)(K * SLOTW * 4);
        if (sq1 < ring + ring_bytes) {
            if (ring_bytes + (unsigned)QB > (unsigned)g.pool_words * 4u && K > g.plan_k) { // the extra slots were room the narrower window left: this one runs with
                K = g.plan_k; ring_bytes = (unsigned)(K * SLOTW * 4);                        // the slots the plan's spill flags were made for
            }
            if (ring_bytes + (unsigned)QB > (unsigned)g.pool_words * 4u) return -1;
            if (sq1 < ring + ring_bytes) { sq1 = ring + ring_bytes; pd = 0xffffffffu; wo->clobber = 1; }
        }
    }
    for (int j = tid; j < QB; j += NT) lds_st_u8(sq1 + j, (j >= 1 && j <= qlen) ? seq_hbm[j - 1] : 4); // shifted: sq1[j] = q[j-1]
    const int qclamp = QB - 4;
    const unsigned long long code_cap = g.cell_cap, ord_cap = g.spill_x > 2 ? g.cell_cap : g.cell_cap / 4; // bytes / ints (arena partition: see the kernel prologue)
    const long long spill_rows = g.cell_cap * g.spill_x > 64 ? (long long)((g.cell_cap * g.spill_x - 64) / ((unsigned long long)SLOTW * 4)) : 0;
    // ---- source row (slot 0, window at column 0) ----
    int end0 = qlen - rem_beg; if (end0 < 0) end0 = 0; end0 += w; if (end0 > qlen) end0 = qlen;
    const int *const hull = g.cert + 6 * (size_t)g.node_cap;
    if (FIXED) { const int hw = glb_ld(hull + bi); end0 = hw >> 16; if ((hw & 65535) != 0) return -1; } // (the source row's interval starts at column 0)
    if (end0 + 2 > WIN) return -1;
    int nsp = 0;
    {
        const bool spf = (g.imap[bi] & 2) != 0;
        if (spf && spill_rows < 1) { wo->status = LCD_ERR_CELLS; return 0; }
        int hh[C], aa[C], bb[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int j = C * tid + k;
            if (j <= end0) {
                const int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
                const int h = j ? imax(f1, f2) : 0;
                hh[k] = h; aa[k] = h - oe1; bb[k] = h - oe2;
            } else { hh[k] = LCD_GUARD; aa[k] = LCD_GUARD; bb[k] = LCD_GUARD; }
        }
        lds_stc<C>(ring + 4 * (C * tid), hh); lds_stc<C>(ring + 4 * (WIN + C * tid), aa); lds_stc<C>(ring + 4 * (2 * WIN + C * tid), bb);
        if (spf) { int *G = g.spill; glb_stc<C>(G + C * tid, hh); glb_stc<C>(G + WIN + C * tid, aa); glb_stc<C>(G + 2 * WIN + C * tid, bb); }
        if (tid == 0) { g.rbeg[bi] = 0; g.rend[bi] = end0; g.roff[bi] = 0; g.ml[bi] = 0; g.mr[bi] = 0; g.spoff[bi] = 0; }
        if (spf) nsp = 1;
    }
    // ring slot meta: lane s of every wavefront holds (beg, end, row-max leftmost / rightmost column) of slot s
    int m_beg = 1, m_end = 0, m_ml = 0, m_mr = 0;
    if (lane == 0) { m_beg = 0; m_end = end0; }
    unsigned long long cused = 0, oused = 0, ncell = (unsigned long long)end0 + 1;
    win_sync<SOLO>();
    const long long t_dp0 = clock64();
    int wbase = -(1 << 20);
    int w_p0 = 0, w_np = 0, w_rem = 1 << 30, w_vb = 4, w_pi0 = 0, w_b0 = 0, w_pi1 = 0, w_b1 = 0, w_sp = 0, w_hull = 1;
    for (int idx = bi + 1; idx < ei; ++idx) {
        if (idx - wbase >= 64) { // plan window: each lane loads the plan of one upcoming row; rows then take it by v_readlane
            if constexpr (NT == 64) if ((unsigned long long)clock64() > g.wd_deadline) { wo->status = LCD_ERR_WATCHDOG; return 0; }
            wbase = idx;
            const int ri = idx + lane;
            w_np = 0; w_rem = 1 << 30;
            if (ri < ei) {
                const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
                w_p0 = s0; w_np = s1 - s0; w_rem = glb_ld(g.pl_rem + ri); w_vb = glb_ld_u8(g.pl_base + ri); w_sp = glb_ld_u8(g.imap + ri) & 2;
                if (FIXED) { w_hull = glb_ld(hull + ri); LCD_PIN(w_hull); }
                if (w_np > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); w_b0 = glb_ld(g.pl_bonus + s0); }
                if (w_np > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); w_b1 = glb_ld(g.pl_bonus + s0 + 1); }
            }
            LCD_PIN(w_p0); LCD_PIN(w_np); LCD_PIN(w_rem); LCD_PIN(w_vb); LCD_PIN(w_sp); LCD_PIN(w_pi0); LCD_PIN(w_b0); LCD_PIN(w_pi1); LCD_PIN(w_b1);
        }
        const int wk = idx - wbase;
        const int np = LCD_RL(w_np, wk), rem = LCD_RL(w_rem, wk), vb = LCD_RL(w_vb, wk);
        const int pi0 = LCD_RL(w_pi0, wk), bz0 = LCD_RL(w_b0, wk);
        const bool spf = LCD_RL(w_sp, wk) != 0;
        // (second predecessor and the plan offset: only rows with more than one predecessor read them)
        int p0 = 0, pi1 = 0, bz1 = 0;
        if (np > 1) { p0 = LCD_RL(w_p0, wk); pi1 = LCD_RL(w_pi1, wk); bz1 = LCD_RL(w_b1, wk); }
        const int s = (idx - bi) & (K - 1);
        if (rem == (1 << 30)) { // not reachable
            if (lane == s) { m_beg = 1; m_end = 0; }
            if (tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
            continue;
        }
        bool synced = false;
        int beg = 0, end = qlen;
        const bool fast1 = BND && np == 1 && idx - pi0 <= K;
        const int f_sp = (pi0 - bi) & (K - 1);
        int f_pb = 1, f_pe = 0;
        if (fast1) { f_pb = LCD_RL(m_beg, f_sp); f_pe = LCD_RL(m_end, f_sp); }
        if (BANDED) { // band: pulled from the predecessors' row-max columns (same values the oracle pushes to successors)
            int mplv = 1 << 30, mprv = 0, minpb = 1 << 30, maxpe = -1;
            if (fast1) { // the usual row: one usable predecessor whose values are still in the ring (its slot metadata read once, here)
                if (f_pb <= f_pe) { minpb = f_pb; maxpe = f_pe; mplv = LCD_RL(m_ml, f_sp) + 1; mprv = LCD_RL(m_mr, f_sp) + 1; }
            } else
            for (int t = 0; t < np; ++t) {
                int pi = t == 0 ? pi0 : pi1;
                if (t > 1) { pi = glb_ld(g.pl_pidx + p0 + t); LCD_PIN(pi); }
                int pb, pe, pml, pmr;
                if (idx - pi <= K) { const int sp = (pi - bi) & (K - 1); pb = LCD_RL(m_beg, sp); pe = LCD_RL(m_end, sp); pml = LCD_RL(m_ml, sp); pmr = LCD_RL(m_mr, sp); }
                else {
                    if (!synced) { win_sync<SOLO>(); synced = true; } // far row: its metadata / spilled values were stored to HBM earlier
                    pb = g.rbeg[pi]; pe = g.rend[pi]; pml = g.ml[pi]; pmr = g.mr[pi];
                    LCD_PIN(pb); LCD_PIN(pe); LCD_PIN(pml); LCD_PIN(pmr);
                }
                if (pb > pe) continue;
                minpb = imin(minpb, pb); maxpe = imax(maxpe, pe);
                mplv = imin(mplv, pml + 1); mprv = imax(mprv, pmr + 1);
            }
            beg = imin(mplv, qlen - rem) - w; if (beg < 0) beg = 0;
            end = imax(mprv, qlen - rem) + w; if (end > qlen) end = qlen;
            if (beg < minpb) beg = minpb;
            if (end > maxpe + 1) end = maxpe + 1;
            if (beg > end) { // empty row
                if (lane == s) { m_beg = 1; m_end = 0; }
                if (tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
                continue;
            }
            if (end - (beg & CM) + 2 > WIN) return -1;
        }
        if (FIXED) { // the row's certified interval (lo | hi << 16; lo > hi: no cell of the row can lie on an optimal path)
            const int hw = LCD_RL(w_hull, wk);
            beg = hw & 65535; end = hw >> 16;
            if (beg > end) {
                if (lane == s) { m_beg = 1; m_end = 0; }
                if (tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
                continue;
            }
            if (end - (beg & CM) + 2 > WIN) return -1;
        }
        const int begc = beg & CM;
        const int jb = begc + C * tid;
        const int x = jb & WM, xm = (jb - 1) & WM;
        bool inb[C]; int sk[C];
        {
            const int jq = imin(jb, qclamp);
            const unsigned qw = (unsigned)lds_ld(sq1 + (jq & ~3)) >> (8 * (jq & 3)); // q[jb-1], q[jb], ...: C <= 4 bytes from one aligned word
#pragma unroll
            for (int k = 0; k < C; ++k) {
                inb[k] = jb + k >= beg && jb + k <= end;
                const int q = (qw >> (8 * k)) & 255;
                sk[k] = (vb >= 4 || q >= 4) ? 0 : (vb == q ? sc.match : -sc.mismatch);
            }
        }
        // ---- phase A: best match / E1 / E2 input of the cells over the predecessors (first maximum keeps its ordinal) ----
        int nn[C], uu[C], vv[C];
#pragma unroll
        for (int k = 0; k < C; ++k) { nn[k] = LCD_NEG; uu[k] = LCD_NEG; vv[k] = LCD_NEG; }
        int om = 0, oa = 0, ob = 0; // ordinals, one byte per cell
        if (fast1) {
            if (f_pb <= f_pe) {
                // a ring slot is addressed by (column mod WIN): columns of this row that lie a full window away from the predecessor's band would read
                // that band's values instead of the filler -- rare (bands close to the window's width, shifted against each other), so those rows mask
                // the loaded values by the predecessor's [beg, end] explicitly instead of giving the whole read up
                const bool risk = f_pe - beg + 2 >= WIN || end - f_pb + 1 >= WIN;
                const unsigned S = ring + 4 * f_sp * SLOTW;
                int hv[C], av[C], bv[C];
                int hm = lds_ld(S + 4 * xm); lds_ldc<C>(S + 4 * x, hv); lds_ldc<C>(S + 4 * (WIN + x), av); lds_ldc<C>(S + 4 * (2 * WIN + x), bv);
                if (risk) {
                    if (jb - 1 < f_pb || jb - 1 > f_pe) hm = LCD_GUARD;
#pragma unroll
                    for (int k = 0; k < C; ++k) if (jb + k < f_pb || jb + k > f_pe) { hv[k] = LCD_GUARD; av[k] = LCD_GUARD; bv[k] = LCD_GUARD; }
                }
#pragma unroll
                for (int k = 0; k < C; ++k) { nn[k] = imax(LCD_NEG, (k == 0 ? hm : hv[k - 1]) + sk[k] + bz0); uu[k] = imax(LCD_NEG, av[k] + bz0); vv[k] = imax(LCD_NEG, bv[k] + bz0); }
            }
        } else
        for (int t = 0; t < np; ++t) {
            int pi = t == 0 ? pi0 : pi1, bz = t == 0 ? bz0 : bz1;
            if (t > 1) { pi = glb_ld(g.pl_pidx + p0 + t); bz = glb_ld(g.pl_bonus + p0 + t); LCD_PIN(pi); LCD_PIN(bz); }
            const bool near = idx - pi <= K;
            const int sp = (pi - bi) & (K - 1);
            bool risk = false; int pb = 0, pe = 0;
            if (BND) {
                if (near) { pb = LCD_RL(m_beg, sp); pe = LCD_RL(m_end, sp); } else { pb = g.rbeg[pi]; pe = g.rend[pi]; LCD_PIN(pb); LCD_PIN(pe); }
                if (pb > pe) continue;
                risk = pe - beg + 2 >= WIN || end - pb + 1 >= WIN; // (see the single-predecessor rows above: masked, not given up)
            }
            int hm, hv[C], av[C], bv[C];
            if (near) {
                const unsigned S = ring + 4 * sp * SLOTW;
                hm = lds_ld(S + 4 * xm); lds_ldc<C>(S + 4 * x, hv); lds_ldc<C>(S + 4 * (WIN + x), av); lds_ldc<C>(S + 4 * (2 * WIN + x), bv);
            } else {
                if (!synced) { win_sync<SOLO>(); synced = true; }
                const int *G = g.spill + (size_t)(unsigned)glb_ld((const int *)g.spoff + pi) * SLOTW;
                hm = glb_ld(G + xm); glb_ldc<C>(G + x, hv); glb_ldc<C>(G + WIN + x, av); glb_ldc<C>(G + 2 * WIN + x, bv);
                LCD_PIN(hm);
#pragma unroll
                for (int k = 0; k < C; ++k) { LCD_PIN(hv[k]); LCD_PIN(av[k]); LCD_PIN(bv[k]); }
            }
            if (risk) {
                if (jb - 1 < pb || jb - 1 > pe) hm = LCD_GUARD;
#pragma unroll
                for (int k = 0; k < C; ++k) if (jb + k < pb || jb + k > pe) { hv[k] = LCD_GUARD; av[k] = LCD_GUARD; bv[k] = LCD_GUARD; }
            }
            const int tt = t > 255 ? 255 : t;
#pragma unroll
            for (int k = 0; k < C; ++k) {
                const int c = (k == 0 ? hm : hv[k - 1]) + sk[k] + bz, a = av[k] + bz, b = bv[k] + bz;
                if (t == 0) { nn[k] = imax(nn[k], c); uu[k] = imax(uu[k], a); vv[k] = imax(vv[k], b); }
                else {
                    if (c > nn[k]) { nn[k] = c; om = (om & ~(255 << (8 * k))) | (tt << (8 * k)); }
                    if (a > uu[k]) { uu[k] = a; oa = (oa & ~(255 << (8 * k))) | (tt << (8 * k)); }
                    if (b > vv[k]) { vv[k] = b; ob = (ob & ~(255 << (8 * k))) | (tt << (8 * k)); }
                }
            }
        }
        if (np > 256) return -1; // ordinals are 8 bits: such a row goes through the generic rows
        // ---- F: A[k] = Hpre[k] + k*e; in-lane inclusive prefix, then one scan pair over the lane totals ----
        int hp[C], spk[C], a1[C], a2[C], p1[C], p2[C];
        const int je1 = jb * e1, je2 = jb * e2;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            hp[k] = imax(nn[k], imax(uu[k], vv[k]));                     // Hpre
            spk[k] = nn[k] == hp[k] ? 0 : uu[k] == hp[k] ? 1 : 2;        // which of match / E1 / E2 gives it (the oracle's priority)
            a1[k] = inb[k] ? hp[k] + je1 + k * e1 : LCD_GUARD; a2[k] = inb[k] ? hp[k] + je2 + k * e2 : LCD_GUARD;
            p1[k] = k ? imax(p1[k - 1], a1[k]) : a1[k]; p2[k] = k ? imax(p2[k - 1], a2[k]) : a2[k];
        }
        int t1 = p1[C - 1], t2 = p2[C - 1];
        scan_max2(t1, t2);
        int x1 = shr1(LCD_GUARD, t1), x2 = shr1(LCD_GUARD, t2); // exclusive prefix over the lanes of this wavefront
        if (NW > 1) {
            const int buf = idx & 1;
            if (lane == 63) { sm.tot1[buf][wave] = t1; sm.tot2[buf][wave] = t2; }
            lds_barrier<NT>();
#pragma unroll
            for (int k = 0; k < NW; ++k) if (k < wave) { x1 = imax(x1, sm.tot1[buf][k]); x2 = imax(x2, sm.tot2[buf][k]); }
        }
        // ---- phase B: F, H, E-out, direction code of the cells ----
        int hh[C], ea[C], eb[C];
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int pf1 = k ? imax(x1, p1[k - 1]) : x1, pf2 = k ? imax(x2, p2[k - 1]) : x2;
            const int f1 = imax(LCD_NEG, pf1 - o1 - je1 - k * e1), f2 = imax(LCD_NEG, pf2 - o2 - je2 - k * e2);
            const int h = imax(hp[k], imax(f1, f2));
            int eo1 = imax(h - oe1, uu[k] - e1), eo2 = imax(h - oe2, vv[k] - e2);
            if (BND) { eo1 = imax(eo1, LCD_NEG); eo2 = imax(eo2, LCD_NEG); }
            const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
            const int hs = hp[k] == h ? spk[k] : fk;
            unsigned fl = 0; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
            { const int q2 = h - oe2, w2 = vv[k] - e2, q1 = h - oe1, w1 = uu[k] - e1, r2 = a2[k], r1 = a1[k];
              LCD_PUSH_GE(fl, q2, w2); LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, r2); LCD_PUSH_GT(fl, pf1, r1); }
            const unsigned cd = (unsigned)hs | (fl << 3) | (((om >> (8 * k)) & 255) ? CB_PM : 0);
            code |= cd << (8 * k);
            hh[k] = inb[k] ? h : LCD_GUARD; ea[k] = inb[k] ? eo1 : LCD_GUARD; eb[k] = inb[k] ? eo2 : LCD_GUARD;
        }
        // ---- row maximum, leftmost / rightmost column (banded rows only: w = qlen never consumes them) ----
        int ml = 0, mr = 0;
        if (BANDED) {
            // (the out-of-band cells already hold LCD_GUARD, below every real H)
            int hb = hh[0];
#pragma unroll
            for (int k = 1; k < C; ++k) hb = imax(hb, hh[k]);
            int bl = jb + C - 1, brr = jb;
#pragma unroll
            for (int k = C - 2; k >= 0; --k) bl = hh[k] == hb ? jb + k : bl;
#pragma unroll
            for (int k = 1; k < C; ++k) brr = hh[k] == hb ? jb + k : brr;
            const int wm = lane63(scan_max(hb));
            const unsigned long long mk = __ballot(hb == wm && hb > LCD_GUARD);
            int wl = 1 << 30, wr = -1;
            if (mk) {
                const int fl = __builtin_amdgcn_readfirstlane(__ffsll((long long)mk) - 1), ll = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)mk));
                wl = LCD_RL(bl, fl); wr = LCD_RL(brr, ll);
            }
            if (NW > 1) {
                if (lane == 0) { sm.bh[wave] = wm; sm.bl[wave] = wl; sm.br[wave] = wr; }
                lds_barrier<NT>();
                int rowmax = sm.bh[0];
#pragma unroll
                for (int k = 1; k < NW; ++k) rowmax = imax(rowmax, sm.bh[k]);
                ml = 1 << 30; mr = -1;
#pragma unroll
                for (int k = 0; k < NW; ++k) if (sm.bh[k] == rowmax) { ml = imin(ml, sm.bl[k]); mr = imax(mr, sm.br[k]); }
            } else { ml = wl; mr = wr; }
        }
        // ---- stores: ring slot (values), HBM (codes; values only for rows a far successor / the end node will read) ----
        const int cw4 = ((end - begc) + 4) & ~3; // cells of this row in HBM, padded to a multiple of 4 (rows stay dword-aligned for every C)
        if (cused + cw4 > code_cap || (np > 1 && oused + cw4 > ord_cap) || (spf && nsp >= spill_rows)) { wo->status = LCD_ERR_CELLS; return 0; }
        {
            const unsigned S = ring + 4 * s * SLOTW;
            lds_stc<C>(S + 4 * x, hh); lds_stc<C>(S + 4 * (WIN + x), ea); lds_stc<C>(S + 4 * (2 * WIN + x), eb);
            if (spf) { int *G = g.spill + (size_t)nsp * SLOTW; glb_stc<C>(G + x, hh); glb_stc<C>(G + WIN + x, ea); glb_stc<C>(G + 2 * WIN + x, eb); }
            if (C * tid < cw4) {
                uint8_t *cp = g.code8 + cused + C * tid;
                if constexpr (C == 4) glb_st(cp, (int)code);
                else if constexpr (C == 2) *(__attribute__((address_space(1))) unsigned short *)cp = (unsigned short)code;
                else *(__attribute__((address_space(1))) uint8_t *)cp = (uint8_t)code;
                if (np > 1) {
                    int ow[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) ow[k] = ((om >> (8 * k)) & 255) | (((oa >> (8 * k)) & 255) << 8) | (((ob >> (8 * k)) & 255) << 16);
                    glb_stc<C>(g.ord + oused + C * tid, ow);
                }
            }
        }
        if (tid == 0) {
            glb_st(g.rbeg + idx, beg); glb_st(g.rend + idx, end); glb_st(g.roff + idx, (int)cused); glb_st(g.ooff + idx, (int)oused);
            if (spf) { glb_st(g.ml + idx, ml); glb_st(g.mr + idx, mr); glb_st(g.spoff + idx, nsp); }
        }
        if (lane == s) { m_beg = beg; m_end = end; m_ml = ml; m_mr = mr; }
        cused += cw4; if (np > 1) oused += cw4; if (spf) ++nsp;
        ncell += (unsigned long long)(end - beg + 1);
        lds_barrier<NT>(); // publish the ring slot to the other wavefronts before the next row's phase A
    }
    win_sync<SOLO>();
    wo->cells = ncell;
    const long long t_bt0 = clock64();
    wo->t_dp = (unsigned long long)(t_bt0 - t_dp0);
    if (wave == 0) code_backtrack(g, sm, pd, bi, ei, qlen, SLOTW, WM, lane, CM);
    win_sync<SOLO>();
    const int n_cig = sm.bc[0];
    wo->status = sm.bc[1];
    wo->cig_pos = sm.bc[4];
    wo->score = sm.bc[5];
    win_sync<SOLO>();
    wo->t_bt = (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

// ================= lean single-wavefront rows (round 3) =================
// The rows of the single-wavefront class are ~90 % of a HiFi-shape step, and in align_windowed<64, ...> a row cost ~450 instructions for 64 - 256 cells:
// the compiler kept half of the row's uniform state (ring slot count, band, predecessor kinds) in vector registers, so uniform decisions became exec-mask
// juggling, uniform HBM reads became flat loads with per-lane 64-bit addresses, and every row took two LDS round trips (predecessor values, query bases) on
// its critical path.  This is the same recurrence (bit-identical codes / ordinals / row metadata: the code-driven backtrack is shared) written for one
// wavefront only:
//   * every uniform value lives in an SGPR by construction (readlane / readfirstlane at the source), the band arithmetic is scalar;
//   * the row plan of 64 rows is ONE packed word per row in a lane window (+ remain / interval, first two predecessors), taken by v_readlane;
//   * BACKBONE rows (one usable predecessor = the row before: 97 % of the rows of a clean-read graph) take the previous row's H / E1 / E2 from REGISTERS:
//     lanes are relative to the row's first column, so the previous row is either in the same lanes or one lane to the left (wave_shl:1 DPP) -- no LDS on
//     the critical path; the LDS ring slot is still written (fire and forget) for the rows that need it: several predecessors, a predecessor further back;
//   * query bases are prefetched one row ahead for the predicted window (same / next lane group);
//   * rbeg / rend / roff go to a 64-row lane window (v_writelane) and are flushed as three coalesced stores per 64 rows instead of four scalar stores per
//     row; rows a far successor or the end node reads (plan flag) store theirs at once.
// MODE 1: the oracle's adaptive band; MODE 2: certified intervals from the table (align_certified).  C cells per lane, window 64 * C columns.
// Returns the number of cigar entries, 0 with wo->status set, or -1 = not representable here (window too narrow / > 254 predecessors / read >= 65 536 bases).
template <int C> struct LeanT { typedef unsigned word; };   // one byte per cell of a lane: direction codes, ordinals, query bases
template <> struct LeanT<8> { typedef unsigned long long word; };
__device__ __forceinline__ int lean_wlane(const int val, const int l, int old) { // old with lane l replaced by val (both wave-uniform; the lane select goes through M0)
    asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(__builtin_amdgcn_readfirstlane(val)), "s"(__builtin_amdgcn_readfirstlane(l)) : "m0");
    return old;
}
// scalar min / max that stay on the scalar ALU: imax(imax(a, b), c) becomes v_max3_i32, which exists on the vector ALU only, and drags a whole chain of
// uniform band arithmetic (and everything derived from it: window base, offsets, loop-carried state) into VGPRs
__device__ __forceinline__ int smax(const int a, const int b) { int r; asm("s_max_i32 %0, %1, %2" : "=s"(r) : "s"(__builtin_amdgcn_readfirstlane(a)), "s"(__builtin_amdgcn_readfirstlane(b)) : "scc"); return r; }
__device__ __forceinline__ int smin(const int a, const int b) { int r; asm("s_min_i32 %0, %1, %2" : "=s"(r) : "s"(__builtin_amdgcn_readfirstlane(a)), "s"(__builtin_amdgcn_readfirstlane(b)) : "scc"); return r; }
__device__ __forceinline__ int dpp_shl1(const int old, const int v) { return __builtin_amdgcn_update_dpp(old, v, 0x130, 0xf, 0xf, false); } // lane i <- lane i + 1 (lane 63 keeps `old`)
template <int MODE, int C>
__device__ __attribute__((noinline)) int align_lean(const Ctx *gp_, const unsigned ring_, const unsigned sq1_, const unsigned pd_ /* 0xffffffff: none */, const LcdScoring sc_, const int w_,
                                                    const int bi_, const int ei_, const int rem_beg_, const uint8_t *seq_hbm_, const int qlen_, WinOut *wo_) {
    const long long t_in0 = clock64();
    constexpr int WIN = 64 * C, WM = WIN - 1, SLOTW = 3 * WIN, CM = ~(C - 1);
    constexpr int CP = C > 4 ? C : 4; // a row's cells in HBM are padded to the lanes' cell groups (and to 4: rows stay dword-aligned for every C)
    constexpr bool BANDED = MODE == 1, FIXED = MODE == 2;
    typedef typename LeanT<C>::word word;
    Smem &sm = g_smem;
    Ctx g = *usgpr(gp_);
    ctx_to_sgpr(g);
    int K = usgpr(g.ring_k);
    const unsigned ring = usgpr(ring_); unsigned sq1 = usgpr(sq1_), pd = usgpr(pd_);
    const int w = usgpr(w_), bi = usgpr(bi_), ei = usgpr(ei_), rem_beg = usgpr(rem_beg_), qlen = usgpr(qlen_);
    const uint8_t *seq_hbm = usgpr(seq_hbm_); WinOut *wo = usgpr(wo_);
    const int s_match = usgpr(sc_.match), s_mism = -usgpr(sc_.mismatch);
    const int o1 = usgpr(sc_.o1), e1 = usgpr(sc_.e1), o2 = usgpr(sc_.o2), e2 = usgpr(sc_.e2), oe1 = o1 + e1, oe2 = o2 + e2;
    const int lane = threadIdx.x & 63;
    if (qlen >= 65535) return -1; // (beg | end << 16 words)
    if (BANDED && (oe1 <= 0 || oe2 <= 0 || e1 < 0 || e2 < 0)) return -1; // (the row maximum is taken from Hpre: a horizontal gap must cost something)
    if (s_match < -32 || s_match > 31 || s_mism < -32 || s_mism > 31) return -1; // (the substitution score of a cell is a 6-bit field of a per-row scalar word, see `lut`)
    // Substitution scores: the query cache holds 6 * min(base, 4), and a row's five scores (its node's base against A C G T N, + 32) are 6-bit fields of ONE scalar
    // word picked per row: the cell's score is a single v_bfe_u32 (was two compares and two selects per cell, plus two moves of the row's match / mismatch into VGPRs)
    unsigned lut4 = 0;
    for (int q = 0; q < 5; ++q) lut4 |= 32u << (6 * q);
    unsigned lutb[4];
    for (int v = 0; v < 4; ++v) { unsigned w_ = 32u << 24; for (int q = 0; q < 4; ++q) w_ |= (unsigned)(32 + (q == v ? s_match : s_mism)) << (6 * q); lutb[v] = usgpr(w_); }
    lut4 = usgpr(lut4);
    // (the five words sit in lanes 0 - 4 of one register and a row takes its own by v_readlane: as a select chain on the scalar side the choice was three branches per row)
    int lutv = (int)lut4;
    for (int v = 0; v < 4; ++v) lutv = lean_wlane((int)lutb[v], v, lutv);
    auto lut_of = [&](const int vb) -> unsigned { return (unsigned)LCD_RL(lutv, smin(vb, 4)); };
    const int QB = (qlen + 12 + 15) & ~15;
    const bool r16 = FIXED && usgpr(g.ring16) != 0; // (16-bit ring values: lds_stc16 / lds_ldc16)
    const unsigned RB = r16 ? 2u : 4u;
    if (FIXED && r16) { // every value of this read inside the int16 range, by the scores' own bounds: best case all matches, worst case one gap over all rows or all columns
        const long long span = (long long)(ei - bi) > qlen ? (long long)(ei - bi) : qlen;
        const long long worst = 2 * ((o1 > o2 ? o1 : o2) + (long long)(e1 < e2 ? e1 : e2) * span) + (o1 + o2) + 2LL * (e1 + e2) + 64; // (no cell is worse than a gap over its rows plus a gap over its columns -- the cheaper extension wins on a long gap; E-out is one open + extend below H)
        // (best case: every base a match AND the heaviest bonus path -- every traversed edge adds ilog2(weight), which on a deep chain is worth more than the matches:
        //  cert_bztop = the largest bonus sum of a source..sink path, from the bound's node arrays)
        if ((long long)qlen * s_match + (long long)usgpr(g.cert_bztop) + 64 > 32000 || worst > 32000) return -1; // (not representable here: the generic rows keep 32-bit values)
    }
    auto ring_st3 = [&](const unsigned slot, const int x, const int (&va)[C], const int (&vb)[C], const int (&vc)[C]) { // slot: byte address of a ring slot; x: its first column here
        if (FIXED && r16) { lds_stc16<C>(slot + 2 * x, va); lds_stc16<C>(slot + 2 * (WIN + x), vb); lds_stc16<C>(slot + 2 * (2 * WIN + x), vc); }
        else { lds_stc<C>(slot + 4 * x, va); lds_stc<C>(slot + 4 * (WIN + x), vb); lds_stc<C>(slot + 4 * (2 * WIN + x), vc); }
    };
    {
        // the pool of a single-wavefront chain is laid out for the window the host expects (PoaChain.wmax columns per ring slot); a wider window moves the
        // query cache up and gives up the first-predecessor distances -- or, if the pool is too small for that, leaves the read to the next wider window
        unsigned ring_bytes = (unsigned)(K * SLOTW) * RB;
        if (sq1 < ring + ring_bytes) {
            if (ring_bytes + (unsigned)QB > (unsigned)g.pool_words * 4u && K > g.plan_k) { K = usgpr(g.plan_k); ring_bytes = (unsigned)(K * SLOTW) * RB; }
            if (ring_bytes + (unsigned)QB > (unsigned)g.pool_words * 4u) return -1;
            if (sq1 < ring + ring_bytes) { sq1 = ring + ring_bytes; pd = 0xffffffffu; wo->clobber = 1; }
        }
    }
    const int KM = K - 1;
    for (int j = lane; j < QB; j += 64) { const int qb_ = (j >= 1 && j <= qlen) ? glb_ld_u8(seq_hbm + (j - 1)) : 4; lds_st_u8(sq1 + j, 6 * (qb_ > 4 ? 4 : qb_)); } // shifted: sq1[j] = 6 * q[j-1] (the field offset into `lut`)
    const unsigned code_cap = (unsigned)(g.cell_cap > 0xfffffff0ull ? 0xfffffff0ull : g.cell_cap);
    const unsigned ord_cap = g.spill_x > 2 ? code_cap : (unsigned)((g.cell_cap / 4) > 0xfffffff0ull ? 0xfffffff0ull : (g.cell_cap / 4));
    const long long spill_rows_ll = g.cell_cap * g.spill_x > 64 ? (long long)((g.cell_cap * g.spill_x - 64) / ((unsigned long long)SLOTW * 4)) : 0;
    const int spill_rows = (int)(spill_rows_ll > 0x7fffffffll ? 0x7fffffffll : spill_rows_ll);
    const int *const hull = g.cert + 6 * (size_t)g.node_cap;
    // ---- source row (slot 0, window at column 0) ----
    int end0 = qlen - rem_beg; if (end0 < 0) end0 = 0; end0 += w; if (end0 > qlen) end0 = qlen;
    if (FIXED) { const int hw = usgpr(glb_ld(hull + bi)); end0 = hw >> 16; if ((hw & 65535) != 0) return -1; } // (the source row's interval starts at column 0)
    if (end0 + 2 > WIN) return -1;
    int nsp = 0;
    int pvh[C], pva[C], pvb[C]; // the previous row's values, lane = (column - pv_begc) / C
    const int cl = C * lane;
    // one cell per lane: the plain rows' lanes follow the diagonal, and the adaptive band stays at column 0 for its first w rows -- the source row's window starts
    // w + LCD_DIAG_SLACK columns LEFT of column 0 (lanes of negative columns are outside every band), as far as its own last column allows
    constexpr int LCD_DIAG_SLACK = 0; // (columns a window is started left of its band: nothing gained once the plain rows could also keep their window, see there)
    const int wb0 = C == 1 && LCD_DIAG_SLACK > 0 ? -smax(0, smin(LCD_DIAG_SLACK + (BANDED ? w : 0), WIN - 2 - end0)) : 0;
    {
        const bool spf = (usgpr(glb_ld_u8(g.imap + bi)) & 2) != 0;
        if (spf && spill_rows < 1) { wo->status = LCD_ERR_CELLS; return 0; }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int j = cl + k + wb0;
            const int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            const int h = j ? imax(f1, f2) : 0;
            const bool in = j >= 0 && j <= end0;
            pvh[k] = in ? h : LCD_GUARD; pva[k] = in ? h - oe1 : LCD_GUARD; pvb[k] = in ? h - oe2 : LCD_GUARD;
        }
        const int x0 = (wb0 + cl) & WM;
        ring_st3(ring, x0, pvh, pva, pvb);
        if (spf) { int *G = g.spill; glb_stc<C>(G + x0, pvh); glb_stc<C>(G + WIN + x0, pva); glb_stc<C>(G + 2 * WIN + x0, pvb); nsp = 1; }
        if (lane == 0) { glb_st(g.rbeg + bi, 0); glb_st(g.rend + bi, end0); glb_st(g.roff + bi, 0); glb_st(g.ml + bi, 0); glb_st(g.mr + bi, 0); glb_st(g.spoff + bi, 0); }
    }
    // ring slot meta: lane s holds (beg | end << 16, ml | mr << 16) of slot s; an empty slot is beg 1, end 0
    int m_be = 1, m_mm = 0;
    if (lane == 0) m_be = end0 << 16;
    int pv_begc = wb0, pv_beg = 0, pv_end = end0, pv_ml = 0, pv_mr = 0; bool pv_ok = true;
    unsigned cused = 0, oused = 0; unsigned long long ncell = (unsigned long long)end0 + 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t_dp0 = clock64();
    wo->t_setup = (unsigned long long)(t_dp0 - t_in0);
    // plan window (lane = row - wbase) and the rows' metadata window
    int w_pk = 0, w_x = 0, w_pi0 = 0, w_pi1 = 0, w_p0 = 0; // w_x: remain (MODE 1) / interval (MODE 2)
    int r_be = 1;
    unsigned cused_w0 = 0; // code offset of the metadata window's first row: the rows' offsets are made from it and the rows' intervals when the window is flushed
    int wbase = bi + 1;
    unsigned long long np_mask = ~0ull; // rows of the plan window that are NOT plain by their plan word (several / far / no predecessors, spilled, unreachable, past the end)
    // packed plan word: #preds (8 bits, 255 = more) | base << 8 | spill << 11 | unreachable << 12 | backbone << 13 | bonus0 << 14 | bonus1 << 19
    auto load_plan = [&](const int base) {
        const int ri = base + lane;
        w_pk = 1 << 12;
        if (ri < ei) {
            const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
            const int cnt = s1 - s0;
            int b0 = 0, b1 = 0;
            w_p0 = s0;
            if (cnt > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); b0 = glb_ld(g.pl_bonus + s0); }
            if (cnt > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); b1 = glb_ld(g.pl_bonus + s0 + 1); }
            const int rem = glb_ld(g.pl_rem + ri);
            w_x = FIXED ? glb_ld(hull + ri) : rem;
            w_pk = imin(cnt, 255) | (glb_ld_u8(g.pl_base + ri) << 8) | ((glb_ld_u8(g.imap + ri) & 2) << 10) | (rem == (1 << 30) ? 1 << 12 : 0)
                 | ((cnt == 1 && w_pi0 == ri - 1) ? 1 << 13 : 0) | (b0 << 14) | (b1 << 19);
        }
        LCD_PIN(w_pk); LCD_PIN(w_x); LCD_PIN(w_pi0); LCD_PIN(w_pi1); LCD_PIN(w_p0);
        np_mask = __ballot((w_pk & 0x38ff) != 0x2001);
        // rows whose ring slot a general row may read: one of the next K rows is not plain by its plan word (the last K rows of a window: always)
        unsigned long long km = 0;
        for (int d = 1; d <= K; ++d) km |= (np_mask >> d) | (1ull << (64 - d));
        w_pk |= (int)((km >> lane) & 1ull) << 24; // (bit 24 of the plan word: keep this row's ring slot)
    };
    auto flush_meta = [&](const int base, const int n) { // rows base .. base + n - 1
        // (a row's cells follow the previous row's in HBM, padded to a multiple of four: its offset is a prefix sum over the window's intervals -- one scan per 64 rows
        //  instead of a lane write per row)
        const int rb_ = r_be & 65535, re_ = (int)((unsigned)r_be >> 16);
        const int cw_ = (lane < n && rb_ <= re_) ? (((re_ - (C == 1 ? rb_ : (rb_ & CM))) + CP) & ~(CP - 1)) : 0;
        const int off_ = (int)cused_w0 + scan_add(cw_) - cw_;
        if (lane < n) { glb_st(g.rbeg + base + lane, rb_); glb_st(g.rend + base + lane, re_); glb_st(g.roff + base + lane, off_); }
    };
    load_plan(wbase);
    // query bases of the lanes' cells for the window that starts at column b: sq1[b + cl + k], k < C, in the low bytes of one word
    // (one LDS load of exactly C bytes -- jq is a multiple of C -- so that a prefetched word needs no arithmetic before the row that uses it)
    auto q_of = [&](const int b) {
        // (no clamp: columns past the cache belong to lanes past the read's end -- never inside a band, their cells are masked -- and an LDS read past the
        //  workgroup's allocation returns 0)
        const unsigned a = (unsigned)cl + (sq1 + (unsigned)b);
        if constexpr (C == 1) return (word)*(const lcd_lds_u8 *)(uintptr_t)a;
        else if constexpr (C == 2) return (word)*(const __attribute__((address_space(3))) unsigned short *)(uintptr_t)a;
        else if constexpr (C == 4) return (word)(unsigned)lds_ld(a);
        else return *(const __attribute__((address_space(3))) unsigned long long *)(uintptr_t)a;
    };
    word q_cur = 0, q_nxt = 0; int q_ok = 0; // windows at pv_begc and pv_begc + C, valid while q_ok (the run loop keeps them; a general row drops them)
    // per-lane constants: (cl + k) * e and -(o + (cl + k) * e)
    int bcle1 = cl * e1 + (1 << 29), bcle2 = cl * e2 + (1 << 29), nbcle1 = -bcle1 - (C == 1 ? o1 : 0), nbcle2 = -bcle2 - (C == 1 ? o2 : 0); // (the plain rows' biased gap prefixes)
    if constexpr (C <= 2) { LCD_PIN(bcle1); LCD_PIN(bcle2); LCD_PIN(nbcle1); LCD_PIN(nbcle2); } // (opaque: the compiler would re-derive them from cl * e and add the bias in an instruction of its own; the wider variants have no registers to spare for that)
    int idx = bi + 1;
    // (statistics builds, tools/ab_build.sh <name> -DLCD_X_ROWSTAT / -DLCD_X_PLANSTAT: plain / general row counts and the kinds of graph change per read through the
    //  LCD_CHAIN_TIMES dump's t_setup / t_bt columns; ticks of the plan-window refreshes through t_plan.  Never defined in the product build.)
#ifdef LCD_X_PLANSTAT
    unsigned long long t_plan_ = 0;
#endif
#ifdef LCD_X_ROWSTAT
    unsigned n_pl_ = 0, n_gn_ = 0;
#endif
    while (idx < ei) {
        if (idx - wbase == 64) {
#ifdef LCD_X_PLANSTAT
            const long long tp0_ = clock64();
#endif
            flush_meta(wbase, 64); wbase = idx; cused_w0 = cused; load_plan(wbase);
#ifdef LCD_X_PLANSTAT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); t_plan_ += (unsigned long long)(clock64() - tp0_);
#endif
            if ((unsigned long long)clock64() > g.wd_deadline) { wo->status = LCD_ERR_WATCHDOG; return 0; }
        }
        int wk = idx - wbase;
        // ===== a run of PLAIN rows: reachable, one usable predecessor = the row before (whose values are in registers), not spilled, window reachable from the
        // previous row's lanes.  Anything else leaves the loop with the row untouched and is handled by the general row below. =====
        // No clamp to LCD_NEG on the way: every value a cell of a plain row can win with is a real one (the row before has a reachable cell on the cell's diagonal or
        // above it, or the cell is reached from its left), fillers stay a factor of two below LCD_NEG whatever a row adds to them, and the E values that go on to the
        // next row keep their floor (max3).  What differs from the clamped rows is the code of cells NO alignment reaches -- never read.
        // The ring slot (and its lane of slot metadata) is written only for rows a general row will look at: one of the next K rows is not plain by its plan word
        // (bit 24 of the plan word; a row that stops being plain at run time has one predecessor, the row before: flushed from the registers when the run ends).
        if (pv_ok) {
#ifdef LCD_X_ROWSTAT
            const int idx0_ = idx;
#endif
            unsigned run_cells = 0;
            // (the run's loop-carried scalars are locals initialised through readfirstlane: as phis of the outer loop the compiler keeps them in VGPRs)
            int l_begc = usgpr(pv_begc), l_beg = usgpr(pv_beg), l_end = usgpr(pv_end), l_ml = usgpr(pv_ml), l_mr = usgpr(pv_mr);
            int pend = 0; // the last row of the run has not been stored to its ring slot
            unsigned l_cused = usgpr(cused); // (as a phi of the outer loop the code offset sat in a vector register, and the capacity test was a 64-bit vector compare)
            if constexpr (C == 1) {
                int q_c1 = lds_ld_u8((unsigned)lane + (sq1 + (unsigned)l_begc)), q_n1 = lds_ld_u8((unsigned)lane + (sq1 + (unsigned)(l_begc + 1))); // query bytes (x 6) of the lanes' columns in the window at hand / one column on
                // ---- one cell per lane: lanes follow the DIAGONAL.  The window moves one column per row whatever the band does, so the cell's diagonal neighbour
                // is the lane's own previous value, the cell above it one lane to the right (two in-place DPP moves, unconditionally: no branch on how the window
                // moved, no register copies where two branches join), and the band [beg, end] drifts inside the 64 lanes (the general rows leave LCD_DIAG_SLACK
                // columns to its left; a band that reaches a window edge ends the run).  Codes in HBM still start at the row's first column. ----
                while (wk < 64) {
                    const int pk = LCD_RL(w_pk, wk);
                    if ((pk & 0x38ff) != 0x2001) break; // #preds == 1, not spilled, reachable, backbone
                    const int xw = LCD_RL(w_x, wk);
                    int beg, end;
                    if (BANDED) {
                        const int diag = qlen - xw;
                        beg = smax(smax(smin(l_ml + 1, diag) - w, 0), l_beg);
                        end = smin(smin(smax(l_mr + 1, diag) + w, qlen), l_end + 1);
                    } else { beg = xw & 65535; end = (int)((unsigned)xw >> 16); }
                    const int adv = smin(beg - l_begc, 1); // (0 or 1: beg >= l_beg >= l_begc) the window follows the band's first column by one column per row (a band that jumps further leaves lanes idle on its left)
                    const int wb = l_begc + adv;        // this row's window: lane = column - wb
                    const int lo = beg - wb, span = end - beg;
                    const int cw4 = (span + CP) & ~(CP - 1);
                    // (one test: each of the four quantities is negative exactly when its condition fails -- empty band; an interval of MODE 2 that starts left of the
                    //  window; band past the window's last lane; code capacity, which is below 2^32 - 16 so the difference fits a signed compare after the shift)
                    if (((end - beg) | (FIXED ? beg - l_begc : 0) | (WIN - 2 - (end - wb)) | (l_cused + (unsigned)cw4 > code_cap ? -1 : 0)) < 0) break; // (the adaptive band never starts left of the row before)
                    const int vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
                    const int s = (idx - bi) & KM;
                    const unsigned lut = lut_of(vb);
                    const int bzb = bz0 - 32; // (the score fields of `lut` are biased by 32)
                    // Hand-over from the row before WITHOUT a branch (two branches that join cost a register copy per value) and without touching the execution mask:
                    // v_cndmask with a DPP source, the condition a scalar mask that is all ones or all zeros.  Window moved (adv): the cell above is one lane to the
                    // right -- E values shift left in place (lane 63 keeps its own old values: its column is never inside a band, end - wb <= WIN - 2, and is
                    // masked) -- the diagonal neighbour is the lane's own H, and the query byte is the one prefetched for the next column.  Window stayed: E values
                    // and query byte stay, H shifts right in place and lane 0 (the column left of the window) gets the filler.
                    asm volatile("s_cmp_lg_u32 %5, 0\n\t"
                                 "s_cselect_b64 vcc, 0, -1\n\ts_nop 1\n\t"                                               // vcc = stayed ? all lanes : none;  D = vcc ? src1 : src0
                                 "v_cndmask_b32_dpp %0, %0, %0, vcc wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_cndmask_b32_dpp %1, %1, %1, vcc wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_cndmask_b32_e32 %3, %4, %3, vcc\n\t"
                                 "s_cselect_b64 vcc, -1, 0\n\ts_nop 1\n\t"                                               // vcc = moved ? all lanes : none
                                 "v_cndmask_b32_dpp %2, %2, %2, vcc wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                 "s_cselect_b64 vcc, 0, 1\n\ts_nop 1\n\t"                                                // vcc = stayed ? lane 0 : none
                                 "v_cndmask_b32_e64 %2, %2, -2.0, vcc"                                                       // (-2.0f is LCD_GUARD's bit pattern)
                                 : "+v"(pva[0]), "+v"(pvb[0]), "+v"(pvh[0]), "+v"(q_c1) : "v"(q_n1), "s"(usgpr(adv)) : "vcc", "scc");
                    static_assert(LCD_GUARD == (int)0xc0000000, "the filler above is written as the inline constant -2.0f");
                    const unsigned q6 = (unsigned)q_c1;
                    q_n1 = lds_ld_u8((unsigned)lane + (sq1 + (unsigned)(wb + 1))); // the next row's byte if its window moves again (the usual case): a row of latency to arrive in
                    const int sk = (int)__builtin_amdgcn_ubfe(lut, q6, 6u);
                    const int nn = pvh[0] + sk + bzb;
                    const int uu = pva[0] + bz0, vv = pvb[0] + bz0;
                    const bool inb = (unsigned)(lane - lo) <= (unsigned)span;
                    const int m_ = imax(uu, vv);
                    int spk = nn >= m_ ? 0 : uu >= vv ? 1 : 2;
                    LCD_PIN(spk); // (made here: left to the code store at the end it keeps the old E values alive past their successors)
                    const int hp = inb ? imax(nn, m_) : LCD_GUARD; // masked once: the gap prefixes, the row maximum and H all take it from here
                    const int a1 = hp + bcle1, a2 = hp + bcle2; // (relative to the window's first column: the row's own offset cancels in every comparison; + 2^29, see x1)
                    int t1 = a1, t2 = a2;
                    int ml = 0, mr = 0;
                    if (BANDED) { // the row maximum and its leftmost / rightmost column, from Hpre (see scan_max3), in the same scan as the gap prefixes
                        int hbs = hp;
                        scan_max3(t1, t2, hbs);
                        const int wm = lane63(hbs);
                        const unsigned long long mk = __ballot(hp == wm); // (the row is not empty: its in-band cells are real values > the filler)
                        ml = wb + usgpr((int)__builtin_ctzll(mk)); mr = wb + usgpr(63 - (int)__builtin_clzll(mk));
                    } else scan_max2(t1, t2);
                    // exclusive prefix over the lanes.  The prefixes carry a bias of 2^29 (every real one is positive, a masked one negative), so the 0 that lane 0
                    // gets from the shift's bound control stands for "nothing to the left" without a register set up for it
                    const int pf1 = __builtin_amdgcn_update_dpp(0, t1, 0x138, 0xf, 0xf, true), pf2 = __builtin_amdgcn_update_dpp(0, t2, 0x138, 0xf, 0xf, true);
                    const int f1 = pf1 + nbcle1, f2 = pf2 + nbcle2; // (the opening penalty is in the lane's constant)
                    const int fm = imax(f1, f2);
                    const int h = imax(hp, fm);
                    const int q1 = h - oe1, w1 = uu - e1, q2 = h - oe2, w2 = vv - e2;
                    const int eo1 = imax(imax(q1, w1), LCD_NEG), eo2 = imax(imax(q2, w2), LCD_NEG);
                    const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
                    const int hs = hp >= fm ? spk : fk;
                    unsigned fl = q2 >= w2 ? 1u : 0u; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
                    LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, a2); LCD_PUSH_GT(fl, pf1, a1);
                    const unsigned code = (unsigned)hs | (fl << 3);
                    pvh[0] = inb ? h : LCD_GUARD; pva[0] = inb ? eo1 : LCD_GUARD; pvb[0] = inb ? eo2 : LCD_GUARD;
                    const int be = beg | (end << 16);
                    if (pk & (1 << 24)) {
                        ring_st3(ring + RB * (unsigned)(s * SLOTW), (wb + lane) & WM, pvh, pva, pvb);
                        m_be = lean_wlane(be, s, m_be); m_mm = lean_wlane(ml | (mr << 16), s, m_mm);
                        pend = 0;
                    } else pend = 1;
                    if (inb) *(__attribute__((address_space(1))) uint8_t *)(g.code8 + (size_t)(l_cused + (unsigned)(lane - lo))) = (uint8_t)code;
                    r_be = lean_wlane(be, wk, r_be);
                    l_begc = wb; l_beg = beg; l_end = end; l_ml = ml; l_mr = mr;
                    l_cused += (unsigned)cw4; run_cells += (unsigned)(span + 1);
                    ++idx; ++wk;
                }
            } else if constexpr (C >= 8) {
                // ---- eight cells per lane (intervals of 257 - 508 columns: a handful of reads per submission, but the longest ones): the rows as they were before
                // round 5 -- clamped, every row to its ring slot.  The leaner formulation below needs more registers at this width than the class has (30 scratch
                // accesses per row against 2) ----
                if (!q_ok) { q_cur = q_of(pv_begc); q_nxt = q_of(pv_begc + C); q_ok = 1; }
                const int cle1 = cl * e1, cle2 = cl * e2, ncle1 = -cle1, ncle2 = -cle2; // (the cells' k * e and the window's begc * e are added on the scalar side)
                while (wk < 64) {
                    const int pk = LCD_RL(w_pk, wk);
                    if ((pk & 0x38ff) != 0x2001) break; // #preds == 1, not spilled, reachable, backbone
                    const int xw = LCD_RL(w_x, wk);
                    int beg, end;
                    if (BANDED) {
                        const int diag = qlen - xw;
                        beg = smax(smax(smin(l_ml + 1, diag) - w, 0), l_beg);
                        end = smin(smin(smax(l_mr + 1, diag) + w, qlen), l_end + 1);
                    } else { beg = xw & 65535; end = (int)((unsigned)xw >> 16); }
                    const int begc = beg & CM;
                    const int sh = begc - l_begc;
                    const int cw4 = ((end - begc) + CP) & ~(CP - 1);
                    if (((end - beg) | (WIN - 2 - (end - begc)) | ((unsigned)sh > (unsigned)C ? -1 : 0) | (l_cused + (unsigned)cw4 > code_cap ? -1 : 0)) < 0) break; // (one test, as above)
                    const int vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
                    const int s = (idx - bi) & KM;
                    const int mt = vb >= 4 ? 0 : s_match, mm = vb >= 4 ? 0 : s_mism;
                    // previous row: same lanes (sh == 0) or one lane to the left (sh == C)
                    int hm, av[C], bv[C], hd[C]; // hd[k]: H of the previous row at this cell's column - 1
                    if (sh == 0) {
                        hm = shr1(LCD_GUARD, pvh[C - 1]);
    #pragma unroll
                        for (int k = 0; k < C; ++k) { hd[k] = k ? pvh[k - 1] : hm; av[k] = pva[k]; bv[k] = pvb[k]; }
                    } else {
                        q_cur = q_nxt; q_nxt = q_of(begc + C);
    #pragma unroll
                        for (int k = 0; k < C; ++k) { hd[k] = k ? dpp_shl1(LCD_GUARD, pvh[k - 1]) : pvh[C - 1]; av[k] = dpp_shl1(LCD_GUARD, pva[k]); bv[k] = dpp_shl1(LCD_GUARD, pvb[k]); }
                    }
                    const int be1 = begc * e1, be2 = begc * e2, lo = beg - begc, span = end - beg;
                    int hp[C], spk[C], a1[C], a2[C], p1[C], p2[C], uu[C], vv[C]; bool inb[C];
    #pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int q = (int)(q_cur >> (8 * k)) & 255; // (6 * base)
                        const int sk = q >= 24 ? 0 : (q == 6 * vb ? mt : mm);
                        const int nn = imax(LCD_NEG, hd[k] + sk + bz0);
                        uu[k] = imax(LCD_NEG, av[k] + bz0); vv[k] = imax(LCD_NEG, bv[k] + bz0);
                        inb[k] = (unsigned)(cl + k - lo) <= (unsigned)span;
                        hp[k] = imax(nn, imax(uu[k], vv[k]));
                        spk[k] = nn == hp[k] ? 0 : uu[k] == hp[k] ? 1 : 2;
                        a1[k] = inb[k] ? hp[k] + cle1 + (be1 + k * e1) : LCD_GUARD; a2[k] = inb[k] ? hp[k] + cle2 + (be2 + k * e2) : LCD_GUARD;
                        p1[k] = k ? imax(p1[k - 1], a1[k]) : a1[k]; p2[k] = k ? imax(p2[k - 1], a2[k]) : a2[k];
                    }
                    int t1 = p1[C - 1], t2 = p2[C - 1];
                    int ml = 0, mr = 0;
                    if (BANDED) { // the row maximum and its leftmost / rightmost column, from Hpre (see scan_max3), in the same scan as the gap prefixes
                        int hm_[C];
    #pragma unroll
                        for (int k = 0; k < C; ++k) hm_[k] = inb[k] ? hp[k] : LCD_GUARD;
                        int hb = hm_[0];
    #pragma unroll
                        for (int k = 1; k < C; ++k) hb = imax(hb, hm_[k]);
                        int hbs = hb;
                        scan_max3(t1, t2, hbs);
                        const int wm = lane63(hbs);
                        const unsigned long long mk = __ballot(hb == wm); // (the row is not empty: its in-band cells are >= LCD_NEG > the filler)
                        const int fl = usgpr((int)__builtin_ctzll(mk)), ll = usgpr(63 - (int)__builtin_clzll(mk));
                        if (C == 1) { ml = begc + fl; mr = begc + ll; }
                        else {
                            int bl = C - 1, brr = 0;
    #pragma unroll
                            for (int k = C - 2; k >= 0; --k) bl = hm_[k] == hb ? k : bl;
    #pragma unroll
                            for (int k = 1; k < C; ++k) brr = hm_[k] == hb ? k : brr;
                            ml = begc + C * fl + LCD_RL(bl, fl); mr = begc + C * ll + LCD_RL(brr, ll);
                        }
                    } else scan_max2(t1, t2);
                    const int x1 = shr1(LCD_GUARD, t1), x2 = shr1(LCD_GUARD, t2);
                    word code = 0;
    #pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int pf1 = k ? imax(x1, p1[k - 1]) : x1, pf2 = k ? imax(x2, p2[k - 1]) : x2;
                        const int f1 = imax(LCD_NEG, pf1 + ncle1 - (o1 + be1 + k * e1)), f2 = imax(LCD_NEG, pf2 + ncle2 - (o2 + be2 + k * e2));
                        const int h = imax(hp[k], imax(f1, f2));
                        const int q1 = h - oe1, w1 = uu[k] - e1, q2 = h - oe2, w2 = vv[k] - e2;
                        const int eo1 = imax(imax(q1, w1), LCD_NEG), eo2 = imax(imax(q2, w2), LCD_NEG);
                        const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
                        const int hs = hp[k] == h ? spk[k] : fk;
                        unsigned fl = 0; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
                        { const int r2 = a2[k], r1 = a1[k]; LCD_PUSH_GE(fl, q2, w2); LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, r2); LCD_PUSH_GT(fl, pf1, r1); }
                        code |= (word)((unsigned)hs | (fl << 3)) << (8 * k);
                        pvh[k] = inb[k] ? h : LCD_GUARD; pva[k] = inb[k] ? eo1 : LCD_GUARD; pvb[k] = inb[k] ? eo2 : LCD_GUARD;
                    }
                    {
                        const int x = (begc + cl) & WM;
                        ring_st3(ring + RB * (unsigned)(s * SLOTW), x, pvh, pva, pvb);
                        if (cl < cw4) {
                            uint8_t *cp = g.code8 + (size_t)(l_cused + (unsigned)cl);
                            if constexpr (C == 8) *(__attribute__((address_space(1))) unsigned long long *)cp = code;
                            else if constexpr (C == 4) glb_st(cp, (int)code);
                            else if constexpr (C == 2) *(__attribute__((address_space(1))) unsigned short *)cp = (unsigned short)code;
                            else *(__attribute__((address_space(1))) uint8_t *)cp = (uint8_t)code;
                        }
                    }
                    const int be = beg | (end << 16);
                    r_be = lean_wlane(be, wk, r_be);
                    m_be = lean_wlane(be, s, m_be); m_mm = lean_wlane(ml | (mr << 16), s, m_mm);
                    l_begc = begc; l_beg = beg; l_end = end; l_ml = ml; l_mr = mr;
                    l_cused += (unsigned)cw4; run_cells += (unsigned)(span + 1);
                    ++idx; ++wk;
                }
            } else {
                if (!q_ok) { q_cur = q_of(pv_begc); q_nxt = q_of(pv_begc + C); q_ok = 1; }
                while (wk < 64) {
                    const int pk = LCD_RL(w_pk, wk);
                    if ((pk & 0x38ff) != 0x2001) break; // #preds == 1, not spilled, reachable, backbone
                    const int xw = LCD_RL(w_x, wk);
                    int beg, end;
                    if (BANDED) {
                        const int diag = qlen - xw;
                        beg = smax(smax(smin(l_ml + 1, diag) - w, 0), l_beg);
                        end = smin(smin(smax(l_mr + 1, diag) + w, qlen), l_end + 1);
                    } else { beg = xw & 65535; end = (int)((unsigned)xw >> 16); }
                    const int begc = beg & CM;
                    const int sh = begc - l_begc;
                    const int cw4 = ((end - begc) + CP) & ~(CP - 1);
                    if (((end - beg) | (WIN - 2 - (end - begc)) | ((unsigned)sh > (unsigned)C ? -1 : 0) | (l_cused + (unsigned)cw4 > code_cap ? -1 : 0)) < 0) break; // (one test, as above)
                    const int vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
                    const int s = (idx - bi) & KM;
                    const unsigned lut = lut_of(vb);
                    const int bzb = bz0 - 32; // (the score fields of `lut` are biased by 32)
                    // previous row: same lanes (sh == 0) or one lane to the left (sh == C); hd[k]: H of the previous row at this cell's column - 1
                    int hd[C];
                    if (sh == 0) {
                        hd[0] = shr1(LCD_GUARD, pvh[C - 1]);
#pragma unroll
                        for (int k = 1; k < C; ++k) hd[k] = pvh[k - 1];
                    } else {
                        q_cur = q_nxt; q_nxt = q_of(begc + C);
                        hd[0] = pvh[C - 1];
#pragma unroll
                        for (int k = 1; k < C; ++k) hd[k] = dpp_shl1(LCD_GUARD, pvh[k - 1]);
#pragma unroll
                        for (int k = 0; k < C; ++k) { pva[k] = dpp_shl1(LCD_GUARD, pva[k]); pvb[k] = dpp_shl1(LCD_GUARD, pvb[k]); }
                    }
                    const int lo = beg - begc, span = end - beg;
                    int hp[C], spk[C], a1[C], a2[C], p1[C], p2[C], uu[C], vv[C]; bool inb[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const unsigned q6 = (unsigned)(q_cur >> (8 * k)) & 255u;
                        const int sk = (int)__builtin_amdgcn_ubfe(lut, q6, 6u);
                        const int nn = hd[k] + sk + bzb;
                        uu[k] = pva[k] + bz0; vv[k] = pvb[k] + bz0;
                        inb[k] = (unsigned)(cl + k - lo) <= (unsigned)span;
                        const int m_ = imax(uu[k], vv[k]);
                        spk[k] = nn >= m_ ? 0 : uu[k] >= vv[k] ? 1 : 2;
                        hp[k] = inb[k] ? imax(nn, m_) : LCD_GUARD; // masked once: the gap prefixes, the row maximum and H all take it from here
                        a1[k] = hp[k] + bcle1 + k * e1; a2[k] = hp[k] + bcle2 + k * e2; // (relative to the window's first column; + 2^29, see x1)
                        p1[k] = k ? imax(p1[k - 1], a1[k]) : a1[k]; p2[k] = k ? imax(p2[k - 1], a2[k]) : a2[k];
                    }
                    int t1 = p1[C - 1], t2 = p2[C - 1];
                    int ml = 0, mr = 0;
                    if (BANDED) { // the row maximum and its leftmost / rightmost column, from Hpre (see scan_max3), in the same scan as the gap prefixes
                        int hb = hp[0];
#pragma unroll
                        for (int k = 1; k < C; ++k) hb = imax(hb, hp[k]);
                        int hbs = hb;
                        scan_max3(t1, t2, hbs);
                        const int wm = lane63(hbs);
                        const unsigned long long mk = __ballot(hb == wm); // (the row is not empty: its in-band cells are real values > the filler)
                        const int fl = usgpr((int)__builtin_ctzll(mk)), ll = usgpr(63 - (int)__builtin_clzll(mk));
                        int bl = C - 1, brr = 0;
#pragma unroll
                        for (int k = C - 2; k >= 0; --k) bl = hp[k] == hb ? k : bl;
#pragma unroll
                        for (int k = 1; k < C; ++k) brr = hp[k] == hb ? k : brr;
                        ml = begc + C * fl + LCD_RL(bl, fl); mr = begc + C * ll + LCD_RL(brr, ll);
                    } else scan_max2(t1, t2);
                    // exclusive prefix over the lanes.  The prefixes carry a bias of 2^29 (every real one is positive, a masked one negative), so the 0 that lane 0
                    // gets from the shift's bound control stands for "nothing to the left" without a register set up for it
                    const int x1 = __builtin_amdgcn_update_dpp(0, t1, 0x138, 0xf, 0xf, true), x2 = __builtin_amdgcn_update_dpp(0, t2, 0x138, 0xf, 0xf, true);
                    word code = 0;
#pragma unroll
                    for (int k = 0; k < C; ++k) {
                        const int pf1 = k ? imax(x1, p1[k - 1]) : x1, pf2 = k ? imax(x2, p2[k - 1]) : x2;
                        const int sn1 = -(o1 + k * e1), sn2 = -(o2 + k * e2);
                        const int f1 = pf1 + nbcle1 + sn1, f2 = pf2 + nbcle2 + sn2;
                        const int fm = imax(f1, f2);
                        const int h = imax(hp[k], fm);
                        const int q1 = h - oe1, w1 = uu[k] - e1, q2 = h - oe2, w2 = vv[k] - e2;
                        const int eo1 = imax(imax(q1, w1), LCD_NEG), eo2 = imax(imax(q2, w2), LCD_NEG);
                        const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
                        const int hs = hp[k] >= fm ? spk[k] : fk;
                        unsigned fl = q2 >= w2 ? 1u : 0u; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
                        { // (eight cells per lane: the cells' own prefix terms are made again here instead of being kept over the scan -- sixteen registers the class does not have)
                          int r2 = a2[k], r1 = a1[k];
                          if constexpr (C >= 8) { int hq = hp[k]; LCD_PIN(hq); r2 = hq + bcle2 + k * e2; r1 = hq + bcle1 + k * e1; }
                          LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, r2); LCD_PUSH_GT(fl, pf1, r1); }
                        code |= (word)((unsigned)hs | (fl << 3)) << (8 * k);
                        pvh[k] = inb[k] ? h : LCD_GUARD; pva[k] = inb[k] ? eo1 : LCD_GUARD; pvb[k] = inb[k] ? eo2 : LCD_GUARD;
                    }
                    const int be = beg | (end << 16);
                    if (pk & (1 << 24)) {
                        ring_st3(ring + RB * (unsigned)(s * SLOTW), (begc + cl) & WM, pvh, pva, pvb);
                        m_be = lean_wlane(be, s, m_be); m_mm = lean_wlane(ml | (mr << 16), s, m_mm);
                        pend = 0;
                    } else pend = 1;
                    if (cl < cw4) {
                        uint8_t *cp = g.code8 + (size_t)(l_cused + (unsigned)cl);
                        if constexpr (C == 8) *(__attribute__((address_space(1))) unsigned long long *)cp = code;
                        else if constexpr (C == 4) glb_st(cp, (int)code);
                        else *(__attribute__((address_space(1))) unsigned short *)cp = (unsigned short)code;
                    }
                    r_be = lean_wlane(be, wk, r_be);
                    l_begc = begc; l_beg = beg; l_end = end; l_ml = ml; l_mr = mr;
                    l_cused += (unsigned)cw4; run_cells += (unsigned)(span + 1);
                    ++idx; ++wk;
                }
            }
            if (pend) { // a general row comes next (or the window ends): it reads its predecessor from the ring
                const int sl = (idx - 1 - bi) & KM;
                ring_st3(ring + RB * (unsigned)(sl * SLOTW), (l_begc + cl) & WM, pvh, pva, pvb);
                m_be = lean_wlane(l_beg | (l_end << 16), sl, m_be); m_mm = lean_wlane(l_ml | (l_mr << 16), sl, m_mm);
            }
            pv_begc = l_begc; pv_beg = l_beg; pv_end = l_end; pv_ml = l_ml; pv_mr = l_mr; cused = l_cused;
#ifdef LCD_X_ROWSTAT
            n_pl_ += (unsigned)(idx - idx0_);
#endif
            ncell += run_cells;
            if (wk == 64 || idx >= ei) continue;
        }
        // ===== general row =====
        q_ok = 0;
        do {
        const int pk = LCD_RL(w_pk, wk);
        const int s = (idx - bi) & KM;
        if (pk & (1 << 12)) { // not reachable from the begin node
            m_be = lean_wlane(1, s, m_be); r_be = lean_wlane(1, wk, r_be);
            pv_ok = false;
            break;
        }
        const int np = pk & 255, vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
        const bool spf = (pk >> 11) & 1;
        if (np == 255) return -1;
        const int xw = LCD_RL(w_x, wk);
        int pi0 = LCD_RL(w_pi0, wk), pi1 = 0, bz1 = 0, p0 = 0;
        if (np > 1) { pi1 = LCD_RL(w_pi1, wk); bz1 = (pk >> 19) & 31; p0 = LCD_RL(w_p0, wk); }
        bool synced = false;
        int beg, end;
        if (BANDED) { // band: pulled from the predecessors' row-max columns (same values the oracle pushes to successors)
            int mplv = 1 << 30, mprv = 0, minpb = 1 << 30, maxpe = -1;
            for (int t = 0; t < np; ++t) {
                int pi = t == 0 ? pi0 : pi1;
                if (t > 1) pi = usgpr(glb_ld(g.pl_pidx + p0 + t));
                int pb, pe, pml, pmr;
                if (idx - pi <= K) { const int sp = (pi - bi) & KM; const int be = LCD_RL(m_be, sp), mm = LCD_RL(m_mm, sp); pb = be & 65535; pe = (int)((unsigned)be >> 16); pml = mm & 65535; pmr = (int)((unsigned)mm >> 16); }
                else {
                    if (!synced) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); synced = true; } // far row: its metadata / spilled values were stored to HBM earlier
                    pb = usgpr(glb_ld(g.rbeg + pi)); pe = usgpr(glb_ld(g.rend + pi)); pml = usgpr(glb_ld(g.ml + pi)); pmr = usgpr(glb_ld(g.mr + pi));
                }
                if (pb > pe) continue;
                minpb = smin(minpb, pb); maxpe = smax(maxpe, pe);
                mplv = smin(mplv, pml + 1); mprv = smax(mprv, pmr + 1);
            }
            const int diag = qlen - xw;
            beg = smax(smax(smin(mplv, diag) - w, 0), minpb);
            end = smin(smin(smax(mprv, diag) + w, qlen), maxpe + 1);
        } else { beg = xw & 65535; end = (int)((unsigned)xw >> 16); } // the row's certified interval (lo > hi: no cell of the row can lie on an optimal path)
        if (beg > end) { // empty row
            m_be = lean_wlane(1, s, m_be); r_be = lean_wlane(1, wk, r_be);
            if (spf && lane == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
            pv_ok = false;
            break;
        }
        if (C == 1 && end - beg + 2 > WIN) return -1;
        // (one cell per lane: the window starts LCD_DIAG_SLACK columns left of the band, as far as the band's width allows -- the plain rows that follow move their
        //  window one column per row, and a band that lags behind the diagonal needs room on its left; cells in HBM start at the band's first column all the same)
        const int begc = C == 1 ? smax(beg - LCD_DIAG_SLACK, end + 2 - WIN) : (beg & CM);
        if (end - begc + 2 > WIN) return -1;
        const int lo_g = C == 1 ? beg - begc : 0; // the band's first lane
        const int jb = begc + cl;
        const word qw = q_of(begc);
        const unsigned lutg = lut_of(vb);
        bool inb[C]; int sk[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            inb[k] = jb + k >= beg && jb + k <= end;
            sk[k] = (int)__builtin_amdgcn_ubfe(lutg, (unsigned)(qw >> (8 * k)) & 255u, 6u) - 32;
        }
        // ---- phase A: best match / E1 / E2 input of the cells over the predecessors (first maximum keeps its ordinal) ----
        int nn[C], uu[C], vv[C];
        word om = 0, oa = 0, ob = 0; // ordinals, one byte per cell
        {
#pragma unroll
            for (int k = 0; k < C; ++k) { nn[k] = LCD_NEG; uu[k] = LCD_NEG; vv[k] = LCD_NEG; }
            const int x = jb & WM, xm = (jb - 1) & WM;
            for (int t = 0; t < np; ++t) {
                int pi = t == 0 ? pi0 : pi1, bz = t == 0 ? bz0 : bz1;
                if (t > 1) { pi = usgpr(glb_ld(g.pl_pidx + p0 + t)); bz = usgpr(glb_ld(g.pl_bonus + p0 + t)); }
                const bool near = idx - pi <= K;
                const int sp = (pi - bi) & KM;
                int pb, pe;
                if (near) { const int be = LCD_RL(m_be, sp); pb = be & 65535; pe = (int)((unsigned)be >> 16); }
                else {
                    if (!synced) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); synced = true; }
                    pb = usgpr(glb_ld(g.rbeg + pi)); pe = usgpr(glb_ld(g.rend + pi));
                }
                if (pb > pe) continue;
                // a ring slot is addressed by (column mod WIN): columns of this row that lie a full window away from the predecessor's band would read that
                // band's values instead of the filler -- rare, so those rows mask the loaded values by the predecessor's [beg, end] explicitly
                const bool risk = pe - beg + 2 >= WIN || end - pb + 1 >= WIN;
                int hm, hv[C], av[C], bv[C];
                if (near) {
                    const unsigned S = ring + RB * (unsigned)(sp * SLOTW);
                    if (FIXED && r16) { hm = lds_ld16(S + 2 * xm); lds_ldc16<C>(S + 2 * x, hv); lds_ldc16<C>(S + 2 * (WIN + x), av); lds_ldc16<C>(S + 2 * (2 * WIN + x), bv); }
                    else { hm = lds_ld(S + 4 * xm); lds_ldc<C>(S + 4 * x, hv); lds_ldc<C>(S + 4 * (WIN + x), av); lds_ldc<C>(S + 4 * (2 * WIN + x), bv); }
                } else {
                    const int *G = g.spill + (size_t)(unsigned)usgpr(glb_ld((const int *)g.spoff + pi)) * SLOTW;
                    hm = glb_ld(G + xm); glb_ldc<C>(G + x, hv); glb_ldc<C>(G + WIN + x, av); glb_ldc<C>(G + 2 * WIN + x, bv);
                    LCD_PIN(hm);
#pragma unroll
                    for (int k = 0; k < C; ++k) { LCD_PIN(hv[k]); LCD_PIN(av[k]); LCD_PIN(bv[k]); }
                }
                if (risk) {
                    if (jb - 1 < pb || jb - 1 > pe) hm = LCD_GUARD;
#pragma unroll
                    for (int k = 0; k < C; ++k) if (jb + k < pb || jb + k > pe) { hv[k] = LCD_GUARD; av[k] = LCD_GUARD; bv[k] = LCD_GUARD; }
                }
                const int tt = t > 255 ? 255 : t;
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const int c = (k == 0 ? hm : hv[k - 1]) + sk[k] + bz, a = av[k] + bz, b = bv[k] + bz;
                    if (t == 0) { nn[k] = imax(nn[k], c); uu[k] = imax(uu[k], a); vv[k] = imax(vv[k], b); }
                    else {
                        if (c > nn[k]) { nn[k] = c; om = (om & ~((word)255 << (8 * k))) | ((word)tt << (8 * k)); }
                        if (a > uu[k]) { uu[k] = a; oa = (oa & ~((word)255 << (8 * k))) | ((word)tt << (8 * k)); }
                        if (b > vv[k]) { vv[k] = b; ob = (ob & ~((word)255 << (8 * k))) | ((word)tt << (8 * k)); }
                    }
                }
            }
        }
        // ---- F: A[k] = Hpre[k] + k*e; in-lane inclusive prefix, then one scan pair over the lane totals ----
        int hp[C], spk[C], a1[C], a2[C], p1[C], p2[C];
        const int je1 = begc * e1, je2 = begc * e2;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            hp[k] = imax(nn[k], imax(uu[k], vv[k]));                     // Hpre
            spk[k] = nn[k] == hp[k] ? 0 : uu[k] == hp[k] ? 1 : 2;        // which of match / E1 / E2 gives it (the oracle's priority)
            a1[k] = inb[k] ? hp[k] + bcle1 + k * e1 : LCD_GUARD; a2[k] = inb[k] ? hp[k] + bcle2 + k * e2 : LCD_GUARD; // (window-relative and biased, like the plain rows': one set of lane constants)
            p1[k] = k ? imax(p1[k - 1], a1[k]) : a1[k]; p2[k] = k ? imax(p2[k - 1], a2[k]) : a2[k];
        }
        int t1 = p1[C - 1], t2 = p2[C - 1];
        scan_max2(t1, t2);
        const int x1 = shr1(LCD_GUARD, t1), x2 = shr1(LCD_GUARD, t2); // exclusive prefix over the lanes
        // ---- phase B: F, H, E-out, direction code of the cells ----
        word code = 0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int pf1 = k ? imax(x1, p1[k - 1]) : x1, pf2 = k ? imax(x2, p2[k - 1]) : x2;
            const int f1 = imax(LCD_NEG, pf1 + nbcle1 - ((C == 1 ? 0 : o1) + k * e1)), f2 = imax(LCD_NEG, pf2 + nbcle2 - ((C == 1 ? 0 : o2) + k * e2));
            const int h = imax(hp[k], imax(f1, f2));
            const int eo1 = imax(imax(h - oe1, uu[k] - e1), LCD_NEG), eo2 = imax(imax(h - oe2, vv[k] - e2), LCD_NEG);
            const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
            const int hs = hp[k] == h ? spk[k] : fk;
            unsigned fl = 0; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
            { const int q2 = h - oe2, w2 = vv[k] - e2, q1 = h - oe1, w1 = uu[k] - e1, r2 = a2[k], r1 = a1[k];
              LCD_PUSH_GE(fl, q2, w2); LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, r2); LCD_PUSH_GT(fl, pf1, r1); }
            const unsigned cd = (unsigned)hs | (fl << 3) | (((om >> (8 * k)) & 255) ? CB_PM : 0);
            code |= (word)cd << (8 * k);
            pvh[k] = inb[k] ? h : LCD_GUARD; pva[k] = inb[k] ? eo1 : LCD_GUARD; pvb[k] = inb[k] ? eo2 : LCD_GUARD;
        }
        // ---- row maximum, leftmost / rightmost column (adaptive band only) ----
        int ml = 0, mr = 0;
        if (BANDED) {
            int hb = pvh[0];
#pragma unroll
            for (int k = 1; k < C; ++k) hb = imax(hb, pvh[k]);
            int bl = C - 1, brr = 0;
#pragma unroll
            for (int k = C - 2; k >= 0; --k) bl = pvh[k] == hb ? k : bl;
#pragma unroll
            for (int k = 1; k < C; ++k) brr = pvh[k] == hb ? k : brr;
            const int wm = lane63(scan_max(hb));
            const unsigned long long mk = __ballot(hb == wm); // (the row is not empty: its in-band cells are >= LCD_NEG > the filler)
            const int fl = usgpr((int)__builtin_ctzll(mk)), ll = usgpr(63 - (int)__builtin_clzll(mk));
            if (C == 1) { ml = begc + fl; mr = begc + ll; }
            else { ml = begc + C * fl + LCD_RL(bl, fl); mr = begc + C * ll + LCD_RL(brr, ll); }
        }
        // ---- stores: ring slot (values), HBM (codes; values only for rows a far successor / the end node will read) ----
        const int cw4 = ((end - begc - lo_g) + CP) & ~(CP - 1); // cells of this row in HBM, padded to a multiple of 4 (rows stay dword-aligned for every C)
        if (cused + (unsigned)cw4 > code_cap || (np > 1 && oused + (unsigned)cw4 > ord_cap) || (spf && nsp >= spill_rows)) { wo->status = LCD_ERR_CELLS; return 0; }
        {
            const int x = jb & WM;
            ring_st3(ring + RB * (unsigned)(s * SLOTW), x, pvh, pva, pvb);
            if (spf) { int *G = g.spill + (size_t)nsp * SLOTW; glb_stc<C>(G + x, pvh); glb_stc<C>(G + WIN + x, pva); glb_stc<C>(G + 2 * WIN + x, pvb); }
            if (C == 1 ? inb[0] : cl < cw4) {
                uint8_t *cp = g.code8 + (size_t)(cused + (unsigned)(cl - lo_g));
                if constexpr (C == 8) *(__attribute__((address_space(1))) unsigned long long *)cp = code;
                else if constexpr (C == 4) glb_st(cp, (int)code);
                else if constexpr (C == 2) *(__attribute__((address_space(1))) unsigned short *)cp = (unsigned short)code;
                else *(__attribute__((address_space(1))) uint8_t *)cp = (uint8_t)code;
                if (np > 1) {
                    int ow[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) ow[k] = (int)((om >> (8 * k)) & 255) | ((int)((oa >> (8 * k)) & 255) << 8) | ((int)((ob >> (8 * k)) & 255) << 16);
                    glb_stc<C>(g.ord + (size_t)(oused + (unsigned)(cl - lo_g)), ow);
                }
            }
        }
        const int be = beg | (end << 16);
        r_be = lean_wlane(be, wk, r_be);
        m_be = lean_wlane(be, s, m_be); m_mm = lean_wlane(ml | (mr << 16), s, m_mm);
        if ((np > 1 || spf) && lane == 0) {
            if (np > 1) glb_st(g.ooff + idx, (int)oused);
            if (spf) { glb_st(g.rbeg + idx, beg); glb_st(g.rend + idx, end); glb_st(g.ml + idx, ml); glb_st(g.mr + idx, mr); glb_st(g.spoff + idx, nsp); }
        }
        pv_begc = begc; pv_beg = beg; pv_end = end; pv_ml = ml; pv_mr = mr; pv_ok = true;
        cused += (unsigned)cw4; if (np > 1) oused += (unsigned)cw4; if (spf) ++nsp;
        ncell += (unsigned long long)(end - beg + 1);
        } while (0);
#ifdef LCD_X_ROWSTAT
        ++n_gn_;
#endif
        ++idx;
    }
    flush_meta(wbase, ei - wbase);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wo->cells = ncell;
#ifdef LCD_X_PLANSTAT
    wo->t_plan = t_plan_;
#endif
#ifdef LCD_X_ROWSTAT
    wo->t_setup = (unsigned long long)n_pl_ | ((unsigned long long)n_gn_ << 24) | ((unsigned long long)(C >= 4 ? n_gn_ + n_pl_ : 0u) << 44);
#endif
    const long long t_bt0 = clock64();
    wo->t_dp = (unsigned long long)(t_bt0 - t_dp0);
    code_backtrack(g, sm, pd, bi, ei, qlen, SLOTW, WM, lane, CM);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int n_cig = sm.bc[0];
    wo->status = sm.bc[1];
    wo->cig_pos = sm.bc[4];
    wo->score = sm.bc[5];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    wo->t_bt = (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

// ================= long chains: certified-band rows on the FOUR wavefronts of a 256-thread workgroup (end of round 3) =================
// The longest chain of a submission is its latency at every depth (32 reads x 4 kb = 200 ms on one wavefront, 78 % of it rows, against 147 ms of work for
// the whole chip), and a row of ~230 cells is 520 instructions for ONE wavefront at four cells per lane.  Here thread t of the workgroup owns the columns
// congruent to t modulo 256 -- one cell per lane, the same 256-column ring slots as align_lean<2, 4> -- and the four wavefronts run the row together:
//   * predecessor values come from the LDS ring (or the HBM spill rows) for EVERY row, masked by the predecessor's interval: the match term's column j - 1 is just the
//     slot entry to the left, whichever wavefront wrote it, so nothing special happens at a wavefront boundary;
//   * the horizontal-gap prefix maximum is one DPP scan pair per wavefront plus the wavefronts' totals through LDS.  The row's columns wrap around the 256 threads
//     at thread (beg mod 256): in that wavefront the lanes below the wrap point hold the row's LAST columns; their cells are kept out of the first columns' scan
//     by an offset of 2^30 on the latter (a max scan: the offset values always win, and are recognised by their size afterwards);
//   * two LDS barriers per row (totals; ring slot) -- the intervals of MODE 2 come from the table, so no row waits for another row's maximum.
// Codes / ordinals / row metadata in HBM are those of the one-cell-per-lane lean rows (a row's cells start at its interval's first column), so the code-driven
// backtrack is shared.  Returns the number of cigar entries, 0 with wo->status set, or -1 = not here (an interval wider than 254 columns, > 254 predecessors).
__device__ __attribute__((noinline)) int align_lean_mw(const Ctx *gp_, const unsigned ring_, const unsigned sq1_, const unsigned pd_ /* 0xffffffff: none */, const LcdScoring sc_,
                                                      const int bi_, const int ei_, const uint8_t *seq_hbm_, const int qlen_, WinOut *wo_) {
    const long long t_in0 = clock64();
    constexpr int NT = 256, WIN = 256, WM = WIN - 1, SLOTW = 3 * WIN, CP = 4, BIG = 1 << 30, BIGT = 1 << 28;
    Smem &sm = g_smem;
    Ctx g = *usgpr(gp_);
    ctx_to_sgpr(g);
    const int K = usgpr(g.ring_k), KM = K - 1;
    const unsigned ring = usgpr(ring_), sq1 = usgpr(sq1_), pd = usgpr(pd_);
    const int bi = usgpr(bi_), ei = usgpr(ei_), qlen = usgpr(qlen_);
    const uint8_t *seq_hbm = usgpr(seq_hbm_); WinOut *wo = usgpr(wo_);
    const int s_match = usgpr(sc_.match), s_mism = -usgpr(sc_.mismatch);
    const int o1 = usgpr(sc_.o1), e1 = usgpr(sc_.e1), o2 = usgpr(sc_.o2), e2 = usgpr(sc_.e2), oe1 = o1 + e1, oe2 = o2 + e2;
    const int tid = threadIdx.x, lane = tid & 63, wave = usgpr(tid >> 6);
    if (qlen >= 65535) return -1;
    const int QB = (qlen + 12 + 15) & ~15;
    if (sq1 < ring + (unsigned)(K * SLOTW * 4) || (K & KM) != 0) return -1; // (the pool is laid out for ring slots at least this wide: PoaChain.wmax >= 256)
    for (int j = tid; j < QB; j += NT) lds_st_u8(sq1 + j, (j >= 1 && j <= qlen) ? glb_ld_u8(seq_hbm + (j - 1)) : 4); // shifted: sq1[j] = q[j-1]
    const int qclamp = QB - CP;
    const unsigned code_cap = (unsigned)(g.cell_cap > 0xfffffff0ull ? 0xfffffff0ull : g.cell_cap);
    const unsigned ord_cap = g.spill_x > 2 ? code_cap : (unsigned)((g.cell_cap / 4) > 0xfffffff0ull ? 0xfffffff0ull : (g.cell_cap / 4));
    const long long spill_rows_ll = g.cell_cap * g.spill_x > 64 ? (long long)((g.cell_cap * g.spill_x - 64) / ((unsigned long long)SLOTW * 4)) : 0;
    const int spill_rows = (int)(spill_rows_ll > 0x7fffffffll ? 0x7fffffffll : spill_rows_ll);
    const int *const hull = g.cert + 6 * (size_t)g.node_cap;
    // ---- source row (slot 0): columns 0 .. end0 ----
    int end0;
    { const int hw = usgpr(glb_ld(hull + bi)); end0 = hw >> 16; if ((hw & 65535) != 0) return -1; }
    if (end0 + 2 > WIN) return -1;
    int nsp = 0;
    {
        const bool spf = (usgpr(glb_ld_u8(g.imap + bi)) & 2) != 0;
        if (spf && spill_rows < 1) { wo->status = LCD_ERR_CELLS; return 0; }
        const int j = tid;
        const int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
        const int h = j ? imax(f1, f2) : 0;
        const bool in = j <= end0;
        const int vh = in ? h : LCD_GUARD, va = in ? h - oe1 : LCD_GUARD, vb2 = in ? h - oe2 : LCD_GUARD;
        *(lcd_lds_i32 *)(uintptr_t)(ring + 4 * tid) = vh; *(lcd_lds_i32 *)(uintptr_t)(ring + 4 * (WIN + tid)) = va; *(lcd_lds_i32 *)(uintptr_t)(ring + 4 * (2 * WIN + tid)) = vb2;
        if (spf) { int *G = g.spill; glb_st(G + tid, vh); glb_st(G + WIN + tid, va); glb_st(G + 2 * WIN + tid, vb2); nsp = 1; }
        if (tid == 0) { glb_st(g.rbeg + bi, 0); glb_st(g.rend + bi, end0); glb_st(g.roff + bi, 0); glb_st(g.ml + bi, 0); glb_st(g.mr + bi, 0); glb_st(g.spoff + bi, 0); }
    }
    int m_be = 1;                       // ring slot meta, lane s of every wavefront: beg | end << 16 of slot s (an empty slot is beg 1, end 0)
    if (lane == 0) m_be = end0 << 16;
    unsigned cused = 0, oused = 0; unsigned long long ncell = (unsigned long long)end0 + 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t_dp0 = clock64();
    wo->t_setup = (unsigned long long)(t_dp0 - t_in0);
    int w_pk = 0, w_x = 0, w_pi0 = 0, w_pi1 = 0, w_p0 = 0; // plan window, lane = row - wbase (every wavefront keeps its own copy)
    int r_be = 1, r_off = 0;
    int wbase = bi + 1;
    auto load_plan = [&](const int base) { // packed word as in align_lean: #preds | base << 8 | spill << 11 | unreachable << 12 | backbone << 13 | bonus0 << 14 | bonus1 << 19
        const int ri = base + lane;
        w_pk = 1 << 12;
        if (ri < ei) {
            const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
            const int cnt = s1 - s0;
            int b0 = 0, b1 = 0;
            w_p0 = s0;
            if (cnt > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); b0 = glb_ld(g.pl_bonus + s0); }
            if (cnt > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); b1 = glb_ld(g.pl_bonus + s0 + 1); }
            const int rem = glb_ld(g.pl_rem + ri);
            w_x = glb_ld(hull + ri);
            w_pk = imin(cnt, 255) | (glb_ld_u8(g.pl_base + ri) << 8) | ((glb_ld_u8(g.imap + ri) & 2) << 10) | (rem == (1 << 30) ? 1 << 12 : 0) | (b0 << 14) | (b1 << 19);
        }
        LCD_PIN(w_pk); LCD_PIN(w_x); LCD_PIN(w_pi0); LCD_PIN(w_pi1); LCD_PIN(w_p0);
    };
    auto flush_meta = [&](const int base, const int n) { // rows base .. base + n - 1 (wavefront 0 stores; the windows are the same in all four)
        if (wave == 0 && lane < n) { glb_st(g.rbeg + base + lane, r_be & 65535); glb_st(g.rend + base + lane, (int)((unsigned)r_be >> 16)); glb_st(g.roff + base + lane, r_off); }
    };
    load_plan(wbase);
    for (int idx = bi + 1; idx < ei; ++idx) {
        if (idx - wbase == 64) { flush_meta(wbase, 64); wbase = idx; load_plan(wbase); }
        const int wk = idx - wbase;
        const int pk = LCD_RL(w_pk, wk);
        const int s = (idx - bi) & KM;
        if (pk & (1 << 12)) { m_be = lean_wlane(1, s, m_be); r_be = lean_wlane(1, wk, r_be); continue; } // not reachable from the begin node
        const int np = pk & 255, vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
        const bool spf = (pk >> 11) & 1;
        if (np == 255) return -1;
        const int xw = LCD_RL(w_x, wk);
        const int pi0 = LCD_RL(w_pi0, wk);
        int pi1 = 0, bz1 = 0, p0 = 0;
        if (np > 1) { pi1 = LCD_RL(w_pi1, wk); bz1 = (pk >> 19) & 31; p0 = LCD_RL(w_p0, wk); }
        const int beg = xw & 65535, end = (int)((unsigned)xw >> 16); // the row's certified interval (beg > end: no cell of the row can lie on an optimal path)
        if (beg > end) {
            m_be = lean_wlane(1, s, m_be); r_be = lean_wlane(1, wk, r_be);
            if (spf && tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
            continue;
        }
        if (end - beg + 2 > WIN) return -1;
        const int j = beg + ((tid - beg) & WM);       // this thread's column of the row
        const bool inb = j <= end;
        const int x = tid, xm = (tid - 1) & WM;       // ring slot entries of columns j and j - 1
        int sk;
        { const int q = lds_ld_u8(sq1 + (unsigned)imin(j, qclamp)); sk = (vb >= 4 || q >= 4) ? 0 : (vb == q ? s_match : s_mism); }
        // ---- phase A: best match / E1 / E2 input over the predecessors (first maximum keeps its ordinal) ----
        int nn = LCD_NEG, uu = LCD_NEG, vv = LCD_NEG, om = 0, oa = 0, ob = 0;
        for (int t = 0; t < np; ++t) {
            int pi = t == 0 ? pi0 : pi1, bz = t == 0 ? bz0 : bz1;
            if (t > 1) { pi = usgpr(glb_ld(g.pl_pidx + p0 + t)); bz = usgpr(glb_ld(g.pl_bonus + p0 + t)); }
            const bool near = idx - pi <= K;
            int pb, pe, hm, av, bv;
            if (near) {
                const int sp = (pi - bi) & KM;
                const int be = LCD_RL(m_be, sp); pb = be & 65535; pe = (int)((unsigned)be >> 16);
                if (pb > pe) continue;
                const unsigned S = ring + 4 * sp * SLOTW;
                hm = lds_ld(S + 4 * xm); av = lds_ld(S + 4 * (WIN + x)); bv = lds_ld(S + 4 * (2 * WIN + x));
            } else { // a far row: its metadata and values were stored to HBM when it was made (and waited for before that row's closing barrier)
                pb = usgpr(glb_ld(g.rbeg + pi)); pe = usgpr(glb_ld(g.rend + pi));
                if (pb > pe) continue;
                const int *G = g.spill + (size_t)(unsigned)usgpr(glb_ld((const int *)g.spoff + pi)) * SLOTW;
                hm = glb_ld(G + xm); av = glb_ld(G + WIN + x); bv = glb_ld(G + 2 * WIN + x);
                LCD_PIN(hm); LCD_PIN(av); LCD_PIN(bv);
            }
            // a slot is addressed by (column mod 256): entries that belong to other columns of the predecessor's interval, or to none, are fillers here
            if (j - 1 < pb || j - 1 > pe) hm = LCD_GUARD;
            if (j < pb || j > pe) { av = LCD_GUARD; bv = LCD_GUARD; }
            const int tt = t > 255 ? 255 : t;
            const int c = hm + sk + bz, a = av + bz, b = bv + bz;
            if (t == 0) { nn = imax(nn, c); uu = imax(uu, a); vv = imax(vv, b); }
            else {
                if (c > nn) { nn = c; om = tt; }
                if (a > uu) { uu = a; oa = tt; }
                if (b > vv) { vv = b; ob = tt; }
            }
        }
        // ---- F: prefix maximum of A[j] = Hpre[j] + j e over the row's columns, which start at thread (beg mod 256) and wrap ----
        const int hp = imax(nn, imax(uu, vv));
        const int spk = nn == hp ? 0 : uu == hp ? 1 : 2;
        const int je1 = j * e1, je2 = j * e2;
        const int a1 = inb ? hp + je1 : LCD_GUARD, a2 = inb ? hp + je2 : LCD_GUARD;
        const int t0 = beg & WM, wrapw = t0 >> 6, wrapl = t0 & 63;
        const bool tail = wave == wrapw && lane < wrapl;   // the row's last columns, in the lanes below the wrap point of that wavefront
        int v1 = (inb && !tail) ? a1 + BIG : a1, v2 = (inb && !tail) ? a2 + BIG : a2;
        scan_max2(v1, v2);
        const int buf = idx & 1;
        if (lane == 63) { sm.tot1[buf][wave] = v1 >= BIGT ? v1 - BIG : LCD_GUARD; sm.tot2[buf][wave] = v2 >= BIGT ? v2 - BIG : LCD_GUARD; } // total of the wavefront's first-columns segment
        int x1 = shr1(LCD_GUARD, v1), x2 = shr1(LCD_GUARD, v2);
        x1 = x1 >= BIGT ? x1 - BIG : (tail ? x1 : LCD_GUARD); x2 = x2 >= BIGT ? x2 - BIG : (tail ? x2 : LCD_GUARD);
        lds_barrier<NT>();
        {
            const int pw = (wave - wrapw) & 3; // this wavefront's place in the row's column order
            int c1 = LCD_GUARD, c2 = LCD_GUARD;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t1k = sm.tot1[buf][k], t2k = sm.tot2[buf][k];
                const bool before = tail || ((k - wrapw) & 3) < pw;
                c1 = before ? imax(c1, t1k) : c1; c2 = before ? imax(c2, t2k) : c2;
            }
            x1 = imax(x1, c1); x2 = imax(x2, c2);
        }
        // ---- phase B: F, H, E-out, direction code ----
        const int f1 = imax(LCD_NEG, x1 - je1 - o1), f2 = imax(LCD_NEG, x2 - je2 - o2);
        const int h = imax(hp, imax(f1, f2));
        const int q1 = h - oe1, w1 = uu - e1, q2 = h - oe2, w2 = vv - e2;
        const int eo1 = imax(imax(q1, w1), LCD_NEG), eo2 = imax(imax(q2, w2), LCD_NEG);
        const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
        const int hs = hp == h ? spk : fk;
        unsigned fl = 0; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
        LCD_PUSH_GE(fl, q2, w2); LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, x2, a2); LCD_PUSH_GT(fl, x1, a1);
        const unsigned cd = (unsigned)hs | (fl << 3) | (om ? CB_PM : 0);
        const int oh = inb ? h : LCD_GUARD, oa1 = inb ? eo1 : LCD_GUARD, oa2 = inb ? eo2 : LCD_GUARD;
        // ---- stores: ring slot (values), HBM (codes; values only for rows a far successor / the end node will read) ----
        const int cw4 = ((end - beg) + CP) & ~(CP - 1);
        if (cused + (unsigned)cw4 > code_cap || (np > 1 && oused + (unsigned)cw4 > ord_cap) || (spf && nsp >= spill_rows)) { wo->status = LCD_ERR_CELLS; return 0; }
        {
            const unsigned S = ring + 4 * s * SLOTW;
            *(lcd_lds_i32 *)(uintptr_t)(S + 4 * x) = oh; *(lcd_lds_i32 *)(uintptr_t)(S + 4 * (WIN + x)) = oa1; *(lcd_lds_i32 *)(uintptr_t)(S + 4 * (2 * WIN + x)) = oa2;
            if (spf) { int *G = g.spill + (size_t)nsp * SLOTW; glb_st(G + x, oh); glb_st(G + WIN + x, oa1); glb_st(G + 2 * WIN + x, oa2); }
            const int cj = j - beg;
            if (cj < cw4) {
                *(__attribute__((address_space(1))) uint8_t *)(g.code8 + (size_t)(cused + (unsigned)cj)) = (uint8_t)cd;
                if (np > 1) glb_st(g.ord + (size_t)(oused + (unsigned)cj), om | (oa << 8) | (ob << 16));
            }
        }
        const int be = beg | (end << 16);
        r_be = lean_wlane(be, wk, r_be); r_off = lean_wlane((int)cused, wk, r_off);
        m_be = lean_wlane(be, s, m_be);
        if ((np > 1 || spf) && tid == 0) {
            if (np > 1) glb_st(g.ooff + idx, (int)oused);
            if (spf) { glb_st(g.rbeg + idx, beg); glb_st(g.rend + idx, end); glb_st(g.ml + idx, 0); glb_st(g.mr + idx, 0); glb_st(g.spoff + idx, nsp); }
        }
        cused += (unsigned)cw4; if (np > 1) oused += (unsigned)cw4;
        if (spf) { ++nsp; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // (a row that others will read from HBM: there before anybody passes the barrier)
        ncell += (unsigned long long)(end - beg + 1);
        lds_barrier<NT>(); // the ring slot, to the other wavefronts
    }
    flush_meta(wbase, ei - wbase);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    wo->cells = ncell;
    const long long t_bt0 = clock64();
    wo->t_dp = (unsigned long long)(t_bt0 - t_dp0);
    if (wave == 0) code_backtrack(g, sm, pd, bi, ei, qlen, SLOTW, WM, lane, ~0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_cig = sm.bc[0];
    wo->status = sm.bc[1];
    wo->cig_pos = sm.bc[4];
    wo->score = sm.bc[5];
    __syncthreads();
    wo->t_bt = (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

// ================= unbanded K2 rows: systolic across the wavefronts of the workgroup =================
// wb < 0 makes every row [0, qlen] (oracle: w = qlen), so nothing about a row depends on the other rows' maxima and the only
// data that crosses a wavefront boundary is (a) H of the last column of the left neighbour's 256-column segment, for the
// match term of lane 0, and (b) the running prefix of the horizontal-gap scan.  Both go through small LDS mailboxes
// (bndH / carry, SYS_D rows deep) and a per-wavefront progress counter: wavefront w starts row r once w-1 has published row
// r, so the workgroup runs as a pipeline (w works on row r while w-1 is on r+1 ...) with NO workgroup barrier in the row
// loop -- with s_barrier twice per row the 16 wavefronts of a 4 096-column row spent 2/3 of their cycles parked.
// Per-lane constants (query bases, column * e) are loaded once per read; out-of-range columns (> qlen) are computed and
// stored like the others and never read, so there is no band mask.  E planes hold E + e ("E-out before the extension
// charge"): max(H - o, Ein), one subtraction less per cell; the reader folds the -e into the edge bonus.
constexpr int SYS_D = 16;
// returns false if the neighbour did not get there within ~2^26 polls (seconds): cannot happen unless the kernel is broken, but a bounded
// spin turns such a bug into an error status instead of a hung GPU
__device__ __forceinline__ bool poll_ge(const int *p, const int v) { // p is in LDS: volatile ds_read, no flat access
    const volatile lcd_lds_i32 *q = (const volatile lcd_lds_i32 *)(uintptr_t)lds_off(p);
    int spins = 0;
    while (*q < v) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 26)) return false; }
    return true;
}

// ================= long chains: certified-band rows SYSTOLIC over the four wavefronts of a 256-thread workgroup (round 4) =================
// The longest chain of a submission is its latency at every depth: 32 reads x 4 kb of a K2 chain whose certified intervals are ~210 columns wide were 195 ms on ONE
// wavefront (align_lean<2, 4>: ~480 instructions per row of four cells per lane) against 141 ms of work for the whole chip.  align_lean_mw put the row on four wavefronts
// with two barriers per row and every predecessor through LDS and came out at 0.85x.  Here the four wavefronts form a PIPELINE, as in align_unbanded:
//   * thread t owns the columns congruent to C t .. C t + C - 1 modulo WIN = 256 C for the whole read, so the previous row's values of a thread's cells are in that
//     thread's REGISTERS whatever the interval does (no lane shifts, no LDS on a backbone row's critical path); the diagonal input of a lane's first cell is one
//     DPP move from the lane to the left, and for lane 0 the boundary H the wavefront to the left has published in its mailbox;
//   * a row's interval covers at most four consecutive 64 C-column blocks -- one per wavefront, no wavefront twice (rows wider than 192 C columns are not taken here) --
//     and runs through them in column order starting at wavefront S = block of its first column: S starts row r as soon as it has finished row r - 1; the next one once
//     its left neighbour has published the row's running gap prefixes (carry1 / carry2) and boundary H; wavefronts the interval does not reach only keep count.
//     Steady state: one row per ~150 instructions of ONE wavefront, the four of them one row apart.  No workgroup barrier inside the read;
//   * rows with several predecessors / predecessors further back read their OWN columns of those rows from the LDS ring (or the HBM spill rows) -- written by the
//     same thread -- masked by the predecessor's interval, and the boundary column from the mailbox / the spill row: they stay inside the pipeline;
//   * a wavefront is at most SYS_D - K - 1 rows ahead of its right neighbour (mailbox slots are re-used modulo SYS_D); every wait is a bounded poll (LCD_ERR_SYNC).
// Codes / ordinals / row metadata in HBM are those of align_lean<2, C> (a row's cells start at its interval's first column rounded down to C), so the code-driven
// backtrack is shared.  Returns the number of cigar entries, 0 with wo->status set, or -1 = not here (an interval wider than the blocks hold, > 254 predecessors).
template <int C>
__device__ __attribute__((noinline)) int align_cyc(const Ctx *gp_, const unsigned ring_, const unsigned sq1_, const unsigned pd_ /* 0xffffffff: none */, const LcdScoring sc_,
                                                   const int bi_, const int ei_, const uint8_t *seq_hbm_, const int qlen_, WinOut *wo_) {
    const long long t_in0 = clock64();
    constexpr int NT = 256, BW = 64 * C, WIN = 256 * C, WM = WIN - 1, SLOTW = 3 * WIN, CP = 4, CM = ~(C - 1), BSH = C == 1 ? 6 : 7;
    constexpr int AHEAD = SYS_D - 3; // a wavefront is at most this many rows ahead of its right neighbour (mailbox slots are re-used modulo SYS_D; readers look back K = 2 rows)
    typedef typename LeanT<C>::word word;
    Smem &sm = g_smem;
    Ctx g = *usgpr(gp_);
    ctx_to_sgpr(g);
    const unsigned ring = usgpr(ring_), sq1 = usgpr(sq1_), pd = usgpr(pd_);
    const int bi = usgpr(bi_), ei = usgpr(ei_), qlen = usgpr(qlen_);
    const uint8_t *seq_hbm = usgpr(seq_hbm_); WinOut *wo = usgpr(wo_);
    const int s_match = usgpr(sc_.match), s_mism = -usgpr(sc_.mismatch);
    const int o1 = usgpr(sc_.o1), e1 = usgpr(sc_.e1), o2 = usgpr(sc_.o2), e2 = usgpr(sc_.e2), oe1 = o1 + e1, oe2 = o2 + e2;
    const int tid = threadIdx.x, lane = tid & 63, wave = usgpr(tid >> 6);
    if (qlen >= 65535 || usgpr(g.ring_k) != 2) return -1; // (two ring slots: the slot metadata of a run of backbone rows is kept as "the last two rows")
    const int QB = (qlen + 12 + 15) & ~15;
    if (sq1 < ring + (unsigned)(2 * SLOTW * 4)) return -1; // (the pool is laid out for ring slots at least this wide: PoaChain.wmax >= WIN)
    for (int j = tid; j < QB; j += NT) lds_st_u8(sq1 + j, (j >= 1 && j <= qlen) ? glb_ld_u8(seq_hbm + (j - 1)) : 4); // shifted: sq1[j] = q[j-1]
    const int qclamp = QB - CP;
    const unsigned code_cap = (unsigned)(g.cell_cap > 0xfffffff0ull ? 0xfffffff0ull : g.cell_cap);
    const unsigned ord_cap = g.spill_x > 2 ? code_cap : (unsigned)((g.cell_cap / 4) > 0xfffffff0ull ? 0xfffffff0ull : (g.cell_cap / 4));
    const long long spill_rows_ll = g.cell_cap * g.spill_x > 64 ? (long long)((g.cell_cap * g.spill_x - 64) / ((unsigned long long)SLOTW * 4)) : 0;
    const int spill_rows = (int)(spill_rows_ll > 0x7fffffffll ? 0x7fffffffll : spill_rows_ll);
    const int *const hull = g.cert + 6 * (size_t)g.node_cap;
    const int cl = C * tid, cll = C * lane;         // this thread's cells inside a ring slot (modulo WIN) / inside its wavefront's block
    const int left = (wave + 3) & 3, right = (wave + 1) & 3;
    // mailboxes (LDS byte offsets): progress counter per wavefront; boundary H, running gap prefixes per (row mod SYS_D, wavefront)
    // (made opaque scalars once: left to itself the compiler re-derives them -- a load from the module's LDS table and a null test -- inside the row loop)
    const unsigned mb_prog = usgpr(lds_off(&g_wide.prog[0])), mb_c1 = usgpr(lds_off(&g_wide.carry1[0][0])), mb_c2 = usgpr(lds_off(&g_wide.carry2[0][0])), mb_h = usgpr(lds_off(&g_wide.bndH[0][0]));
    const unsigned a_pl = usgpr(mb_prog + 4 * left), a_pr = usgpr(mb_prog + 4 * right), a_me = usgpr(mb_prog + 4 * wave);
    const unsigned l_c1 = usgpr(mb_c1 + 4 * left), l_c2 = usgpr(mb_c2 + 4 * left), l_h = usgpr(mb_h + 4 * left);   // the left neighbour's mailbox column, row 0
    const unsigned m_c1 = usgpr(mb_c1 + 4 * wave), m_c2 = usgpr(mb_c2 + 4 * wave), m_h = usgpr(mb_h + 4 * wave);   // this wavefront's
    // ---- source row (slot 0): columns 0 .. end0, all of them in the blocks 0 .. 2 ----
    int end0;
    { const int hw = usgpr(glb_ld(hull + bi)); end0 = hw >> 16; if ((hw & 65535) != 0) return -1; }
    if (end0 + 2 > WIN - BW) return -1;
    int nsp = 0;
    int pvh[C], pva[C], pvb[C]; // this thread's cells of the row before (fillers where that row's interval does not reach)
    {
        const bool spf = (usgpr(glb_ld_u8(g.imap + bi)) & 2) != 0;
        if (spf && spill_rows < 1) { wo->status = LCD_ERR_CELLS; return 0; }
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int j = cl + k;
            const int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            const int h = j ? imax(f1, f2) : 0;
            const bool in = j <= end0;
            pvh[k] = in ? h : LCD_GUARD; pva[k] = in ? h - oe1 : LCD_GUARD; pvb[k] = in ? h - oe2 : LCD_GUARD;
        }
        lds_stc<C>(ring + 4 * cl, pvh); lds_stc<C>(ring + 4 * (WIN + cl), pva); lds_stc<C>(ring + 4 * (2 * WIN + cl), pvb);
        if (spf) { int *G = g.spill; glb_stc<C>(G + cl, pvh); glb_stc<C>(G + WIN + cl, pva); glb_stc<C>(G + 2 * WIN + cl, pvb); nsp = 1; }
        if (tid == 0) { glb_st(g.rbeg + bi, 0); glb_st(g.rend + bi, end0); glb_st(g.roff + bi, 0); glb_st(g.ml + bi, 0); glb_st(g.mr + bi, 0); glb_st(g.spoff + bi, 0); sm.bc[7] = LCD_OK; }
        if (lane == 63) { // the source row's mailbox entries: boundary H of this wavefront's last column; progress = bi
            *(lcd_lds_i32 *)(uintptr_t)(m_h + (unsigned)((bi & (SYS_D - 1)) * (MAXW * 4))) = pvh[C - 1];
            *(lcd_lds_i32 *)(uintptr_t)a_me = bi;
        }
    }
    // intervals (beg | end << 16; 1 = none) of the last two rows = of the two ring slots: l_be is row idx - 1 (whose values the pv registers hold), l_be2 row idx - 2
    int l_be = end0 << 16, l_be2 = 1;
    unsigned cused = 0, oused = 0; unsigned long long ncell = (unsigned long long)end0 + 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t_dp0 = clock64();
    wo->t_setup = (unsigned long long)(t_dp0 - t_in0);
    int w_pk = 0, w_x = 0, w_pi0 = 0, w_pi1 = 0, w_p0 = 0; // plan window, lane = row - wbase (every wavefront keeps its own copy)
    int r_be = 1, r_off = 0;                              // row metadata window (wavefront 0 stores it)
    int wbase = bi + 1;
    auto load_plan = [&](const int base) { // packed word as in align_lean: #preds | base << 8 | spill << 11 | unreachable << 12 | backbone << 13 | bonus0 << 14 | bonus1 << 19
        const int ri = base + lane;
        w_pk = 1 << 12;
        if (ri < ei) {
            const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
            const int cnt = s1 - s0;
            int b0 = 0, b1 = 0;
            w_p0 = s0;
            if (cnt > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); b0 = glb_ld(g.pl_bonus + s0); }
            if (cnt > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); b1 = glb_ld(g.pl_bonus + s0 + 1); }
            const int rem = glb_ld(g.pl_rem + ri);
            w_x = glb_ld(hull + ri);
            w_pk = imin(cnt, 255) | (glb_ld_u8(g.pl_base + ri) << 8) | ((glb_ld_u8(g.imap + ri) & 2) << 10) | (rem == (1 << 30) ? 1 << 12 : 0)
                 | ((cnt == 1 && w_pi0 == ri - 1) ? 1 << 13 : 0) | (b0 << 14) | (b1 << 19);
        }
        LCD_PIN(w_pk); LCD_PIN(w_x); LCD_PIN(w_pi0); LCD_PIN(w_pi1); LCD_PIN(w_p0);
    };
    auto flush_meta = [&](const int base, const int n) { // rows base .. base + n - 1
        if (wave == 0 && lane < n) { glb_st(g.rbeg + base + lane, r_be & 65535); glb_st(g.rend + base + lane, (int)((unsigned)r_be >> 16)); glb_st(g.roff + base + lane, r_off); }
    };
    // "the left neighbour has published row `need_l` and the right one row `need_r`" -- and, read behind the same counters (LDS operations of a wavefront are in
    // order, the writer publishes its data before its counter), this row's running gap prefixes and the boundary H of row `rb`.  Bounded: false = timed out
    auto sync_rows = [&](const int need_l_, const int need_r_, const int rc, const int rb, int &c1, int &c2, int &bh) {
        const int need_l = usgpr(need_l_), need_r = usgpr(need_r_);
        const unsigned ac = (unsigned)usgpr((rc & (SYS_D - 1)) * (MAXW * 4)), ab = (unsigned)usgpr((rb & (SYS_D - 1)) * (MAXW * 4));
        for (int spins = 0;; ++spins) {
            const int pl = *(const volatile lcd_lds_i32 *)(uintptr_t)a_pl, pr = *(const volatile lcd_lds_i32 *)(uintptr_t)a_pr;
            const int v1 = *(const volatile lcd_lds_i32 *)(uintptr_t)(l_c1 + ac), v2 = *(const volatile lcd_lds_i32 *)(uintptr_t)(l_c2 + ac), vh = *(const volatile lcd_lds_i32 *)(uintptr_t)(l_h + ab);
            if (usgpr(pl) >= need_l && usgpr(pr) >= need_r) { c1 = v1; c2 = v2; bh = vh; return true; }
            if (usgpr(spins) > (1 << 24)) return false;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto publish = [&](const int idx) { // "this wavefront is done with row idx": its LDS writes (ring slot, mailbox) first
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 63) *(volatile lcd_lds_i32 *)(uintptr_t)a_me = idx;
    };
    int fail = 0; // -1: not representable here; > 0: an error status.  Every wavefront takes the same decision at the same row (same plan, same counters)
    unsigned long long t_poll = 0;
    load_plan(wbase);
    int idx = bi + 1;
    while (idx < ei) {
        if (idx - wbase == 64) {
            flush_meta(wbase, 64); wbase = idx; load_plan(wbase);
            if ((unsigned long long)clock64() > g.wd_deadline) { fail = LCD_ERR_WATCHDOG; break; } // (the others reach the same row; a wavefront that waits for this one runs into its poll bound)
        }
        int wk = idx - wbase;
        // ===== a run of backbone rows: reachable, one usable predecessor = the row before (in the registers), not spilled.  Anything else leaves the loop with the
        // row untouched and is the general row below. =====
        {
            // (the run's loop-carried scalars are locals, and made scalars again at the top of every row: as phis of the outer loop -- or behind the polling loop,
            //  whose exit the compiler takes for divergent -- they end up in VGPRs, and with them every decision of the row)
            int f_be = usgpr(l_be), f_be2 = usgpr(l_be2), f_idx = usgpr(idx), f_wk = usgpr(wk); unsigned run_cells = 0, f_cused = usgpr(cused);
            while (true) {
                f_be = usgpr(f_be); f_be2 = usgpr(f_be2); f_idx = usgpr(f_idx); f_wk = usgpr(f_wk); f_cused = usgpr(f_cused); run_cells = usgpr(run_cells);
                const int idx = f_idx, wk = f_wk; const unsigned cused = f_cused;
                if (wk >= 64) break;
                const int pk = LCD_RL(w_pk, wk);
                if ((pk & 0x38ff) != 0x2001) break;
                const int xw = LCD_RL(w_x, wk);
                const int beg = xw & 65535, end = (int)((unsigned)xw >> 16);
                const int begc = beg & CM, blk0 = begc & ~(BW - 1);
                const int cw4 = ((end - begc) + CP) & ~(CP - 1);
                const int pb = f_be & 65535, pe = (int)((unsigned)f_be >> 16);
                // (not here: an empty row / row before, an interval that would come round to its first wavefront, a full DP region, and columns a whole window away
                //  from the row before -- a register would hold the value of another column)
                if (beg > end || pb > pe || end - blk0 + 2 > WIN || cused + (unsigned)cw4 > code_cap || pe - begc + 2 >= WIN || end - pb + 1 >= WIN) break;
                const int pw = (wave - (blk0 >> BSH)) & 3;                 // this wavefront's place in the row's column order (0: the row's first)
                const int j0 = blk0 + BW * pw;                             // the column of its lane 0
                const int j = j0 + cll;                                    // this lane's first column
                const int vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
                const int mt = vb >= 4 ? 0 : s_match, mm = vb >= 4 ? 0 : s_mism;
                word qw;
                {
                    const unsigned a = sq1 + (unsigned)imin(j, qclamp);
                    if constexpr (C == 1) qw = (word)*(const lcd_lds_u8 *)(uintptr_t)a;
                    else qw = (word)*(const __attribute__((address_space(3))) unsigned short *)(uintptr_t)a;
                }
                // the first wavefront of a row does not wait for its left neighbour -- the LAST wavefront of the rows before -- unless the row before reaches into that
                // block: then the neighbour was that row's first wavefront and is ahead anyway
                const bool need_b = j0 - 1 >= pb && j0 - 1 <= pe;
                int cin1, cin2, bnd;
                if (!sync_rows(pw > 0 ? idx : (need_b ? idx - 1 : bi), idx - AHEAD, idx, idx - 1, cin1, cin2, bnd)) { fail = LCD_ERR_SYNC; break; }
                if (pw == 0) { cin1 = LCD_GUARD; cin2 = LCD_GUARD; }
                if (!need_b) bnd = LCD_GUARD;
                const int hm = __builtin_amdgcn_update_dpp(bnd, pvh[C - 1], 0x138, 0xf, 0xf, false); // wave_shr:1, lane 0 keeps the boundary
                const int je1 = __mul24(j, e1), je2 = __mul24(j, e2);
                int hp[C], spk[C], a1[C], a2[C], p1[C], p2[C], uu[C], vv[C]; bool inb[C];
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const int q = (int)(qw >> (8 * k)) & 255;
                    const int sk = q >= 4 ? 0 : (q == vb ? mt : mm);
                    const int nn = imax(LCD_NEG, (k ? pvh[k - 1] : hm) + sk + bz0);
                    uu[k] = imax(LCD_NEG, pva[k] + bz0); vv[k] = imax(LCD_NEG, pvb[k] + bz0);
                    inb[k] = (unsigned)(j + k - beg) <= (unsigned)(end - beg);
                    hp[k] = imax(nn, imax(uu[k], vv[k]));
                    spk[k] = nn == hp[k] ? 0 : uu[k] == hp[k] ? 1 : 2;
                    a1[k] = inb[k] ? hp[k] + (je1 + k * e1) : LCD_GUARD; a2[k] = inb[k] ? hp[k] + (je2 + k * e2) : LCD_GUARD;
                    p1[k] = k ? imax(p1[k - 1], a1[k]) : a1[k]; p2[k] = k ? imax(p2[k - 1], a2[k]) : a2[k];
                }
                int t1 = p1[C - 1], t2 = p2[C - 1];
                scan_max2(t1, t2);
                const int x1 = imax(shr1(LCD_GUARD, t1), cin1), x2 = imax(shr1(LCD_GUARD, t2), cin2);
                word code = 0;
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const int pf1 = k ? imax(x1, p1[k - 1]) : x1, pf2 = k ? imax(x2, p2[k - 1]) : x2;
                    const int f1 = imax(LCD_NEG, pf1 - (o1 + je1 + k * e1)), f2 = imax(LCD_NEG, pf2 - (o2 + je2 + k * e2));
                    const int h = imax(hp[k], imax(f1, f2));
                    const int q1 = h - oe1, w1 = uu[k] - e1, q2 = h - oe2, w2 = vv[k] - e2;
                    const int eo1 = imax(imax(q1, w1), LCD_NEG), eo2 = imax(imax(q2, w2), LCD_NEG);
                    const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
                    const int hs = hp[k] == h ? spk[k] : fk;
                    unsigned fl = 0; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
                    { const int r2 = a2[k], r1 = a1[k]; LCD_PUSH_GE(fl, q2, w2); LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, r2); LCD_PUSH_GT(fl, pf1, r1); }
                    code |= (word)((unsigned)hs | (fl << 3)) << (8 * k);
                    pvh[k] = inb[k] ? h : LCD_GUARD; pva[k] = inb[k] ? eo1 : LCD_GUARD; pvb[k] = inb[k] ? eo2 : LCD_GUARD;
                }
                {
                    const unsigned SL = ring + 4 * ((idx - bi) & 1) * SLOTW;
                    lds_stc<C>(SL + 4 * cl, pvh); lds_stc<C>(SL + 4 * (WIN + cl), pva); lds_stc<C>(SL + 4 * (2 * WIN + cl), pvb);
                    const int cj = j - begc;
                    if ((unsigned)cj < (unsigned)cw4) {
                        uint8_t *cp = g.code8 + (size_t)(cused + (unsigned)cj);
                        if constexpr (C == 2) *(__attribute__((address_space(1))) unsigned short *)cp = (unsigned short)code;
                        else *(__attribute__((address_space(1))) uint8_t *)cp = (uint8_t)code;
                    }
                    if (lane == 63) {
                        const unsigned mo = (unsigned)((idx & (SYS_D - 1)) * (MAXW * 4));
                        *(lcd_lds_i32 *)(uintptr_t)(m_h + mo) = pvh[C - 1];
                        *(lcd_lds_i32 *)(uintptr_t)(m_c1 + mo) = imax(cin1, t1); *(lcd_lds_i32 *)(uintptr_t)(m_c2 + mo) = imax(cin2, t2);
                    }
                }
                const int be = beg | (end << 16);
                if (wave == 0) { r_be = lean_wlane(be, wk, r_be); r_off = lean_wlane((int)cused, wk, r_off); }
                f_be2 = f_be; f_be = be;
                f_cused = cused + (unsigned)cw4; run_cells += (unsigned)(end - beg + 1);
                publish(idx);
                f_idx = idx + 1; f_wk = wk + 1;
            }
            l_be = usgpr(f_be); l_be2 = usgpr(f_be2); ncell += usgpr(run_cells); idx = usgpr(f_idx); wk = usgpr(f_wk); cused = usgpr(f_cused);
            if (fail != 0) break;
            if (wk == 64 || idx >= ei) continue;
        }
        // ===== general row: several predecessors, a predecessor further back, a spilled / empty / unreachable row =====
        do {
        const int pk = LCD_RL(w_pk, wk);
        if (pk & (1 << 12)) { // not reachable from the begin node
            if (wave == 0) r_be = lean_wlane(1, wk, r_be);
#pragma unroll
            for (int k = 0; k < C; ++k) { pvh[k] = LCD_GUARD; pva[k] = LCD_GUARD; pvb[k] = LCD_GUARD; }
            l_be2 = l_be; l_be = 1; publish(idx);
            break;
        }
        const int np = pk & 255, vb = (pk >> 8) & 7, bz0 = (pk >> 14) & 31;
        const bool spf = (pk >> 11) & 1;
        if (np == 255) { fail = -1; break; }
        const int xw = LCD_RL(w_x, wk);
        const int pi0 = LCD_RL(w_pi0, wk);
        const int beg = xw & 65535, end = (int)((unsigned)xw >> 16); // the row's certified interval (beg > end: no cell of the row can lie on an optimal path)
        if (beg > end) {
            if (wave == 0) r_be = lean_wlane(1, wk, r_be);
            if (spf && tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
#pragma unroll
            for (int k = 0; k < C; ++k) { pvh[k] = LCD_GUARD; pva[k] = LCD_GUARD; pvb[k] = LCD_GUARD; }
            l_be2 = l_be; l_be = 1; publish(idx);
            break;
        }
        const int begc = beg & CM, blk0 = begc & ~(BW - 1);
        if (end - blk0 + 2 > WIN) { fail = -1; break; } // the interval would come round to its first wavefront again
        const int cw4 = ((end - begc) + CP) & ~(CP - 1);
        if (cused + (unsigned)cw4 > code_cap || (np > 1 && oused + (unsigned)cw4 > ord_cap) || (spf && nsp >= spill_rows)) { fail = LCD_ERR_CELLS; break; }
        const int pw = (wave - (blk0 >> BSH)) & 3;
        const int j0 = blk0 + BW * pw, j = j0 + cll;
        int pi1 = 0, bz1 = 0, p0 = 0;
        if (np > 1) { pi1 = LCD_RL(w_pi1, wk); bz1 = (pk >> 19) & 31; p0 = LCD_RL(w_p0, wk); }
        // the rows this one reads of its left neighbour: every wavefront but the row's first needs the row itself (gap prefixes); a predecessor's boundary column needs
        // that predecessor.  (Simply: the row itself, or the newest predecessor whose interval holds column j0 - 1.)
        int need_l = bi;
        if (pw > 0) need_l = idx;
        else {
            for (int t = 0; t < np; ++t) {
                int pi = t == 0 ? pi0 : pi1;
                if (t > 1) pi = usgpr(glb_ld(g.pl_pidx + p0 + t));
                int pbe;
                if (idx - pi <= 2) pbe = idx - pi == 1 ? l_be : l_be2;
                else pbe = usgpr(glb_ld(hull + pi)); // (a far row: its interval is the table's)
                const int pb = pbe & 65535, pe = (int)((unsigned)pbe >> 16);
                if (pb <= pe && j0 - 1 >= pb && j0 - 1 <= pe) need_l = smax(need_l, pi);
            }
        }
        int cin1, cin2, bnd_unused;
        const long long tq0 = clock64();
        if (!sync_rows(need_l, idx - AHEAD, idx, idx, cin1, cin2, bnd_unused)) { fail = LCD_ERR_SYNC; break; }
        t_poll += (unsigned long long)(clock64() - tq0);
        if (pw == 0) { cin1 = LCD_GUARD; cin2 = LCD_GUARD; }
        word qw;
        {
            const unsigned a = sq1 + (unsigned)imin(j, qclamp);
            if constexpr (C == 1) qw = (word)*(const lcd_lds_u8 *)(uintptr_t)a;
            else qw = (word)*(const __attribute__((address_space(3))) unsigned short *)(uintptr_t)a;
        }
        bool inb[C]; int sk[C];
#pragma unroll
        for (int k = 0; k < C; ++k) {
            inb[k] = j + k >= beg && j + k <= end;
            const int q = (int)(qw >> (8 * k)) & 255;
            sk[k] = (vb >= 4 || q >= 4) ? 0 : (vb == q ? s_match : s_mism);
        }
        // ---- phase A: best match / E1 / E2 input over the predecessors (first maximum keeps its ordinal); own columns from the ring slot / spill row this thread wrote ----
        int nn[C], uu[C], vv[C];
        word om = 0, oa = 0, ob = 0;
#pragma unroll
        for (int k = 0; k < C; ++k) { nn[k] = LCD_NEG; uu[k] = LCD_NEG; vv[k] = LCD_NEG; }
        {
            bool synced = false;
            for (int t = 0; t < np; ++t) {
                int pi = t == 0 ? pi0 : pi1, bz = t == 0 ? bz0 : bz1;
                if (t > 1) { pi = usgpr(glb_ld(g.pl_pidx + p0 + t)); bz = usgpr(glb_ld(g.pl_bonus + p0 + t)); }
                const bool near = idx - pi <= 2;
                int pb, pe, hm, hv[C], av[C], bv[C];
                if (near) {
                    const int pbe = idx - pi == 1 ? l_be : l_be2; pb = pbe & 65535; pe = (int)((unsigned)pbe >> 16);
                    if (pb > pe) continue;
                    const unsigned SL = ring + 4 * ((pi - bi) & 1) * SLOTW;
                    hm = lds_ld(SL + 4 * ((cl - 1) & WM)); lds_ldc<C>(SL + 4 * cl, hv); lds_ldc<C>(SL + 4 * (WIN + cl), av); lds_ldc<C>(SL + 4 * (2 * WIN + cl), bv);
                    if (j0 - 1 >= pb && j0 - 1 <= pe) { // lane 0: the boundary column from the left neighbour's mailbox
                        const int bh = *(const volatile lcd_lds_i32 *)(uintptr_t)(l_h + (unsigned)((pi & (SYS_D - 1)) * (MAXW * 4)));
                        if (lane == 0) hm = bh;
                    }
                } else { // a far row: its metadata and values were stored to HBM when it was made (and waited for before that row was published)
                    if (!synced) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); synced = true; }
                    { const int pbe = usgpr(glb_ld(hull + pi)); pb = pbe & 65535; pe = (int)((unsigned)pbe >> 16); } // (the table's interval: the row's metadata in HBM is another wavefront's store)
                    if (pb > pe) continue;
                    const int *G = g.spill + (size_t)(unsigned)usgpr(glb_ld((const int *)g.spoff + pi)) * SLOTW; // (every wavefront stores the same spoff: this one reads its own store)
                    hm = glb_ld(G + ((cl - 1) & WM)); glb_ldc<C>(G + cl, hv); glb_ldc<C>(G + WIN + cl, av); glb_ldc<C>(G + 2 * WIN + cl, bv);
                    LCD_PIN(hm);
#pragma unroll
                    for (int k = 0; k < C; ++k) { LCD_PIN(hv[k]); LCD_PIN(av[k]); LCD_PIN(bv[k]); }
                }
                // a slot is addressed by (column mod WIN): entries that belong to other columns of the predecessor's interval, or to none, are fillers here
                if (j - 1 < pb || j - 1 > pe) hm = LCD_GUARD;
#pragma unroll
                for (int k = 0; k < C; ++k) if (j + k < pb || j + k > pe) { hv[k] = LCD_GUARD; av[k] = LCD_GUARD; bv[k] = LCD_GUARD; }
                const int tt = t > 255 ? 255 : t;
#pragma unroll
                for (int k = 0; k < C; ++k) {
                    const int c = (k == 0 ? hm : hv[k - 1]) + sk[k] + bz, a = av[k] + bz, b = bv[k] + bz;
                    if (t == 0) { nn[k] = imax(nn[k], c); uu[k] = imax(uu[k], a); vv[k] = imax(vv[k], b); }
                    else {
                        if (c > nn[k]) { nn[k] = c; om = (om & ~((word)255 << (8 * k))) | ((word)tt << (8 * k)); }
                        if (a > uu[k]) { uu[k] = a; oa = (oa & ~((word)255 << (8 * k))) | ((word)tt << (8 * k)); }
                        if (b > vv[k]) { vv[k] = b; ob = (ob & ~((word)255 << (8 * k))) | ((word)tt << (8 * k)); }
                    }
                }
            }
        }
        // ---- F: A[k] = Hpre[k] + column * e; in-lane inclusive prefix, one scan pair over the lanes, carry from the wavefront to the left ----
        int hp[C], spk[C], a1[C], a2[C], p1[C], p2[C];
        const int je1 = __mul24(j, e1), je2 = __mul24(j, e2);
#pragma unroll
        for (int k = 0; k < C; ++k) {
            hp[k] = imax(nn[k], imax(uu[k], vv[k]));
            spk[k] = nn[k] == hp[k] ? 0 : uu[k] == hp[k] ? 1 : 2;
            a1[k] = inb[k] ? hp[k] + (je1 + k * e1) : LCD_GUARD; a2[k] = inb[k] ? hp[k] + (je2 + k * e2) : LCD_GUARD;
            p1[k] = k ? imax(p1[k - 1], a1[k]) : a1[k]; p2[k] = k ? imax(p2[k - 1], a2[k]) : a2[k];
        }
        int t1 = p1[C - 1], t2 = p2[C - 1];
        scan_max2(t1, t2);
        const int x1 = imax(shr1(LCD_GUARD, t1), cin1), x2 = imax(shr1(LCD_GUARD, t2), cin2);
        // ---- phase B: F, H, E-out, direction code ----
        word code = 0;
#pragma unroll
        for (int k = 0; k < C; ++k) {
            const int pf1 = k ? imax(x1, p1[k - 1]) : x1, pf2 = k ? imax(x2, p2[k - 1]) : x2;
            const int f1 = imax(LCD_NEG, pf1 - (o1 + je1 + k * e1)), f2 = imax(LCD_NEG, pf2 - (o2 + je2 + k * e2));
            const int h = imax(hp[k], imax(f1, f2));
            const int q1 = h - oe1, w1 = uu[k] - e1, q2 = h - oe2, w2 = vv[k] - e2;
            const int eo1 = imax(imax(q1, w1), LCD_NEG), eo2 = imax(imax(q2, w2), LCD_NEG);
            const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;
            const int hs = hp[k] == h ? spk[k] : fk;
            unsigned fl = 0; // O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code
            { const int r2 = a2[k], r1 = a1[k]; LCD_PUSH_GE(fl, q2, w2); LCD_PUSH_GE(fl, q1, w1); LCD_PUSH_GT(fl, pf2, r2); LCD_PUSH_GT(fl, pf1, r1); }
            const unsigned cd = (unsigned)hs | (fl << 3) | (((om >> (8 * k)) & 255) ? CB_PM : 0);
            code |= (word)cd << (8 * k);
            pvh[k] = inb[k] ? h : LCD_GUARD; pva[k] = inb[k] ? eo1 : LCD_GUARD; pvb[k] = inb[k] ? eo2 : LCD_GUARD;
        }
        // ---- stores: ring slot (values), HBM (codes; values only for rows a far successor / the end node will read), mailbox ----
        {
            const unsigned SL = ring + 4 * ((idx - bi) & 1) * SLOTW;
            lds_stc<C>(SL + 4 * cl, pvh); lds_stc<C>(SL + 4 * (WIN + cl), pva); lds_stc<C>(SL + 4 * (2 * WIN + cl), pvb);
            if (spf) { int *G = g.spill + (size_t)nsp * SLOTW; glb_stc<C>(G + cl, pvh); glb_stc<C>(G + WIN + cl, pva); glb_stc<C>(G + 2 * WIN + cl, pvb); }
            const int cj = j - begc;
            if ((unsigned)cj < (unsigned)cw4) {
                uint8_t *cp = g.code8 + (size_t)(cused + (unsigned)cj);
                if constexpr (C == 2) *(__attribute__((address_space(1))) unsigned short *)cp = (unsigned short)code;
                else *(__attribute__((address_space(1))) uint8_t *)cp = (uint8_t)code;
                if (np > 1) {
                    int ow[C];
#pragma unroll
                    for (int k = 0; k < C; ++k) ow[k] = (int)((om >> (8 * k)) & 255) | ((int)((oa >> (8 * k)) & 255) << 8) | ((int)((ob >> (8 * k)) & 255) << 16);
                    glb_stc<C>(g.ord + (size_t)(oused + (unsigned)cj), ow);
                }
            }
            if (lane == 63) {
                const unsigned mo = (unsigned)((idx & (SYS_D - 1)) * (MAXW * 4));
                *(lcd_lds_i32 *)(uintptr_t)(m_h + mo) = pvh[C - 1];
                *(lcd_lds_i32 *)(uintptr_t)(m_c1 + mo) = imax(cin1, t1); *(lcd_lds_i32 *)(uintptr_t)(m_c2 + mo) = imax(cin2, t2);
            }
        }
        const int be = beg | (end << 16);
        if (wave == 0) { r_be = lean_wlane(be, wk, r_be); r_off = lean_wlane((int)cused, wk, r_off); }
        l_be2 = l_be; l_be = be;
        if ((np > 1 || spf) && lane == 0) { // (all four wavefronts, the same values: each reads back only what it stored itself)
            if (np > 1) glb_st(g.ooff + idx, (int)oused);
            if (spf) { glb_st(g.rbeg + idx, beg); glb_st(g.rend + idx, end); glb_st(g.ml + idx, 0); glb_st(g.mr + idx, 0); glb_st(g.spoff + idx, nsp); }
        }
        cused += (unsigned)cw4; if (np > 1) oused += (unsigned)cw4;
        if (spf) { ++nsp; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // (a row that others will read from HBM: there before it is published)
        ncell += (unsigned long long)(end - beg + 1);
        publish(idx);
        } while (0);
        if (fail != 0) break;
        ++idx;
    }
    if (fail != 0) { // nobody may wait for this wavefront any more; the others reach the same row and the same decision
        if (lane == 63) *(volatile lcd_lds_i32 *)(uintptr_t)a_me = 0x7ffffff0;
        if (fail == LCD_ERR_SYNC && lane == 0) sm.bc[7] = LCD_ERR_SYNC;
    }
    if (fail == 0) flush_meta(wbase, ei - wbase);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (sm.bc[7] != LCD_OK) fail = sm.bc[7]; // (a wavefront that timed out waiting for another: everybody leaves with the error)
    __syncthreads();
    if (fail < 0) return -1;
    if (fail > 0) { wo->status = fail; return 0; }
    wo->cells = ncell;
    const long long t_bt0 = clock64();
    wo->t_dp = (unsigned long long)(t_bt0 - t_dp0); wo->t_poll = t_poll;
    if (wave == 0) code_backtrack(g, sm, pd, bi, ei, qlen, SLOTW, WM, lane, CM);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    const int n_cig = sm.bc[0];
    wo->status = sm.bc[1];
    wo->cig_pos = sm.bc[4];
    wo->score = sm.bc[5];
    __syncthreads();
    wo->t_bt = (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

// BAND: the rows are restricted to the certified intervals of the table (g.cert hull, align_certified_sys): a wavefront whose 256 columns miss a row's interval
// only publishes fillers to its mailboxes (a few dozen instructions instead of a row), cells of an active wavefront outside the interval are stored as fillers,
// and a row's codes / ordinals in HBM start at the interval's 4-cell group (rbeg / rend / roff say where, as for the windowed rows).
template <int NT, bool BAND = false>
__device__ __attribute__((noinline)) int align_unbanded(const Ctx *gp_, const unsigned ring_, const unsigned pd_ /* 0xffffffff: none */, const LcdScoring sc_,
                                                        const int bi_, const int ei_, const uint8_t *seq_hbm_, const int qlen_, WinOut *wo_) {
    constexpr int NW = NT / 64, K = Cfg<NT>::K;
    constexpr bool SYS = NW > 1;
    Smem &sm = g_smem;
    Ctx g = *usgpr(gp_); // (by pointer: a by-value context is 440 B of outgoing-argument stack per call site and per lane)
    ctx_to_sgpr(g);
    const unsigned ring = usgpr(ring_), pd = usgpr(pd_);
    const int bi = usgpr(bi_), ei = usgpr(ei_), qlen = usgpr(qlen_);
    const uint8_t *seq_hbm = usgpr(seq_hbm_); WinOut *wo = usgpr(wo_);
    LcdScoring sc; sc.match = usgpr(sc_.match); sc.mismatch = usgpr(sc_.mismatch); sc.o1 = usgpr(sc_.o1); sc.e1 = usgpr(sc_.e1); sc.o2 = usgpr(sc_.o2); sc.e2 = usgpr(sc_.e2); sc.dbg = usgpr(sc_.dbg);
    constexpr int WIN = 4 * NT, WM = WIN - 1, SLOTW = 3 * WIN;
    const int tid = threadIdx.x, lane = tid & 63, wave = usgpr(tid >> 6); // (wave in an SGPR: branches on it are scalar)
    const int o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2;
    // Reads longer than the window are swept in COLUMN TILES of WIN columns (1 024-thread class only): tile t runs every row over the columns
    // [t * WIN, (t + 1) * WIN); what crosses a tile boundary is, per row, H of the tile's last column (the diagonal input of the next tile's first
    // column) and the two running F prefixes -- three ints per row through HBM (g.tb), exactly what the wavefronts of one tile hand each other
    // through the LDS mailboxes.  Codes / ordinals are stored row-major over the WHOLE read, so the code-driven backtrack does not know about
    // tiles; spilled value rows are per tile and only the last tile's are read afterwards (the end node looks at column qlen).
    const int n_tiles = qlen + 2 > WIN ? (qlen >> (NT == 1024 ? 12 : 30)) + 1 : 1;
    if (n_tiles > 1 && (NT != 1024 || BAND)) return -1;
    if (BAND && qlen >= 65535) return -1;
    const int *const hull = g.cert + 6 * (size_t)g.node_cap;
    const int seg_lo = 256 * wave, seg_hi = seg_lo + 255; // this wavefront's columns (BAND: single tile)
    const int jl = 4 * tid;                  // first of this lane's four columns inside the tile
    const int cw4_full = ((qlen >> 2) + 1) << 2;  // cells of a full row in HBM, padded to the lanes' 4-cell groups
    const unsigned long long code_cap = g.cell_cap, ord_cap = g.spill_x > 2 ? g.cell_cap : g.cell_cap / 4;
    const long long spill_rows = g.cell_cap * g.spill_x > 64 ? (long long)((g.cell_cap * g.spill_x - 64) / ((unsigned long long)SLOTW * 4)) : 0;
    {
        int has_n = 0;
        for (int j = tid; j < qlen; j += NT) has_n |= seq_hbm[j] >= 4;
        if (__syncthreads_or(has_n)) return -1; // reads with N bases (score 0 against everything) take the windowed rows
    }
    int *const tb_h[2] = {g.tb, g.tb + g.node_cap}; int *const tb_c1 = g.tb + 2 * (size_t)g.node_cap, *const tb_c2 = g.tb + 3 * (size_t)g.node_cap;
    unsigned long long ncell = 0, t_plan = 0, t_poll = 0;
    int AW = 1;
    const long long t_dp0 = clock64();
  for (int tile = 0; tile < n_tiles; ++tile) {
    const int tbase = tile * WIN, jb = tbase + jl, par = tile & 1;
    const bool more = tile + 1 < n_tiles;
    AW = imin(NW, ((imin(qlen, tbase + WIN - 1) - tbase) >> 8) + 1); // wavefronts that own a column <= qlen in this tile
    // per-lane constants of this read
    // (one packed register: q[jb-1], q[jb], q[jb+1], q[jb+2] in bytes 0..3; columns outside the read get 15, which matches no base)
    unsigned qpk = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = jb + k;
        int b = 15;
        if (j >= 1 && j <= qlen) b = seq_hbm[j - 1];
        qpk |= (unsigned)b << (8 * k);
    }
    const int je1 = jb * e1, je2 = jb * e2;            // A[k] = Hpre[k] + (jb + k) * e
    const int noj1 = -(o1 + je1), noj2 = -(o2 + je2);  // F[k] = prefix + noj - k * e
    const unsigned lofs = 16u * (unsigned)tid;         // this lane's 4 cells inside a plane (bytes; tile-local)
    const unsigned PL = 4u * (unsigned)WIN;            // plane stride (bytes)
    int nsp = 0;
    int src_be = qlen << 16; // (BAND: the source row's interval)
    // ---- source row (slot 0) ----
    {
        const bool spf = (g.imap[bi] & 2) != 0;
        if (spf && spill_rows < 1) { wo->status = LCD_ERR_CELLS; return 0; }
        int end0 = qlen;
        if (BAND) { const int hw = usgpr(glb_ld(hull + bi)); end0 = hw >> 16; if ((hw & 65535) != 0) return -1; src_be = end0 << 16; }
        int hh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = jb + k;
            const int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            hh[k] = j <= qlen ? (j ? imax(f1, f2) : 0) : LCD_GUARD;
            if (BAND && j > end0) hh[k] = LCD_GUARD;
        }
        const int4 H4 = make_int4(hh[0], hh[1], hh[2], hh[3]);
        int4 A4 = make_int4(hh[0] - o1, hh[1] - o1, hh[2] - o1, hh[3] - o1), B4 = make_int4(hh[0] - o2, hh[1] - o2, hh[2] - o2, hh[3] - o2);
        if (BAND) { // (fillers stay fillers)
            if (jb > end0) { A4.x = LCD_GUARD; B4.x = LCD_GUARD; } if (jb + 1 > end0) { A4.y = LCD_GUARD; B4.y = LCD_GUARD; }
            if (jb + 2 > end0) { A4.z = LCD_GUARD; B4.z = LCD_GUARD; } if (jb + 3 > end0) { A4.w = LCD_GUARD; B4.w = LCD_GUARD; }
        }
        {
            lds_st4(ring + lofs, H4); lds_st4(ring + PL + lofs, A4); lds_st4(ring + 2 * PL + lofs, B4);
            if (spf) { int *G = g.spill; glb_st4(G + jl, H4); glb_st4(G + WIN + jl, A4); glb_st4(G + 2 * WIN + jl, B4); }
        }
        if (SYS && lane == 63) { g_wide.bndH[bi & (SYS_D - 1)][wave] = H4.w; g_wide.prog[wave] = bi; }
        if (more && tid == NT - 1) glb_st(tb_h[par] + bi, H4.w); // the source row's H at the tile's last column
        if (tid == 0) { g.rbeg[bi] = 0; g.rend[bi] = end0; g.roff[bi] = 0; g.ml[bi] = 0; g.mr[bi] = 0; g.spoff[bi] = 0; sm.bc[7] = LCD_OK; }
        if (spf) nsp = 1;
    }
    unsigned long long cused = 0, oused = 0;
    const unsigned long long tcols = (unsigned long long)(imin(qlen, tbase + WIN - 1) - tbase + 1);
    ncell += BAND ? (unsigned long long)(src_be >> 16) + 1 : tcols;
    __syncthreads();
    int err = LCD_OK;
    if (wave < AW) {
        int wbase = -(1 << 20);
        int w_x = 0;     // BAND: the rows' intervals (lo | hi << 16)
        int m_be = 1;    // BAND: lane s = interval of the row in ring slot s (lo > hi: none)
        if (BAND) m_be = lane == 0 ? src_be : 1;
        int w_pk = 0, w_pi0 = 0, w_pi1 = 0; // w_pk: #preds (16 bits) | base << 16 | spill << 19 | unreachable << 20 | bonus0 << 21 | bonus1 << 26
        const int ke1 = e1, ke2 = e2;
        for (int idx = bi + 1; idx < ei; ++idx) {
            if (idx - wbase >= 64) { // plan window: each lane loads the plan of one upcoming row; rows then take it by v_readlane
                if ((unsigned long long)clock64() > g.wd_deadline) { err = LCD_ERR_WATCHDOG; break; } // (every wavefront on its own: the others run into it, or into their poll bound)
                const long long tp0 = clock64();
                wbase = idx;
                const int ri = idx + lane;
                w_pk = 1 << 20;
                if (ri < ei) {
                    const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
                    const int cnt = s1 - s0;
                    int b0 = 0, b1 = 0;
                    if (cnt > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); b0 = glb_ld(g.pl_bonus + s0); }
                    if (cnt > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); b1 = glb_ld(g.pl_bonus + s0 + 1); }
                    if (BAND) w_x = glb_ld(hull + ri);
                    w_pk = imin(cnt, 65535) | (glb_ld_u8(g.pl_base + ri) << 16) | ((glb_ld_u8(g.imap + ri) & 2) << 18) | (glb_ld(g.pl_rem + ri) == (1 << 30) ? 1 << 20 : 0) | (b0 << 21) | (b1 << 26);
                }
                LCD_PIN(w_pk); LCD_PIN(w_pi0); LCD_PIN(w_pi1); if (BAND) LCD_PIN(w_x);
                t_plan += (unsigned long long)(clock64() - tp0);
            }
            const int wk = idx - wbase;
            const int pk = LCD_RL(w_pk, wk), pi0 = LCD_RL(w_pi0, wk), pi1 = LCD_RL(w_pi1, wk);
            const int np = pk & 65535, vb = (pk >> 16) & 7, bz0 = (pk >> 21) & 31, bz1 = (pk >> 26) & 31;
            const bool spf = (pk >> 19) & 1;
            if ((pk >> 20) & 1) { // not reachable from the begin node (sub-graph alignments only)
                if (tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
                if (BAND) m_be = lean_wlane(1, (idx - bi) & (K - 1), m_be);
                if (SYS && lane == 63) *(volatile lcd_lds_i32 *)(uintptr_t)lds_off(&g_wide.prog[wave]) = idx; // (a row nobody reads is done as soon as it is passed: a long run of them must not look like a stalled neighbour)
                continue;
            }
            int beg = 0, end = qlen, begc = 0;
            if (BAND) { const int xw = LCD_RL(w_x, wk); beg = xw & 65535; end = (int)((unsigned)xw >> 16); begc = beg & ~3; }
            const int cw4 = BAND ? ((end - begc) + 4) & ~3 : cw4_full;
            if (BAND && beg > end) { // empty row: no cell of it can lie on an optimal path
                if (tid == 0) { glb_st(g.rbeg + idx, 1); glb_st(g.rend + idx, 0); }
                m_be = lean_wlane(1, (idx - bi) & (K - 1), m_be);
                if (SYS && lane == 63) *(volatile lcd_lds_i32 *)(uintptr_t)lds_off(&g_wide.prog[wave]) = idx;
                continue; // (nobody reads this row's mailboxes: every reader looks at the row's interval first)
            }
            // every wavefront takes the same decisions from the same plan, so an error leaves the loop in all of them at the same row
            if (cused + cw4 > code_cap || (np > 1 && oused + cw4 > ord_cap) || (spf && nsp >= spill_rows)) { err = LCD_ERR_CELLS; break; }
            if (np > 256) { err = LCD_FALLBACK; break; } // ordinals are 8 bits (and the packed plan word holds 16): generic rows
            const int s = (idx - bi) & (K - 1);
            if (BAND && (seg_lo > end || seg_hi < begc)) { // none of this wavefront's columns is in the row's interval: fillers to the mailboxes, the row's bookkeeping, on
                if (SYS) {
                    if (wave + 1 < AW && !poll_ge(&g_wide.prog[wave + 1], idx - (SYS_D - K - 1))) { err = LCD_ERR_SYNC; if (lane == 0) g_smem.bc[7] = LCD_ERR_SYNC; break; }
                    if (lane == 63) { g_wide.bndH[idx & (SYS_D - 1)][wave] = LCD_GUARD; g_wide.carry1[idx & (SYS_D - 1)][wave] = LCD_GUARD; g_wide.carry2[idx & (SYS_D - 1)][wave] = LCD_GUARD; }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 63) *(volatile lcd_lds_i32 *)(uintptr_t)lds_off(&g_wide.prog[wave]) = idx;
                }
                if (tid == 0) {
                    glb_st(g.rbeg + idx, beg); glb_st(g.rend + idx, end); glb_st(g.roff + idx, (int)cused); glb_st(g.ooff + idx, (int)oused);
                    if (spf) { glb_st(g.ml + idx, 0); glb_st(g.mr + idx, 0); glb_st(g.spoff + idx, nsp); }
                }
                m_be = lean_wlane(beg | (end << 16), s, m_be);
                cused += cw4; if (np > 1) oused += cw4; if (spf) ++nsp;
                ncell += (unsigned long long)(end - beg + 1);
                continue;
            }
            const long long tq0 = clock64();
            if (SYS) {
                bool ok = true;
                if (wave + 1 < AW) ok = poll_ge(&g_wide.prog[wave + 1], idx - (SYS_D - K - 1)); // mailbox slot (idx mod SYS_D) is free again
                if (ok && wave > 0) ok = poll_ge(&g_wide.prog[wave - 1], idx);                   // left neighbour has published this row
                if (!ok) { err = LCD_ERR_SYNC; if (lane == 0) g_smem.bc[7] = LCD_ERR_SYNC; break; }
                asm volatile("" ::: "memory");
            }
            t_poll += (unsigned long long)(clock64() - tq0);
            const int d0 = idx - pi0, d1 = idx - pi1;
            int s0, s1, s2, s3;
            if (vb < 4) {
                const int mt = sc.match, mm = -sc.mismatch;
                s0 = (int)(qpk & 255) == vb ? mt : mm; s1 = (int)((qpk >> 8) & 255) == vb ? mt : mm;
                s2 = (int)((qpk >> 16) & 255) == vb ? mt : mm; s3 = (int)(qpk >> 24) == vb ? mt : mm;
            } else { s0 = s1 = s2 = s3 = 0; } // N in the graph scores 0 against everything
            // ---- phase A: best match / E1 / E2 input of the four cells ----
            int n0, n1, n2, n3, u0, u1, u2, u3, v0, v1, v2, v3;
            int om = 0, oa = 0, ob = 0; // ordinals of the first maximum, one byte per cell
            auto fill_row = [&](int &hm, int4 &hv, int4 &av, int4 &bv) { hm = LCD_GUARD; hv = make_int4(LCD_GUARD, LCD_GUARD, LCD_GUARD, LCD_GUARD); av = hv; bv = hv; };
            auto near_row = [&](const int pi, int &hm, int4 &hv, int4 &av, int4 &bv) {
                if (BAND) { // the slot holds this wavefront's columns of row pi only if they met that row's interval
                    const int be = LCD_RL(m_be, (pi - bi) & (K - 1)), pb = be & 65535, pe = (int)((unsigned)be >> 16);
                    if (pb > pe || seg_lo > pe || seg_hi < (pb & ~3)) { fill_row(hm, hv, av, bv); return; }
                }
                const unsigned S = ring + 4u * (unsigned)(((pi - bi) & (K - 1)) * SLOTW);
                hm = lds_ld(S + lofs - 4); hv = lds_ld4(S + lofs); av = lds_ld4(S + PL + lofs); bv = lds_ld4(S + 2 * PL + lofs);
                if (lane == 0) hm = wave == 0 ? (tile ? glb_ld(tb_h[par ^ 1] + pi) : LCD_GUARD) : (SYS ? g_wide.bndH[pi & (SYS_D - 1)][wave - 1] : hm);
            };
            auto far_row = [&](const int pi, int &hm, int4 &hv, int4 &av, int4 &bv) {
                int pb = 0, pe = qlen;
                if (BAND) {
                    const int be = usgpr(glb_ld(hull + pi)); pb = be & 65535; pe = (int)((unsigned)be >> 16);
                    if (pb > pe || seg_lo > pe || seg_hi < (pb & ~3)) { fill_row(hm, hv, av, bv); return; }
                }
                const int *G = g.spill + (size_t)(unsigned)glb_ld((const int *)g.spoff + pi) * SLOTW;
                hm = tid ? glb_ld(G + jl - 1) : (tile ? glb_ld(tb_h[par ^ 1] + pi) : LCD_GUARD); hv = glb_ld4(G + jl); av = glb_ld4(G + WIN + jl); bv = glb_ld4(G + 2 * WIN + jl);
                LCD_PIN(hm); LCD_PIN(hv.x); LCD_PIN(hv.y); LCD_PIN(hv.z); LCD_PIN(hv.w); LCD_PIN(av.x); LCD_PIN(av.y); LCD_PIN(av.z); LCD_PIN(av.w);
                LCD_PIN(bv.x); LCD_PIN(bv.y); LCD_PIN(bv.z); LCD_PIN(bv.w);
                if (BAND && (jb - 1 < pb || jb - 1 > pe)) hm = LCD_GUARD; // (lane 0: the left neighbour's column, written only if it met the row's interval)
            };
            if (np == 1 && d0 <= K) {
                int hm; int4 hv, av, bv;
                near_row(pi0, hm, hv, av, bv);
                const int be1 = bz0 - e1, be2 = bz0 - e2;
                n0 = hm + s0 + bz0; n1 = hv.x + s1 + bz0; n2 = hv.y + s2 + bz0; n3 = hv.z + s3 + bz0;
                u0 = av.x + be1; u1 = av.y + be1; u2 = av.z + be1; u3 = av.w + be1;
                v0 = bv.x + be2; v1 = bv.y + be2; v2 = bv.z + be2; v3 = bv.w + be2;
            } else {
                if ((np > 0 && d0 > K) || (np > 1 && d1 > K) || np > 2) { // a far row may be read from HBM: its stores (own ones included) must have landed
                    if (SYS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else __syncthreads(); // (systolic: the neighbours waited before publishing a spilled row)
                }
                n0 = n1 = n2 = n3 = u0 = u1 = u2 = u3 = v0 = v1 = v2 = v3 = LCD_NEG;
                int p0 = 0;
                if (np > 2) { p0 = glb_ld(g.pl_start + idx); LCD_PIN(p0); }
                for (int t = 0; t < np; ++t) {
                    int pi = t == 0 ? pi0 : pi1, bz = t == 0 ? bz0 : bz1;
                    if (t > 1) { pi = glb_ld(g.pl_pidx + p0 + t); bz = glb_ld(g.pl_bonus + p0 + t); LCD_PIN(pi); LCD_PIN(bz); }
                    int hm; int4 hv, av, bv;
                    if (idx - pi <= K) near_row(pi, hm, hv, av, bv); else far_row(pi, hm, hv, av, bv);
                    const int be1 = bz - e1, be2 = bz - e2;
                    const int c0 = hm + s0 + bz, c1 = hv.x + s1 + bz, c2 = hv.y + s2 + bz, c3 = hv.z + s3 + bz;
                    const int a0 = av.x + be1, a1 = av.y + be1, a2 = av.z + be1, a3 = av.w + be1;
                    const int b0 = bv.x + be2, b1 = bv.y + be2, b2 = bv.z + be2, b3 = bv.w + be2;
                    const int tt = t > 255 ? 255 : t;
#define LCD_UPD(cur, cand, ordv, sh) if ((cand) > (cur)) { cur = (cand); ordv = (ordv & ~(255 << (sh))) | (tt << (sh)); }
                    LCD_UPD(n0, c0, om, 0) LCD_UPD(n1, c1, om, 8) LCD_UPD(n2, c2, om, 16) LCD_UPD(n3, c3, om, 24)
                    LCD_UPD(u0, a0, oa, 0) LCD_UPD(u1, a1, oa, 8) LCD_UPD(u2, a2, oa, 16) LCD_UPD(u3, a3, oa, 24)
                    LCD_UPD(v0, b0, ob, 0) LCD_UPD(v1, b1, ob, 8) LCD_UPD(v2, b2, ob, 16) LCD_UPD(v3, b3, ob, 24)
#undef LCD_UPD
                }
            }
            const int h0 = imax(n0, imax(u0, v0)), h1 = imax(n1, imax(u1, v1)), h2 = imax(n2, imax(u2, v2)), h3 = imax(n3, imax(u3, v3)); // Hpre
            // which of match / E1 / E2 gives Hpre (the oracle's priority): used when H == Hpre
            const int sp0 = n0 == h0 ? 0 : u0 == h0 ? 1 : 2, sp1 = n1 == h1 ? 0 : u1 == h1 ? 1 : 2, sp2 = n2 == h2 ? 0 : u2 == h2 ? 1 : 2, sp3 = n3 == h3 ? 0 : u3 == h3 ? 1 : 2;
            // ---- F: A[k] = Hpre[k] + k*e; in-lane inclusive prefix, one scan pair over the lane totals, carry from the left wavefront ----
            const bool i0 = !BAND || (jb >= beg && jb <= end), i1 = !BAND || (jb + 1 >= beg && jb + 1 <= end), i2 = !BAND || (jb + 2 >= beg && jb + 2 <= end), i3 = !BAND || (jb + 3 >= beg && jb + 3 <= end);
            const int a10 = i0 ? h0 + je1 : LCD_GUARD, a11 = i1 ? h1 + je1 + ke1 : LCD_GUARD, a12 = i2 ? h2 + je1 + 2 * ke1 : LCD_GUARD, a13 = i3 ? h3 + je1 + 3 * ke1 : LCD_GUARD;
            const int a20 = i0 ? h0 + je2 : LCD_GUARD, a21 = i1 ? h1 + je2 + ke2 : LCD_GUARD, a22 = i2 ? h2 + je2 + 2 * ke2 : LCD_GUARD, a23 = i3 ? h3 + je2 + 3 * ke2 : LCD_GUARD;
            const int p10 = a10, p11 = imax(p10, a11), p12 = imax(p11, a12);
            const int p20 = a20, p21 = imax(p20, a21), p22 = imax(p21, a22);
            int t1 = imax(p12, a13), t2 = imax(p22, a23);
            scan_max2(t1, t2);
            int x1 = shr1(LCD_GUARD, t1), x2 = shr1(LCD_GUARD, t2); // exclusive prefix over the lanes of this wavefront
            int cin1 = LCD_GUARD, cin2 = LCD_GUARD;
            if (SYS && wave > 0) { cin1 = g_wide.carry1[idx & (SYS_D - 1)][wave - 1]; cin2 = g_wide.carry2[idx & (SYS_D - 1)][wave - 1]; x1 = imax(x1, cin1); x2 = imax(x2, cin2); }
            else if (tile) { cin1 = glb_ld(tb_c1 + idx); cin2 = glb_ld(tb_c2 + idx); LCD_PIN(cin1); LCD_PIN(cin2); x1 = imax(x1, cin1); x2 = imax(x2, cin2); } // the row's F prefixes over the tiles before
            // ---- phase B: F, H, E-out, direction code of the four cells ----
            int hh0, hh1, hh2, hh3, ea0, ea1, ea2, ea3, eb0, eb1, eb2, eb3;
            unsigned code = 0;
#define LCD_CELL(k, hp, spk, ev1, ev2, pf1, pf2, ak1, ak2, HO, AO, BO)                                                        \
            {                                                                                                               \
                const int f1 = (pf1) + noj1 - (k) * ke1, f2 = (pf2) + noj2 - (k) * ke2;                                     \
                const int h = imax(hp, imax(f1, f2));                                                                       \
                const int ho1 = h - o1, ho2 = h - o2;                                                                       \
                const int fk = f1 == h ? (f2 == h ? 5 : 3) : 4;                                                             \
                unsigned fl = 0; /* O2, O1, Y2, Y1 pushed in this order = bits 6, 5, 4, 3 of the code */                    \
                { const int w2 = (ev2), w1 = (ev1), q2 = (pf2), r2 = (ak2), q1 = (pf1), r1 = (ak1);                         \
                  LCD_PUSH_GE(fl, ho2, w2); LCD_PUSH_GE(fl, ho1, w1); LCD_PUSH_GT(fl, q2, r2); LCD_PUSH_GT(fl, q1, r1); }   \
                const unsigned cd = (unsigned)((hp) == h ? (spk) : fk) | (fl << 3) | (((om >> (8 * (k))) & 255) ? CB_PM : 0); \
                code |= cd << (8 * (k));                                                                                    \
                HO = h; AO = imax(ho1, ev1); BO = imax(ho2, ev2);                                                           \
            }
            LCD_CELL(0, h0, sp0, u0, v0, x1, x2, a10, a20, hh0, ea0, eb0)
            LCD_CELL(1, h1, sp1, u1, v1, imax(x1, p10), imax(x2, p20), a11, a21, hh1, ea1, eb1)
            LCD_CELL(2, h2, sp2, u2, v2, imax(x1, p11), imax(x2, p21), a12, a22, hh2, ea2, eb2)
            LCD_CELL(3, h3, sp3, u3, v3, imax(x1, p12), imax(x2, p22), a13, a23, hh3, ea3, eb3)
#undef LCD_CELL
            if (BAND) { // cells outside the interval are fillers; cells inside keep to the value range (>= LCD_NEG: a cell whose predecessors are all fillers)
                hh0 = i0 ? imax(hh0, LCD_NEG) : LCD_GUARD; ea0 = i0 ? imax(ea0, LCD_NEG) : LCD_GUARD; eb0 = i0 ? imax(eb0, LCD_NEG) : LCD_GUARD;
                hh1 = i1 ? imax(hh1, LCD_NEG) : LCD_GUARD; ea1 = i1 ? imax(ea1, LCD_NEG) : LCD_GUARD; eb1 = i1 ? imax(eb1, LCD_NEG) : LCD_GUARD;
                hh2 = i2 ? imax(hh2, LCD_NEG) : LCD_GUARD; ea2 = i2 ? imax(ea2, LCD_NEG) : LCD_GUARD; eb2 = i2 ? imax(eb2, LCD_NEG) : LCD_GUARD;
                hh3 = i3 ? imax(hh3, LCD_NEG) : LCD_GUARD; ea3 = i3 ? imax(ea3, LCD_NEG) : LCD_GUARD; eb3 = i3 ? imax(eb3, LCD_NEG) : LCD_GUARD;
            }
            // ---- stores: ring slot (values), HBM (codes; values only for rows a far successor / the end node will read) ----
            const int4 H4 = make_int4(hh0, hh1, hh2, hh3), A4 = make_int4(ea0, ea1, ea2, ea3), B4 = make_int4(eb0, eb1, eb2, eb3);
            {
                const unsigned S = ring + 4u * (unsigned)(s * SLOTW);
                lds_st4(S + lofs, H4); lds_st4(S + PL + lofs, A4); lds_st4(S + 2 * PL + lofs, B4);
                if (spf) { int *G = g.spill + (size_t)nsp * SLOTW; glb_st4(G + jl, H4); glb_st4(G + WIN + jl, A4); glb_st4(G + 2 * WIN + jl, B4); }
                if (BAND ? (jb >= begc && jb - begc < cw4) : jb < cw4) {
                    const int jo = jb - begc; // (begc = 0 without a band)
                    glb_st(g.code8 + cused + jo, (int)code);
                    if (np > 1) glb_st4(g.ord + oused + jo, make_int4((om & 255) | ((oa & 255) << 8) | ((ob & 255) << 16),
                                                                         ((om >> 8) & 255) | (((oa >> 8) & 255) << 8) | (((ob >> 8) & 255) << 16),
                                                                         ((om >> 16) & 255) | (((oa >> 16) & 255) << 8) | (((ob >> 16) & 255) << 16),
                                                                         ((om >> 24) & 255) | (((oa >> 24) & 255) << 8) | (((ob >> 24) & 255) << 16)));
                }
            }
            if (tid == 0) {
                glb_st(g.rbeg + idx, beg); glb_st(g.rend + idx, end); glb_st(g.roff + idx, (int)cused); glb_st(g.ooff + idx, (int)oused);
                if (spf) { glb_st(g.ml + idx, 0); glb_st(g.mr + idx, 0); glb_st(g.spoff + idx, nsp); }
            }
            if (SYS) {
                if (lane == 63) {
                    g_wide.bndH[idx & (SYS_D - 1)][wave] = hh3;
                    g_wide.carry1[idx & (SYS_D - 1)][wave] = imax(cin1, t1); g_wide.carry2[idx & (SYS_D - 1)][wave] = imax(cin2, t2);
                    if (more && wave == NW - 1) { glb_st(tb_h[par] + idx, hh3); glb_st(tb_c1 + idx, imax(cin1, t1)); glb_st(tb_c2 + idx, imax(cin2, t2)); } // -> the next tile
                }
                // the row is complete in LDS (and, for a spilled row, in HBM) before the neighbours may look at it
                if (spf) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 63) *(volatile lcd_lds_i32 *)(uintptr_t)lds_off(&g_wide.prog[wave]) = idx; // (LDS-typed: a generic volatile store is a flat store + vmcnt(0))
            }
            if (BAND) m_be = lean_wlane(beg | (end << 16), s, m_be);
            cused += cw4; if (np > 1) oused += cw4; if (spf) ++nsp;
            ncell += BAND ? (unsigned long long)(end - beg + 1) : tcols;
        }
    }
    if (tid == 0 && err != LCD_OK) sm.bc[7] = err;
    __syncthreads(); // (also: every store of this tile -- boundary columns, spilled rows -- has landed before the next tile reads them)
    if (sm.bc[7] != LCD_OK) { const int e = sm.bc[7]; __syncthreads(); if (e == LCD_FALLBACK) return -1; wo->status = e; return 0; }
  } // tiles
    wo->cells = ncell;
    if (tid == (AW - 1) * 64) { sm.prof[0] = t_plan; sm.prof[1] = t_poll; } // (profiling aid: the LAST active wavefront's view)
    const long long t_bt0 = clock64();
    wo->t_dp = (unsigned long long)(t_bt0 - t_dp0);
    if (wave == 0) code_backtrack(g, sm, pd, bi, ei, qlen, SLOTW, WM, lane, ~3);
    __syncthreads();
    const int n_cig = sm.bc[0];
    wo->status = sm.bc[1];
    wo->cig_pos = sm.bc[4];
    wo->score = sm.bc[5];
    wo->t_plan = sm.prof[0]; wo->t_poll = sm.prof[1];
    __syncthreads();
    wo->t_bt = (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

// ================= certified band for the unbanded K2 alignments (single-wavefront class) =================
// The oracle's K2 rows span the whole read (wb = -1).  Its backtrack, though, only ever visits cells that lie on an OPTIMAL alignment path, and it takes
// the same decisions there in any DP that (a) contains every cell of every optimal path and (b) treats the cells it leaves out as unreachable: a value
// on an optimal path is reached through predecessors on optimal paths, a candidate that loses a comparison in the full DP can only lose by more when it
// is under-estimated, and a candidate that ties is on an optimal path itself (DESIGN.md "Certified band" has the argument incl. the insertion-run flags).
// A cell (v, j) is provably on NO optimal path if an upper bound on every alignment through it is below the score S of some alignment:
//   UB(v, j) = Bp(v) + Bs(v) + max(o1, o2)
//            + M * min(j, maxD(v)) - G(j - maxD(v)) - G(minD(v) - j)                      (prefix: j bases against a source..v path of minD..maxD nodes)
//            + M * min(r, maxR(v)) - G(r - maxR(v)) - G(minR(v) - r),   r = qlen - j       (suffix)
// with M the match score, G(k) = min(o1 + e1 k, o2 + e2 k) for k > 0 (the cheapest way to pay k forced gap columns; G is concave, so splitting them costs
// more -- except for ONE run cut in two by the cell itself, hence the max(o1, o2)), Bp / Bs the largest sums of edge bonuses (ilog2 of the weight,
// inc_path_score) over source..v / v..sink paths.  The rows' intervals [lo, hi] = hull{j : UB(v, j) >= S_est} are computed BEFORE the DP from a guess
// S_est (the bound at the end cell minus the largest slack the chain's earlier reads needed, plus a margin); the DP over the intervals returns S.  If
// S >= S_est the guess was a true lower bound of the optimum and the alignment is the oracle's; otherwise S itself is one and the read is redone with
// S_est = S, which cannot fail.  oracle/poa.c carries the same bound as a checker (LCDO_CERT_STATS; tests/test_oracle_poa.py).
// Node arrays by topological index in g.cert: minD | maxD | Bp | minR | maxR | Bs | hull (lo | hi << 16).
constexpr int CERT_INF = 1 << 29;
__device__ __forceinline__ int wlane(const int val, const int l, int old) { // old with lane l replaced by val (both wave-uniform)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(val), "s"(l) : "m0"); // (gfx9: one SGPR per VALU instruction; the lane select goes through M0)
    return old;
}
__device__ __forceinline__ int cert_G(const int k, const int o1, const int e1, const int o2, const int e2) { return k <= 0 ? 0 : imin(o1 + e1 * k, o2 + e2 * k); }
// forward / backward passes over the rows.  One wavefront; the rows are walked one by one with the values of the current and the previous 64-row block in
// registers (lane = row - block base), so a predecessor / successor within 64..127 rows costs a v_readlane / v_writelane and only far ones go to HBM
// SIDES: 1 = the source-side sweep, 2 = the sink-side sweep, 3 = both (the two are independent: a workgroup with a second wavefront runs them side by side)
template <int SIDES>
__device__ __attribute__((noinline)) void cert_node_arrays(const Ctx *gp_, const int bi_, const int ei_) {
    Ctx g = *usgpr(gp_); ctx_to_sgpr(g);
    const int bi = usgpr(bi_), ei = usgpr(ei_);
    const int lane = threadIdx.x & 63;
    const size_t cap = (size_t)g.node_cap;
    int *const dmin = g.cert, *const dmax = g.cert + cap, *const bp = g.cert + 2 * cap, *const rmin = g.cert + 3 * cap, *const rmax = g.cert + 4 * cap, *const bs = g.cert + 5 * cap;
    if constexpr ((SIDES & 1) != 0) {
    // ---- source side ----
    int prev_mn = CERT_INF, prev_mx = -1, prev_b = LCD_NEG;
    for (int base = bi; base < ei; base += 64) {
        const int ri = base + lane;
        int w_np = 0, w_p0 = 0, w_pi0 = 0, w_b0 = 0, w_pi1 = 0, w_b1 = 0;
        if (ri < ei) {
            const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
            w_p0 = s0; w_np = s1 - s0;
            if (w_np > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); w_b0 = glb_ld(g.pl_bonus + s0); }
            if (w_np > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); w_b1 = glb_ld(g.pl_bonus + s0 + 1); }
        }
        LCD_PIN(w_np); LCD_PIN(w_p0); LCD_PIN(w_pi0); LCD_PIN(w_b0); LCD_PIN(w_pi1); LCD_PIN(w_b1);
        int cur_mn = CERT_INF, cur_mx = -1, cur_b = LCD_NEG;
        const int nrow = imin(64, ei - base);
        // Rows that hang on the row before them (the backbone between two bubbles: ~97 % of the rows) are taken a RUN at a time -- path lengths count up, the
        // bonus is a prefix sum -- and only the others (several predecessors, or one that is not the row before) one by one
        const bool chain_row = ri >= ei || (ri != bi && w_np == 1 && w_pi0 == ri - 1);
        const unsigned long long odd = __ballot(!chain_row) & (nrow == 64 ? ~0ull : (1ull << nrow) - 1);   // rows to take one by one
        const int cbw = scan_add(ri < ei && ri != bi ? w_b0 : 0);
        for (int k = 0; k < nrow;) {
            if (!((odd >> k) & 1)) { // a run of backbone rows [k, k1)
                const unsigned long long rest = odd >> k;
                const int k1 = rest ? k + __builtin_ctzll(rest) : nrow;
                const int s_mn = k ? LCD_RL(cur_mn, k - 1) : LCD_RL(prev_mn, 63), s_mx = k ? LCD_RL(cur_mx, k - 1) : LCD_RL(prev_mx, 63), s_b = k ? LCD_RL(cur_b, k - 1) : LCD_RL(prev_b, 63);
                const int c0 = k ? LCD_RL(cbw, k - 1) : 0;
                if (s_mx >= 0 && lane >= k && lane < k1) { cur_mn = s_mn + (lane - k + 1); cur_mx = s_mx + (lane - k + 1); cur_b = s_b + (cbw - c0); }
                k = k1;
                continue;
            }
            int mn = CERT_INF, mx = -1, bb = LCD_NEG;
            if (base + k == bi) { mn = 0; mx = 0; bb = 0; }
            else {
                const int np = LCD_RL(w_np, k);
                int p0 = 0;
                if (np > 2) p0 = LCD_RL(w_p0, k);
                for (int t = 0; t < np; ++t) {
                    int pi = t == 0 ? LCD_RL(w_pi0, k) : LCD_RL(w_pi1, k), bz = t == 0 ? LCD_RL(w_b0, k) : LCD_RL(w_b1, k);
                    if (t > 1) { pi = usgpr(glb_ld(g.pl_pidx + p0 + t)); bz = usgpr(glb_ld(g.pl_bonus + p0 + t)); }
                    int pm, px, pb;
                    if (pi >= base) { pm = LCD_RL(cur_mn, pi - base); px = LCD_RL(cur_mx, pi - base); pb = LCD_RL(cur_b, pi - base); }
                    else if (pi >= base - 64) { pm = LCD_RL(prev_mn, pi - base + 64); px = LCD_RL(prev_mx, pi - base + 64); pb = LCD_RL(prev_b, pi - base + 64); }
                    else { pm = usgpr(glb_ld(dmin + pi)); px = usgpr(glb_ld(dmax + pi)); pb = usgpr(glb_ld(bp + pi)); }
                    if (px >= 0) { mn = imin(mn, pm + 1); mx = imax(mx, px + 1); bb = imax(bb, pb + bz); }
                }
            }
            cur_mn = wlane(mn, k, cur_mn); cur_mx = wlane(mx, k, cur_mx); cur_b = wlane(bb, k, cur_b);
            ++k;
        }
        if (ri < ei) { glb_st(dmin + ri, cur_mn); glb_st(dmax + ri, cur_mx); glb_st(bp + ri, cur_b); }
        prev_mn = cur_mn; prev_mx = cur_mx; prev_b = cur_b;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // a far successor reads these back
    }
    }
    if constexpr ((SIDES & 2) == 0) return;
    // ---- sink side: every row pushes (minR + 1, maxR + 1, Bs + bonus) to its predecessors, rows in descending order; the sink itself counts no node ----
    for (int i = bi + lane; i < ei; i += 64) { glb_st(rmin + i, CERT_INF); glb_st(rmax + i, -1); glb_st(bs + i, LCD_NEG); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int top = bi + ((ei - 1 - bi) >> 6 << 6);
    int cur_mn = CERT_INF, cur_mx = -1, cur_b = LCD_NEG, nxt_mn = CERT_INF, nxt_mx = -1, nxt_b = LCD_NEG; // accumulators of the block's rows / of the block below
    int base = top;
    auto push = [&](const int pi, const int cm, const int cx, const int cb) {
        if (pi >= base) {
            const int l = pi - base;
            cur_mn = wlane(imin(LCD_RL(cur_mn, l), cm), l, cur_mn); cur_mx = wlane(imax(LCD_RL(cur_mx, l), cx), l, cur_mx);
            cur_b = wlane(imax(LCD_RL(cur_b, l), cb), l, cur_b);
        } else if (pi >= base - 64) {
            const int l = pi - base + 64;
            nxt_mn = wlane(imin(LCD_RL(nxt_mn, l), cm), l, nxt_mn); nxt_mx = wlane(imax(LCD_RL(nxt_mx, l), cx), l, nxt_mx);
            nxt_b = wlane(imax(LCD_RL(nxt_b, l), cb), l, nxt_b);
        } else { // far predecessor: read-modify-write in HBM (one wavefront, program order; drained before anything reads it back)
            const int om = usgpr(glb_ld(rmin + pi)), ox = usgpr(glb_ld(rmax + pi)), ob = usgpr(glb_ld(bs + pi));
            if (lane == 0) { glb_st(rmin + pi, imin(om, cm)); glb_st(rmax + pi, imax(ox, cx)); glb_st(bs + pi, imax(ob, cb)); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    { // the sink's predecessors: nothing left to align after them
        const int p0 = usgpr(glb_ld(g.pl_start + ei)), np = usgpr(glb_ld(g.pl_start + ei + 1)) - p0;
        for (int t = 0; t < np; ++t) { const int pi = usgpr(glb_ld(g.pl_pidx + p0 + t)), bz = usgpr(glb_ld(g.pl_bonus + p0 + t)); push(pi, 0, 0, bz); }
    }
    for (; base >= bi; base -= 64) {
        const int ri = base + lane;
        int w_np = 0, w_p0 = 0, w_pi0 = 0, w_b0 = 0, w_pi1 = 0, w_b1 = 0;
        if (ri < ei) {
            const int s0 = glb_ld(g.pl_start + ri), s1 = glb_ld(g.pl_start + ri + 1);
            w_p0 = s0; w_np = s1 - s0;
            if (w_np > 0) { w_pi0 = glb_ld(g.pl_pidx + s0); w_b0 = glb_ld(g.pl_bonus + s0); }
            if (w_np > 1) { w_pi1 = glb_ld(g.pl_pidx + s0 + 1); w_b1 = glb_ld(g.pl_bonus + s0 + 1); }
            // far pushes that went through HBM
            cur_mn = imin(cur_mn, glb_ld(rmin + ri)); cur_mx = imax(cur_mx, glb_ld(rmax + ri)); cur_b = imax(cur_b, glb_ld(bs + ri));
        }
        LCD_PIN(w_np); LCD_PIN(w_p0); LCD_PIN(w_pi0); LCD_PIN(w_b0); LCD_PIN(w_pi1); LCD_PIN(w_b1); LCD_PIN(cur_mn); LCD_PIN(cur_mx); LCD_PIN(cur_b);
        const int nrow = imin(64, ei - base);
        // descending: a run of backbone rows [a, k] first lets what was pushed into its rows (from rows above it, in or outside the block) flow down the
        // backbone -- (min, +) / (max, +) suffix scans over the run's lanes (lane order reversed, prefix scan, reversed back) -- and then its first row
        // pushes to the row before it; the other rows push to their predecessors one by one
        const bool chain_row = ri < ei && ri != bi && w_np == 1 && w_pi0 == ri - 1;
        const unsigned long long cm = __ballot(chain_row);
        const int cbl = scan_add(chain_row ? w_b0 : 0);                                    // bonus of the backbone edges up to this row (inside a run)
        const int rl = 63 - lane;
        for (int k = nrow - 1; k >= 0;) {
            if ((cm >> k) & 1) {
                const unsigned long long below = ~cm & ((k == 63 ? ~0ull : (1ull << (k + 1)) - 1)); // rows <= k that are NOT backbone rows
                const int a = below ? 64 - __builtin_clzll(below) : 0;                               // the run is [a, k]
                const bool in = lane >= a && lane <= k;
                const int kmn = in ? cur_mn + lane : (1 << 30), kmx = in && cur_mx >= 0 ? cur_mx + lane : -CERT_INF, kb = in && cur_mx >= 0 ? cur_b + cbl : 2 * LCD_NEG;
                const int smn = __shfl(scan_min(__shfl(kmn, rl)), rl), smx = __shfl(scan_max(__shfl(kmx, rl)), rl), sb = __shfl(scan_max(__shfl(kb, rl)), rl);
                if (in) { if (smx > -CERT_INF / 2) { cur_mn = smn - lane; cur_mx = smx - lane; cur_b = sb - cbl; } else { cur_mn = CERT_INF; cur_mx = -1; cur_b = LCD_NEG; } }
                const int mx0 = LCD_RL(cur_mx, a);
                if (mx0 >= 0) push(base + a - 1, LCD_RL(cur_mn, a) + 1, mx0 + 1, LCD_RL(cur_b, a) + LCD_RL(w_b0, a)); // (a backbone row is never the source: base + a - 1 >= bi)
                k = a - 1;
                continue;
            }
            const int mx = LCD_RL(cur_mx, k);
            if (!(mx < 0 || base + k == bi)) { // (the sink cannot be reached from this row / the source has no predecessor: nothing to push)
                const int mn = LCD_RL(cur_mn, k), bb = LCD_RL(cur_b, k);
                const int np = LCD_RL(w_np, k);
                int p0 = 0;
                if (np > 2) p0 = LCD_RL(w_p0, k);
                for (int t = 0; t < np; ++t) {
                    int pi = t == 0 ? LCD_RL(w_pi0, k) : LCD_RL(w_pi1, k), bz = t == 0 ? LCD_RL(w_b0, k) : LCD_RL(w_b1, k);
                    if (t > 1) { pi = usgpr(glb_ld(g.pl_pidx + p0 + t)); bz = usgpr(glb_ld(g.pl_bonus + p0 + t)); }
                    push(pi, mn + 1, mx + 1, bb + bz);
                }
            }
            --k;
        }
        if (ri < ei) { glb_st(rmin + ri, cur_mn); glb_st(rmax + ri, cur_mx); glb_st(bs + ri, cur_b); }
        cur_mn = nxt_mn; cur_mx = nxt_mx; cur_b = nxt_b; nxt_mn = CERT_INF; nxt_mx = -1; nxt_b = LCD_NEG;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// the bound at the end cell: the best any alignment of the whole read can score (the guess S_est is taken below it)
__device__ int cert_ubtop(const Ctx &g, const int ei, const int qlen, const LcdScoring &sc, int *bztop = nullptr) {
    const size_t cap = (size_t)g.node_cap;
    const int *dmin = g.cert, *dmax = g.cert + cap, *bp = g.cert + 2 * cap;
    const int p0 = g.pl_start[ei], np = g.pl_start[ei + 1] - p0;
    int ub = LCD_NEG, bz = 0;
    for (int t = 0; t < np; ++t) {
        const int pi = g.pl_pidx[p0 + t], mx = dmax[pi], mn = dmin[pi];
        if (mx < 0) continue;
        bz = imax(bz, bp[pi] + g.pl_bonus[p0 + t]);
        ub = imax(ub, bp[pi] + g.pl_bonus[p0 + t] + sc.match * imin(qlen, mx) - cert_G(qlen - mx, sc.o1, sc.e1, sc.o2, sc.e2) - cert_G(mn - qlen, sc.o1, sc.e1, sc.o2, sc.e2));
    }
    if (bztop) *bztop = bz; // the largest sum of edge bonuses over a source..sink path: no cell's value is above qlen * match + this
    return ub;
}
// rows' intervals for the score bound `sest`, lane = row.  On every segment between two consecutive breakpoints {minD, maxD, qlen - maxR, qlen - minR} the
// bound is a linear function minus concave functions of linear functions, i.e. CONVEX: its super-level set on a segment is a prefix and / or a suffix of
// it, so the hull's ends are found by one bisection each, on the first / last segment that has an end point at or above the bound.
// Returns the widest window a row needs (columns from lo rounded down to the lane's 4-cell group to hi, + 2), or -1 when the source row has no interval
// starting at column 0 (the guess was above the optimum).
// (wv of nw wavefronts: every nw-th block of 64 rows; the caller combines the wavefronts' results -- any -1, else the maximum)
__device__ __attribute__((noinline)) int cert_hull(const Ctx *gp_, const int bi_, const int ei_, const int qlen_, const int sest_, const LcdScoring sc_, const int wv_ = 0, const int nw_ = 1) {
    Ctx g = *usgpr(gp_); ctx_to_sgpr(g);
    const int bi = usgpr(bi_), ei = usgpr(ei_), qlen = usgpr(qlen_), sest = usgpr(sest_);
    const int M = usgpr(sc_.match), o1 = usgpr(sc_.o1), e1 = usgpr(sc_.e1), o2 = usgpr(sc_.o2), e2 = usgpr(sc_.e2);
    const int lane = threadIdx.x & 63;
    const size_t cap = (size_t)g.node_cap;
    const int *dmin = g.cert, *dmax = g.cert + cap, *bp = g.cert + 2 * cap, *rmin = g.cert + 3 * cap, *rmax = g.cert + 4 * cap, *bs = g.cert + 5 * cap;
    int *hull = g.cert + 6 * cap;
    const int O = imax(o1, o2);
    int maxw = 0, src_ok = 1;
    const int wv = usgpr(wv_), nw = usgpr(nw_);
    for (int base = bi + 64 * wv; base < ei; base += 64 * nw) {
        const int ri = base + lane;
        int lo = 1, hi = 0;
        if (ri < ei) {
            const int dn = glb_ld(dmin + ri), dx = glb_ld(dmax + ri), rn = glb_ld(rmin + ri), rx = glb_ld(rmax + ri);
            if (dx >= 0 && rx >= 0) {
                const int K0 = glb_ld(bp + ri) + glb_ld(bs + ri) + O - sest; // UB(j) - sest >= 0 <=> keep
                auto ub = [&](const int j) {
                    const int r = qlen - j;
                    return K0 + M * (imin(j, dx) + imin(r, rx)) - cert_G(j - dx, o1, e1, o2, e2) - cert_G(dn - j, o1, e1, o2, e2) - cert_G(r - rx, o1, e1, o2, e2) - cert_G(rn - r, o1, e1, o2, e2);
                };
                auto clampq = [&](const int v) { return imax(0, imin(qlen, v)); };
                int a = clampq(dn), b = clampq(dx), c = clampq(qlen - rx), d = clampq(qlen - rn); // a <= b, c <= d: merge the two sorted pairs
                int p1 = imin(a, c), t1 = imax(a, c), t2 = imin(b, d), p4 = imax(b, d);
                int p2 = imin(t1, t2), p3 = imax(t1, t2);
                const int p0 = 0, p5 = qlen;
                const int f0 = ub(p0), f1 = ub(p1), f2 = ub(p2), f3 = ub(p3), f4 = ub(p4), f5 = ub(p5);
                // first / last breakpoint at or above the bound
                int fl = -1, fh = -1, ll = -1, lh = -1; bool any = false;
#define LCD_SEG_LO(fa, pa, fb, pb) if (fl < 0 && (fb) >= 0) { fl = ((fa) >= 0) ? (pa) : (pa); fh = (pb); any = true; if ((fa) >= 0) fh = (pa); }
                // walk the breakpoints left to right: the hull's left end lies in the segment that ends at the first qualifying breakpoint
                if (f0 >= 0) { lo = 0; any = true; }
                else if (f1 >= 0) { fl = p0; fh = p1; any = true; }
                else if (f2 >= 0) { fl = p1; fh = p2; any = true; }
                else if (f3 >= 0) { fl = p2; fh = p3; any = true; }
                else if (f4 >= 0) { fl = p3; fh = p4; any = true; }
                else if (f5 >= 0) { fl = p4; fh = p5; any = true; }
#undef LCD_SEG_LO
                if (any) {
                    // bracket search with an invariant (ub(l) < 0 <= ub(h)): the first steps guess the crossing by interpolation -- the bound is piecewise
                    // linear with a handful of pieces, so two or three of them usually close the bracket -- then plain bisection; exact either way
                    if (f0 < 0) {
                        int l = fl, h = fh, vl = f1 >= 0 ? f0 : f2 >= 0 ? f1 : f3 >= 0 ? f2 : f4 >= 0 ? f3 : f4, vh = f1 >= 0 ? f1 : f2 >= 0 ? f2 : f3 >= 0 ? f3 : f4 >= 0 ? f4 : f5;
                        for (int it = 0; h - l > 1; ++it) {
                            int m = (l + h) >> 1;
                            if (it < 5) { m = l + (int)((float)(h - l) * (float)(-vl) / (float)(vh - vl)); m = imax(l + 1, imin(h - 1, m)); }
                            const int v = ub(m);
                            if (v >= 0) { h = m; vh = v; } else { l = m; vl = v; }
                        }
                        lo = h;
                    }
                    if (f5 >= 0) hi = qlen;
                    else {
                        if (f4 >= 0) { ll = p4; lh = p5; } else if (f3 >= 0) { ll = p3; lh = p4; } else if (f2 >= 0) { ll = p2; lh = p3; } else if (f1 >= 0) { ll = p1; lh = p2; } else { ll = p0; lh = p1; }
                        int l = ll, h = lh, vl = f4 >= 0 ? f4 : f3 >= 0 ? f3 : f2 >= 0 ? f2 : f1 >= 0 ? f1 : f0, vh = f4 >= 0 ? f5 : f3 >= 0 ? f4 : f2 >= 0 ? f3 : f1 >= 0 ? f2 : f1; // ub(l) >= 0 > ub(h)
                        for (int it = 0; h - l > 1; ++it) {
                            int m = (l + h) >> 1;
                            if (it < 5) { m = l + (int)((float)(h - l) * (float)vl / (float)(vl - vh)); m = imax(l + 1, imin(h - 1, m)); }
                            const int v = ub(m);
                            if (v >= 0) { l = m; vl = v; } else { h = m; vh = v; }
                        }
                        hi = l;
                    }
                }
            }
            glb_st(hull + ri, lo <= hi ? (lo | (hi << 16)) : 1);
            if (ri == bi && !(lo == 0 && hi >= 0)) src_ok = 0;
        }
        const int wd = lo <= hi ? hi - (lo & ~3) + 2 : 0;
        maxw = imax(maxw, wd);
    }
    maxw = lane63(scan_max(maxw));
    const int bad = __builtin_amdgcn_readfirstlane((int)(__ballot(!src_ok) != 0));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return bad ? -1 : maxw;
}

// The certified-band alignment of one read (see the comment block above cert_node_arrays).  Not inlined: inside align_to_subgraph -- i.e. inside the chain
// kernel's body -- its loops and lambdas cost every chain of every class another 50 - 110 B of scratch (the kernel body is what spills).
// Returns the number of cigar entries (status LCD_OK), or 0 with gp->status = LCD_ERR_CERT / an error.
template <int NT>
__device__ __attribute__((noinline)) int align_certified(Ctx *gp, const unsigned ro, const unsigned so, const unsigned pdo_, const LcdScoring sc, const int w, const int bi, const int ei,
                                                         const int rem_beg, const uint8_t *seq_hbm, const int qlen, unsigned long long *cells_acc) {
    Smem &sm = g_smem;
    Ctx &g = *gp; // (the caller's context itself: the few fields this function changes are changed in place)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned pdo = pdo_; // (a window wider than the pool was laid out for overwrites the first-predecessor distances: no attempt after it may use them -- WinOut.clobber)
    WinOut wo; wo.status = g.status; wo.t_dp = wo.t_bt = wo.cells = 0; wo.cig_pos = 0; wo.t_plan = wo.t_poll = 0; wo.score = LCD_NEG; wo.t_setup = 0; wo.clobber = 0;
    int nc = -1;
    auto leave = [&](const int r) { return r; };
            // 64-thread class: the workgroup IS that wavefront.  256-thread class (long chains: the critical path of a submission): wavefront 0 runs the same
            // rows while the others wait; the per-read phases around them (graph update, re-sort, plan), a third to a half of such a chain in a 64-thread
            // workgroup, run on all four wavefronts
            constexpr bool SOLO = NT > 64;
            constexpr int WINC = 504; // the widest window the rows below hold (8 cells per lane, intervals start on a multiple of 8: hull widths are counted from a multiple of 4)
            if (qlen >= 65536 || ei - bi < 2) { g.status = LCD_ERR_CERT; return leave(0); }
            const long long tca0 = clock64();
            if constexpr (NT >= 128) { if (wave == 0) cert_node_arrays<1>(&g, bi, ei); else if (wave == 1) cert_node_arrays<2>(&g, bi, ei); } // (source side / sink side; the others only meet the barrier)
            else cert_node_arrays<3>(&g, bi, ei);
            __syncthreads();
            g.t_plan += (unsigned long long)(clock64() - tca0); // (profiling: the bound's node arrays)
            LCD_BS(13, 90ull * (unsigned long long)(ei - bi + 1)); // (six node arrays written 24 B and the plan read for them 20 B per row, twice (both sweeps); the intervals 4 B written, 4 read by the rows)
            const unsigned long long cells_before = *cells_acc;
            int bztop = 0;
            const int ubtop = cert_ubtop(g, ei, qlen, sc, &bztop);
            g.cert_bztop = bztop;
            // the guess: the bound at the end cell minus a slack -- the largest one an earlier read of this chain needed (+ 25 % + 32), or a small one for
            // the first alignment.  An attempt that comes back below its guess doubles the slack; the best score seen so far is a TRUE lower bound of the
            // optimum and takes over as soon as it is the tighter of the two (that attempt cannot fail)
            // the slack over the largest one the chain has needed so far: + 1/8 + 8.  It was + 1/4 + 32 until the end of round 3: every unit of slack widens every row's
            // interval, and a wider widest interval means more cells per lane for the whole read; an attempt that falls short costs one more pass over the (narrower)
            // intervals and leaves a true lower bound.  Measured at the driver's flags / 1 / 4 / 16 batches in flight: 100.5 k regions/s, 119 / 158 / 217 ms against
            // 100.4 k, 119 / 175 / 223 ms; + 1/8 + 1 .. 4: 122 / 155 / 217; + 1/8 + 32: 120 / 174 / 223.  (LCD_DBG bits 8-11 / 12-19 override the eighths / the constant)
            const int d_a = (sc.dbg >> 8) & 15, d_b = (sc.dbg >> 12) & 255;
            int delta = g.cert_hist < 0 ? 48 + qlen / 32 : g.cert_hist + g.cert_hist * (d_a ? d_a : 1) / 8 + (d_b ? d_b : 8);
            int sbest = LCD_NEG;
            bool done = false;
            auto hull_of = [&](const int sest) { // every row's interval for this score bound (table in g.cert); the widest window a row needs, or -1
                const long long th0 = clock64();
                { const int mv = cert_hull(&g, bi, ei, qlen, sest, sc, wave, NT / 64); if (lane == 0) sm.scan[wave] = mv; } // (the rows are independent: every wavefront takes its share)
                __syncthreads();
                int m = 0;
#pragma unroll
                for (int k = 0; k < NT / 64; ++k) { const int v = sm.scan[k]; m = (m < 0 || v < 0) ? -1 : imax(m, v); }
                __syncthreads();
                g.t_poll += (unsigned long long)(clock64() - th0); // (profiling: the rows' intervals)
                return m;
            };
            // A read whose intervals do not fit the window goes through the GENERIC rows (any width, values in HBM, 2 - 3x slower per row) over the intervals of
            // a looser guess -- a dozen reads of a chain at most, and the chain stays inside its round instead of coming back with full rows in the next one
            auto to_generic = [&](const int slack) {
                if (sc.dbg & 16) { g.status = LCD_ERR_CERT; return 0; } // (test switch: give the chain up instead, so that the host's re-run with full rows is exercised)
                const int sest = imax(sbest, ubtop - slack);
                const int m = hull_of(sest);
                if (m < 0) { g.status = LCD_ERR_CERT; g.t_plan = 6000000ull; return 0; }
                g.cert_generic = 1; g.cert_generic_seen = 1; g.cert_sest = sest; g.cert_ubtop = ubtop; g.cert_cells0 = cells_before;
                return -2;
            };
            for (int attempt = 0; attempt < 10 && !done; ++attempt, delta *= 2) {
                int sest = imax(sbest, ubtop - delta);
                int mw = hull_of(sest);
                bool fitted = false;
                if (mw > WINC && g.cert_generic_seen) return leave(to_generic(ubtop - sest)); // (an earlier read of the chain already had to: no windowed attempt at a tighter guess first -- tried: 39.6 k instead of 56.2 k regions/s at 20 batches)
                if (mw > WINC) {
                    // the intervals of this guess do not fit the window: take the LARGEST slack whose intervals do (they grow with the slack; bisection,
                    // ~16 instructions per row and step) -- if the alignment over those verifies, nothing wider was needed
                    int fit = 0, wide = ubtop - sest;
                    while (wide - fit > 4) { const int mid = (fit + wide) >> 1; const int m = hull_of(ubtop - mid); if (m > WINC) wide = mid; else fit = mid; }
                    sest = ubtop - fit; mw = hull_of(sest); fitted = true;
                    if (mw > WINC) return leave(to_generic((ubtop - sest) + (ubtop - sest) / 2 + 48));
                }
                if (mw < 0) { // not even the source row qualifies: the guess is above the optimum
                    if (fitted) { g.status = LCD_ERR_CERT; g.t_plan = 4000000ull; return leave(0); }
                    continue;
                }
                wo.status = g.status; wo.score = LCD_NEG;
                if (sc.dbg & 32) g.t_setup += 1ull << (12 * (mw <= 60 ? 0 : mw <= 124 ? 1 : mw <= 188 ? 2 : mw <= 380 ? 3 : 4)); // (LCD_DBG=32: reads per widest-interval class, 12 bits each, in the t_setup slot)
                // one, two or four cells per lane by the widest interval (hull widths are computed for 4-cell groups: the narrower variants keep a margin)
                nc = -1;
                bool by_all = false;
                if constexpr (NT == 256) if (SOLO && g.solo == 2 && mw <= 256) { // the rows on all four wavefronts (align_lean_mw); -1: not there, wavefront 0 takes the read as before
                    nc = align_lean_mw(&g, ro, so, pdo, sc, bi, ei, seq_hbm, qlen, &wo);
                    __syncthreads();
                    by_all = nc >= 0; // (every wavefront has the result)
                    if (!by_all) { wo.status = g.status; wo.score = LCD_NEG; }
                }
                if constexpr (NT == 256) if (SOLO && g.solo == 3 && mw > 60 && mw <= 380) { // the rows as a pipeline over the four wavefronts (align_cyc): one or two cells per lane by the widest interval
                    nc = mw <= 188 ? align_cyc<1>(&g, ro, so, pdo, sc, bi, ei, seq_hbm, qlen, &wo) : -1;
                    if (nc < 0) { __syncthreads(); nc = align_cyc<2>(&g, ro, so, pdo, sc, bi, ei, seq_hbm, qlen, &wo); }
                    __syncthreads();
                    by_all = nc >= 0;
                    if (by_all && !(sc.dbg & 32)) g.t_setup += wo.t_poll; // (profiling: wavefront 0's mailbox polls, in the "row setup" slot)
                    if (!by_all) { wo.status = g.status; wo.score = LCD_NEG; }
                }
                if (!by_all && (!SOLO || wave == 0)) {
                    if (mw <= 60) nc = align_lean<2, 1>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo);
                    if (wo.clobber) pdo = 0xffffffffu;
                    if (nc < 0 && mw <= 124) { win_sync<SOLO>(); nc = align_lean<2, 2>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
                    if (wo.clobber) pdo = 0xffffffffu;
                    if (nc < 0 && mw <= 256) { win_sync<SOLO>(); nc = align_lean<2, 4>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
                    if (wo.clobber) pdo = 0xffffffffu;
                    if (nc < 0) { win_sync<SOLO>(); nc = align_lean<2, 8>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); } // (512 columns: a region whose reads differ by an SV-size indel)
                    if (wo.clobber) pdo = 0xffffffffu; // (and the attempts after this one)
                    if (SOLO && lane == 0) { g_wide.ppi[0] = nc; g_wide.ppi[1] = wo.status; g_wide.ppi[2] = wo.score; g_wide.ppi[3] = wo.cig_pos; g_wide.po[0] = (unsigned)wo.cells; g_wide.po[1] = (unsigned)(wo.cells >> 32);
                                             g_wide.po[2] = (unsigned)wo.t_dp; g_wide.po[3] = (unsigned)(wo.t_dp >> 32); g_wide.po[4] = (unsigned)wo.t_bt; g_wide.po[5] = (unsigned)(wo.t_bt >> 32); g_wide.ppi[4] = wo.clobber; }
                }
                if (SOLO && !by_all) { // the result of wavefront 0 to everybody
                    __syncthreads();
                    if (g_wide.ppi[4]) { pdo = 0xffffffffu; wo.clobber = 1; }
                    nc = g_wide.ppi[0]; wo.status = g_wide.ppi[1]; wo.score = g_wide.ppi[2]; wo.cig_pos = g_wide.ppi[3]; wo.cells = g_wide.po[0] | ((unsigned long long)g_wide.po[1] << 32);
                    wo.t_dp = g_wide.po[2] | ((unsigned long long)g_wide.po[3] << 32); wo.t_bt = g_wide.po[4] | ((unsigned long long)g_wide.po[5] << 32); wo.t_setup = 0;
                    __syncthreads();
                }
                if (nc < 0) return leave(to_generic((ubtop - sest) + (ubtop - sest) / 2 + 48)); // (the window's alias checks: rare)
                if (wo.status != LCD_OK) { g.status = wo.status; return leave(0); }
                g.t_dp += wo.t_dp; g.t_bt += wo.t_bt; *cells_acc += wo.cells; if (!(sc.dbg & 32)) g.t_setup += wo.t_setup; wo.t_setup = 0;
                const int S = wo.score;
                if (S > LCD_NEG / 2) {
                    sbest = imax(sbest, S);
                    if (S >= sest) { done = true; g.cert_hist = imax(g.cert_hist, ubtop - S); }
                }
                __syncthreads();
                if (!done && fitted) return leave(to_generic((ubtop - sest) + (ubtop - sest) / 2 + 48)); // the optimum is below every bound whose intervals fit: the window is too narrow for this read
            }
            if (!done) { g.status = LCD_ERR_CERT; g.t_plan = 3000000ull; return leave(0); }
            g.alg_adjust += (long long)(ei - bi) * (qlen + 1) - (long long)(*cells_acc - cells_before); // what align_unbanded would have counted for this read
            g.status = wo.status;
            g.cig_node = g.cig_node0 + wo.cig_pos; g.cig_qpos = g.cig_qpos0 + wo.cig_pos;
            return leave(nc);
}

// The certified band of a NOISY read's K2 alignment is wider than any single wavefront's window (a third to a half of the read), so those chains stay in the class
// their reads' length asks for and run the systolic rows over the intervals (align_unbanded<NT, true>): same bound, same guess-and-verify loop as align_certified,
// no window to fit.  Returns the number of cigar entries, 0 with g.status set, or -3: not here (the caller's full rows take the read).
template <int NT>
__device__ __attribute__((noinline)) int align_certified_sys(Ctx *gp, const unsigned ro, const unsigned pdo, const LcdScoring sc, const int bi, const int ei,
                                                             const uint8_t *seq_hbm, const int qlen, unsigned long long *cells_acc) {
    Smem &sm = g_smem;
    Ctx &g = *gp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (qlen >= 65535 || ei - bi < 2 || qlen + 2 > 4 * NT) return -3;
    const long long tca0 = clock64();
    if constexpr (NT >= 128) { if (wave == 0) cert_node_arrays<1>(&g, bi, ei); else if (wave == 1) cert_node_arrays<2>(&g, bi, ei); }
    else cert_node_arrays<3>(&g, bi, ei);
    __syncthreads();
    g.t_plan += (unsigned long long)(clock64() - tca0);
    const unsigned long long cells_before = *cells_acc;
    const int ubtop = cert_ubtop(g, ei, qlen, sc);
    int delta = g.cert_hist < 0 ? 48 + qlen / 8 : g.cert_hist + g.cert_hist / 4 + 32; // (first read of a chain of noisy reads: every eighth base an error's worth of slack)
    int sbest = LCD_NEG, nc = 0;
    bool done = false;
    WinOut wo; wo.status = g.status; wo.t_dp = wo.t_bt = wo.cells = 0; wo.cig_pos = 0; wo.t_plan = wo.t_poll = 0; wo.score = LCD_NEG; wo.t_setup = 0; wo.clobber = 0;
    for (int attempt = 0; attempt < 12 && !done; ++attempt, delta *= 2) {
        const int sest = imax(sbest, ubtop - delta);
        const long long th0 = clock64();
        if (wave == 0) { const int m = cert_hull(&g, bi, ei, qlen, sest, sc); if (lane == 0) sm.bc[6] = m; }
        __syncthreads();
        const int mw = sm.bc[6];
        __syncthreads();
        g.t_poll += (unsigned long long)(clock64() - th0);
        if (mw < 0) continue; // not even the source row qualifies: the guess is above the optimum
        wo.status = g.status; wo.score = LCD_NEG; wo.t_dp = wo.t_bt = wo.cells = 0;
        nc = align_unbanded<NT, true>(&g, ro, pdo, sc, bi, ei, seq_hbm, qlen, &wo);
        if (nc < 0) return -3;
        if (wo.status != LCD_OK) { g.status = wo.status; return 0; }
        g.t_dp += wo.t_dp; g.t_bt += wo.t_bt; *cells_acc += wo.cells;
        const int S = wo.score;
        if (S > LCD_NEG / 2) {
            sbest = imax(sbest, S);
            if (S >= sest) { done = true; g.cert_hist = imax(g.cert_hist, ubtop - S); }
        }
        __syncthreads();
    }
    if (!done) return -3;
    g.alg_adjust += (long long)(ei - bi) * (qlen + 1) - (long long)(*cells_acc - cells_before); // what the full rows would have counted for this read
    g.status = wo.status;
    g.cig_node = g.cig_node0 + wo.cig_pos; g.cig_qpos = g.cig_qpos0 + wo.cig_pos;
    return nc;
}

// the generic rows' end node + value backtrack (one thread, serial): a function of its own -- inlined, its loops were part of the chain kernel's body, which is what
// spills (the same build with a call inside this block had 70 instead of 400 spilled VGPRs in the 64-thread kernel and ran 5 % faster).  Returns the
// number of cigar entries; *best_out = the end cell's score.
__device__ __attribute__((noinline)) int generic_backtrack(Ctx *gp, const uint8_t *seq, const LcdScoring sc, const int bi, const int ei, const int qlen, const int fixedg_, int *best_out) {
    Ctx &g = *gp;
    const bool fixedg = fixedg_ != 0;
    const int o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2, oe1 = o1 + e1, oe2 = o2 + e2;
        int n_cig = 0;
        int best = LCD_NEG, br = -1;
        {
            const int p0 = g.pl_start[ei], np = g.pl_start[ei + 1] - p0;
            for (int t = 0; t < np; ++t) {
                const int pi = g.pl_pidx[p0 + t];
                if (qlen < g.rbeg[pi] || qlen > g.rend[pi]) continue;
                int c = g.H[g.roff[pi] + (qlen - g.rbeg[pi])] + g.pl_bonus[p0 + t];
                if (c > best) { best = c; br = pi; }
            }
        }
        if (fixedg && (br < 0 || best < g.cert_sest)) { g.status = LCD_ERR_CERT; br = -1; } // the guess was above the optimum: the host re-runs the chain with full rows
        *best_out = best;
        if (br >= 0 && best > LCD_NEG / 2) {
            int pos = qlen;
            int i = br, j = qlen, st = 0;
#define CELLH(pi, jj) g.H[g.roff[pi] + ((jj) - g.rbeg[pi])]
#define INB(pi, jj) ((jj) >= g.rbeg[pi] && (jj) <= g.rend[pi])
            while (i != bi && j > 0 && g.status == LCD_OK) {
                const int v = g.idx2node[i];
                const int p0 = g.pl_start[i], np = g.pl_start[i + 1] - p0;
                if (st == 0) {
                    const int hv = CELLH(i, j);
                    bool hit = false;
                    const uint8_t vb = g.base[v], qb = seq[j - 1];
                    const int s = (vb >= 4 || qb >= 4) ? 0 : (vb == qb ? sc.match : -sc.mismatch);
                    for (int t = 0; t < np && !hit; ++t) {
                        const int pi = g.pl_pidx[p0 + t];
                        if (!INB(pi, j - 1)) continue;
                        if (CELLH(pi, j - 1) + s + g.pl_bonus[p0 + t] == hv) {
                            --pos; g.cig_node[pos] = v; g.cig_qpos[pos] = j - 1; i = pi; --j; hit = true;
                        }
                    }
                    for (int c = 1; c <= 2 && !hit; ++c) {
                        const int *E = c == 1 ? g.E1 : g.E2;
                        for (int t = 0; t < np && !hit; ++t) {
                            const int pi = g.pl_pidx[p0 + t];
                            if (!INB(pi, j)) continue;
                            if (E[g.roff[pi] + (j - g.rbeg[pi])] + g.pl_bonus[p0 + t] == hv) { i = pi; st = c; hit = true; }
                        }
                    }
                    if (!hit) {
                        const int rb = g.rbeg[i];
                        for (int k = j - 1; k >= rb && !hit; --k) {
                            int len = j - k, hk = CELLH(i, k);
                            if (hk - o1 - len * e1 == hv || hk - o2 - len * e2 == hv) {
                                for (int t = j; t > k; --t) { --pos; g.cig_node[pos] = -1; g.cig_qpos[pos] = t - 1; }
                                j = k; hit = true;
                            }
                        }
                    }
                    if (!hit) g.status = LCD_ERR_BACKTRACK;
                } else {
                    const int oe = st == 1 ? oe1 : oe2, ee = st == 1 ? e1 : e2;
                    const int *E = st == 1 ? g.E1 : g.E2;
                    const int ev = E[g.roff[i] + (j - g.rbeg[i])];
                    if (CELLH(i, j) - oe == ev) { st = 0; continue; }
                    bool hit = false;
                    for (int t = 0; t < np && !hit; ++t) {
                        const int pi = g.pl_pidx[p0 + t];
                        if (!INB(pi, j)) continue;
                        if (E[g.roff[pi] + (j - g.rbeg[pi])] + g.pl_bonus[p0 + t] - ee == ev) { i = pi; hit = true; }
                    }
                    if (!hit) g.status = LCD_ERR_BACKTRACK;
                }
            }
#undef CELLH
#undef INB
            while (j > 0) { --pos; g.cig_node[pos] = -1; g.cig_qpos[pos] = j - 1; --j; }
            n_cig = qlen - pos;
            if (pos > 0) for (int t = 0; t < n_cig; ++t) { g.cig_node[t] = g.cig_node[t + pos]; g.cig_qpos[t] = g.cig_qpos[t + pos]; }
        }
    return n_cig;
}

// banded convex-gap global alignment of seq[0..qlen) to the sub-graph (beg_node,end_node); returns #cigar
// entries written to g.cig_node/g.cig_qpos in start->end order (block-uniform result).
//
// Rows that fit one sweep of the workgroup take the windowed path above (values stay in LDS, direction codes go to HBM).
// The generic rows below are the fallback for rows wider than the window (reads longer than 4*NT columns, band drift of
// noisy reads): every row is written to HBM as H/E1/E2 (the value backtrack and far predecessors read it there) AND, when
// it fits, into the K-slot LDS ring.  HBM rows are only read once a full barrier has drained the stores issued before it
// (tracked with last_full).
template <int NT>
__device__ __attribute__((noinline)) int align_to_subgraph(Ctx &g, Smem &sm, int *ring, uint8_t *sseq, const LcdScoring &sc, const int wb, int wf_milli, int beg_node, int end_node,
                                 const uint8_t *seq_hbm, int qlen, unsigned long long *cells_acc) {
    constexpr int NW = NT / 64, K = Cfg<NT>::K;
    constexpr int MP = NW == 1 ? 0 : MAXP; // predecessors staged in LDS per row (single-wavefront rows keep them in registers or read the plan)
    constexpr int WMAX = 4 * NT; // ring slot capacity in columns (the widest window of the class)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    g.cig_node = g.cig_node0; g.cig_qpos = g.cig_qpos0;
    if (qlen <= 0) return 0;
    const int bi = g.node2idx[beg_node], ei = g.node2idx[end_node];
    const int o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2, oe1 = o1 + e1, oe2 = o2 + e2;
    // w = wb<0 ? qlen : wb + (int)(wf*qlen);  wf is 0.01 or 0 on this path: (int)(0.01*q) == q/100
    const int w = wb < 0 ? qlen : wb + (int)(((long long)wf_milli * qlen) / 1000);
    const int remain_end = g.remain[end_node];
    const int n = g.n_node;
    const uint8_t *seq = seq_hbm;
    // LDS after the ring: [query cache | first-predecessor distances]
    const int QB = (qlen + 12 + 15) & ~15;
    if (QB > g.seq_cap) { g.status = LCD_ERR_LDS; return 0; } // read slice longer than the LDS query cache (host sizes the class)
    uint8_t *pd = (QB + (ei - bi) + 16 <= g.seq_cap) ? sseq + QB : nullptr;
    { const long long tb0 = clock64();
    if (g.plan_valid && g.plan_bi == bi && g.plan_ei == ei && g.plan_rend == remain_end) {
        // the graph has the nodes, edges, order and `remain` it had when this plan was built (the reads since only added weight, and add_alignment_block patched the
        // bonuses): only the first-predecessor distances are made again -- they live in the LDS pool behind the query cache, whose place depends on the read
        LCD_PT0();
        if (pd) LCD_BS(2, 12ull * (unsigned long long)(ei - bi + 1)); // (pl_start twice, pl_pidx once per row)
        if (pd) { // (two dependent trips per row: four rows per thread in flight)
            for (int b = bi + tid; b <= ei; b += 4 * NT) {
                int p0[4], np[4], pi[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int idx = imin(b + u * NT, ei); p0[u] = g.pl_start[idx]; np[u] = g.pl_start[idx + 1] - p0[u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) pi[u] = g.pl_pidx[np[u] > 0 ? p0[u] : 0];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int idx = b + u * NT; if (idx <= ei) { const int d = np[u] > 0 ? idx - pi[u] : 255; pd[idx - bi] = (uint8_t)(d < 255 ? d : 255); } }
            }
        }
        __syncthreads();
        LCD_PT(19);
    } else {
        build_plan<NT>(g, sm, bi, ei, remain_end, pd, (NT == 64 || g.solo) ? g.plan_k : K, ring, (int)((uint8_t *)sseq - (uint8_t *)ring));
        // plan build, per row: read imap 1, idx2node 4, in_head 4, remain 4, base 1, and for each of the two in-edge slots taken straight-line e_from 4, e_next_in 4,
        // e_w 4, node2idx 4, imap 1 (= 48 B); written pl_start 4, pl_rem 4, pl_base 1, and per plan entry pl_pidx 4, pl_bonus 4, e_slot 4 (~1.05 entries per row);
        // before it imap (1 B per node) and e_slot (4 B per edge) are reset
        LCD_BS(7, 48ull * (unsigned long long)(ei - bi + 1)); LCD_BS(8, 9ull * (unsigned long long)(ei - bi + 1) + 12ull * (unsigned long long)g.pl_start[ei + 1] + (unsigned long long)g.n_node + 4ull * (unsigned long long)g.n_edge);
        g.plan_valid = 1; g.plan_bi = bi; g.plan_ei = ei; g.plan_rend = remain_end;
    }
    g.t_bp += (unsigned long long)(clock64() - tb0); } // (rows to spill: those with a successor further away than the windowed rows' ring -- the smallest ring a window of this chain may run with)
    if (!(sc.dbg & 8)) {
        WinOut wo; wo.status = g.status; wo.t_dp = wo.t_bt = wo.cells = 0; wo.cig_pos = 0; wo.t_plan = wo.t_poll = 0; wo.t_setup = 0; wo.clobber = 0;
        const int rem_beg = g.remain[beg_node] - remain_end;
        const unsigned pdo = pd ? lds_off(pd) : 0xffffffffu, ro = lds_off(ring), so = lds_off(sseq);
        int nc = -1;
        bool to_generic_rows = false;
        if (wb < 0 && g.cert_on == 2) {
            const int r = align_certified_sys<NT>(&g, ro, pdo, sc, bi, ei, seq_hbm, qlen, cells_acc);
            if (r != -3) return r;
            __syncthreads();
        }
        if constexpr (NT <= 256) if (wb < 0 && g.cert_on == 1) {
            const int r = align_certified<NT>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, cells_acc);
            if (r != -2) return r;
            to_generic_rows = true; // (g.cert_generic is set: the rows below take their intervals from the table)
        }
        if (!to_generic_rows) {
        if (wb < 0) {
            nc = align_unbanded<NT>(&g, ro, pdo, sc, bi, ei, seq_hbm, qlen, &wo);
            if (nc < 0) { __syncthreads(); nc = align_windowed<NT, 0, 4>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
        } else {
            // the host's preferred window (PoaChain.wmax) first; a band that outgrows it is re-run in the next wider one
            if constexpr (NT == 64) { // single wavefront: the lean rows (align_lean); a band that outgrows 256 columns takes the generic rows below
                // (a window wider than the pool was laid out for puts its ring over the first-predecessor distances -- eight slots of 128 columns in a 16 KB pool did,
                //  and the four-cells-per-lane rows that ran next, on two slots, followed the overwritten distances in their backtrack: the round-3 "hang", a backtrack
                //  that walked out of a row's band and stepped from a row to itself for ever.  WinOut.clobber: nothing after such a window uses them)
                unsigned pdl = pdo;
                if (g.wmax <= 64) nc = align_lean<1, 1>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo);
                if (wo.clobber) pdl = 0xffffffffu;
                if (nc < 0 && g.wmax <= 128) { __syncthreads(); nc = align_lean<1, 2>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
                if (wo.clobber) pdl = 0xffffffffu;
                if (nc < 0) { __syncthreads(); nc = align_lean<1, 4>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
            } else if (NT == 256 && g.solo) { // a long K1 chain in a 256-thread workgroup: wavefront 0 runs the lean rows, the others wait for its result
                if (wave == 0) {
                    unsigned pdl = pdo;
                    if (g.wmax <= 64) nc = align_lean<1, 1>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo);
                    if (wo.clobber) pdl = 0xffffffffu;
                    if (nc < 0 && g.wmax <= 128) { win_sync<true>(); nc = align_lean<1, 2>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
                    if (wo.clobber) pdl = 0xffffffffu;
                    if (nc < 0 && g.wmax <= 256) { win_sync<true>(); nc = align_lean<1, 4>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
                    if (wo.clobber) pdl = 0xffffffffu;
                    if (nc < 0) { win_sync<true>(); nc = align_lean<1, 8>(&g, ro, so, pdl, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo); }
                    if (lane == 0) { g_wide.ppi[0] = nc; g_wide.ppi[1] = wo.status; g_wide.ppi[2] = wo.score; g_wide.ppi[3] = wo.cig_pos; g_wide.po[0] = (unsigned)wo.cells; g_wide.po[1] = (unsigned)(wo.cells >> 32);
                                     g_wide.po[2] = (unsigned)wo.t_dp; g_wide.po[3] = (unsigned)(wo.t_dp >> 32); g_wide.po[4] = (unsigned)wo.t_bt; g_wide.po[5] = (unsigned)(wo.t_bt >> 32); }
                }
                __syncthreads();
                nc = g_wide.ppi[0]; wo.status = g_wide.ppi[1]; wo.score = g_wide.ppi[2]; wo.cig_pos = g_wide.ppi[3]; wo.cells = g_wide.po[0] | ((unsigned long long)g_wide.po[1] << 32);
                wo.t_dp = g_wide.po[2] | ((unsigned long long)g_wide.po[3] << 32); wo.t_bt = g_wide.po[4] | ((unsigned long long)g_wide.po[5] << 32); wo.t_setup = 0;
                __syncthreads();
            } else nc = align_windowed<NT, 1, 4>(&g, ro, so, pdo, sc, w, bi, ei, rem_beg, seq_hbm, qlen, &wo);
        }
        if (nc >= 0) {
            // rows: 1 B of direction code per cell; 12 B of row metadata per row (rbeg, rend, roff); 18 B of plan per row (pl_start, pl_pidx, pl_bonus, pl_rem 4 B each,
            // pl_base, imap 1 B each); the read once.  Backtrack: per path entry 8 B of row metadata + 1 B of code + 4 B of order (idx2node); path out 8 B, read back 8 B
            LCD_BS(0, wo.cells); LCD_BS(1, 12ull * (unsigned long long)(ei - bi + 1)); LCD_BS(2, 18ull * (unsigned long long)(ei - bi + 1)); LCD_BS(3, qlen);
            LCD_BS(4, 13ull * (unsigned long long)nc); LCD_BS(5, 16ull * (unsigned long long)nc);
            g.status = wo.status; g.t_dp += wo.t_dp; g.t_bt += wo.t_bt; *cells_acc += wo.cells; g.t_plan += wo.t_plan; g.t_poll += wo.t_poll; g.t_setup += wo.t_setup;
            g.cig_node = g.cig_node0 + wo.cig_pos; g.cig_qpos = g.cig_qpos0 + wo.cig_pos;
            return nc;
        }
        }
        __syncthreads();
    }
    const bool fixedg = g.cert_generic != 0;   // certified band, intervals from the table (wider than the windowed rows hold)
    const int *const hullg = g.cert + 6 * (size_t)g.node_cap;
    g.cert_generic = 0;
    // ring slots of the generic rows: RW columns each -- the class's widest window, or what the host laid the pool out for in the single-wavefront class
    // (64 / 128 columns for banded chains, 384 for certified-band chains: their few reads that come here have intervals of 260 - 380 columns, and a row that
    // fits its slot needs neither the HBM round trip nor the store drain of a row that does not)
    const int RW = (NT == 64 || g.solo) ? (g.ring16 ? g.wmax / 2 : g.wmax) : WMAX; // (a 16-bit ring of wmax columns is half as many int32 columns here)
    const bool ring_ok = RW >= 64;
    for (int i = tid; i < qlen; i += NT) sseq[i] = seq_hbm[i];
    __syncthreads();
    unsigned long long used = 0;
    const long long t_dp0 = clock64();
    // ring bookkeeping (block-uniform registers): which row each slot holds; meta of ring rows lives in sm.rm
    int si0 = -1, si1 = -1, si2 = -1, si3 = -1; // row held by ring slot 0..3 (scalars, not an array: keeps them out of scratch)
#define LCD_SLOT_OF(pi) ((pi) == si0 ? 0 : (K > 1 && (pi) == si1) ? 1 : (K > 2 && (pi) == si2) ? 2 : (K > 3 && (pi) == si3) ? 3 : -1)
    int next_slot = 0;
    int last_full = bi; // every row with index < last_full is complete in HBM
    // ---- source row ----
    int last_idx, last_beg, last_end, last_ml, last_mr, last_slot; unsigned last_off;
    {
        int r = g.remain[beg_node] - remain_end;
        int end = qlen - r; if (end < 0) end = 0; end += w; if (end > qlen) end = qlen;
        if (fixedg) end = hullg[bi] >> 16;
        if ((unsigned long long)end + 1 > g.cell_cap / 3) { g.status = LCD_ERR_CELLS; return 0; }
        const bool fits = ring_ok && end + 1 <= RW;
        if (tid == 0) { g.rbeg[bi] = 0; g.rend[bi] = end; g.roff[bi] = 0; g.ml[bi] = 0; g.mr[bi] = 0; }
        for (int j = tid; j <= end; j += NT) {
            int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            int h = j ? imax(f1, f2) : 0;
            g.H[j] = h; g.E1[j] = h - oe1; g.E2[j] = h - oe2;
            if (fits) { ring[j] = h; ring[RW + j] = h - oe1; ring[2 * RW + j] = h - oe2; }
        }
        used = end + 1;
        last_idx = bi; last_beg = 0; last_end = end; last_ml = 0; last_mr = 0; last_off = 0; last_slot = fits ? 0 : -1;
        if (fits) {
            si0 = bi; next_slot = 1 % K;
            if (tid == 0) { sm.rm[0][0] = 0; sm.rm[0][1] = end; sm.rm[0][2] = 0; sm.rm[0][3] = 0; sm.rm[0][4] = 0; }
        }
        __syncthreads();
        last_full = bi + 1;
    }
    // Plan window: every 64 rows each lane loads the plan of one upcoming row (start, #preds, remain, base and the first two
    // predecessor entries); rows then take it by v_readlane.  The row loop therefore issues no HBM load in the common case,
    // so it never waits (vmcnt is in-order) behind the row stores that are still draining.
    int wbase = -(1 << 20);
    int w_p0 = 0, w_np = 0, w_rem = 1 << 30, w_vb = 4, w_pi0 = 0, w_b0 = 0, w_pi1 = 0, w_b1 = 0;
    // ---- rows ----
    for (int idx = bi + 1; idx < ei; ++idx) {
        if (idx - wbase >= 64) {
            if constexpr (NT == 64) if ((unsigned long long)clock64() > g.wd_deadline) { g.status = LCD_ERR_WATCHDOG; return 0; }
            wbase = idx;
            const int ri = idx + lane;
            w_np = 0; w_rem = 1 << 30;
            if (ri < ei) {
                const int s0 = g.pl_start[ri], s1 = g.pl_start[ri + 1];
                w_p0 = s0; w_np = s1 - s0; w_rem = g.pl_rem[ri]; w_vb = g.pl_base[ri];
                if (w_np > 0) { w_pi0 = g.pl_pidx[s0]; w_b0 = g.pl_bonus[s0]; }
                if (w_np > 1) { w_pi1 = g.pl_pidx[s0 + 1]; w_b1 = g.pl_bonus[s0 + 1]; }
            }
        }
        const int wk = idx - wbase;
        const int p0 = LCD_RL(w_p0, wk), np = LCD_RL(w_np, wk);
        const int rem = LCD_RL(w_rem, wk); const uint8_t vb = (uint8_t)LCD_RL(w_vb, wk);
        int my_pi = 0, my_bonus = 0;
        if (np <= 2) {
            const int a0 = LCD_RL(w_pi0, wk), a1 = LCD_RL(w_pi1, wk), c0 = LCD_RL(w_b0, wk), c1 = LCD_RL(w_b1, wk);
            my_pi = lane == 0 ? a0 : a1; my_bonus = lane == 0 ? c0 : c1;
        } else if (tid < np && tid < MAXP) { my_pi = g.pl_pidx[p0 + tid]; my_bonus = g.pl_bonus[p0 + tid]; }
        if (rem == (1 << 30)) { // not reachable
            if (tid == 0) { g.rbeg[idx] = 1; g.rend[idx] = 0; g.roff[idx] = (uint32_t)used; g.ml[idx] = 0; g.mr[idx] = 0; }
            continue;
        }
        // predecessor metadata.  One wavefront (NT == 64): lane t keeps predecessor t in registers and the row loops
        // broadcast it with v_readlane (no LDS round trip).  More wavefronts: staged in LDS for everybody.
        const bool regstage = (NW == 1) && np <= 64;
        int r_pb = 1, r_pe = 0, r_pml = 0, r_pmr = 0, r_slot = -1, r_bonus = my_bonus; unsigned r_po = 0;
        if (regstage) {
            const bool mine = lane < np;
            if (mine) r_slot = LCD_SLOT_OF(my_pi);
            // a predecessor row (or its metadata) must be read from HBM: drain the stores issued since the last full barrier
            if (__any(mine && r_slot < 0 && my_pi >= last_full)) { __syncthreads(); last_full = idx; }
            if (mine) {
                if (my_pi == last_idx) { r_pb = last_beg; r_pe = last_end; r_po = last_off; r_pml = last_ml; r_pmr = last_mr; }
                else if (r_slot >= 0) { r_pb = sm.rm[r_slot][0]; r_pe = sm.rm[r_slot][1]; r_po = (unsigned)sm.rm[r_slot][2]; r_pml = sm.rm[r_slot][3]; r_pmr = sm.rm[r_slot][4]; }
                else { r_pb = g.rbeg[my_pi]; r_pe = g.rend[my_pi]; r_po = g.roff[my_pi]; r_pml = g.ml[my_pi]; r_pmr = g.mr[my_pi]; }
            }
        } else if constexpr (NW == 1) {
            __syncthreads(); last_full = idx; // > 64 predecessors on a single-wavefront row: everything through the plan and HBM
        } else {
            // pass 1: where does each predecessor row live (ring slot or HBM only)
            if (tid < np && tid < MP) {
                const int slot = LCD_SLOT_OF(my_pi);
                g_wide.bonus[tid] = my_bonus; g_wide.ppi[tid] = my_pi; g_wide.pslot[tid] = slot;
            }
            lds_barrier<NT>();
            {
                int far_pi = np > MP ? (1 << 30) : -1;
                const int ns = imin(np, MP);
                for (int t = 0; t < ns; ++t) if (g_wide.pslot[t] < 0) far_pi = imax(far_pi, g_wide.ppi[t]);
                if (far_pi >= last_full) { __syncthreads(); last_full = idx; }
            }
            // pass 2: metadata from registers (row just computed), ring meta (LDS) or HBM
            if (tid < np && tid < MP) {
                const int pi = my_pi, slot = g_wide.pslot[tid];
                if (pi == last_idx) { g_wide.pb[tid] = last_beg; g_wide.pe[tid] = last_end; g_wide.po[tid] = last_off; g_wide.pml[tid] = last_ml; g_wide.pmr[tid] = last_mr; }
                else if (slot >= 0) { g_wide.pb[tid] = sm.rm[slot][0]; g_wide.pe[tid] = sm.rm[slot][1]; g_wide.po[tid] = (unsigned)sm.rm[slot][2]; g_wide.pml[tid] = sm.rm[slot][3]; g_wide.pmr[tid] = sm.rm[slot][4]; }
                else { g_wide.pb[tid] = g.rbeg[pi]; g_wide.pe[tid] = g.rend[pi]; g_wide.po[tid] = g.roff[pi]; g_wide.pml[tid] = g.ml[pi]; g_wide.pmr[tid] = g.mr[pi]; }
            }
            lds_barrier<NT>();
        }
        // band: pulled from the predecessors' row-max columns (same values the oracle pushes to successors)
        int mplv = 1 << 30, mprv = 0, minpb = 1 << 30, maxpe = -1;
        for (int t = 0; t < np; ++t) {
            int pb, pe, pml, pmr;
            if (regstage) { pb = LCD_RL(r_pb, t); pe = LCD_RL(r_pe, t); pml = LCD_RL(r_pml, t); pmr = LCD_RL(r_pmr, t); }
            else if (t < MP) { pb = g_wide.pb[t]; pe = g_wide.pe[t]; pml = g_wide.pml[t]; pmr = g_wide.pmr[t]; }
            else { const int pi = g.pl_pidx[p0 + t]; pb = g.rbeg[pi]; pe = g.rend[pi]; pml = g.ml[pi]; pmr = g.mr[pi]; }
            if (pb > pe) continue;
            minpb = imin(minpb, pb); maxpe = imax(maxpe, pe);
            mplv = imin(mplv, pml + 1); mprv = imax(mprv, pmr + 1);
        }
        int beg = imin(mplv, qlen - rem) - w; if (beg < 0) beg = 0;
        int end = imax(mprv, qlen - rem) + w; if (end > qlen) end = qlen;
        if (beg < minpb) beg = minpb;
        if (end > maxpe + 1) end = maxpe + 1;
        if (fixedg) { const int hw = hullg[idx]; beg = hw & 65535; end = hw >> 16; } // (lo > hi: no cell of the row can be on an optimal path)
        if (beg > end) {
            if (tid == 0) { g.rbeg[idx] = 1; g.rend[idx] = 0; g.roff[idx] = (uint32_t)used; g.ml[idx] = 0; g.mr[idx] = 0; }
            lds_barrier<NT>();
            continue;
        }
        const unsigned long long off = used;
        const int width = end - beg + 1;
        used += (unsigned long long)width;
        if (used > g.cell_cap / 3) { g.status = LCD_ERR_CELLS; return 0; }
        if (tid == 0) { g.rbeg[idx] = beg; g.rend[idx] = end; g.roff[idx] = (uint32_t)off; }
        const int nchunks = (width + 63) >> 6;
        bool fits = ring_ok && width <= RW;
        // a row of more than one sweep (slots wider than NW * RMAX * 64 columns exist in the single-wavefront class only) writes its first sweep into its slot before the
        // second sweep has read the predecessors: it may not take the slot of a row it still reads (a predecessor K rows back)
        if (fits && width > NW * RMAX * 64) { if (!regstage) fits = false; else if (__any(lane < np && r_slot == next_slot)) fits = false; }
        const int slot = fits ? next_slot : -1;
        int *rH = ring + (size_t)(slot < 0 ? 0 : slot) * 3 * RW, *rE1 = rH + RW, *rE2 = rH + 2 * RW;
        int carry1 = LCD_NEG * 2, carry2 = LCD_NEG * 2; // running max over the sweeps already done (rows wider than WMAX)
        int best_h = LCD_NEG - 64, best_l = 1 << 30, best_r = -1;
        int sweep = 0;
        for (int cb = 0; cb < nchunks; cb += NW * RMAX, ++sweep) {
            const int left = nchunks - cb;
            const int R = imin(RMAX, (left + NW - 1) / NW); // chunks per wavefront in this sweep (contiguous per wavefront)
            // per-chunk values live in named scalars (h0.., not arrays: hipcc put the array form into scratch memory)
            int hp0 = LCD_NEG, hp1 = LCD_NEG, hp2 = LCD_NEG, hp3 = LCD_NEG, ea0 = LCD_NEG, ea1 = LCD_NEG, ea2 = LCD_NEG, ea3 = LCD_NEG;
            int eb0 = LCD_NEG, eb1 = LCD_NEG, eb2 = LCD_NEG, eb3 = LCD_NEG;
            int pa0 = LCD_NEG * 2, pa1 = LCD_NEG * 2, pa2 = LCD_NEG * 2, pa3 = LCD_NEG * 2, pb0 = LCD_NEG * 2, pb1 = LCD_NEG * 2, pb2 = LCD_NEG * 2, pb3 = LCD_NEG * 2;
            int wc1 = LCD_NEG * 2, wc2 = LCD_NEG * 2; // in-wavefront carry
            auto phaseA = [&](const int r, int &hp, int &ev1, int &ev2, int &pr1, int &pr2) {
                const int rel = ((cb + wave * R + r) << 6) + lane;
                const int j = beg + rel;
                const bool act = j <= end;
                int mx = LCD_NEG, e1i = LCD_NEG, e2i = LCD_NEG;
                int s = 0;
                if (act && j >= 1) { const uint8_t qb = sseq[j - 1]; s = (vb >= 4 || qb >= 4) ? 0 : (vb == qb ? sc.match : -sc.mismatch); }
                for (int t = 0; t < np; ++t) { // uniform loop: every lane takes part in the readlane broadcasts
                    int pb, pe, bonus, ps; unsigned po;
                    if (regstage) { pb = LCD_RL(r_pb, t); pe = LCD_RL(r_pe, t); po = (unsigned)LCD_RL((int)r_po, t); bonus = LCD_RL(r_bonus, t); ps = LCD_RL(r_slot, t); }
                    else if (t < MP) { pb = g_wide.pb[t]; pe = g_wide.pe[t]; po = g_wide.po[t]; bonus = g_wide.bonus[t]; ps = g_wide.pslot[t]; }
                    else { const int pi = g.pl_pidx[p0 + t]; pb = g.rbeg[pi]; pe = g.rend[pi]; po = g.roff[pi]; bonus = g.pl_bonus[p0 + t]; ps = -1; }
                    if (!act) continue;
                    if (ps >= 0) {
                        const int *qH = ring + (size_t)ps * 3 * RW;
                        if (j >= 1 && j - 1 >= pb && j - 1 <= pe) mx = imax(mx, qH[j - 1 - pb] + s + bonus);
                        if (j >= pb && j <= pe) { e1i = imax(e1i, qH[RW + (j - pb)] + bonus); e2i = imax(e2i, qH[2 * RW + (j - pb)] + bonus); }
                    } else {
                        if (j >= 1 && j - 1 >= pb && j - 1 <= pe) mx = imax(mx, g.H[po + (j - 1 - pb)] + s + bonus);
                        if (j >= pb && j <= pe) { e1i = imax(e1i, g.E1[po + (j - pb)] + bonus); e2i = imax(e2i, g.E2[po + (j - pb)] + bonus); }
                    }
                }
                const int hpre = imax(mx, imax(e1i, e2i));
                hp = hpre; ev1 = e1i; ev2 = e2i;
                // F via prefix max of A[k] = Hpre[k] + (k-beg)*e
                const int a1 = act ? hpre + rel * e1 : LCD_NEG * 2, a2 = act ? hpre + rel * e2 : LCD_NEG * 2;
                const int i1 = scan_max(a1), i2 = scan_max(a2);
                pr1 = imax(shr1(LCD_NEG * 2, i1), wc1); pr2 = imax(shr1(LCD_NEG * 2, i2), wc2);
                wc1 = imax(wc1, lane63(i1)); wc2 = imax(wc2, lane63(i2));
            };
            if (0 < R) phaseA(0, hp0, ea0, eb0, pa0, pb0);
            if (1 < R) phaseA(1, hp1, ea1, eb1, pa1, pb1);
            if (2 < R) phaseA(2, hp2, ea2, eb2, pa2, pb2);
            if (3 < R) phaseA(3, hp3, ea3, eb3, pa3, pb3);
            int cin1 = carry1, cin2 = carry2;
            if (NW > 1) {
                const int buf = sweep & 1;
                if (lane == 0) { sm.tot1[buf][wave] = wc1; sm.tot2[buf][wave] = wc2; }
                lds_barrier<NT>();
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    const int t1 = sm.tot1[buf][k], t2 = sm.tot2[buf][k];
                    if (k < wave) { cin1 = imax(cin1, t1); cin2 = imax(cin2, t2); }
                    carry1 = imax(carry1, t1); carry2 = imax(carry2, t2);
                }
            } else { carry1 = imax(carry1, wc1); carry2 = imax(carry2, wc2); }
            auto phaseB = [&](const int r, const int hp, const int ev1, const int ev2, const int pr1, const int pr2) {
                const int rel = ((cb + wave * R + r) << 6) + lane;
                const int j = beg + rel;
                if (j <= end) {
                    const int p1 = imax(pr1, cin1), p2 = imax(pr2, cin2);
                    const int f1 = (j > beg) ? imax(LCD_NEG, p1 - o1 - rel * e1) : LCD_NEG;
                    const int f2 = (j > beg) ? imax(LCD_NEG, p2 - o2 - rel * e2) : LCD_NEG;
                    int h = imax(hp, imax(f1, f2)); if (h < LCD_NEG) h = LCD_NEG;
                    int eo1 = imax(h - oe1, ev1 - e1), eo2 = imax(h - oe2, ev2 - e2);
                    if (eo1 < LCD_NEG) eo1 = LCD_NEG;
                    if (eo2 < LCD_NEG) eo2 = LCD_NEG;
                    g.H[off + rel] = h; g.E1[off + rel] = eo1; g.E2[off + rel] = eo2;
                    if (fits) { rH[rel] = h; rE1[rel] = eo1; rE2[rel] = eo2; }
                    if (h > best_h) { best_h = h; best_l = j; best_r = j; } else if (h == best_h) best_r = j;
                }
            };
            if (0 < R) phaseB(0, hp0, ea0, eb0, pa0, pb0);
            if (1 < R) phaseB(1, hp1, ea1, eb1, pa1, pb1);
            if (2 < R) phaseB(2, hp2, ea2, eb2, pa2, pb2);
            if (3 < R) phaseB(3, hp3, ea3, eb3, pa3, pb3);
        }
        // row maximum, leftmost / rightmost column
        int ml = 0, mr = 0;
        if (wb >= 0) { // (unbanded rows: w = qlen makes every band [0, qlen], the row-max columns are never consumed)
            const int wmax = lane63(scan_max(best_h));
            const int wl = lane63(scan_min(best_h == wmax ? best_l : (1 << 30)));
            const int wr = lane63(scan_max(best_h == wmax ? best_r : -1));
            if (NW > 1) {
                if (lane == 0) { sm.bh[wave] = wmax; sm.bl[wave] = wl; sm.br[wave] = wr; }
                lds_barrier<NT>();
                int rowmax = sm.bh[0];
#pragma unroll
                for (int k = 1; k < NW; ++k) rowmax = imax(rowmax, sm.bh[k]);
                ml = 1 << 30; mr = -1;
#pragma unroll
                for (int k = 0; k < NW; ++k) if (sm.bh[k] == rowmax) { ml = imin(ml, sm.bl[k]); mr = imax(mr, sm.br[k]); }
            } else { ml = wl; mr = wr; }
        }
        if (tid == 0) {
            g.ml[idx] = ml; g.mr[idx] = mr;
            if (fits) { sm.rm[slot][0] = beg; sm.rm[slot][1] = end; sm.rm[slot][2] = (int)(unsigned)off; sm.rm[slot][3] = ml; sm.rm[slot][4] = mr; }
        }
        if (fits) {
            if (slot == 0) si0 = idx; else if (slot == 1) si1 = idx; else if (slot == 2) si2 = idx; else si3 = idx;
            next_slot = (slot + 1) % K;
        } else {
            // the row lives in HBM only: forget any ring row it might alias and make it readable before it is used
            __syncthreads(); last_full = idx + 1;
        }
        last_idx = idx; last_beg = beg; last_end = end; last_ml = ml; last_mr = mr; last_off = (unsigned)off; last_slot = slot;
    }
    __syncthreads();
    *cells_acc += used;
    LCD_BS(14, 24ull * used); // (generic rows: H, E1, E2 written and read back, 4 B each per cell)
    const long long t_bt0 = clock64();
    g.t_dp += (unsigned long long)(t_bt0 - t_dp0);
    // ---- end node: best predecessor at column qlen, then backtrack (thread 0) ----
    if (tid == 0) {
        int best = LCD_NEG;
        const int n_cig = generic_backtrack(&g, seq, sc, bi, ei, qlen, fixedg ? 1 : 0, &best);
        sm.bc[5] = best;
        sm.bc[0] = n_cig; sm.bc[1] = g.status;
    }
    __syncthreads();
    const int n_cig = sm.bc[0];
    g.status = sm.bc[1];
    if (fixedg && g.status == LCD_OK) { // certified: the slack this read needed, and what full rows would have counted
        g.cert_hist = imax(g.cert_hist, g.cert_ubtop - sm.bc[5]);
        g.alg_adjust += (long long)(ei - bi) * (qlen + 1) - (long long)(*cells_acc - g.cert_cells0);
    }
    __syncthreads();
    g.t_bt += (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

// A chain's DP region (code / ordinal / spilled-row planes, or the H / E1 / E2 planes of the generic rows) is sized by the host from an estimate.  When a read's rows
// do not fit, the workgroup takes a region four times larger (up to the worst case: every node a row of the longest read) from the launch set's spare pool and the
// caller repeats the read: nothing of the chain is lost, where a re-run by the host starts again from the first read -- and a submission's re-runs, a few long
// chains, were a second round as long as the first on the SV shape.  The graph arrays stay where they are; the old region is simply left behind (bump
// allocation; the host resets the pool between rounds).  false: no pool, pool exhausted, or the region already at its worst case.
template <int NT>
__device__ __attribute__((noinline)) bool grow_dp_region(Ctx &g, Smem &sm, const PoaChain &ch, PoaSpare *sp) {
    if (!sp || ch.cert == 1) return false;
    // the most a read can need: every node a row of every column -- or every row SPILLED (12 bytes per window column whatever the row's width: in the graphs of noisy
    // reads nearly every row has a successor beyond the ring, and the wide classes' windows are thousands of columns: K2 chains of 59 x 1 - 2 kb ended here at their
    // 55th read with the cells' bound reached and a third of the rows' worth of spill space)
    unsigned long long worst = (unsigned long long)g.node_cap * (unsigned long long)(ch.max_len + 1);
    {
        const unsigned long long win = (NT == 64 || ch.solo) ? (unsigned long long)ch.wmax : 4ull * NT;
        const unsigned long long spill_all = ((unsigned long long)g.node_cap * 3ull * win * 4ull) / (unsigned long long)(g.spill_x > 2 ? g.spill_x : 2) + 128;
        if (spill_all > worst) worst = spill_all;
    }
    if (g.cell_cap >= worst) return false;
    unsigned long long nc = g.cell_cap * 4ull; if (nc > worst) nc = worst;
    const unsigned long long a = lcd_align_up(nc, 16), bytes = lcd_align_up(a * (unsigned long long)(g.spill_x > 2 ? 1 + 4 + g.spill_x : 4) + 64, 256);
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long old = __hip_atomic_load(&sp->used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), off = ~0ull;
        const unsigned long long cap = sp->cap;
        for (;;) {
            if (old + bytes > cap) break;
            const unsigned long long seen = atomicCAS(&sp->used, old, old + bytes);
            if (seen == old) { off = old; break; }
            old = seen;
        }
        atomicAdd(off != ~0ull ? &sp->n_grown : &sp->n_refused, 1u);
        sm.bc[6] = (int)(unsigned)(off & 0xffffffffu); sm.bc[7] = (int)(unsigned)(off >> 32);
    }
    __syncthreads();
    const unsigned long long off = (unsigned long long)(unsigned)sm.bc[6] | ((unsigned long long)(unsigned)sm.bc[7] << 32);
    __syncthreads();
    if (off == ~0ull) return false;
    uint8_t *p = (uint8_t *)(uintptr_t)sp->base + off;
    g.H = (int *)p; g.E1 = g.H + nc / 3; g.E2 = g.E1 + nc / 3;
    g.code8 = p; g.ord = (int *)(p + a); g.spill = (int *)(p + (g.spill_x > 2 ? 5 : 2) * a);
    g.cell_cap = nc;
    return true;
}

// the output phase of a chain (MSA rank, rows, clusters, consensus: oracle/poa.c poa_output): a function of its own, like the per-read phases (the chain kernel's body
// is what spills)
template <int NT>
__device__ __attribute__((noinline)) void chain_output(Ctx &g, Smem &sm, const PoaChain &ch, uint8_t *outpool, PoaChainOut &out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_seq = ch.n_reads;
    (void)wave;
    uint8_t *ob = outpool + ch.out_off;
    const int nc_cap = ch.node_cap;
    uint8_t *cons0 = ob, *cons1 = ob + nc_cap;
    uint8_t *msa = ob + 2 * (size_t)nc_cap; // rows: n_seq reads, then cons rows n_seq, n_seq+1
    int *clu_ids = (int *)(ob + lcd_align_up((uint64_t)(n_seq + 4) * nc_cap, 16)); // [2][n_seq]
    if (g.status == LCD_OK && g.n_node > 2) {
        const int n = g.n_node;
        int *rank = g.deg;
        // MSA column of every node = the number of aligned rings whose FIRST node (in topological order) comes before this ring's first node (the oracle hands the
        // columns out in one serial pass over the order: oracle/poa.c poa_output).  In parallel: every row finds its ring's smallest index (rings are short), the
        // rows that ARE that smallest index are numbered by a prefix sum, and every node takes its ring leader's number.  (The serial pass on one lane -- three
        // dependent loads per node -- was 6 % of a HiFi-shape step.)
        int *lead_idx = g.pl_start, *colnum = g.queue; // (free after the last read)
        int ncol = 0;
        {
            if (tid == 0) { rank[0] = -1; rank[1] = -1; }
            int carry = 0;
            for (int base = 1; base < n - 1; base += NT) {
                const int idx = base + tid;
                int isl = 0;
                if (idx < n - 1) {
                    const int v = g.idx2node[idx];
                    int m = idx;
                    for (int a = g.aligned[v]; a != v; a = g.aligned[a]) m = imin(m, g.node2idx[a]);
                    lead_idx[idx] = m; isl = m == idx;
                }
                int tot;
                const int r = block_excl_scan<NT>(isl, sm, &tot);
                if (idx < n - 1) colnum[idx] = carry + r;
                carry += tot;
            }
            ncol = carry;
            __syncthreads();
            for (int idx = 1 + tid; idx < n - 1; idx += NT) rank[g.idx2node[idx]] = colnum[lead_idx[idx]];
        }
        __syncthreads();
        for (size_t t = tid; t < (size_t)(n_seq + 2) * ncol; t += NT) msa[(t / ncol) * (size_t)nc_cap + (t % ncol)] = LCD_GAP;
        __syncthreads();
        for (int v = 2 + tid; v < n; v += NT) {
            const int col = rank[v]; const uint8_t b = g.base[v];
            for (int e = g.out_head[v]; e >= 0; e = g.e_next_out[e])
                for (int wd = 0; wd < g.rid_words; ++wd) {
                    unsigned long long bits = g.rid[(size_t)e * g.rid_words + wd];
                    while (bits) { int r = __ffsll((long long)bits) - 1; bits &= bits - 1; msa[(size_t)(wd * 64 + r) * nc_cap + col] = b; }
                }
        }
        for (int r = tid; r < n_seq; r += NT) g.clu[r] = 0;
        __syncthreads();
        // clustering + consensus run on wavefront 0 (ballot compactions); the other wavefronts only meet the barriers
        int n_clu = 1;
        if (ch.mode == 1 && n_seq >= 2) {
            const int min_w = (int)ch.min_w;
            if (wave == 0) {
                int n_het = 0;
                for (int c0 = 0; c0 < ncol; c0 += 64) {
                    int c = c0 + lane, ishet = 0;
                    if (c < ncol) {
                        Cnt6 cnt;
                        for (int r = 0; r < n_seq; ++r) cnt.add(msa[(size_t)r * nc_cap + c]);
                        int k = 0;
                        for (int a = 0; a < 6; ++a) k += cnt.get(a) >= min_w;
                        ishet = k >= 2;
                    }
                    unsigned long long m = __ballot(ishet);
                    if (ishet) g.het[n_het + __popcll(m & ((1ull << lane) - 1))] = c;
                    n_het += __popcll(m);
                }
                if (lane == 0) sm.bc[1] = n_het;
            }
            __syncthreads();
            const int n_het = sm.bc[1];
            if (n_het > 0) {
                if (wave == 0) {
                    int bv2 = -1, bh = 1 << 30, ba0 = 0, ba1 = 0;
                    for (int h = lane; h < n_het; h += 64) {
                        Cnt6 cnt;
                        for (int r = 0; r < n_seq; ++r) cnt.add(msa[(size_t)r * nc_cap + g.het[h]]);
                        int m0 = 0; for (int a = 1; a < 6; ++a) if (cnt.get(a) > cnt.get(m0)) m0 = a;
                        int m1 = -1; for (int a = 0; a < 6; ++a) if (a != m0 && (m1 < 0 || cnt.get(a) > cnt.get(m1))) m1 = a;
                        if (cnt.get(m1) > bv2) { bv2 = cnt.get(m1); bh = h; ba0 = m0; ba1 = m1; }
                    }
                    int gv2 = wave_max(bv2);
                    int gh = wave_min(bv2 == gv2 ? bh : (1 << 30));
                    int src = __ffsll((long long)__ballot(bv2 == gv2 && bh == gh)) - 1;
                    const int a0 = __shfl(ba0, src), a1 = __shfl(ba1, src);
                    const int pcol = g.het[gh];
                    for (int r = lane; r < n_seq; r += 64) { int al = msa[(size_t)r * nc_cap + pcol]; g.clu[r] = al == a0 ? 0 : al == a1 ? 1 : -1; }
                }
                __syncthreads();
                for (int it = 0; it < 10; ++it) {
                    for (int t = tid; t < 2 * n_het; t += NT) {
                        int c = t / n_het, h = t % n_het;
                        Cnt6 cnt;
                        for (int r = 0; r < n_seq; ++r) if (g.clu[r] == c) cnt.add(msa[(size_t)r * nc_cap + g.het[h]]);
                        int m0 = 0; for (int a = 1; a < 6; ++a) if (cnt.get(a) > cnt.get(m0)) m0 = a;
                        g.prof[t] = (uint8_t)m0;
                    }
                    if (tid == 0) sm.bc[2] = 0;
                    __syncthreads();
                    int changed = 0;
                    for (int r = tid; r < n_seq; r += NT) {
                        int d0 = 0, d1 = 0;
                        for (int h = 0; h < n_het; ++h) {
                            uint8_t al = msa[(size_t)r * nc_cap + g.het[h]];
                            d0 += al != g.prof[h]; d1 += al != g.prof[n_het + h];
                        }
                        int cur = g.clu[r];
                        int nc = d0 < d1 ? 0 : d1 < d0 ? 1 : (cur >= 0 ? cur : 0);
                        if (nc != cur) changed = 1;
                        g.nclu[r] = nc;
                    }
                    if (changed) sm.bc[2] = 1;
                    __syncthreads();
                    for (int r = tid; r < n_seq; r += NT) g.clu[r] = g.nclu[r];
                    const int any_changed = sm.bc[2];
                    __syncthreads();
                    if (!any_changed) break;
                }
                if (wave == 0) {
                    int c1 = 0;
                    for (int r = lane; r < n_seq; r += 64) c1 += g.clu[r];
                    for (int d = 32; d >= 1; d >>= 1) c1 += __shfl_xor(c1, d);
                    if (lane == 0) sm.bc[3] = c1;
                }
                __syncthreads();
                const int c1 = sm.bc[3], c0n = n_seq - c1;
                if (c0n >= min_w && c1 >= min_w) {
                    n_clu = 2;
                    if (c1 > c0n) for (int r = tid; r < n_seq; r += NT) g.clu[r] ^= 1;
                } else
                    for (int r = tid; r < n_seq; r += NT) g.clu[r] = 0;
                __syncthreads();
            }
        }
        out.n_cons = n_clu; out.msa_len = ncol;
        for (int c = 0; c < n_clu; ++c) {
            if (wave == 0) {
                int csize = 0;
                for (int r0 = 0; r0 < n_seq; r0 += 64) {
                    int r = r0 + lane, in = r < n_seq && g.clu[r] == c;
                    unsigned long long m = __ballot(in);
                    if (in) clu_ids[c * n_seq + csize + __popcll(m & ((1ull << lane) - 1))] = r;
                    csize += __popcll(m);
                }
                if (lane == 0) sm.bc[4] = csize;
            }
            __syncthreads();
            const int csize = sm.bc[4];
            out.clu_n[c] = csize;
            uint8_t *crow = msa + (size_t)(n_seq + c) * nc_cap;
            uint8_t *cons = c == 0 ? cons0 : cons1;
            if (wave == 0) {
                int cl = 0;
                for (int c0 = 0; c0 < ncol; c0 += 64) {
                    int col = c0 + lane, emit = 0, mb = 0;
                    if (col < ncol) {
                        Cnt6 cnt;
                        for (int k = 0; k < csize; ++k) cnt.add(msa[(size_t)clu_ids[c * n_seq + k] * nc_cap + col]);
                        for (int a = 1; a < 5; ++a) if (cnt.get(a) > cnt.get(mb)) mb = a;
                        emit = cnt.get(mb) > 0 && cnt.get(mb) >= cnt.get(5);
                    }
                    unsigned long long m = __ballot(emit);
                    if (emit) { cons[cl + __popcll(m & ((1ull << lane) - 1))] = (uint8_t)mb; crow[col] = (uint8_t)mb; }
                    cl += __popcll(m);
                }
                if (lane == 0) sm.bc[5] = cl;
            }
            __syncthreads();
            out.cons_len[c] = sm.bc[5];
            __syncthreads();
        }
    }
}

} // namespace

// Register budget per class (second launch-bound argument = wavefronts per SIMD the kernel must allow: 4 -> 128 VGPRs, 3 -> 168, 2 -> 256).  LCD_MINW64 / LCD_MINW256
// are build-time switches (Makefile); the per-read phase functions are instantiated per NT and inherit their kernel's budget.
#ifndef LCD_MINW64
#define LCD_MINW64 4
#endif
#ifndef LCD_MINW256
#define LCD_MINW256 4
#endif
#define LCD_MINW(NT) ((NT) == 64 ? LCD_MINW64 : (NT) == 256 ? LCD_MINW256 : 4)
template <int NT>
__global__ void __launch_bounds__(NT, LCD_MINW(NT)) lcd_poa_chain_kernel(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool,
                                                           uint8_t *arena, uint8_t *outpool, PoaChainOut *outs, LcdScoring sc,
                                                           int n_chains, int *gate, PoaSpare *spare) {
    const int cid = blockIdx.x;
    if (cid >= n_chains) return;
    if (gate && threadIdx.x == 0) atomicAdd(gate, 1); // "this workgroup is resident" (see lcd_gate_kernel)
    Smem &sm = g_smem;
    extern __shared__ int lds_pool[]; // [row ring | query cache], re-used by the re-sort; sized per launch (PoaChain.lds_words)
    int *ring = lds_pool;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PoaChain &ch = chains[cid]; // (read where it lies: a copy is 200 B of every lane's private segment, and grow_dp_region / chain_output take it by reference anyway)
    // ring slots hold PoaChain.wmax columns: the class's widest window (4 * NT), or -- single-wavefront banded chains -- the narrower
    // window the host expects the band to fit (64 / 128 columns: LDS per chain is what limits how many of them share a CU)
    const int ring_cols = (NT == 64 || ch.solo) ? ch.wmax : 4 * NT;
    const int ring_k = (NT == 64 || ch.solo) && ch.ring_k > 2 ? ch.ring_k : Cfg<NT>::K;
    const int ring16 = (NT == 64 && !ch.solo && ch.cert == 1 && ch.ring16) ? 1 : 0;
    const int ring_words = ring_k * 3 * ring_cols / (ring16 ? 2 : 1);
    uint8_t *sseq = (uint8_t *)(lds_pool + ring_words);
    const PoaLayout L = poa_layout(ch.node_cap, ch.edge_cap, ch.rid_words, ch.max_len, ch.cell_cap, ch.n_reads, ch.spill_x, ch.cert);
    uint8_t *ws = arena + ch.ws_off;
    int my_slot = -1;
    if (ch.slot_flags) { // pooled arenas: claim a slot (this CU's own ones first; anything free otherwise; wait if the pool is exhausted)
        if (tid == 0) {
            int *flags = (int *)(uintptr_t)ch.slot_flags;
            const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), xc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
            const unsigned raw = ((xc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
            const int *rank = (const int *)(uintptr_t)ch.cu_rank;
            int r = rank ? rank[raw] : -1;
            if (r < 0) r = (int)raw;
            const unsigned ns = (unsigned)ch.n_slots, start = ((unsigned)r * (unsigned)ch.per_cu) % ns;
            int slot = -1;
            for (unsigned sweeps = 0; slot < 0 && sweeps < (1u << 20); ++sweeps) {
                for (unsigned t = 0; t < ns; ++t) {
                    const unsigned q = (start + t) % ns;
                    if (atomicCAS(flags + q, 0, 1) == 0) { slot = (int)q; break; }
                }
                if (slot < 0) __builtin_amdgcn_s_sleep(64);
            }
            sm.bc[0] = slot;
        }
        __syncthreads();
        my_slot = sm.bc[0];
        __syncthreads();
        if (my_slot < 0) { // cannot happen while slot holders make progress; never run on somebody else's arena
            if (tid == 0) { PoaChainOut o = PoaChainOut(); o.status = LCD_ERR_SYNC; outs[cid] = o; }
            return;
        }
        ws = (uint8_t *)(uintptr_t)ch.ws_off + (uint64_t)my_slot * ch.slot_bytes;
    }
    // The chain's context (sixty pointers into its arena, capacities, counters) is passed by reference to the non-inlined phase functions.  As a local it lives in
    // the private segment: 700 bytes per LANE, every uniform field 64 times over, and each phase call re-reads the frame through the vector memory path -- with
    // 4 096 resident chains those frames do not stay in the caches.  One copy per WAVEFRONT in LDS instead for the classes up to 256 threads (the wavefronts of
    // a workgroup keep separate copies exactly as their lanes did: every update is made by all of them, unsynchronised); the two widest classes have no LDS to
    // spare and keep theirs in the private segment.
    constexpr bool CTX_LDS = NT <= 256;
    __shared__ Ctx s_ctx[CTX_LDS ? NT / 64 : 1];
    Ctx g_private;
    Ctx &g = *(CTX_LDS ? &s_ctx[threadIdx.x >> 6] : &g_private);
    // DP region: 4 * cell_cap bytes.  Windowed / systolic rows: [direction codes: cell_cap B | predecessor ordinals: cell_cap B (one row
    // in four may have >= 2 predecessors) | spilled value rows: 2 * cell_cap B].  Generic rows: int32 H, E1, E2 planes of cell_cap / 3
    // cells.  Whatever does not fit ends the chain with LCD_ERR_CELLS and the host re-runs it with a larger arena.
    g.H = (int *)(ws + L.H); g.E1 = g.H + ch.cell_cap / 3; g.E2 = g.E1 + ch.cell_cap / 3;
    g.code8 = ws + L.H; g.ord = (int *)(ws + L.H + lcd_align_up(ch.cell_cap, 16)); g.spill = (int *)(ws + L.H + (ch.spill_x > 2 ? 5 : 2) * lcd_align_up(ch.cell_cap, 16));
    g.ooff = (uint32_t *)(ws + L.ooff); g.spoff = (uint32_t *)(ws + L.spoff);
    g.rbeg = (int *)(ws + L.rbeg); g.rend = (int *)(ws + L.rend); g.roff = (uint32_t *)(ws + L.roff);
    g.ml = (int *)(ws + L.mpl); g.mr = (int *)(ws + L.mpr);
    g.idx2node = (int *)(ws + L.idx2node); g.node2idx = (int *)(ws + L.node2idx); g.remain = (int *)(ws + L.remain);
    g.deg = (int *)(ws + L.deg); g.queue = (int *)(ws + L.queue);
    g.out_head = (int *)(ws + L.n_out_head); g.out_tail = (int *)(ws + L.n_out_tail);
    g.in_head = (int *)(ws + L.n_in_head); g.in_tail = (int *)(ws + L.n_in_tail);
    g.nin = (int *)(ws + L.n_nin); g.aligned = (int *)(ws + L.n_aligned);
    g.e_from = (int *)(ws + L.e_from); g.e_to = (int *)(ws + L.e_to); g.e_w = (int *)(ws + L.e_w);
    g.e_next_out = (int *)(ws + L.e_next_out); g.e_next_in = (int *)(ws + L.e_next_in);
    g.rid = (unsigned long long *)(ws + L.rid);
    g.cig_node0 = g.cig_node = (int *)(ws + L.cig_node); g.cig_qpos0 = g.cig_qpos = (int *)(ws + L.cig_qpos);
    g.base = ws + L.n_base; g.imap = ws + L.imap;
    g.het = (int *)(ws + L.het); g.clu = (int *)(ws + L.clu); g.nclu = (int *)(ws + L.nclu); g.prof = ws + L.prof;
    g.pl_start = (int *)(ws + L.pl_start); g.pl_pidx = (int *)(ws + L.pl_pidx); g.pl_bonus = (int *)(ws + L.pl_bonus);
    g.pl_rem = (int *)(ws + L.pl_rem); g.pl_base = ws + L.pl_base;
    g.e_slot = (int *)(ws + L.e_slot); g.plan_valid = 0; g.plan_bi = g.plan_ei = g.plan_rend = 0;
    g.aa_node = (int *)(ws + L.aa_node); g.aa_flag = (int *)(ws + L.aa_flag); g.aa_eid = (int *)(ws + L.aa_eid);
    g.tb = (int *)(ws + L.tb); g.cert = (int *)(ws + L.cert); g.cert_on = ch.cert == 2 ? 2 : NT <= 256 ? ch.cert : 0; g.cert_hist = -1; g.alg_adjust = 0; g.cert_generic = 0; g.cert_generic_seen = 0; g.cert_sest = 0; g.cert_ubtop = 0; g.cert_bztop = 0; g.cert_cells0 = 0;
    g.node_cap = ch.node_cap; g.edge_cap = ch.edge_cap; g.rid_words = ch.rid_words; g.cell_cap = ch.cell_cap;
    g.spill_x = ch.spill_x < 2 ? 2 : ch.spill_x; g.wmax = ch.wmax; g.pool_words = ch.lds_words; g.seq_cap = (ch.lds_words - ring_words) * 4; g.ring16 = ring16; g.ring_k = ring_k; g.plan_k = (NT == 64 || ch.solo) && ring_k > 2 && ch.wmax < 256 ? 2 : ring_k; // (slots beyond 2 of a chain laid out for a narrow window: not there when a read needs a wider one)
    {   // deadline = a floor (LcdScoring.wd_s, 30 s) + the chain's own work at a rate far below the slowest class on a crowded chip (5e7 cells/s; the generic rows
        // alone on a CU run 2.4e8): a legitimately long chain -- ultra-long noisy reads, 60x depth, unbanded K2 rows over a graph twice the read -- is not a hang
        const unsigned long long width = ch.mode ? 2ull * (unsigned long long)(ch.max_len > 0 ? ch.max_len : 1) : 1024ull;
        unsigned long long work_s = (unsigned long long)(ch.n_reads > 0 ? ch.n_reads : 1) * (unsigned long long)(ch.max_len > 0 ? ch.max_len : 1) * width / 50000000ull;
        { const unsigned long long cap_s = 10ull * (unsigned long long)(sc.wd_s > 0 ? sc.wd_s : 30); if (work_s > cap_s) work_s = cap_s; } // (the unbanded worst case of 60 reads of 50 kb is 6 000 s -- a hung long chain must still be caught: at most ten floors on top of the floor)
        g.wd_deadline = (unsigned long long)clock64() + ((unsigned long long)(sc.wd_s > 0 ? sc.wd_s : 30) + work_s) * 2400000000ull; // (~2.4 GHz shader clock: the bound is about seconds, not exact)
    }
#ifdef LCD_X_INCSTAT
    for (int k_ = 0; k_ < 13; ++k_) g.inc_stat[k_] = 0;
#endif
#ifdef LCD_X_PHASESTAT
    for (int k_ = 0; k_ < 24; ++k_) g.pstat[k_] = 0;
#endif
#ifdef LCD_X_BYTESTAT
    for (int k_ = 0; k_ < 16; ++k_) g.bstat[k_] = 0;
#endif
    g.cut = g.prof; g.cut_valid = 0; g.upd_new = g.upd_newe = g.upd_moved = 0; g.inc_off = (sc.dbg >> 10) & 1;
    g.mm_valid = 0; g.topo_mode = (sc.dbg >> 6) & 7; g.solo = NT == 256 ? ch.solo : 0; g.n_node = 2; g.n_edge = 0; g.status = LCD_OK; g.t_dp = g.t_bt = 0; g.t_plan = g.t_poll = 0; g.t_kahn = 0; g.t_bp = 0; g.t_setup = 0;
    // The long chains are the latency of a submission at every depth, and next to 8 - 12 other wavefronts of their CU each of theirs issues when the arbiter gets round
    // to it: highest wavefront priority for them (the others fill the slots their dependency stalls leave).  LCD_DBG bit 512: off
    if constexpr (NT == 256) { if (ch.solo && !(sc.dbg & 512)) __builtin_amdgcn_s_setprio(3); }
    const long long t_begin = clock64();
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
    unsigned long long t_graph = 0, t_sub = 0, t_add = 0;
    if (tid == 0)
        for (int i = 0; i < 2; ++i) {
            g.base[i] = 4; g.out_head[i] = g.out_tail[i] = g.in_head[i] = g.in_tail[i] = -1; g.nin[i] = 0; g.aligned[i] = i;
        }
    __syncthreads();
    unsigned long long cells = 0, aligned_bases = 0;
#ifdef LCD_X_ROWSTAT
    unsigned long long chg_stat_ = 0;
#endif
    int n_aligned_reads = 0, backbone_len = 0;
    const int n_seq = ch.n_reads;
    const PoaRead *rd = reads + ch.read0;
    for (int i = 0; i < n_seq && g.status == LCD_OK; ++i) {
        const PoaRead r = rd[i];
        if (r.skip) continue;
        if (__syncthreads_or((unsigned long long)clock64() > g.wd_deadline)) { g.status = LCD_ERR_WATCHDOG; break; } // (one decision for the workgroup: its wavefronts meet at barriers inside the phases)
        int exc_beg = 0, exc_end = 1, beg_cut = 0, end_cut = 0;
        if (ch.mode == 0 && i != 0) {
            const long long ts0 = clock64();
            beg_cut = r.read_beg - 1; end_cut = r.len - r.read_end;
            // A read anchored at the backbone's first and last base: node 2 hangs on the source (index 0) and the last backbone node leads to the sink (the last
            // index), so the sweeps of oracle/poa.c subgraph_nodes end at once with (source, sink) whatever else the graph holds -- no sweep, no per-row extremes
            if (!(backbone_len > 0 && r.ref_beg == 1 && r.ref_end == backbone_len)) {
                if (!g.mm_valid && g.cut_valid) { compute_mm<NT>(g); LCD_BS(15, 60ull * (unsigned long long)g.n_node); }
                if (wave == 0) {
                    int eb, ee;
                    subgraph_nodes_wave0(g, lane, r.ref_beg + 1, r.ref_end + 1, &eb, &ee);
                    if (lane == 0) { sm.bc[2] = eb; sm.bc[3] = ee; }
                }
                __syncthreads();
                exc_beg = sm.bc[2]; exc_end = sm.bc[3];
                __syncthreads();
            }
            t_sub += (unsigned long long)(clock64() - ts0);
        }
        const uint8_t *seq = pool + r.seq_off + beg_cut;
        const int len = r.len - beg_cut - end_cut;
        int n_cig = 0;
        if (g.n_node > 2) {
            for (;;) { // (a read whose rows outgrow the chain's DP region is repeated in a larger one from the launch set's spare pool: grow_dp_region)
                const unsigned long long cells0 = cells; const long long adj0 = g.alg_adjust;
                n_cig = align_to_subgraph<NT>(g, sm, ring, sseq, sc, ch.mode == 0 ? 10 : -1, ch.mode == 0 ? 10 : 0, exc_beg, exc_end, seq, len, &cells);
                if (g.status != LCD_ERR_CELLS || !grow_dp_region<NT>(g, sm, ch, spare)) break;
                cells = cells0; g.alg_adjust = adj0; g.status = LCD_OK;
            }
            if (len > 0) { aligned_bases += len; n_aligned_reads++; }
        }
        // graph update + re-sort: serial pointer work on thread 0; results published through LDS
        const long long tg0 = clock64();
        int changed = 0;
        if (len > 0 && g.status == LCD_OK && g.n_node == 2) backbone_len = len; // (this read becomes the backbone: node id of its base k is k + 1)
        if (len > 0 && g.status == LCD_OK) changed = add_alignment_block<NT>(g, sm, exc_beg, exc_end, seq, len, n_cig, i);
        // graph update, per path entry: path 8 (read), read base 1, node base 1, path scratch written 8 (aa_node, aa_flag) and read back 16, out_head 4, the first
        // out-edge's e_w, e_to, e_next_out 12, e_w written 4, read set read + written 16, aa_eid 4 (= 74 B); a new node 28 B of node arrays, a new edge 20 B + its read set
        if (len > 0) LCD_BS(6, n_cig > 0 ? 74ull * (unsigned long long)n_cig + 28ull * (unsigned long long)g.upd_new + (20ull + 8ull * (unsigned long long)g.rid_words) * (unsigned long long)g.upd_newe
                                         : (unsigned long long)len * (30ull + 20ull + 8ull * (unsigned long long)g.rid_words));
        t_add += (unsigned long long)(clock64() - tg0);
#ifdef LCD_X_ROWSTAT
        if (len > 0) { chg_stat_ += changed == 2 ? 1ull : changed == 1 ? (1ull << 16) : changed == 3 ? (1ull << 32) : 0ull; chg_stat_ += 1ull << 48; }
#endif
        if (changed == 1 || changed == 2) g.plan_valid = 0; // (3: weights only, every heaviest out-edge the same -- order, remain and the plan's structure stand; its bonuses were patched)
        if (changed == 2 && g.status == LCD_OK) {
            LCD_PT0();
            // full re-sort, per node: staged nin, out_head, aligned 12; jump tables written and read ~16; node2idx, idx2node, cut, remain written 13; heaviest successor
            // e_w 8 -- ~50 B -- and per edge e_to, e_next_out 8 (twice when the walk runs on the words in HBM: not counted).  Incremental: the path scratch once (12 B per
            // entry), ~60 B per row of each 64-row window fetched (counted as two windows + one per 3 new nodes), 13 B per row of the order behind the first new node (half the order on average)
            if (!topo_sort_incremental<NT>(g, sm, lds_off(lds_pool), exc_beg, exc_end, n_cig)) { LCD_PT(22); topo_sort_block<NT>(g, sm, lds_pool, ch.mode == 0); LCD_PT(21); LCD_BS(9, 50ull * (unsigned long long)g.n_node + 8ull * (unsigned long long)g.n_edge); }
            else {
                LCD_PT(22);
                LCD_BS(10, 12ull * (unsigned long long)n_cig + 3840ull * (2ull + (unsigned long long)g.upd_new / 3ull) + (g.upd_new > 0 ? 13ull * (unsigned long long)g.n_node / 2ull : 0ull));
                if (g.upd_moved) LCD_BS(11, 34ull * (unsigned long long)g.n_node);
                if (g.upd_moved) { topo_remain_block<NT>(g, sm, lds_pool); LCD_PT(20); }
#ifdef LCD_X_VERIFY_INC
                {   // (experiment) the full re-sort must give the same order, the same `remain`, and a cut wherever the incremental one says so (the DP region is free here)
                    const int n_ = g.n_node;
                    int *sv_i2n = g.H, *sv_n2i = g.H + n_, *sv_rem = g.H + 2 * n_; uint8_t *sv_cut = (uint8_t *)(g.H + 3 * n_);
                    if ((unsigned long long)n_ * 13 + 64 < g.cell_cap * 4) {
                        for (int t = tid; t < n_; t += NT) { sv_i2n[t] = g.idx2node[t]; sv_n2i[t] = g.node2idx[t]; sv_rem[t] = g.remain[t]; sv_cut[t] = g.cut[t]; }
                        __syncthreads();
                        topo_sort_block<NT>(g, sm, lds_pool, 0);
                        int bad = 0;
                        for (int t = tid; t < n_; t += NT) {
                            if (sv_i2n[t] != g.idx2node[t]) { if (!(bad & 1)) printf("[vi] chain %d read %d n %d new %d: idx %d inc %d (its n2i %d) full %d (its n2i %d)\n", cid, i, n_, g.upd_new, t, sv_i2n[t], sv_n2i[sv_i2n[t]], g.idx2node[t], g.node2idx[g.idx2node[t]]); bad |= 1; }
                            if (sv_n2i[t] != g.node2idx[t]) bad |= 2;
                            if (sv_rem[t] != g.remain[t]) { if (!(bad & 4)) printf("[vr] chain %d read %d n %d new %d: node %d remain inc %d full %d\n", cid, i, n_, g.upd_new, t, sv_rem[t], g.remain[t]); bad |= 4; }
                            if (sv_cut[t] && !g.cut[t]) { if (!(bad & 8)) printf("[vc] chain %d read %d n %d new %d: idx %d cut inc %d full %d\n", cid, i, n_, g.upd_new, t, (int)sv_cut[t], (int)g.cut[t]); bad |= 8; }
                        }
                        bad = (__syncthreads_or(bad & 1) ? 1 : 0) | (__syncthreads_or(bad & 2) ? 2 : 0) | (__syncthreads_or(bad & 4) ? 4 : 0) | (__syncthreads_or(bad & 8) ? 8 : 0);
                        if (bad) { if (tid == 0) printf("[verify-inc] chain %d read %d n %d new %d newe %d: mismatch %d\n", cid, i, n_, g.upd_new, g.upd_newe, bad); g.status = LCD_ERR_TOPO; }
                        g.mm_valid = 0;
                    }
                }
#endif
            }
        } else if (changed == 1 && g.status == LCD_OK) { topo_remain_block<NT>(g, sm, lds_pool); LCD_BS(11, 34ull * (unsigned long long)g.n_node); } // (out_head, e_w x 2, e_to x 2, e_next_out x 2, remain: ~34 B per node)
        t_graph += (unsigned long long)(clock64() - tg0);
    }
    const long long t_out0 = clock64();
    // ---------------- output: MSA rank, rows, clusters, consensus (oracle/poa.c poa_output) ----------------
    PoaChainOut out;
    out.status = g.status; out.n_cons = 0; out.cons_len[0] = out.cons_len[1] = 0; out.msa_len = 0; out.clu_n[0] = out.clu_n[1] = 0;
    out.n_node = g.n_node; out.n_edge = g.n_edge; out.n_aligned_reads = n_aligned_reads; out.cells = cells; out.cells_alg = (unsigned long long)((long long)cells + g.alg_adjust); out.aligned_bases = aligned_bases;
    chain_output<NT>(g, sm, ch, outpool, out);
    // output: the MSA rows ((reads + 2) x columns bytes written, read again for the column profile and the consensus), rank / column arrays ~24 B per node
    LCD_BS(12, 2ull * (unsigned long long)(n_seq + 2) * (unsigned long long)g.n_node + 24ull * (unsigned long long)g.n_node);
    if (tid == 0) {
        const long long t_end = clock64();
        out.t_total = (unsigned long long)(t_end - t_begin); out.t_dp = g.t_dp; out.t_bt = g.t_bt; out.t_graph = t_graph; out.t_sub = t_sub; out.t_plan = g.t_plan; out.t_poll = NT == 64 && ch.cert ? g.t_poll : g.t_kahn; /* (profiling: t_poll slot reports the serial Kahn walk; certified-band chains: t_plan / t_poll = node arrays / intervals) */
        out.hw_id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); out.xcc_id = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        out.rt_begin = rt_begin; out.rt_end = __builtin_amdgcn_s_memrealtime();
        out.t_out = (unsigned long long)(t_end - t_out0);
        out.t_bp = g.t_bp; out.t_add = t_add; out.t_sort = t_graph - t_add; out.t_setup = g.t_setup;
#ifdef LCD_X_ROWSTAT
        out.t_bt = chg_stat_;
#endif
#ifdef LCD_X_BYTESTAT
        {   // totals of the process by (threads, kind), printed whenever a launch has finished: the last line of a run holds every chain (tools/hbm_account.py)
            const int cls = (NT == 64 ? 0 : NT == 128 ? 1 : NT == 256 ? 2 : NT == 512 ? 3 : 4) * 3 + (ch.mode == 0 ? 0 : ch.cert ? 1 : 2);
            for (int k_ = 0; k_ < 16; ++k_) atomicAdd(&g_bs_tot[cls][k_], g.bstat[k_]);
            atomicAdd(&g_bs_tot[cls][16], 1ull); atomicAdd(&g_bs_tot[cls][17], (unsigned long long)n_aligned_reads); atomicAdd(&g_bs_tot[cls][18], cells);
            __threadfence();
            const unsigned key_ = (unsigned)(((unsigned long long)(uintptr_t)chains >> 8) * 2654435761ull >> 16) & 255u;
            if (atomicAdd(&g_bs_done[key_], 1u) + 1u == (unsigned)n_chains) {
                g_bs_done[key_] = 0;
                for (int c_ = 0; c_ < 15; ++c_) if (g_bs_tot[c_][16])
                    printf("[bs] %d %d %d %llu %llu %llu : %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", c_ / 3 == 0 ? 64 : c_ / 3 == 1 ? 128 : c_ / 3 == 2 ? 256 : c_ / 3 == 3 ? 512 : 1024, c_ % 3 == 0 ? 0 : 1, c_ % 3 == 1 ? 1 : 0,
                           g_bs_tot[c_][16], g_bs_tot[c_][17], g_bs_tot[c_][18], g_bs_tot[c_][0], g_bs_tot[c_][1], g_bs_tot[c_][2], g_bs_tot[c_][3], g_bs_tot[c_][4], g_bs_tot[c_][5], g_bs_tot[c_][6], g_bs_tot[c_][7],
                           g_bs_tot[c_][8], g_bs_tot[c_][9], g_bs_tot[c_][10], g_bs_tot[c_][11], g_bs_tot[c_][12], g_bs_tot[c_][13], g_bs_tot[c_][14], g_bs_tot[c_][15]);
                printf("[bs-end]\n");
            }
        }
#endif
#ifdef LCD_X_PHASESTAT
        if ((cid & 127) == 0 && NT == 64) printf("[pt] chain %d mode %d reads %d nodes %d total %llu : %llu %llu %llu %llu %llu %llu %llu %llu | %llu %llu %llu %llu %llu %llu %llu | %llu %llu %llu %llu %llu %llu %llu %llu %llu\n", cid, ch.mode, n_seq, g.n_node, (unsigned long long)(clock64() - t_begin),
            g.pstat[0], g.pstat[1], g.pstat[2], g.pstat[3], g.pstat[4], g.pstat[5], g.pstat[6], g.pstat[7], g.pstat[8], g.pstat[9], g.pstat[10], g.pstat[11], g.pstat[12], g.pstat[13], g.pstat[14],
            g.pstat[15], g.pstat[16], g.pstat[17], g.pstat[18], g.pstat[19], g.pstat[20], g.pstat[21], g.pstat[22], g.pstat[23]);
#endif
#ifdef LCD_X_INCSTAT
        if ((cid & 511) == 0) printf("[inc] chain %d mode %d reads %d nodes %d: refused pre %u ml %u new %u cnt %u cut %u cap %u q %u dry %u el %u left %u | ok %u walked %u pieces %u\n", cid, ch.mode, n_seq, g.n_node,
            g.inc_stat[0], g.inc_stat[1], g.inc_stat[2], g.inc_stat[3], g.inc_stat[4], g.inc_stat[5], g.inc_stat[6], g.inc_stat[7], g.inc_stat[8], g.inc_stat[9], g.inc_stat[10], g.inc_stat[11], g.inc_stat[12]);
#endif
        if (g.status == LCD_ERR_WATCHDOG) { out.t_plan = sm.prof[0]; out.t_poll = sm.prof[1]; out.t_bp = sm.prof[2]; out.t_add = sm.prof[3]; } // (the backtrack's last states, if that is where it was)
        outs[cid] = out;
    }
    if (my_slot >= 0) { // every store of this workgroup into the slot has completed (barrier = vmcnt(0) per wavefront) before the next owner may start
        __syncthreads();
        if (tid == 0) { __threadfence(); atomicExch((int *)(uintptr_t)ch.slot_flags + my_slot, 0); }
    }
}

// One pass over the chip recording which (XCC, SE, SH, CU) ids exist: the host turns it into the compact CU index of the arena slots.
__global__ void lcd_cu_probe_kernel(int *seen) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)), xc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        seen[((xc & 15) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)] = 1;
        for (int i = 0; i < 200; ++i) __builtin_amdgcn_s_sleep(127); // stay resident long enough for the dispatcher to reach every CU
    }
}
void lcd_launch_cu_probe(int *seen, hipStream_t stream) { hipLaunchKernelGGL(lcd_cu_probe_kernel, dim3(16384), dim3(64), 0, stream, seen); }

// Holds a stream until `target` workgroups of the wide classes have started.  A 1 024-thread chain needs ALL the vector registers of a
// CU; if the thousands of 64-thread chains of the same step are dispatched at the same time they take a few wavefront slots on every
// CU and the wide chains -- the longest ones, the step's critical path -- wait for a CU to drain completely (measured: 1.09 s instead
// of 0.39 s for the wide launch).  The narrow classes' streams therefore start with this one-lane kernel.
__global__ void lcd_gate_kernel(int *ctr, int target0, int target1) {
    // bounded (about 2 s): the gate is a scheduling hint, never a correctness condition -- if the wide launches failed or are held
    // back by something else the narrow classes simply start
    for (int spins = 0; spins < (1 << 18); ++spins) {
        // (read with a device-scope compare-and-swap that can never succeed: the counters are bumped from all 8 XCDs, whose L2s are not
        //  coherent with each other; a relaxed load -- and atomicAdd(p, 0), which the compiler folds into one -- is served from this XCD's L2
        //  and saw the counters ~100 ms late)
        if (atomicCAS(ctr, -1, -1) >= target0 && atomicCAS(ctr + 1, -1, -1) >= target1) { if (spins > ctr[4]) ctr[4] = spins; break; } // (ctr[4]: longest wait in polls, LCD_GATE_DEBUG)
        __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); // ~7 us between polls: the counters move on a millisecond scale
    }
}
void lcd_launch_gate(int *ctr, int target0, int target1, hipStream_t stream) { hipLaunchKernelGGL(lcd_gate_kernel, dim3(1), dim3(1), 0, stream, ctr, target0, target1); }

void lcd_launch_poa(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool, uint8_t *arena, uint8_t *outpool,
                    PoaChainOut *outs, LcdScoring sc, int n_chains, int threads, int lds_bytes, hipStream_t stream, int *gate, PoaSpare *spare) {
    if (n_chains <= 0) return;
    static std::once_flag attr_once[16]; // (concurrent submitters: lcd_batch_run_many is thread-safe; function attributes are per device)
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    std::call_once(attr_once[dev], [] { // allow > 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    });
#define LCD_LAUNCH(NT) hipLaunchKernelGGL(lcd_poa_chain_kernel<NT>, dim3(n_chains), dim3(NT), lds_bytes, stream, chains, reads, pool, arena, outpool, outs, sc, n_chains, gate, spare)
    if (threads <= 64) LCD_LAUNCH(64);
    else if (threads <= 128) LCD_LAUNCH(128);
    else if (threads <= 256) LCD_LAUNCH(256);
    else if (threads <= 512) LCD_LAUNCH(512);
    else LCD_LAUNCH(1024);
#undef LCD_LAUNCH
}
