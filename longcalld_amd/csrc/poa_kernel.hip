// poa_kernel.hip -- K1/K2: partial-order alignment chains on gfx950 (CDNA4, wave64).
//
// Replaces what src/align.c:762-857 (abpoa_partial_aln_msa_cons) and :872-943 (abpoa_aln_msa_cons) ask
// of abPOA.  One workgroup owns one chain (= one abpoa_t life: region x haplotype for K1, region for K2)
// and keeps the whole graph build device-resident: align read -> backtrack -> add alignment -> re-sort ->
// next read, then MSA / clustering / consensus.  No host round trip inside a chain.
//
// Workgroup size follows the DP row width: 64 threads for banded HiFi rows (<=128 columns), 256 / 1024 for
// the unbanded K2 rows, so a 4 000-column row is 4 rounds of 16 wavefronts instead of 63 serial chunks.
// Integer DP only (no MFMA):
//   * per read a "row plan" (CSR of the usable predecessors + edge bonus of every row) is built in parallel,
//     so a row costs one dependent load level instead of a linked-list walk;
//   * a row's predecessor metadata is staged once in LDS and shared by all wavefronts;
//   * the horizontal-gap recurrence is a wave-level exclusive prefix max with an LDS carry across wavefronts
//     (F[j] = max_k<j Hpre[k] - o - (j-k)e  ==  prefixmax(Hpre[k]+k e) - o - j e);
//   * the adaptive band is pulled from the predecessors' row-max columns (no scatter);
//   * only H, E1, E2 are stored (12 B/cell); the insertion run is re-derived from H in the backtrack.
// Semantics are defined by oracle/poa.c (see its header); this file must match it bit for bit.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {

constexpr int MAXP = 64;   // predecessors staged in LDS per row (more are read from the plan in HBM)
constexpr int MAXW = 16;   // wavefronts per workgroup

struct Smem {
    int tot1[2][MAXW], tot2[2][MAXW];
    int bh[MAXW], bl[MAXW], br[MAXW];
    int pb[MAXP], pe[MAXP], bonus[MAXP], pml[MAXP], pmr[MAXP];
    unsigned po[MAXP];
    int ppi[MAXP], pslot[MAXP];
    int rm[4][5];          // ring slot meta: beg, end, HBM offset, row-max leftmost / rightmost column
    int scan[MAXW];
    int bc[8];
};

template <int NT> struct Cfg;
// One LDS pool per workgroup: [row ring | query cache] during the DP, re-used as 16-bit graph arrays by the re-sort.
// (sizes are per launch: PoaChain.wmax columns per ring slot, PoaChain.lds_words in total; only the slot count is per class)
template <> struct Cfg<64> { static constexpr int K = 4; };
template <> struct Cfg<256> { static constexpr int K = 2; };
template <> struct Cfg<1024> { static constexpr int K = 2; };

struct Ctx {
    int *H, *E1, *E2;
    int *rbeg, *rend; uint32_t *roff;
    int *ml, *mr, *idx2node, *node2idx, *remain, *deg, *queue;
    int *out_head, *out_tail, *in_head, *in_tail, *nin, *aligned;
    int *e_from, *e_to, *e_w, *e_next_out, *e_next_in;
    unsigned long long *rid;
    int *cig_node, *cig_qpos;
    uint8_t *base, *imap;
    int *het, *clu, *nclu; uint8_t *prof;
    int *pl_start, *pl_pidx, *pl_bonus, *pl_rem; uint8_t *pl_base;
    int *aa_node, *aa_flag, *aa_eid;
    int wmax, seq_cap, pool_words;
    int n_node, n_edge, node_cap, edge_cap, rid_words;
    unsigned long long cell_cap;
    int status;
    unsigned long long t_dp, t_bt;
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
__device__ __forceinline__ int shr1(int identity, int v) { return dpp_take<0x138, 0xf>(identity, v); } // wave_shr:1
__device__ __forceinline__ int lane63(int v) { return __builtin_amdgcn_readlane(v, 63); }


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
    if (lane == 63) sm.scan[wave] = incl;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) { const int t = sm.scan[k]; if (k < wave) woff += t; tot += t; }
    __syncthreads();
    *total = tot;
    return woff + incl - v;
}

template <int NT>
__device__ bool add_alignment_block(Ctx &g, Smem &sm, int beg_node, int end_node, const uint8_t *seq, int len, int n_cig, int read_id) {
    const int tid = threadIdx.x;
    const int rw = read_id >> 6; const unsigned long long rbit = 1ull << (read_id & 63);
    if (g.n_node == 2) { // first read: a chain source -> bases -> sink (abpoa_add_graph_sequence)
        if (len + 2 > g.node_cap || len + 1 > g.edge_cap) { g.status = LCD_ERR_NODES; return false; }
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
        __syncthreads();
        return true;
    }
    if (n_cig == 0) return false;
    // phase 1: the node each entry lands on: existing (>= 0), new and aligned to an anchor, or plain new
    int carry = 0;
    for (int base = 0; base < n_cig; base += NT) {
        const int i = base + tid;
        int isnew = 0, flag = -2, T = 0;
        if (i < n_cig) {
            const uint8_t b = seq[g.cig_qpos[i]];
            const int node = g.cig_node[i];
            if (node >= 0) {
                if (g.base[node] == b) T = node;
                else {
                    int a = -1;
                    for (int x = g.aligned[node]; x != node; x = g.aligned[x]) if (g.base[x] == b) { a = x; break; }
                    if (a >= 0) T = a; else { isnew = 1; flag = node; }
                }
            } else { isnew = 1; flag = -1; }
        }
        int tot;
        const int rank = block_excl_scan<NT>(isnew, sm, &tot);
        if (i < n_cig) { g.aa_node[i] = isnew ? g.n_node + carry + rank : T; g.aa_flag[i] = flag; }
        carry += tot;
    }
    const int n_new = carry;
    if (g.n_node + n_new > g.node_cap) { g.status = LCD_ERR_NODES; return false; }
    __syncthreads();
    // phase 2: new nodes
    for (int i = tid; i < n_cig; i += NT) {
        const int flag = g.aa_flag[i];
        if (flag == -2) continue;
        const int id = g.aa_node[i];
        g.base[id] = seq[g.cig_qpos[i]]; g.out_head[id] = g.out_tail[id] = g.in_head[id] = g.in_tail[id] = -1; g.nin[id] = 0;
        if (flag >= 0) { g.aligned[id] = g.aligned[flag]; g.aligned[flag] = id; } else g.aligned[id] = id;
    }
    // phase 3: edges j = 0..n_cig (from path[j-1] to path[j]); existing ones gain weight + read id, the others are numbered
    carry = 0;
    for (int base = 0; base <= n_cig; base += NT) {
        const int j = base + tid;
        int need = 0;
        if (j <= n_cig) {
            const int from = j == 0 ? beg_node : g.aa_node[j - 1], to = j == n_cig ? end_node : g.aa_node[j];
            const bool from_new = j > 0 && g.aa_flag[j - 1] != -2, to_new = j < n_cig && g.aa_flag[j] != -2;
            need = 1;
            if (!from_new && !to_new)
                for (int e = g.out_head[from]; e >= 0; e = g.e_next_out[e])
                    if (g.e_to[e] == to) { g.e_w[e] += 1; g.rid[(size_t)e * g.rid_words + rw] |= rbit; need = 0; break; }
        }
        int tot;
        const int rank = block_excl_scan<NT>(need, sm, &tot);
        if (j <= n_cig) g.aa_eid[j] = need ? g.n_edge + carry + rank : -1;
        carry += tot;
    }
    const int n_newe = carry;
    if (g.n_edge + n_newe > g.edge_cap) { g.status = LCD_ERR_EDGES; return false; }
    __syncthreads(); // new-node fields (phase 2) and edge numbers are visible
    // phase 4: create + link the new edges
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
    __syncthreads();
    return true;
}

// Kahn BFS order + remain for the whole workgroup: the pointer-chasing part still runs on one lane (the FIFO order is
// inherently serial) but on 16-bit copies of the graph staged in LDS, so each dependent step costs an LDS access (~60 clk)
// instead of an HBM/L2 access (~500 clk); staging in and out is a coalesced parallel copy.  Falls back to HBM when the
// graph does not fit the pool.
template <int NT>
__device__ void topo_sort_block(Ctx &g, Smem &sm, int *lds_pool) {
    const int tid = threadIdx.x;
    const int n = g.n_node, E = g.n_edge;
    const bool fits = n < 65535 && E < 65535 && (size_t)14 * n + (size_t)6 * E + 64 <= (size_t)g.pool_words * 4;
    if (!fits) {
        if (tid == 0) { topo_sort(g); sm.bc[6] = g.status; }
        __syncthreads();
        g.status = sm.bc[6];
        __syncthreads();
        return;
    }
    unsigned short *deg = (unsigned short *)lds_pool, *queue = deg + n, *oh = queue + n, *al = oh + n, *i2n = al + n, *n2i = i2n + n, *rem = n2i + n;
    unsigned short *en = rem + n, *et = en + E, *ew = et + E;
    for (int i = tid; i < n; i += NT) { deg[i] = (unsigned short)g.nin[i]; oh[i] = (unsigned short)(g.out_head[i] + 1); al[i] = (unsigned short)g.aligned[i]; }
    for (int e = tid; e < E; e += NT) { en[e] = (unsigned short)(g.e_next_out[e] + 1); et[e] = (unsigned short)g.e_to[e]; const int w = g.e_w[e]; ew[e] = (unsigned short)(w > 65535 ? 65535 : w); }
    __syncthreads();
    if (tid == 0) {
        int qh = 0, qt = 0, index = 0;
        queue[qt++] = 0;
        while (qh < qt) {
            const int cur = queue[qh++];
            i2n[index] = (unsigned short)cur; n2i[cur] = (unsigned short)index; ++index;
            if (cur == 1) break;
            for (int e = oh[cur]; e != 0; e = en[e - 1]) {
                const int out = et[e - 1];
                const int d = deg[out] - 1; deg[out] = (unsigned short)d;
                if (d == 0) {
                    bool ok = true;
                    for (int a = al[out]; a != out; a = al[a]) if (deg[a] != 0) { ok = false; break; }
                    if (!ok) continue;
                    queue[qt++] = (unsigned short)out;
                    for (int a = al[out]; a != out; a = al[a]) queue[qt++] = (unsigned short)a;
                }
            }
        }
        if (index != n) g.status = LCD_ERR_TOPO;
        else {
            rem[1] = 0; // remain + 1
            for (int i = n - 2; i >= 0; --i) {
                const int v = i2n[i]; int mw = -1, mid = 1;
                for (int e = oh[v]; e != 0; e = en[e - 1]) if ((int)ew[e - 1] > mw) { mw = ew[e - 1]; mid = et[e - 1]; }
                rem[v] = (unsigned short)(rem[mid] + 1);
            }
        }
        sm.bc[6] = g.status;
    }
    __syncthreads();
    g.status = sm.bc[6];
    if (g.status == LCD_OK)
        for (int i = tid; i < n; i += NT) { g.idx2node[i] = i2n[i]; g.node2idx[i] = n2i[i]; g.remain[i] = (int)rem[i] - 1; }
    __syncthreads();
}

// sub-graph boundaries (oracle/poa.c subgraph_nodes): min/max sweeps on wavefront 0, result broadcast through LDS
__device__ void subgraph_nodes_wave0(Ctx &g, int lane, int inc_beg, int inc_end, int *exc_beg, int *exc_end) {
    const int bi = g.node2idx[inc_beg], ei = g.node2idx[inc_end];
    int b = bi, e = ei, up, down;
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

// LDS-only workgroup barrier: orders LDS traffic without draining the HBM store queue (vmcnt), which a
// __syncthreads() would do.  Single-wavefront workgroups need no s_barrier at all (LDS ops of one wave are in order).
template <int NT>
__device__ __forceinline__ void lds_barrier() {
    if (NT > 64) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("" ::: "memory");
}

#define LCD_RL(v, t) __builtin_amdgcn_readlane((v), (t))

// six per-symbol counters packed 3 x 21 bits into two 64-bit words (register-resident, no dynamic array indexing)
struct Cnt6 {
    unsigned long long a, b;
    __device__ __forceinline__ Cnt6() : a(0), b(0) {}
    __device__ __forceinline__ void add(int sym) { if (sym < 3) a += 1ull << (21 * sym); else b += 1ull << (21 * (sym - 3)); }
    __device__ __forceinline__ int get(int sym) const { return (int)(((sym < 3 ? a >> (21 * sym) : b >> (21 * (sym - 3)))) & 0x1fffff); }
};
constexpr int RMAX = 4; // 64-column chunks per wavefront per sweep: NW*RMAX*64 == WMAX

// banded convex-gap global alignment of seq[0..qlen) to the sub-graph (beg_node,end_node); returns #cigar
// entries written to cig_node/cig_qpos in start->end order (block-uniform result).
//
// Row data flow: every row is written to HBM (the backtrack and far predecessors read it there) AND, when it
// fits, into a K-slot LDS ring; a predecessor that is still in the ring is read from LDS, so the common
// row-to-row dependency never waits for an HBM store->load round trip.  HBM rows are only read once a full
// barrier has drained the stores issued before it (tracked with last_full).
template <int NT>
__device__ int align_to_subgraph(Ctx &g, Smem &sm, int *ring, uint8_t *sseq, const LcdScoring &sc, const int wb, int wf_milli, int beg_node, int end_node,
                                 const uint8_t *seq_hbm, int qlen, unsigned long long *cells_acc) {
    constexpr int NW = NT / 64, K = Cfg<NT>::K;
    const int WMAX = g.wmax; // ring slot capacity in columns: chosen per launch from the chains' lengths (dynamic LDS)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (qlen <= 0) return 0;
    const int bi = g.node2idx[beg_node], ei = g.node2idx[end_node];
    const int o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2, oe1 = o1 + e1, oe2 = o2 + e2;
    // w = wb<0 ? qlen : wb + (int)(wf*qlen);  wf is 0.01 or 0 on this path: (int)(0.01*q) == q/100
    const int w = wb < 0 ? qlen : wb + (int)(((long long)wf_milli * qlen) / 1000);
    const int remain_end = g.remain[end_node];
    const int n = g.n_node;
    // the read's bases are re-read by every row: keep them in LDS when they fit
    // (two explicit pointers, never one that may be either: a maybe-LDS pointer compiles to FLAT loads, whose s_waitcnt
    //  couples vmcnt and lgkmcnt and would stall every chunk behind the row stores still draining to HBM)
    const uint8_t *seq = seq_hbm;
    if (qlen > g.seq_cap) { g.status = LCD_ERR_NODES; return 0; } // read slice longer than the LDS query cache (host sizes the class)
    for (int i = tid; i < qlen; i += NT) sseq[i] = seq_hbm[i];
    // ---- reachability map over [bi, ei] ----
    if (bi == 0 && ei == n - 1) {
        for (int i = tid; i < n; i += NT) g.imap[i] = 1;
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
    __syncthreads();
    // ---- row plan: per row (by topological index) remain / base / CSR of usable predecessors + edge bonus ----
    {
        int carry = 0;
        for (int base = bi; base <= ei; base += NT) {
            const int idx = base + tid;
            int cnt = 0, v = -1;
            if (idx <= ei && g.imap[idx]) {
                v = g.idx2node[idx];
                for (int e = g.in_head[v]; e >= 0; e = g.e_next_in[e]) {
                    int pi = g.node2idx[g.e_from[e]];
                    cnt += (pi >= bi && pi < ei && g.imap[pi]);
                }
            }
            const int incl = scan_add(cnt);
            if (lane == 63) sm.scan[wave] = incl;
            __syncthreads();
            int woff = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < NW; ++k) { const int t = sm.scan[k]; if (k < wave) woff += t; tot += t; }
            const int start = carry + woff + incl - cnt;
            if (idx <= ei) {
                g.pl_start[idx] = start;
                g.pl_rem[idx] = v >= 0 ? g.remain[v] - remain_end : (1 << 30); // 1<<30: row not reachable from beg
                g.pl_base[idx] = v >= 0 ? g.base[v] : 4;
                if (v >= 0) {
                    int k = start;
                    for (int e = g.in_head[v]; e >= 0; e = g.e_next_in[e]) {
                        int pi = g.node2idx[g.e_from[e]];
                        if (pi >= bi && pi < ei && g.imap[pi]) { g.pl_pidx[k] = pi; g.pl_bonus[k] = ilog2_32(g.e_w[e]); ++k; }
                    }
                }
            }
            carry += tot;
            __syncthreads();
        }
        if (tid == 0) { g.pl_start[ei + 1] = carry; g.pl_start[ei + 2] = carry; g.pl_start[ei + 3] = carry; }
    }
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
        if ((unsigned long long)end + 1 > g.cell_cap) { g.status = LCD_ERR_CELLS; return 0; }
        const bool fits = end + 1 <= WMAX;
        if (tid == 0) { g.rbeg[bi] = 0; g.rend[bi] = end; g.roff[bi] = 0; g.ml[bi] = 0; g.mr[bi] = 0; }
        for (int j = tid; j <= end; j += NT) {
            int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            int h = j ? imax(f1, f2) : 0;
            g.H[j] = h; g.E1[j] = h - oe1; g.E2[j] = h - oe2;
            if (fits) { ring[j] = h; ring[WMAX + j] = h - oe1; ring[2 * WMAX + j] = h - oe2; }
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
    // ================= unbanded fast path (K2, wb < 0) =================
    // w = qlen makes every row [0, qlen] (oracle: beg = max(0, .. - qlen) = 0, end = min(qlen, .. + qlen) = qlen, and the
    // source row already spans it), every node is reachable, so row metadata is implicit: rbeg = 0, rend = qlen,
    // roff = (idx - bi) * W1p (row stride padded to 4 cells).  No staging, no band, no row-max.
    // FOUR cells per lane: a wavefront covers 256 columns, so a row is one sweep of the workgroup.  Predecessor rows
    // come from the LDS ring as ds_read_b128, the row goes to HBM as global_store_dwordx4, the horizontal-gap prefix is
    // 3 in-lane max + ONE interleaved DPP scan pair per 256 cells.  (The chain is issue-bound: this is ~4x fewer
    // instructions per cell than one cell per lane.)
    // Ring slot layout (words): H plane = [.. guard @3 | H(0..qlen) @4..], E1 plane @WMAX, E2 plane @2*WMAX; the guard
    // (-2^30) stands in for H[j-1] at j = 0 so the match term needs no bounds test.  Query bases are kept shifted by one
    // byte (q[j-1] at byte j) so a lane's four bases are one aligned ds_read_b32.
    if (wb < 0 && bi == 0 && ei == n - 1 && qlen + 5 <= WMAX && qlen + 5 <= NT * 4 && 2 * (qlen + 24) <= g.seq_cap) {
        const int W1 = qlen + 1, W1p = (W1 + 3) & ~3;
        if ((unsigned long long)(ei - bi) * W1p > g.cell_cap) { g.status = LCD_ERR_CELLS; return 0; }
        for (int i = bi + 1 + tid; i < ei; i += NT) { g.rbeg[i] = 0; g.rend[i] = qlen; g.roff[i] = (uint32_t)((i - bi) * (unsigned)W1p); }
        used = (unsigned long long)(ei - bi) * W1p;
        const int SLOTW = 3 * WMAX; // words per ring slot
        uint8_t *sq1 = sseq + ((qlen + 8 + 15) & ~15); // shifted copy: sq1[j] = q[j-1]
        for (int j = tid; j <= qlen + 3; j += NT) sq1[j] = (j >= 1 && j <= qlen) ? seq_hbm[j - 1] : 4;
        // source row into slot 0; guards of every slot
        for (int j = tid; j <= qlen; j += NT) {
            const int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            const int h = j ? imax(f1, f2) : 0;
            ring[4 + j] = h; ring[WMAX + j] = h - oe1; ring[2 * WMAX + j] = h - oe2;
        }
        if (tid < K) ring[tid * SLOTW + 3] = LCD_NEG * 2;
        si0 = bi; si1 = -1; si2 = -1; si3 = -1; next_slot = 1 % K;
        __syncthreads();
        last_full = bi + 1;
        // this lane's four columns
        const int c = (wave << 8) + (lane << 2);
        const bool k0 = c <= qlen, k1 = c + 1 <= qlen, k2 = c + 2 <= qlen, k3 = c + 3 <= qlen;
        const int c1a = c * e1, c2a = c * e2; // a-offsets of cell 0 (cell k adds k*e)
        int wbase = -(1 << 20);
        int w_p0 = 0, w_np = 0, w_vb = 4, w_pi0 = 0, w_b0 = 0, w_pi1 = 0, w_b1 = 0;
        for (int idx = bi + 1; idx < ei; ++idx) {
            if (idx - wbase >= 64) {
                wbase = idx;
                const int ri = idx + lane;
                w_np = 0;
                if (ri < ei) {
                    const int s0 = g.pl_start[ri], s1 = g.pl_start[ri + 1];
                    w_p0 = s0; w_np = s1 - s0; w_vb = g.pl_base[ri];
                    if (w_np > 0) { w_pi0 = g.pl_pidx[s0]; w_b0 = g.pl_bonus[s0]; }
                    if (w_np > 1) { w_pi1 = g.pl_pidx[s0 + 1]; w_b1 = g.pl_bonus[s0 + 1]; }
                }
            }
            const int wk = idx - wbase;
            const int p0 = LCD_RL(w_p0, wk), np = LCD_RL(w_np, wk);
            const int vb = LCD_RL(w_vb, wk);
            const int pi0 = LCD_RL(w_pi0, wk), bz0 = LCD_RL(w_b0, wk), pi1 = LCD_RL(w_pi1, wk), bz1 = LCD_RL(w_b1, wk);
            const int sl0 = np > 0 ? LCD_SLOT_OF(pi0) : 0, sl1 = np > 1 ? LCD_SLOT_OF(pi1) : 0;
            const bool fastrow = np >= 1 && np <= 2 && sl0 >= 0 && sl1 >= 0;
            if (!fastrow) { // predecessors that are not in the ring are read from HBM: their stores must have drained
                int far = -1;
                if (np > 0 && sl0 < 0) far = pi0;
                if (np > 1 && sl1 < 0) far = imax(far, pi1);
                if (np > 2) far = 1 << 30;
                if (far >= last_full) { __syncthreads(); last_full = idx; }
            }
            const size_t off = (size_t)(idx - bi) * W1p;
            const int slot = next_slot;
            int *rS = ring + slot * SLOTW;
            // ---- phase A: Hpre of the four cells ----
            int h0, h1, h2, h3, u0, u1, u2, u3, v0, v1, v2, v3; // Hpre, E1in, E2in
            if (fastrow) {
                const unsigned qw = *(const unsigned *)(sq1 + c); // q[c-1], q[c], q[c+1], q[c+2]
                int s0, s1, s2, s3;
                {
                    const int q0 = qw & 255, q1 = (qw >> 8) & 255, q2 = (qw >> 16) & 255, q3 = qw >> 24;
                    s0 = (vb >= 4 || q0 >= 4) ? 0 : (vb == q0 ? sc.match : -sc.mismatch);
                    s1 = (vb >= 4 || q1 >= 4) ? 0 : (vb == q1 ? sc.match : -sc.mismatch);
                    s2 = (vb >= 4 || q2 >= 4) ? 0 : (vb == q2 ? sc.match : -sc.mismatch);
                    s3 = (vb >= 4 || q3 >= 4) ? 0 : (vb == q3 ? sc.match : -sc.mismatch);
                }
                const int *q = ring + sl0 * SLOTW;
                const int hm = q[3 + c];                                        // H[c-1] (guard at c == 0)
                const int4 hv = *(const int4 *)(q + 4 + c);                     // H[c..c+3]
                const int4 av = *(const int4 *)(q + WMAX + c), bv = *(const int4 *)(q + 2 * WMAX + c);
                const int m0 = hm + s0 + bz0, m1 = hv.x + s1 + bz0, m2 = hv.y + s2 + bz0, m3 = hv.z + s3 + bz0;
                u0 = av.x + bz0; u1 = av.y + bz0; u2 = av.z + bz0; u3 = av.w + bz0;
                v0 = bv.x + bz0; v1 = bv.y + bz0; v2 = bv.z + bz0; v3 = bv.w + bz0;
                int n0 = m0, n1 = m1, n2 = m2, n3 = m3;
                if (np > 1) {
                    const int *r = ring + sl1 * SLOTW;
                    const int gm = r[3 + c];
                    const int4 gv = *(const int4 *)(r + 4 + c);
                    const int4 cv = *(const int4 *)(r + WMAX + c), dv = *(const int4 *)(r + 2 * WMAX + c);
                    n0 = imax(n0, gm + s0 + bz1); n1 = imax(n1, gv.x + s1 + bz1); n2 = imax(n2, gv.y + s2 + bz1); n3 = imax(n3, gv.z + s3 + bz1);
                    u0 = imax(u0, cv.x + bz1); u1 = imax(u1, cv.y + bz1); u2 = imax(u2, cv.z + bz1); u3 = imax(u3, cv.w + bz1);
                    v0 = imax(v0, dv.x + bz1); v1 = imax(v1, dv.y + bz1); v2 = imax(v2, dv.z + bz1); v3 = imax(v3, dv.w + bz1);
                }
                h0 = imax(n0, imax(u0, v0)); h1 = imax(n1, imax(u1, v1)); h2 = imax(n2, imax(u2, v2)); h3 = imax(n3, imax(u3, v3));
            } else {
                auto slow = [&](const int cc, const bool act, int &hp, int &ev1, int &ev2) {
                    int mx = LCD_NEG, e1i = LCD_NEG, e2i = LCD_NEG;
                    if (act) {
                        int s = 0;
                        if (cc >= 1) { const uint8_t qb = sq1[cc]; s = (vb >= 4 || qb >= 4) ? 0 : (vb == qb ? sc.match : -sc.mismatch); }
                        for (int t = 0; t < np; ++t) {
                            const int pi = t == 0 ? pi0 : t == 1 ? pi1 : g.pl_pidx[p0 + t];
                            const int bonus = t == 0 ? bz0 : t == 1 ? bz1 : g.pl_bonus[p0 + t];
                            const int sl = t == 0 ? sl0 : t == 1 ? sl1 : -1;
                            if (sl >= 0) {
                                const int *q = ring + sl * SLOTW;
                                if (cc >= 1) mx = imax(mx, q[3 + cc] + s + bonus);
                                e1i = imax(e1i, q[WMAX + cc] + bonus); e2i = imax(e2i, q[2 * WMAX + cc] + bonus);
                            } else {
                                const size_t po = (size_t)(pi - bi) * W1p;
                                if (cc >= 1) mx = imax(mx, g.H[po + cc - 1] + s + bonus);
                                e1i = imax(e1i, g.E1[po + cc] + bonus); e2i = imax(e2i, g.E2[po + cc] + bonus);
                            }
                        }
                    }
                    hp = imax(mx, imax(e1i, e2i)); ev1 = e1i; ev2 = e2i;
                };
                slow(c, k0, h0, u0, v0); slow(c + 1, k1, h1, u1, v1); slow(c + 2, k2, h2, u2, v2); slow(c + 3, k3, h3, u3, v3);
            }
            // ---- F: A[k] = Hpre[k] + k*e; in-lane inclusive prefix, then one scan pair over the lane totals ----
            const int a10 = k0 ? h0 + c1a : LCD_NEG * 2, a11 = imax(a10, k1 ? h1 + c1a + e1 : LCD_NEG * 2), a12 = imax(a11, k2 ? h2 + c1a + 2 * e1 : LCD_NEG * 2);
            const int a20 = k0 ? h0 + c2a : LCD_NEG * 2, a21 = imax(a20, k1 ? h1 + c2a + e2 : LCD_NEG * 2), a22 = imax(a21, k2 ? h2 + c2a + 2 * e2 : LCD_NEG * 2);
            int t1 = imax(a12, k3 ? h3 + c1a + 3 * e1 : LCD_NEG * 2), t2 = imax(a22, k3 ? h3 + c2a + 3 * e2 : LCD_NEG * 2);
            scan_max2(t1, t2);
            int x1 = shr1(LCD_NEG * 2, t1), x2 = shr1(LCD_NEG * 2, t2); // exclusive prefix over the lanes of this wavefront
            if (NW > 1) {
                const int buf = idx & 1;
                if (lane == 63) { sm.tot1[buf][wave] = t1; sm.tot2[buf][wave] = t2; }
                lds_barrier<NT>();
#pragma unroll
                for (int k = 0; k < NW; ++k) if (k < wave) { x1 = imax(x1, sm.tot1[buf][k]); x2 = imax(x2, sm.tot2[buf][k]); }
            }
            // ---- phase B: F, H, E of the four cells; row to the ring slot and to HBM ----
            // (at column 0 the prefix is -2^30, so f clamps to LCD_NEG exactly as the oracle's "j > beg" test does)
            if (k0) {
                int4 H4, A4, B4;
#define LCD_CELL(k, hp, ev1, ev2, p1, p2, HO, AO, BO)                                                   \
                {                                                                                        \
                    const int f1 = imax(LCD_NEG, (p1) - o1 - c1a - (k) * e1), f2 = imax(LCD_NEG, (p2) - o2 - c2a - (k) * e2); \
                    int h = imax(hp, imax(f1, f2)); if (h < LCD_NEG) h = LCD_NEG;                          \
                    int eo1 = imax(h - oe1, (ev1) - e1), eo2 = imax(h - oe2, (ev2) - e2);                  \
                    if (eo1 < LCD_NEG) eo1 = LCD_NEG;                                                      \
                    if (eo2 < LCD_NEG) eo2 = LCD_NEG;                                                      \
                    HO = h; AO = eo1; BO = eo2;                                                            \
                }
                LCD_CELL(0, h0, u0, v0, x1, x2, H4.x, A4.x, B4.x)
                LCD_CELL(1, h1, u1, v1, imax(x1, a10), imax(x2, a20), H4.y, A4.y, B4.y)
                LCD_CELL(2, h2, u2, v2, imax(x1, a11), imax(x2, a21), H4.z, A4.z, B4.z)
                LCD_CELL(3, h3, u3, v3, imax(x1, a12), imax(x2, a22), H4.w, A4.w, B4.w)
#undef LCD_CELL
                if (k3) {
                    if (!(sc.dbg & 1)) { *(int4 *)(g.H + off + c) = H4; *(int4 *)(g.E1 + off + c) = A4; *(int4 *)(g.E2 + off + c) = B4; }
                    if (!(sc.dbg & 2)) *(int4 *)(rS + 4 + c) = H4; *(int4 *)(rS + WMAX + c) = A4; *(int4 *)(rS + 2 * WMAX + c) = B4;
                } else {
                    g.H[off + c] = H4.x; g.E1[off + c] = A4.x; g.E2[off + c] = B4.x; rS[4 + c] = H4.x; rS[WMAX + c] = A4.x; rS[2 * WMAX + c] = B4.x;
                    if (k1) { g.H[off + c + 1] = H4.y; g.E1[off + c + 1] = A4.y; g.E2[off + c + 1] = B4.y; rS[5 + c] = H4.y; rS[WMAX + c + 1] = A4.y; rS[2 * WMAX + c + 1] = B4.y; }
                    if (k2) { g.H[off + c + 2] = H4.z; g.E1[off + c + 2] = A4.z; g.E2[off + c + 2] = B4.z; rS[6 + c] = H4.z; rS[WMAX + c + 2] = A4.z; rS[2 * WMAX + c + 2] = B4.z; }
                }
            }
            if (slot == 0) si0 = idx; else if (slot == 1) si1 = idx; else if (slot == 2) si2 = idx; else si3 = idx;
            next_slot = (slot + 1) % K;
            // publish the ring slot to the other wavefronts before the next row's phase A
            lds_barrier<NT>();
        }
    } else {
    // Plan window: every 64 rows each lane loads the plan of one upcoming row (start, #preds, remain, base and the first two
    // predecessor entries); rows then take it by v_readlane.  The row loop therefore issues no HBM load in the common case,
    // so it never waits (vmcnt is in-order) behind the row stores that are still draining.
    int wbase = -(1 << 20);
    int w_p0 = 0, w_np = 0, w_rem = 1 << 30, w_vb = 4, w_pi0 = 0, w_b0 = 0, w_pi1 = 0, w_b1 = 0;
    // ---- rows ----
    for (int idx = bi + 1; idx < ei; ++idx) {
        if (idx - wbase >= 64) {
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
        } else {
            // pass 1: where does each predecessor row live (ring slot or HBM only)
            if (tid < np && tid < MAXP) {
                const int slot = LCD_SLOT_OF(my_pi);
                sm.bonus[tid] = my_bonus; sm.ppi[tid] = my_pi; sm.pslot[tid] = slot;
            }
            lds_barrier<NT>();
            {
                int far_pi = np > MAXP ? (1 << 30) : -1;
                const int ns = imin(np, MAXP);
                for (int t = 0; t < ns; ++t) if (sm.pslot[t] < 0) far_pi = imax(far_pi, sm.ppi[t]);
                if (far_pi >= last_full) { __syncthreads(); last_full = idx; }
            }
            // pass 2: metadata from registers (row just computed), ring meta (LDS) or HBM
            if (tid < np && tid < MAXP) {
                const int pi = my_pi, slot = sm.pslot[tid];
                if (pi == last_idx) { sm.pb[tid] = last_beg; sm.pe[tid] = last_end; sm.po[tid] = last_off; sm.pml[tid] = last_ml; sm.pmr[tid] = last_mr; }
                else if (slot >= 0) { sm.pb[tid] = sm.rm[slot][0]; sm.pe[tid] = sm.rm[slot][1]; sm.po[tid] = (unsigned)sm.rm[slot][2]; sm.pml[tid] = sm.rm[slot][3]; sm.pmr[tid] = sm.rm[slot][4]; }
                else { sm.pb[tid] = g.rbeg[pi]; sm.pe[tid] = g.rend[pi]; sm.po[tid] = g.roff[pi]; sm.pml[tid] = g.ml[pi]; sm.pmr[tid] = g.mr[pi]; }
            }
            lds_barrier<NT>();
        }
        // band: pulled from the predecessors' row-max columns (same values the oracle pushes to successors)
        int mplv = 1 << 30, mprv = 0, minpb = 1 << 30, maxpe = -1;
        for (int t = 0; t < np; ++t) {
            int pb, pe, pml, pmr;
            if (regstage) { pb = LCD_RL(r_pb, t); pe = LCD_RL(r_pe, t); pml = LCD_RL(r_pml, t); pmr = LCD_RL(r_pmr, t); }
            else if (t < MAXP) { pb = sm.pb[t]; pe = sm.pe[t]; pml = sm.pml[t]; pmr = sm.pmr[t]; }
            else { const int pi = g.pl_pidx[p0 + t]; pb = g.rbeg[pi]; pe = g.rend[pi]; pml = g.ml[pi]; pmr = g.mr[pi]; }
            if (pb > pe) continue;
            minpb = imin(minpb, pb); maxpe = imax(maxpe, pe);
            mplv = imin(mplv, pml + 1); mprv = imax(mprv, pmr + 1);
        }
        int beg = imin(mplv, qlen - rem) - w; if (beg < 0) beg = 0;
        int end = imax(mprv, qlen - rem) + w; if (end > qlen) end = qlen;
        if (beg < minpb) beg = minpb;
        if (end > maxpe + 1) end = maxpe + 1;
        if (beg > end) {
            if (tid == 0) { g.rbeg[idx] = 1; g.rend[idx] = 0; g.roff[idx] = (uint32_t)used; g.ml[idx] = 0; g.mr[idx] = 0; }
            lds_barrier<NT>();
            continue;
        }
        const unsigned long long off = used;
        const int width = end - beg + 1;
        used += (unsigned long long)width;
        if (used > g.cell_cap) { g.status = LCD_ERR_CELLS; return 0; }
        if (tid == 0) { g.rbeg[idx] = beg; g.rend[idx] = end; g.roff[idx] = (uint32_t)off; }
        const int nchunks = (width + 63) >> 6;
        const bool fits = width <= WMAX;
        const int slot = fits ? next_slot : -1;
        int *rH = ring + (size_t)(slot < 0 ? 0 : slot) * 3 * WMAX, *rE1 = rH + WMAX, *rE2 = rH + 2 * WMAX;
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
                    else if (t < MAXP) { pb = sm.pb[t]; pe = sm.pe[t]; po = sm.po[t]; bonus = sm.bonus[t]; ps = sm.pslot[t]; }
                    else { const int pi = g.pl_pidx[p0 + t]; pb = g.rbeg[pi]; pe = g.rend[pi]; po = g.roff[pi]; bonus = g.pl_bonus[p0 + t]; ps = -1; }
                    if (!act) continue;
                    if (ps >= 0) {
                        const int *qH = ring + (size_t)ps * 3 * WMAX;
                        if (j >= 1 && j - 1 >= pb && j - 1 <= pe) mx = imax(mx, qH[j - 1 - pb] + s + bonus);
                        if (j >= pb && j <= pe) { e1i = imax(e1i, qH[WMAX + (j - pb)] + bonus); e2i = imax(e2i, qH[2 * WMAX + (j - pb)] + bonus); }
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
    } // banded / generic rows
    __syncthreads();
    *cells_acc += used;
    const long long t_bt0 = clock64();
    g.t_dp += (unsigned long long)(t_bt0 - t_dp0);
    // ---- end node: best predecessor at column qlen, then backtrack (thread 0) ----
    if (tid == 0) {
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
        if (br >= 0 && best > LCD_NEG / 2 && !(sc.dbg & 5)) {
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
        sm.bc[0] = n_cig; sm.bc[1] = g.status;
    }
    __syncthreads();
    const int n_cig = sm.bc[0];
    g.status = sm.bc[1];
    __syncthreads();
    g.t_bt += (unsigned long long)(clock64() - t_bt0);
    return n_cig;
}

} // namespace

template <int NT>
__global__ void __launch_bounds__(NT) lcd_poa_chain_kernel(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool,
                                                           uint8_t *arena, uint8_t *outpool, PoaChainOut *outs, LcdScoring sc,
                                                           int n_chains) {
    const int cid = blockIdx.x;
    if (cid >= n_chains) return;
    __shared__ Smem sm;
    extern __shared__ int lds_pool[]; // [row ring | query cache], re-used by the re-sort; sized per launch (PoaChain.lds_words)
    int *ring = lds_pool;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const PoaChain ch = chains[cid];
    uint8_t *sseq = (uint8_t *)(lds_pool + Cfg<NT>::K * 3 * ch.wmax);
    const PoaLayout L = poa_layout(ch.node_cap, ch.edge_cap, ch.rid_words, ch.max_len, ch.cell_cap, ch.n_reads);
    uint8_t *ws = arena + ch.ws_off;
    Ctx g;
    g.H = (int *)(ws + L.H); g.E1 = (int *)(ws + L.E1); g.E2 = (int *)(ws + L.E2);
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
    g.cig_node = (int *)(ws + L.cig_node); g.cig_qpos = (int *)(ws + L.cig_qpos);
    g.base = ws + L.n_base; g.imap = ws + L.imap;
    g.het = (int *)(ws + L.het); g.clu = (int *)(ws + L.clu); g.nclu = (int *)(ws + L.nclu); g.prof = ws + L.prof;
    g.pl_start = (int *)(ws + L.pl_start); g.pl_pidx = (int *)(ws + L.pl_pidx); g.pl_bonus = (int *)(ws + L.pl_bonus);
    g.pl_rem = (int *)(ws + L.pl_rem); g.pl_base = ws + L.pl_base;
    g.aa_node = (int *)(ws + L.aa_node); g.aa_flag = (int *)(ws + L.aa_flag); g.aa_eid = (int *)(ws + L.aa_eid);
    g.node_cap = ch.node_cap; g.edge_cap = ch.edge_cap; g.rid_words = ch.rid_words; g.cell_cap = ch.cell_cap;
    g.wmax = ch.wmax; g.pool_words = ch.lds_words; g.seq_cap = (ch.lds_words - Cfg<NT>::K * 3 * ch.wmax) * 4;
    g.n_node = 2; g.n_edge = 0; g.status = LCD_OK; g.t_dp = g.t_bt = 0;
    const long long t_begin = clock64();
    unsigned long long t_graph = 0, t_sub = 0;
    if (tid == 0)
        for (int i = 0; i < 2; ++i) {
            g.base[i] = 4; g.out_head[i] = g.out_tail[i] = g.in_head[i] = g.in_tail[i] = -1; g.nin[i] = 0; g.aligned[i] = i;
        }
    __syncthreads();
    unsigned long long cells = 0, aligned_bases = 0;
    int n_aligned_reads = 0;
    const int n_seq = ch.n_reads;
    const PoaRead *rd = reads + ch.read0;
    for (int i = 0; i < n_seq && g.status == LCD_OK; ++i) {
        const PoaRead r = rd[i];
        if (r.skip) continue;
        int exc_beg = 0, exc_end = 1, beg_cut = 0, end_cut = 0;
        if (ch.mode == 0 && i != 0) {
            const long long ts0 = clock64();
            beg_cut = r.read_beg - 1; end_cut = r.len - r.read_end;
            if (wave == 0) {
                int eb, ee;
                subgraph_nodes_wave0(g, lane, r.ref_beg + 1, r.ref_end + 1, &eb, &ee);
                if (lane == 0) { sm.bc[2] = eb; sm.bc[3] = ee; }
            }
            __syncthreads();
            exc_beg = sm.bc[2]; exc_end = sm.bc[3];
            __syncthreads();
            t_sub += (unsigned long long)(clock64() - ts0);
        }
        const uint8_t *seq = pool + r.seq_off + beg_cut;
        const int len = r.len - beg_cut - end_cut;
        int n_cig = 0;
        if (g.n_node > 2) {
            n_cig = align_to_subgraph<NT>(g, sm, ring, sseq, sc, ch.mode == 0 ? 10 : -1, ch.mode == 0 ? 10 : 0, exc_beg, exc_end, seq, len, &cells);
            if (len > 0) { aligned_bases += len; n_aligned_reads++; }
        }
        // graph update + re-sort: serial pointer work on thread 0; results published through LDS
        const long long tg0 = clock64();
        bool changed = false;
        if (len > 0 && g.status == LCD_OK) changed = add_alignment_block<NT>(g, sm, exc_beg, exc_end, seq, len, n_cig, i);
        if (changed && g.status == LCD_OK) topo_sort_block<NT>(g, sm, lds_pool);
        t_graph += (unsigned long long)(clock64() - tg0);
    }
    const long long t_out0 = clock64();
    // ---------------- output: MSA rank, rows, clusters, consensus (oracle/poa.c poa_output) ----------------
    PoaChainOut out;
    out.status = g.status; out.n_cons = 0; out.cons_len[0] = out.cons_len[1] = 0; out.msa_len = 0; out.clu_n[0] = out.clu_n[1] = 0;
    out.n_node = g.n_node; out.n_edge = g.n_edge; out.n_aligned_reads = n_aligned_reads; out.cells = cells; out.aligned_bases = aligned_bases;
    uint8_t *ob = outpool + ch.out_off;
    const int nc_cap = ch.node_cap;
    uint8_t *cons0 = ob, *cons1 = ob + nc_cap;
    uint8_t *msa = ob + 2 * (size_t)nc_cap; // rows: n_seq reads, then cons rows n_seq, n_seq+1
    int *clu_ids = (int *)(ob + lcd_align_up((uint64_t)(n_seq + 4) * nc_cap, 16)); // [2][n_seq]
    if (g.status == LCD_OK && g.n_node > 2) {
        const int n = g.n_node;
        int *rank = g.deg;
        if (tid == 0) {
            int nc = 0;
            for (int i = 0; i < n; ++i) rank[i] = -1;
            for (int idx = 1; idx < n - 1; ++idx) {
                int v = g.idx2node[idx];
                if (rank[v] >= 0) continue;
                rank[v] = nc;
                for (int a = g.aligned[v]; a != v; a = g.aligned[a]) rank[a] = nc;
                ++nc;
            }
            sm.bc[0] = nc;
        }
        __syncthreads();
        const int ncol = sm.bc[0];
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
    if (tid == 0) {
        const long long t_end = clock64();
        out.t_total = (unsigned long long)(t_end - t_begin); out.t_dp = g.t_dp; out.t_bt = g.t_bt; out.t_graph = t_graph; out.t_sub = t_sub;
        out.t_out = (unsigned long long)(t_end - t_out0);
        outs[cid] = out;
    }
}

void lcd_launch_poa(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool, uint8_t *arena, uint8_t *outpool,
                    PoaChainOut *outs, LcdScoring sc, int n_chains, int threads, int lds_bytes, hipStream_t stream) {
    if (n_chains <= 0) return;
    static bool attr_set = false;
    if (!attr_set) { // allow > 64 KB of dynamic LDS (gfx950 has 160 KB per CU)
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        hipFuncSetAttribute((const void *)lcd_poa_chain_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set = true;
    }
    if (threads <= 64)
        hipLaunchKernelGGL(lcd_poa_chain_kernel<64>, dim3(n_chains), dim3(64), lds_bytes, stream, chains, reads, pool, arena, outpool, outs, sc, n_chains);
    else if (threads <= 256)
        hipLaunchKernelGGL(lcd_poa_chain_kernel<256>, dim3(n_chains), dim3(256), lds_bytes, stream, chains, reads, pool, arena, outpool, outs, sc, n_chains);
    else
        hipLaunchKernelGGL(lcd_poa_chain_kernel<1024>, dim3(n_chains), dim3(1024), lds_bytes, stream, chains, reads, pool, arena, outpool, outs, sc, n_chains);
}
