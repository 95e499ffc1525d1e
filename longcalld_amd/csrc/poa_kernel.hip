// poa_kernel.hip -- K1/K2: partial-order alignment chains on gfx950 (CDNA4, wave64).
//
// Replaces what src/align.c:762-857 (abpoa_partial_aln_msa_cons) and :872-943 (abpoa_aln_msa_cons) ask
// of abPOA.  One 64-lane wavefront owns one chain (= one abpoa_t life: region x haplotype for K1, region
// for K2) and keeps the whole graph build device-resident: align read -> backtrack -> add alignment ->
// re-sort -> next read, then MSA / clustering / consensus.  No host round trip inside a chain.
//
// Integer DP only (no MFMA): a DP row is one coalesced 64-column sweep; the horizontal-gap recurrence is a
// wave-level exclusive prefix max (F[j] = max_k<j Hpre[k] - o - (j-k)e  ==  prefixmax(Hpre[k]+k e) - o - j e);
// only H, E1, E2 are stored (12 B/cell), the insertion run is re-derived from H in the backtrack.
// Semantics are defined by oracle/poa.c (see its header); this file must match it bit for bit.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {

struct Ctx {
    int *H, *E1, *E2;
    int *rbeg, *rend; uint32_t *roff;
    int *mpl, *mpr, *idx2node, *node2idx, *remain, *deg, *queue;
    int *out_head, *out_tail, *in_head, *in_tail, *nin, *aligned;
    int *e_from, *e_to, *e_w, *e_next_out, *e_next_in;
    unsigned long long *rid;
    int *cig_node, *cig_qpos;
    uint8_t *base, *imap;
    int *het, *clu, *nclu; uint8_t *prof;
    int n_node, n_edge, node_cap, edge_cap, rid_words;
    unsigned long long cell_cap;
    int status;
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
// exclusive prefix max over lanes (lane 0 gets `identity`)
__device__ __forceinline__ int wave_excl_prefix_max(int v, int lane, int identity) {
    int x = v;
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d);
        if (lane >= d) x = imax(x, y);
    }
    int e = __shfl_up(x, 1);
    return lane == 0 ? identity : e;
}

// ---------------- graph mutation (lane 0 only) ----------------
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

__device__ void add_alignment(Ctx &g, int beg_node, int end_node, const uint8_t *seq, int len, int n_cig, int read_id) {
    if (g.n_node == 2) {
        int last = 0;
        for (int i = 0; i < len; ++i) { int id = add_node(g, seq[i]); add_edge(g, last, id, 0, read_id); last = id; }
        add_edge(g, last, 1, 0, read_id);
        topo_sort(g);
        return;
    }
    if (n_cig == 0) return;
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
    if (g.status == LCD_OK) topo_sort(g);
}

// sub-graph boundaries (oracle/poa.c subgraph_nodes), wave-parallel min/max sweeps
__device__ void subgraph_nodes(Ctx &g, int lane, int inc_beg, int inc_end, int *exc_beg, int *exc_end) {
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

// banded convex-gap global alignment of seq[0..qlen) to the sub-graph (beg_node,end_node); returns #cigar
// entries written to cig_node/cig_qpos in start->end order (lane-uniform result).
__device__ int align_to_subgraph(Ctx &g, int lane, const LcdScoring &sc, int wb, int wf_milli, int beg_node, int end_node,
                                 const uint8_t *seq, int qlen, unsigned long long *cells_acc) {
    if (qlen <= 0) return 0;
    const int bi = g.node2idx[beg_node], ei = g.node2idx[end_node];
    const int o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2, oe1 = o1 + e1, oe2 = o2 + e2;
    // w = wb<0 ? qlen : wb + (int)(wf*qlen);  wf is 0.01 or 0 on this path: (int)(0.01*q) == q/100 for q < 2^31/100
    const int w = wb < 0 ? qlen : wb + (int)(((long long)wf_milli * qlen) / 1000);
    const int remain_end = g.remain[end_node];
    const int n = g.n_node;
    // reachability map over [bi, ei]
    if (bi == 0 && ei == n - 1) {
        for (int i = lane; i < n; i += 64) g.imap[i] = 1;
    } else {
        for (int i = bi + lane; i <= ei; i += 64) g.imap[i] = 0;
        __syncthreads();
        if (lane == 0) {
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
    for (int i = lane; i < n; i += 64) { g.mpl[i] = 1 << 30; g.mpr[i] = 0; }
    __syncthreads();
    unsigned long long used = 0;
    // ---- source row ----
    {
        int r = g.remain[beg_node] - remain_end;
        int end = qlen - r; if (end < 0) end = 0; end += w; if (end > qlen) end = qlen;
        if ((unsigned long long)end + 1 > g.cell_cap) { g.status = LCD_ERR_CELLS; return 0; }
        if (lane == 0) { g.rbeg[bi] = 0; g.rend[bi] = end; g.roff[bi] = 0; }
        for (int j = lane; j <= end; j += 64) {
            int f1 = j ? -(o1 + e1 * j) : LCD_NEG, f2 = j ? -(o2 + e2 * j) : LCD_NEG;
            int h = j ? imax(f1, f2) : 0;
            g.H[j] = h; g.E1[j] = h - oe1; g.E2[j] = h - oe2;
        }
        used = end + 1;
        if (lane == 0)
            for (int e = g.out_head[beg_node]; e >= 0; e = g.e_next_out[e]) {
                int o = g.e_to[e];
                g.mpl[o] = imin(g.mpl[o], 1); g.mpr[o] = imax(g.mpr[o], 1);
            }
        __syncthreads();
    }
    // ---- rows ----
    for (int idx = bi + 1; idx < ei; ++idx) {
        if (!g.imap[idx]) { if (lane == 0) { g.rbeg[idx] = 1; g.rend[idx] = 0; g.roff[idx] = (uint32_t)used; } __syncthreads(); continue; }
        const int v = g.idx2node[idx];
        const int rem = g.remain[v] - remain_end;
        int beg = imin(g.mpl[v], qlen - rem) - w; if (beg < 0) beg = 0;
        int end = imax(g.mpr[v], qlen - rem) + w; if (end > qlen) end = qlen;
        int minpb = 1 << 30, maxpe = -1;
        for (int e = g.in_head[v]; e >= 0; e = g.e_next_in[e]) {
            int pi = g.node2idx[g.e_from[e]];
            if (pi < bi || pi >= ei || !g.imap[pi]) continue;
            int pb = g.rbeg[pi], pe = g.rend[pi];
            if (pb > pe) continue;
            minpb = imin(minpb, pb); maxpe = imax(maxpe, pe);
        }
        if (beg < minpb) beg = minpb;
        if (end > maxpe + 1) end = maxpe + 1;
        if (beg > end) { if (lane == 0) { g.rbeg[idx] = 1; g.rend[idx] = 0; g.roff[idx] = (uint32_t)used; } __syncthreads(); continue; }
        const unsigned long long off = used;
        used += (unsigned long long)(end - beg + 1);
        if (used > g.cell_cap) { g.status = LCD_ERR_CELLS; return 0; }
        if (lane == 0) { g.rbeg[idx] = beg; g.rend[idx] = end; g.roff[idx] = (uint32_t)off; }
        const uint8_t vb = g.base[v];
        int carry1 = LCD_NEG, carry2 = LCD_NEG; // running max of Hpre[k]+k*e over previous chunks (k*e can be large: use offsets from beg)
        int best_h = LCD_NEG - 64, best_l = 1 << 30, best_r = -1;
        for (int c0 = beg; c0 <= end; c0 += 64) {
            const int j = c0 + lane;
            const bool act = j <= end;
            int mx = LCD_NEG, e1i = LCD_NEG, e2i = LCD_NEG;
            int s = 0;
            if (act && j >= 1) { uint8_t qb = seq[j - 1]; s = (vb >= 4 || qb >= 4) ? 0 : (vb == qb ? sc.match : -sc.mismatch); }
            for (int e = g.in_head[v]; e >= 0; e = g.e_next_in[e]) {
                int pi = g.node2idx[g.e_from[e]];
                if (pi < bi || pi >= ei || !g.imap[pi]) continue;
                const int pb = g.rbeg[pi], pe = g.rend[pi];
                const uint32_t po = g.roff[pi];
                const int bonus = ilog2_32(g.e_w[e]);
                if (act) {
                    if (j >= 1 && j - 1 >= pb && j - 1 <= pe) mx = imax(mx, g.H[po + (j - 1 - pb)] + s + bonus);
                    if (j >= pb && j <= pe) {
                        e1i = imax(e1i, g.E1[po + (j - pb)] + bonus);
                        e2i = imax(e2i, g.E2[po + (j - pb)] + bonus);
                    }
                }
            }
            int hpre = imax(mx, imax(e1i, e2i));
            // F via exclusive prefix max of A[k] = Hpre[k] + (k-beg)*e
            const int rel = j - beg;
            int a1 = act ? hpre + rel * e1 : LCD_NEG * 2, a2 = act ? hpre + rel * e2 : LCD_NEG * 2;
            int p1 = wave_excl_prefix_max(a1, lane, LCD_NEG * 2), p2 = wave_excl_prefix_max(a2, lane, LCD_NEG * 2);
            p1 = imax(p1, carry1); p2 = imax(p2, carry2);
            int f1 = (j > beg) ? imax(LCD_NEG, p1 - o1 - rel * e1) : LCD_NEG;
            int f2 = (j > beg) ? imax(LCD_NEG, p2 - o2 - rel * e2) : LCD_NEG;
            carry1 = imax(carry1, wave_max(a1)); carry2 = imax(carry2, wave_max(a2));
            int h = imax(hpre, imax(f1, f2)); if (h < LCD_NEG) h = LCD_NEG;
            int eo1 = imax(h - oe1, e1i - e1), eo2 = imax(h - oe2, e2i - e2);
            if (eo1 < LCD_NEG) eo1 = LCD_NEG;
            if (eo2 < LCD_NEG) eo2 = LCD_NEG;
            if (act) {
                g.H[off + rel] = h; g.E1[off + rel] = eo1; g.E2[off + rel] = eo2;
                if (h > best_h) { best_h = h; best_l = j; best_r = j; } else if (h == best_h) best_r = j;
            }
        }
        // row maximum, leftmost / rightmost column
        int rowmax = wave_max(best_h);
        int ml = wave_min(best_h == rowmax ? best_l : (1 << 30));
        int mr = wave_max(best_h == rowmax ? best_r : -1);
        if (lane == 0)
            for (int e = g.out_head[v]; e >= 0; e = g.e_next_out[e]) {
                int o = g.e_to[e];
                g.mpl[o] = imin(g.mpl[o], ml + 1); g.mpr[o] = imax(g.mpr[o], mr + 1);
            }
        __syncthreads();
    }
    *cells_acc += used;
    // ---- end node: best predecessor at column qlen, then backtrack (lane 0) ----
    int n_cig = 0;
    if (lane == 0) {
        int best = LCD_NEG, br = -1;
        for (int e = g.in_head[end_node]; e >= 0; e = g.e_next_in[e]) {
            int pi = g.node2idx[g.e_from[e]];
            if (pi < bi || pi >= ei || !g.imap[pi]) continue;
            if (qlen < g.rbeg[pi] || qlen > g.rend[pi]) continue;
            int c = g.H[g.roff[pi] + (qlen - g.rbeg[pi])] + ilog2_32(g.e_w[e]);
            if (c > best) { best = c; br = pi; }
        }
        if (br >= 0 && best > LCD_NEG / 2) {
            int pos = qlen;
            int i = br, j = qlen, st = 0;
#define CELLH(pi, jj) g.H[g.roff[pi] + ((jj) - g.rbeg[pi])]
#define INB(pi, jj) ((jj) >= g.rbeg[pi] && (jj) <= g.rend[pi])
            while (i != bi && j > 0 && g.status == LCD_OK) {
                const int v = g.idx2node[i];
                if (st == 0) {
                    const int hv = CELLH(i, j);
                    bool hit = false;
                    uint8_t vb = g.base[v], qb = seq[j - 1];
                    const int s = (vb >= 4 || qb >= 4) ? 0 : (vb == qb ? sc.match : -sc.mismatch);
                    for (int e = g.in_head[v]; e >= 0 && !hit; e = g.e_next_in[e]) {
                        int pi = g.node2idx[g.e_from[e]];
                        if (pi < bi || pi >= ei || !g.imap[pi] || !INB(pi, j - 1)) continue;
                        if (CELLH(pi, j - 1) + s + ilog2_32(g.e_w[e]) == hv) {
                            --pos; g.cig_node[pos] = v; g.cig_qpos[pos] = j - 1; i = pi; --j; hit = true;
                        }
                    }
                    for (int c = 1; c <= 2 && !hit; ++c) {
                        const int *E = c == 1 ? g.E1 : g.E2;
                        for (int e = g.in_head[v]; e >= 0 && !hit; e = g.e_next_in[e]) {
                            int pi = g.node2idx[g.e_from[e]];
                            if (pi < bi || pi >= ei || !g.imap[pi] || !INB(pi, j)) continue;
                            if (E[g.roff[pi] + (j - g.rbeg[pi])] + ilog2_32(g.e_w[e]) == hv) { i = pi; st = c; hit = true; }
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
                    for (int e = g.in_head[v]; e >= 0 && !hit; e = g.e_next_in[e]) {
                        int pi = g.node2idx[g.e_from[e]];
                        if (pi < bi || pi >= ei || !g.imap[pi] || !INB(pi, j)) continue;
                        if (E[g.roff[pi] + (j - g.rbeg[pi])] + ilog2_32(g.e_w[e]) - ee == ev) { i = pi; hit = true; }
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
    }
    n_cig = __shfl(n_cig, 0);
    __syncthreads();
    return n_cig;
}

} // namespace

// one wavefront per chain
__global__ void __launch_bounds__(64) lcd_poa_chain_kernel(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool,
                                                           uint8_t *arena, uint8_t *outpool, PoaChainOut *outs, LcdScoring sc,
                                                           int n_chains) {
    const int cid = blockIdx.x;
    if (cid >= n_chains) return;
    const int lane = threadIdx.x;
    const PoaChain ch = chains[cid];
    const PoaLayout L = poa_layout(ch.node_cap, ch.edge_cap, ch.rid_words, ch.max_len, ch.cell_cap, ch.n_reads);
    uint8_t *ws = arena + ch.ws_off;
    Ctx g;
    g.H = (int *)(ws + L.H); g.E1 = (int *)(ws + L.E1); g.E2 = (int *)(ws + L.E2);
    g.rbeg = (int *)(ws + L.rbeg); g.rend = (int *)(ws + L.rend); g.roff = (uint32_t *)(ws + L.roff);
    g.mpl = (int *)(ws + L.mpl); g.mpr = (int *)(ws + L.mpr);
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
    g.node_cap = ch.node_cap; g.edge_cap = ch.edge_cap; g.rid_words = ch.rid_words; g.cell_cap = ch.cell_cap;
    g.n_node = 2; g.n_edge = 0; g.status = LCD_OK;
    if (lane == 0)
        for (int i = 0; i < 2; ++i) {
            g.base[i] = 4; g.out_head[i] = g.out_tail[i] = g.in_head[i] = g.in_tail[i] = -1; g.nin[i] = 0; g.aligned[i] = i;
        }
    __syncthreads();
    unsigned long long cells = 0, aligned_bases = 0;
    int n_aligned_reads = 0;
    const int n_seq = ch.n_reads;
    const PoaRead *rd = reads + ch.read0;
    const int backbone_len = rd[0].len;
    (void)backbone_len;
    for (int i = 0; i < n_seq && g.status == LCD_OK; ++i) {
        const PoaRead r = rd[i];
        if (r.skip) continue;
        int exc_beg = 0, exc_end = 1, beg_cut = 0, end_cut = 0;
        if (ch.mode == 0 && i != 0) {
            beg_cut = r.read_beg - 1; end_cut = r.len - r.read_end;
            subgraph_nodes(g, lane, r.ref_beg + 1, r.ref_end + 1, &exc_beg, &exc_end);
        }
        const uint8_t *seq = pool + r.seq_off + beg_cut;
        const int len = r.len - beg_cut - end_cut;
        int n_cig = 0;
        if (g.n_node > 2) {
            n_cig = align_to_subgraph(g, lane, sc, ch.mode == 0 ? 10 : -1, ch.mode == 0 ? 10 : 0, exc_beg, exc_end, seq, len, &cells);
            if (len > 0) { aligned_bases += len; n_aligned_reads++; }
        }
        // graph update + re-sort: serial pointer work, lane 0; results published through memory
        int nn = 0, ne = 0, st = 0;
        if (lane == 0) {
            if (len > 0 && g.status == LCD_OK) add_alignment(g, exc_beg, exc_end, seq, len, n_cig, i);
            nn = g.n_node; ne = g.n_edge; st = g.status;
        }
        g.n_node = __shfl(nn, 0); g.n_edge = __shfl(ne, 0); g.status = __shfl(st, 0);
        __syncthreads();
    }
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
        int ncol = 0;
        if (lane == 0) {
            for (int i = 0; i < n; ++i) rank[i] = -1;
            for (int idx = 1; idx < n - 1; ++idx) {
                int v = g.idx2node[idx];
                if (rank[v] >= 0) continue;
                rank[v] = ncol;
                for (int a = g.aligned[v]; a != v; a = g.aligned[a]) rank[a] = ncol;
                ++ncol;
            }
        }
        ncol = __shfl(ncol, 0);
        __syncthreads();
        for (size_t t = lane; t < (size_t)(n_seq + 2) * ncol; t += 64) msa[(t / ncol) * (size_t)nc_cap + (t % ncol)] = LCD_GAP;
        __syncthreads();
        for (int v = 2 + lane; v < n; v += 64) {
            const int col = rank[v]; const uint8_t b = g.base[v];
            for (int e = g.out_head[v]; e >= 0; e = g.e_next_out[e])
                for (int wd = 0; wd < g.rid_words; ++wd) {
                    unsigned long long bits = g.rid[(size_t)e * g.rid_words + wd];
                    while (bits) { int r = __ffsll((long long)bits) - 1; bits &= bits - 1; msa[(size_t)(wd * 64 + r) * nc_cap + col] = b; }
                }
        }
        __syncthreads();
        // clustering
        int n_clu = 1;
        for (int r = lane; r < n_seq; r += 64) g.clu[r] = 0;
        __syncthreads();
        if (ch.mode == 1 && n_seq >= 2) {
            const int min_w = (int)ch.min_w;
            // het columns (ordered compaction, 64 columns per step)
            int n_het = 0;
            for (int c0 = 0; c0 < ncol; c0 += 64) {
                int c = c0 + lane, ishet = 0;
                if (c < ncol) {
                    int cnt[6] = {0, 0, 0, 0, 0, 0};
                    for (int r = 0; r < n_seq; ++r) cnt[msa[(size_t)r * nc_cap + c]]++;
                    int k = 0;
                    for (int a = 0; a < 6; ++a) k += cnt[a] >= min_w;
                    ishet = k >= 2;
                }
                unsigned long long m = __ballot(ishet);
                if (ishet) g.het[n_het + __popcll(m & ((1ull << lane) - 1))] = c;
                n_het += __popcll(m);
            }
            __syncthreads();
            if (n_het > 0) {
                // pivot: most balanced het column (largest runner-up count, leftmost)
                int bv2 = -1, bh = 1 << 30, ba0 = 0, ba1 = 0;
                for (int h = lane; h < n_het; h += 64) {
                    int cnt[6] = {0, 0, 0, 0, 0, 0};
                    for (int r = 0; r < n_seq; ++r) cnt[msa[(size_t)r * nc_cap + g.het[h]]]++;
                    int m0 = 0; for (int a = 1; a < 6; ++a) if (cnt[a] > cnt[m0]) m0 = a;
                    int m1 = -1; for (int a = 0; a < 6; ++a) if (a != m0 && (m1 < 0 || cnt[a] > cnt[m1])) m1 = a;
                    if (cnt[m1] > bv2) { bv2 = cnt[m1]; bh = h; ba0 = m0; ba1 = m1; }
                }
                int gv2 = wave_max(bv2);
                int gh = wave_min(bv2 == gv2 ? bh : (1 << 30));
                int src = __ffsll((long long)__ballot(bv2 == gv2 && bh == gh)) - 1;
                const int a0 = __shfl(ba0, src), a1 = __shfl(ba1, src);
                const int pcol = g.het[gh];
                for (int r = lane; r < n_seq; r += 64) { int al = msa[(size_t)r * nc_cap + pcol]; g.clu[r] = al == a0 ? 0 : al == a1 ? 1 : -1; }
                __syncthreads();
                for (int it = 0; it < 10; ++it) {
                    for (int t = lane; t < 2 * n_het; t += 64) {
                        int c = t / n_het, h = t % n_het;
                        int cnt[6] = {0, 0, 0, 0, 0, 0};
                        for (int r = 0; r < n_seq; ++r) if (g.clu[r] == c) cnt[msa[(size_t)r * nc_cap + g.het[h]]]++;
                        int m0 = 0; for (int a = 1; a < 6; ++a) if (cnt[a] > cnt[m0]) m0 = a;
                        g.prof[t] = (uint8_t)m0;
                    }
                    __syncthreads();
                    int changed = 0;
                    for (int r = lane; r < n_seq; r += 64) {
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
                    __syncthreads();
                    for (int r = lane; r < n_seq; r += 64) g.clu[r] = g.nclu[r];
                    __syncthreads();
                    if (!__any(changed)) break;
                }
                int c1 = 0;
                for (int r = lane; r < n_seq; r += 64) c1 += g.clu[r];
                for (int d = 32; d >= 1; d >>= 1) c1 += __shfl_xor(c1, d);
                const int c0n = n_seq - c1;
                if (c0n >= min_w && c1 >= min_w) {
                    n_clu = 2;
                    if (c1 > c0n) for (int r = lane; r < n_seq; r += 64) g.clu[r] ^= 1;
                } else
                    for (int r = lane; r < n_seq; r += 64) g.clu[r] = 0;
                __syncthreads();
            }
        }
        out.n_cons = n_clu; out.msa_len = ncol;
        for (int c = 0; c < n_clu; ++c) {
            // member list, ascending read index (ordered compaction)
            int csize = 0;
            for (int r0 = 0; r0 < n_seq; r0 += 64) {
                int r = r0 + lane, in = r < n_seq && g.clu[r] == c;
                unsigned long long m = __ballot(in);
                if (in) clu_ids[c * n_seq + csize + __popcll(m & ((1ull << lane) - 1))] = r;
                csize += __popcll(m);
            }
            __syncthreads();
            out.clu_n[c] = csize;
            uint8_t *crow = msa + (size_t)(n_seq + c) * nc_cap;
            uint8_t *cons = c == 0 ? cons0 : cons1;
            int cl = 0;
            for (int c0 = 0; c0 < ncol; c0 += 64) {
                int col = c0 + lane, emit = 0, mb = 0;
                if (col < ncol) {
                    int cnt[6] = {0, 0, 0, 0, 0, 0};
                    for (int k = 0; k < csize; ++k) cnt[msa[(size_t)clu_ids[c * n_seq + k] * nc_cap + col]]++;
                    for (int a = 1; a < 5; ++a) if (cnt[a] > cnt[mb]) mb = a;
                    emit = cnt[mb] > 0 && cnt[mb] >= cnt[5];
                }
                unsigned long long m = __ballot(emit);
                if (emit) { cons[cl + __popcll(m & ((1ull << lane) - 1))] = (uint8_t)mb; crow[col] = (uint8_t)mb; }
                cl += __popcll(m);
            }
            out.cons_len[c] = cl;
            __syncthreads();
        }
    }
    if (lane == 0) outs[cid] = out;
}

extern "C++" void lcd_launch_poa(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool, uint8_t *arena, uint8_t *outpool,
                                 PoaChainOut *outs, LcdScoring sc, int n_chains, hipStream_t stream) {
    if (n_chains <= 0) return;
    hipLaunchKernelGGL(lcd_poa_chain_kernel, dim3(n_chains), dim3(64), 0, stream, chains, reads, pool, arena, outpool, outs, sc, n_chains);
}
