/*
 * poa.c -- ORACLE (test infrastructure only; see lcd_oracle.h).  PARITY UNPINNED (byte level).
 *
 * Restates what src/align.c asks of abPOA (github.com/yangao07/abPOA, git submodule `abPOA`, pin
 * unknown, ABSENT from /root/reference):
 *   K1  abpoa_partial_aln_msa_cons   src/align.c:762-857  (sub-graph incremental POA, wb=10 wf=0.01,
 *       sub_aln=1, inc_path_score=1, cons_algrm=ABPOA_MF, max_n_cons=1)
 *   K2  abpoa_aln_msa_cons           src/align.c:872-943  (abpoa_msa, wb=-1 i.e. unbanded, max_n_cons=2)
 * following the published algorithm (Gao et al., "abPOA: an SIMD-based C library for fast partial
 * order alignment using adaptive band", Bioinformatics 2021) and the library behaviour recalled in
 * SURVEY.md Appendix D.  Because upstream cannot be read here, every tie-break below is THIS
 * PROJECT'S definition (documented in DESIGN.md "POA semantics"); the HIP kernel must match this file
 * bit for bit, and this file is pinned only at score level (independent DAG DP, dag_dp_score) and by
 * MSA/consensus invariants (tests/test_oracle_poa.py).
 *
 * Semantics in one place:
 *   graph     node 0 = source, 1 = sink; node k+1 = k-th base (1-based) of the first read (src/align.c:797).
 *             edges keep insertion order; each edge carries a weight and the set of read ids using it.
 *   order     Kahn BFS from the source; a node is released only when it and all nodes aligned to it have
 *             in-degree 0, then it and its aligned ring are queued together.
 *   remain    remain[sink] = -1; remain[v] = remain[heaviest out-neighbour (first on ties)] + 1.
 *   band      w = wb<0 ? qlen : wb + (int)(wf*qlen);  r = remain[v]-remain[end];
 *             beg = max(0, min(max_pos_left[v], qlen-r) - w), end = min(qlen, max(max_pos_right[v], qlen-r) + w),
 *             then clipped to [min pred beg, max pred end + 1].
 *   scores    match +M, mismatch -X, any N: 0; every traversed edge adds ilog2(weight) (inc_path_score).
 *             convex gap: open o1/ext e1 and open o2/ext e2, F from the pre-F row maximum (scan form).
 *   backtrack state machine on the stored H/E1/E2 (F is never stored): in H  M (preds in edge order) > E1 > E2 >
 *             insertion run (closest k with H[k]-gap(j-k)==H[j]); in E  "open" (H-oe==E) tested before "extend".
 *   consensus ABPOA_MF: per MSA column the most frequent symbol among the cluster's reads; bases beat the
 *             gap on ties, lower code beats higher; gap-majority columns are dropped.
 *   clusters  (max_n_cons=2) columns with >=2 symbols of count >= max(2,(int)(n*min_freq)) are het; the most
 *             balanced het column seeds two groups, then <=10 Jacobi rounds of 2-medians on the Hamming
 *             distance over het columns; a cluster smaller than that threshold collapses everything to 1.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

#define NEG (-(1 << 29))

typedef struct {
    int from, to, w, next_out, next_in;
} edge_t;
typedef struct {
    uint8_t base;
    int out_head, out_tail, in_head, in_tail, n_in, n_out;
    int aligned_next; /* ring of mutually aligned nodes (self if alone) */
} node_t;

struct lcdo_poa_s {
    node_t *node; int n_node, m_node;
    edge_t *edge; int n_edge, m_edge;
    uint64_t *rid; int rid_words;
    int n_seq;
    int *idx2node, *node2idx, *remain; int m_idx;
    int last_score;
};

static int g_dbg_banded, g_dbg_unbanded;

static inline int ilog2_32(uint32_t v) { int r = 0; while (v >>= 1) ++r; return r; }

static lcdo_poa_t *poa_init(int n_seq) {
    lcdo_poa_t *g = (lcdo_poa_t *)calloc(1, sizeof(*g));
    g->n_seq = n_seq; g->rid_words = (n_seq + 63) / 64; if (g->rid_words < 1) g->rid_words = 1;
    g->m_node = 1024; g->node = (node_t *)malloc(g->m_node * sizeof(node_t));
    g->m_edge = 2048; g->edge = (edge_t *)malloc(g->m_edge * sizeof(edge_t));
    g->rid = (uint64_t *)calloc((size_t)g->m_edge * g->rid_words, 8);
    for (int i = 0; i < 2; ++i) {
        node_t *n = &g->node[i];
        n->base = 4; n->out_head = n->out_tail = n->in_head = n->in_tail = -1; n->n_in = n->n_out = 0; n->aligned_next = i;
    }
    g->n_node = 2;
    return g;
}
static void poa_free(lcdo_poa_t *g) {
    free(g->node); free(g->edge); free(g->rid); free(g->idx2node); free(g->node2idx); free(g->remain); free(g);
}
static int add_node(lcdo_poa_t *g, uint8_t base) {
    if (g->n_node == g->m_node) { g->m_node *= 2; g->node = (node_t *)realloc(g->node, g->m_node * sizeof(node_t)); }
    node_t *n = &g->node[g->n_node];
    n->base = base; n->out_head = n->out_tail = n->in_head = n->in_tail = -1; n->n_in = n->n_out = 0;
    n->aligned_next = g->n_node;
    return g->n_node++;
}
static void add_edge(lcdo_poa_t *g, int from, int to, int check, int read_id) {
    if (check) {
        for (int e = g->node[from].out_head; e >= 0; e = g->edge[e].next_out)
            if (g->edge[e].to == to) {
                g->edge[e].w += 1;
                g->rid[(size_t)e * g->rid_words + (read_id >> 6)] |= 1ull << (read_id & 63);
                return;
            }
    }
    if (g->n_edge == g->m_edge) {
        g->m_edge *= 2;
        g->edge = (edge_t *)realloc(g->edge, g->m_edge * sizeof(edge_t));
        g->rid = (uint64_t *)realloc(g->rid, (size_t)g->m_edge * g->rid_words * 8);
        memset(g->rid + (size_t)g->n_edge * g->rid_words, 0, (size_t)(g->m_edge - g->n_edge) * g->rid_words * 8);
    }
    int e = g->n_edge++;
    edge_t *E = &g->edge[e];
    E->from = from; E->to = to; E->w = 1; E->next_out = E->next_in = -1;
    memset(g->rid + (size_t)e * g->rid_words, 0, g->rid_words * 8);
    g->rid[(size_t)e * g->rid_words + (read_id >> 6)] |= 1ull << (read_id & 63);
    node_t *f = &g->node[from], *t = &g->node[to];
    if (f->out_tail < 0) f->out_head = e; else g->edge[f->out_tail].next_out = e;
    f->out_tail = e; f->n_out++;
    if (t->in_tail < 0) t->in_head = e; else g->edge[t->in_tail].next_in = e;
    t->in_tail = e; t->n_in++;
}
static int get_aligned_id(lcdo_poa_t *g, int node_id, uint8_t base) {
    for (int a = g->node[node_id].aligned_next; a != node_id; a = g->node[a].aligned_next)
        if (g->node[a].base == base) return a;
    return -1;
}
static void add_aligned(lcdo_poa_t *g, int node_id, int new_id) {
    g->node[new_id].aligned_next = g->node[node_id].aligned_next;
    g->node[node_id].aligned_next = new_id;
}

/* Kahn BFS order + remain */
static void topo_sort(lcdo_poa_t *g) {
    int n = g->n_node;
    if (g->m_idx < n) {
        g->m_idx = n * 2;
        g->idx2node = (int *)realloc(g->idx2node, g->m_idx * sizeof(int));
        g->node2idx = (int *)realloc(g->node2idx, g->m_idx * sizeof(int));
        g->remain = (int *)realloc(g->remain, g->m_idx * sizeof(int));
    }
    int *deg = (int *)malloc(n * sizeof(int)), *q = (int *)malloc(n * sizeof(int));
    for (int i = 0; i < n; ++i) deg[i] = g->node[i].n_in;
    int qh = 0, qt = 0, index = 0;
    q[qt++] = 0;
    while (qh < qt) {
        int cur = q[qh++];
        g->idx2node[index] = cur; g->node2idx[cur] = index++;
        if (cur == 1) break;
        for (int e = g->node[cur].out_head; e >= 0; e = g->edge[e].next_out) {
            int out = g->edge[e].to;
            if (--deg[out] == 0) {
                int ok = 1;
                for (int a = g->node[out].aligned_next; a != out; a = g->node[a].aligned_next)
                    if (deg[a] != 0) { ok = 0; break; }
                if (!ok) continue;
                q[qt++] = out;
                for (int a = g->node[out].aligned_next; a != out; a = g->node[a].aligned_next) q[qt++] = a;
            }
        }
    }
    assert(index == n);
    /* remain: reverse order suffices (all successors have larger index) */
    g->remain[1] = -1;
    for (int i = n - 2; i >= 0; --i) {
        int v = g->idx2node[i], mw = -1, mid = -1;
        for (int e = g->node[v].out_head; e >= 0; e = g->edge[e].next_out)
            if (g->edge[e].w > mw) { mw = g->edge[e].w; mid = g->edge[e].to; }
        g->remain[v] = g->remain[mid] + 1;
    }
    free(deg); free(q);
}

/* sub-graph boundaries: smallest exclusive window such that every in-edge of (up,beg] starts at >= up
 * and every out-edge of [end,down) ends at <= down.  (abpoa_subgraph_nodes as recalled) */
static void subgraph_nodes(lcdo_poa_t *g, int inc_beg, int inc_end, int *exc_beg, int *exc_end) {
    int bi = g->node2idx[inc_beg], ei = g->node2idx[inc_end];
    /* upstream */
    int b = bi, e = ei, up;
    for (;;) {
        int mn = b;
        for (int i = b; i <= e; ++i)
            for (int ed = g->node[g->idx2node[i]].in_head; ed >= 0; ed = g->edge[ed].next_in) {
                int x = g->node2idx[g->edge[ed].from];
                if (x < mn) mn = x;
            }
        int full = 1;
        for (int i = mn + 1; i <= b && full; ++i)
            for (int ed = g->node[g->idx2node[i]].in_head; ed >= 0; ed = g->edge[ed].next_in)
                if (g->node2idx[g->edge[ed].from] < mn) { full = 0; break; }
        if (full) { up = mn; break; }
        e = b; b = mn;
    }
    /* downstream */
    b = bi; e = ei; int down;
    for (;;) {
        int mx = e;
        for (int i = b; i <= e; ++i)
            for (int ed = g->node[g->idx2node[i]].out_head; ed >= 0; ed = g->edge[ed].next_out) {
                int x = g->node2idx[g->edge[ed].to];
                if (x > mx) mx = x;
            }
        int full = 1;
        for (int i = e; i < mx && full; ++i)
            for (int ed = g->node[g->idx2node[i]].out_head; ed >= 0; ed = g->edge[ed].next_out)
                if (g->node2idx[g->edge[ed].to] > mx) { full = 0; break; }
        if (full) { down = mx; break; }
        b = e; e = mx;
    }
    *exc_beg = g->idx2node[up]; *exc_end = g->idx2node[down];
}

typedef struct {
    int op;   /* 0 = M(node,qpos), 1 = I(qpos) */
    int node, qpos;
} gcig_t;

static inline int sc_mat(const lcdo_opt_t *o, uint8_t a, uint8_t b) {
    if (a >= 4 || b >= 4) return 0;
    return a == b ? o->match : -o->mismatch;
}

/* banded (wb,wf) convex-gap global alignment of seq[0..qlen) to the sub-graph (beg_node, end_node) exclusive.
 * returns number of cigar entries (start->end order) in *cig (malloc'd). */

/* ---- checker of the product's CERTIFIED BAND for K2 (poa_kernel.hip align_certified; DESIGN.md "Certified band"), switched on by LCDO_CERT_STATS:
 * upper bounds on the score of any alignment through a cell, from per-node path-length ranges (minD..maxD nodes from the source, minR..maxR to the sink) and
 * maximal edge-bonus sums; a cell whose bound is below a known alignment score cannot lie on an optimal path.  Inside the unbanded DP below the checker
 *   - tests the prefix half of the bound against EVERY cell's H (g_cert_viol_prefix),
 *   - builds each row's interval for the score of a banded alignment of the same read and tests that every matched cell of the backtrack is inside
 *     (g_cert_viol_path),
 *   - replays the product's guessing policy (bound at the end cell minus the largest slack seen so far in the chain + a margin) and counts cells, retries and
 *     rows wider than the 256-column window.
 * lcdo_poa_cert_stats returns the counters (tests/test_oracle_poa.py::test_band_certificate). ---- */
static int g_cert_on = -1, g_cert_slb = NEG;
static long long g_cert_cells_full, g_cert_cells_hull, g_cert_rows, g_cert_reads, g_cert_viol_prefix, g_cert_viol_path, g_cert_maxw, g_cert_wide;
static int g_cert_hist_delta = -1; static long long g_cert_est_cells, g_cert_est_retry, g_cert_est_ovf_reads, g_cert_true_ovf_reads, g_cert_regions_allok, g_cert_regions, g_cert_region_bad;
static inline int cert_G(const lcdo_opt_t *o, int k) { if (k <= 0) return 0; int a = o->gap_open1 + o->gap_ext1 * k, b = o->gap_open2 + o->gap_ext2 * k; return a < b ? a : b; }
static int align_to_subgraph(lcdo_poa_t *g, const lcdo_opt_t *opt, int wb, double wf, int beg_node, int end_node,
                             const uint8_t *seq, int qlen, gcig_t **cig_out, int *score_out) {
    *cig_out = NULL; if (score_out) *score_out = NEG;
    if (qlen <= 0) return 0;
    const int bi = g->node2idx[beg_node], ei = g->node2idx[end_node];
    const int nrow = ei - bi; /* rows bi..ei-1 */
    const int o1 = opt->gap_open1, e1 = opt->gap_ext1, o2 = opt->gap_open2, e2 = opt->gap_ext2;
    const int oe1 = o1 + e1, oe2 = o2 + e2;
    const int w = wb < 0 ? qlen : wb + (int)(wf * qlen);
    const int remain_end = g->remain[end_node];
    uint8_t *imap = (uint8_t *)calloc(nrow + 1, 1);
    int *rbeg = (int *)malloc(nrow * sizeof(int)), *rend = (int *)malloc(nrow * sizeof(int));
    size_t *roff = (size_t *)malloc(nrow * sizeof(size_t));
    int *mpl = (int *)malloc(g->n_node * sizeof(int)), *mpr = (int *)malloc(g->n_node * sizeof(int));
    for (int i = 0; i < g->n_node; ++i) { mpl[i] = 1 << 30; mpr[i] = 0; }
    imap[0] = 1; imap[nrow] = 1;
    for (int r = 0; r < nrow; ++r) {
        if (!imap[r]) continue;
        int v = g->idx2node[bi + r];
        for (int e = g->node[v].out_head; e >= 0; e = g->edge[e].next_out) {
            int x = g->node2idx[g->edge[e].to] - bi;
            if (x >= 0 && x <= nrow) imap[x] = 1;
        }
    }
    size_t cap = 1 << 16, used = 0;
    int *H = (int *)malloc(cap * 3 * sizeof(int));
#define GROW(need) do { if (used + (need) > cap) { while (used + (need) > cap) cap *= 2; H = (int *)realloc(H, cap * 3 * sizeof(int)); } } while (0)
    /* cell (r,j) lives at H[3*(roff[r]+j-rbeg[r]) + {0:H,1:E1,2:E2}]; F1/F2 are row-local (never stored):
     * the backtrack re-derives an insertion run from H alone (closest k with H[k]-gap(j-k)==H[j]). */
#define CELL(r, j, c) H[3 * (roff[r] + (size_t)((j) - rbeg[r])) + (c)]
#define INBAND(r, j) ((j) >= rbeg[r] && (j) <= rend[r])
    /* row 0 : source */
    {
        int r = g->remain[beg_node] - remain_end;
        int end = qlen - r; if (end < 0) end = 0; end += w; if (end > qlen) end = qlen;
        rbeg[0] = 0; rend[0] = end; roff[0] = used; GROW((size_t)end + 1); used += end + 1;
        for (int j = 0; j <= end; ++j) {
            int f1 = j ? -(o1 + e1 * j) : NEG, f2 = j ? -(o2 + e2 * j) : NEG;
            int h = j ? (f1 > f2 ? f1 : f2) : 0;
            CELL(0, j, 0) = h; CELL(0, j, 1) = h - oe1; CELL(0, j, 2) = h - oe2;
        }
        for (int e = g->node[beg_node].out_head; e >= 0; e = g->edge[e].next_out) {
            int o = g->edge[e].to;
            if (1 < mpl[o]) mpl[o] = 1;
            if (1 > mpr[o]) mpr[o] = 1;
        }
    }
    for (int r = 1; r < nrow; ++r) {
        rbeg[r] = 1; rend[r] = 0; roff[r] = used;
        if (!imap[r]) continue;
        int v = g->idx2node[bi + r];
        int rem = g->remain[v] - remain_end;
        int beg = mpl[v] < qlen - rem ? mpl[v] : qlen - rem; beg -= w; if (beg < 0) beg = 0;
        int end = mpr[v] > qlen - rem ? mpr[v] : qlen - rem; end += w; if (end > qlen) end = qlen;
        int minpb = 1 << 30, maxpe = -1;
        for (int e = g->node[v].in_head; e >= 0; e = g->edge[e].next_in) {
            int pr = g->node2idx[g->edge[e].from] - bi;
            if (pr < 0 || pr >= nrow || !imap[pr] || rbeg[pr] > rend[pr]) continue;
            if (rbeg[pr] < minpb) minpb = rbeg[pr];
            if (rend[pr] > maxpe) maxpe = rend[pr];
        }
        if (beg < minpb) beg = minpb;
        if (end > maxpe + 1) end = maxpe + 1;
        if (beg > end) continue; /* empty row */
        rbeg[r] = beg; rend[r] = end; GROW((size_t)(end - beg + 1)); used += end - beg + 1;
        int f1 = NEG, f2 = NEG, hpre_prev = NEG, rowmax = NEG, ml = beg, mr = beg;
        for (int j = beg; j <= end; ++j) {
            int mx = NEG, e1i = NEG, e2i = NEG;
            for (int e = g->node[v].in_head; e >= 0; e = g->edge[e].next_in) {
                int pr = g->node2idx[g->edge[e].from] - bi;
                if (pr < 0 || pr >= nrow || !imap[pr]) continue;
                int bonus = ilog2_32(g->edge[e].w);
                if (j >= 1 && INBAND(pr, j - 1)) {
                    int c = CELL(pr, j - 1, 0) + sc_mat(opt, g->node[v].base, seq[j - 1]) + bonus;
                    if (c > mx) mx = c;
                }
                if (INBAND(pr, j)) {
                    int c = CELL(pr, j, 1) + bonus; if (c > e1i) e1i = c;
                    c = CELL(pr, j, 2) + bonus; if (c > e2i) e2i = c;
                }
            }
            int hpre = mx; if (e1i > hpre) hpre = e1i; if (e2i > hpre) hpre = e2i;
            if (j > beg) {
                int a = hpre_prev - oe1, b = f1 - e1; f1 = a > b ? a : b;
                a = hpre_prev - oe2; b = f2 - e2; f2 = a > b ? a : b;
            } else f1 = f2 = NEG;
            if (f1 < NEG) f1 = NEG;
            if (f2 < NEG) f2 = NEG;
            int h = hpre; if (f1 > h) h = f1; if (f2 > h) h = f2;
            if (h < NEG) h = NEG;
            int eo1 = h - oe1 > e1i - e1 ? h - oe1 : e1i - e1, eo2 = h - oe2 > e2i - e2 ? h - oe2 : e2i - e2;
            if (eo1 < NEG) eo1 = NEG;
            if (eo2 < NEG) eo2 = NEG;
            CELL(r, j, 0) = h; CELL(r, j, 1) = eo1; CELL(r, j, 2) = eo2;
            hpre_prev = hpre;
            if (h > rowmax) { rowmax = h; ml = mr = j; } else if (h == rowmax) mr = j;
        }
        for (int e = g->node[v].out_head; e >= 0; e = g->edge[e].next_out) {
            int o = g->edge[e].to;
            if (ml + 1 < mpl[o]) mpl[o] = ml + 1;
            if (mr + 1 > mpr[o]) mpr[o] = mr + 1;
        }
    }
    /* best predecessor of the end node at column qlen */
    int best = NEG, br = -1;
    for (int e = g->node[end_node].in_head; e >= 0; e = g->edge[e].next_in) {
        int pr = g->node2idx[g->edge[e].from] - bi;
        if (pr < 0 || pr >= nrow || !imap[pr] || !INBAND(pr, qlen)) continue;
        int c = CELL(pr, qlen, 0) + ilog2_32(g->edge[e].w);
        if (c > best) { best = c; br = pr; }
    }
    int n = 0;
    gcig_t *cig = NULL;
    if (br >= 0 && best > NEG / 2) {
        if (score_out) *score_out = best;
        cig = (gcig_t *)malloc((size_t)(qlen + 1) * sizeof(gcig_t));
        /* reverse fill from the back */
        int pos = qlen; /* next free slot is pos-1 ; at most qlen entries (one per query base) */
        int i = br, j = qlen, st = 0; /* st: 0 H, 1 E1out, 2 E2out */
        while (i != 0 && j > 0) {
            int v = g->idx2node[bi + i];
            if (st == 0) {
                int hv = CELL(i, j, 0), hit = 0;
                int s = sc_mat(opt, g->node[v].base, seq[j - 1]);
                for (int e = g->node[v].in_head; e >= 0 && !hit; e = g->edge[e].next_in) {
                    int pr = g->node2idx[g->edge[e].from] - bi;
                    if (pr < 0 || pr >= nrow || !imap[pr] || !INBAND(pr, j - 1)) continue;
                    if (CELL(pr, j - 1, 0) + s + ilog2_32(g->edge[e].w) == hv) {
                        cig[--pos] = (gcig_t){0, v, j - 1}; i = pr; --j; hit = 1;
                    }
                }
                for (int c = 1; c <= 2 && !hit; ++c)
                    for (int e = g->node[v].in_head; e >= 0 && !hit; e = g->edge[e].next_in) {
                        int pr = g->node2idx[g->edge[e].from] - bi;
                        if (pr < 0 || pr >= nrow || !imap[pr] || !INBAND(pr, j)) continue;
                        if (CELL(pr, j, c) + ilog2_32(g->edge[e].w) == hv) { i = pr; st = c; hit = 1; }
                    }
                if (!hit) { /* insertion run: closest k with H[i][k] - gap(j-k) == H[i][j] */
                    for (int k = j - 1; k >= rbeg[i] && !hit; --k) {
                        int len = j - k, hk = CELL(i, k, 0);
                        if (hk - o1 - len * e1 == hv || hk - o2 - len * e2 == hv) {
                            for (int t = j; t > k; --t) cig[--pos] = (gcig_t){1, -1, t - 1};
                            j = k; hit = 1;
                        }
                    }
                }
                if (!hit) { fprintf(stderr, "[lcdo_poa] backtrack failed at H(%d,%d)\n", i, j); abort(); }
            } else {
                const int oe = st == 1 ? oe1 : oe2, ee = st == 1 ? e1 : e2;
                int ev = CELL(i, j, st);
                if (CELL(i, j, 0) - oe == ev) { st = 0; continue; }
                int hit = 0;
                for (int e = g->node[v].in_head; e >= 0 && !hit; e = g->edge[e].next_in) {
                    int pr = g->node2idx[g->edge[e].from] - bi;
                    if (pr < 0 || pr >= nrow || !imap[pr] || !INBAND(pr, j)) continue;
                    if (CELL(pr, j, st) + ilog2_32(g->edge[e].w) - ee == ev) { i = pr; hit = 1; }
                }
                if (!hit) { fprintf(stderr, "[lcdo_poa] backtrack failed at E%d(%d,%d)\n", st, i, j); abort(); }
            }
        }
        while (j > 0) { cig[--pos] = (gcig_t){1, -1, j - 1}; --j; }
        n = qlen - pos;
        memmove(cig, cig + pos, (size_t)n * sizeof(gcig_t));
    }

    if (g_cert_on > 0 && wb < 0 && br >= 0 && best > NEG / 2 && g_cert_slb > NEG / 2) {
        const int M = opt->match, O = opt->gap_open1 > opt->gap_open2 ? opt->gap_open1 : opt->gap_open2;
        int *minD = (int *)malloc(sizeof(int) * (nrow + 1)), *maxD = (int *)malloc(sizeof(int) * (nrow + 1)), *Bp = (int *)malloc(sizeof(int) * (nrow + 1));
        int *minR = (int *)malloc(sizeof(int) * (nrow + 1)), *maxR = (int *)malloc(sizeof(int) * (nrow + 1)), *Bs = (int *)malloc(sizeof(int) * (nrow + 1));
        int *lo = (int *)malloc(sizeof(int) * (nrow + 1)), *hi = (int *)malloc(sizeof(int) * (nrow + 1));
        minD[0] = maxD[0] = 0; Bp[0] = 0;
        for (int r = 1; r < nrow; ++r) {
            minD[r] = 1 << 29; maxD[r] = -1; Bp[r] = NEG;
            if (!imap[r]) continue;
            int v = g->idx2node[bi + r];
            for (int e = g->node[v].in_head; e >= 0; e = g->edge[e].next_in) {
                int pr = g->node2idx[g->edge[e].from] - bi;
                if (pr < 0 || pr >= nrow || !imap[pr] || maxD[pr] < 0) continue;
                if (minD[pr] + 1 < minD[r]) minD[r] = minD[pr] + 1;
                if (maxD[pr] + 1 > maxD[r]) maxD[r] = maxD[pr] + 1;
                int b = Bp[pr] + ilog2_32(g->edge[e].w); if (b > Bp[r]) Bp[r] = b;
            }
        }
        for (int r = nrow - 1; r >= 0; --r) {
            minR[r] = 1 << 29; maxR[r] = -1; Bs[r] = NEG;
            if (!imap[r]) continue;
            int v = g->idx2node[bi + r];
            for (int e = g->node[v].out_head; e >= 0; e = g->edge[e].next_out) {
                int to = g->edge[e].to, bz = ilog2_32(g->edge[e].w);
                if (to == end_node) { if (0 < minR[r]) minR[r] = 0; if (0 > maxR[r]) maxR[r] = 0; if (bz > Bs[r]) Bs[r] = bz; continue; }
                int x = g->node2idx[to] - bi;
                if (x <= r || x >= nrow || !imap[x] || maxR[x] < 0) continue;
                if (minR[x] + 1 < minR[r]) minR[r] = minR[x] + 1;
                if (maxR[x] + 1 > maxR[r]) maxR[r] = maxR[x] + 1;
                if (Bs[x] + bz > Bs[r]) Bs[r] = Bs[x] + bz;
            }
        }
#define UBP(r, j) (Bp[r] + M * ((j) < maxD[r] ? (j) : maxD[r]) - cert_G(opt, (j) - maxD[r]) - cert_G(opt, minD[r] - (j)))
#define UBS(r, j) (Bs[r] + M * ((qlen - (j)) < maxR[r] ? (qlen - (j)) : maxR[r]) - cert_G(opt, (qlen - (j)) - maxR[r]) - cert_G(opt, minR[r] - (qlen - (j))))
        for (int r = 0; r < nrow; ++r) {
            lo[r] = 1; hi[r] = 0;
            if (!imap[r] || rbeg[r] > rend[r] || maxD[r] < 0 || maxR[r] < 0) continue;
            ++g_cert_rows; g_cert_cells_full += rend[r] - rbeg[r] + 1;
            for (int j = rbeg[r]; j <= rend[r]; ++j) {
                if (CELL(r, j, 0) > UBP(r, j)) ++g_cert_viol_prefix;       /* the prefix half of the bound, checked on every cell of the full DP */
                if (UBP(r, j) + UBS(r, j) + O >= g_cert_slb) { if (lo[r] > hi[r]) lo[r] = j; hi[r] = j; }
            }
            if (lo[r] <= hi[r]) { g_cert_cells_hull += hi[r] - lo[r] + 1; if (hi[r] - lo[r] + 1 > g_cert_maxw) g_cert_maxw = hi[r] - lo[r] + 1; if (hi[r] - lo[r] + 4 > 256) ++g_cert_wide; }
        }
        /* every cell the backtrack visited must be inside its row's hull: matched cells from the cigar */
        for (int t = 0; t < n; ++t) if (cig[t].op == 0) { int r = g->node2idx[cig[t].node] - bi, j = cig[t].qpos + 1; if (j < lo[r] || j > hi[r]) ++g_cert_viol_path; }
        { /* policy: S_est = UB at the end cell - (largest slack seen in this chain so far + margin) */
          int ubtop = NEG;
          for (int e = g->node[end_node].in_head; e >= 0; e = g->edge[e].next_in) { int pr = g->node2idx[g->edge[e].from] - bi; if (pr < 0 || pr >= nrow || !imap[pr] || maxD[pr] < 0) continue; int u = UBP(pr, qlen) + ilog2_32(g->edge[e].w); if (u > ubtop) ubtop = u; }
          const int margin = getenv("LCDO_CERT_MARGIN") ? atoi(getenv("LCDO_CERT_MARGIN")) : 32;
          int delta = g_cert_hist_delta < 0 ? 64 + qlen / 8 : g_cert_hist_delta + g_cert_hist_delta / 4 + margin;
          int sest = ubtop - delta, retry = 0;
          if (sest > best) { retry = 1; sest = best; ++g_cert_est_retry; }     /* the first pass would come back with S' < S_est (pessimistically: S' = S*) */
          int ovf = 0, tovf = 0;
          for (int r = 0; r < nrow; ++r) {
              if (!imap[r] || rbeg[r] > rend[r] || maxD[r] < 0 || maxR[r] < 0) continue;
              int l2 = 1, h2 = 0;
              for (int j = rbeg[r]; j <= rend[r]; ++j) if (UBP(r, j) + UBS(r, j) + O >= sest) { if (l2 > h2) l2 = j; h2 = j; }
              if (l2 <= h2) { g_cert_est_cells += (h2 - l2 + 1) * (retry ? 2 : 1); if (h2 - (l2 & ~3) + 2 > 256) ovf = 1; }
              if (lo[r] <= hi[r] && hi[r] - (lo[r] & ~3) + 2 > 256) tovf = 1;
          }
          g_cert_est_ovf_reads += ovf; g_cert_true_ovf_reads += tovf; if (ovf) g_cert_region_bad = 1;
          if (ubtop - best > g_cert_hist_delta) g_cert_hist_delta = ubtop - best;
        }
        ++g_cert_reads;
        free(minD); free(maxD); free(Bp); free(minR); free(maxR); free(Bs); free(lo); free(hi);
    }
#undef CELL
#undef INBAND
#undef GROW
    free(H); free(imap); free(rbeg); free(rend); free(roff); free(mpl); free(mpr);
    *cig_out = cig;
    return n;
}

/* independent unbanded DAG DP (max score only), different formulation: node-major over ALL nodes in
 * topological order with reachability from beg expressed by NEG, three states per cell. */
static int dag_dp_score(lcdo_poa_t *g, const lcdo_opt_t *opt, int beg_node, int end_node, const uint8_t *seq, int qlen) {
    const int o1 = opt->gap_open1, e1 = opt->gap_ext1, o2 = opt->gap_open2, e2 = opt->gap_ext2;
    int n = g->n_node, W = qlen + 1;
    int *Hh = (int *)malloc((size_t)n * W * sizeof(int)), *D1 = (int *)malloc((size_t)n * W * sizeof(int)),
        *D2 = (int *)malloc((size_t)n * W * sizeof(int));
    for (size_t i = 0; i < (size_t)n * W; ++i) Hh[i] = D1[i] = D2[i] = NEG;
    int bi = g->node2idx[beg_node], ei = g->node2idx[end_node];
    int best = NEG;
    for (int idx = bi; idx <= ei; ++idx) {
        int v = g->idx2node[idx];
        int *h = Hh + (size_t)v * W, *d1 = D1 + (size_t)v * W, *d2 = D2 + (size_t)v * W;
        if (v == beg_node) {
            h[0] = 0;
        } else if (v == end_node) {
            for (int e = g->node[v].in_head; e >= 0; e = g->edge[e].next_in) {
                int p = g->edge[e].from; int pi = g->node2idx[p];
                if (pi < bi) continue;
                int c = Hh[(size_t)p * W + qlen];
                if (c > NEG / 2 && c + ilog2_32(g->edge[e].w) > best) best = c + ilog2_32(g->edge[e].w);
            }
            break;
        } else {
            for (int e = g->node[v].in_head; e >= 0; e = g->edge[e].next_in) {
                int p = g->edge[e].from; if (g->node2idx[p] < bi) continue;
                int bonus = ilog2_32(g->edge[e].w);
                const int *ph = Hh + (size_t)p * W, *pd1 = D1 + (size_t)p * W, *pd2 = D2 + (size_t)p * W;
                for (int j = 0; j <= qlen; ++j) {
                    /* deletion of node v (vertical): enter from pred's H (open) or pred's D (extend) */
                    if (ph[j] > NEG / 2) {
                        int c = ph[j] - o1 - e1 + bonus; if (c > d1[j]) d1[j] = c;
                        c = ph[j] - o2 - e2 + bonus; if (c > d2[j]) d2[j] = c;
                    }
                    if (pd1[j] > NEG / 2 && pd1[j] - e1 + bonus > d1[j]) d1[j] = pd1[j] - e1 + bonus;
                    if (pd2[j] > NEG / 2 && pd2[j] - e2 + bonus > d2[j]) d2[j] = pd2[j] - e2 + bonus;
                    if (j >= 1 && ph[j - 1] > NEG / 2) {
                        int c = ph[j - 1] + sc_mat(opt, g->node[v].base, seq[j - 1]) + bonus;
                        if (c > h[j]) h[j] = c;
                    }
                }
            }
            for (int j = 0; j <= qlen; ++j) { if (d1[j] > h[j]) h[j] = d1[j]; if (d2[j] > h[j]) h[j] = d2[j]; }
        }
        /* horizontal gaps within the row: O(L^2)-free two-state scan on the pre-F maximum */
        int f1 = NEG, f2 = NEG, prev = NEG;
        for (int j = 0; j <= qlen; ++j) {
            int cur = h[j];
            if (j > 0) {
                int a = prev > NEG / 2 ? prev - o1 - e1 : NEG, b = f1 > NEG / 2 ? f1 - e1 : NEG; f1 = a > b ? a : b;
                a = prev > NEG / 2 ? prev - o2 - e2 : NEG; b = f2 > NEG / 2 ? f2 - e2 : NEG; f2 = a > b ? a : b;
                if (f1 > h[j]) h[j] = f1;
                if (f2 > h[j]) h[j] = f2;
            }
            prev = cur;
        }
    }
    free(Hh); free(D1); free(D2);
    return best;
}

/* add the first read as a chain (abpoa_add_graph_sequence) */
static void add_sequence(lcdo_poa_t *g, const uint8_t *seq, int len, int read_id) {
    int last = 0;
    for (int i = 0; i < len; ++i) { int id = add_node(g, seq[i]); add_edge(g, last, id, 0, read_id); last = id; }
    add_edge(g, last, 1, 0, read_id);
}

static void add_alignment(lcdo_poa_t *g, int beg_node, int end_node, const uint8_t *seq, int len, const gcig_t *cig,
                          int n_cig, int read_id) {
    if (g->n_node == 2) { add_sequence(g, seq, len, read_id); topo_sort(g); return; }
    if (n_cig == 0) return;
    int last = beg_node, last_new = 0;
    for (int i = 0; i < n_cig; ++i) {
        uint8_t b = seq[cig[i].qpos];
        if (cig[i].op == 0) {
            int node = cig[i].node;
            if (g->node[node].base != b) {
                int a = get_aligned_id(g, node, b);
                if (a != -1) { add_edge(g, last, a, 1 - last_new, read_id); last = a; last_new = 0; }
                else {
                    int nid = add_node(g, b);
                    add_edge(g, last, nid, 0, read_id); last = nid; last_new = 1;
                    add_aligned(g, node, nid);
                }
            } else { add_edge(g, last, node, 1 - last_new, read_id); last = node; last_new = 0; }
        } else {
            int nid = add_node(g, b);
            add_edge(g, last, nid, 0, read_id); last = nid; last_new = 1;
        }
    }
    add_edge(g, last, end_node, 1 - last_new, read_id);
    topo_sort(g);
}

/* ---- output: MSA rank, rows, clusters, consensus ---- */
static void poa_output(lcdo_poa_t *g, const lcdo_opt_t *opt, int max_n_cons, lcdo_poa_result_t *res) {
    int n = g->n_node, n_seq = g->n_seq;
    int *rank = (int *)malloc(n * sizeof(int));
    for (int i = 0; i < n; ++i) rank[i] = -1;
    int ncol = 0;
    for (int idx = 1; idx < n - 1; ++idx) { /* idx 0 = source, n-1 = sink */
        int v = g->idx2node[idx];
        if (rank[v] >= 0) continue;
        rank[v] = ncol;
        for (int a = g->node[v].aligned_next; a != v; a = g->node[a].aligned_next) rank[a] = ncol;
        ++ncol;
    }
    memset(res, 0, sizeof(*res));
    res->n_seq = n_seq; res->msa_len = ncol;
    res->msa = (uint8_t **)calloc(n_seq + 2, sizeof(uint8_t *));
    for (int i = 0; i < n_seq + 2; ++i) { res->msa[i] = (uint8_t *)malloc(ncol > 0 ? ncol : 1); memset(res->msa[i], LCDO_GAP, ncol); }
    for (int v = 2; v < n; ++v)
        for (int e = g->node[v].out_head; e >= 0; e = g->edge[e].next_out)
            for (int r = 0; r < n_seq; ++r)
                if (g->rid[(size_t)e * g->rid_words + (r >> 6)] >> (r & 63) & 1) res->msa[r][rank[v]] = g->node[v].base;
    /* clustering */
    int *clu = (int *)calloc(n_seq, sizeof(int)); /* cluster of each read */
    int n_clu = 1;
    if (max_n_cons > 1 && n_seq >= 2) {
        int min_w = (int)(n_seq * opt->min_af); if (min_w < 2) min_w = 2;
        int *het = (int *)malloc((ncol > 0 ? ncol : 1) * sizeof(int)), n_het = 0;
        for (int c = 0; c < ncol; ++c) {
            int cnt[6] = {0, 0, 0, 0, 0, 0}, k = 0;
            for (int r = 0; r < n_seq; ++r) cnt[res->msa[r][c]]++;
            for (int a = 0; a < 6; ++a) if (cnt[a] >= min_w) ++k;
            if (k >= 2) het[n_het++] = c;
        }
        if (n_het > 0) {
            /* pivot = the most balanced het column (largest runner-up count, leftmost on ties) */
            int pivot = -1, pv2 = -1, a0 = 0, a1 = 0;
            for (int h = 0; h < n_het; ++h) {
                int cnt[6] = {0, 0, 0, 0, 0, 0};
                for (int r = 0; r < n_seq; ++r) cnt[res->msa[r][het[h]]]++;
                int m0 = 0; for (int a = 1; a < 6; ++a) if (cnt[a] > cnt[m0]) m0 = a;
                int m1 = -1; for (int a = 0; a < 6; ++a) if (a != m0 && (m1 < 0 || cnt[a] > cnt[m1])) m1 = a;
                if (cnt[m1] > pv2) { pv2 = cnt[m1]; pivot = h; a0 = m0; a1 = m1; }
            }
            for (int r = 0; r < n_seq; ++r) {
                int al = res->msa[r][het[pivot]];
                clu[r] = al == a0 ? 0 : al == a1 ? 1 : -1;
            }
            /* Jacobi 2-medians on Hamming distance over the het columns, <= 10 rounds */
            uint8_t *prof = (uint8_t *)malloc((size_t)2 * n_het);
            int *nclu = (int *)malloc(n_seq * sizeof(int));
            for (int it = 0; it < 10; ++it) {
                for (int c = 0; c < 2; ++c)
                    for (int h = 0; h < n_het; ++h) {
                        int cnt[6] = {0, 0, 0, 0, 0, 0};
                        for (int r = 0; r < n_seq; ++r) if (clu[r] == c) cnt[res->msa[r][het[h]]]++;
                        int m0 = 0; for (int a = 1; a < 6; ++a) if (cnt[a] > cnt[m0]) m0 = a;
                        prof[c * n_het + h] = (uint8_t)m0;
                    }
                int changed = 0;
                for (int r = 0; r < n_seq; ++r) {
                    int d0 = 0, d1 = 0;
                    for (int h = 0; h < n_het; ++h) {
                        d0 += res->msa[r][het[h]] != prof[h];
                        d1 += res->msa[r][het[h]] != prof[n_het + h];
                    }
                    nclu[r] = d0 < d1 ? 0 : d1 < d0 ? 1 : (clu[r] >= 0 ? clu[r] : 0);
                    if (nclu[r] != clu[r]) changed = 1;
                }
                memcpy(clu, nclu, n_seq * sizeof(int));
                if (!changed) break;
            }
            int c0 = 0, c1 = 0;
            for (int r = 0; r < n_seq; ++r) if (clu[r]) ++c1; else ++c0;
            if (c0 >= min_w && c1 >= min_w) {
                n_clu = 2;
                if (c1 > c0) for (int r = 0; r < n_seq; ++r) clu[r] ^= 1; /* larger cluster first */
            } else memset(clu, 0, n_seq * sizeof(int));
            free(prof); free(nclu);
        }
        free(het);
    }
    /* consensus per cluster */
    res->n_cons = n_clu;
    for (int c = 0; c < n_clu; ++c) {
        int csize = 0;
        res->clu_read_ids[c] = (int *)malloc(n_seq * sizeof(int));
        for (int r = 0; r < n_seq; ++r) if (clu[r] == c) res->clu_read_ids[c][csize++] = r;
        res->clu_n_seq[c] = csize;
        res->cons_seq[c] = (uint8_t *)malloc(ncol > 0 ? ncol : 1);
        int cl = 0;
        uint8_t *crow = res->msa[n_seq + c];
        for (int col = 0; col < ncol; ++col) {
            int cnt[6] = {0, 0, 0, 0, 0, 0};
            for (int k = 0; k < csize; ++k) cnt[res->msa[res->clu_read_ids[c][k]][col]]++;
            int mb = 0;
            for (int a = 1; a < 5; ++a) if (cnt[a] > cnt[mb]) mb = a;
            if (cnt[mb] > 0 && cnt[mb] >= cnt[5]) { res->cons_seq[c][cl++] = (uint8_t)mb; crow[col] = (uint8_t)mb; }
        }
        res->cons_len[c] = cl;
    }
    if (n_clu == 1) { free(res->msa[n_seq + 1]); res->msa[n_seq + 1] = NULL; }
    free(rank); free(clu);
}

void lcdo_poa_result_free(lcdo_poa_result_t *r) {
    if (!r->msa) return;
    for (int i = 0; i < r->n_seq + 2; ++i) free(r->msa[i]);
    free(r->msa);
    for (int c = 0; c < 2; ++c) { free(r->cons_seq[c]); free(r->clu_read_ids[c]); }
    memset(r, 0, sizeof(*r));
}

/* K1: src/align.c:762-857 */
int lcdo_poa_partial_aln_msa_cons(const lcdo_opt_t *opt, int sampling_reads, int n_reads, uint8_t **read_seqs,
                                  const int *read_lens, const int *read_full_cover, lcdo_poa_result_t *res) {
    lcdo_poa_t *g = poa_init(n_reads);
    for (int i = 0; i < n_reads; ++i) {
        int exc_beg = 0, exc_end = 1, beg_cut = 0, end_cut = 0;
        if (i != 0) {
            int ref_beg, ref_end, read_beg, read_end;
            if (lcdo_collect_partial_aln_beg_end(opt, sampling_reads, read_seqs[0], read_lens[0], read_full_cover[0],
                                                 read_seqs[i], read_lens[i], read_full_cover[i], &ref_beg, &ref_end,
                                                 &read_beg, &read_end) == 0) continue;
            beg_cut = read_beg - 1; end_cut = read_lens[i] - read_end;
            subgraph_nodes(g, ref_beg + 1, ref_end + 1, &exc_beg, &exc_end);
        }
        const uint8_t *seq = read_seqs[i] + beg_cut; int len = read_lens[i] - beg_cut - end_cut;
        gcig_t *cig = NULL; int n_cig = 0, sc = NEG;
        if (g->n_node > 2) {
            n_cig = align_to_subgraph(g, opt, 10, 0.01, exc_beg, exc_end, seq, len, &cig, &sc);
            g_dbg_banded = sc;
            if (getenv("LCDO_POA_CHECK_UNBANDED")) g_dbg_unbanded = dag_dp_score(g, opt, exc_beg, exc_end, seq, len);
        }
        if (len > 0) add_alignment(g, exc_beg, exc_end, seq, len, cig, n_cig, i);
        free(cig);
    }
    if (g->idx2node == NULL) topo_sort(g);
    poa_output(g, opt, 1, res);
    int n_cons = res->n_cons;
    poa_free(g);
    return n_cons;
}

/* K1 with the anchors GIVEN (test infrastructure: a chain dumped by the library, LCD_DUMP_CHAIN, carries the anchors its anchor stage computed -- the loop is the one
   above with collect_partial_aln_beg_end's results read from `anchors[4 * i]` = {ref_beg, ref_end, read_beg, read_end} and reads with skip[i] != 0 left out) */
int lcdo_poa_partial_aln_msa_cons_anchored(const lcdo_opt_t *opt, int n_reads, uint8_t **read_seqs, const int *read_lens, const int *anchors, const int *skip,
                                           lcdo_poa_result_t *res) {
    lcdo_poa_t *g = poa_init(n_reads);
    for (int i = 0; i < n_reads; ++i) {
        int exc_beg = 0, exc_end = 1, beg_cut = 0, end_cut = 0;
        if (skip && skip[i]) continue;
        if (i != 0) {
            const int ref_beg = anchors[4 * i], ref_end = anchors[4 * i + 1], read_beg = anchors[4 * i + 2], read_end = anchors[4 * i + 3];
            beg_cut = read_beg - 1; end_cut = read_lens[i] - read_end;
            subgraph_nodes(g, ref_beg + 1, ref_end + 1, &exc_beg, &exc_end);
        }
        const uint8_t *seq = read_seqs[i] + beg_cut; int len = read_lens[i] - beg_cut - end_cut;
        gcig_t *cig = NULL; int n_cig = 0, sc = NEG;
        if (g->n_node > 2) n_cig = align_to_subgraph(g, opt, 10, 0.01, exc_beg, exc_end, seq, len, &cig, &sc);
        if (len > 0) add_alignment(g, exc_beg, exc_end, seq, len, cig, n_cig, i);
        free(cig);
    }
    if (g->idx2node == NULL) topo_sort(g);
    poa_output(g, opt, 1, res);
    int n_cons = res->n_cons;
    poa_free(g);
    return n_cons;
}

/* K2: src/align.c:872-943 (abpoa_msa on all reads, unbanded, <= max_n_cons consensus) */
int lcdo_poa_aln_msa_cons(const lcdo_opt_t *opt, int n_reads, uint8_t **read_seqs, const int *read_lens, int max_n_cons,
                          lcdo_poa_result_t *res) {
    lcdo_poa_t *g = poa_init(n_reads);
    g_cert_hist_delta = -1; if (g_cert_regions) g_cert_regions_allok += !g_cert_region_bad; ++g_cert_regions; g_cert_region_bad = 0;
    for (int i = 0; i < n_reads; ++i) {
        gcig_t *cig = NULL; int n_cig = 0, sc = NEG;
        if (g->n_node > 2) {
            g_cert_on = getenv("LCDO_CERT_STATS") ? 1 : 0;
            if (g_cert_on) { gcig_t *c1 = NULL; int s1 = NEG; align_to_subgraph(g, opt, 10, 0.01, 0, 1, read_seqs[i], read_lens[i], &c1, &s1); free(c1); g_cert_slb = s1; }
            n_cig = align_to_subgraph(g, opt, -1, 0.0, 0, 1, read_seqs[i], read_lens[i], &cig, &sc);
            g_dbg_banded = sc;
            if (getenv("LCDO_POA_CHECK_UNBANDED")) g_dbg_unbanded = dag_dp_score(g, opt, 0, 1, read_seqs[i], read_lens[i]);
        }
        if (read_lens[i] > 0) add_alignment(g, 0, 1, read_seqs[i], read_lens[i], cig, n_cig, i);
        free(cig);
    }
    if (g->idx2node == NULL) topo_sort(g);
    poa_output(g, opt, max_n_cons, res);
    int n_cons = res->n_cons;
    poa_free(g);
    return n_cons;
}

void lcdo_poa_cert_stats(long long *out) { /* experiment: rows, full cells, hull cells, reads, prefix-bound violations, path violations, widest hull, rows wider than 252 */
    out[0] = g_cert_rows; out[1] = g_cert_cells_full; out[2] = g_cert_cells_hull; out[3] = g_cert_reads; out[4] = g_cert_viol_prefix; out[5] = g_cert_viol_path; out[6] = g_cert_maxw; out[7] = g_cert_wide; out[8] = 0; out[9] = 0; out[10] = 0; out[11] = g_cert_est_cells; out[12] = g_cert_est_retry; out[13] = g_cert_est_ovf_reads; out[14] = g_cert_true_ovf_reads; out[15] = g_cert_regions_allok; out[16] = g_cert_regions;
}
int lcdo_poa_debug_last_scores(int *banded, int *unbanded) {
    *banded = g_dbg_banded; *unbanded = g_dbg_unbanded;
    return 0;
}
