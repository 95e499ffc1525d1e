/*
 * wfa2p.c -- ORACLE (test infrastructure only; see lcd_oracle.h).  PARITY UNPINNED (byte level).
 *
 * Restates what src/align.c:374-460 (wfa_end2end_aln) asks of WFA2-lib for the only configuration
 * that is live on the germline path: distance_metric = gap_affine_2p, match = 0, heuristic = none,
 * end-to-end span, memory_mode = high (all wavefronts kept, direct backtrace).
 *
 * WFA2-lib itself (github.com/smarco/WFA2-lib, git submodule `WFA2-lib`, pin unknown) is NOT in
 * /root/reference, so this file follows the published algorithm (Marco-Sola et al., "Fast gap-affine
 * pairwise alignment using the wavefront algorithm", 2021; "Optimal gap-affine alignment in O(s)
 * space", 2023) plus the library behaviour recalled in SURVEY.md Appendix D:
 *   I1[s][k] = max(M[s-o1-e1][k-1], I1[s-e1][k-1]) + 1      (I = consumes TEXT)
 *   D1[s][k] = max(M[s-o1-e1][k+1], D1[s-e1][k+1])          (D = consumes PATTERN)
 *   M [s][k] = max(M[s-x][k]+1, I1, I2, D1, D2), nulled when h>tlen or v>plen, then extended.
 * Backtrace: in M, all nine sources compete as (offset<<4 | type) with
 *   M(misms)=9 > D2_ext=8 > D2_open=7 > D1_ext=6 > D1_open=5 > I2_ext=4 > I2_open=3 > I1_ext=2 > I1_open=1,
 * matches = offset - best source offset are emitted first, then the operation.
 * What IS pinned: the optimal score, against the independent O(nm) Gotoh DP in gotoh2p.c, and CIGAR
 * validity (tests/test_oracle_wfa.py).  The tie-break order above is this project's definition until
 * upstream sources can be read.
 */
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

#define WF_NULL (-(1 << 29))

typedef struct {
    int lo, hi;       /* lo > hi: null wavefront */
    int *m, *i1, *i2, *d1, *d2; /* indexed [k - lo] */
} wf_t;

typedef struct {
    wf_t *wf;
    int n, cap;
} wfset_t;

static inline int wf_get(const wfset_t *S, int comp, int s, int k) {
    if (s < 0 || s >= S->n) return WF_NULL;
    const wf_t *w = &S->wf[s];
    if (w->lo > w->hi || k < w->lo || k > w->hi) return WF_NULL;
    const int *a = comp == 0 ? w->m : comp == 1 ? w->i1 : comp == 2 ? w->i2 : comp == 3 ? w->d1 : w->d2;
    return a[k - w->lo];
}

static inline int wf_exists(const wfset_t *S, int s) { return s >= 0 && s < S->n && S->wf[s].lo <= S->wf[s].hi; }

static void wf_alloc(wf_t *w, int lo, int hi) {
    int n = hi - lo + 1;
    w->lo = lo; w->hi = hi;
    w->m = (int *)malloc(5 * (size_t)n * sizeof(int));
    w->i1 = w->m + n; w->i2 = w->i1 + n; w->d1 = w->i2 + n; w->d2 = w->d1 + n;
}

static inline int max2(int a, int b) { return a > b ? a : b; }

/* forward WFA + backtrace; ops out as chars 'M','X','I','D' in start->end order. returns score */
static int wfa2p_core(const uint8_t *p, int plen, const uint8_t *t, int tlen, int x, int o1, int e1, int o2, int e2,
                      char **ops_out, int *n_ops_out) {
    wfset_t S = {0, 0, 0};
    S.cap = 64; S.wf = (wf_t *)malloc(S.cap * sizeof(wf_t));
    const int k_end = tlen - plen;
    int s = 0;
    wf_alloc(&S.wf[0], 0, 0);
    S.wf[0].m[0] = 0; S.wf[0].i1[0] = S.wf[0].i2[0] = S.wf[0].d1[0] = S.wf[0].d2[0] = WF_NULL;
    S.n = 1;
    for (;;) {
        wf_t *w = &S.wf[s];
        if (w->lo <= w->hi) {
            /* extend */
            for (int k = w->lo; k <= w->hi; ++k) {
                int h = w->m[k - w->lo];
                if (h < 0) continue;
                int v = h - k;
                while (v < plen && h < tlen && p[v] == t[h]) { ++v; ++h; }
                w->m[k - w->lo] = h;
            }
            if (k_end >= w->lo && k_end <= w->hi && w->m[k_end - w->lo] >= tlen) break;
        }
        /* next score */
        ++s;
        if (s >= S.cap) { S.cap *= 2; S.wf = (wf_t *)realloc(S.wf, S.cap * sizeof(wf_t)); }
        S.n = s + 1;
        wf_t *nw = &S.wf[s];
        nw->lo = 1; nw->hi = 0; nw->m = NULL;
        int src[7] = {s - x, s - o1 - e1, s - o2 - e2, s - e1, s - e2, s - e1, s - e2};
        int lo = 1 << 30, hi = -(1 << 30), any = 0;
        for (int i = 0; i < 5; ++i) { /* M[s-x], M[s-o1-e1], M[s-o2-e2], {I1,D1}[s-e1], {I2,D2}[s-e2] */
            if (wf_exists(&S, src[i])) {
                any = 1;
                if (S.wf[src[i]].lo < lo) lo = S.wf[src[i]].lo;
                if (S.wf[src[i]].hi > hi) hi = S.wf[src[i]].hi;
            }
        }
        if (!any) continue;
        lo -= 1; hi += 1;
        wf_alloc(nw, lo, hi);
        for (int k = lo; k <= hi; ++k) {
            int i1 = max2(wf_get(&S, 0, s - o1 - e1, k - 1), wf_get(&S, 1, s - e1, k - 1)) + 1;
            int i2 = max2(wf_get(&S, 0, s - o2 - e2, k - 1), wf_get(&S, 2, s - e2, k - 1)) + 1;
            int d1 = max2(wf_get(&S, 0, s - o1 - e1, k + 1), wf_get(&S, 3, s - e1, k + 1));
            int d2 = max2(wf_get(&S, 0, s - o2 - e2, k + 1), wf_get(&S, 4, s - e2, k + 1));
            int mm = wf_get(&S, 0, s - x, k) + 1;
            int m = max2(max2(mm, max2(i1, i2)), max2(d1, d2));
            if (i1 < 0) i1 = WF_NULL;
            if (i2 < 0) i2 = WF_NULL;
            if (d1 < 0) d1 = WF_NULL;
            if (d2 < 0) d2 = WF_NULL;
            if (m < 0 || m > tlen || m - k > plen || m - k < 0) m = WF_NULL; /* out of the DP matrix */
            nw->m[k - lo] = m; nw->i1[k - lo] = i1; nw->i2[k - lo] = i2; nw->d1[k - lo] = d1; nw->d2[k - lo] = d2;
        }
    }
    const int score = s;
    if (ops_out) {
        /* backtrace */
        char *rev = (char *)malloc(plen + tlen + 2);
        int n = 0, k = k_end, off = tlen, type = 0; /* 0 M, 1 I1, 2 I2, 3 D1, 4 D2 */
        int h = off, v = off - k;
        while (v > 0 && h > 0 && s > 0) {
            const int mism = s - x, go1 = s - o1 - e1, ge1 = s - e1, go2 = s - o2 - e2, ge2 = s - e2;
            long long best = (long long)WF_NULL * 16, c;
#define CAND(val, ty) do { c = (long long)(val) * 16 + (ty); if ((val) >= 0 && c > best) best = c; } while (0)
            if (type == 0) {
                CAND(wf_get(&S, 0, mism, k) + 1, 9);
                CAND(wf_get(&S, 0, go1, k - 1) + 1, 1); CAND(wf_get(&S, 1, ge1, k - 1) + 1, 2);
                CAND(wf_get(&S, 0, go2, k - 1) + 1, 3); CAND(wf_get(&S, 2, ge2, k - 1) + 1, 4);
                CAND(wf_get(&S, 0, go1, k + 1), 5); CAND(wf_get(&S, 3, ge1, k + 1), 6);
                CAND(wf_get(&S, 0, go2, k + 1), 7); CAND(wf_get(&S, 4, ge2, k + 1), 8);
            } else if (type == 1) {
                CAND(wf_get(&S, 0, go1, k - 1) + 1, 1); CAND(wf_get(&S, 1, ge1, k - 1) + 1, 2);
            } else if (type == 2) {
                CAND(wf_get(&S, 0, go2, k - 1) + 1, 3); CAND(wf_get(&S, 2, ge2, k - 1) + 1, 4);
            } else if (type == 3) {
                CAND(wf_get(&S, 0, go1, k + 1), 5); CAND(wf_get(&S, 3, ge1, k + 1), 6);
            } else {
                CAND(wf_get(&S, 0, go2, k + 1), 7); CAND(wf_get(&S, 4, ge2, k + 1), 8);
            }
#undef CAND
            if (best < 0) break; /* no source: cannot happen for a consistent wavefront set */
            const int boff = (int)(best / 16), bty = (int)(best % 16);
            if (type == 0) {
                int nm = off - boff;
                for (int i = 0; i < nm; ++i) rev[n++] = 'M';
                off = boff; h = off; v = off - k;
                if (v <= 0 || h <= 0) break;
            }
            switch (bty) {
            case 9: s = mism; type = 0; rev[n++] = 'X'; --off; break;
            case 1: s = go1; type = 0; rev[n++] = 'I'; --k; --off; break;
            case 2: s = ge1; type = 1; rev[n++] = 'I'; --k; --off; break;
            case 3: s = go2; type = 0; rev[n++] = 'I'; --k; --off; break;
            case 4: s = ge2; type = 2; rev[n++] = 'I'; --k; --off; break;
            case 5: s = go1; type = 0; rev[n++] = 'D'; ++k; break;
            case 6: s = ge1; type = 3; rev[n++] = 'D'; ++k; break;
            case 7: s = go2; type = 0; rev[n++] = 'D'; ++k; break;
            case 8: s = ge2; type = 4; rev[n++] = 'D'; ++k; break;
            }
            h = off; v = off - k;
        }
        if (v > 0 && h > 0) {
            int nm = v < h ? v : h;
            for (int i = 0; i < nm; ++i) rev[n++] = 'M';
            v -= nm; h -= nm;
        }
        while (v > 0) { rev[n++] = 'D'; --v; }
        while (h > 0) { rev[n++] = 'I'; --h; }
        char *ops = (char *)malloc(n + 1);
        for (int i = 0; i < n; ++i) ops[i] = rev[n - 1 - i];
        ops[n] = 0;
        free(rev);
        *ops_out = ops; *n_ops_out = n;
    }
    for (int i = 0; i < S.n; ++i) if (S.wf[i].lo <= S.wf[i].hi) free(S.wf[i].m);
    free(S.wf);
    return score;
}

/* src/align.c:374-460 for heuristic = NONE, affine_gap = 2P */
int lcdo_wfa_end2end_aln(const uint8_t *pattern, int plen, const uint8_t *text, int tlen, int gap_aln, int b, int q, int e,
                         int q2, int e2, uint32_t **cigar_buf, int *cigar_length, uint8_t **pattern_alg,
                         uint8_t **text_alg, int *alg_length, int *score) {
    uint8_t *p = (uint8_t *)pattern, *t = (uint8_t *)text;
    if (gap_aln == LCDO_GAP_LEFT_ALN) { /* :409-414 */
        p = (uint8_t *)malloc(plen + 1); t = (uint8_t *)malloc(tlen + 1);
        for (int i = 0; i < plen; ++i) p[i] = pattern[plen - i - 1];
        for (int i = 0; i < tlen; ++i) t[i] = text[tlen - i - 1];
    }
    char *ops; int n_ops;
    int sc = wfa2p_core(p, plen, t, tlen, b, q, e, q2, e2, &ops, &n_ops);
    if (score) *score = sc;
    if (cigar_buf && cigar_length) { /* cigar_get_CIGAR(show_mismatches=true) then optional reversal, :430-443 */
        uint32_t *tmp = (uint32_t *)malloc((n_ops + 1) * sizeof(uint32_t));
        int nc = 0;
        for (int i = 0; i < n_ops;) {
            int j = i;
            while (j < n_ops && ops[j] == ops[i]) ++j;
            uint32_t op = ops[i] == 'M' ? LCDO_CEQUAL : ops[i] == 'X' ? LCDO_CDIFF : ops[i] == 'I' ? LCDO_CINS : LCDO_CDEL;
            tmp[nc++] = ((uint32_t)(j - i) << 4) | op;
            i = j;
        }
        *cigar_buf = (uint32_t *)malloc((nc > 0 ? nc : 1) * sizeof(uint32_t));
        for (int i = 0; i < nc; ++i) (*cigar_buf)[i] = gap_aln == LCDO_GAP_LEFT_ALN ? tmp[nc - i - 1] : tmp[i];
        *cigar_length = nc;
        free(tmp);
    }
    if (pattern_alg && text_alg) { /* wfa_collect_pretty_alignment :277-329, then reversal :445-455 */
        const int maxl = tlen + plen + 1;
        uint8_t *mem = (uint8_t *)calloc(2 * (size_t)maxl, 1);
        uint8_t *pa = mem, *ta = mem + maxl;
        int n = 0, pp = 0, tp = 0;
        for (int i = 0; i < n_ops; ++i) {
            switch (ops[i]) {
            case 'M': case 'X': pa[n] = p[pp++]; ta[n++] = t[tp++]; break;
            case 'I': pa[n] = LCDO_GAP; ta[n++] = t[tp++]; break;
            case 'D': pa[n] = p[pp++]; ta[n++] = LCDO_GAP; break;
            }
        }
        if (gap_aln == LCDO_GAP_LEFT_ALN) {
            for (int i = 0; i < n / 2; ++i) {
                uint8_t x = pa[i]; pa[i] = pa[n - i - 1]; pa[n - i - 1] = x;
                x = ta[i]; ta[i] = ta[n - i - 1]; ta[n - i - 1] = x;
            }
        }
        *pattern_alg = pa; *text_alg = ta; *alg_length = n;
    }
    free(ops);
    if (gap_aln == LCDO_GAP_LEFT_ALN) { free(p); free(t); }
    return 0;
}

/* ---- independent 2-piece Gotoh (score pin) ---- */
int lcdo_gotoh2p_score(const uint8_t *p, int plen, const uint8_t *t, int tlen, int x, int o1, int e1, int o2, int e2) {
    const int INF = 1 << 29;
    int W = tlen + 1;
    int *H = (int *)malloc(5 * (size_t)W * sizeof(int));
    int *V1 = H + W, *V2 = V1 + W; /* vertical gaps (consume pattern) ending at column j */
    int *Hn = V2 + W;
    int *Hp = H;
    /* row 0 */
    Hp[0] = 0; V1[0] = V2[0] = INF;
    for (int j = 1; j <= tlen; ++j) {
        int a = o1 + e1 * j, b2 = o2 + e2 * j;
        Hp[j] = a < b2 ? a : b2; V1[j] = V2[j] = INF;
    }
    for (int i = 1; i <= plen; ++i) {
        int h1 = INF, h2 = INF; /* horizontal gaps (consume text) */
        for (int j = 0; j <= tlen; ++j) {
            int v1 = V1[j] + e1, v2 = V2[j] + e2;
            if (Hp[j] + o1 + e1 < v1) v1 = Hp[j] + o1 + e1;
            if (Hp[j] + o2 + e2 < v2) v2 = Hp[j] + o2 + e2;
            V1[j] = v1; V2[j] = v2;
            int best = v1 < v2 ? v1 : v2;
            if (j > 0) {
                int a = h1 + e1, b2 = h2 + e2;
                if (Hn[j - 1] + o1 + e1 < a) a = Hn[j - 1] + o1 + e1;
                if (Hn[j - 1] + o2 + e2 < b2) b2 = Hn[j - 1] + o2 + e2;
                h1 = a; h2 = b2;
                if (h1 < best) best = h1;
                if (h2 < best) best = h2;
                int d = Hp[j - 1] + (p[i - 1] == t[j - 1] ? 0 : x);
                if (d < best) best = d;
            }
            Hn[j] = best;
        }
        int *sw = Hp; Hp = Hn; Hn = sw;
    }
    int r = Hp[tlen];
    free(H);
    return r;
}

int lcdo_cigar_score2p(const uint32_t *cigar, int n_cigar, const uint8_t *p, int plen, const uint8_t *t, int tlen, int x,
                       int o1, int e1, int o2, int e2) {
    int pi = 0, ti = 0, sc = 0;
    for (int i = 0; i < n_cigar; ++i) {
        int op = cigar[i] & 0xf, len = cigar[i] >> 4;
        if (len <= 0) return -1;
        if (op == LCDO_CEQUAL || op == LCDO_CDIFF) {
            if (pi + len > plen || ti + len > tlen) return -1;
            for (int j = 0; j < len; ++j) {
                int eq = p[pi + j] == t[ti + j];
                if (eq != (op == LCDO_CEQUAL)) return -1;
                if (!eq) sc += x;
            }
            pi += len; ti += len;
        } else if (op == LCDO_CINS || op == LCDO_CDEL) {
            /* adjacent same-type runs would have been merged, so each run is one gap */
            int a = o1 + e1 * len, b2 = o2 + e2 * len;
            sc += a < b2 ? a : b2;
            if (op == LCDO_CINS) ti += len; else pi += len;
            if (pi > plen || ti > tlen) return -1;
        } else return -1;
    }
    if (pi != plen || ti != tlen) return -1;
    return sc;
}
