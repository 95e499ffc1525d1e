/*
 * edlib_nw.c -- ORACLE (test infrastructure only; see lcd_oracle.h).
 *
 * Plain-C restatement of the part of edlib that longcallD's germline path uses:
 * edlibAlign(query, target, {k=-1, EDLIB_MODE_NW, EDLIB_TASK_PATH})  (src/align.c:222-232).
 *
 * edlib computes the optimal edit distance with Myers' bit-vector algorithm inside an Ukkonen
 * band (edlib/src/edlib.cpp:730-928) and then extracts ONE specific optimal path:
 *   - if (2*8+4)*ceil(q/64)*t + 8*t < 1 MiB: stored-matrix traceback with precedence
 *     Up(INSERT) -> Left(DELETE) -> Diagonal (edlib.cpp:942-1141, tests at :1020,:1054,:1085);
 *   - else Hirschberg: split the target at t/2, take the FIRST query row (ascending) where
 *     left+right == best, then the two boundary cases, recurse (edlib.cpp:1231-1396).
 * Every cell on an optimal path is inside the band and exact, and a banded over-estimate can
 * never satisfy an equality test of the traceback (|delta| <= 1 between neighbours), so a plain
 * full-matrix DP with the same precedence and the same split rule reproduces edlib's path
 * byte for byte.  That claim is not taken on trust: tests/test_oracle_edlib.py checks this file
 * against the reference's real edlib (oracle/_ref) and tests/golden/edlib_golden.json.
 */
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

typedef struct {
    uint8_t *ops;
    int n, m;
} opbuf_t;

static void op_push(opbuf_t *b, uint8_t op, int cnt) {
    if (b->n + cnt > b->m) {
        b->m = (b->n + cnt) * 2 + 16;
        b->ops = (uint8_t *)realloc(b->ops, b->m);
    }
    memset(b->ops + b->n, op, cnt);
    b->n += cnt;
}

/* column of NW scores after consuming ncols target chars: col[r] = D(query[0..r], target[0..ncols-1]).
 * rev=1 walks both strings from their ends (the reverse pass of edlib.cpp:1262-1265). */
static void nw_column(const uint8_t *q, int qlen, const uint8_t *t, int tlen_total, int ncols, int rev, int *col) {
    for (int r = 0; r < qlen; ++r) col[r] = r + 1; /* boundary column -1 */
    for (int c = 0; c < ncols; ++c) {
        uint8_t tc = rev ? t[tlen_total - 1 - c] : t[c];
        int diag = c;    /* D[-1][c-1] */
        int up = c + 1;  /* D[-1][c]   */
        for (int r = 0; r < qlen; ++r) {
            uint8_t qc = rev ? q[qlen - 1 - r] : q[r];
            int left = col[r];
            int v = diag + (qc != tc);
            if (up + 1 < v) v = up + 1;
            if (left + 1 < v) v = left + 1;
            diag = left;
            col[r] = v;
            up = v;
        }
    }
}

/* stored-matrix traceback, edlib.cpp:942-1141; appends ops in start->end order */
static void traceback_full(const uint8_t *q, int qlen, const uint8_t *t, int tlen, opbuf_t *out) {
    int W = tlen + 1;
    int *D = (int *)malloc((size_t)(qlen + 1) * W * sizeof(int));
    for (int c = 0; c <= tlen; ++c) D[c] = c;
    for (int r = 1; r <= qlen; ++r) {
        int *row = D + (size_t)r * W, *prow = row - W;
        row[0] = r;
        for (int c = 1; c <= tlen; ++c) {
            int v = prow[c - 1] + (q[r - 1] != t[c - 1]);
            if (prow[c] + 1 < v) v = prow[c] + 1;
            if (row[c - 1] + 1 < v) v = row[c - 1] + 1;
            row[c] = v;
        }
    }
    uint8_t *rev = (uint8_t *)malloc(qlen + tlen + 1);
    int n = 0, r = qlen, c = tlen;
    while (r > 0 && c > 0) {
        int cur = D[(size_t)r * W + c];
        if (D[(size_t)(r - 1) * W + c] + 1 == cur) { /* Up first (edlib.cpp:1020) */
            rev[n++] = LCDO_EDOP_INSERT; --r;
        } else if (D[(size_t)r * W + c - 1] + 1 == cur) { /* then Left (:1054) */
            rev[n++] = LCDO_EDOP_DELETE; --c;
        } else { /* Diagonal (:1085) */
            rev[n++] = D[(size_t)(r - 1) * W + c - 1] == cur ? LCDO_EDOP_MATCH : LCDO_EDOP_MISMATCH;
            --r; --c;
        }
    }
    while (c > 0) { rev[n++] = LCDO_EDOP_DELETE; --c; }
    while (r > 0) { rev[n++] = LCDO_EDOP_INSERT; --r; }
    for (int i = n - 1; i >= 0; --i) op_push(out, rev[i], 1);
    free(rev); free(D);
}

static int obtain_alignment(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int best, opbuf_t *out);

/* edlib.cpp:1231-1396 */
static int obtain_hirschberg(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int best, opbuf_t *out) {
    int left_w = tlen / 2, right_w = tlen - left_w;
    int *left = (int *)malloc(qlen * sizeof(int)), *rrev = (int *)malloc(qlen * sizeof(int));
    nw_column(q, qlen, t, tlen, left_w, 0, left);
    nw_column(q, qlen, t, tlen, right_w, 1, rrev);
    /* right[idx] = rrev[qlen-1-idx] = cost of query[idx..] vs target[left_w..] */
    int found = 0, q_idx = -1, lscore = -1, rscore = -1;
    for (int i = 0; i <= qlen - 2; ++i) {
        lscore = left[i]; rscore = rrev[qlen - 1 - (i + 1)];
        if (lscore + rscore == best) { q_idx = i; found = 1; break; }
    }
    if (!found) {
        lscore = left_w; rscore = rrev[qlen - 1];
        if (lscore + rscore == best) { q_idx = -1; found = 1; }
    }
    if (!found) {
        lscore = left[qlen - 1]; rscore = right_w;
        if (lscore + rscore == best) { q_idx = qlen - 1; found = 1; }
    }
    free(left); free(rrev);
    if (!found) return -1;
    int ul_h = q_idx + 1, lr_h = qlen - ul_h;
    if (obtain_alignment(q, ul_h, t, left_w, lscore, out) < 0) return -1;
    if (obtain_alignment(q + ul_h, lr_h, t + left_w, right_w, rscore, out) < 0) return -1;
    return 0;
}

/* edlib.cpp:1161-1213 */
static int obtain_alignment(const uint8_t *q, int qlen, const uint8_t *t, int tlen, int best, opbuf_t *out) {
    if (qlen == 0 || tlen == 0) {
        op_push(out, qlen == 0 ? LCDO_EDOP_DELETE : LCDO_EDOP_INSERT, qlen + tlen);
        return 0;
    }
    long long max_blocks = (qlen + 63) / 64;
    long long data_size = (2ll * 8 + 4) * max_blocks * tlen + 2ll * 4 * tlen;
    if (data_size < 1024 * 1024) {
        traceback_full(q, qlen, t, tlen, out);
        return 0;
    }
    return obtain_hirschberg(q, qlen, t, tlen, best, out);
}

int lcdo_edlib_nw(const uint8_t *query, int qlen, const uint8_t *target, int tlen, uint8_t **aln, int *aln_len) {
    if (aln) { *aln = NULL; *aln_len = 0; }
    if (qlen == 0 || tlen == 0) return qlen > tlen ? qlen : tlen; /* edlib.cpp:166-173: no alignment produced */
    int *col = (int *)malloc(qlen * sizeof(int));
    nw_column(query, qlen, target, tlen, tlen, 0, col);
    int best = col[qlen - 1];
    free(col);
    if (aln) {
        opbuf_t out = {0, 0, 0};
        if (obtain_alignment(query, qlen, target, tlen, best, &out) < 0) { free(out.ops); return -1; }
        *aln = out.ops; *aln_len = out.n;
    }
    return best;
}

/* src/align.c:189-208 */
static int aln_to_xgaps(const uint8_t *a, int n) {
    int n_gaps = 0, n_mis = 0;
    for (int i = 0; i < n; ++i) {
        if (a[i] == LCDO_EDOP_MATCH) continue;
        else if (a[i] == LCDO_EDOP_MISMATCH) n_mis++;
        else if (i == 0 || a[i - 1] != a[i]) n_gaps++;
    }
    return n_mis + n_gaps;
}

int lcdo_edlib_xgaps(const uint8_t *target, int tlen, const uint8_t *query, int qlen) {
    uint8_t *aln; int n;
    if (lcdo_edlib_nw(query, qlen, target, tlen, &aln, &n) < 0) return -1;
    int x = aln_to_xgaps(aln, n);
    free(aln);
    return x;
}

int lcdo_edlib_edit_distance(const uint8_t *target, int tlen, const uint8_t *query, int qlen) {
    return lcdo_edlib_nw(query, qlen, target, tlen, NULL, NULL);
}

/* src/align.c:234-254 with edlibAlignmentToXID :164-187 */
int lcdo_edlib_end2end_aln(const uint8_t *target, int tlen, const uint8_t *query, int qlen, int *n_eq, int *n_xid) {
    uint8_t *aln; int n;
    int d = lcdo_edlib_nw(query, qlen, target, tlen, &aln, &n);
    if (d < 0) return -1;
    if (n_eq && n_xid) {
        int eq = 0, x = 0;
        for (int i = 0; i < n; ++i) { if (aln[i] == LCDO_EDOP_MATCH) eq++; else x++; }
        *n_eq = eq; *n_xid = x;
    }
    free(aln);
    return d;
}


/* ---------------- HW (infix) mode: edlibAlign(query, target, {k = -1, EDLIB_MODE_HW, EDLIB_TASK_PATH}), src/align.c:256-275 ----------------
 * edlib/src/edlib.cpp:146-280: (1) the best score over all end positions of the target with a free start (first DP row all zero) and the end positions that reach
 * it, in increasing order; (2) for an end position the start: the SAME problem reversed -- reversed query against the reversed target prefix that ends there, this
 * time with the prefix anchored (SHW: first row 0, 1, 2, ...) -- whose LAST position with the best score is taken (:243-262: `positionsSHW[numPositionsSHW - 1]`),
 * i.e. the longest target stretch; (3) the path = the NW alignment (same traceback as above) of the query against target[start0 .. end0] for the FIRST end
 * position.  Full matrices here: test infrastructure, sizes of a few thousand. */
int lcdo_edlib_hw(const uint8_t *query, int qlen, const uint8_t *target, int tlen, int *start_out, int *end_out, uint8_t **aln, int *aln_len) {
    if (aln) { *aln = NULL; *aln_len = 0; }
    if (start_out) *start_out = -1;
    if (end_out) *end_out = -1;
    if (qlen == 0 || tlen == 0) return qlen; /* edlib.cpp:166-177: in HW (and SHW) mode an empty side gives editDistance = queryLength -- an empty query matches the empty infix --
                                               * and endLocations[0] = -1, no path (NW: max of the lengths).  Pinned by the round-4 HW vectors of tests/golden/edlib_golden.json */
    int *prev = (int *)malloc((size_t)(tlen + 1) * sizeof(int)), *cur = (int *)malloc((size_t)(tlen + 1) * sizeof(int));
    for (int j = 0; j <= tlen; ++j) prev[j] = 0;                       /* free start anywhere in the target */
    for (int i = 1; i <= qlen; ++i) {
        cur[0] = i;
        for (int j = 1; j <= tlen; ++j) {
            int d = prev[j - 1] + (query[i - 1] != target[j - 1]), u = prev[j] + 1, l = cur[j - 1] + 1;
            cur[j] = d < u ? (d < l ? d : l) : (u < l ? u : l);
        }
        int *t = prev; prev = cur; cur = t;
    }
    /* the first end position with the best score.  Position -1 -- the query placed in front of the target, score qlen -- is a candidate too (edlib.cpp:659-691: the
     * padded last block reports it through "c - W" / "targetLength - W + i"; :222-236 then takes 0 as its start), so a query that matches nothing ends at -1 */
    int best = qlen, end0 = -1;
    for (int j = 1; j <= tlen; ++j) if (prev[j] < best) { best = prev[j]; end0 = j - 1; }
    if (end0 < 0) { /* all of the query inserted in front of the target: start 0 (edlib.cpp:236), the path of an empty target (obtainAlignment: queryLength insertions) */
        free(prev); free(cur);
        if (start_out) *start_out = 0;
        if (aln) { *aln = (uint8_t *)malloc((size_t)qlen + 1); for (int i = 0; i < qlen; ++i) (*aln)[i] = LCDO_EDOP_INSERT; *aln_len = qlen; }
        return best;
    }
    /* start: reversed query vs the reversed prefix target[0 .. end0], anchored at its first character */
    const int plen = end0 + 1;
    for (int j = 0; j <= plen; ++j) prev[j] = j;
    for (int i = 1; i <= qlen; ++i) {
        cur[0] = i;
        for (int j = 1; j <= plen; ++j) {
            int d = prev[j - 1] + (query[qlen - i] != target[end0 - (j - 1)]), u = prev[j] + 1, l = cur[j - 1] + 1;
            cur[j] = d < u ? (d < l ? d : l) : (u < l ? u : l);
        }
        int *t = prev; prev = cur; cur = t;
    }
    int best2 = 1 << 30, last = -1;
    for (int j = 1; j <= plen; ++j) { if (prev[j] < best2) { best2 = prev[j]; last = j - 1; } else if (prev[j] == best2) last = j - 1; }
    free(prev); free(cur);
    if (best2 != best) return -1;
    const int start0 = end0 - last;
    if (start_out) *start_out = start0;
    if (end_out) *end_out = end0;
    if (aln) {
        opbuf_t out = {0, 0, 0};
        if (obtain_alignment(query, qlen, target + start0, end0 - start0 + 1, best, &out) < 0) { free(out.ops); return -1; }
        *aln = out.ops; *aln_len = out.n;
    }
    return best;
}

/* src/align.c:256-275 with edlibAlignmentToXID :164-187 */
int lcdo_edlib_infix_aln(const uint8_t *target, int tlen, const uint8_t *query, int qlen, int *n_eq, int *n_xid) {
    uint8_t *aln; int n, s0, e0;
    int d = lcdo_edlib_hw(query, qlen, target, tlen, &s0, &e0, &aln, &n);
    if (d < 0) return -1;
    if (n_eq && n_xid) {
        int eq = 0, x = 0;
        for (int i = 0; i < n; ++i) { if (aln[i] == LCDO_EDOP_MATCH) eq++; else x++; }
        *n_eq = eq; *n_xid = x;
    }
    free(aln);
    return d;
}
