/*
 * align_glue.c -- ORACLE (test infrastructure only; see lcd_oracle.h).
 *
 * Source-level restatement of the src/align.c glue around K1-K4 (all of that source IS present in
 * /root/reference, so this follows it statement by statement; "parity unpinned" only because the
 * reference binary cannot be built here to produce golden outputs -- htslib/abPOA/WFA2 headers absent).
 * Every function cites the lines it follows.
 */
#include <assert.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

/* ---------- src/seq.c:429-436 ---------- */
static double calc_read_error_rate(int len, const uint8_t *qual) {
    if (len <= 0 || qual == NULL) return 0.0;
    double expected_errors = 0.0;
    for (int i = 0; i < len; ++i) expected_errors += pow(10.0, -((double)qual[i]) / 10.0);
    return expected_errors / len;
}

/* ---------- src/align.c:945-952 ---------- */
static int full_cover_cmp(int c1, int c2) {
    if (c1 == c2) return 0;
    if (LCDO_IS_BOTH_COVER(c1)) return 1;
    else if (LCDO_IS_BOTH_COVER(c2)) return -1;
    if (LCDO_IS_LEFT_COVER(c1) && LCDO_IS_LEFT_COVER(c2)) return 0;
    if (LCDO_IS_RIGHT_COVER(c1) && LCDO_IS_RIGHT_COVER(c2)) return 0;
    return c1 - c2;
}

/* ---------- src/align.c:955-987 : exchange sort, exact swap sequence ---------- */
void lcdo_sort_noisy_region_reads(lcdo_region_reads_t *r, int use_err) {
    int n = r->n_reads;
    double *err = NULL;
    if (use_err) {
        err = (double *)malloc(n * sizeof(double));
        for (int i = 0; i < n; ++i) err[i] = calc_read_error_rate(r->lens[i], r->quals ? r->quals[i] : NULL);
    }
    for (int i = 0; i < n - 1; ++i) {
        for (int j = i + 1; j < n; ++j) {
            int cc = full_cover_cmp(r->fully_covers[i], r->fully_covers[j]);
            if (cc < 0 || (cc == 0 && use_err && err[i] > err[j]) ||
                (cc == 0 && ((use_err && err[i] == err[j]) || !use_err) && r->lens[i] < r->lens[j])) {
#define SWAP(T, a) do { T t_ = (a)[i]; (a)[i] = (a)[j]; (a)[j] = t_; } while (0)
                SWAP(int, r->fully_covers); SWAP(int, r->read_ids); SWAP(int, r->lens); SWAP(uint8_t *, r->seqs);
                if (r->quals) SWAP(uint8_t *, r->quals);
                SWAP(int, r->haps); SWAP(int64_t, r->phase_sets);
                if (use_err) SWAP(double, err);
#undef SWAP
            }
        }
    }
    free(err);
}

/* ---------- src/align.c:1225-1279 ---------- */
int64_t lcdo_collect_phase_set_with_both_haps(const lcdo_region_reads_t *r, int min_full, int min_all) {
    int n = r->n_reads, n_uniq = 0;
    int64_t *uniq = (int64_t *)calloc(n > 0 ? n : 1, sizeof(int64_t));
    int (*full)[2] = (int (*)[2])calloc(n > 0 ? n : 1, sizeof(int[2]));
    int (*all)[2] = (int (*)[2])calloc(n > 0 ? n : 1, sizeof(int[2]));
    int (*minlen)[2] = (int (*)[2])malloc((n > 0 ? n : 1) * sizeof(int[2]));
    for (int i = 0; i < n; ++i) minlen[i][0] = minlen[i][1] = INT32_MAX;
    for (int i = 0; i < n; ++i) {
        if (r->haps[i] == 0) continue;
        int k;
        for (k = 0; k < n_uniq; ++k) if (uniq[k] == r->phase_sets[i]) break; /* add_phase_set :989 */
        if (k == n_uniq) uniq[n_uniq++] = r->phase_sets[i];
        int h = r->haps[i] - 1;
        if (LCDO_IS_BOTH_COVER(r->fully_covers[i])) {
            full[k][h]++; all[k][h]++;
            if (minlen[k][h] > r->lens[i]) minlen[k][h] = r->lens[i];
        } else if (LCDO_IS_LEFT_COVER(r->fully_covers[i]) || LCDO_IS_RIGHT_COVER(r->fully_covers[i])) {
            if (r->lens[i] >= minlen[k][h]) all[k][h]++;
        }
    }
    int64_t max_ps = -1; int max_i = -1, m1 = -1, m2 = -1;
    for (int i = 0; i < n_uniq; ++i) {
        int c1 = full[i][0] < full[i][1] ? full[i][0] : full[i][1];
        int c2 = full[i][0] > full[i][1] ? full[i][0] : full[i][1];
        if (c1 > m1) { m1 = c1; m2 = c2; max_ps = uniq[i]; max_i = i; }
        else if (c1 == m1 && c2 > m2) { m2 = c2; max_ps = uniq[i]; max_i = i; }
    }
    if (m1 < min_full) max_ps = -1;
    if (max_ps != -1 && max_i != -1)
        if (all[max_i][0] < min_all || all[max_i][1] < min_all) max_ps = -1;
    free(uniq); free(full); free(all); free(minlen);
    return max_ps;
}

/* ---------- src/align.c:1000-1026 ---------- */
static int is_homopolymer(const uint8_t *seq, int seq_len, int flank_len) {
    if (seq_len < 2 * flank_len || seq_len > 2 * flank_len + 50) return 0;
    int min_hp_len = 5, hp_start = -1, hp_len = 0;
    for (int i = flank_len - 1; i < seq_len - flank_len + 1; ++i) {
        if (seq[i] == seq[i - 1]) {
            if (hp_start == -1) hp_start = i - 2;
            hp_len++;
        } else {
            if (hp_len >= min_hp_len) return 1;
            else { hp_start = -1; hp_len = 0; }
        }
    }
    if (hp_len >= min_hp_len) return 1;
    return 0;
}

/* ---------- src/align.c:496-562 ---------- */
void lcdo_wfa_trim_aln_str(int fc, lcdo_aln_str_t *s) {
    if (LCDO_IS_NOT_COVER(fc) || LCDO_IS_BOTH_COVER(fc)) return;
    if ((LCDO_IS_LEFT_COVER(fc) && LCDO_IS_RIGHT_GAP(fc)) || (LCDO_IS_RIGHT_COVER(fc) && LCDO_IS_LEFT_GAP(fc))) {
        s->target_beg = 0; s->target_end = s->aln_len - 1; s->query_beg = 0; s->query_end = s->aln_len - 1;
        return;
    }
    if (LCDO_IS_LEFT_COVER(fc)) {
        int target_end = -1, query_end = -1;
        for (int i = s->aln_len - 1; i >= 0; --i) {
            if (query_end == -1 && s->query_aln[i] != 5 && s->target_aln[i] == s->query_aln[i]) query_end = i;
            if (target_end == -1 && s->target_aln[i] != 5) target_end = i;
            if (target_end != -1 && query_end != -1) break;
        }
        if (query_end == -1) query_end = target_end;
        assert(query_end <= target_end);
        s->aln_len = target_end + 1;
        s->target_beg = 0; s->target_end = target_end; s->query_beg = 0; s->query_end = query_end;
        for (int i = query_end + 1; i < s->aln_len; ++i) s->query_aln[i] = 5;
    } else if (LCDO_IS_RIGHT_COVER(fc)) {
        int query_start = -1, target_start = -1;
        for (int i = 0; i < s->aln_len; ++i) {
            if (query_start == -1 && s->query_aln[i] != 5 && s->target_aln[i] == s->query_aln[i]) query_start = i;
            if (target_start == -1 && s->target_aln[i] != 5) target_start = i;
            if (target_start != -1 && query_start != -1) break;
        }
        if (query_start == -1) query_start = target_start;
        assert(query_start >= target_start);
        s->aln_len = s->aln_len - target_start;
        if (target_start != 0) {
            uint8_t *t = (uint8_t *)malloc((size_t)s->aln_len * 2 + 1);
            for (int i = 0; i < s->aln_len; ++i) t[i] = s->target_aln[i + target_start];
            for (int i = 0; i < s->aln_len; ++i) t[i + s->aln_len] = s->query_aln[i + target_start];
            free(s->target_aln);
            s->target_aln = t; s->query_aln = t + s->aln_len;
        }
        s->target_beg = 0; s->target_end = s->aln_len - 1;
        s->query_beg = query_start - target_start; s->query_end = s->aln_len - 1;
        for (int i = 0; i < s->query_beg; ++i) s->query_aln[i] = 5;
    }
}

/* ---------- src/align.c:630-663 ---------- */
static void collect_aln_beg_end(const uint32_t *cigar, int n, int ext, int ref_len, int *ref_beg, int *ref_end, int read_len,
                                int *read_beg, int *read_end) {
    *ref_beg = 1; *read_beg = 1; *ref_end = ref_len; *read_end = read_len;
    if (ext == LCDO_EXT_LEFT_TO_RIGHT) {
        int tr = 0, tq = 0;
        for (int i = 0; i < n; ++i) {
            int op = cigar[i] & 0xf, len = cigar[i] >> 4;
            if (op == LCDO_CEQUAL || op == LCDO_CMATCH) { tr += len; tq += len; *ref_end = tr; *read_end = tq; }
            else if (op == LCDO_CDIFF) { tr += len; tq += len; }
            else if (op == LCDO_CDEL) tr += len;
            else if (op == LCDO_CINS) tq += len;
        }
    } else {
        int tr = ref_len + 1, tq = read_len + 1;
        for (int i = n - 1; i >= 0; --i) {
            int op = cigar[i] & 0xf, len = cigar[i] >> 4;
            if (op == LCDO_CEQUAL || op == LCDO_CMATCH) { tr -= len; tq -= len; *ref_beg = tr; *read_beg = tq; }
            else if (op == LCDO_CDIFF) { tr -= len; tq -= len; }
            else if (op == LCDO_CDEL) tr -= len;
            else if (op == LCDO_CINS) tq -= len;
        }
    }
}

/* ---------- src/align.c:667-707 ---------- */
static int cal_wfa_partial_aln_beg_end(int ext, const lcdo_opt_t *opt, const uint8_t *_target, int _tlen,
                                       const uint8_t *_query, int _qlen, int *tb, int *te, int *qb, int *qe) {
    int gap_aln = opt->gap_aln;
    double ratio = opt->partial_aln_ratio;
    int tlen = _tlen, qlen = _qlen;
    const uint8_t *target = _target, *query = _query;
    if (ext == LCDO_EXT_LEFT_TO_RIGHT) {
        if (_tlen > _qlen * ratio) tlen = (int)(_qlen * ratio);
        else if (_qlen > _tlen * ratio) qlen = (int)(_tlen * ratio);
    } else {
        if (_tlen > _qlen * ratio) { target = _target + _tlen - (int)(_qlen * ratio); tlen = (int)(_qlen * ratio); }
        else if (_qlen > _tlen * ratio) { query = _query + _qlen - (int)(_tlen * ratio); qlen = (int)(_tlen * ratio); }
    }
    if (ext == LCDO_EXT_LEFT_TO_RIGHT) gap_aln = (gap_aln == LCDO_GAP_RIGHT_ALN) ? LCDO_GAP_LEFT_ALN : LCDO_GAP_RIGHT_ALN;
    int min_len = tlen < qlen ? tlen : qlen;
    if (ext == LCDO_EXT_LEFT_TO_RIGHT) {
        int x = lcdo_edlib_xgaps(target, min_len, query, min_len);
        if (x > min_len * 0.10) return 0;
    } else {
        int x = lcdo_edlib_xgaps(target + tlen - min_len, min_len, query + qlen - min_len, min_len);
        if (x > min_len * 0.10) return 0;
    }
    uint32_t *cigar = NULL; int ret = 1, n_cigar = 0;
    lcdo_wfa_end2end_aln(target, tlen, query, qlen, gap_aln, opt->mismatch, opt->gap_open1, opt->gap_ext1, opt->gap_open2,
                         opt->gap_ext2, &cigar, &n_cigar, NULL, NULL, NULL, NULL);
    if (n_cigar == 0) ret = 0;
    else collect_aln_beg_end(cigar, n_cigar, ext, _tlen, tb, te, _qlen, qb, qe);
    free(cigar);
    return ret;
}

/* ---------- src/align.c:709-745 ---------- */
int lcdo_collect_partial_aln_beg_end(const lcdo_opt_t *opt, int sampling_reads, const uint8_t *target, int tlen, int tfc,
                                     const uint8_t *query, int qlen, int qfc, int *tb, int *te, int *qb, int *qe) {
    *tb = 1; *te = tlen; *qb = 1; *qe = qlen;
    int ret = 1;
    assert(LCDO_IS_BOTH_COVER(tfc) != 0);
    if (LCDO_IS_BOTH_COVER(qfc) || (LCDO_IS_LEFT_COVER(qfc) && LCDO_IS_RIGHT_GAP(qfc)) ||
        (LCDO_IS_RIGHT_COVER(qfc) && LCDO_IS_LEFT_GAP(qfc))) {
        if (sampling_reads) {
            int x = lcdo_edlib_xgaps(target, tlen, query, qlen);
            if (x > (tlen < qlen ? tlen : qlen) * 0.10) return 0;
        }
        return 1;
    } else {
        if (LCDO_IS_LEFT_COVER(qfc)) ret = cal_wfa_partial_aln_beg_end(LCDO_EXT_LEFT_TO_RIGHT, opt, target, tlen, query, qlen, tb, te, qb, qe);
        else if (LCDO_IS_RIGHT_COVER(qfc)) ret = cal_wfa_partial_aln_beg_end(LCDO_EXT_RIGHT_TO_LEFT, opt, target, tlen, query, qlen, tb, te, qb, qe);
    }
    return ret;
}

/* ---------- src/align.c:565-572 (only the BOTH_COVER branch is reachable, SURVEY 2.1 K3') ---------- */
static void wfa_collect_aln_str_both(const lcdo_opt_t *opt, const uint8_t *target, int tlen, const uint8_t *query, int qlen,
                                     lcdo_aln_str_t *s) {
    s->target_aln = 0; s->query_aln = 0; s->aln_len = 0;
    lcdo_wfa_end2end_aln(target, tlen, query, qlen, opt->gap_aln, opt->mismatch, opt->gap_open1, opt->gap_ext1,
                         opt->gap_open2, opt->gap_ext2, NULL, NULL, &s->target_aln, &s->query_aln, &s->aln_len, NULL);
    s->target_beg = 0; s->target_end = s->aln_len - 1; s->query_beg = 0; s->query_end = s->aln_len - 1;
}

/* ---------- src/align.c:1029-1054 ---------- */
static int make_cons_read_aln_str(const uint8_t *cons_str, const uint8_t *read_str, int msa_len, int full_cover, lcdo_aln_str_t *s) {
    int aln_len = 0;
    s->target_aln = (uint8_t *)malloc((size_t)msa_len * 2 + 1);
    s->query_aln = s->target_aln + msa_len;
    for (int i = 0; i < msa_len; ++i)
        if (read_str[i] != 5 || cons_str[i] != 5) { s->target_aln[aln_len] = cons_str[i]; s->query_aln[aln_len] = read_str[i]; aln_len++; }
    s->aln_len = aln_len;
    s->target_beg = 0; s->target_end = aln_len - 1; s->query_beg = 0; s->query_end = aln_len - 1;
    lcdo_wfa_trim_aln_str(full_cover, s);
    return aln_len;
}

/* ---------- src/align.c:1056-1146 ---------- */
static int make_ref_read_aln_str(const lcdo_opt_t *opt, const lcdo_aln_str_t *rc, const lcdo_aln_str_t *cr, lcdo_aln_str_t *rr) {
    int aln_len = 0, max_len = rc->aln_len + cr->aln_len;
    rr->target_aln = (uint8_t *)malloc((size_t)max_len * 2 + 1);
    rr->query_aln = rr->target_aln + max_len;
    int i = 0, j = 0;
    while (i < rc->aln_len && j < cr->aln_len) {
        if (rc->query_aln[i] == 5 && cr->target_aln[j] == 5) {
            int rd = 1, qd = 1;
            while (i + rd < rc->aln_len && rc->query_aln[i + rd] == 5) rd++;
            while (j + qd < cr->aln_len && cr->target_aln[j + qd] == 5) qd++;
            uint8_t *ra = 0, *qa = 0; int dl = 0;
            lcdo_wfa_end2end_aln(rc->target_aln + i, rd, cr->query_aln + j, qd, opt->gap_aln, opt->mismatch, opt->gap_open1,
                                 opt->gap_ext1, opt->gap_open2, opt->gap_ext2, NULL, NULL, &ra, &qa, &dl, NULL);
            for (int k = 0; k < dl; ++k) { rr->target_aln[aln_len] = ra[k]; rr->query_aln[aln_len] = qa[k]; aln_len++; }
            i += rd; j += qd;
            free(ra);
        } else if (rc->query_aln[i] != 5 && cr->target_aln[j] != 5) {
            rr->target_aln[aln_len] = rc->target_aln[i]; rr->query_aln[aln_len] = cr->query_aln[j]; aln_len++; i++; j++;
        } else if (rc->query_aln[i] == 5) {
            rr->target_aln[aln_len] = rc->target_aln[i]; rr->query_aln[aln_len] = 5; aln_len++; i++;
        } else {
            rr->target_aln[aln_len] = 5; rr->query_aln[aln_len] = cr->query_aln[j]; aln_len++; j++;
        }
    }
    while (i < rc->aln_len) { rr->target_aln[aln_len] = rc->target_aln[i]; rr->query_aln[aln_len] = 5; aln_len++; i++; }
    while (j < cr->aln_len) { rr->target_aln[aln_len] = 5; rr->query_aln[aln_len] = cr->query_aln[j]; aln_len++; j++; }
    rr->aln_len = aln_len;
    rr->target_beg = rr->target_end = rr->query_beg = rr->query_end = -1;
    return aln_len;
}

/* ---------- src/align.c:1286-1375 ---------- */
static int with_ps_hap(const lcdo_opt_t *opt, int sampling_reads, const lcdo_region_reads_t *r, int64_t ps, const uint8_t *ref_seq,
                       int ref_len, int *clu_n_seqs, int **clu_read_ids, lcdo_aln_str_t **aln_strs) {
    int n = r->n_reads, n_cons = 0;
    int *ids = (int *)malloc((n + 2) * sizeof(int)), *lens = (int *)malloc((n + 2) * sizeof(int)),
        *fcs = (int *)calloc(n + 2, sizeof(int));
    uint8_t **seqs = (uint8_t **)malloc((n + 2) * sizeof(uint8_t *));
    lcdo_poa_result_t res[2]; memset(res, 0, sizeof(res));
    int have[2] = {0, 0};
    int use_non_full = !is_homopolymer(ref_seq, ref_len, opt->noisy_reg_flank_len);
    lens[0] = 0;
    for (int hap = 1; hap <= 2; ++hap) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            if (r->lens[i] <= 0 || r->phase_sets[i] != ps || r->haps[i] != hap) continue;
            if (use_non_full == 0 && LCDO_IS_BOTH_COVER(r->fully_covers[i]) == 0) continue;
            ids[m] = r->read_ids[i]; lens[m] = r->lens[i]; seqs[m] = r->seqs[i]; fcs[m] = r->fully_covers[i]; m++;
        }
        if (lens[0] >= opt->max_noisy_reg_len) break;
        if (m == 0) continue;
        n_cons += lcdo_poa_partial_aln_msa_cons(opt, sampling_reads, m, seqs, lens, fcs, &res[hap - 1]);
        have[hap - 1] = 1;
        clu_n_seqs[hap - 1] = m; /* src/align.c:832-834: single consensus => all input reads, skipped ones included */
        clu_read_ids[hap - 1] = (int *)malloc(m * sizeof(int));
        for (int i = 0; i < m; ++i) clu_read_ids[hap - 1][i] = ids[i];
    }
    if (n_cons != 2) n_cons = 0;
    else {
        for (int hap = 1; hap <= 2; ++hap) {
            lcdo_aln_str_t *cs = aln_strs[hap - 1];
            lcdo_poa_result_t *R = &res[hap - 1];
            wfa_collect_aln_str_both(opt, ref_seq, ref_len, R->cons_seq[0], R->cons_len[0], &cs[0]);
            int k = 0;
            for (int i = 0; i < n; ++i) {
                if (r->lens[i] <= 0 || r->phase_sets[i] != ps || r->haps[i] != hap) continue;
                if (use_non_full == 0 && LCDO_IS_BOTH_COVER(r->fully_covers[i]) == 0) continue;
                make_cons_read_aln_str(R->msa[clu_n_seqs[hap - 1]], R->msa[k], R->msa_len, r->fully_covers[i], &cs[2 * k + 1]);
                if (opt->collect_ref_read_aln_str) make_ref_read_aln_str(opt, &cs[0], &cs[2 * k + 1], &cs[2 * k + 2]);
                k++;
            }
        }
    }
    for (int h = 0; h < 2; ++h) if (have[h]) lcdo_poa_result_free(&res[h]);
    free(ids); free(lens); free(fcs); free(seqs);
    return n_cons;
}

/* ---------- src/align.c:1148-1211 ---------- */
static int no_ps_hap(const lcdo_opt_t *opt, const lcdo_region_reads_t *r, const uint8_t *ref_seq, int ref_len, int *clu_n_seqs,
                     int **clu_read_ids, lcdo_aln_str_t **aln_strs) {
    int n = r->n_reads, nf = 0, n_cons = 0;
    int *fids = (int *)malloc((n + 2) * sizeof(int)), *flens = (int *)malloc((n + 2) * sizeof(int));
    uint8_t **fseqs = (uint8_t **)malloc((n + 2) * sizeof(uint8_t *));
    for (int i = 0; i < n; ++i) {
        if (r->lens[i] <= 0 || LCDO_IS_BOTH_COVER(r->fully_covers[i]) == 0) continue;
        fids[nf] = i; flens[nf] = r->lens[i]; fseqs[nf] = r->seqs[i]; nf++;
    }
    if (nf == 0 || flens[0] >= opt->max_noisy_reg_len) { free(fids); free(flens); free(fseqs); return 0; }
    lcdo_poa_result_t R;
    n_cons = lcdo_poa_aln_msa_cons(opt, nf, fseqs, flens, 2, &R);
    /* src/align.c:907-920 */
    if (n_cons == 2) {
        for (int c = 0; c < 2; ++c) {
            clu_n_seqs[c] = R.clu_n_seq[c];
            clu_read_ids[c] = (int *)malloc(R.clu_n_seq[c] * sizeof(int));
            for (int j = 0; j < R.clu_n_seq[c]; ++j) clu_read_ids[c][j] = fids[R.clu_read_ids[c][j]];
        }
    } else {
        clu_n_seqs[0] = nf;
        clu_read_ids[0] = (int *)malloc(nf * sizeof(int));
        for (int i = 0; i < nf; ++i) clu_read_ids[0][i] = fids[i];
    }
    for (int c = 0; c < n_cons; ++c) {
        lcdo_aln_str_t *cs = aln_strs[c];
        wfa_collect_aln_str_both(opt, ref_seq, ref_len, R.cons_seq[c], R.cons_len[c], &cs[0]);
        int k = 0;
        for (int j = 0; j < clu_n_seqs[c]; ++j) {
            int read_i = clu_read_ids[c][j];
            clu_read_ids[c][j] = r->read_ids[read_i];
            /* msa row of the j-th member of cluster c (src/align.c:929-934); fully_covers[c] quirk (:1194) */
            const uint8_t *row = R.msa[R.clu_read_ids[c][j]];
            make_cons_read_aln_str(R.msa[R.n_seq + c], row, R.msa_len, r->fully_covers[c], &cs[2 * k + 1]);
            if (opt->collect_ref_read_aln_str) make_ref_read_aln_str(opt, &cs[0], &cs[2 * k + 1], &cs[2 * k + 2]);
            k++;
        }
    }
    lcdo_poa_result_free(&R);
    free(fids); free(flens); free(fseqs);
    return n_cons;
}

/* ---------- src/align.c:1760-1813 (after collect_noisy_read_info; digar rewrite :1803 not on the BASELINE path) ---------- */
int lcdo_collect_noisy_reg_aln_strs(const lcdo_opt_t *opt, int64_t reg_len, lcdo_region_reads_t *reads, const uint8_t *ref_seq,
                                    int ref_seq_len, int *clu_n_seqs, int **clu_read_ids, lcdo_aln_str_t **aln_strs) {
    if (reads->n_reads <= 0) return 0;
    int sampling_reads = reg_len >= opt->min_noisy_reg_size_to_sample_reads;
    lcdo_sort_noisy_region_reads(reads, sampling_reads);
    int64_t ps = lcdo_collect_phase_set_with_both_haps(reads, opt->min_hap_full_reads, opt->min_hap_reads);
    int n_full = 0;
    for (int i = 0; i < reads->n_reads; ++i) if (LCDO_IS_BOTH_COVER(reads->fully_covers[i])) n_full++;
    int n_cons = 0;
    if (ps > 0) n_cons = with_ps_hap(opt, sampling_reads, reads, ps, ref_seq, ref_seq_len, clu_n_seqs, clu_read_ids, aln_strs);
    else if (n_full >= opt->min_dp) n_cons = no_ps_hap(opt, reads, ref_seq, ref_seq_len, clu_n_seqs, clu_read_ids, aln_strs);
    return n_cons;
}

/* ---------- src/align.c:1392-1458, one read ---------- */
void lcdo_read_region_slice(const lcdo_digar1_t *digars, int n_digar, int qlen, int64_t reg_beg, int64_t reg_end, int flank,
                            int *rb, int *re, int *cover_out) {
    int64_t reg_digar_beg = -1, reg_digar_end = -1;
    int reg_read_beg = 0, reg_read_end = qlen - 1; /* digar2qlen()-1 */
    if (digars[0].type == LCDO_CHARD_CLIP) reg_read_beg = digars[0].len;
    if (digars[n_digar - 1].type == LCDO_CHARD_CLIP) reg_read_end = digars[n_digar - 1].qi - 1;
    int beg_is_del = 0, end_is_del = 0, cover = 0;
    for (int d = 0; d < n_digar; ++d) {
        int64_t digar_beg = digars[d].pos, digar_end;
        int op = digars[d].type, len = digars[d].len, qi = digars[d].qi;
        if (op == LCDO_CSOFT_CLIP || op == LCDO_CHARD_CLIP) continue;
        if (op == LCDO_CDIFF || op == LCDO_CEQUAL || op == LCDO_CDEL) digar_end = digar_beg + len - 1;
        else digar_end = digar_beg;
        if (digar_beg > reg_end) break;
        if (digar_end < reg_beg) continue;
        if (digar_beg <= reg_beg && digar_end >= reg_beg) {
            if (op == LCDO_CDEL) { reg_digar_beg = reg_beg; reg_read_beg = qi; if (len > flank) beg_is_del = 1; }
            else { reg_digar_beg = reg_beg; reg_read_beg = qi + (int)(reg_beg - digar_beg); }
        }
        if (digar_beg <= reg_end && digar_end >= reg_end) {
            if (op == LCDO_CDEL) { reg_digar_end = reg_end; reg_read_end = qi - 1; if (len > flank) end_is_del = 1; }
            else { reg_digar_end = reg_end; reg_read_end = qi + (int)(reg_end - digar_beg); }
        }
    }
    if (reg_digar_beg == reg_beg && reg_digar_end == reg_end) {
        if (!beg_is_del && !end_is_del) cover = LCDO_LEFT_COVER | LCDO_RIGHT_COVER;
        else if (!beg_is_del && end_is_del) cover = LCDO_LEFT_COVER | LCDO_RIGHT_GAP;
        else if (beg_is_del && !end_is_del) cover = LCDO_LEFT_GAP | LCDO_RIGHT_COVER;
        else cover = LCDO_LEFT_GAP | LCDO_RIGHT_GAP;
    } else if (reg_digar_beg == reg_beg) cover = beg_is_del ? LCDO_LEFT_GAP : LCDO_LEFT_COVER;
    else if (reg_digar_end == reg_end) cover = end_is_del ? LCDO_RIGHT_GAP : LCDO_RIGHT_COVER;
    else cover = 0;
    *rb = reg_read_beg; *re = reg_read_end; *cover_out = cover;
}
