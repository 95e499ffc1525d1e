/* oracle/digar.c -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
 *
 * SURVEY 8(f) row f2, first part: one read's EQX CIGAR -> digar list + the read's noisy windows.  Restates
 *   collect_digar_from_eqx_cigar     src/bam_utils.c:701-842
 *   push_xid_size_queue_win          src/bam_utils.c:161-200  (xid_queue_t :123-159)
 *   collect_noisy_region_len         src/bam_utils.c:631-638
 *   is_overlap_reg                   src/bam_utils.h:150-153
 *   cr_index ordering of the read's intervals (src/cgranges.c; restated as lcdo_cr_sorted_order in assign_hap.c)
 * Not restated: is_ont_palindrome_clip (reads the SA tag through htslib) -- its result comes in as two flags; the copies of the read
 * buffers and the base-quality histogram (longcalld_copy_digar_read_buffers, :96-103), which are I/O bookkeeping.
 * Parity: UNPINNED (no golden vectors for this path in the reference; its binary cannot be built here).  The bundled test_data BAM is
 * decoded by tests/golden/make_testdata_fixture.py and its CIGARs are the real-input case of tests/test_gpu_digar.py.
 */
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

#include "digar_priv.h"

int lcdo_collect_digar_from_eqx_cigar(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const uint8_t *bseq,
                                      const uint8_t *qual, int qlen, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome,
                                      int right_clip_is_palindrome, lcdo_digar_t **digars_out, int *n_digar_out, int64_t **noisy_out, int *n_noisy_out,
                                      int64_t **chunk_noisy_out, int *n_chunk_noisy_out, int64_t *beg_out, int64_t *end_out, int *n_cand_out) {
    (void)bseq; /* the digar alt_seq copies are not part of this interface: bases stay in the packed read */
    int64_t pos = read_pos0 + 1; int qi = 0;
    int rlen = 0, cap = 16, nd = 0, n_cand = 0;
    for (int i = 0; i < n_cigar; ++i) { const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4); if (op == CMATCH || op == CDEL || op == CREF_SKIP || op == CEQUAL || op == CDIFF) rlen += len; }
    lcdo_digar_t *d = (lcdo_digar_t *)malloc(sizeof(lcdo_digar_t) * cap);
#define PUSH_D(P, T, L, Q, LQ) do { if (nd == cap) { cap *= 2; d = (lcdo_digar_t *)realloc(d, sizeof(lcdo_digar_t) * cap); } \
        d[nd].pos = (P); d[nd].type = (T); d[nd].len = (L); d[nd].qi = (Q); d[nd].is_low_qual = (LQ); ++nd; } while (0)
    xidq_t q; q.cap = rlen > 16 ? rlen : 16; q.pos = (int64_t *)malloc(sizeof(int64_t) * q.cap); q.lens = (int *)malloc(sizeof(int) * q.cap);
    q.counts = (int *)malloc(sizeof(int) * q.cap); q.front = 0; q.rear = -1; q.count = 0; q.max_s = opt->noisy_reg_max_xgaps; q.win = opt->noisy_reg_slide_win;
    ivlist_t cr = {NULL, 0, 0};
    int64_t cur_s = -1, cur_e = -1; int q_s = -1, q_e = -1;
    int bad = 0;
    for (int i = 0; i < n_cigar && !bad; ++i) {
        const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4);
        if (op == CDIFF) {
            for (int j = 0; j < len; ++j) {
                if (qual[qi] >= opt->min_bq) { q_push(&q, pos, 1, 1, &cr, &cur_s, &cur_e, &q_s, &q_e); PUSH_D(pos, op, 1, qi, 0); }
                else PUSH_D(pos, op, 1, qi, 1);
                ++n_cand; ++pos; ++qi;
            }
        } else if (op == CEQUAL) { PUSH_D(pos, op, len, qi, 0); pos += len; qi += len; }
        else if (op == CDEL) {
            const int qr = qi < qlen ? qi : qlen - 1; /* (a deletion is never the last operation of an alignment; guard only) */
            if ((qi == 0 || qual[qi - 1] >= opt->min_bq) && qual[qr] >= opt->min_bq) { q_push(&q, pos, len, len, &cr, &cur_s, &cur_e, &q_s, &q_e); PUSH_D(pos, op, len, qi, 0); }
            else PUSH_D(pos, op, len, qi, 1);
            ++n_cand; pos += len;
        } else if (op == CINS) {
            int low = 1;
            for (int k = 0; k < len; ++k) if (qual[qi + k] >= opt->min_bq) { low = 0; break; }
            if (!low) q_push(&q, pos, 0, len, &cr, &cur_s, &cur_e, &q_s, &q_e);
            PUSH_D(pos, op, len, qi, low);
            ++n_cand; qi += len;
        } else if (op == CSOFT || op == CHARD) {
            const int pal = (i == 0 && left_clip_is_palindrome) || (i != 0 && right_clip_is_palindrome);
            PUSH_D(pos, pal ? CHARD : op, len, qi, 0);
            if ((i == 0 && pos > 10) || (i != 0 && pos < whole_ref_len - 10)) {
                if (len > opt->end_clip_reg) {
                    if (i == 0 && !left_clip_is_palindrome) { if (pos > 1) iv_add(&cr, pos - 1, pos + opt->end_clip_reg_flank_win, 0); ++n_cand; }
                    else if (i != 0 && !right_clip_is_palindrome) { if (pos < whole_ref_len) iv_add(&cr, pos - 1 - opt->end_clip_reg_flank_win, pos, 0); ++n_cand; }
                }
            }
            if (op == CSOFT) qi += len;
        } else if (op == CREF_SKIP) pos += len;
        else bad = 1; /* 'M' in an EQX CIGAR: the reference exits */
    }
    if (cur_s != -1) {
        int vs = 0;
        for (int i = q_s; i <= q_e; ++i) vs += q.counts[i];
        if (vs < (int)(cur_e - cur_s + 1)) vs = (int)(cur_e - cur_s + 1);
        iv_add(&cr, cur_s - 1, cur_e, vs);
    }
    /* cr_index: as added when already sorted by start, otherwise cgranges' radix sort */
    int *order = (int *)malloc(sizeof(int) * (cr.n + 1)), *st = (int *)malloc(sizeof(int) * (cr.n + 1)), *en = (int *)malloc(sizeof(int) * (cr.n + 1));
    for (int i = 0; i < cr.n; ++i) { st[i] = (int)cr.v[i].st; en[i] = (int)cr.v[i].en; }
    lcdo_cr_sorted_order(cr.n, st, en, order);
    int64_t *noisy = (int64_t *)malloc(sizeof(int64_t) * 3 * (cr.n + 1)), *cn = (int64_t *)malloc(sizeof(int64_t) * 3 * (cr.n + 1));
    int total = 0, ncn = 0;
    for (int i = 0; i < cr.n; ++i) { const iv_t *v = &cr.v[order[i]]; noisy[3 * i] = v->st; noisy[3 * i + 1] = v->en; noisy[3 * i + 2] = v->label; total += (int)(v->en - v->st + 1); }
    const int64_t beg = read_pos0 + 1, end = read_pos0 + rlen; /* digar->beg, digar->end = bam_endpos */
    const int mapped = (int)(end - beg + 1);
    int skip = 0;
    if (total > mapped * opt->max_noisy_frac_per_read || n_cand > mapped * opt->max_var_ratio_per_read) skip = 1;
    else
        for (int i = 0; i < cr.n; ++i)
            if (!(noisy[3 * i] + 1 > reg_end || noisy[3 * i + 1] < reg_beg)) { cn[3 * ncn] = noisy[3 * i]; cn[3 * ncn + 1] = noisy[3 * i + 1]; cn[3 * ncn + 2] = noisy[3 * i + 2]; ++ncn; }
    *digars_out = d; *n_digar_out = nd; *noisy_out = noisy; *n_noisy_out = cr.n; *chunk_noisy_out = cn; *n_chunk_noisy_out = ncn;
    *beg_out = beg; *end_out = end; *n_cand_out = n_cand;
    free(order); free(st); free(en); free(cr.v); free(q.pos); free(q.lens); free(q.counts);
    if (bad) return -2;
    return skip ? -1 : 0;
}
