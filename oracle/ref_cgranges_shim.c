/*
 * ref_cgranges_shim.c -- thin C entry points over the REFERENCE's own interval index (src/cgranges.c, present in the reference tree).
 *
 * Built only where /root/reference exists (this container), by oracle/Makefile, together with /root/reference/src/cgranges.c compiled from
 * where it lies, into oracle/_ref/libcgranges_ref.so (git-ignored, travels with gpurun).  No reference source is copied into this repo.
 * It pins the two places where results of the hot path depend on cgranges' behaviour: the order in which cr_index() leaves the intervals
 * (K5 seeds reads in that order, src/assign_hap.c:512-526; a read's noisy windows are reported in it, src/bam_utils.c:806) and the order
 * in which cr_overlap() reports hits (K5, src/assign_hap.c:345-422).
 */
#include <stdlib.h>
#include "cgranges.h"

/* labels (= insertion indices) of the n intervals in the order cr_index() leaves them */
void ref_cr_sorted_labels(int n, const int *st, const int *en, int *labels_out) {
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n; ++i) cr_add(cr, "cr", st[i], en[i], i);
    cr_index(cr);
    for (int64_t i = 0; i < cr->n_r; ++i) labels_out[i] = cr_label(cr, i);
    cr_destroy(cr);
}

/* labels of the intervals overlapping [qst, qen), in the order cr_overlap() returns them; returns their number */
int ref_cr_overlap_labels(int n, const int *st, const int *en, int qst, int qen, int *labels_out, int cap) {
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n; ++i) cr_add(cr, "cr", st[i], en[i], i);
    cr_index(cr);
    int64_t *b = 0, m = 0;
    const int64_t k = cr_overlap(cr, "cr", qst, qen, &b, &m);
    for (int64_t i = 0; i < k && i < cap; ++i) labels_out[i] = cr_label(cr, b[i]);
    free(b);
    cr_destroy(cr);
    return (int)k;
}

/* pre_process_noisy_regs (src/collect_var.c:557-638) with cr_extend_noisy_regs_with_low_comp (:538-553) and low_comp_cr_start_end (:466-478):
 * the control flow is restated here, every interval operation (cr_add / cr_index / cr_merge / cr_overlap) is the reference's own code.
 * in:  chunk_noisy (st, en, label) triples in cr_add order; low_comp (st, en) pairs (chunk->low_comp_cr, may be empty); per read (in
 *      ordered_read_ids order, skipped reads left out by the caller): digar beg / end and its own noisy intervals (CSR, st/en pairs).
 * out: the surviving regions (st, en, label) in index order; returns their number. */
int ref_pre_process_noisy_regs(int n_noisy, const int *noisy, int n_low, const int *low_comp, int n_reads, const long long *read_beg,
                               const long long *read_end, const int *read_iv_off, const int *read_iv, int noisy_reg_merge_dis, int min_sv_len,
                               int min_alt_dp, float min_af, int *out, int out_cap) {
    if (n_noisy == 0) return 0;
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n_noisy; ++i) cr_add(cr, "cr", noisy[3 * i], noisy[3 * i + 1], noisy[3 * i + 2]);
    cr_index(cr);
    if (n_low > 0) { /* cr_extend_noisy_regs_with_low_comp */
        cgranges_t *lc = cr_init();
        for (int i = 0; i < n_low; ++i) cr_add(lc, "cr", low_comp[2 * i], low_comp[2 * i + 1], 0);
        cr_index(lc);
        cgranges_t *nw = cr_init();
        for (int64_t i = 0; i < cr->n_r; ++i) {
            const int32_t start = cr_start(cr, i) + 1, end = cr_end(cr, i);
            int32_t ns = start, ne = end;
            int64_t *b = 0, m = 0;
            const int64_t k = cr_overlap(lc, "cr", start - 1, end, &b, &m);
            for (int64_t j = 0; j < k; ++j) { const int32_t s = cr_start(lc, b[j]) + 1, e = cr_end(lc, b[j]); if (s < ns) ns = s; if (e > ne) ne = e; }
            free(b);
            cr_add(nw, "cr", ns - 1, ne, cr_label(cr, i));
        }
        cr_index(nw); cr_destroy(cr); cr = nw; cr_destroy(lc);
    }
    cr = cr_merge(cr, -1, noisy_reg_merge_dis, min_sv_len); /* inside cr_extend_noisy_regs_with_low_comp (:552) */
    cr = cr_merge(cr, -1, noisy_reg_merge_dis, min_sv_len); /* and again in pre_process_noisy_regs (:568) */
    const int64_t nr = cr->n_r;
    int *tot = (int *)calloc(nr + 1, sizeof(int)), *nz = (int *)calloc(nr + 1, sizeof(int));
    int64_t *ob = 0, mb = 0;
    for (int r = 0; r < n_reads; ++r) {
        const int64_t on = cr_overlap(cr, "cr", (int32_t)(read_beg[r] - 1), (int32_t)read_end[r], &ob, &mb);
        cgranges_t *rc = cr_init();
        for (int k = read_iv_off[r]; k < read_iv_off[r + 1]; ++k) cr_add(rc, "cr", read_iv[2 * k], read_iv[2 * k + 1], 0);
        cr_index(rc);
        for (int64_t q = 0; q < on; ++q) {
            const int ri = (int)ob[q];
            tot[ri]++;
            const int rs = cr_start(cr, ri) + 1, re = cr_end(cr, ri);
            int64_t *nb = 0, nm = 0;
            if (rc->n_r > 0 && cr_overlap(rc, "cr", rs - 1, re, &nb, &nm) > 0) nz[ri]++;
            free(nb);
        }
        cr_destroy(rc);
    }
    free(ob);
    int n_out = 0;
    for (int64_t i = 0; i < nr; ++i) {
        if (nz[i] < min_alt_dp || (float)nz[i] / tot[i] < min_af) continue;
        if (n_out < out_cap) { out[3 * n_out] = cr_start(cr, i); out[3 * n_out + 1] = cr_end(cr, i); out[3 * n_out + 2] = cr_label(cr, i); }
        ++n_out;
    }
    free(tot); free(nz); cr_destroy(cr);
    return n_out;
}

/* the reference's own symmetric-DUST (src/sdust.c, compiled from where it lies): low-complexity intervals of a sequence, as chunk->low_comp_cr
 * is filled (src/bam_utils.c:1573-1581).  out: (start, finish) pairs, 0-based half-open as sdust returns them; returns their number. */
#include "sdust.h"
int ref_sdust(const unsigned char *seq, int len, int T, int W, int *out, int cap) {
    int n = 0;
    uint64_t *r = sdust(0, seq, len, T, W, &n);
    for (int i = 0; i < n && i < cap; ++i) { out[2 * i] = (int)(r[i] >> 32); out[2 * i + 1] = (int)(uint32_t)r[i]; }
    free(r);
    return n;
}

/* post_process_noisy_regs (src/collect_var.c:640-660) with collect_noisy_reg_start_end (:481-536): regions grown to flanks free of candidate
 * variants, then merged (cr_merge(cr, 0, -1, -1)).  Control flow restated, interval operations by the reference's cgranges.
 * in: regions (st, en, label) in index order; candidate variants (pos, ref_len, cate) in chunk order.  out: (st, en, label); returns the number. */
int ref_post_process_noisy_regs(int n_regs, const int *regs, int n_vars, const int *var_pos, const int *var_ref_len, const int *var_cate, int flank_len,
                                int *out, int out_cap) {
    const int NOT_CAND = 0x800 | 0x001 | 0x002;
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n_regs; ++i) cr_add(cr, "cr", regs[3 * i], regs[3 * i + 1], regs[3 * i + 2]);
    cr_index(cr);
    const int n = (int)cr->n_r;
    int *maxl = (int *)malloc(sizeof(int) * (n + 1)), *minr = (int *)malloc(sizeof(int) * (n + 1)), *st = (int *)malloc(sizeof(int) * (n + 1)), *en = (int *)malloc(sizeof(int) * (n + 1));
    for (int i = 0; i < n; ++i) maxl[i] = minr[i] = -1;
    for (int ri = 0, vi = 0; ri < n && vi < n_vars;) {
        if (var_cate[vi] & NOT_CAND) { vi++; continue; }
        const int vs = var_pos[vi], ve = var_pos[vi] + var_ref_len[vi] - 1;
        const int rs = cr_start(cr, ri) + 1, re = cr_end(cr, ri);
        if (vs > re) { if (minr[ri] == -1) minr[ri] = vi; ri++; }
        else if (ve < rs) { maxl[ri] = vi; vi++; }
        else vi++;
    }
    for (int ri = 0; ri < n; ++ri) {
        if (maxl[ri] == -1) maxl[ri] = n_vars - 1 < 0 ? n_vars - 1 : 0;
        if (minr[ri] == -1) minr[ri] = 0 > n_vars - 1 ? 0 : n_vars - 1;
        const int os = cr_start(cr, ri) + 1, oe = cr_end(cr, ri);
        int cs = os - flank_len, ce = oe + flank_len;
        for (int vi = maxl[ri]; vi >= 0; --vi) {
            if (var_cate[vi] & NOT_CAND) continue;
            const int vs = var_pos[vi], ve = var_pos[vi] + var_ref_len[vi] - 1;
            if (ve < cs - 1) break; else if (vs - flank_len < cs) cs = vs - flank_len;
        }
        for (int vi = minr[ri]; vi < n_vars; ++vi) {
            if (var_cate[vi] & NOT_CAND) continue;
            const int vs = var_pos[vi], ve = var_pos[vi] + var_ref_len[vi] - 1;
            if (vs > ce + 1) break; else if (ve + flank_len > ce) ce = ve + flank_len;
        }
        st[ri] = cs; en[ri] = ce;
    }
    cgranges_t *nw = cr_init();
    for (int ri = 0; ri < n; ++ri) cr_add(nw, "cr", st[ri], en[ri], cr_label(cr, ri));
    cr_index(nw); cr_destroy(cr);
    nw = cr_merge(nw, 0, -1, -1);
    int m = 0;
    for (int64_t i = 0; i < nw->n_r; ++i, ++m) if (m < out_cap) { out[3 * m] = cr_start(nw, i); out[3 * m + 1] = cr_end(nw, i); out[3 * m + 2] = cr_label(nw, i); }
    cr_destroy(nw); free(maxl); free(minr); free(st); free(en);
    return m;
}

/* cr_merge(cr, fixed_merge_win, dynamic_merge_win, dynamic_merge_label_min) (src/cgranges.c:289) on n labelled intervals: the merged intervals (st, en, label) in
 * index order; returns their number.  (collect_var.c calls it with (-1, noisy_reg_merge_dis, min_sv_len) and with (0, -1, -1).) */
int ref_cr_merge(int n, const int *st, const int *en, const int *label, int fixed_win, int dyn_win, int dyn_label_min, int *out, int cap) {
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n; ++i) cr_add(cr, "cr", st[i], en[i], label[i]);
    cr_index(cr);
    cr = cr_merge(cr, fixed_win, dyn_win, dyn_label_min);
    int k = 0;
    for (int64_t i = 0; i < cr->n_r && k < cap; ++i, ++k) { out[3 * k] = cr_start(cr, i); out[3 * k + 1] = cr_end(cr, i); out[3 * k + 2] = cr_label(cr, i); }
    const int total = (int)cr->n_r;
    cr_destroy(cr);
    return total;
}
