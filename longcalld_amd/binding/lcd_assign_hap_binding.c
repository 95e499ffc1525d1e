/*
 * lcd_assign_hap_binding.c -- the reference-side binding of K5: what a longcallD maintainer adds to src/assign_hap.c so that
 *
 *     int assign_hap_based_on_germline_het_vars_kmeans(const call_var_opt_t *opt, bam_chunk_t *chunk, int target_var_cate)
 *
 * (src/assign_hap.h:12, definition src/assign_hap.c:473-547; callers src/collect_var.c:2944,2972) runs on liblcd_hotpath.so.  It flattens the
 * pointer-rich chunk state -- cand_var_t (src/collect_var.h:71-95), read_var_profile_t (:98-104), chunk->read_var_cr (cgranges, src/cgranges.h) and
 * the per-read arrays of bam_chunk_t (src/bam_utils.h:45-92) -- into lcd_hap_problem_t, calls lcd_assign_hap_germline, and writes back exactly
 * what the reference function mutates: chunk->haps / phase_sets / phase_scores / n_clean_agree_snps / n_clean_conflict_snps and, per candidate
 * variant of the target categories, phase_set, hap_to_cons_alle[0..2] and hap_to_alle_profile[0..2][allele].
 *
 * In the longcallD tree this file is compiled as is (it includes the reference's own headers).  tests/c/ compiles it against a header that
 * declares only the fields used here (same names and types) and round-trips a chunk through it on the GPU (tests/test_gpu_hap.py).
 */
#include <stdlib.h>
#include <string.h>
#ifdef LCD_BINDING_STRUCTS_H
#include LCD_BINDING_STRUCTS_H
#else
#include "call_var_main.h"
#include "bam_utils.h"
#include "collect_var.h"
#include "cgranges.h"
#endif
#include "lcd_hotpath.h"

#ifndef LONGCALLD_DEF_PLOID
#define LONGCALLD_DEF_PLOID 2
#endif

int lcd_bind_assign_hap_based_on_germline_het_vars_kmeans(const call_var_opt_t *opt, bam_chunk_t *chunk, int target_var_cate) {
    const int R = chunk->n_reads, V = chunk->n_cand_vars;
    cand_var_t *cv = chunk->cand_vars;
    read_var_profile_t *p = chunk->read_var_profile;
    const cgranges_t *cr = chunk->read_var_cr;
    int i, h, a, any = 0;
    for (i = 0; i < V; ++i) any |= (chunk->var_i_to_cate[i] & target_var_cate) != 0;
    if (!any) return 0;                                                        /* src/assign_hap.c:482-485: nothing is touched */
    /* ---- variants: SoA + CSR over the alleles ---- */
    int64_t *var_pos = (int64_t *)malloc((size_t)(V + 1) * sizeof(int64_t)), *var_ps = (int64_t *)malloc((size_t)(V + 1) * sizeof(int64_t));
    int *var_type = (int *)malloc((size_t)(V + 1) * sizeof(int)), *is_hp = (int *)malloc((size_t)(V + 1) * sizeof(int));
    int *total_cov = (int *)malloc((size_t)(V + 1) * sizeof(int)), *alle_off = (int *)malloc((size_t)(V + 2) * sizeof(int));
    alle_off[0] = 0;
    for (i = 0; i < V; ++i) {
        var_pos[i] = cv[i].pos; var_ps[i] = cv[i].phase_set; var_type[i] = cv[i].var_type; is_hp[i] = cv[i].is_homopolymer_indel;
        total_cov[i] = cv[i].total_cov; alle_off[i + 1] = alle_off[i] + cv[i].n_uniq_alles;
    }
    const int TA = alle_off[V];
    int *alle_covs = (int *)malloc((size_t)(TA + 1) * sizeof(int));
    int *cons = (int *)malloc((size_t)(3 * V + 1) * sizeof(int)), *prof = (int *)calloc((size_t)(3 * TA + 1), sizeof(int));
    for (i = 0; i < V; ++i) {
        const int is_target = (chunk->var_i_to_cate[i] & target_var_cate) != 0;
        memcpy(alle_covs + alle_off[i], cv[i].alle_covs, (size_t)cv[i].n_uniq_alles * sizeof(int));
        if (is_target && cv[i].hap_to_alle_profile == NULL) {                  /* first call for this variant: allocate as var_init_hap_profile_cons_allele does (:42-45) */
            cv[i].hap_to_alle_profile = (int **)malloc((LONGCALLD_DEF_PLOID + 1) * sizeof(int *));
            for (h = 0; h <= LONGCALLD_DEF_PLOID; ++h) cv[i].hap_to_alle_profile[h] = (int *)calloc((size_t)cv[i].n_uniq_alles, sizeof(int));
            cv[i].hap_to_cons_alle = (int *)malloc((LONGCALLD_DEF_PLOID + 1) * sizeof(int));
            for (h = 0; h <= LONGCALLD_DEF_PLOID; ++h) cv[i].hap_to_cons_alle[h] = -1;
        }
        for (h = 0; h < 3; ++h) {
            cons[i * 3 + h] = cv[i].hap_to_cons_alle ? cv[i].hap_to_cons_alle[h] : -1;
            if (cv[i].hap_to_alle_profile) for (a = 0; a < cv[i].n_uniq_alles; ++a) prof[h * TA + alle_off[i] + a] = cv[i].hap_to_alle_profile[h][a];
        }
    }
    /* ---- reads: CSR over read_var_profile_t.alleles (one int per variant of [start_var_idx, end_var_idx]) ---- */
    int *start_var = (int *)malloc((size_t)(R + 1) * sizeof(int)), *end_var = (int *)malloc((size_t)(R + 1) * sizeof(int));
    int *allele_off = (int *)malloc((size_t)(R + 2) * sizeof(int));
    allele_off[0] = 0;
    for (i = 0; i < R; ++i) {
        start_var[i] = p[i].start_var_idx; end_var[i] = p[i].end_var_idx;
        allele_off[i + 1] = allele_off[i] + (p[i].start_var_idx >= 0 && p[i].end_var_idx >= p[i].start_var_idx ? p[i].end_var_idx - p[i].start_var_idx + 1 : 0);
    }
    int *alleles = (int *)malloc((size_t)(allele_off[R] + 1) * sizeof(int));
    for (i = 0; i < R; ++i) if (allele_off[i + 1] > allele_off[i]) memcpy(alleles + allele_off[i], p[i].alleles, (size_t)(allele_off[i + 1] - allele_off[i]) * sizeof(int));
    /* ---- chunk->read_var_cr: the labels (read ids) in the index's sorted interval order; cr_overlap reports hits in that order (:511) ---- */
    const int n_cr = (int)cr->n_r;
    int *cr_read = (int *)malloc((size_t)(n_cr + 1) * sizeof(int));
    for (i = 0; i < n_cr; ++i) cr_read[i] = cr->r[i].label;
    lcd_hap_problem_t q;
    memset(&q, 0, sizeof(q));
    q.n_reads = R; q.n_vars = V; q.is_ont = opt->is_ont;
    q.var_pos = var_pos; q.var_type = var_type; q.var_cate = chunk->var_i_to_cate; q.is_homopolymer_indel = is_hp; q.total_cov = total_cov;
    q.alle_off = alle_off; q.alle_covs = alle_covs; q.start_var_idx = start_var; q.end_var_idx = end_var; q.allele_off = allele_off; q.alleles = alleles;
    q.ordered_read_ids = chunk->ordered_read_ids; q.is_skipped = chunk->is_skipped; q.n_cr = n_cr; q.cr_read = cr_read;
    q.haps = chunk->haps; q.phase_sets = (int64_t *)chunk->phase_sets;              /* hts_pos_t == int64_t */
    q.n_clean_agree_snps = chunk->n_clean_agree_snps; q.n_clean_conflict_snps = chunk->n_clean_conflict_snps;
    q.var_phase_set = var_ps; q.hap_to_cons_alle = cons; q.hap_to_alle_profile = prof;
    const int rc = lcd_assign_hap_germline(&q, target_var_cate);
    if (rc == 0) {
        /* ---- write back what the reference function mutates ---- */
        for (i = 0; i < R; ++i) chunk->phase_scores[i] = 0;                        /* read_init_hap_phase_set, src/assign_hap.c:18 */
        for (i = 0; i < V; ++i) {
            if ((chunk->var_i_to_cate[i] & target_var_cate) == 0) continue;
            cv[i].phase_set = var_ps[i];
            for (h = 0; h < 3; ++h) {
                cv[i].hap_to_cons_alle[h] = cons[i * 3 + h];
                for (a = 0; a < cv[i].n_uniq_alles; ++a) cv[i].hap_to_alle_profile[h][a] = prof[h * TA + alle_off[i] + a];
            }
        }
    }
    free(var_pos); free(var_ps); free(var_type); free(is_hp); free(total_cov); free(alle_off); free(alle_covs); free(cons); free(prof);
    free(start_var); free(end_var); free(allele_off); free(alleles); free(cr_read);
    return rc;   /* 0, as the reference returns; < 0: lcd_last_error() (the stub in src/assign_hap.c turns that into _err_error_exit) */
}
