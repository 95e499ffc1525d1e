/* k5_roundtrip.c -- TEST ONLY: builds the reference-shaped (array-of-structs) chunk state from flat test arrays, runs the K5 binding on it,
 * and hands the mutated state back as flat arrays.  tests/test_gpu_hap.py compares them with the oracle. */
#include <stdlib.h>
#include <string.h>
#include "ref_fields_min.h"
int lcd_bind_assign_hap_based_on_germline_het_vars_kmeans(const call_var_opt_t *opt, bam_chunk_t *chunk, int target_var_cate);

/* two calls in a row, as collect_var_main does (clean categories, then all germline categories: src/collect_var.c:2944,2972) */
int k5_roundtrip(int n_reads, int n_vars, int is_ont, const int64_t *var_pos, const int *var_type, int *var_cate, const int *is_hp, const int *total_cov,
                 const int *alle_off, const int *alle_covs, const int *start_var, const int *end_var, const int *allele_off, const int *alleles,
                 int *ordered, uint8_t *is_skipped, int n_cr, const int *cr_read, int n_targets, const int *targets,
                 int *haps, int64_t *phase_sets, int *agree, int *conflict, int64_t *var_ps, int *cons, int *prof) {
    int i, h, a, t, rc = 0;
    cand_var_t *cv = (cand_var_t *)calloc((size_t)n_vars + 1, sizeof(cand_var_t));
    for (i = 0; i < n_vars; ++i) {
        cv[i].pos = var_pos[i]; cv[i].phase_set = -1; cv[i].var_type = var_type[i]; cv[i].is_homopolymer_indel = is_hp[i]; cv[i].total_cov = total_cov[i];
        cv[i].n_uniq_alles = alle_off[i + 1] - alle_off[i];
        cv[i].alle_covs = (int *)malloc((size_t)(cv[i].n_uniq_alles + 1) * sizeof(int));
        memcpy(cv[i].alle_covs, alle_covs + alle_off[i], (size_t)cv[i].n_uniq_alles * sizeof(int));
    }
    read_var_profile_t *p = (read_var_profile_t *)calloc((size_t)n_reads + 1, sizeof(read_var_profile_t));
    for (i = 0; i < n_reads; ++i) {
        const int n = allele_off[i + 1] - allele_off[i];
        p[i].read_id = i; p[i].start_var_idx = start_var[i]; p[i].end_var_idx = end_var[i];
        p[i].alleles = (int *)malloc((size_t)(n + 1) * sizeof(int));
        memcpy(p[i].alleles, alleles + allele_off[i], (size_t)n * sizeof(int));
    }
    cgranges_t cr; cr.n_r = cr.m_r = n_cr; cr.r = (cr_intv_t *)calloc((size_t)n_cr + 1, sizeof(cr_intv_t));
    for (i = 0; i < n_cr; ++i) { cr.r[i].label = cr_read[i]; cr.r[i].x = ((uint64_t)(uint32_t)start_var[cr_read[i]] << 32) | (uint32_t)(end_var[cr_read[i]] + 1); }
    bam_chunk_t ch; memset(&ch, 0, sizeof(ch));
    ch.n_reads = n_reads; ch.ordered_read_ids = ordered; ch.n_clean_agree_snps = agree; ch.n_clean_conflict_snps = conflict; ch.is_skipped = is_skipped;
    ch.n_cand_vars = n_vars; ch.cand_vars = cv; ch.var_i_to_cate = var_cate; ch.read_var_profile = p; ch.read_var_cr = &cr;
    ch.phase_scores = (int *)calloc((size_t)n_reads + 1, sizeof(int)); ch.haps = haps; ch.phase_sets = phase_sets;
    call_var_opt_t opt; opt.is_ont = is_ont;
    for (t = 0; t < n_targets && rc == 0; ++t) rc = lcd_bind_assign_hap_based_on_germline_het_vars_kmeans(&opt, &ch, targets[t]);
    const int TA = alle_off[n_vars];
    for (i = 0; i < n_vars; ++i) {
        var_ps[i] = cv[i].phase_set;
        for (h = 0; h < 3; ++h) {
            cons[i * 3 + h] = cv[i].hap_to_cons_alle ? cv[i].hap_to_cons_alle[h] : -1;
            for (a = 0; a < cv[i].n_uniq_alles; ++a) prof[h * TA + alle_off[i] + a] = cv[i].hap_to_alle_profile ? cv[i].hap_to_alle_profile[h][a] : 0;
        }
        if (cv[i].hap_to_alle_profile) { for (h = 0; h < 3; ++h) free(cv[i].hap_to_alle_profile[h]); free(cv[i].hap_to_alle_profile); free(cv[i].hap_to_cons_alle); }
        free(cv[i].alle_covs);
    }
    for (i = 0; i < n_reads; ++i) free(p[i].alleles);
    free(cv); free(p); free(cr.r); free(ch.phase_scores);
    return rc;
}
