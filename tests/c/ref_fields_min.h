/* ref_fields_min.h -- TEST ONLY.  The fields of the reference's chunk state that longcalld_amd/binding/lcd_assign_hap_binding.c touches, with the
 * reference's names and types (src/collect_var.h:71-104, src/bam_utils.h:45-92, src/cgranges.h:36-49, src/call_var_main.h:128-180), so that the
 * binding can be compiled and round-tripped without htslib.  Not a reference header and never used to build reference code. */
#ifndef LCD_TEST_REF_FIELDS_MIN_H
#define LCD_TEST_REF_FIELDS_MIN_H
#include <stdint.h>
typedef int64_t hts_pos_t;
typedef struct { uint64_t x; uint32_t y : 31, rev : 1; int32_t label; } cr_intv_t;
typedef struct { int64_t n_r, m_r; cr_intv_t *r; } cgranges_t;
typedef struct cand_var_t {
    hts_pos_t pos, phase_set;
    int var_type, is_homopolymer_indel, total_cov, n_uniq_alles;
    int *alle_covs;
    int **hap_to_alle_profile;
    int *hap_to_cons_alle;
} cand_var_t;
typedef struct read_var_profile_t { int read_id, start_var_idx, end_var_idx; int *alleles; } read_var_profile_t;
typedef struct bam_chunk_t {
    int n_reads; int *ordered_read_ids;
    int *n_clean_agree_snps, *n_clean_conflict_snps;
    uint8_t *is_skipped;
    int n_cand_vars; cand_var_t *cand_vars; int *var_i_to_cate;
    read_var_profile_t *read_var_profile; cgranges_t *read_var_cr;
    int *phase_scores, *haps; hts_pos_t *phase_sets;
} bam_chunk_t;
typedef struct call_var_opt_t { int is_ont; } call_var_opt_t;
#endif
