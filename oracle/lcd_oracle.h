/*
 * lcd_oracle.h -- CPU ORACLE for the longcallD per-region alignment/phasing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call anything under oracle/.  The product
 * path (longcalld_amd/csrc, liblcd_hotpath.so) never includes this header.
 *
 * Each function is a plain-C, sequential restatement of what the reference computes on
 * this path; every one cites the reference file:line it follows (paths relative to
 * /root/reference).  Parity status per kernel:
 *   K4 edlib   : PINNED   -- checked against the reference's own vendored edlib
 *                            (oracle/_ref/libedlib_ref.so, built from edlib/src/edlib.cpp)
 *                            and the golden vectors in tests/golden/edlib_*.json.
 *   K5 hap     : source-restated from src/assign_hap.c:16-547 (all source present);
 *                "parity unpinned" (reference cannot be built: htslib headers absent).
 *   align glue : source-restated from src/align.c; "parity unpinned" (same reason).
 *   K3 WFA2    : "parity unpinned" -- WFA2-lib (github.com/smarco/WFA2-lib, submodule,
 *                pin unknown) is absent; restates the published WFA gap-affine-2p
 *                algorithm (Marco-Sola et al. 2021/2023).  Scores are pinned against an
 *                independent O(nm) Gotoh DP (oracle/gotoh2p.c).
 *   K1/K2 POA  : "parity unpinned" -- abPOA (github.com/yangao07/abPOA, submodule, pin
 *                unknown) is absent; restates the published adaptive-banded POA
 *                (Gao et al. 2021).  Scores pinned against an unbanded DAG DP.
 */
#ifndef LCD_ORACLE_H
#define LCD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- byte codes (src/seq.c:14-31, src/align.c:316,321) ---- */
#define LCDO_GAP 5

/* ---- cover flags (src/align.h:6-18) ---- */
#define LCDO_RIGHT_GAP 0x1
#define LCDO_LEFT_GAP 0x2
#define LCDO_RIGHT_COVER 0x4
#define LCDO_LEFT_COVER 0x8
#define LCDO_BOTH_COVER 0xC
#define LCDO_IS_BOTH_COVER(c) (((c) & LCDO_LEFT_COVER) && ((c) & LCDO_RIGHT_COVER))
#define LCDO_IS_LEFT_COVER(c) (((c) & LCDO_LEFT_COVER) && ((c) & LCDO_RIGHT_COVER) == 0)
#define LCDO_IS_LEFT_GAP(c) ((c) & LCDO_LEFT_GAP)
#define LCDO_IS_RIGHT_COVER(c) (((c) & LCDO_LEFT_COVER) == 0 && ((c) & LCDO_RIGHT_COVER))
#define LCDO_IS_RIGHT_GAP(c) ((c) & LCDO_RIGHT_GAP)
#define LCDO_IS_NOT_COVER(c) (((c) & LCDO_LEFT_COVER) == 0 && ((c) & LCDO_RIGHT_COVER) == 0)

#define LCDO_GAP_LEFT_ALN 1
#define LCDO_GAP_RIGHT_ALN 2
#define LCDO_EXT_LEFT_TO_RIGHT 1
#define LCDO_EXT_RIGHT_TO_LEFT 2

/* BAM cigar op codes used by the path (htslib/sam.h values) */
#define LCDO_CMATCH 0
#define LCDO_CINS 1
#define LCDO_CDEL 2
#define LCDO_CSOFT_CLIP 4
#define LCDO_CHARD_CLIP 5
#define LCDO_CEQUAL 7
#define LCDO_CDIFF 8

/* edlib edit ops (edlib/include/edlib.h:39-42) */
#define LCDO_EDOP_MATCH 0
#define LCDO_EDOP_INSERT 1
#define LCDO_EDOP_DELETE 2
#define LCDO_EDOP_MISMATCH 3

/* the subset of call_var_opt_t the path reads (src/call_var_main.h:128-180) */
typedef struct {
    int match, mismatch, gap_open1, gap_ext1, gap_open2, gap_ext2; /* 2,6,6,2,24,1 */
    int gap_aln;                                                   /* 1 = left */
    double min_af;                                                 /* 0.20 */
    int min_dp;                                                    /* 5 */
    double partial_aln_ratio;                                      /* 1.1 */
    int min_noisy_reg_size_to_sample_reads;                        /* 10000 */
    int max_noisy_reg_len;                                         /* 50000 */
    int noisy_reg_flank_len;                                       /* 10 */
    int min_hap_full_reads, min_hap_reads;                         /* 1, 2 */
    int collect_ref_read_aln_str; /* (refine_bam && out_aln_fp) || out_somatic */
    int is_ont;
} lcdo_opt_t;

void lcdo_opt_default(lcdo_opt_t *opt); /* src/call_var_main.c:140-224 */

/* ---------------- K4: edlib NW (edlib/src/edlib.cpp) ---------------- */
/* edit distance + edlib's exact traceback path (ops start->end); returns distance, -1 on error.
 * *aln is malloc'd (caller frees) when aln != NULL. */
int lcdo_edlib_nw(const uint8_t *query, int qlen, const uint8_t *target, int tlen, uint8_t **aln, int *aln_len);
int lcdo_edlib_xgaps(const uint8_t *target, int tlen, const uint8_t *query, int qlen);        /* src/align.c:222 */
int lcdo_edlib_edit_distance(const uint8_t *target, int tlen, const uint8_t *query, int qlen); /* src/align.c:210 */
int lcdo_edlib_end2end_aln(const uint8_t *target, int tlen, const uint8_t *query, int qlen, int *n_eq, int *n_xid); /* :234 */
/* HW (infix) mode, edlibAlign(.., EDLIB_MODE_HW, EDLIB_TASK_PATH): distance, the first end position and its start in the target, the NW path of the query against that stretch */
int lcdo_edlib_hw(const uint8_t *query, int qlen, const uint8_t *target, int tlen, int *start_out, int *end_out, uint8_t **aln, int *aln_len);
int lcdo_edlib_infix_aln(const uint8_t *target, int tlen, const uint8_t *query, int qlen, int *n_eq, int *n_xid);   /* src/align.c:256 */

/* ---------------- K3: WFA gap-affine-2p (src/align.c:374-460) ---------------- */
/* Mirrors wfa_end2end_aln for heuristic==NONE, affine_gap==2P. cigar_buf/pattern_alg malloc'd. returns 0. */
int lcdo_wfa_end2end_aln(const uint8_t *pattern, int plen, const uint8_t *text, int tlen, int gap_aln, int b, int q, int e,
                         int q2, int e2, uint32_t **cigar_buf, int *cigar_length, uint8_t **pattern_alg,
                         uint8_t **text_alg, int *alg_length, int *score);
/* independent O(nm) 2-piece Gotoh optimum (penalty, >= 0) -- pins the WFA score */
int lcdo_gotoh2p_score(const uint8_t *pattern, int plen, const uint8_t *text, int tlen, int b, int q, int e, int q2, int e2);
/* re-score a BAM-style cigar (=,X,I,D) under the 2-piece model; -1 if it does not consume both strings or mislabels =/X */
int lcdo_cigar_score2p(const uint32_t *cigar, int n_cigar, const uint8_t *pattern, int plen, const uint8_t *text, int tlen,
                       int b, int q, int e, int q2, int e2);

/* ---------------- K1/K2: POA (abPOA restatement) ---------------- */
typedef struct lcdo_poa_s lcdo_poa_t;

typedef struct {
    int n_cons;
    int cons_len[2];
    uint8_t *cons_seq[2];
    int clu_n_seq[2];
    int *clu_read_ids[2]; /* indices into the input read list */
    int n_seq, msa_len;
    uint8_t **msa; /* n_seq + n_cons rows of msa_len */
} lcdo_poa_result_t;
void lcdo_poa_result_free(lcdo_poa_result_t *r);

/* K1: src/align.c:762-857 (sub-graph incremental POA, 1 consensus).  Returns n_cons. */
int lcdo_poa_partial_aln_msa_cons(const lcdo_opt_t *opt, int sampling_reads, int n_reads, uint8_t **read_seqs,
                                  const int *read_lens, const int *read_full_cover, lcdo_poa_result_t *res);
/* K2: src/align.c:872-943 (unbanded de-novo MSA, <=2 consensus) */
int lcdo_poa_aln_msa_cons(const lcdo_opt_t *opt, int n_reads, uint8_t **read_seqs, const int *read_lens, int max_n_cons,
                          lcdo_poa_result_t *res);
/* score-level pin: unbanded optimum of aligning seq to the CURRENT graph of a chain built from the first n reads.
 * (used by tests only) */
int lcdo_poa_debug_last_scores(int *banded, int *unbanded);
/* counters of the certified-band checker (env LCDO_CERT_STATS=1 while lcdo_poa_aln_msa_cons runs; oracle/poa.c): out[0] rows, [1] cells of the full rows,
 * [2] cells inside the intervals for the true lower bound, [3] reads, [4] cells whose H exceeds the prefix bound (must be 0), [5] matched backtrack cells outside
 * their row's interval (must be 0), [6] widest interval, [7] rows wider than the 256-column window, [11] cells under the product's guessing policy (a retried
 * read counted twice), [12] retries, [13] reads with a row wider than the window under the policy, [14] same for the true bound, [15] / [16] regions without / all */
void lcdo_poa_cert_stats(long long *out);

/* ---------------- align.c glue ---------------- */
typedef struct {
    uint8_t *target_aln, *query_aln; /* one malloc block, target first (src/collect_var.h:106-112) */
    int aln_len, target_beg, target_end, query_beg, query_end;
} lcdo_aln_str_t;

/* flattened per-read inputs of collect_noisy_reg_aln_strs AFTER collect_noisy_read_info (src/align.c:1377) */
typedef struct {
    int n_reads;
    int *read_ids;      /* chunk-level ids (permuted in place by the sort, src/align.c:1774) */
    int *lens;
    uint8_t **seqs;
    uint8_t **quals;
    int *fully_covers;
    int *haps;
    int64_t *phase_sets;
} lcdo_region_reads_t;

void lcdo_sort_noisy_region_reads(lcdo_region_reads_t *r, int use_error_rate);                  /* src/align.c:955 */
int64_t lcdo_collect_phase_set_with_both_haps(const lcdo_region_reads_t *r, int min_full, int min_all); /* :1225 */
void lcdo_wfa_trim_aln_str(int full_cover, lcdo_aln_str_t *s);                                  /* :496 */
int lcdo_collect_partial_aln_beg_end(const lcdo_opt_t *opt, int sampling_reads, const uint8_t *target, int tlen,
                                     int target_full_cover, const uint8_t *query, int qlen, int query_full_cover,
                                     int *target_beg, int *target_end, int *query_beg, int *query_end); /* :709 */
/* region driver after read-info extraction: src/align.c:1760-1813 minus collect_noisy_read_info.
 * reg_len = noisy_reg_end - noisy_reg_beg + 1.  aln_strs[c] must hold 1+2*n_reads zeroed entries. returns n_cons */
int lcdo_collect_noisy_reg_aln_strs(const lcdo_opt_t *opt, int64_t reg_len, lcdo_region_reads_t *reads,
                                    const uint8_t *ref_seq, int ref_seq_len, int *clu_n_seqs, int **clu_read_ids,
                                    lcdo_aln_str_t **aln_strs);

/* digar walk of collect_noisy_read_info for ONE read (src/align.c:1392-1458) */
typedef struct {
    int64_t pos;
    int type, len, qi;
} lcdo_digar1_t;
void lcdo_read_region_slice(const lcdo_digar1_t *digars, int n_digar, int qlen, int64_t reg_beg, int64_t reg_end,
                            int noisy_reg_flank_len, int *reg_read_beg, int *reg_read_end, int *cover);

/* ---------------- SURVEY 8(f) f2 (first part): EQX CIGAR -> digars + the read's noisy windows (src/bam_utils.c:701-842) ---------------- */
typedef struct {
    int min_bq;                       /* 10 */
    int noisy_reg_max_xgaps;          /* 5 */
    int noisy_reg_slide_win;          /* 100 (HiFi) / 25 (ONT) */
    int end_clip_reg, end_clip_reg_flank_win; /* 30, 100 */
    double max_noisy_frac_per_read;   /* 0.5 */
    double max_var_ratio_per_read;    /* 0.05 */
} lcdo_digar_opt_t;
typedef struct { int64_t pos; int type, len, qi, is_low_qual; } lcdo_digar_t; /* digar1_t without the alt_seq copy */
/* returns 0, -1 (read skipped: too noisy) or -2 ('M' operation).  noisy / chunk_noisy: (start, end, label) triples as cr_add stores them
 * (start = first position - 1), in cr_index order; chunk_noisy = those overlapping [reg_beg, reg_end] (none when skipped).  malloc()'d. */
int lcdo_collect_digar_from_eqx_cigar(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const uint8_t *bseq,
                                      const uint8_t *qual, int qlen, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len,
                                      int left_clip_is_palindrome, int right_clip_is_palindrome, lcdo_digar_t **digars, int *n_digar,
                                      int64_t **noisy, int *n_noisy, int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg, int64_t *end,
                                      int *n_total_cand_vars);
void lcdo_cr_sorted_order(int n, const int *st, const int *en, int *order_out);
/* oracle/digar_tags.c: the same outputs from a cs:Z tag (src/bam_utils.c:844), an MD:Z tag (:1010) or the reference bases (:1179; bseq = BAM 4-bit bases,
 * ref_seq[0] = position ref_beg, ref_end inclusive); -2 where the reference stops the program */
int lcdo_collect_digar_from_cs_tag(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const char *cs, const uint8_t *qual, int qlen,
                                   int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome, int right_clip_is_palindrome,
                                   lcdo_digar_t **digars, int *n_digar, int64_t **noisy, int *n_noisy, int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg,
                                   int64_t *end, int *n_total_cand_vars);
int lcdo_collect_digar_from_MD_tag(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const char *md, const uint8_t *qual, int qlen,
                                   int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome, int right_clip_is_palindrome,
                                   lcdo_digar_t **digars, int *n_digar, int64_t **noisy, int *n_noisy, int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg,
                                   int64_t *end, int *n_total_cand_vars);
int lcdo_collect_digar_from_ref_seq(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const uint8_t *bseq, const uint8_t *qual, int qlen,
                                    const char *ref_seq, int64_t ref_beg, int64_t ref_end, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len,
                                    int left_clip_is_palindrome, int right_clip_is_palindrome, lcdo_digar_t **digars, int *n_digar, int64_t **noisy, int *n_noisy,
                                    int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg, int64_t *end, int *n_total_cand_vars);

/* ---------------- region alignment strings -> candidate variants + read x variant profile ---------------- */
typedef struct {
    int64_t pos;                 /* cand_var_t.pos (1-based reference position) */
    int var_type, ref_len, alt_len; /* BAM_CDIFF 8 / BAM_CINS 1 / BAM_CDEL 2 */
    int cate;                    /* LONGCALLD_NOISY_CAND_HET_VAR 0x100 / _HOM_VAR 0x200 */
    int from_cons;               /* 1: consensus 1, 2: consensus 2, 3: both (var_from_cons_idx) */
    int is_homopolymer_indel;
    int ref_base, alt_ref_base;
    int total_cov, alle_covs[2];
    int alt_off;                 /* alt_seq = alt_pool + alt_off, alt_len bytes */
} lcdo_noisy_var_t;
/* make_vars_from_msa_cons_aln, src/collect_var.c:2279.  Profile rows: the reads of cluster 0, then those of cluster 1, in
 * clu_read_ids order; prof_alleles is n_rows x n_vars (row-major), -1 where the read does not fully cover the variant. Returns n_vars;
 * all outputs are malloc()'d (NULL when there is nothing). */
int lcdo_make_vars_from_msa_cons_aln(int min_sv_len, const uint8_t *chunk_ref, int64_t chunk_ref_beg, int64_t chunk_ref_len,
                                     int64_t noisy_reg_beg, int n_cons, const int *clu_n_seqs, lcdo_aln_str_t **aln_strs,
                                     lcdo_noisy_var_t **vars, uint8_t **alt_pool, int **prof_start, int **prof_end, int **prof_alleles);

/* ---------------- K5: hap assignment (src/assign_hap.c:473-547) ---------------- */
/* bam_chunk_t / cand_var_t / read_var_profile_t flattened (src/collect_var.h:71-104, src/bam_utils.h:45-92).
 * Identical field list to lcd_hap_problem_t in include/lcd_hotpath.h (kept separate on purpose: the oracle never includes
 * product headers). */
typedef struct {
    int n_reads, n_vars, is_ont;
    /* per var (index = position in chunk->cand_vars, sorted by position) */
    const int64_t *var_pos;          /* cand_var_t.pos */
    const int *var_type;             /* BAM_CDIFF 8 / BAM_CINS 1 / BAM_CDEL 2 */
    const int *var_cate;             /* chunk->var_i_to_cate */
    const int *is_homopolymer_indel;
    const int *total_cov;
    const int *alle_off;             /* n_vars+1: CSR offsets into alle_covs and the profile planes */
    const int *alle_covs;
    /* per read */
    const int *start_var_idx, *end_var_idx; /* read_var_profile_t; -1 = no var */
    const int *allele_off;           /* n_reads+1: CSR offsets into alleles */
    const int *alleles;              /* 0 ref, 1 alt, -1, -2 */
    const int *ordered_read_ids;     /* chunk->ordered_read_ids, n_reads */
    const uint8_t *is_skipped;
    /* chunk->read_var_cr after cr_index(): labels (read ids) in the sorted interval order cr_overlap reports them */
    int n_cr;
    const int *cr_read;
    /* in/out state */
    int *haps;                       /* n_reads */
    int64_t *phase_sets;             /* n_reads */
    int *n_clean_agree_snps, *n_clean_conflict_snps; /* n_reads */
    int64_t *var_phase_set;          /* n_vars */
    int *hap_to_cons_alle;           /* n_vars*3 */
    int *hap_to_alle_profile;        /* 3 planes of alle_off[n_vars] ints: [h*total + alle_off[v] + a] */
} lcdo_hap_problem_t;
int lcdo_assign_hap_germline(lcdo_hap_problem_t *p, int target_var_cate);

/* ---------------- update_digars_from_msa1 (oracle/digar_rewrite.c) ---------------- */
int lcdo_update_digars_from_msa1(const lcdo_digar_t *digars, int n_digar, int qlen, int msa_len, const uint8_t *ref_str, const uint8_t *read_str, int full_cover,
                                 int64_t noisy_reg_beg, int64_t noisy_reg_end, int read_beg, int read_end, lcdo_digar_t **out, int *n_out);

/* ---------------- stitching, genotype records, VCF body text (oracle/emit.c) ---------------- */
typedef struct { double log_p, log_1p, log_2; int max_gq, max_qual, min_sv_len, min_dp, min_alt_dp, out_amb_base; } lcdo_call_opt_t;
typedef struct {
    int64_t pos, PS;
    int type, ref_len, n_alt_allele, alt_len[2];
    uint8_t *ref_bases, *alt_bases[2];
    int GT[2], DP, AD[3], QUAL, GQ, is_sv, is_clean, n_alt_reads; /* AD[2]: the int the reference's formatter finds behind AD[1] (third allele's coverage, else the GT bytes) */
    int *alt_read_i;
    int cand_i;
    int tsd_len, polya_len, te_seq_i, te_is_rev;   /* var1_t's retrotransposon members (src/collect_var.c:1504-1520), filled by lcdo_annotate_te */
    int64_t tsd_pos1, tsd_pos2;
    uint8_t *tsd_seq;
} lcdo_var1_t;
typedef struct {
    int tid, n_reads, n_vars;
    const int *ordered_read_ids; const uint8_t *is_skipped;
    int *haps; int64_t *phase_sets; int64_t *var_phase_set; int *hap_to_cons_alle;
    int n_up_ovlp, n_down_ovlp; const int *up_ovlp_read_i, *down_ovlp_read_i;
    int flip_hap; int64_t flip_pre_PS, flip_cur_PS;
} lcdo_chunk_phase_t;
int lcdo_make_variants(const lcdo_call_opt_t *opt, const lcdo_hap_problem_t *p, const int *var_ref_len, const int *var_alt_len, const uint64_t *alt_off,
                       const uint8_t *alt_pool, const uint8_t *alt_ref_base, const char *ref_seq, int64_t ref_beg, int64_t reg_beg, int64_t reg_end,
                       lcdo_var1_t **vars_out);
void lcdo_free_variants(lcdo_var1_t *v, int n);
int lcdo_flip_variant_hap(lcdo_chunk_phase_t *pre_chunk, lcdo_chunk_phase_t *cur_chunk, int out_aln);
int lcdo_format_vcf(const lcdo_call_opt_t *opt, const char *chrom, const lcdo_var1_t *vars, int n_vars, char **text_out);
/* the same with the INFO keys of annotated records (src/vcf_utils.c:184-195); te_names may be NULL */
int lcdo_format_vcf_te(const lcdo_call_opt_t *opt, const char *chrom, const lcdo_var1_t *vars, int n_vars, const char *const *te_names, char **text_out);
struct lcdo_te_lib;
/* collect_te_info_from_cons for the candidates behind finished records (oracle/te_info.c); te_lib may be NULL */
int lcdo_annotate_te(const lcdo_call_opt_t *opt, int min_tsd_len, int max_tsd_len, int min_polya_len, float min_polya_ratio, const struct lcdo_te_lib *te_lib,
                     const char *ref_seq, int64_t ref_beg, int64_t ref_end, lcdo_var1_t *vars, int n_vars);
/* order of intervals (st[i], en[i], label i) after cr_index(): cr_is_sorted / radix_sort_cr_intv (src/cgranges.c:13-86,162,350) */
void lcdo_cr_sorted_order(int n, const int *st, const int *en, int *order_out);
/* oracle/digar_tags.c: the same outputs from a cs:Z tag (src/bam_utils.c:844), an MD:Z tag (:1010) or the reference bases (:1179; bseq = BAM 4-bit bases,
 * ref_seq[0] = position ref_beg, ref_end inclusive); -2 where the reference stops the program */
int lcdo_collect_digar_from_cs_tag(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const char *cs, const uint8_t *qual, int qlen,
                                   int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome, int right_clip_is_palindrome,
                                   lcdo_digar_t **digars, int *n_digar, int64_t **noisy, int *n_noisy, int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg,
                                   int64_t *end, int *n_total_cand_vars);
int lcdo_collect_digar_from_MD_tag(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const char *md, const uint8_t *qual, int qlen,
                                   int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome, int right_clip_is_palindrome,
                                   lcdo_digar_t **digars, int *n_digar, int64_t **noisy, int *n_noisy, int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg,
                                   int64_t *end, int *n_total_cand_vars);
int lcdo_collect_digar_from_ref_seq(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const uint8_t *bseq, const uint8_t *qual, int qlen,
                                    const char *ref_seq, int64_t ref_beg, int64_t ref_end, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len,
                                    int left_clip_is_palindrome, int right_clip_is_palindrome, lcdo_digar_t **digars, int *n_digar, int64_t **noisy, int *n_noisy,
                                    int64_t **chunk_noisy, int *n_chunk_noisy, int64_t *beg, int64_t *end, int *n_total_cand_vars);

/* ---- SURVEY a14 (oracle/te_info.c): collect_te_info (src/align.c:32-83), collect_te_info_from_cons (:139-163), check_te_seq and the k-mer sets (src/kmer.c) ---- */
typedef struct lcdo_te_lib lcdo_te_lib_t;
lcdo_te_lib_t *lcdo_te_lib_create(int n_seqs, const char *const *seqs, const int *lens, int k);
void lcdo_te_lib_destroy(lcdo_te_lib_t *L);
int lcdo_check_te_seq(const lcdo_te_lib_t *L, const uint8_t *seq, int len, int *is_rev);
int lcdo_collect_te_info(int min_tsd_len, int max_tsd_len, int min_polya_len, float min_polya_ratio, const lcdo_te_lib_t *lib, int var_type,
                         const uint8_t *gap_seq, const uint8_t *flank_ref_seq, int gap_len, int64_t gap_pos, uint8_t *tsd_seq, int64_t *tsd_pos1,
                         int64_t *tsd_pos2, int *tsd_polya_len, int *te_seq_i, int *te_is_rev);
int lcdo_collect_te_info_from_cons(int min_tsd_len, int max_tsd_len, int min_polya_len, float min_polya_ratio, const lcdo_te_lib_t *lib, const char *ref_seq,
                                   int64_t ref_beg, int64_t ref_end, int64_t gap_ref_start, int msa_gap_start, int var_type, int gap_len,
                                   const uint8_t *cons_msa_seq, uint8_t *tsd_seq, int64_t *tsd_pos1, int64_t *tsd_pos2, int *tsd_polya_len, int *te_seq_i,
                                   int *te_is_rev);

#ifdef __cplusplus
}
#endif
#endif
