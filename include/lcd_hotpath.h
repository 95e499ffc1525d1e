/*
 * lcd_hotpath.h -- C ABI of liblcd_hotpath.so: the MI355X (gfx950) implementation of longcallD's
 * per-noisy-region alignment/phasing hot path.  Plain pointers and sizes only; no torch, no C++ types.
 *
 * Every entry point names the reference interface it replaces (paths relative to the longcallD tree).
 * INTEGRATION.md shows the few lines a longcallD maintainer adds to src/align.c / src/assign_hap.c /
 * src/collect_var.c to route the reference's own symbols here.
 *
 * PARITY STATUS (read before relying on byte identity with a real longcallD build):
 *   pinned to the reference's own code compiled in this project's container: K4 (edlib: distance, path, xgaps), the cgranges interval order, sdust;
 *   PARITY UNPINNED for K1/K2 (abPOA) and K3 (WFA2): the abPOA / WFA2-lib submodules are empty in the reference checkout, so consensus, MSA and CIGAR
 *   tie-breaks follow this project's restatement of the published algorithms (oracle/poa.c, oracle/wfa2p.c), pinned at score level only;
 *   K5, the align.c glue, f1 and f2 are line-by-line restatements of source that IS present, but no reference binary can be built here to confirm them.
 *   tests/test_replay_reference_dump.py replays `longcallD call -V 3` dumps of a real build once one is available (skipped until then).
 *
 * Conventions kept from the reference:
 *   - byte codes A0 C1 G2 T3 N4, gap 5 (src/seq.c:14-31, src/align.c:316,321);
 *   - buffers handed back are libc malloc()'d and owned by the caller, with the reference's interior
 *     pointer layout (aln_str_t: one block, target row first; only target_aln is free()d,
 *     src/collect_var.c:2718-2724);
 *   - return values: n_cons for the region call, distance/-1 for edlib, 0 for the WFA wrapper with
 *     failure signalled by *cigar_length == 0 (src/align.c:703);
 *   - unrecoverable states (no GPU, kernel error, arena exhaustion after retries) return a negative
 *     code and set lcd_last_error(); there is NO CPU fallback.
 *   - all entry points are thread-safe: each call (or each lcd_batch_t) owns its HIP stream and buffers.
 */
#ifndef LCD_HOTPATH_H
#define LCD_HOTPATH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the fields of call_var_opt_t (src/call_var_main.h:128-180) that the path reads */
typedef struct lcd_opt_t {
    int match, mismatch, gap_open1, gap_ext1, gap_open2, gap_ext2; /* src/align.h:21-26 */
    int gap_aln;                                                   /* LONGCALLD_GAP_LEFT_ALN = 1 */
    double min_af;
    int min_dp;
    double partial_aln_ratio;
    int min_noisy_reg_size_to_sample_reads, max_noisy_reg_len, noisy_reg_flank_len;
    int min_hap_full_reads, min_hap_reads;
    int collect_ref_read_aln_str; /* (refine_bam && out_aln_fp) || out_somatic, src/align.c:1785-1786 */
    int is_ont;
    /* additive: 1 = also run make_vars_from_msa_cons_aln (src/collect_var.c:2279) on the device after the strings are
     * built and return it through lcd_batch_region_vars; 2 = the same, and lcd_batch_download leaves the alignment strings in HBM
     * (lcd_batch_region_result then fails): only variants and alleles cross PCIe */
    int collect_noisy_vars;
    int min_sv_len;               /* call_var_opt_t.min_sv_len (LONGCALLD_MIN_SV_LEN 30, src/call_var_main.h:54,175) */
} lcd_opt_t;

/* == aln_str_t, src/collect_var.h:106-112 */
typedef struct lcd_aln_str_t {
    uint8_t *target_aln;
    uint8_t *query_aln;
    int aln_len;
    int target_beg, target_end, query_beg, query_end;
} lcd_aln_str_t;

/* digar1_t / digar_t views (src/bam_utils.h:27-43) for the read-slicing step (src/align.c:1377-1461) */
typedef struct lcd_digar1_t {
    int64_t pos;
    int type, len, qi;
} lcd_digar1_t;
typedef struct lcd_read_view_t {
    const lcd_digar1_t *digars;
    int n_digar;
    int qlen;            /* digar2qlen() */
    const uint8_t *bseq; /* BAM 4-bit packed bases (bam_get_seq) */
    const uint8_t *qual;
    int hap;             /* chunk->haps[read] */
    int64_t phase_set;   /* chunk->phase_sets[read] */
} lcd_read_view_t;

void lcd_opt_default(lcd_opt_t *opt);          /* src/call_var_main.c:140-224 */
int lcd_init(int device);                      /* process default device; 0 ok; <0: no usable gfx950 device */
/* One process, many GPUs (the reference's kt_for workers are threads of one process, src/call_var_main.c:773): a device belongs to an
 * lcd_batch_t (lcd_batch_create_on) or, for the per-call mirrors and K5, to the calling thread (lcd_set_thread_device; < 0 = process default). */
int lcd_device_count(void);
long long lcd_alloc_events(void);           /* device allocations made so far by the library's grow-only buffers (a steady state makes none) */
long long lcd_device_bytes(int device);     /* bytes those buffers hold on a device right now */
int lcd_set_thread_device(int device);
const char *lcd_last_error(void);
/* Host threads this process uses beside the calling thread: `team` = threads of a submission's short parallel loops (LCD_HOST_TEAM), `arena_threads` = threads that lay
 * results out on the host (LCD_ARENA_THREADS).  Their defaults are the CPUs this process may use (`cpus`: affinity mask cut by the cgroup's CPU quota) divided by the
 * processes of the job on this host (`local_world`: LOCAL_WORLD_SIZE, else WORLD_SIZE, else 1), at most 8 / 16 -- so N ranks on one host (one per GPU) share the host
 * instead of each taking what a lone process takes.  Additive (the reference sizes its one thread pool by -t, src/call_var_main.c:773).  Any pointer may be NULL; returns 0. */
int lcd_host_threads(int *team, int *arena_threads, int *cpus, int *local_world);
const char *lcd_version(void);

/* ---- drop-in mirrors of src/align.h:50-65 ---- */

/* replaces wfa_end2end_aln (definition src/align.c:374-376; parameter meaning follows the DEFINITION:
 * gap_aln, b, q, e, q2, e2, heuristic, affine_gap).  Only heuristic == 0 (none) and affine_gap == 1 (2-piece)
 * -- the germline-live configuration -- are implemented; others return -2. */
int lcd_wfa_end2end_aln(uint8_t *pattern, int plen, uint8_t *text, int tlen, int gap_aln, int b, int q, int e, int q2,
                        int e2, int heuristic, int affine_gap, uint32_t **cigar_buf, int *cigar_length,
                        uint8_t **pattern_alg, uint8_t **text_alg, int *alg_length);
/* replaces edlib_end2end_aln (src/align.c:234), edlib_xgaps (:222), edlib_edit_distance (:210) */
int lcd_edlib_end2end_aln(uint8_t *target, int tlen, uint8_t *query, int qlen, int *n_eq, int *n_xid);
int lcd_edlib_xgaps(uint8_t *target, int tlen, uint8_t *query, int qlen);
int lcd_edlib_edit_distance(uint8_t *target, int tlen, uint8_t *query, int qlen);

/* replaces end2end_aln (src/align.c:610; exported src/align.h:56, no caller in src) and wfa_collect_diff_ins_seq (src/align.c:463; callers
 * src/collect_var.c:203 and, with -s, src/assign_hap.c:1280): both are thin wrappers over the 2-piece WFA above */
int lcd_end2end_aln(const lcd_opt_t *opt, char *tseq, int tlen, uint8_t *qseq, int qlen, uint32_t **cigar_buf);
int lcd_wfa_collect_diff_ins_seq(const lcd_opt_t *opt, uint8_t *large_seq, int large_len, uint8_t *small_seq, int small_len, uint8_t **diff_seq);
/* replaces edlib_infix_aln (src/align.c:256-275, src/align.h:54; every caller is somatic-mode code): edlibAlign(query, target, {k = -1, EDLIB_MODE_HW,
 * EDLIB_TASK_PATH}) -- the query against the best-matching stretch of the target (free start and end in the target), the path on the first end position's
 * stretch (edlib/src/edlib.cpp:146-280).  Returns the edit distance (-1 on error); *n_eq / *n_xid as edlibAlignmentToXID counts the path (:164).  Byte-pinned to
 * the reference's own edlib (tests/golden/edlib_golden.json `hw_cases`). */
int lcd_edlib_infix_aln(uint8_t *target, int tlen, uint8_t *query, int qlen, int *n_eq, int *n_xid);
/* src/align.h:60: wfa_heuristic_aln (x-drop; NO caller in longcallD) is exported so that longcallD links against this library alone; it returns -2, sets
 * lcd_last_error() and prints to stderr -- never a silent result */
int lcd_wfa_heuristic_aln(uint8_t *pattern, int plen, uint8_t *text, int tlen, int a, int b, int q, int e, int q2, int e2, int *n_eq, int *n_xid);

/* replaces collect_noisy_reg_aln_strs (src/align.c:1760) with bam_chunk_t flattened to per-read views.
 * noisy_reads[] is permuted in place exactly as the reference does (src/align.c:1774). */
int lcd_collect_noisy_reg_aln_strs(const lcd_opt_t *opt, const lcd_read_view_t *chunk_reads, int64_t noisy_reg_beg,
                                   int64_t noisy_reg_end, int noisy_reg_i, int n_noisy_reg_reads, int *noisy_reads,
                                   const uint8_t *ref_seq, int ref_seq_len, int *clu_n_seqs, int **clu_read_ids,
                                   lcd_aln_str_t **aln_strs);

/* ---- K5: replaces assign_hap_based_on_germline_het_vars_kmeans (src/assign_hap.h:12, src/assign_hap.c:473-547) ----
 * bam_chunk_t / cand_var_t / read_var_profile_t flattened (src/collect_var.h:71-104, src/bam_utils.h:45-92). */
typedef struct lcd_hap_problem_t {
    int n_reads, n_vars, is_ont;
    const int64_t *var_pos;            /* cand_var_t.pos */
    const int *var_type;               /* BAM_CDIFF 8 / BAM_CINS 1 / BAM_CDEL 2 */
    const int *var_cate;               /* chunk->var_i_to_cate */
    const int *is_homopolymer_indel;
    const int *total_cov;
    const int *alle_off;               /* n_vars+1 CSR offsets into alle_covs and the three profile planes */
    const int *alle_covs;
    const int *start_var_idx, *end_var_idx; /* read_var_profile_t (-1 / -2: no variant) */
    const int *allele_off;             /* n_reads+1 CSR offsets into alleles */
    const int *alleles;
    const int *ordered_read_ids;
    const uint8_t *is_skipped;
    int n_cr;
    const int *cr_read;                /* labels of chunk->read_var_cr in its sorted interval order (cr->r[i].label) */
    int *haps;                         /* out: chunk->haps */
    int64_t *phase_sets;               /* out: chunk->phase_sets */
    int *n_clean_agree_snps, *n_clean_conflict_snps;
    int64_t *var_phase_set;            /* out: cand_var_t.phase_set */
    int *hap_to_cons_alle;             /* in/out n_vars*3 */
    int *hap_to_alle_profile;          /* in/out 3 planes of alle_off[n_vars] */
} lcd_hap_problem_t;
int lcd_assign_hap_germline(lcd_hap_problem_t *p, int target_var_cate);
int lcd_assign_hap_batch(int n, lcd_hap_problem_t *probs, const int *target_var_cates);

/* ---- batched form (additive; legal because regions of one pass are independent, SURVEY CS-2) ---- */
typedef struct lcd_batch_s lcd_batch_t;

typedef struct lcd_batch_stats_t {
    int n_regions, n_regions_resolved; /* regions reaching K1/K2 ; regions with n_cons > 0 */
    int n_chains, n_anchor_jobs, n_wfa_jobs, n_edlib_jobs;
    uint64_t poa_aligned_bases, poa_cells, wfa_offsets, edlib_blocks; /* poa_cells: DP cells of the reference's algorithm (K1: adaptive band, K2: full rows) */
    uint64_t poa_alg_bytes; /* SURVEY 8d B_poa summed over aligned reads (C = poa_cells) */
    double ms_total, ms_anchor, ms_poa, ms_wfa, ms_strings; /* HIP-event times on the batch stream */
    double ms_upload, ms_download, ms_host;
    double ms_poa_kernel;   /* HIP events tight around the POA chain kernel launch(es) only */
    int n_poa_launches;
    int poa_retries;
    double ms_vars;         /* stage S6 (opt.collect_noisy_vars) */
    uint64_t poa_cells_computed; /* cells the kernels actually computed: < poa_cells where K2 ran over a certified band (same alignments) */
    int poa_grown;               /* DP regions enlarged in place by a chain whose estimate was too small for a read (spare DP memory); such a chain is not re-run */
} lcd_batch_stats_t;

#define LCD_DEVICE_ANY (-2)
lcd_batch_t *lcd_batch_create(const lcd_opt_t *opt);                 /* on the calling thread's device */
/* device >= 0: that GPU; -1: the calling thread's; LCD_DEVICE_ANY: a host-only job buffer whose device is chosen at upload / dispatch time.
 * Batches of one lcd_batch_run_many call share a device. */
lcd_batch_t *lcd_batch_create_on(const lcd_opt_t *opt, int device);
void lcd_batch_destroy(lcd_batch_t *b);
void lcd_batch_clear(lcd_batch_t *b);
/* add one region AFTER read slicing (the outputs of collect_noisy_read_info, src/align.c:1377): returns region index */
int lcd_batch_add_region(lcd_batch_t *b, int64_t reg_len, int n_reads, const int *read_ids, const int *lens,
                         const uint8_t *const *seqs, const uint8_t *const *quals, const int *fully_covers, const int *haps,
                         const int64_t *phase_sets, const uint8_t *ref_seq, int ref_seq_len);
/* add one region from chunk views (does the digar walk of src/align.c:1392-1458 on the host) */
int lcd_batch_add_region_from_chunk(lcd_batch_t *b, const lcd_read_view_t *chunk_reads, int64_t noisy_reg_beg,
                                    int64_t noisy_reg_end, int n_noisy_reg_reads, const int *noisy_reads,
                                    const uint8_t *ref_seq, int ref_seq_len);
/* the same region with the reads' bases left as they are in their BAM records (4 bits per base): the host copies the slices' bytes, lcd_batch_upload unpacks them
 * on the device (seq_nt16_int[bam_seqi()], src/align.c:1445-1448) into the batch's input pool -- no per-base loop on the host, half the bytes over PCIe for the
 * reads.  Same results as lcd_batch_add_region_from_chunk. */
int lcd_batch_add_region_from_chunk_packed(lcd_batch_t *b, const lcd_read_view_t *chunk_reads, int64_t noisy_reg_beg,
                                    int64_t noisy_reg_end, int n_noisy_reg_reads, const int *noisy_reads,
                                    const uint8_t *ref_seq, int ref_seq_len);
int lcd_batch_upload(lcd_batch_t *b);  /* host -> HBM (not part of the timed hot path) */
int lcd_batch_run(lcd_batch_t *b);     /* anchors -> POA chains -> ref/cons WFA -> strings; inputs and outputs stay in HBM */
/* the same for n uploaded batches JOINTLY (one set of launches per stage over the jobs / chains of all of them: a chain is a
 * sequential object that occupies at most one CU, so several chunks' regions in flight are what fills 256 CUs -- the reference's
 * kt_for workers, src/call_var_main.c:321, are the natural source of such concurrent batches).  batches[0] lends its stream and work
 * buffers; every batch must be downloaded before batches[0] runs again or is destroyed.  Results are those of n separate lcd_batch_run calls.
 * Thread-safe for submissions with different batches[0]: two submitter threads with ~32 batches each is the measured optimum on one MI355X
 * (one submission's anchor / ref-cons / string stages run under the other's chains); a submission that exceeds the device
 * memory budget (LCD_MEM_FRACTION, default 0.92) is split in halves and retried. */
int lcd_batch_run_many(lcd_batch_t **batches, int n);
int lcd_batch_download(lcd_batch_t *b);/* HBM -> host */
/* ---- one process, all GPUs of the node (the GPU-side analogue of kt_for's work stealing, src/kthread.c:24-64, src/call_var_main.c:773) ----
 * lcd_dispatch_run orders the batches (job buffers created with LCD_DEVICE_ANY, regions added, not uploaded) by estimated DP work, longest first; one
 * submitter thread per device takes the next `coalesce` of them whenever its previous submission is done: bind + upload, lcd_batch_run_many, download.
 * Returns when every batch is downloaded (results through lcd_batch_region_result / _vars as usual); device_of[i] (optional) = the GPU batch i ran on.
 * Chunks are independent until stitch_var_main: no data-path exchange between devices. */
typedef struct lcd_dispatch_s lcd_dispatch_t;
lcd_dispatch_t *lcd_dispatch_create(int n_devices, const int *devices, int coalesce); /* n_devices <= 0: every visible GPU; coalesce <= 0: 16 */
void lcd_dispatch_destroy(lcd_dispatch_t *d);
int lcd_dispatch_n_devices(const lcd_dispatch_t *d);
int lcd_dispatch_run(lcd_dispatch_t *d, lcd_batch_t **batches, int n, int *device_of);
/* flags bit 0: no lcd_batch_download after a submission (results stay on the device) */
void lcd_dispatch_set_flags(lcd_dispatch_t *d, int flags);
/* per device (dispatcher order): ms its submitter thread spent inside submissions during the last lcd_dispatch_run, number of submissions; returns #devices */
int lcd_dispatch_busy(const lcd_dispatch_t *d, double *busy_ms, int *n_submissions);
double lcd_batch_cost(const lcd_batch_t *b);   /* the work estimate the queue is ordered by (DP cells of the batch's chains) */
/* longest-processing-time assignment of n costs to n_bins bins (static sharding across processes / ranks: bench.py --job-mb) */
void lcd_lpt_assign(int n, const double *cost, int n_bins, int *bin_of, double *bin_load);
/* ---- one process per GPU: cross-rank rebalancing of region queues (the reference's analogue inside one process is kt_for's work stealing,
 * src/kthread.c:24-64, called at src/call_var_main.c:773).  One epoch before the hot path: every rank packs its region jobs chunk by chunk
 * (lcd_region_jobs_pack: the arguments of lcd_batch_add_region, byte for byte) and prices them (lcd_region_job_cost); lcd_rebalance_exchange all-gathers the queue
 * depths and the (cost, bytes) tables over RCCL, computes the same plan on every rank (lcd_rebalance_plan) and moves WHOLE packed buffers with ncclSend / ncclRecv;
 * the receiver adds them to its batches with lcd_batch_add_packed.  No data-path collective (chunks are independent until stitch_var_main, src/collect_var.c:2983).
 * librccl is opened lazily by lcd_comm_create. */
typedef struct lcd_region_job_t {      /* the arguments of lcd_batch_add_region */
    int64_t reg_len; int n_reads;
    const int *read_ids, *lens; const uint8_t *const *seqs; const uint8_t *const *quals /* NULL, or NULL entries: zeros travel */;
    const int *fully_covers, *haps; const int64_t *phase_sets; const uint8_t *ref_seq; int ref_seq_len;
} lcd_region_job_t;
typedef struct lcd_move_t { int src, index /* in src's queue */, dst; } lcd_move_t;
typedef struct lcd_rebalance_stats_t { int n_moves, world; uint64_t moved_bytes; double imbalance_before, imbalance_after /* max load / mean load over the ranks */;
                                       double load_before_mine, load_after_mine; int jobs_before_mine, jobs_after_mine; } lcd_rebalance_stats_t;
typedef struct lcd_comm_s lcd_comm_t;
double lcd_region_job_cost(const lcd_region_job_t *job);                                  /* DP-cell estimate: banded K1 chains if any read is phased, else the unbanded K2 chain */
uint64_t lcd_region_jobs_pack(int n, const lcd_region_job_t *jobs, uint8_t *buf);         /* buf == NULL: the size; else fills buf (that many bytes) */
int lcd_batch_add_packed(lcd_batch_t *b, const uint8_t *buf, uint64_t nbytes);            /* -> regions added (lcd_batch_add_region each), < 0: malformed (lcd_rebalance_last_error) */
/* costs: every rank's job costs, rank after rank (n_jobs[r] each); moves: room for sum(n_jobs) entries.  From the most loaded rank to the least loaded one, the job
 * that brings the pair closest to equal -- or, when every job of the most loaded rank is at least as large as the gap (SV-heavy chunks), the swap of one job of
 * each whose difference does (two moves) -- until the most loaded rank is within tol of the mean or max_moves (< 0: no limit) are made; a job moves at most once.
 * Deterministic.  Returns the number of moves; load_before / load_after (world entries each, nullable). */
int lcd_rebalance_plan(int world, const int *n_jobs, const double *costs, double tol, int max_moves, lcd_move_t *moves, double *load_before, double *load_after);
int lcd_rccl_unique_id(uint8_t id[128]);                                                   /* ncclGetUniqueId on rank 0; the caller gives the bytes to the other ranks */
lcd_comm_t *lcd_comm_create(int world, int rank, const uint8_t id[128], int device);       /* ncclCommInitRank; NULL on failure */
void lcd_comm_destroy(lcd_comm_t *c);
int lcd_comm_info(lcd_comm_t *c, int *nccl_world, int *nccl_rank, int *nccl_device);      /* ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the communicator: what RCCL itself sees */
/* One epoch.  In: this rank's queue.  Out (arrays malloc()'d): its new queue -- kept jobs (bufs_out[i] is the caller's pointer, owned_out[i] = 0) and received ones
 * (malloc()'d buffers, owned_out[i] = 1: the caller frees them). */
int lcd_rebalance_exchange(lcd_comm_t *c, int n_jobs, const double *cost, const uint64_t *nbytes, const uint8_t *const *bufs, double tol,
                           int *n_out, double **cost_out, uint64_t **nbytes_out, uint8_t ***bufs_out, uint8_t **owned_out, lcd_rebalance_stats_t *stats);
const char *lcd_rebalance_last_error(void);
/* region results; clu_read_ids[c] and aln_strs[c][j].target_aln are malloc()'d (aln_strs[c] must hold 1+2*n_reads zeroed entries) */
int lcd_batch_region_result(lcd_batch_t *b, int region, int *clu_n_seqs, int **clu_read_ids, lcd_aln_str_t **aln_strs);
/* ---- candidate variants of a region + the read x variant allele profile (opt.collect_noisy_vars) ----
 * == make_vars_from_msa_cons_aln (src/collect_var.c:2279-2347: make_cand_vars_from_msa :1855, update_cand_var_profile_from_cons_aln_str1/2
 * :2164/:2206) computed on the device from the strings of lcd_batch_run.  What stays with the caller, as host code in the reference too:
 * TSD / polyA / TE annotation of gaps >= min_sv_len (collect_te_info_from_cons, :1815/:1834, SURVEY a14: lcd_collect_te_info_from_cons below, applied by the
 * caller to the INS / DEL records it gets) and merge_var_profile (:2712). */
typedef struct lcd_noisy_var_t {   /* the cand_var_t fields make_cand_vars0 (src/collect_var.c:1746) and the profile update fill */
    int64_t pos;
    int var_type, ref_len, alt_len; /* BAM_CDIFF 8 / BAM_CINS 1 / BAM_CDEL 2 */
    int cate;                       /* LONGCALLD_NOISY_CAND_HET_VAR 0x100 / LONGCALLD_NOISY_CAND_HOM_VAR 0x200 */
    int from_cons;                  /* var_from_cons_idx: 1 | 2 | 3 (:2216-2226) */
    int is_homopolymer_indel;       /* var_is_homopolymer_indel (:1720), from chunk_ref_seq; 0 for gaps >= min_sv_len */
    int ref_base, alt_ref_base;
    int total_cov, alle_covs[2];
    uint8_t *alt_seq;               /* malloc()'d alt_len bytes, NULL for deletions */
} lcd_noisy_var_t;
/* returns n_vars (>= 0) or < 0.  Rows of the profile = the reads of cluster 0 then cluster 1 in clu_read_ids order (row_read_ids);
 * prof_start/prof_end = read_var_profile_t.start_var_idx/end_var_idx (-1/-2: none); prof_alleles = n_rows x n_vars, -1 where unset.
 * chunk_ref_seq[k] = base code at reference position chunk_ref_beg + k (chunk->ref_seq / ref_beg); may be NULL (flag stays 0).
 * Every output array is malloc()'d (free vars[i].alt_seq, vars, row_read_ids, prof_*). */
int lcd_batch_region_vars(lcd_batch_t *b, int region, int64_t noisy_reg_beg, const uint8_t *chunk_ref_seq, int64_t chunk_ref_beg,
                          int64_t chunk_ref_len, lcd_noisy_var_t **vars, int *n_rows, int **row_read_ids, int **prof_start,
                          int **prof_end, int **prof_alleles);
int lcd_batch_region_sorted_ids(lcd_batch_t *b, int region, int *read_ids_out); /* the in-place permutation of noisy_reads */
/* per read of a region in its sorted order: chunk read id, cover flag and the read's slice read_reg_beg / read_reg_end as collect_noisy_read_info computed
 * them (src/align.c:1392-1458; -1 / -2 for regions added after slicing): what update_digars_from_aln_str (:1745) hands to lcd_update_digars_from_msa1 */
int lcd_batch_region_read_slices(lcd_batch_t *b, int region, int *read_ids, int *covers, int *read_beg, int *read_end);
int lcd_batch_get_stats(lcd_batch_t *b, lcd_batch_stats_t *st);
/* the K4 job set of the batch's anchor stage (edlib_xgaps calls of src/align.c:694,698,722): offsets into the batch's host pool (valid until the
 * batch is cleared); returns the number of jobs, fills at most cap of them.  bench.py times the reference's own edlib on it. */
int lcd_batch_k4_jobs(lcd_batch_t *b, int cap, uint64_t *t_off, int *tlen, uint64_t *q_off, int *qlen, const uint8_t **pool, uint64_t *pool_len);
/* a 64-bit FNV-1a digest over every region's results (n_cons, clusters, all alignment rows) -- cheap whole-batch parity check */
uint64_t lcd_batch_digest(lcd_batch_t *b);
/* lcd_batch_region_result for every region of a downloaded batch, results freed again: the host-side cost of holding every result the way the per-call
 * mirror hands it over (malloc()'d rows), without the digest's hashing; returns the bytes of alignment rows handed out */
uint64_t lcd_batch_materialize(lcd_batch_t *b);
/* Every region's results of a downloaded batch in ONE host block (additive entry; the reference's contract -- one malloc() per row, freed row by row,
 * src/collect_var.c:2670-2724 -- stays with lcd_batch_region_result).  *results_out[r] describes region r exactly as lcd_batch_region_result fills the caller's
 * arrays: n_cons, clu_n_seqs[2], clu_read_ids[2], aln_strs[2] (each an array of n_aln_strs = 1 + 2 * n_reads zero-initialised lcd_aln_str_t: [0] ref<->cons,
 * [2i+1] cons<->read i, [2i+2] ref<->read i).  The table, the id lists, the aln_str_t arrays and every row are INTERIOR pointers into *arena_out:
 * free(*arena_out) releases everything and nothing inside may be freed on its own.  Filled by host threads.  Returns the number of regions, < 0 on error. */
typedef struct lcd_region_result_t { int n_cons, n_aln_strs; int clu_n_seqs[2]; int *clu_read_ids[2]; lcd_aln_str_t *aln_strs[2]; } lcd_region_result_t;
int lcd_batch_region_results_arena(lcd_batch_t *b, lcd_region_result_t **results_out, void **arena_out, uint64_t *arena_bytes);

/* ---- SURVEY 8(f) f2, first part: EQX CIGARs -> digar lists + each read's noisy windows (additive) ----
 * == collect_digar_from_eqx_cigar (src/bam_utils.c:701-842, with push_xid_size_queue_win :161-200) for all reads of a chunk in one launch.
 * Inputs are what bam1_t holds: 0-based position, CIGAR words (bam_get_cigar), qualities; pal_flags bit0 / bit1 = is_ont_palindrome_clip
 * for the left / right clip (the caller reads the SA tag; 0 for HiFi).  Outputs (all malloc()'d, CSR over the reads):
 *   digars[digar_off[r] .. digar_off[r+1])   digar1_t fields (alt_seq copies are not made: bases stay in the packed read);
 *   ivs[iv_off[r] .. iv_off[r+1])            the read's digar->noisy_regs in cr_index order: start (= first position - 1), end, label;
 *   iv_in_chunk[k]                           1 if interval k is added to chunk->chunk_noisy_regs (read not skipped, overlaps [reg_beg, reg_end]);
 *   status[r] 0 / -1 (the function's return value: read skipped as too noisy) / -2 ('M' operation); beg/end = digar->beg/end; n_cand_vars. */
typedef struct lcd_digar_opt_t {
    int min_bq, noisy_reg_max_xgaps, noisy_reg_slide_win, end_clip_reg, end_clip_reg_flank_win; /* src/call_var_main.h:19-38: 10, 5, 100 | 25, 30, 100 */
    double max_noisy_frac_per_read, max_var_ratio_per_read;                                     /* 0.5, 0.05 */
} lcd_digar_opt_t;
typedef struct lcd_digar_t { int64_t pos; int type, len, qi, is_low_qual; } lcd_digar_t;
typedef struct lcd_noisy_iv_t { int64_t start, end; int label, pad; } lcd_noisy_iv_t;
void lcd_digar_opt_default(lcd_digar_opt_t *o, int is_ont);
int lcd_digar_batch(const lcd_digar_opt_t *opt, int n_reads, const int64_t *pos0, const uint32_t *cigar_pool, const uint64_t *cigar_off,
                    const int *n_cigar, const uint8_t *qual_pool, const uint64_t *qual_off, const int *qlen, const uint8_t *pal_flags,
                    int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, uint64_t **digar_off, lcd_digar_t **digars, uint64_t **iv_off,
                    lcd_noisy_iv_t **ivs, uint8_t **iv_in_chunk, int *status, int64_t *beg, int64_t *end, int *n_cand_vars);
/* the reference's three other digar sources (src/collect_var.c:1072-1079 picks one per read: EQX CIGAR, else cs tag, else MD tag, else the reference
 * bases); same outputs as lcd_digar_batch.
 *   lcd_digar_batch_tags, mode LCD_DIGAR_CS == collect_digar_from_cs_tag (src/bam_utils.c:844-1008): tags[r] = the read's cs:Z string (short or long
 *     form); clips come from the first / last CIGAR operation, with that function's own clip rule (:884-888, :969-972);
 *   lcd_digar_batch_tags, mode LCD_DIGAR_MD == collect_digar_from_MD_tag (:1010-1177): tags[r] = the MD:Z string, the CIGAR has 'M';
 *   lcd_digar_batch_ref == collect_digar_from_ref_seq (:1179-1328): seq_pool + seq_off[r] = bam_get_seq (4-bit bases), ref_seq[0] is reference position
 *     ref_beg (1-based), ref_end inclusive (chunk->ref_seq / ref_beg / ref_end); the base comparison runs on the device.
 * The tag strings are parsed on the host (they are as long as the read has events) into EQX-shaped operations for the same kernel.  A tag that does
 * not fit its CIGAR, an unknown cs character or '=' / 'X' next to an MD tag -- where the reference stops the program -- gives status[r] = -2. */
#define LCD_DIGAR_CS 1
#define LCD_DIGAR_MD 2
int lcd_digar_batch_tags(const lcd_digar_opt_t *opt, int mode, int n_reads, const int64_t *pos0, const uint32_t *cigar_pool, const uint64_t *cigar_off,
                         const int *n_cigar, const char *const *tags, const uint8_t *qual_pool, const uint64_t *qual_off, const int *qlen,
                         const uint8_t *pal_flags, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, uint64_t **digar_off, lcd_digar_t **digars,
                         uint64_t **iv_off, lcd_noisy_iv_t **ivs, uint8_t **iv_in_chunk, int *status, int64_t *beg, int64_t *end, int *n_cand_vars);
int lcd_digar_batch_ref(const lcd_digar_opt_t *opt, int n_reads, const int64_t *pos0, const uint32_t *cigar_pool, const uint64_t *cigar_off,
                        const int *n_cigar, const uint8_t *seq_pool, const uint64_t *seq_off, const uint8_t *qual_pool, const uint64_t *qual_off,
                        const int *qlen, const uint8_t *pal_flags, const char *ref_seq, int64_t ref_beg, int64_t ref_end, int64_t reg_beg,
                        int64_t reg_end, int64_t whole_ref_len, uint64_t **digar_off, lcd_digar_t **digars, uint64_t **iv_off, lcd_noisy_iv_t **ivs,
                        uint8_t **iv_in_chunk, int *status, int64_t *beg, int64_t *end, int *n_cand_vars);

/* ---- retrotransposon annotation of SV-size gaps (host code in the reference: src/align.c:32-163, src/kmer.c; host code here) ----
 * The variant builder calls collect_te_info_from_cons for every INS / DEL of at least min_sv_len bases (src/collect_var.c:1817,1834; :801 for a candidate
 * variant through collect_te_info_from_var); its results are cand_var_t.tsd_len / tsd_seq / tsd_pos1 / tsd_pos2 / polya_len / te_seq_i / te_is_rev.
 *   lcd_te_lib_create      == make_te_kmer_idx (src/kmer.c:120-150) without the file reading: per TE sequence the set of its overlapping k-mers and the set of
 *                             their reverse complements (kmer_len = opt->te_kmer_len, 15; 1..15), "simple" k-mers left out by the reference's own rule (:16-24);
 *   lcd_check_te_seq       == check_te_seq (:218-253): index of the TE sequence sharing most of seq's non-overlapping k-mers (>= 3), -1 if none; *is_rev
 *                             untouched when seq yields no k-mer;
 *   lcd_collect_te_info    == collect_te_info (src/align.c:32-83): returns the target-site-duplication length (0: not a TE candidate) and, if > 0, its bases
 *                             in tsd_seq (caller's buffer of opt->max_tsd_len bytes), its positions, the poly-A (> 0) / poly-T (< 0) length and the TE hit;
 *                             bases are codes 0-4; var_type 1 = BAM_CINS, 2 = BAM_CDEL; lib may be NULL (opt->n_te_seqs == 0);
 *   lcd_collect_te_info_from_cons == collect_te_info_from_cons (:139-163): the gap and the reference behind it taken from the consensus row / the chunk's reference
 *                             (ref_seq[0] = position ref_beg, letters or codes; outside [ref_beg, ref_end] = N); with cons_msa_seq = alt_seq and
 *                             msa_gap_start = 0 it is collect_te_info_from_var (:87-131). */
typedef struct lcd_te_lib_t lcd_te_lib_t;
typedef struct lcd_te_opt_t { int min_tsd_len, max_tsd_len, min_polya_len; float min_polya_ratio; } lcd_te_opt_t;   /* 2, 100, 10, 0.8 (src/call_var_main.h:55-58) */
void lcd_te_opt_default(lcd_te_opt_t *o);
lcd_te_lib_t *lcd_te_lib_create(int n_seqs, const char *const *seqs, const int *lens, int kmer_len);
void lcd_te_lib_destroy(lcd_te_lib_t *lib);
int lcd_te_lib_n_seqs(const lcd_te_lib_t *lib);
int lcd_check_te_seq(const lcd_te_lib_t *lib, const uint8_t *seq, int len, int *is_rev);
int lcd_collect_te_info(const lcd_te_opt_t *opt, const lcd_te_lib_t *lib, int var_type, const uint8_t *gap_seq, const uint8_t *flank_ref_seq, int gap_len,
                        int64_t gap_pos, uint8_t *tsd_seq, int64_t *tsd_pos1, int64_t *tsd_pos2, int *tsd_polya_len, int *te_seq_i, int *te_is_rev);
int lcd_collect_te_info_from_cons(const lcd_te_opt_t *opt, const lcd_te_lib_t *lib, const char *ref_seq, int64_t ref_beg, int64_t ref_end, int64_t gap_ref_start,
                                  int msa_gap_start, int var_type, int gap_len, const uint8_t *cons_msa_seq, uint8_t *tsd_seq, int64_t *tsd_pos1, int64_t *tsd_pos2,
                                  int *tsd_polya_len, int *te_seq_i, int *te_is_rev);

/* ---- SURVEY 8(f) f2 -> region jobs: collect_noisy_read_info's digar walk (src/align.c:1392-1456) for many (region, read) pairs in one launch ----
 * pair i = read pair_read[i] (index into digar_off / qlen) against the region [pair_reg_beg[i], pair_reg_end[i]] (1-based, flanks included); digars as
 * lcd_digar_batch returns them.  Out per pair: read_reg_beg / read_reg_end (the read's query interval over the region, src/align.c:1458) and the cover flag
 * (LONGCALLD_NOISY_{LEFT,RIGHT}_{COVER,GAP}; a deletion longer than noisy_reg_flank_len at a region end makes that end a gap). */
int lcd_region_read_slices_batch(int n_pairs, const int *pair_read, const int64_t *pair_reg_beg, const int64_t *pair_reg_end, int n_reads,
                                 const uint64_t *digar_off, const lcd_digar_t *digars, const int *qlen, int noisy_reg_flank_len,
                                 int *read_beg, int *read_end, int *cover);

/* ---- a DEVICE-RESIDENT chunk: records -> digars -> region slices -> a batch's read bases without host round trips ----
 * lcd_chunk_create uploads a chunk's reads ONCE (CIGARs, qualities, the records' 4-bit bases: seq_pool / seq_off as lcd_bam_load_region returns them), makes the
 * digars (lcd_digar_batch's kernel) and KEEPS them in HBM.  The host gets what its glue needs: lcd_chunk_read_info (status / digar->beg / end / #candidates /
 * #digars per read), lcd_chunk_intervals (the noisy windows, pointers into the handle), lcd_chunk_region_slices (per (region, read) pair the query interval and the
 * cover flag, computed on the digars in HBM) and lcd_batch_add_region_from_chunk_dev (a region job whose read bases are unpacked on the device from the chunk at
 * lcd_batch_upload).  No digar and no base crosses PCIe after lcd_chunk_create (lcd_copy_counters: [0] digar bytes D2H, [1] digar bytes H2D, [2] read-base bytes H2D,
 * [3] read-base bytes D2H since process start).  Results == the host path (lcd_digar_batch -> lcd_region_read_slices_batch -> lcd_batch_add_region_from_chunk). */
typedef struct lcd_chunk_s lcd_chunk_t;
struct lcd_bam_reads_t; /* (below: f3) */
lcd_chunk_t *lcd_chunk_create(const lcd_digar_opt_t *opt, int n_reads, const int64_t *pos0, const uint32_t *cigar_pool, const uint64_t *cigar_off, const int *n_cigar,
                              const uint8_t *qual_pool, const uint64_t *qual_off, const int *qlen, const uint8_t *pal_flags, const uint8_t *seq_pool,
                              const uint64_t *seq_off, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len);
/* f3 on the device, in front of the chunk: sam_itr_queryi + sam_itr_next (htslib: bgzf_read_block + inflate + bam_read1) and the record loop of
 * collect_ref_seq_bam_main (src/bam_utils.c:1672-1706), then collect_digar_from_eqx_cigar (:701-842), for one region of an indexed BAM.  The region's BGZF blocks
 * (the .bai's bins + linear index) are read from the file and uploaded COMPRESSED, inflated on the device (lcd_bgzf_inflate_dev's kernel), the records are found,
 * measured and filtered in HBM (tid, overlap with (reg_beg - 1, reg_end], BAM_FUNMAP / FSECONDARY / FSUPPLEMENTARY, MAPQ >= min_mapq, CG-tag CIGARs of reads with more
 * than 65 535 operations; file order; the loader's stop rule) and their digars made and kept there.  Bases and qualities are never moved: the region jobs unpack
 * bases from the inflated stream, the sampling rule of long regions (calc_read_error_rate, src/seq.c:429) runs on the qualities in HBM.  The host receives 80 bytes
 * per record and, with `meta`, the per-read scalars of lcd_bam_reads_t and the read names (cigar_pool / seq_pool / qual_pool stay NULL; seq_off / qual_off are
 * offsets of the device stream; free with lcd_bam_reads_free).  whole_ref_len = the contig's length in the BAM header; pal_flags = 0 (is_ont_palindrome_clip needs
 * the caller's clip test).  Result == lcd_bam_load_region_indexed + lcd_chunk_create on the same region.  A region without reads gives a chunk of 0 reads.
 * NULL on failure (lcd_last_error()); no host path: without a HIP device the call fails. */
lcd_chunk_t *lcd_chunk_create_from_bam(const lcd_digar_opt_t *opt, const char *bam_path, const char *bai_path, const char *chrom, int64_t reg_beg, int64_t reg_end,
                                       int min_mapq, int verify_crc, struct lcd_bam_reads_t *meta);
void lcd_chunk_destroy(lcd_chunk_t *c);
int lcd_chunk_n_reads(const lcd_chunk_t *c);
int lcd_chunk_read_info(const lcd_chunk_t *c, int *status, int64_t *beg, int64_t *end, int *n_cand_vars, int *n_digars);
int lcd_chunk_intervals(const lcd_chunk_t *c, const uint64_t **iv_off, const lcd_noisy_iv_t **ivs, const uint8_t **iv_in_chunk);
int lcd_chunk_region_slices(const lcd_chunk_t *c, int n_pairs, const int *pair_read, const int64_t *pair_reg_beg, const int64_t *pair_reg_end, int noisy_reg_flank_len,
                            int *read_beg, int *read_end, int *cover);
int lcd_batch_add_region_from_chunk_dev(lcd_batch_t *b, const lcd_chunk_t *c, int64_t reg_beg, int64_t reg_end, int n, const int *read_ids, const int *read_beg,
                                        const int *read_end, const int *cover, const int *haps, const int64_t *phase_sets, const uint8_t *ref_seq, int ref_seq_len);
void lcd_copy_counters(unsigned long long out[4]);

/* ---- SURVEY 8(f) f2, chunk level: pre_process_noisy_regs (src/collect_var.c:557-638) ----
 * chunk_noisy: the intervals cr_add()'ed to chunk->chunk_noisy_regs while the reads were loaded (lcd_digar_batch: ivs[k] with iv_in_chunk[k]), in
 * that order; low_comp: chunk->low_comp_cr as (start, end) pairs (sdust output, may be empty); reads in ordered_read_ids order with the skipped
 * ones left out: digar beg / end and each read's own noisy intervals (CSR, as lcd_digar_batch returns them).  Steps: cr_index, extension to the
 * overlapping low-complexity intervals (:538-553), cr_merge with the label-dependent distance (src/cgranges.c:225-300, twice as the reference
 * does), then per region the reads spanning it and the reads noisy in it (device), kept when noisy >= min_alt_dp and noisy / total >= min_af.
 * Returns the number of surviving regions; *regs_out is malloc()'d, in index order (start = first position - 1, end, label). */
int lcd_pre_process_noisy_regs(const lcd_noisy_iv_t *chunk_noisy, int n_noisy, const int64_t *low_comp, int n_low, int n_reads, const int64_t *read_beg,
                               const int64_t *read_end, const uint64_t *read_iv_off, const lcd_noisy_iv_t *read_ivs, int min_alt_dp, float min_af,
                               lcd_noisy_iv_t **regs_out);

/* cr_merge (src/cgranges.h:73, src/cgranges.c:289): the interval merge chunk_noisy_regs goes through (src/collect_var.c:552, :568 with a negative fixed window =
 * "the smaller of the two labels"; :657 with 0).  Pinned to the reference's own cgranges by tests/golden/cgranges_golden.json.  *out malloc()'d. */
int lcd_cr_merge(const lcd_noisy_iv_t *iv, int n, int fixed_merge_win, lcd_noisy_iv_t **out);

/* post_process_noisy_regs (src/collect_var.c:640-660, collect_noisy_reg_start_end :481-536): every region is grown by noisy_reg_flank_len and
 * further while a candidate variant (categories outside LONGCALLD_NOT_CAND_VAR_CATE, src/collect_var.h:28) sits within the flank, then overlapping /
 * touching regions are merged (cr_merge(cr, 0, -1, -1)).  Host code, as in the reference (a two-pointer walk over tens of regions); it closes the
 * region pipeline lcd_digar_batch -> lcd_pre_process_noisy_regs -> here.  regs in index order; vars in chunk order.  *regs_out malloc()'d. */
int lcd_post_process_noisy_regs(const lcd_noisy_iv_t *regs, int n_regs, int n_vars, const int64_t *var_pos, const int *var_ref_len, const int *var_cate,
                                int noisy_reg_flank_len, lcd_noisy_iv_t **regs_out);

/* ---- low-complexity intervals of a chunk's reference: sdust (src/sdust.c), as chunk->low_comp_cr is filled (src/bam_utils.c:1573-1581) ----
 * seq: raw codes 0..3 (4+ = N) or letters; T, W: LONGCALLD_SDUST_T 5 / LONGCALLD_SDUST_W 20 (src/call_var_main.h:82-83), W <= 64.
 * *intervals_out: malloc()'d (start, finish) pairs exactly as sdust() returns them (0-based, half-open); returns their number or < 0. */
int lcd_sdust(const uint8_t *seq, int64_t len, int T, int W, int64_t **intervals_out);
/* the same for the references of many chunks in ONE launch (the recommended form: a single sequence is latency-bound -- 21 ms per 500 kb chunk against 9 ms
 * for the reference's sdust() on one core -- while a pipeline step's worth of chunks shares that latency).  intervals_out[q] malloc()'d, n_out[q] pairs. */
int lcd_sdust_batch(int n_seqs, const uint8_t *const *seqs, const int64_t *lens, int T, int W, int64_t **intervals_out, int *n_out);

/* ---- cross-chunk stitching, genotype emission, tag values (host code in the reference and here: small and serial) ----
 * lcd_flip_variant_hap == flip_variant_hap + update_chunk_{var,read}_hap_phase_set1 (src/collect_var.c:1565-1680): the reads that overlap both chunks
 * vote (same haplotype in both: -1, different: +1); a non-zero score joins the phase sets (cur's smallest read PS becomes pre's largest) and, if
 * positive, swaps haplotypes 1 / 2 of the current chunk's variants (and reads, when update_reads: opt->out_aln_fp != NULL) in that phase set.
 * Overlap lists are the concatenation over the input BAMs of up_ovlp_read_i / down_ovlp_read_i (src/bam_utils.h:64-65). */
typedef struct lcd_chunk_phase_t {
    int tid, n_reads, n_vars;
    const int *ordered_read_ids;
    const uint8_t *is_skipped;
    int *haps; int64_t *phase_sets;                    /* in/out, n_reads */
    int64_t *var_phase_set; int *hap_to_cons_alle;     /* in/out, n_vars and n_vars*3 */
    int n_up_ovlp, n_down_ovlp;
    const int *up_ovlp_read_i, *down_ovlp_read_i;
    int flip_hap; int64_t flip_pre_PS, flip_cur_PS;    /* out (flip_hap starts at 0, src/bam_utils.c:1368) */
} lcd_chunk_phase_t;
int lcd_flip_variant_hap(lcd_chunk_phase_t *pre, lcd_chunk_phase_t *cur, int update_reads);  /* 0, or -6 when the overlap counts disagree (the reference exits) */
int lcd_stitch_chunks(lcd_chunk_phase_t *chunks, int n_chunks, int update_reads);            /* stitch_var_main, src/collect_var.c:2983-2989 */
/* lcd_make_variants == make_variants (src/collect_var.c:1465-1601), germline fields: candidate variants of the output categories inside [reg_beg, reg_end]
 * -> VCF-ready records.  p supplies the chunk state K5 works on (positions, types, categories, coverages, the read x variant profile, var_phase_set,
 * hap_to_cons_alle); var_ref_len / var_alt_len / alt_off + alt_pool / alt_ref_base are the remaining cand_var_t fields; ref_seq is chunk->ref_seq (letters
 * or codes, nst_nt4_table applies) starting at ref_beg.  Returns the number of records; *vars_out malloc()'d (free with lcd_free_variants). */
typedef struct lcd_call_opt_t { double log_p, log_1p, log_2; int max_gq, max_qual, min_sv_len, min_dp, min_alt_dp, out_amb_base; } lcd_call_opt_t;
typedef struct lcd_var1_t {          /* var1_t, src/call_var_main.h:108-121 (somatic fields left out) */
    int64_t pos, PS;
    int type, ref_len, n_alt_allele, alt_len[2];
    uint8_t *ref_bases, *alt_bases[2];
    int GT[2], DP, AD[3], QUAL, GQ, is_sv, is_clean, n_alt_reads; /* AD[2]: what the reference's formatter reads for a two-alt record -- var1_t has `int DP, AD[2]; uint8_t GT[2]`,
                                                                    * so its AD[2] is the third allele's coverage when there is one (the store at src/collect_var.c:1561 runs on), else the GT bytes */
    int *alt_read_i;
    int cand_i;                          /* the candidate (index into the lcd_hap_problem_t) the record was made from */
    int tsd_len, polya_len, te_seq_i, te_is_rev;   /* SURVEY a14 (var1_t's retrotransposon members): 0 / 0 / -1 / 0 from lcd_make_variants, filled by lcd_annotate_te */
    int64_t tsd_pos1, tsd_pos2;          /* -1 until then */
    uint8_t *tsd_seq;                    /* malloc()'d, tsd_len codes; freed by lcd_free_variants */
} lcd_var1_t;
void lcd_call_opt_default(lcd_call_opt_t *o);   /* src/call_var_main.c:156-157,209,217-219 */
int lcd_make_variants(const lcd_call_opt_t *opt, const lcd_hap_problem_t *p, const int *var_ref_len, const int *var_alt_len, const uint64_t *alt_off,
                      const uint8_t *alt_pool, const uint8_t *alt_ref_base, const char *ref_seq, int64_t ref_beg, int64_t reg_beg, int64_t reg_end,
                      lcd_var1_t **vars_out);
void lcd_free_variants(lcd_var1_t *vars, int n);
/* the VCF body lines write_var_to_vcf (src/vcf_utils.c:97-268) emits for these records (filters DP / AD / ambiguous bases applied); *text_out malloc()'d,
 * NUL-terminated; returns the number of lines */
int lcd_format_vcf(const lcd_call_opt_t *opt, const char *chrom, const lcd_var1_t *vars, int n_vars, char **text_out);
/* SURVEY a14 in the output records.  The reference computes the annotation when a candidate is made (collect_te_info_from_cons, src/collect_var.c:1817,1834) and copies
 * it into the record (:1504-1520); the values depend only on the candidate's own position, type, length and inserted bases, so here they are computed for the
 * finished records: every INS / DEL record whose gap (without the anchor base) has at least opt->min_sv_len bases.  te_lib may be NULL.  Returns the number of
 * records that got a target-site duplication.  lcd_format_vcf_te == lcd_format_vcf plus the INFO keys write_var_to_vcf adds for them (src/vcf_utils.c:184-195: MEI,
 * TSD, TSDLEN, POLYALEN, TSDPOS1, TSDPOS2, REPNAME = strand + te_names[te_seq_i]); te_names may be NULL (no REPNAME). */
int lcd_annotate_te(const lcd_call_opt_t *opt, const lcd_te_opt_t *te_opt, const lcd_te_lib_t *te_lib, const char *ref_seq, int64_t ref_beg, int64_t ref_end,
                    lcd_var1_t *vars, int n_vars);
int lcd_format_vcf_te(const lcd_call_opt_t *opt, const char *chrom, const lcd_var1_t *vars, int n_vars, const char *const *te_names, char **text_out);
/* HP / PS aux tags of write_processed_read_to_bam (src/bam_utils.c:1955-2006): HP:i is written iff hap != 0, PS:i iff phase set > 0 (an existing tag with
 * another value is replaced, one that should not be there is deleted): has_hp / has_ps say whether the record ends up carrying the tag */
void lcd_read_tags(int n_reads, const int *haps, const int64_t *phase_sets, uint8_t *has_hp, int *hp, uint8_t *has_ps, int64_t *ps);

/* ---- update_digars_from_msa1 (src/align.c:1701-1743; only with --refine-aln -b / -s, host code in the reference too) ----
 * rebuilds one read's digar list around a noisy region from its ref<->read alignment string (aln_strs[c][2k+2], opt.collect_ref_read_aln_str): the old
 * digars left / right of the region (collect_left_digars :1463, collect_right_digars :1500) around per-column digars of the string (collect_full / left /
 * right_msa_digars :1543-1699, by cover flag), joined with the reference's push rule (same_digar1, src/bam_utils.c:557).  digars as lcd_digar_batch returns
 * them (alt_seq is implicit: the read's bases [qi, qi + len)); read_beg / read_end = the read's slice of the region (collect_noisy_read_info's
 * read_reg_beg / read_reg_end).  Returns 0 and the new list (malloc()'d; *n_out may be 0), 1 when double_check_digar rejects the result (the read keeps its
 * old digars, as in the reference), 2 for a read that covers neither end (untouched). */
int lcd_update_digars_from_msa1(const lcd_digar_t *digars, int n_digar, int qlen, int msa_len, const uint8_t *ref_str, const uint8_t *read_str, int full_cover,
                                int64_t noisy_reg_beg, int64_t noisy_reg_end, int read_beg, int read_end, lcd_digar_t **out, int *n_out);

/* ---- the data formats in front of the path, without htslib (host code; lcd_io.cpp) ----
 * lcd_bam_load_region == the record loop of collect_ref_seq_bam_main (src/bam_utils.c:1672-1706) for one input BAM: reads of `chrom` overlapping
 * [reg_beg, reg_end] (1-based, i.e. sam_itr_queryi on (reg_beg - 1, reg_end]) that are mapped, primary (not BAM_FSECONDARY / BAM_FSUPPLEMENTARY) and of
 * MAPQ >= min_mapq (opt->min_mq, 30), in file order, as the flat arrays lcd_digar_batch (pos0, cigar_pool / cigar_off / n_cigar, qual_pool / qual_off, qlen)
 * and lcd_read_view_t (seq_pool + seq_off[r] = bam_get_seq: BAM 4-bit bases) take.  BGZF blocks are inflated in parallel on n_threads host threads
 * (0 = all); the region is a scan of the sorted file.  Returns n_reads or < 0 (lcd_io_last_error()); free with lcd_bam_reads_free. */
typedef struct lcd_bam_reads_t {
    int n_reads, tid, n_targets; int64_t target_len;
    int64_t *pos0, *end_pos;               /* bam1_core_t.pos ; bam_endpos (0-based, exclusive) */
    int *mapq, *flag, *n_cigar, *qlen;
    uint64_t *cigar_off; uint32_t *cigar_pool;   /* words, bam_get_cigar */
    uint64_t *seq_off; uint8_t *seq_pool;        /* bytes, bam_get_seq */
    uint64_t *qual_off; uint8_t *qual_pool;      /* bytes, bam_get_qual */
    uint64_t *name_off; char *name_pool;         /* NUL-terminated, bam_get_qname */
} lcd_bam_reads_t;
int lcd_bam_load_region(const char *bam_path, const char *chrom, int64_t reg_beg, int64_t reg_end, int min_mapq, int n_threads, lcd_bam_reads_t *out);
/* the same records through the BAM's .bai (bins of the region + the linear index's offset, SAM specification 5.2-5.3): only the BGZF blocks of the region's
 * chunks are read and inflated -- what sam_itr_queryi does for the reference (src/bam_utils.c:1673) */
int lcd_bam_load_region_indexed(const char *bam_path, const char *bai_path, const char *chrom, int64_t reg_beg, int64_t reg_end, int min_mapq, lcd_bam_reads_t *out);
void lcd_bam_reads_free(lcd_bam_reads_t *r);
/* BGZF blocks inflated ON THE DEVICE (inflate_kernel.hip) -- what bgzf_read_block + zlib's inflate do for the reference one block at a time on the calling thread
 * (htslib behind sam_itr_next, src/bam_utils.c:1673-1675).  `file` is an image of a BGZF file or of a run of whole blocks (e.g. a .bai chunk): the block table
 * comes from the BSIZE fields on the host (a hop per block), the compressed bytes are uploaded once, every block is one wavefront (dynamic / fixed / stored
 * deflate blocks, decode tables and the 32 KB history in LDS), and the inflated stream -- the blocks' outputs back to back, exactly the bytes bgzf_read
 * delivers -- stays in HBM: lcd_inflated_dev_ptr / lcd_inflated_size.  verify_crc != 0 checks every block's CRC-32 on the device (ISIZE is always checked).
 * Returns NULL on a malformed container / deflate stream / CRC mismatch (lcd_io_last_error() names the block) and when there is no HIP device: this entry
 * point has no host path (lcd_bam_load_region inflates on host threads).  lcd_inflated_to_host copies a range of the stream back (the record walk of a
 * loader, tests); lcd_inflated_kernel_ms / lcd_inflated_upload_ms: HIP-event times of the decode kernel and of the upload, for the measurement. */
typedef struct lcd_inflated_s lcd_inflated_t;
lcd_inflated_t *lcd_bgzf_inflate_dev(const uint8_t *file, size_t n, int verify_crc);
uint64_t lcd_inflated_dev_ptr(const lcd_inflated_t *h);
size_t lcd_inflated_size(const lcd_inflated_t *h);
size_t lcd_inflated_n_blocks(const lcd_inflated_t *h);
double lcd_inflated_kernel_ms(const lcd_inflated_t *h);
double lcd_inflated_upload_ms(const lcd_inflated_t *h);
int lcd_inflated_to_host(const lcd_inflated_t *h, size_t off, size_t n, uint8_t *out);
void lcd_inflated_free(lcd_inflated_t *h);
/* faidx_fetch_seq of chrom:[beg, end] (1-based inclusive, clipped to the contig) through <fa_path>.fai, as byte codes A0 C1 G2 T3 N4 (get_bam_chunk_reg_ref_seq0,
 * src/bam_utils.c:1558); returns the length, *codes_out malloc()'d */
int64_t lcd_fasta_fetch(const char *fa_path, const char *chrom, int64_t beg, int64_t end, uint8_t **codes_out);
/* the header lines write_vcf_header appends (src/vcf_utils.c:17-96) + the column line.  htslib's bcf_hdr_write decides the final order / ##fileformat line of
 * the real tool and is absent from the reference checkout: this text is not pinned (the body lines, lcd_format_vcf, are). */
int lcd_vcf_header(const char *source_version, const char *cmdline, const char *date_yyyymmdd, int n_contigs, const char *const *contig_names,
                   const int64_t *contig_lens, const char *sample_name, char **text_out);
const char *lcd_io_last_error(void);

/* ---- kernel-level batches (also what the per-call mirrors above run on) ---- */
int lcd_edlib_batch(int n, const uint8_t *pool, uint64_t pool_len, const uint64_t *q_off, const int *qlen,
                    const uint64_t *t_off, const int *tlen, int *dist, int *xgaps, int *n_eq, int *n_xid);
/* the same in HW (infix) mode; start / end (nullable): the target stretch [start, end] the path was taken on = edlib's startLocations[0] / endLocations[0] */
int lcd_edlib_batch_hw(int n, const uint8_t *pool, uint64_t pool_len, const uint64_t *q_off, const int *qlen,
                       const uint64_t *t_off, const int *tlen, int *dist, int *xgaps, int *n_eq, int *n_xid, int *start, int *end);
/* want bit0: cigars into cigars[i*cigar_stride ..], bit1: rows into rows[i*2*row_stride ..] (pattern row, then text row at +row_stride) */
int lcd_wfa_batch(int n, const uint8_t *pool, uint64_t pool_len, const uint64_t *p_off, const int *plen, const uint64_t *t_off,
                  const int *tlen, const int *gap_aln, int b, int q, int e, int q2, int e2, int want, int *score,
                  uint32_t *cigars, int cigar_stride, int *n_cigar, uint8_t *rows, int row_stride, int *aln_len);
/* bytes of device work arena one alignment of optimal score <= score_bound occupies: the decision bytes of one block of scores (1 B per diagonal,
 * 16 MB blocks) + the snapshots of the value ring, instead of 20 B x score^2 of retained wavefronts */
uint64_t lcd_wfa_arena_bytes(int plen, int tlen, int score_bound, int b, int q, int e, int q2, int e2);
/* POA chains with the anchor results supplied by the caller: mode 0 = K1 (src/align.c:762), 1 = K2 (:872).
 * anchors: 4 ints per read (ref_beg, ref_end, read_beg, read_end; 1-based).  Outputs are copied into caller arrays:
 * cons (2*cons_stride per chain), msa ((max_reads+2)*msa_stride per chain, row r at r*msa_stride), clu_ids (2*max_reads per chain). */
int lcd_poa_batch(const lcd_opt_t *opt, int n_chains, const int *mode, const int *chain_read0, const int *chain_n_reads,
                  int n_reads_total, const uint64_t *seq_off, const int *len, const int *skip, const int *anchors,
                  const uint8_t *pool, uint64_t pool_len, int *status, int *n_cons, int *cons_len, int *msa_len, int *clu_n,
                  uint8_t *cons, int cons_stride, uint8_t *msa, int msa_stride, int max_reads, int *clu_ids);

#ifdef __cplusplus
}
#endif
#endif
