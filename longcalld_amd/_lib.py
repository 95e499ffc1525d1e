"""ctypes loader for liblcd_hotpath.so (built in-tree by __graft_entry__.build / csrc/Makefile)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblcd_hotpath.so")


class LcdError(RuntimeError):
    pass


class LcdOpt(C.Structure):
    """lcd_opt_t == the call_var_opt_t fields the path reads (src/call_var_main.h:128-180)."""
    _fields_ = [(n, C.c_int) for n in ("match", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2", "gap_aln")] + [
        ("min_af", C.c_double), ("min_dp", C.c_int), ("partial_aln_ratio", C.c_double)] + [
        (n, C.c_int) for n in ("min_noisy_reg_size_to_sample_reads", "max_noisy_reg_len", "noisy_reg_flank_len",
                               "min_hap_full_reads", "min_hap_reads", "collect_ref_read_aln_str", "is_ont", "collect_noisy_vars", "min_sv_len")]


class LcdAlnStr(C.Structure):
    """lcd_aln_str_t == aln_str_t (src/collect_var.h:106-112)."""
    _fields_ = [("target_aln", C.POINTER(C.c_uint8)), ("query_aln", C.POINTER(C.c_uint8)), ("aln_len", C.c_int),
                ("target_beg", C.c_int), ("target_end", C.c_int), ("query_beg", C.c_int), ("query_end", C.c_int)]


class LcdRegionResult(C.Structure):
    """lcd_region_result_t (include/lcd_hotpath.h): one region of lcd_batch_region_results_arena"""
    _fields_ = [("n_cons", C.c_int), ("n_aln_strs", C.c_int), ("clu_n_seqs", C.c_int * 2), ("clu_read_ids", C.POINTER(C.c_int) * 2), ("aln_strs", C.POINTER(LcdAlnStr) * 2)]


class LcdDigar1(C.Structure):
    _fields_ = [("pos", C.c_int64), ("type", C.c_int), ("len", C.c_int), ("qi", C.c_int)]


class LcdNoisyVar(C.Structure):
    """lcd_noisy_var_t (include/lcd_hotpath.h): the cand_var_t fields of a noisy-region variant"""
    _fields_ = [("pos", C.c_int64)] + [(n, C.c_int) for n in ("var_type", "ref_len", "alt_len", "cate", "from_cons", "is_homopolymer_indel",
                                                               "ref_base", "alt_ref_base", "total_cov")] + [("alle_covs", C.c_int * 2),
                                                                                                              ("alt_seq", C.POINTER(C.c_uint8))]


class LcdDigarOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("min_bq", "noisy_reg_max_xgaps", "noisy_reg_slide_win", "end_clip_reg", "end_clip_reg_flank_win")] + [
        ("max_noisy_frac_per_read", C.c_double), ("max_var_ratio_per_read", C.c_double)]


class LcdDigar(C.Structure):
    _fields_ = [("pos", C.c_int64), ("type", C.c_int), ("len", C.c_int), ("qi", C.c_int), ("is_low_qual", C.c_int)]


class LcdNoisyIv(C.Structure):
    _fields_ = [("start", C.c_int64), ("end", C.c_int64), ("label", C.c_int), ("pad", C.c_int)]


class LcdBamReads(C.Structure):
    """lcd_bam_reads_t"""
    _fields_ = [("n_reads", C.c_int), ("tid", C.c_int), ("n_targets", C.c_int), ("target_len", C.c_int64), ("pos0", C.POINTER(C.c_int64)), ("end_pos", C.POINTER(C.c_int64)),
                ("mapq", C.POINTER(C.c_int)), ("flag", C.POINTER(C.c_int)), ("n_cigar", C.POINTER(C.c_int)), ("qlen", C.POINTER(C.c_int)), ("cigar_off", C.POINTER(C.c_uint64)),
                ("cigar_pool", C.POINTER(C.c_uint32)), ("seq_off", C.POINTER(C.c_uint64)), ("seq_pool", C.POINTER(C.c_uint8)), ("qual_off", C.POINTER(C.c_uint64)),
                ("qual_pool", C.POINTER(C.c_uint8)), ("name_off", C.POINTER(C.c_uint64)), ("name_pool", C.POINTER(C.c_char))]


class LcdReadView(C.Structure):
    _fields_ = [("digars", C.POINTER(LcdDigar1)), ("n_digar", C.c_int), ("qlen", C.c_int), ("bseq", C.POINTER(C.c_uint8)),
                ("qual", C.POINTER(C.c_uint8)), ("hap", C.c_int), ("phase_set", C.c_int64)]


class LcdBatchStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n_regions", "n_regions_resolved", "n_chains", "n_anchor_jobs", "n_wfa_jobs", "n_edlib_jobs")] + [
        (n, C.c_uint64) for n in ("poa_aligned_bases", "poa_cells", "wfa_offsets", "edlib_blocks", "poa_alg_bytes")] + [
        (n, C.c_double) for n in ("ms_total", "ms_anchor", "ms_poa", "ms_wfa", "ms_strings", "ms_upload", "ms_download", "ms_host", "ms_poa_kernel")] + [
        ("n_poa_launches", C.c_int), ("poa_retries", C.c_int), ("ms_vars", C.c_double), ("poa_cells_computed", C.c_uint64), ("poa_grown", C.c_int)]


class LcdHapProblem(C.Structure):
    """lcd_hap_problem_t: bam_chunk_t / cand_var_t / read_var_profile_t flattened for K5 (src/assign_hap.c:473)"""
    _i32p, _i64p, _u8p = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint8)
    _fields_ = [("n_reads", C.c_int), ("n_vars", C.c_int), ("is_ont", C.c_int), ("var_pos", _i64p), ("var_type", _i32p), ("var_cate", _i32p),
                ("is_homopolymer_indel", _i32p), ("total_cov", _i32p), ("alle_off", _i32p), ("alle_covs", _i32p), ("start_var_idx", _i32p),
                ("end_var_idx", _i32p), ("allele_off", _i32p), ("alleles", _i32p), ("ordered_read_ids", _i32p), ("is_skipped", _u8p),
                ("n_cr", C.c_int), ("cr_read", _i32p), ("haps", _i32p), ("phase_sets", _i64p), ("n_clean_agree_snps", _i32p),
                ("n_clean_conflict_snps", _i32p), ("var_phase_set", _i64p), ("hap_to_cons_alle", _i32p), ("hap_to_alle_profile", _i32p)]


_lib = None

# every symbol include/lcd_hotpath.h declares (tests check the .so exports all of them)
EXPORTS = [
    "lcd_opt_default", "lcd_init", "lcd_device_count", "lcd_alloc_events", "lcd_device_bytes", "lcd_set_thread_device", "lcd_batch_create_on", "lcd_last_error", "lcd_host_threads", "lcd_version", "lcd_wfa_end2end_aln", "lcd_edlib_end2end_aln",
    "lcd_edlib_xgaps", "lcd_edlib_edit_distance", "lcd_end2end_aln", "lcd_wfa_collect_diff_ins_seq", "lcd_edlib_infix_aln", "lcd_wfa_heuristic_aln", "lcd_collect_noisy_reg_aln_strs", "lcd_batch_create", "lcd_batch_destroy",
    "lcd_batch_clear", "lcd_batch_region_vars", "lcd_digar_opt_default", "lcd_digar_batch", "lcd_digar_batch_tags", "lcd_digar_batch_ref", "lcd_region_read_slices_batch", "lcd_te_opt_default", "lcd_te_lib_create", "lcd_te_lib_destroy", "lcd_te_lib_n_seqs", "lcd_check_te_seq", "lcd_collect_te_info", "lcd_collect_te_info_from_cons", "lcd_annotate_te", "lcd_format_vcf_te", "lcd_pre_process_noisy_regs", "lcd_post_process_noisy_regs", "lcd_cr_merge", "lcd_sdust", "lcd_sdust_batch", "lcd_batch_add_region", "lcd_batch_add_region_from_chunk", "lcd_batch_add_region_from_chunk_packed", "lcd_batch_upload", "lcd_batch_run", "lcd_batch_run_many",
    "lcd_batch_download", "lcd_dispatch_create", "lcd_dispatch_destroy", "lcd_dispatch_n_devices", "lcd_dispatch_run", "lcd_dispatch_set_flags", "lcd_dispatch_busy", "lcd_batch_cost", "lcd_lpt_assign", "lcd_batch_region_result", "lcd_batch_region_sorted_ids", "lcd_batch_region_read_slices", "lcd_batch_get_stats", "lcd_batch_k4_jobs", "lcd_chunk_create", "lcd_chunk_create_from_bam", "lcd_chunk_destroy", "lcd_chunk_n_reads", "lcd_chunk_read_info", "lcd_chunk_intervals", "lcd_chunk_region_slices", "lcd_batch_add_region_from_chunk_dev", "lcd_copy_counters", "lcd_batch_digest", "lcd_batch_materialize", "lcd_batch_region_results_arena",
    "lcd_edlib_batch", "lcd_edlib_batch_hw", "lcd_wfa_batch", "lcd_wfa_arena_bytes", "lcd_poa_batch", "lcd_assign_hap_germline", "lcd_assign_hap_batch", "lcd_flip_variant_hap", "lcd_stitch_chunks", "lcd_call_opt_default", "lcd_make_variants", "lcd_free_variants", "lcd_format_vcf", "lcd_read_tags", "lcd_update_digars_from_msa1", "lcd_bam_load_region", "lcd_bam_load_region_indexed", "lcd_bam_reads_free", "lcd_fasta_fetch", "lcd_vcf_header", "lcd_io_last_error",
    "lcd_region_job_cost", "lcd_region_jobs_pack", "lcd_batch_add_packed", "lcd_rebalance_plan", "lcd_rccl_unique_id", "lcd_comm_create", "lcd_comm_destroy", "lcd_comm_info", "lcd_rebalance_exchange", "lcd_rebalance_last_error",
    "lcd_bgzf_inflate_dev", "lcd_inflated_dev_ptr", "lcd_inflated_size", "lcd_inflated_n_blocks", "lcd_inflated_kernel_ms", "lcd_inflated_upload_ms", "lcd_inflated_to_host", "lcd_inflated_free",
]


def load_library():
    """Load liblcd_hotpath.so; raises LcdError (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LcdError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    u8p, i32p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    lib.lcd_last_error.restype = C.c_char_p
    lib.lcd_version.restype = C.c_char_p
    lib.lcd_host_threads.argtypes = [i32p, i32p, i32p, i32p]
    lib.lcd_opt_default.argtypes = [C.POINTER(LcdOpt)]
    lib.lcd_init.argtypes = [C.c_int]
    lib.lcd_batch_create.restype = C.c_void_p
    lib.lcd_batch_create.argtypes = [C.POINTER(LcdOpt)]
    lib.lcd_dispatch_create.restype = C.c_void_p
    lib.lcd_dispatch_create.argtypes = [C.c_int, i32p, C.c_int]
    lib.lcd_dispatch_destroy.argtypes = [C.c_void_p]
    lib.lcd_dispatch_destroy.restype = None
    lib.lcd_dispatch_n_devices.argtypes = [C.c_void_p]
    lib.lcd_dispatch_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, i32p]
    lib.lcd_batch_cost.restype = C.c_double
    lib.lcd_batch_cost.argtypes = [C.c_void_p]
    lib.lcd_lpt_assign.restype = None
    lib.lcd_lpt_assign.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int, i32p, C.POINTER(C.c_double)]
    lib.lcd_batch_region_read_slices.argtypes = [C.c_void_p, C.c_int, i32p, i32p, i32p, i32p]
    lib.lcd_batch_k4_jobs.argtypes = [C.c_void_p, C.c_int, u64p, i32p, u64p, i32p, C.POINTER(u8p), u64p]
    lib.lcd_end2end_aln.argtypes = [C.POINTER(LcdOpt), C.c_char_p, C.c_int, u8p, C.c_int, C.POINTER(u32p)]
    lib.lcd_wfa_collect_diff_ins_seq.argtypes = [C.POINTER(LcdOpt), u8p, C.c_int, u8p, C.c_int, C.POINTER(u8p)]
    lib.lcd_edlib_infix_aln.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p]
    lib.lcd_wfa_heuristic_aln.argtypes = [u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 6 + [i32p, i32p]
    lib.lcd_wfa_arena_bytes.restype = C.c_uint64
    lib.lcd_wfa_arena_bytes.argtypes = [C.c_int] * 8
    lib.lcd_batch_create_on.restype = C.c_void_p
    lib.lcd_batch_create_on.argtypes = [C.POINTER(LcdOpt), C.c_int]
    lib.lcd_set_thread_device.argtypes = [C.c_int]
    lib.lcd_alloc_events.restype = C.c_longlong
    lib.lcd_device_bytes.restype = C.c_longlong
    lib.lcd_device_bytes.argtypes = [C.c_int]
    lib.lcd_batch_destroy.argtypes = [C.c_void_p]
    lib.lcd_batch_destroy.restype = None
    lib.lcd_batch_clear.argtypes = [C.c_void_p]
    lib.lcd_batch_clear.restype = None
    lib.lcd_batch_add_region.argtypes = [C.c_void_p, C.c_int64, C.c_int, i32p, i32p, C.POINTER(u8p), C.POINTER(u8p), i32p, i32p,
                                         C.POINTER(C.c_int64), u8p, C.c_int]
    lib.lcd_batch_add_region_from_chunk.argtypes = [C.c_void_p, C.POINTER(LcdReadView), C.c_int64, C.c_int64, C.c_int, i32p, u8p, C.c_int]
    lib.lcd_batch_add_region_from_chunk_packed.argtypes = lib.lcd_batch_add_region_from_chunk.argtypes
    for f in ("lcd_batch_upload", "lcd_batch_run", "lcd_batch_download"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.lcd_batch_run_many.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.lcd_batch_region_result.argtypes = [C.c_void_p, C.c_int, i32p, C.POINTER(i32p), C.POINTER(C.POINTER(LcdAlnStr))]
    lib.lcd_batch_region_vars.argtypes = [C.c_void_p, C.c_int, C.c_int64, u8p, C.c_int64, C.c_int64, C.POINTER(C.POINTER(LcdNoisyVar)), i32p,
                                          C.POINTER(i32p), C.POINTER(i32p), C.POINTER(i32p), C.POINTER(i32p)]
    i64p, u64p_ = C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    lib.lcd_digar_opt_default.argtypes = [C.POINTER(LcdDigarOpt), C.c_int]
    lib.lcd_digar_opt_default.restype = None
    lib.lcd_digar_batch.argtypes = [C.POINTER(LcdDigarOpt), C.c_int, i64p, C.POINTER(C.c_uint32), u64p_, i32p, u8p, u64p_, i32p, u8p, C.c_int64, C.c_int64, C.c_int64,
                                    C.POINTER(u64p_), C.POINTER(C.POINTER(LcdDigar)), C.POINTER(u64p_), C.POINTER(C.POINTER(LcdNoisyIv)), C.POINTER(u8p), i32p, i64p, i64p, i32p]
    _dg_head = [C.POINTER(LcdDigarOpt), C.c_int, i64p, C.POINTER(C.c_uint32), u64p_, i32p]
    _dg_tail = [C.c_int64, C.c_int64, C.c_int64, C.POINTER(u64p_), C.POINTER(C.POINTER(LcdDigar)), C.POINTER(u64p_), C.POINTER(C.POINTER(LcdNoisyIv)), C.POINTER(u8p), i32p, i64p, i64p, i32p]
    lib.lcd_digar_batch_tags.argtypes = [_dg_head[0], C.c_int] + _dg_head[1:] + [C.POINTER(C.c_char_p), u8p, u64p_, i32p, u8p] + _dg_tail
    lib.lcd_digar_batch_ref.argtypes = _dg_head + [u8p, u64p_, u8p, u64p_, i32p, u8p, C.c_char_p, C.c_int64, C.c_int64] + _dg_tail
    lib.lcd_region_read_slices_batch.argtypes = [C.c_int, i32p, i64p, i64p, C.c_int, u64p_, C.POINTER(LcdDigar), i32p, C.c_int, i32p, i32p, i32p]
    lib.lcd_pre_process_noisy_regs.argtypes = [C.POINTER(LcdNoisyIv), C.c_int, i64p, C.c_int, C.c_int, i64p, i64p, u64p_, C.POINTER(LcdNoisyIv), C.c_int, C.c_float,
                                               C.POINTER(C.POINTER(LcdNoisyIv))]
    lib.lcd_post_process_noisy_regs.argtypes = [C.POINTER(LcdNoisyIv), C.c_int, C.c_int, i64p, i32p, i32p, C.c_int, C.POINTER(C.POINTER(LcdNoisyIv))]
    lib.lcd_sdust.argtypes = [u8p, C.c_int64, C.c_int, C.c_int, C.POINTER(i64p)]
    lib.lcd_batch_region_sorted_ids.argtypes = [C.c_void_p, C.c_int, i32p]
    lib.lcd_batch_get_stats.argtypes = [C.c_void_p, C.POINTER(LcdBatchStats)]
    lib.lcd_batch_digest.argtypes = [C.c_void_p]
    lib.lcd_batch_digest.restype = C.c_uint64
    lib.lcd_batch_materialize.argtypes = [C.c_void_p]
    lib.lcd_batch_materialize.restype = C.c_uint64
    lib.lcd_edlib_batch.argtypes = [C.c_int, u8p, C.c_uint64, u64p, i32p, u64p, i32p, i32p, i32p, i32p, i32p]
    lib.lcd_edlib_batch_hw.argtypes = [C.c_int, u8p, C.c_uint64, u64p, i32p, u64p, i32p, i32p, i32p, i32p, i32p, i32p, i32p]
    lib.lcd_wfa_batch.argtypes = [C.c_int, u8p, C.c_uint64, u64p, i32p, u64p, i32p, i32p] + [C.c_int] * 6 + [i32p, u32p, C.c_int, i32p, u8p, C.c_int, i32p]
    lib.lcd_poa_batch.argtypes = [C.POINTER(LcdOpt), C.c_int, i32p, i32p, i32p, C.c_int, u64p, i32p, i32p, i32p, u8p, C.c_uint64, i32p,
                                  i32p, i32p, i32p, i32p, u8p, C.c_int, u8p, C.c_int, C.c_int, i32p]
    lib.lcd_assign_hap_germline.argtypes = [C.POINTER(LcdHapProblem), C.c_int]
    lib.lcd_assign_hap_batch.argtypes = [C.c_int, C.POINTER(LcdHapProblem), i32p]
    lib.lcd_wfa_end2end_aln.argtypes = [u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 8 + [C.POINTER(u32p), i32p, C.POINTER(u8p), C.POINTER(u8p), i32p]
    lib.lcd_edlib_end2end_aln.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p]
    lib.lcd_edlib_xgaps.argtypes = [u8p, C.c_int, u8p, C.c_int]
    lib.lcd_edlib_edit_distance.argtypes = [u8p, C.c_int, u8p, C.c_int]
    lib.lcd_collect_noisy_reg_aln_strs.argtypes = [C.POINTER(LcdOpt), C.POINTER(LcdReadView), C.c_int64, C.c_int64, C.c_int, C.c_int, i32p,
                                                   u8p, C.c_int, i32p, C.POINTER(i32p), C.POINTER(C.POINTER(LcdAlnStr))]
    _lib = lib
    return lib


def check(rc, lib=None):
    if rc < 0:
        lib = lib or load_library()
        raise LcdError(f"liblcd_hotpath error {rc}: {lib.lcd_last_error().decode()}")
    return rc
