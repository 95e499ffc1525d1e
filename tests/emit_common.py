"""ctypes drivers for SURVEY 8(f) f4 (stitching, genotype records, VCF text): the product library (liblcd_hotpath.so, prefix lcd_) and the oracle
(liblcd_oracle.so, prefix lcdo_) declare the same flattened structs, so one driver serves both -- the tests compare their outputs."""
import ctypes as C

import numpy as np

i32p, i64p, u8p, u64p = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class CallOpt(C.Structure):
    _fields_ = [("log_p", C.c_double), ("log_1p", C.c_double), ("log_2", C.c_double), ("max_gq", C.c_int), ("max_qual", C.c_int), ("min_sv_len", C.c_int),
                ("min_dp", C.c_int), ("min_alt_dp", C.c_int), ("out_amb_base", C.c_int)]


class Var1(C.Structure):
    _fields_ = [("pos", C.c_int64), ("PS", C.c_int64), ("type", C.c_int), ("ref_len", C.c_int), ("n_alt_allele", C.c_int), ("alt_len", C.c_int * 2),
                ("ref_bases", u8p), ("alt_bases", u8p * 2), ("GT", C.c_int * 2), ("DP", C.c_int), ("AD", C.c_int * 3), ("QUAL", C.c_int), ("GQ", C.c_int),
                ("is_sv", C.c_int), ("is_clean", C.c_int), ("n_alt_reads", C.c_int), ("alt_read_i", i32p),
                ("cand_i", C.c_int), ("tsd_len", C.c_int), ("polya_len", C.c_int), ("te_seq_i", C.c_int), ("te_is_rev", C.c_int), ("tsd_pos1", C.c_int64), ("tsd_pos2", C.c_int64),
                ("tsd_seq", u8p)]


class TeOpt(C.Structure):
    _fields_ = [("min_tsd_len", C.c_int), ("max_tsd_len", C.c_int), ("min_polya_len", C.c_int), ("min_polya_ratio", C.c_float)]


class ChunkPhase(C.Structure):
    _fields_ = [("tid", C.c_int), ("n_reads", C.c_int), ("n_vars", C.c_int), ("ordered_read_ids", i32p), ("is_skipped", u8p), ("haps", i32p), ("phase_sets", i64p),
                ("var_phase_set", i64p), ("hap_to_cons_alle", i32p), ("n_up_ovlp", C.c_int), ("n_down_ovlp", C.c_int), ("up_ovlp_read_i", i32p),
                ("down_ovlp_read_i", i32p), ("flip_hap", C.c_int), ("flip_pre_PS", C.c_int64), ("flip_cur_PS", C.c_int64)]


def default_call_opt():
    o = CallOpt()
    o.log_p, o.log_1p, o.log_2 = -3.0, float(np.log10(1 - 0.001)), 0.301023      # src/call_var_main.c:217
    o.max_gq, o.max_qual, o.min_sv_len, o.min_dp, o.min_alt_dp, o.out_amb_base = 60, 60, 30, 5, 2, 0
    return o


def make_variants(lib, prefix, hap_struct, opt, extra, ref_seq, ref_beg, reg_beg, reg_end, te=None):
    """-> (records as dicts, VCF text); `hap_struct` is the filled lcd(o)_hap_problem_t, `extra` = dict(var_ref_len, var_alt_len, alt_off, alt_pool, alt_ref_base);
    te = dict(lib=<TE library handle of this side or None>, names=[bytes]): the records are annotated and written with the TE keys"""
    fn = getattr(lib, prefix + "make_variants")
    vp = C.POINTER(Var1)()
    keep = [np.ascontiguousarray(extra["var_ref_len"], np.int32), np.ascontiguousarray(extra["var_alt_len"], np.int32), np.ascontiguousarray(extra["alt_off"], np.uint64),
            np.ascontiguousarray(extra["alt_pool"], np.uint8), np.ascontiguousarray(extra["alt_ref_base"], np.uint8)]
    fn.restype = C.c_int
    n = fn(C.byref(opt), C.byref(hap_struct), keep[0].ctypes.data_as(i32p), keep[1].ctypes.data_as(i32p), keep[2].ctypes.data_as(u64p), keep[3].ctypes.data_as(u8p),
           keep[4].ctypes.data_as(u8p), C.c_char_p(ref_seq), C.c_int64(ref_beg), C.c_int64(reg_beg), C.c_int64(reg_end), C.byref(vp))
    assert n >= 0, n
    if te is not None:
        ann = getattr(lib, prefix + "annotate_te"); ann.restype = C.c_int
        ref_end = ref_beg + len(ref_seq) - 1
        if prefix == "lcd_":
            to = TeOpt(2, 100, 10, 0.8)
            ann.argtypes = [C.POINTER(CallOpt), C.POINTER(TeOpt), C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(Var1), C.c_int]
            n_ann = ann(C.byref(opt), C.byref(to), te["lib"], ref_seq, ref_beg, ref_end, vp, n)
        else:
            ann.argtypes = [C.POINTER(CallOpt), C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(Var1), C.c_int]
            n_ann = ann(C.byref(opt), 2, 100, 10, 0.8, te["lib"], ref_seq, ref_beg, ref_end, vp, n)
        te["n_annotated"] = n_ann
    recs = []
    for i in range(n):
        v = vp[i]
        recs.append(dict(cand_i=v.cand_i, tsd=bytes(v.tsd_seq[j] for j in range(v.tsd_len)), polya_len=v.polya_len, te_seq_i=v.te_seq_i, te_is_rev=v.te_is_rev,
                         tsd_pos1=v.tsd_pos1, tsd_pos2=v.tsd_pos2, pos=v.pos, PS=v.PS, type=v.type, ref_len=v.ref_len, n_alt=v.n_alt_allele, alt_len=list(v.alt_len)[:v.n_alt_allele],
                         ref=bytes(v.ref_bases[j] for j in range(v.ref_len)), alt=[bytes(v.alt_bases[a][j] for j in range(v.alt_len[a])) for a in range(v.n_alt_allele)],
                         GT=list(v.GT), DP=v.DP, AD=list(v.AD), QUAL=v.QUAL, GQ=v.GQ, is_sv=v.is_sv, is_clean=v.is_clean,
                         alt_reads=[v.alt_read_i[j] for j in range(v.n_alt_reads)]))
    tp = C.c_char_p()
    tptr = C.c_void_p()
    if te is None:
        fmt = getattr(lib, prefix + "format_vcf")
        fmt.restype = C.c_int
        n_lines = fmt(C.byref(opt), C.c_char_p(b"chr11"), vp, n, C.byref(tptr))
    else:
        fmt = getattr(lib, prefix + "format_vcf_te")
        fmt.restype = C.c_int
        names = (C.c_char_p * max(len(te["names"]), 1))(*te["names"])
        n_lines = fmt(C.byref(opt), C.c_char_p(b"chr11"), vp, n, names, C.byref(tptr))
    text = C.string_at(tptr).decode()
    _libc.free(tptr)
    fr = getattr(lib, prefix + "free_variants")
    fr.restype = None
    fr(vp, n)
    assert text.count("\n") == n_lines
    return recs, text


def flip(lib, prefix, pre, cur, update_reads):
    """pre / cur: dicts with tid, ordered_read_ids, is_skipped, haps, phase_sets, var_phase_set, hap_to_cons_alle, up_ovlp, down_ovlp (numpy arrays, mutated)"""
    def fill(d):
        s = ChunkPhase()
        s.tid, s.n_reads, s.n_vars = d["tid"], len(d["haps"]), len(d["var_phase_set"])
        s.ordered_read_ids = d["ordered_read_ids"].ctypes.data_as(i32p); s.is_skipped = d["is_skipped"].ctypes.data_as(u8p)
        s.haps = d["haps"].ctypes.data_as(i32p); s.phase_sets = d["phase_sets"].ctypes.data_as(i64p)
        s.var_phase_set = d["var_phase_set"].ctypes.data_as(i64p); s.hap_to_cons_alle = d["hap_to_cons_alle"].ctypes.data_as(i32p)
        s.n_up_ovlp, s.n_down_ovlp = len(d["up_ovlp"]), len(d["down_ovlp"])
        s.up_ovlp_read_i = d["up_ovlp"].ctypes.data_as(i32p); s.down_ovlp_read_i = d["down_ovlp"].ctypes.data_as(i32p)
        s.flip_hap, s.flip_pre_PS, s.flip_cur_PS = 0, -7, -7
        return s
    a, b = fill(pre), fill(cur)
    fn = getattr(lib, prefix + "flip_variant_hap")
    fn.restype = C.c_int
    rc = fn(C.byref(a), C.byref(b), int(update_reads))
    return rc, (b.flip_hap, b.flip_pre_PS, b.flip_cur_PS)
