"""ctypes binding of the CPU oracle (oracle/liblcd_oracle.so) and the reference's edlib (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under longcalld_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
u8p, i32p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_uint32)
_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class Opt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("match", "mismatch", "gap_open1", "gap_ext1", "gap_open2", "gap_ext2", "gap_aln")] + [
        ("min_af", C.c_double), ("min_dp", C.c_int), ("partial_aln_ratio", C.c_double)] + [
        (n, C.c_int) for n in ("min_noisy_reg_size_to_sample_reads", "max_noisy_reg_len", "noisy_reg_flank_len",
                               "min_hap_full_reads", "min_hap_reads", "collect_ref_read_aln_str", "is_ont")]


class PoaRes(C.Structure):
    _fields_ = [("n_cons", C.c_int), ("cons_len", C.c_int * 2), ("cons_seq", u8p * 2), ("clu_n_seq", C.c_int * 2), ("clu_read_ids", i32p * 2),
                ("n_seq", C.c_int), ("msa_len", C.c_int), ("msa", C.POINTER(u8p))]


class AlnStr(C.Structure):
    _fields_ = [("target_aln", u8p), ("query_aln", u8p), ("aln_len", C.c_int), ("target_beg", C.c_int), ("target_end", C.c_int),
                ("query_beg", C.c_int), ("query_end", C.c_int)]


class RegionReads(C.Structure):
    _fields_ = [("n_reads", C.c_int), ("read_ids", i32p), ("lens", i32p), ("seqs", C.POINTER(u8p)), ("quals", C.POINTER(u8p)),
                ("fully_covers", i32p), ("haps", i32p), ("phase_sets", C.POINTER(C.c_int64))]


_lib = None
_ref = None


def build(force=False):
    so = os.path.join(_HERE, "liblcd_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.lcdo_edlib_nw.argtypes = [u8p, C.c_int, u8p, C.c_int, C.POINTER(u8p), i32p]
        L.lcdo_edlib_xgaps.argtypes = [u8p, C.c_int, u8p, C.c_int]
        L.lcdo_edlib_end2end_aln.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p]
        L.lcdo_wfa_end2end_aln.argtypes = [u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 6 + [C.POINTER(u32p), i32p, C.POINTER(u8p), C.POINTER(u8p), i32p, i32p]
        L.lcdo_gotoh2p_score.argtypes = [u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 5
        L.lcdo_cigar_score2p.argtypes = [u32p, C.c_int, u8p, C.c_int, u8p, C.c_int] + [C.c_int] * 5
        L.lcdo_poa_partial_aln_msa_cons.argtypes = [C.POINTER(Opt), C.c_int, C.c_int, C.POINTER(u8p), i32p, i32p, C.POINTER(PoaRes)]
        L.lcdo_poa_aln_msa_cons.argtypes = [C.POINTER(Opt), C.c_int, C.POINTER(u8p), i32p, C.c_int, C.POINTER(PoaRes)]
        L.lcdo_collect_partial_aln_beg_end.argtypes = [C.POINTER(Opt), C.c_int, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
        L.lcdo_collect_noisy_reg_aln_strs.argtypes = [C.POINTER(Opt), C.c_int64, C.POINTER(RegionReads), u8p, C.c_int, i32p, C.POINTER(i32p),
                                                      C.POINTER(C.POINTER(AlnStr))]
    return _lib


def ref_edlib():
    """the reference's own vendored edlib, compiled by oracle/Makefile into oracle/_ref (None if not built)"""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libedlib_ref.so")
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        _ref.ref_edlib_nw_path.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int, i32p]
        _ref.ref_edlib_distance.argtypes = [u8p, C.c_int, u8p, C.c_int]
        _ref.ref_edlib_hw_path.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int, i32p, i32p, i32p, i32p]
    return _ref


def default_opt():
    o = Opt()
    lib().lcdo_opt_default(C.byref(o))
    return o


def _p(a):
    return a.ctypes.data_as(u8p)


def _c8(a):
    return np.ascontiguousarray(a, np.uint8)


def edlib_nw(query, target):
    """-> (distance, ops uint8 array) with edlib's exact path"""
    q, t = _c8(query), _c8(target)
    ap, n = u8p(), C.c_int()
    d = lib().lcdo_edlib_nw(_p(q), len(q), _p(t), len(t), C.byref(ap), C.byref(n))
    ops = np.ctypeslib.as_array(ap, shape=(max(n.value, 1),))[:n.value].copy() if n.value else np.zeros(0, np.uint8)
    if ap:
        _libc.free(ap)
    return d, ops


_ref_cg = None


def ref_cgranges():
    """the reference's own src/cgranges.c, compiled by oracle/Makefile into oracle/_ref (None if not built)"""
    global _ref_cg
    if _ref_cg is None:
        p = os.path.join(_HERE, "_ref", "libcgranges_ref.so")
        if not os.path.exists(p):
            return None
        _ref_cg = C.CDLL(p)
        _ref_cg.ref_cr_sorted_labels.argtypes = [C.c_int, i32p, i32p, i32p]
        _ref_cg.ref_cr_sorted_labels.restype = None
        _ref_cg.ref_cr_overlap_labels.argtypes = [C.c_int, i32p, i32p, C.c_int, C.c_int, i32p, C.c_int]
    return _ref_cg


def ref_cr_sorted_order(st, en):
    st = np.ascontiguousarray(st, np.int32); en = np.ascontiguousarray(en, np.int32)
    out = np.zeros(len(st), np.int32)
    ref_cgranges().ref_cr_sorted_labels(len(st), st.ctypes.data_as(i32p), en.ctypes.data_as(i32p), out.ctypes.data_as(i32p))
    return out


def ref_cr_overlap(st, en, qst, qen):
    st = np.ascontiguousarray(st, np.int32); en = np.ascontiguousarray(en, np.int32)
    out = np.zeros(len(st) + 1, np.int32)
    k = ref_cgranges().ref_cr_overlap_labels(len(st), st.ctypes.data_as(i32p), en.ctypes.data_as(i32p), int(qst), int(qen), out.ctypes.data_as(i32p), len(out))
    return out[:k].copy()


def ref_pre_process_noisy_regs(chunk_noisy, low_comp, read_beg, read_end, read_ivs, min_alt_dp=2, min_af=0.2, merge_dis=500, min_sv_len=30):
    """pre_process_noisy_regs with the REFERENCE's cgranges doing every interval operation (oracle/ref_cgranges_shim.c)"""
    L = ref_cgranges()
    L.ref_pre_process_noisy_regs.argtypes = [C.c_int, i32p, C.c_int, i32p, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_float,
                                             i32p, C.c_int]
    cn = np.ascontiguousarray(np.asarray(chunk_noisy, np.int64).reshape(-1, 3), np.int32)
    lc = np.ascontiguousarray(np.asarray(low_comp, np.int64).reshape(-1, 2), np.int32)
    rb = np.ascontiguousarray(read_beg, np.int64); re_ = np.ascontiguousarray(read_end, np.int64)
    off = np.concatenate([[0], np.cumsum([len(x) for x in read_ivs])]).astype(np.int32) if len(rb) else np.zeros(1, np.int32)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(x, np.int64).reshape(-1, 3)[:, :2] for x in read_ivs] + [np.zeros((0, 2), np.int64)]), np.int32) if len(rb) else np.zeros((1, 2), np.int32)
    if len(flat) == 0:
        flat = np.zeros((1, 2), np.int32)
    out = np.zeros(3 * (len(cn) + 1), np.int32)
    ll = C.POINTER(C.c_longlong)
    n = L.ref_pre_process_noisy_regs(len(cn), cn.ctypes.data_as(i32p), len(lc), (lc if len(lc) else np.zeros((1, 2), np.int32)).ctypes.data_as(i32p), len(rb),
                                     (rb if len(rb) else np.zeros(1, np.int64)).ctypes.data_as(ll), (re_ if len(rb) else np.zeros(1, np.int64)).ctypes.data_as(ll),
                                     off.ctypes.data_as(i32p), flat.ctypes.data_as(i32p), int(merge_dis), int(min_sv_len), int(min_alt_dp), float(min_af), out.ctypes.data_as(i32p), len(cn) + 1)
    return out[:3 * n].reshape(-1, 3).astype(np.int64)


def ref_post_process_noisy_regs(regs, var_pos, var_ref_len, var_cate, flank=10):
    L = ref_cgranges()
    L.ref_post_process_noisy_regs.argtypes = [C.c_int, i32p, C.c_int, i32p, i32p, i32p, C.c_int, i32p, C.c_int]
    r = np.ascontiguousarray(np.asarray(regs, np.int64).reshape(-1, 3), np.int32)
    vp = np.ascontiguousarray(var_pos, np.int32); vl = np.ascontiguousarray(var_ref_len, np.int32); vc = np.ascontiguousarray(var_cate, np.int32)
    nv = len(vp)
    if nv == 0:
        vp = np.zeros(1, np.int32); vl = np.zeros(1, np.int32); vc = np.zeros(1, np.int32)
    out = np.zeros(3 * (len(r) + 1), np.int32)
    n = L.ref_post_process_noisy_regs(len(r), r.ctypes.data_as(i32p), nv, vp.ctypes.data_as(i32p), vl.ctypes.data_as(i32p), vc.ctypes.data_as(i32p), int(flank), out.ctypes.data_as(i32p), len(r) + 1)
    return out[:3 * n].reshape(-1, 3).astype(np.int64)


def ref_sdust(seq, T=5, W=20):
    """the reference's own sdust() (src/sdust.c) -> (n, 2) array of (start, finish)"""
    L = ref_cgranges()
    L.ref_sdust.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i32p, C.c_int]
    a = _c8(seq)
    out = np.zeros(2 * (len(a) // 2 + 8), np.int32)
    n = L.ref_sdust(_p(a), len(a), int(T), int(W), out.ctypes.data_as(i32p), len(out) // 2)
    return out[:2 * n].reshape(-1, 2).astype(np.int64)


def ref_edlib_nw(query, target):
    r = ref_edlib()
    q, t = _c8(query), _c8(target)
    buf = np.zeros(len(q) + len(t) + 8, np.uint8)
    n = C.c_int()
    d = r.ref_edlib_nw_path(_p(q), len(q), _p(t), len(t), _p(buf), len(buf), C.byref(n))
    return d, buf[:n.value].copy()


def ref_edlib_hw(query, target):
    """the reference's own edlib in HW (infix) mode with the path: -> (distance, start0, end0, ops)"""
    r = ref_edlib()
    q, t = _c8(query), _c8(target)
    buf = np.zeros(len(q) + len(t) + 8, np.uint8)
    n, s0, e0, nl = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    d = r.ref_edlib_hw_path(_p(q), len(q), _p(t), len(t), _p(buf), len(buf), C.byref(n), C.byref(s0), C.byref(e0), C.byref(nl))
    return d, s0.value, e0.value, buf[:n.value].copy()


def edlib_hw(query, target):
    """oracle/edlib_nw.c lcdo_edlib_hw: -> (distance, start0, end0, ops)"""
    q, t = _c8(query), _c8(target)
    ap, n, s0, e0 = u8p(), C.c_int(), C.c_int(), C.c_int()
    L = lib()
    L.lcdo_edlib_hw.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p, C.POINTER(u8p), i32p]
    d = L.lcdo_edlib_hw(_p(q), len(q), _p(t), len(t), C.byref(s0), C.byref(e0), C.byref(ap), C.byref(n))
    ops = np.ctypeslib.as_array(ap, shape=(max(n.value, 1),))[:n.value].copy() if n.value else np.zeros(0, np.uint8)
    if ap:
        _libc.free(ap)
    return d, s0.value, e0.value, ops


def edlib_infix_aln(target, query):
    t, q = _c8(target), _c8(query)
    a, b = C.c_int(), C.c_int()
    L = lib()
    L.lcdo_edlib_infix_aln.argtypes = [u8p, C.c_int, u8p, C.c_int, i32p, i32p]
    d = L.lcdo_edlib_infix_aln(_p(t), len(t), _p(q), len(q), C.byref(a), C.byref(b))
    return d, a.value, b.value


def ops_to_xgaps(ops):
    """edlibAlignmentToXGAPS, src/align.c:189-208"""
    x = 0
    for i, o in enumerate(ops):
        if o == 3:
            x += 1
        elif o in (1, 2) and (i == 0 or ops[i - 1] != o):
            x += 1
    return x


def edlib_xgaps(target, query):
    t, q = _c8(target), _c8(query)
    return lib().lcdo_edlib_xgaps(_p(t), len(t), _p(q), len(q))


def edlib_end2end_aln(target, query):
    t, q = _c8(target), _c8(query)
    a, b = C.c_int(), C.c_int()
    d = lib().lcdo_edlib_end2end_aln(_p(t), len(t), _p(q), len(q), C.byref(a), C.byref(b))
    return d, a.value, b.value


def wfa_end2end_aln(pattern, text, gap_aln=1, b=6, q=6, e=2, q2=24, e2=1):
    """-> dict(score, cigar, pattern_alg, text_alg)"""
    p, t = _c8(pattern), _c8(text)
    cb, cl, pa, ta, al, sc = u32p(), C.c_int(), u8p(), u8p(), C.c_int(), C.c_int()
    lib().lcdo_wfa_end2end_aln(_p(p), len(p), _p(t), len(t), gap_aln, b, q, e, q2, e2, C.byref(cb), C.byref(cl), C.byref(pa), C.byref(ta), C.byref(al), C.byref(sc))
    out = dict(score=sc.value, cigar=np.ctypeslib.as_array(cb, shape=(max(cl.value, 1),))[:cl.value].copy(),
               pattern_alg=np.ctypeslib.as_array(pa, shape=(max(al.value, 1),))[:al.value].copy(),
               text_alg=np.ctypeslib.as_array(ta, shape=(max(al.value, 1),))[:al.value].copy())
    _libc.free(cb)
    _libc.free(pa)
    return out


def gotoh2p_score(pattern, text, b=6, q=6, e=2, q2=24, e2=1):
    p, t = _c8(pattern), _c8(text)
    return lib().lcdo_gotoh2p_score(_p(p), len(p), _p(t), len(t), b, q, e, q2, e2)


def cigar_score2p(cigar, pattern, text, b=6, q=6, e=2, q2=24, e2=1):
    p, t = _c8(pattern), _c8(text)
    c = np.ascontiguousarray(cigar, np.uint32)
    return lib().lcdo_cigar_score2p(c.ctypes.data_as(u32p), len(c), _p(p), len(p), _p(t), len(t), b, q, e, q2, e2)


def _poa_unpack(res, n, nc):
    out = dict(n_cons=nc, msa_len=res.msa_len,
               cons=[np.ctypeslib.as_array(res.cons_seq[c], shape=(max(res.cons_len[c], 1),))[:res.cons_len[c]].copy() for c in range(nc)],
               msa=[np.ctypeslib.as_array(res.msa[i], shape=(max(res.msa_len, 1),))[:res.msa_len].copy() for i in range(n + nc)],
               clu=[np.array(res.clu_read_ids[c][:res.clu_n_seq[c]], np.int32) for c in range(nc)])
    lib().lcdo_poa_result_free(C.byref(res))
    return out


def poa_partial_aln_msa_cons(reads, covers, opt=None, sampling=0):
    """K1, src/align.c:762"""
    opt = opt or default_opt()
    reads = [_c8(r) for r in reads]
    n = len(reads)
    arr = (u8p * n)(*[_p(r) for r in reads])
    lens = (C.c_int * n)(*[len(r) for r in reads])
    fc = (C.c_int * n)(*[int(c) for c in covers])
    res = PoaRes()
    nc = lib().lcdo_poa_partial_aln_msa_cons(C.byref(opt), sampling, n, arr, lens, fc, C.byref(res))
    return _poa_unpack(res, n, nc)


def poa_partial_aln_msa_cons_anchored(reads, anchors, skip=None, opt=None):
    """K1 with given anchors [(ref_beg, ref_end, read_beg, read_end)] (a chain dumped by LCD_DUMP_CHAIN: tools/replay_chain.py --oracle)"""
    opt = opt or default_opt()
    reads = [_c8(r) for r in reads]
    n = len(reads)
    arr = (u8p * n)(*[_p(r) for r in reads])
    lens = (C.c_int * n)(*[len(r) for r in reads])
    an = (C.c_int * (4 * n))(*[int(x) for a in anchors for x in a])
    sk = (C.c_int * n)(*[int(x) for x in (skip or [0] * n)])
    res = PoaRes()
    L = lib()
    L.lcdo_poa_partial_aln_msa_cons_anchored.restype = C.c_int
    nc = L.lcdo_poa_partial_aln_msa_cons_anchored(C.byref(opt), n, arr, lens, an, sk, C.byref(res))
    return _poa_unpack(res, n, nc)


def poa_cert_stats():
    """counters of oracle/poa.c's certified-band checker (run poa_aln_msa_cons with LCDO_CERT_STATS=1 in the environment first)"""
    L = lib()
    out = (C.c_longlong * 20)()
    L.lcdo_poa_cert_stats.argtypes = [C.POINTER(C.c_longlong)]; L.lcdo_poa_cert_stats.restype = None
    L.lcdo_poa_cert_stats(out)
    o = list(out)
    return dict(rows=o[0], full_cells=o[1], hull_cells=o[2], reads=o[3], prefix_violations=o[4], path_violations=o[5], widest=o[6], wide_rows=o[7],
                policy_cells=o[11], policy_retries=o[12], policy_wide_reads=o[13], true_wide_reads=o[14])


def poa_aln_msa_cons(reads, max_n_cons=2, opt=None):
    """K2, src/align.c:872"""
    opt = opt or default_opt()
    reads = [_c8(r) for r in reads]
    n = len(reads)
    arr = (u8p * n)(*[_p(r) for r in reads])
    lens = (C.c_int * n)(*[len(r) for r in reads])
    res = PoaRes()
    nc = lib().lcdo_poa_aln_msa_cons(C.byref(opt), n, arr, lens, max_n_cons, C.byref(res))
    return _poa_unpack(res, n, nc)


def collect_partial_aln_beg_end(target, tcover, query, qcover, opt=None, sampling=0):
    opt = opt or default_opt()
    t, q = _c8(target), _c8(query)
    a = [C.c_int() for _ in range(4)]
    r = lib().lcdo_collect_partial_aln_beg_end(C.byref(opt), sampling, _p(t), len(t), tcover, _p(q), len(q), qcover, *[C.byref(x) for x in a])
    return r, tuple(x.value for x in a)


def collect_noisy_reg_aln_strs(reg, opt=None):
    """region driver (src/align.c:1760 after read slicing) -> same dict layout as longcalld_amd.align.RegionBatch.result, plus sorted_ids"""
    opt = opt or default_opt()
    n = len(reg["seqs"])
    seqs = [_c8(s) for s in reg["seqs"]]
    quals = reg.get("quals")
    quals = [_c8(s) for s in quals] if quals is not None else [np.zeros(max(len(s), 1), np.uint8) for s in seqs]
    sp = (u8p * n)(*[_p(s) for s in seqs])
    qp = (u8p * n)(*[_p(s) for s in quals])
    ids = np.array(reg["read_ids"], np.int32).copy()
    lens = np.array([len(s) for s in seqs], np.int32)
    cov = np.array(reg["covers"], np.int32).copy()
    haps = np.array(reg["haps"], np.int32).copy()
    pss = np.array(reg["phase_sets"], np.int64).copy()
    rr = RegionReads(n, ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), sp, qp, cov.ctypes.data_as(i32p), haps.ctypes.data_as(i32p),
                     pss.ctypes.data_as(C.POINTER(C.c_int64)))
    ref = _c8(reg["ref"])
    m = 1 + 2 * n
    clu_n = (C.c_int * 2)(0, 0)
    clu_ids = (i32p * 2)()
    a0, a1 = (AlnStr * m)(), (AlnStr * m)()
    arr = (C.POINTER(AlnStr) * 2)(C.cast(a0, C.POINTER(AlnStr)), C.cast(a1, C.POINTER(AlnStr)))
    nc = lib().lcdo_collect_noisy_reg_aln_strs(C.byref(opt), int(reg["reg_len"]), C.byref(rr), _p(ref), len(ref), clu_n, clu_ids, arr)
    res = dict(n_cons=nc, clu_n_seqs=[int(clu_n[0]), int(clu_n[1])], clu_read_ids=[], aln_strs=[[], []], sorted_ids=ids.copy())
    for c in range(2):
        if clu_ids[c]:
            res["clu_read_ids"].append(np.ctypeslib.as_array(clu_ids[c], shape=(max(clu_n[c], 1),))[:clu_n[c]].copy())
            _libc.free(clu_ids[c])
        else:
            res["clu_read_ids"].append(None)
        for j in range(m):
            s = (a0, a1)[c][j]
            if not s.target_aln:
                res["aln_strs"][c].append(None)
                continue
            L = s.aln_len
            t = np.ctypeslib.as_array(s.target_aln, shape=(max(L, 1),))[:L].copy()
            q = np.ctypeslib.as_array(s.query_aln, shape=(max(L, 1),))[:L].copy()
            res["aln_strs"][c].append(dict(target=t, query=q, aln_len=L, target_beg=s.target_beg, target_end=s.target_end,
                                           query_beg=s.query_beg, query_end=s.query_end))
            _libc.free(s.target_aln)
    return res


class NoisyVar(C.Structure):
    _fields_ = [("pos", C.c_int64)] + [(n, C.c_int) for n in ("var_type", "ref_len", "alt_len", "cate", "from_cons", "is_homopolymer_indel",
                                                               "ref_base", "alt_ref_base", "total_cov")] + [("alle_covs", C.c_int * 2), ("alt_off", C.c_int)]


VAR_KEYS = ("pos", "var_type", "ref_len", "alt_len", "cate", "from_cons", "is_homopolymer_indel", "ref_base", "alt_ref_base", "total_cov")


def make_vars_from_msa_cons_aln(res, noisy_reg_beg, chunk_ref, chunk_ref_beg, min_sv_len=30):
    """SURVEY 8(f) f1 oracle (oracle/cand_vars.c) on a region result dict (collect_noisy_reg_aln_strs above or RegionBatch.result):
    -> dict(n_vars, <VAR_KEYS arrays>, alle_covs (n,2), alt_seqs [arrays], prof_start, prof_end, prof_alleles (rows x n_vars))"""
    nc = res["n_cons"]
    keep, arrs = [], []
    for c in range(2):
        m = len(res["aln_strs"][c])
        a = (AlnStr * max(m, 1))()
        for j, s in enumerate(res["aln_strs"][c] if c < nc else []):
            if s is None:
                continue
            t = _c8(s["target"]); q = _c8(s["query"])
            keep += [t, q]
            a[j].target_aln, a[j].query_aln, a[j].aln_len = _p(t), _p(q), int(s["aln_len"])
            a[j].target_beg, a[j].target_end, a[j].query_beg, a[j].query_end = (int(s[k]) for k in ("target_beg", "target_end", "query_beg", "query_end"))
        arrs.append(a)
    pa = (C.POINTER(AlnStr) * 2)(C.cast(arrs[0], C.POINTER(AlnStr)), C.cast(arrs[1], C.POINTER(AlnStr)))
    clu_n = (C.c_int * 2)(*[int(x) for x in res["clu_n_seqs"]])
    cref = _c8(chunk_ref)
    vars_p, pool_p = C.POINTER(NoisyVar)(), u8p()
    ps, pe, pal = i32p(), i32p(), i32p()
    L = lib()
    L.lcdo_make_vars_from_msa_cons_aln.argtypes = [C.c_int, u8p, C.c_int64, C.c_int64, C.c_int64, C.c_int, i32p, C.POINTER(C.POINTER(AlnStr)),
                                                   C.POINTER(C.POINTER(NoisyVar)), C.POINTER(u8p), C.POINTER(i32p), C.POINTER(i32p), C.POINTER(i32p)]
    n = L.lcdo_make_vars_from_msa_cons_aln(int(min_sv_len), _p(cref), int(chunk_ref_beg), len(cref), int(noisy_reg_beg), nc, clu_n, pa,
                                           C.byref(vars_p), C.byref(pool_p), C.byref(ps), C.byref(pe), C.byref(pal))
    rows = sum(int(res["clu_n_seqs"][c]) for c in range(nc))
    out = dict(n_vars=n, n_rows=rows)
    for k in VAR_KEYS:
        out[k] = np.array([getattr(vars_p[i], k) for i in range(n)], np.int64)
    out["alle_covs"] = np.array([[vars_p[i].alle_covs[0], vars_p[i].alle_covs[1]] for i in range(n)], np.int32).reshape(n, 2)
    out["alt_seqs"] = [np.array([pool_p[vars_p[i].alt_off + k] for k in range(vars_p[i].alt_len)], np.uint8) for i in range(n)]
    if n > 0 and rows > 0:
        out["prof_start"] = np.ctypeslib.as_array(ps, shape=(rows,)).copy()
        out["prof_end"] = np.ctypeslib.as_array(pe, shape=(rows,)).copy()
        out["prof_alleles"] = np.ctypeslib.as_array(pal, shape=(rows * n,)).copy().reshape(rows, n)
    else:
        out["prof_start"] = np.full(rows, -1, np.int32); out["prof_end"] = np.full(rows, -2, np.int32)
        out["prof_alleles"] = np.full((rows, n), -1, np.int32)
    for p in (vars_p, pool_p, ps, pe, pal):
        if p:
            _libc.free(C.cast(p, C.c_void_p))
    return out


class DigarOpt(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("min_bq", "noisy_reg_max_xgaps", "noisy_reg_slide_win", "end_clip_reg", "end_clip_reg_flank_win")] + [
        ("max_noisy_frac_per_read", C.c_double), ("max_var_ratio_per_read", C.c_double)]


class DigarLQ(C.Structure):
    _fields_ = [("pos", C.c_int64), ("type", C.c_int), ("len", C.c_int), ("qi", C.c_int), ("is_low_qual", C.c_int)]


def digar_opt(is_ont=0):
    """src/call_var_main.h:19-38 defaults"""
    return DigarOpt(10, 5, 25 if is_ont else 100, 30, 100, 0.5, 0.05)


def _collect_digar(fn, mid_types, mid_args, pos0, cigar, qual, reg_beg, reg_end, whole_ref_len, opt, left_pal, right_pal, pre_types=(), pre_args=()):
    opt = opt or digar_opt()
    cg = np.ascontiguousarray(cigar, np.uint32); ql = _c8(qual)
    dp, nz, cz = C.POINTER(DigarLQ)(), C.POINTER(C.c_int64)(), C.POINTER(C.c_int64)()
    nd, nn, nc, ncand = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    beg, end = C.c_int64(), C.c_int64()
    f = getattr(lib(), fn)
    f.argtypes = [C.POINTER(DigarOpt), C.c_int64, C.POINTER(C.c_uint32), C.c_int] + list(pre_types) + [u8p, C.c_int] + list(mid_types) + [C.c_int64, C.c_int64, C.c_int64,
                  C.c_int, C.c_int, C.POINTER(C.POINTER(DigarLQ)), i32p, C.POINTER(C.POINTER(C.c_int64)), i32p,
                  C.POINTER(C.POINTER(C.c_int64)), i32p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), i32p]
    rc = f(C.byref(opt), int(pos0), cg.ctypes.data_as(C.POINTER(C.c_uint32)), len(cg), *pre_args, _p(ql), len(ql), *mid_args, int(reg_beg), int(reg_end),
           int(whole_ref_len), int(left_pal), int(right_pal), C.byref(dp), C.byref(nd), C.byref(nz), C.byref(nn), C.byref(cz), C.byref(nc),
           C.byref(beg), C.byref(end), C.byref(ncand))
    dg = np.array([[dp[i].pos, dp[i].type, dp[i].len, dp[i].qi, dp[i].is_low_qual] for i in range(nd.value)], np.int64).reshape(-1, 5)
    noisy = np.array([nz[i] for i in range(3 * nn.value)], np.int64).reshape(-1, 3)
    cn = np.array([cz[i] for i in range(3 * nc.value)], np.int64).reshape(-1, 3)
    for p in (dp, nz, cz):
        if p:
            _libc.free(C.cast(p, C.c_void_p))
    return dict(rc=rc, digars=dg, noisy=noisy, chunk_noisy=cn, beg=beg.value, end=end.value, n_cand=ncand.value)


def collect_digar_from_eqx_cigar(pos0, cigar, qual, reg_beg, reg_end, whole_ref_len, opt=None, left_pal=0, right_pal=0):
    """SURVEY 8(f) f2 oracle (oracle/digar.c): -> dict(rc, digars (n,5), noisy (m,3), chunk_noisy (k,3), beg, end, n_cand)"""
    return _collect_digar("lcdo_collect_digar_from_eqx_cigar", (), (), pos0, cigar, qual, reg_beg, reg_end, whole_ref_len, opt, left_pal, right_pal, (u8p,), (None,))


def collect_digar_from_cs_tag(pos0, cigar, cs, qual, reg_beg, reg_end, whole_ref_len, opt=None, left_pal=0, right_pal=0):
    """oracle/digar_tags.c: collect_digar_from_cs_tag (src/bam_utils.c:844)"""
    return _collect_digar("lcdo_collect_digar_from_cs_tag", (), (), pos0, cigar, qual, reg_beg, reg_end, whole_ref_len, opt, left_pal, right_pal, (C.c_char_p,), (bytes(cs),))


def collect_digar_from_MD_tag(pos0, cigar, md, qual, reg_beg, reg_end, whole_ref_len, opt=None, left_pal=0, right_pal=0):
    """oracle/digar_tags.c: collect_digar_from_MD_tag (src/bam_utils.c:1010)"""
    return _collect_digar("lcdo_collect_digar_from_MD_tag", (), (), pos0, cigar, qual, reg_beg, reg_end, whole_ref_len, opt, left_pal, right_pal, (C.c_char_p,), (bytes(md),))


def collect_digar_from_ref_seq(pos0, cigar, bseq, qual, ref_seq, ref_beg, ref_end, reg_beg, reg_end, whole_ref_len, opt=None, left_pal=0, right_pal=0):
    """oracle/digar_tags.c: collect_digar_from_ref_seq (src/bam_utils.c:1179); bseq = BAM 4-bit packed bases"""
    bs = _c8(bseq)
    return _collect_digar("lcdo_collect_digar_from_ref_seq", (C.c_char_p, C.c_int64, C.c_int64), (bytes(ref_seq), int(ref_beg), int(ref_end)), pos0, cigar, qual,
                          reg_beg, reg_end, whole_ref_len, opt, left_pal, right_pal, (u8p,), (_p(bs),))


class Digar1(C.Structure):
    _fields_ = [("pos", C.c_int64), ("type", C.c_int), ("len", C.c_int), ("qi", C.c_int)]


def read_region_slice(digars, qlen, reg_beg, reg_end, flank=10):
    """one read of collect_noisy_read_info (src/align.c:1392-1458): digars = (n, 4) rows (pos, type, len, qi) -> (read_beg, read_end, cover)"""
    d = np.asarray(digars, np.int64)
    da = (Digar1 * max(len(d), 1))()
    for k in range(len(d)):
        da[k].pos, da[k].type, da[k].len, da[k].qi = int(d[k, 0]), int(d[k, 1]), int(d[k, 2]), int(d[k, 3])
    rb, re, cv = C.c_int(), C.c_int(), C.c_int()
    L = lib()
    L.lcdo_read_region_slice.argtypes = [C.POINTER(Digar1), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int, i32p, i32p, i32p]
    L.lcdo_read_region_slice.restype = None
    L.lcdo_read_region_slice(da, len(d), int(qlen), int(reg_beg), int(reg_end), int(flank), C.byref(rb), C.byref(re), C.byref(cv))
    return rb.value, re.value, cv.value


class HapProblem(C.Structure):
    _i64p = C.POINTER(C.c_int64)
    _fields_ = [("n_reads", C.c_int), ("n_vars", C.c_int), ("is_ont", C.c_int), ("var_pos", _i64p), ("var_type", i32p), ("var_cate", i32p),
                ("is_homopolymer_indel", i32p), ("total_cov", i32p), ("alle_off", i32p), ("alle_covs", i32p), ("start_var_idx", i32p),
                ("end_var_idx", i32p), ("allele_off", i32p), ("alleles", i32p), ("ordered_read_ids", i32p), ("is_skipped", u8p),
                ("n_cr", C.c_int), ("cr_read", i32p), ("haps", i32p), ("phase_sets", _i64p), ("n_clean_agree_snps", i32p),
                ("n_clean_conflict_snps", i32p), ("var_phase_set", _i64p), ("hap_to_cons_alle", i32p), ("hap_to_alle_profile", i32p)]


def assign_hap_germline(prob, target_var_cate, state=None):
    """K5 oracle (oracle/assign_hap.c) on the same flattened problem dict the HIP mirror takes"""
    R, V, TA = prob["n_reads"], prob["n_vars"], int(prob["alle_off"][-1])
    state = state or dict(haps=np.zeros(R, np.int32), phase_sets=np.full(R, -1, np.int64), n_clean_agree_snps=np.zeros(R, np.int32),
                          n_clean_conflict_snps=np.zeros(R, np.int32), var_phase_set=np.full(V, -1, np.int64),
                          hap_to_cons_alle=np.full(V * 3, -1, np.int32), hap_to_alle_profile=np.zeros(3 * TA, np.int32))
    keep = []
    s = HapProblem()
    s.n_reads, s.n_vars, s.is_ont, s.n_cr = R, V, prob["is_ont"], len(prob["cr_read"])
    i64p = C.POINTER(C.c_int64)
    for name, ty in (("var_pos", i64p), ("var_type", i32p), ("var_cate", i32p), ("is_homopolymer_indel", i32p), ("total_cov", i32p),
                     ("alle_off", i32p), ("alle_covs", i32p), ("start_var_idx", i32p), ("end_var_idx", i32p), ("allele_off", i32p),
                     ("alleles", i32p), ("ordered_read_ids", i32p), ("cr_read", i32p)):
        a = np.ascontiguousarray(prob[name], np.int64 if ty is i64p else np.int32)
        if a.size == 0:
            a = np.zeros(1, a.dtype)
        keep.append(a)
        setattr(s, name, a.ctypes.data_as(ty))
    sk = np.ascontiguousarray(prob["is_skipped"], np.uint8)
    s.is_skipped = sk.ctypes.data_as(u8p)
    for name, ty in (("haps", i32p), ("phase_sets", i64p), ("n_clean_agree_snps", i32p), ("n_clean_conflict_snps", i32p), ("var_phase_set", i64p),
                     ("hap_to_cons_alle", i32p), ("hap_to_alle_profile", i32p)):
        setattr(s, name, state[name].ctypes.data_as(ty))
    L = lib()
    L.lcdo_assign_hap_germline.argtypes = [C.POINTER(HapProblem), C.c_int]
    L.lcdo_assign_hap_germline(C.byref(s), int(target_var_cate))
    return state


def cr_sorted_order(st, en):
    """order in which cgranges holds intervals after cr_index() (src/cgranges.c:350-353 + radix sort :13-86)"""
    st = np.ascontiguousarray(st, np.int32); en = np.ascontiguousarray(en, np.int32)
    out = np.zeros(len(st), np.int32)
    L = lib()
    L.lcdo_cr_sorted_order.argtypes = [C.c_int, i32p, i32p, i32p]
    L.lcdo_cr_sorted_order(len(st), st.ctypes.data_as(i32p), en.ctypes.data_as(i32p), out.ctypes.data_as(i32p))
    return out
