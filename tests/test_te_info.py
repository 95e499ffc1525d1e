"""SURVEY a14 on CPU: retrotransposon annotation of SV-size gaps -- the library's host code (lcd_collect_te_info, lcd_collect_te_info_from_cons,
lcd_check_te_seq, lcd_te_lib_create: longcalld_amd/csrc/lcd_emit.cpp) against oracle/te_info.c, the restatement of src/align.c:32-163 and src/kmer.c, and
both against known answers derived by hand from the reference's text (the target-site duplication with its one mismatch, poly-A before poly-T, the 20-base
search window, the reference's own "simple k-mer" rule, the strand tie).  No GPU: this is host code in the reference and here."""
import ctypes as C

import numpy as np
import pytest

from longcalld_amd import _lib

u8p, i64p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int64), C.POINTER(C.c_int)


class TeOpt(C.Structure):
    _fields_ = [("min_tsd_len", C.c_int), ("max_tsd_len", C.c_int), ("min_polya_len", C.c_int), ("min_polya_ratio", C.c_float)]


@pytest.fixture(scope="module")
def libs(oracle):
    prod, orc = C.CDLL(_lib.LIB_PATH), oracle.lib()
    prod.lcd_te_lib_create.restype = C.c_void_p; prod.lcd_te_lib_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), i32p, C.c_int]
    orc.lcdo_te_lib_create.restype = C.c_void_p; orc.lcdo_te_lib_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), i32p, C.c_int]
    prod.lcd_te_lib_destroy.argtypes = [C.c_void_p]; orc.lcdo_te_lib_destroy.argtypes = [C.c_void_p]
    prod.lcd_check_te_seq.argtypes = [C.c_void_p, u8p, C.c_int, i32p]; orc.lcdo_check_te_seq.argtypes = [C.c_void_p, u8p, C.c_int, i32p]
    outs = [u8p, i64p, i64p, i32p, i32p, i32p]
    prod.lcd_collect_te_info.argtypes = [C.POINTER(TeOpt), C.c_void_p, C.c_int, u8p, u8p, C.c_int, C.c_int64] + outs
    orc.lcdo_collect_te_info.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, u8p, u8p, C.c_int, C.c_int64] + outs
    prod.lcd_collect_te_info_from_cons.argtypes = [C.POINTER(TeOpt), C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, u8p] + outs
    orc.lcdo_collect_te_info_from_cons.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, u8p] + outs
    prod.lcd_te_opt_default.argtypes = [C.POINTER(TeOpt)]
    return prod, orc


def _mk_lib(fn, seqs, k):
    arr = (C.c_char_p * max(len(seqs), 1))(*[bytes(s) for s in seqs]); lens = (C.c_int * max(len(seqs), 1))(*[len(s) for s in seqs])
    return fn(len(seqs), arr, lens, k)


def _p(a):
    return a.ctypes.data_as(u8p)


def _outs():
    return np.zeros(128, np.uint8), C.c_int64(), C.c_int64(), C.c_int(), C.c_int(), C.c_int()


def _res(r, o):
    tsd, p1, p2, pa, ti, tr = o
    return (r, bytes(tsd[:max(r, 0)]), p1.value, p2.value, pa.value, ti.value, tr.value)


def te_info(libs, which, opt, lib, var_type, gap, flank, pos):
    prod, orc = libs
    gap = np.ascontiguousarray(gap, np.uint8); flank = np.ascontiguousarray(flank, np.uint8)
    o = _outs()
    tail = (_p(o[0]), C.byref(o[1]), C.byref(o[2]), C.byref(o[3]), C.byref(o[4]), C.byref(o[5]))
    if which == 0:
        r = prod.lcd_collect_te_info(C.byref(opt), lib, var_type, _p(gap), _p(flank), len(gap), pos, *tail)
    else:
        r = orc.lcdo_collect_te_info(opt.min_tsd_len, opt.max_tsd_len, opt.min_polya_len, opt.min_polya_ratio, lib, var_type, _p(gap), _p(flank), len(gap), pos, *tail)
    return _res(r, o)


def _opt(libs):
    o = TeOpt(); libs[0].lcd_te_opt_default(C.byref(o))
    assert (o.min_tsd_len, o.max_tsd_len, o.min_polya_len) == (2, 100, 10) and abs(o.min_polya_ratio - 0.8) < 1e-7      # src/call_var_main.h:55-58
    return o


def codes(s):
    return np.array(["ACGTN".index(c) for c in s], np.uint8)


def test_known_answers(libs):
    opt = _opt(libs)
    # an insertion: TSD "ACGTAC" (then two mismatches), body, 12 A's at the end -> tsd 6, poly-A 12 (the last qualifying suffix length is what is kept:
    # the walk goes on past the A's while fewer than 21 bases were seen, and every A further in that still gives >= 80 % updates it)
    gap = codes("ACGTAC" + "GGCCGGCCGGTC" + "A" * 12)
    flank = codes("ACGTAC" + "TTAATTAATTAATTAATTAATTAA")
    for w in (0, 1):
        assert te_info(libs, w, opt, None, 1, gap, flank, 1000) == (6, bytes(codes("ACGTAC")), 1000, -1, 12, -1, 0)
        # the same gap as a deletion: second TSD position = pos + gap_len
        assert te_info(libs, w, opt, None, 2, gap, flank, 1000) == (6, bytes(codes("ACGTAC")), 1000, 1000 + len(gap), 12, -1, 0)
    # one mismatch inside the duplication is allowed, the second ends it: ACGTxACG|y -> tsd 8 ; bases copied are the REFERENCE's (flank), not the gap's
    gap = codes("ACGTTACG" + "CCCC" + "A" * 10); flank = codes("ACGTGACG" + "GGGG" + "C" * 10)
    for w in (0, 1):
        assert te_info(libs, w, opt, None, 1, gap, flank, 5)[:2] == (8, bytes(codes("ACGTGACG")))
    # no poly-A at the end but poly-T right behind the duplication: negative length
    gap = codes("GATTACA" + "T" * 11 + "GCGCGCGCGCGCGCGCGCGCGCGCGC"); flank = codes("GATTACA" + "C" * 37)
    for w in (0, 1):
        r = te_info(libs, w, opt, None, 1, gap, flank, 7)
        assert r[0] == 7 and r[4] == -11
    # a duplication of one base is below min_tsd_len; so is none at all; poly-A alone is not enough
    for g, f in (("ACCCCCCCCCC" + "A" * 12, "AGGGGGGGGGG" + "G" * 12), ("CCCC" + "A" * 12, "GGGG" + "G" * 12)):
        for w in (0, 1):
            assert te_info(libs, w, opt, None, 1, codes(g), codes(f), 1)[0] == 0
    # TSD but neither tail: 0, and the outputs keep their "nothing" values
    for w in (0, 1):
        assert te_info(libs, w, opt, None, 1, codes("ACGTACGGCCGGCCGGCCGGCC"), codes("ACGTACTTTTTTTTTTTTTTTT"), 9) == (0, b"", -1, -1, -1, -1, 0)
    # longer than max_tsd_len (the whole gap repeats): not a TSD
    g = codes("ACGT" * 30 + "A" * 12)
    for w in (0, 1):
        assert te_info(libs, w, opt, None, 1, g, g, 9)[0] == 0


def test_simple_kmer_rule_and_strand_tie(libs):
    """the reference's not_simple_kmer keeps every k-mer with a non-A base above its LAST base: AAAAC is 'simple' (dropped), AAACA is kept, CCCCC is kept;
    a tie between the strands -- 0 : 0 included -- reports the reverse one"""
    prod, orc = libs
    k = 5
    te = [b"AAACATTGCAGGCTAAAAC", b"GGGGGCCCCCAAAAAAAAAAC"]
    for mk, chk, kill in ((prod.lcd_te_lib_create, prod.lcd_check_te_seq, prod.lcd_te_lib_destroy), (orc.lcdo_te_lib_create, orc.lcdo_check_te_seq, orc.lcdo_te_lib_destroy)):
        L = _mk_lib(mk, te, k)
        def q(s):
            a = codes(s); r = C.c_int(7)
            return chk(L, _p(a), len(a), C.byref(r)), r.value
        assert q("AAAAC" * 4) == (-1, 7)                            # only simple k-mers: no query k-mer at all, is_rev untouched
        assert q("AAACATTGCAGGCTA") == (0, 0)                        # three non-overlapping k-mers of TE 0, forward
        assert q("TAGCCTGCAATGTTT") == (0, 1)                        # its reverse complement
        assert q("GGGGGCCCCC") == (-1, 1)                            # two hits < 3 on the forward strand -- and GGGGG CCCCC reversed is itself: 2 : 2, tie -> reverse, too few
        assert q("ACGTGACGTGACGTG") == (-1, 1)                       # nothing anywhere: 0 : 0 -> reverse strand flag, -1
        kill(L)


def test_fuzz_product_equals_oracle(libs):
    prod, orc = libs
    rng = np.random.default_rng(77)
    tes = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(200, 900))).astype(np.uint8)) for _ in range(3)]
    tes[1] = tes[1][:100] + b"N" * 3 + tes[1][100:] + b"acgtacgtnnacgt"
    Lp, Lo = _mk_lib(prod.lcd_te_lib_create, tes, 15), _mk_lib(orc.lcdo_te_lib_create, tes, 15)
    opt = _opt(libs)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    n_te = n_tsd = n_t = 0
    for it in range(3000):
        L = int(rng.integers(1, 400))
        gap = rng.integers(0, 4, L).astype(np.uint8)
        if it % 3 == 0:                                   # a piece of a TE sequence, either strand, a few errors
            t = tes[int(rng.integers(0, 3))].upper().replace(b"N", b"A")
            a = int(rng.integers(0, len(t) - 60)); piece = t[a:a + int(rng.integers(50, 300))]
            if rng.random() < 0.5:
                piece = piece.translate(comp)[::-1]
            gap = np.array([b"ACGT".index(c) for c in piece], np.uint8); L = len(gap)
            for _ in range(int(rng.integers(0, 4))):
                gap[int(rng.integers(0, L))] = int(rng.integers(0, 5))
        flank = rng.integers(0, 4, L).astype(np.uint8)
        t = int(rng.integers(0, min(L, 30) + 1)); flank[:t] = gap[:t]
        if t > 3 and rng.random() < 0.5:
            flank[int(rng.integers(0, t))] ^= 1
        x = rng.random()
        if x < 0.4:
            n = int(rng.integers(5, 30)); gap[max(L - n, 0):] = 0
            if rng.random() < 0.5 and L > 4:
                gap[L - int(rng.integers(1, min(L, 25) + 1))] = 2
        elif x < 0.7:
            n = int(rng.integers(5, 30)); gap[t:t + n] = 3
        vt = 1 + it % 2
        a = te_info(libs, 0, opt, Lp, vt, gap, flank, 1000 + it)
        b = te_info(libs, 1, opt, Lo, vt, gap, flank, 1000 + it)
        assert a == b, it
        ra, rb = C.c_int(-5), C.c_int(-5)
        assert prod.lcd_check_te_seq(Lp, _p(gap), L, C.byref(ra)) == orc.lcdo_check_te_seq(Lo, _p(gap), L, C.byref(rb)) and ra.value == rb.value
        n_tsd += a[0] > 0; n_te += a[5] >= 0; n_t += a[4] < -1
    assert n_tsd > 300 and n_te > 50 and n_t > 50
    prod.lcd_te_lib_destroy(Lp); orc.lcdo_te_lib_destroy(Lo)


def test_from_cons_and_reference_window(libs):
    """collect_te_info_from_cons: the gap from the consensus row (INS) or the reference (DEL), the bases behind it from the reference; positions outside the loaded
    window read as N (get_bseq1)"""
    prod, orc = libs
    opt = _opt(libs)
    rng = np.random.default_rng(5)
    ref = bytes(rng.choice(list(b"ACGTacgt"), 600).astype(np.uint8)); ref_beg = 5000
    for it in range(400):
        gl = int(rng.integers(30, 120)); start = ref_beg + int(rng.integers(-20, 560)); vt = 1 + it % 2
        row = rng.integers(0, 6, 300).astype(np.uint8); ms = int(rng.integers(0, 150))
        if vt == 1:   # make it look like a TE insertion every other time
            seg = np.array([b"ACGT".index(bytes([c]).upper()) if ref_beg <= start + i <= ref_beg + 599 else 4 for i, c in enumerate(ref[max(start - ref_beg, 0):max(start - ref_beg, 0) + 8])], np.uint8)
            row[ms:ms + len(seg)] = seg; row[ms + gl - 12:ms + gl] = 0
        res = []
        for w in (0, 1):
            o = _outs(); tail = (_p(o[0]), C.byref(o[1]), C.byref(o[2]), C.byref(o[3]), C.byref(o[4]), C.byref(o[5]))
            if w == 0:
                r = prod.lcd_collect_te_info_from_cons(C.byref(opt), None, ref, ref_beg, ref_beg + 599, start, ms, vt, gl, _p(row), *tail)
            else:
                r = orc.lcdo_collect_te_info_from_cons(2, 100, 10, 0.8, None, ref, ref_beg, ref_beg + 599, start, ms, vt, gl, _p(row), *tail)
            res.append(_res(r, o))
        assert res[0] == res[1], it
