"""SURVEY 8(f) f4 -- cross-chunk stitching (flip_variant_hap, src/collect_var.c:1618), genotype records (make_variants :1465, GQ / QUAL :1435-1459) and the
VCF body text (write_var_to_vcf, src/vcf_utils.c:97): the product's host code (liblcd_hotpath.so, lcd_emit.cpp) against the oracle's statement-by-statement
restatement (oracle/emit.c).  Host code on both sides: these run without a GPU."""
import ctypes as C

import numpy as np
import pytest

import emit_common as ec


@pytest.fixture(scope="module")
def libs(oracle):
    from longcalld_amd import _lib
    return C.CDLL(_lib.LIB_PATH), oracle.lib()


def _chunk(seed, nv=220, nr=300, three_alleles=True):
    """a chunk after K5 (the oracle's own, on the CPU) + the cand_var_t allele fields make_variants reads"""
    from longcalld_amd import jobs
    from oracle import pyoracle
    rng = np.random.default_rng(seed)
    p = jobs.make_hap_problem(rng, nv, nr)
    st = pyoracle.assign_hap_germline(p, jobs.GERMLINE_CLEAN)
    st = pyoracle.assign_hap_germline(p, jobs.GERMLINE_ALL, st)
    V = p["n_vars"]
    vtype = p["var_type"]
    ref_len = np.where(vtype == 8, 1, np.where(vtype == 2, rng.integers(1, 60, V), 0)).astype(np.int32)
    alt_len = np.where(vtype == 8, 1, np.where(vtype == 1, rng.integers(1, 60, V), 0)).astype(np.int32)
    alt_off = np.concatenate([[0], np.cumsum(alt_len)]).astype(np.uint64)
    alt_pool = rng.integers(0, 4, int(alt_off[-1]) + 1).astype(np.uint8)
    alt_pool[rng.random(len(alt_pool)) < 0.01] = 4                                   # an N in an alt allele: the line is dropped unless --amb-base
    alt_ref_base = rng.choice([0, 1, 2, 3, 4], V, p=[0.2, 0.2, 0.2, 0.2, 0.2]).astype(np.uint8)
    ref_beg = int(p["var_pos"].min()) - 100
    L = int(p["var_pos"].max()) - ref_beg + 200
    ref = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, L, p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002])].tobytes()
    extra = dict(var_ref_len=ref_len, var_alt_len=alt_len, alt_off=alt_off[:-1].copy(), alt_pool=alt_pool, alt_ref_base=alt_ref_base)
    return p, st, extra, ref, ref_beg


def _hap_struct(p, st, keep):
    from longcalld_amd import _lib, align
    return align._fill_hap_struct(_lib.LcdHapProblem, p, st, keep)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_make_variants_and_vcf_text_match_oracle(libs, seed):
    prod, orc = libs
    p, st, extra, ref, ref_beg = _chunk(seed)
    opt = ec.default_call_opt()
    keep = []
    hs = _hap_struct(p, st, keep)
    reg_beg, reg_end = int(p["var_pos"][5]), int(p["var_pos"][-6])                  # variants outside the chunk's own region are left to its neighbours
    a_recs, a_txt = ec.make_variants(prod, "lcd_", hs, opt, extra, ref, ref_beg, reg_beg, reg_end)
    b_recs, b_txt = ec.make_variants(orc, "lcdo_", hs, opt, extra, ref, ref_beg, reg_beg, reg_end)
    assert len(a_recs) == len(b_recs) > 100
    for x, y in zip(a_recs, b_recs):
        assert x == y
    assert a_txt == b_txt and a_txt.count("\n") > 50
    # spot checks of the format itself (src/vcf_utils.c:165-230)
    line = a_txt.splitlines()[0].split("\t")
    assert line[0] == "chr11" and line[2] == "." and line[6] == "PASS" and line[8].startswith("GT:DP:AD:VAF:GQ")
    assert any(";SVTYPE=" in l for l in a_txt.splitlines()) and any("|" in l.split("\t")[9] for l in a_txt.splitlines())
    n3 = int(((p["alle_off"][1:] - p["alle_off"][:-1]) > 2).sum())
    assert n3 > 0   # three-allele sites exist: the AD[] overrun of var1_t (GT bytes) is exercised and agreed on
    opt.out_amb_base = 1
    a2 = ec.make_variants(prod, "lcd_", hs, opt, extra, ref, ref_beg, reg_beg, reg_end)[1]
    b2 = ec.make_variants(orc, "lcdo_", hs, opt, extra, ref, ref_beg, reg_beg, reg_end)[1]
    assert a2 == b2 and a2.count("\n") >= a_txt.count("\n")


def test_two_alt_alleles_ad_field(libs):
    """a record whose haplotypes carry two DIFFERENT alt alleles (GT 1|2): the formatter prints three AD values and two VAFs.  var1_t has AD[2] only -- its third
    value is what the reference's store at src/collect_var.c:1561 left behind AD[1], the third allele's coverage -- so the product's and the oracle's records
    carry that value as AD[2] (neither reads out of bounds) and the text agrees"""
    prod, orc = libs
    p, st, extra, ref, ref_beg = _chunk(7)
    n_alle = p["alle_off"][1:] - p["alle_off"][:-1]
    three = np.flatnonzero(n_alle > 2)
    assert len(three) > 3
    hc = st["hap_to_cons_alle"].reshape(-1, 3)
    for v in three:
        hc[v, 1], hc[v, 2] = 1, 2
    opt = ec.default_call_opt()
    opt.min_dp = 0; opt.min_alt_dp = 0
    keep = []
    hs = _hap_struct(p, st, keep)
    reg_beg, reg_end = int(p["var_pos"][0]), int(p["var_pos"][-1])
    a_recs, a_txt = ec.make_variants(prod, "lcd_", hs, opt, extra, ref, ref_beg, reg_beg, reg_end)
    b_recs, b_txt = ec.make_variants(orc, "lcdo_", hs, opt, extra, ref, ref_beg, reg_beg, reg_end)
    assert a_recs == b_recs and a_txt == b_txt
    two = [r for r in a_recs if r["n_alt"] == 2]
    assert len(two) > 3
    for r in two:
        cov = p["alle_covs"][p["alle_off"][r["cand_i"]]:p["alle_off"][r["cand_i"] + 1]]
        assert r["AD"][2] == cov[2]                                                  # the third allele's coverage, as in the reference's memory
    lines = [l.split("\t") for l in a_txt.splitlines()]
    both = [l for l in lines if "," in l[4]]                                         # two ALT sequences
    assert both and all(len(l[9].split(":")[2].split(",")) == 3 and len(l[9].split(":")[3].split(",")) == 2 for l in both)


def test_call_opt_defaults(libs):
    prod, _ = libs
    o = ec.CallOpt()
    prod.lcd_call_opt_default(C.byref(o))
    d = ec.default_call_opt()
    for f, _t in ec.CallOpt._fields_:
        assert abs(getattr(o, f) - getattr(d, f)) < 1e-15, f


@pytest.mark.parametrize("seed,update_reads", [(5, 0), (6, 1), (7, 1), (8, 0), (9, 1), (10, 1)])
def test_flip_variant_hap_matches_oracle(libs, seed, update_reads):
    prod, orc = libs
    rng = np.random.default_rng(seed)

    def chunk(nr, nv, ps_vals):
        haps = rng.integers(0, 3, nr).astype(np.int32)
        ps = np.where(haps == 0, -1, rng.choice(ps_vals[1:], nr)).astype(np.int64)     # a read with a haplotype has a phase set (update_read_phase_set)
        return dict(tid=3, ordered_read_ids=rng.permutation(nr).astype(np.int32), is_skipped=(rng.random(nr) < 0.05).astype(np.uint8),
                    haps=haps, phase_sets=ps,
                    var_phase_set=rng.choice(ps_vals, nv).astype(np.int64), hap_to_cons_alle=rng.integers(-1, 2, nv * 3).astype(np.int32),
                    up_ovlp=np.zeros(0, np.int32), down_ovlp=np.zeros(0, np.int32))
    pre, cur = chunk(200, 80, [-1, 1000, 5000]), chunk(220, 90, [-1, 9000, 12000])
    n_ov = 40
    pre["down_ovlp"] = rng.choice(200, n_ov, replace=False).astype(np.int32)
    cur["up_ovlp"] = rng.choice(220, n_ov, replace=False).astype(np.int32)
    if seed % 2:    # make the overlap reads mostly disagree so that the flip branch is taken
        cur["haps"][cur["up_ovlp"]] = 3 - np.maximum(pre["haps"][pre["down_ovlp"]], 1)
        cur["phase_sets"][cur["up_ovlp"]] = np.where(cur["phase_sets"][cur["up_ovlp"]] < 0, 9000, cur["phase_sets"][cur["up_ovlp"]])
    import copy
    pa, ca, pb, cb = copy.deepcopy(pre), copy.deepcopy(cur), copy.deepcopy(pre), copy.deepcopy(cur)
    ra = ec.flip(prod, "lcd_", pa, ca, update_reads)
    rb = ec.flip(orc, "lcdo_", pb, cb, update_reads)
    assert ra == rb and ra[0] == 0 and ra[1][0] in (0, 1)
    voted = ra[1][1] != -7          # (a tied vote leaves the chunk alone, src/collect_var.c:1669)
    assert voted or seed % 2 == 0
    for k in ("haps", "phase_sets", "var_phase_set", "hap_to_cons_alle"):
        assert (ca[k] == cb[k]).all(), k
        assert (pa[k] == pre[k]).all()                                              # the previous chunk is only read
    if not update_reads:
        assert (ca["haps"] == cur["haps"]).all() and (ca["phase_sets"] == cur["phase_sets"]).all()
    if voted:
        assert (ca["var_phase_set"] != cur["var_phase_set"]).any()                  # the phase sets were joined
    else:
        assert all((ca[k] == cur[k]).all() for k in ("haps", "phase_sets", "var_phase_set", "hap_to_cons_alle"))
    # different contigs: untouched; overlap counts that disagree: the reference exits, both sides report it
    c2 = copy.deepcopy(cur); c2["tid"] = 4
    assert ec.flip(prod, "lcd_", copy.deepcopy(pre), c2, 1)[1][1] == -7
    c3 = copy.deepcopy(cur); c3["up_ovlp"] = c3["up_ovlp"][:-1].copy()
    assert ec.flip(prod, "lcd_", copy.deepcopy(pre), c3, 1)[0] == -6 == ec.flip(orc, "lcdo_", copy.deepcopy(pre), copy.deepcopy(c3), 1)[0]


def test_read_tags_policy(libs):
    """write_processed_read_to_bam (src/bam_utils.c:1955-2006): HP:i iff hap != 0, PS:i iff phase set > 0"""
    prod, _ = libs
    haps = np.array([0, 1, 2, 0, 1], np.int32); ps = np.array([-1, 500, 0, 900, -1], np.int64)
    has_hp, has_ps = np.zeros(5, np.uint8), np.zeros(5, np.uint8)
    hp, pso = np.zeros(5, np.int32), np.zeros(5, np.int64)
    prod.lcd_read_tags(5, haps.ctypes.data_as(ec.i32p), ps.ctypes.data_as(ec.i64p), has_hp.ctypes.data_as(ec.u8p), hp.ctypes.data_as(ec.i32p),
                       has_ps.ctypes.data_as(ec.u8p), pso.ctypes.data_as(ec.i64p))
    assert list(has_hp) == [0, 1, 1, 0, 1] and list(has_ps) == [0, 1, 0, 1, 0] and list(hp[[1, 2, 4]]) == [1, 2, 1] and pso[1] == 500


def test_te_annotation_in_records_and_vcf_text(libs):
    """SURVEY a14 in f4: SV-size insertions / deletions built to look like retrotransposon insertions (target-site duplication, TE body, poly-A or poly-T) next to
    ordinary ones: lcd_annotate_te + lcd_format_vcf_te == the oracle's (oracle/emit.c, oracle/te_info.c) record by record and byte by byte, and the text carries
    the keys write_var_to_vcf adds (src/vcf_utils.c:184-195)"""
    prod, orc = libs
    p, st, extra, ref, ref_beg = _chunk(21, nv=400, nr=300)
    rng = np.random.default_rng(99)
    tes = [bytes(rng.choice(list(b"ACGT"), 700).astype(np.uint8)), bytes(rng.choice(list(b"ACGT"), 500).astype(np.uint8))]
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    refa = bytearray(ref)
    V = p["n_vars"]; vtype = p["var_type"]; pos = p["var_pos"]
    alt_len = extra["var_alt_len"].copy(); ref_len = extra["var_ref_len"].copy()
    alts = [extra["alt_pool"][int(extra["alt_off"][i]):int(extra["alt_off"][i]) + int(alt_len[i])].copy() for i in range(V)]
    code = {65: 0, 67: 1, 71: 2, 84: 3, 78: 4}
    n_made = 0
    for i in range(V):
        o = int(pos[i]) - ref_beg
        if vtype[i] == 1 and i % 2 == 0 and o + 40 < len(refa):          # insertion: TSD copied from the reference behind it, TE piece, tail
            t = int(rng.integers(4, 16)); te = tes[i % 4 // 2]; a = int(rng.integers(0, len(te) - 130)); piece = te[a:a + 120]
            if i % 8 >= 4:
                piece = piece.translate(comp)[::-1]
            body = [code[c] for c in piece]
            tsd = [code.get(c, 4) for c in refa[o:o + t]]
            seq = tsd + ([3] * 14 + body if i % 3 == 0 else body + [0] * 14)
            alts[i] = np.array(seq, np.uint8); alt_len[i] = len(seq); n_made += 1
        elif vtype[i] == 2 and i % 2 == 0 and o + 2 * 70 < len(refa):   # deletion of 60 bases: the bases behind it repeat its first ones, it ends in poly-A
            ref_len[i] = 60; t = int(rng.integers(4, 16))
            refa[o + 60 - 13:o + 60] = b"A" * 13
            refa[o + 60:o + 60 + t] = refa[o:o + t]
            n_made += 1
    off = np.concatenate([[0], np.cumsum(alt_len)]).astype(np.uint64)
    extra2 = dict(var_ref_len=ref_len, var_alt_len=alt_len, alt_off=off[:-1].copy(), alt_pool=np.concatenate(alts + [np.zeros(1, np.uint8)]), alt_ref_base=extra["alt_ref_base"])
    ref2 = bytes(refa)
    import test_te_info as tt
    prod.lcd_te_lib_create.restype = C.c_void_p; orc.lcdo_te_lib_create.restype = C.c_void_p
    prod.lcd_te_lib_create.argtypes = orc.lcdo_te_lib_create.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int), C.c_int]
    Lp, Lo = tt._mk_lib(prod.lcd_te_lib_create, tes, 15), tt._mk_lib(orc.lcdo_te_lib_create, tes, 15)
    opt = ec.default_call_opt()
    keep = []
    hs = _hap_struct(p, st, keep)
    reg_beg, reg_end = int(pos[2]), int(pos[-3])
    names = [b"AluYa5", b"L1HS"]
    ta, tb = dict(lib=Lp, names=names), dict(lib=Lo, names=names)
    a_recs, a_txt = ec.make_variants(prod, "lcd_", hs, opt, extra2, ref2, ref_beg, reg_beg, reg_end, te=ta)
    b_recs, b_txt = ec.make_variants(orc, "lcdo_", hs, opt, extra2, ref2, ref_beg, reg_beg, reg_end, te=tb)
    assert len(a_recs) == len(b_recs) and all(x == y for x, y in zip(a_recs, b_recs))
    assert a_txt == b_txt and ta["n_annotated"] == tb["n_annotated"] >= 10 and n_made >= 20
    lines = a_txt.splitlines()
    mei = [l for l in lines if "\tCLEAN;MEI;" in l or "\tMEI;" in l]
    assert len(mei) >= 5 and all(";TSD=" in l and ";TSDLEN=" in l and ";POLYALEN=" in l and ";TSDPOS1=" in l and ";REPNAME=" in l for l in mei)
    assert any(";REPNAME=+AluYa5" in l or ";REPNAME=+L1HS" in l for l in mei) and any(";REPNAME=-" in l for l in mei)
    assert any(";POLYALEN=-" in l for l in lines)                                   # poly-T behind the duplication
    assert any(";TSDPOS2=" in l for l in lines)                                     # deletions carry the second position
    assert any(";SVTYPE=" in l and ";TSD=" not in l for l in lines)                 # ordinary SVs stay as they were
    # without the annotation step the text is that of lcd_format_vcf
    plain = ec.make_variants(prod, "lcd_", hs, opt, extra2, ref2, ref_beg, reg_beg, reg_end)[1]
    assert "TSD=" not in plain and plain.count("\n") == a_txt.count("\n")
    prod.lcd_te_lib_destroy.argtypes = [C.c_void_p]; orc.lcdo_te_lib_destroy.argtypes = [C.c_void_p]
    prod.lcd_te_lib_destroy(Lp); orc.lcdo_te_lib_destroy(Lo)
