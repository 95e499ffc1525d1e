"""oracle/digar_tags.c on CPU: the restatements of collect_digar_from_cs_tag / _MD_tag / _ref_seq (src/bam_utils.c:844, :1010, :1179) against the
restatement of collect_digar_from_eqx_cigar (:701) on alignments given in all four shapes (tests/digar_inputs.py).  The reference's four functions are
written to agree on a well-formed alignment; each restatement was written from its own function, so their agreement here checks the restatements against
one another.  Where the reference's functions DIFFER the test pins the difference: the cs function's clip rule, '=' runs of the MD string carried over an
insertion, bases outside the loaded reference window, malformed tags."""
import numpy as np

import digar_inputs as di


def _same(a, b, what=""):
    assert a["rc"] == b["rc"], what
    for k in ("digars", "noisy", "chunk_noisy"):
        assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), (what, k)
    assert (a["beg"], a["end"], a["n_cand"]) == (b["beg"], b["end"], b["n_cand"]), what


def test_four_sources_agree(oracle):
    orc = oracle
    rng = np.random.default_rng(11)
    n_windows = 0
    for i in range(120):
        ops = di.eqx_ops(rng, noisy=i % 3 == 0, n_skip=False)
        pos0 = int(rng.integers(1000, 800000))
        a = di.build(rng, ops, pos0, ref_pad=(int(rng.integers(0, 50)), int(rng.integers(0, 50))), lower=i % 2 == 0)
        q = di.quals(rng, a["qlen"])
        for is_ont in (0, 1):
            opt = orc.digar_opt(is_ont)
            pal = (int(rng.integers(0, 2)), int(rng.integers(0, 2))) if i % 5 == 0 else (0, 0)
            args = (200000, 700000, 1000000, opt, pal[0], pal[1])
            e = orc.collect_digar_from_eqx_cigar(pos0, a["eqx"], q, *args)
            _same(e, orc.collect_digar_from_cs_tag(pos0, a["eqx"], a["cs"], q, *args), "cs")
            _same(e, orc.collect_digar_from_cs_tag(pos0, a["mcig"], a["cs_long"], q, *args), "cs long")
            _same(e, orc.collect_digar_from_MD_tag(pos0, a["mcig"], a["md"], q, *args), "MD")
            _same(e, orc.collect_digar_from_ref_seq(pos0, a["mcig"], a["bseq"], q, a["ref_seq"], a["ref_beg"], a["ref_end"], *args), "ref")
            _same(e, orc.collect_digar_from_ref_seq(pos0, a["eqx"], a["bseq"], q, a["ref_seq"], a["ref_beg"], a["ref_end"], *args), "ref on =/X")   # (:1201: M, = and X alike)
            n_windows += len(e["noisy"])
    assert n_windows > 100


def test_reference_skips_in_md_and_ref(oracle):
    """'N' operations: MD and reference comparison step over them like the EQX path; (the cs function would need '~', which it does not advance over)"""
    orc = oracle
    rng = np.random.default_rng(12)
    for i in range(30):
        ops = di.eqx_ops(rng, n_skip=True)
        a = di.build(rng, ops, 5000 + i)
        q = di.quals(rng, a["qlen"])
        args = (1, 10 ** 6, 10 ** 7, None, 0, 0)
        e = orc.collect_digar_from_eqx_cigar(a["pos0"], a["eqx"], q, *args)
        _same(e, orc.collect_digar_from_MD_tag(a["pos0"], a["mcig"], a["md"], q, *args))
        _same(e, orc.collect_digar_from_ref_seq(a["pos0"], a["mcig"], a["bseq"], q, a["ref_seq"], a["ref_beg"], a["ref_end"], *args))


def test_cs_clip_rule_differs_near_the_contig_ends(oracle):
    """a long left clip at pos <= 10: the EQX / MD / ref functions do nothing (:772), the cs function still counts a candidate (:884-888); a long right clip
    at pos >= tlen - 10 likewise (:969-972)"""
    orc = oracle
    rng = np.random.default_rng(13)
    ops = [(4, 80), (7, 400), (8, 1), (7, 300), (4, 90)]
    a = di.build(rng, ops, 4)                       # pos = 5
    q = np.full(a["qlen"], 40, np.uint8)
    tlen = 4 + 701 + 3                              # the right clip sits at pos = 706 >= tlen - 10
    args = (1, 10 ** 6, tlen, None, 0, 0)
    e = orc.collect_digar_from_eqx_cigar(4, a["eqx"], q, *args)
    c = orc.collect_digar_from_cs_tag(4, a["eqx"], a["cs"], q, *args)
    assert (e["digars"] == c["digars"]).all() and len(e["noisy"]) == 0 and len(c["noisy"]) == 0
    assert e["n_cand"] == 1 and c["n_cand"] == 3
    m = orc.collect_digar_from_MD_tag(4, a["mcig"], a["md"], q, *args)
    _same(e, m)
    # away from the ends all of them flag both flanks
    a = di.build(rng, ops, 5000)
    args = (1, 10 ** 6, 10 ** 6, None, 0, 0)
    e = orc.collect_digar_from_eqx_cigar(5000, a["eqx"], q, *args)
    _same(e, orc.collect_digar_from_cs_tag(5000, a["eqx"], a["cs"], q, *args))
    assert len(e["noisy"]) == 2 and e["n_cand"] == 3


def test_md_run_carried_over_an_insertion(oracle):
    orc = oracle
    cig = np.array([(10 << 4) | 0, (3 << 4) | 1, (12 << 4) | 0, (2 << 4) | 2, (5 << 4) | 0], np.uint32)   # 10M3I12M2D5M
    q = np.full(30, 40, np.uint8)
    loose = orc.DigarOpt(10, 5, 100, 30, 100, 1.0, 1.0)       # (4 events in 29 bp: the default ratios would skip the read)
    r = orc.collect_digar_from_MD_tag(99, cig, b"15A6^CG0T4", q, 1, 10 ** 6, 10 ** 6, loose)
    assert r["rc"] == 0
    exp = [(100, 7, 10, 0), (110, 1, 3, 10), (110, 7, 5, 13), (115, 8, 1, 18), (116, 7, 6, 19), (122, 2, 2, 25), (124, 8, 1, 25), (125, 7, 4, 26)]
    assert [tuple(int(v) for v in d[:4]) for d in r["digars"]] == exp and r["end"] == 128
    # a tag that runs out before the CIGAR does, letters where the CIGAR has '=' : the reference stops the program
    assert orc.collect_digar_from_MD_tag(99, cig, b"15A6^CG0T", q, 1, 10 ** 6, 10 ** 6)["rc"] == -2
    assert orc.collect_digar_from_MD_tag(99, np.array([(30 << 4) | 7], np.uint32), b"30", q, 1, 10 ** 6, 10 ** 6)["rc"] == -2
    assert orc.collect_digar_from_cs_tag(99, cig, b":10+acg:5!", q, 1, 10 ** 6, 10 ** 6)["rc"] == -2


def test_ref_window_shorter_than_the_read(oracle):
    """bases outside [ref_beg, ref_end] are stepped over without a digar (:1206-1212); a '=' run interrupted by them is flushed at pos - eq_len"""
    orc = oracle
    rng = np.random.default_rng(14)
    ops = [(7, 100), (8, 1), (7, 200), (1, 4), (7, 150)]
    a = di.build(rng, ops, 1000, ref_pad=(-30, -40))
    q = np.full(a["qlen"], 40, np.uint8)
    r = orc.collect_digar_from_ref_seq(1000, a["mcig"], a["bseq"], q, a["ref_seq"], a["ref_beg"], a["ref_end"], 1, 10 ** 6, 10 ** 6)
    d = [tuple(int(v) for v in x[:4]) for x in r["digars"]]
    assert d == [(1031, 7, 70, 30), (1101, 8, 1, 100), (1102, 7, 200, 101), (1302, 1, 4, 301), (1342, 7, 110, 345)]   # the last run: 110 bases, then 40 stepped over, flushed at pos - eq_len
    assert (r["beg"], r["end"]) == (1001, 1451)
