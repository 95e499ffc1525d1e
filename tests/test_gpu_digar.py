"""SURVEY 8(f) f2, first part, on the GPU: EQX CIGARs -> digar lists + per-read noisy windows (digar_kernel.hip through lcd_digar_batch) vs the
oracle's restatement of collect_digar_from_eqx_cigar (src/bam_utils.c:701): every digar, interval, flag and counter identical -- on the
CIGARs of the reference's bundled HG002 chunk (rebuilt from the fixture's digar lists) and on seeded synthetic alignments with clips,
reference skips, low-quality bases, long gaps and the 'M' error case."""
import numpy as np
import pytest

import testdata_common as tc

pytestmark = pytest.mark.gpu


def _same(exp, got):
    assert exp["rc"] == got["rc"]
    if exp["rc"] == -2:   # an 'M' operation: the reference exits the process (src/bam_utils.c:791); nothing else is defined
        return
    assert exp["digars"].shape == got["digars"].shape and (exp["digars"] == got["digars"]).all()
    assert exp["noisy"].shape == got["noisy"].shape and (exp["noisy"] == got["noisy"]).all()
    assert exp["chunk_noisy"].shape == got["chunk_noisy"].shape and (exp["chunk_noisy"] == got["chunk_noisy"]).all()
    assert (exp["beg"], exp["end"], exp["n_cand"]) == (got["beg"], got["end"], got["n_cand"])


def _cigar_of(digars):
    out = []
    for pos, t, l, qi in digars:
        t, l = int(t), int(l)
        if out and t == 8 and (out[-1] & 0xf) == 8:
            out[-1] += 1 << 4
        else:
            out.append((l << 4) | t)
    return np.array(out, np.uint32)


def test_real_chunk_cigars(lcd, oracle):
    ch = tc.Chunk()
    rng = np.random.default_rng(7)
    cigs = [_cigar_of(d) for d in ch.digars]
    pos0 = [int(d[0][0]) - 1 for d in ch.digars]
    o = int(ch.z["ref_beg"]); reg_beg, reg_end = o + 20000, o + 180000
    for label, quals in (("high", [np.full(int(q), 40, np.uint8) for q in ch.qlen]),
                         ("mixed", [rng.choice([3, 8, 12, 30, 40], int(q), p=[0.05, 0.05, 0.1, 0.3, 0.5]).astype(np.uint8) for q in ch.qlen])):
        got = lcd.digar_batch(pos0, cigs, quals, reg_beg, reg_end, 135086622)
        n_win = 0
        for i in range(ch.n_reads):
            exp = oracle.collect_digar_from_eqx_cigar(pos0[i], cigs[i], quals[i], reg_beg, reg_end, 135086622)
            _same(exp, got[i])
            if label == "high":   # the digars the fixture was built from come back (positions, types, lengths, query offsets)
                assert (got[i]["digars"][:, :4] == ch.digars[i]).all()
            n_win += len(got[i]["noisy"])
        assert n_win > 100


def _synthetic(rng, noisy):
    ops, qlen = [], 0
    def add(op, ln):
        nonlocal qlen
        ops.append((ln << 4) | op)
        if op in (7, 8, 1, 4):
            qlen += ln
    if rng.random() < 0.4:
        if rng.random() < 0.3:
            add(5, int(rng.integers(1, 200)))
        add(4, int(rng.integers(1, 120)))
    for _ in range(int(rng.integers(20, 400))):
        add(7, int(rng.integers(1, 300 if not noisy else 30)))
        x = rng.random()
        if x < 0.45:
            add(8, int(rng.integers(1, 4)))
        elif x < 0.7:
            add(1, int(rng.integers(1, 60 if rng.random() < 0.1 else 5)))
        elif x < 0.95:
            add(2, int(rng.integers(1, 80 if rng.random() < 0.1 else 5)))
        else:
            add(3, int(rng.integers(10, 2000)))
    add(7, int(rng.integers(5, 100)))
    if rng.random() < 0.4:
        add(4, int(rng.integers(1, 150)))
    return np.array(ops, np.uint32), qlen


def test_synthetic_alignments(lcd, oracle):
    rng = np.random.default_rng(99)
    cigs, quals, pos0, pal = [], [], [], []
    for i in range(300):
        c, ql = _synthetic(rng, noisy=i % 3 == 0)
        cigs.append(c); quals.append(rng.choice([2, 9, 10, 25, 40], ql, p=[0.03, 0.04, 0.08, 0.35, 0.5]).astype(np.uint8))
        pos0.append(int(rng.integers(0, 5)) if i % 17 == 0 else int(rng.integers(1000, 900000))); pal.append(int(rng.integers(0, 4)) if i % 5 == 0 else 0)
    cigs.append(np.array([(50 << 4) | 7, (10 << 4) | 0, (20 << 4) | 7], np.uint32)); quals.append(np.full(80, 30, np.uint8)); pos0.append(5000); pal.append(0)   # 'M'
    for is_ont in (0, 1):
        got = lcd.digar_batch(pos0, cigs, quals, 200000, 700000, 1000000, is_ont=is_ont, pal_flags=np.array(pal, np.uint8))
        n_skip = n_chunk = 0
        for i in range(len(cigs)):
            exp = oracle.collect_digar_from_eqx_cigar(pos0[i], cigs[i], quals[i], 200000, 700000, 1000000, oracle.digar_opt(is_ont), pal[i] & 1, (pal[i] >> 1) & 1)
            _same(exp, got[i])
            n_skip += got[i]["rc"] == -1; n_chunk += len(got[i]["chunk_noisy"])
        assert got[-1]["rc"] == -2 and n_skip >= 1 and n_chunk > 50


def test_pre_process_noisy_regs(lcd, oracle):
    """f2, chunk level: lcd_pre_process_noisy_regs vs the same control flow over the REFERENCE's own cgranges (cr_index / cr_merge / cr_overlap
    compiled from /root/reference/src/cgranges.c into oracle/_ref): merged and filtered regions identical -- on the windows lcd_digar_batch
    finds in the bundled reads and on synthetic chunks with many tied starts, > 64 intervals and label-dependent merge distances"""
    if oracle.ref_cgranges() is None:
        pytest.skip("oracle/_ref/libcgranges_ref.so not built")
    ch = tc.Chunk()
    cigs = [_cigar_of(d) for d in ch.digars]; pos0 = [int(d[0][0]) - 1 for d in ch.digars]
    o = int(ch.z["ref_beg"])
    got = lcd.digar_batch(pos0, cigs, [np.full(int(q), 40, np.uint8) for q in ch.qlen], o, o + 210000, 135086622)
    keep = [g for g in got if g["rc"] == 0]
    chunk_noisy = np.concatenate([g["chunk_noisy"] for g in keep])
    rb = [g["beg"] for g in keep]; re_ = [g["end"] for g in keep]; ivs = [g["noisy"] for g in keep]
    for low in (np.zeros((0, 2), np.int64), np.array([[chunk_noisy[5][0] - 40, chunk_noisy[5][1] + 25], [chunk_noisy[40][0] + 3, chunk_noisy[40][1] + 300]], np.int64)):
        exp = oracle.ref_pre_process_noisy_regs(chunk_noisy, low, rb, re_, ivs)
        res = lcd.pre_process_noisy_regs(chunk_noisy, low, rb, re_, ivs)
        assert exp.shape == res.shape and (exp == res).all() and 10 < len(res) < len(chunk_noisy)
    rng = np.random.default_rng(5)
    for trial in range(6):
        n_reads = int(rng.integers(5, 400))
        centers = rng.integers(1000, 200000, int(rng.integers(3, 60)))
        rb, re_, ivs, cn = [], [], [], []
        for r in range(n_reads):
            b = int(rng.integers(0, 180000)); e = b + int(rng.integers(2000, 30000))
            mine = []
            for c in centers:
                if b < c < e and rng.random() < 0.5:
                    s = int(c + rng.integers(-3, 4) * (trial % 2)); mine.append([s, s + int(rng.integers(5, 400)), int(rng.choice([6, 12, 60, 300, 700]))])
            mine.sort()
            rb.append(b + 1); re_.append(e); ivs.append(np.array(mine, np.int64).reshape(-1, 3)); cn += mine
        cn = np.array(cn, np.int64).reshape(-1, 3)
        low = np.stack([centers[:5] - 50, centers[:5] + 80], 1) if trial % 3 == 0 else np.zeros((0, 2), np.int64)
        for min_dp, min_af in ((2, 0.2), (1, 0.0), (5, 0.5)):
            exp = oracle.ref_pre_process_noisy_regs(cn, low, rb, re_, ivs, min_dp, min_af)
            res = lcd.pre_process_noisy_regs(cn, low, rb, re_, ivs, min_dp, min_af)
            assert exp.shape == res.shape and (exp == res).all(), (trial, min_dp)


def test_pre_process_noisy_regs_committed_reference_vectors(lcd):
    """the pre_process_noisy_regs cases of tests/golden/cgranges_golden.json (expected regions from the reference's own cgranges, generated by
    tests/golden/make_cgranges_golden.py): no dependence on oracle/_ref being present on the box"""
    import json, os
    from conftest import ROOT
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "cgranges_golden.json")))["pre_process"]
    assert len(cases) >= 20
    for c in cases:
        got = lcd.pre_process_noisy_regs(np.array(c["chunk_noisy"], np.int64).reshape(-1, 3), np.array(c["low_comp"], np.int64).reshape(-1, 2), c["read_beg"], c["read_end"],
                                         [np.array(x, np.int64).reshape(-1, 3) for x in c["read_ivs"]], c["min_alt_dp"], c["min_af"])
        assert got.tolist() == c["out"]


def _lowcomp_seq(rng, n, n_frac):
    s = rng.integers(0, 4, n).astype(np.uint8)
    for _ in range(n // 400):
        p = int(rng.integers(0, n - 300)); k = int(rng.integers(1, 7)); ln = int(rng.integers(8, 250))
        unit = rng.integers(0, 4, k).astype(np.uint8)
        s[p:p + ln] = np.resize(unit, ln)
        if rng.random() < 0.3:      # imperfect repeat
            q = rng.integers(p, p + ln, max(1, ln // 15)); s[q] = rng.integers(0, 4, len(q))
    for _ in range(int(n * n_frac / 50) + (1 if n_frac else 0)):
        p = int(rng.integers(0, n - 10)); ln = int(rng.choice([1, 2, 3, 10, 60, 700, 3000]))
        s[p:p + ln] = 4
    return s


def test_sdust_matches_reference(lcd, oracle):
    """low-complexity intervals (chunk->low_comp_cr): the segmented device sdust vs the REFERENCE's own src/sdust.c on random sequences with
    tandem repeats / homopolymers / runs of N of every size (windows share words across an N), several (T, W), lengths around the segment
    size, and on the bundled chr11 slice"""
    if oracle.ref_cgranges() is None:
        pytest.skip("oracle/_ref/libcgranges_ref.so not built")
    rng = np.random.default_rng(21)
    n_iv = 0
    for n, nf in ((0, 0), (2, 0), (3, 0), (40, 0), (511, 0), (512, 0.0), (513, 0.01), (5000, 0), (5000, 0.02), (60000, 0.0), (60000, 0.01), (300000, 0.002)):
        s = _lowcomp_seq(rng, n, nf) if n > 400 else rng.integers(0, 2, n).astype(np.uint8)
        for T, W in ((5, 20), (20, 64), (10, 33)) if n <= 5000 else ((5, 20), (10, 33)):   # (W = 64: thousands of perfect intervals per window, slow on a lane)
            exp = oracle.ref_sdust(s, T, W); got = lcd.sdust(s, T, W)
            assert exp.shape == got.shape and (exp == got).all(), (n, nf, T, W)
            n_iv += len(got)
    ch = tc.Chunk()
    ref = ch.z["ref"]
    exp = oracle.ref_sdust(ref, 5, 20); got = lcd.sdust(ref, 5, 20)
    assert exp.shape == got.shape and (exp == got).all() and len(got) > 500
    letters = np.frombuffer(b"ACGTN", np.uint8)[np.minimum(ref[:50000], 4)]      # faidx gives letters (src/bam_utils.c:1564): same intervals
    assert (lcd.sdust(letters, 5, 20) == oracle.ref_sdust(ref[:50000], 5, 20)).all()
    assert n_iv > 1000


def test_sdust_batch_many_chunks_one_launch(lcd, oracle):
    """lcd_sdust_batch: the references of many chunks in one launch (sequences of different lengths, an empty one, an all-N one) -- every
    sequence's intervals equal its own single-sequence result and the reference's sdust()"""
    if oracle.ref_cgranges() is None:
        pytest.skip("oracle/_ref/libcgranges_ref.so not built")
    rng = np.random.default_rng(23)
    seqs = [_lowcomp_seq(rng, n, nf) for n, nf in ((50000, 0.0), (500, 0.0), (120000, 0.01), (700, 0.02), (33333, 0.0))] + [np.zeros(0, np.uint8), np.full(400, 4, np.uint8)]
    got = lcd.sdust_batch(seqs, 5, 20)
    assert len(got) == len(seqs)
    for s, g in zip(seqs, got):
        exp = oracle.ref_sdust(s, 5, 20) if len(s) else np.zeros((0, 2), np.int64)
        assert exp.shape == g.shape and (exp == g).all()
    assert sum(len(g) for g in got) > 100


def test_f2_edge_inputs(lcd, oracle):
    """empty and degenerate inputs of the f2 entry points: no reads, a read that is one clip + one match, all-N / very short sequences, no
    windows at all, regions nobody supports"""
    assert lcd.digar_batch([], [], [], 0, 100, 1000) == []
    cig = [np.array([(40 << 4) | 4, (25 << 4) | 7], np.uint32), np.array([(5 << 4) | 7], np.uint32), np.array([(3 << 4) | 5, (9 << 4) | 7, (2 << 4) | 5], np.uint32)]
    ql = [np.full(65, 30, np.uint8), np.full(5, 30, np.uint8), np.full(9, 30, np.uint8)]
    got = lcd.digar_batch([100, 0, 7], cig, ql, 0, 1000, 5000)
    for i in range(3):
        exp = oracle.collect_digar_from_eqx_cigar([100, 0, 7][i], cig[i], ql[i], 0, 1000, 5000)
        _same(exp, got[i])
    assert len(got[0]["noisy"]) == 1 and got[1]["digars"].shape == (1, 5)      # the 40-base soft clip marks its flank; a 5-base read is one digar
    for s in (np.zeros(0, np.uint8), np.array([1], np.uint8), np.array([0, 0], np.uint8), np.full(300, 4, np.uint8), np.zeros(1000, np.uint8)):
        got_s = lcd.sdust(s, 5, 20)
        if oracle.ref_cgranges() is not None and len(s):
            assert (got_s == oracle.ref_sdust(s, 5, 20)).all()
    assert len(lcd.sdust(np.zeros(1000, np.uint8), 5, 20)) == 1                  # a homopolymer is one interval
    assert len(lcd.pre_process_noisy_regs(np.zeros((0, 3), np.int64), np.zeros((0, 2), np.int64), [], [], [])) == 0
    one = np.array([[500, 560, 20]], np.int64)
    assert len(lcd.pre_process_noisy_regs(one, np.zeros((0, 2), np.int64), [], [], [])) == 0                       # no read spans it
    assert len(lcd.pre_process_noisy_regs(one, np.zeros((0, 2), np.int64), [1, 1], [2000, 2000], [one, one])) == 1   # two noisy reads: kept
    assert len(lcd.pre_process_noisy_regs(one, np.zeros((0, 2), np.int64), [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], [2000] * 11, [one] + [np.zeros((0, 3), np.int64)] * 10)) == 0  # 1 of 11: below min_alt_dp


# ---- the reference's three other digar sources (src/collect_var.c:1072-1079): cs tag, MD tag, reference bases ----
def _tag_inputs(rng, n, n_skip, pads=(0, 50)):
    import digar_inputs as di
    al, qs, pal = [], [], []
    for i in range(n):
        a = di.build(rng, di.eqx_ops(rng, noisy=i % 3 == 0, n_skip=n_skip), int(rng.integers(0, 12)) if i % 13 == 0 else int(rng.integers(1000, 900000)),
                     ref_pad=(int(rng.integers(*pads)), int(rng.integers(*pads))), lower=i % 2 == 0)
        al.append(a); qs.append(di.quals(rng, a["qlen"])); pal.append(int(rng.integers(0, 4)) if i % 5 == 0 else 0)
    return al, qs, np.array(pal, np.uint8)


def test_cs_and_md_tags(lcd, oracle):
    """lcd_digar_batch_tags vs oracle/digar_tags.c (collect_digar_from_cs_tag src/bam_utils.c:844, collect_digar_from_MD_tag :1010): short and long cs
    forms, MD runs carried over insertions, clips next to the contig ends (where the cs function's own clip rule shows), palindromic clips, both window
    sizes; malformed tags come back as -2"""
    rng = np.random.default_rng(21)
    al, qs, pal = _tag_inputs(rng, 200, n_skip=False)
    pos0 = [a["pos0"] for a in al]
    bad_cs = dict(al[3]); bad_cs["cs"] = al[3]["cs"][:40] + b"!" + al[3]["cs"][40:]
    for is_ont in (0, 1):
        opt = oracle.digar_opt(is_ont)
        tlen = 1000000
        for form, cig in (("cs", "eqx"), ("cs_long", "mcig")):
            got = lcd.digar_batch(pos0, [a[cig] for a in al], qs, 200000, 700000, tlen, is_ont=is_ont, pal_flags=pal, cs=[a[form] for a in al])
            for i, a in enumerate(al):
                _same(oracle.collect_digar_from_cs_tag(a["pos0"], a[cig], a[form], qs[i], 200000, 700000, tlen, opt, pal[i] & 1, (pal[i] >> 1) & 1), got[i])
        got = lcd.digar_batch(pos0, [a["mcig"] for a in al], qs, 200000, 700000, tlen, is_ont=is_ont, pal_flags=pal, md=[a["md"] for a in al])
        n_win = 0
        for i, a in enumerate(al):
            _same(oracle.collect_digar_from_MD_tag(a["pos0"], a["mcig"], a["md"], qs[i], 200000, 700000, tlen, opt, pal[i] & 1, (pal[i] >> 1) & 1), got[i])
            n_win += len(got[i]["noisy"])
        assert n_win > 100
    # reads with 'N' (MD only), and the error exits
    al2, qs2, _ = _tag_inputs(rng, 40, n_skip=True)
    got = lcd.digar_batch([a["pos0"] for a in al2], [a["mcig"] for a in al2], qs2, 1, 10 ** 6, 10 ** 7, md=[a["md"] for a in al2])
    for i, a in enumerate(al2):
        _same(oracle.collect_digar_from_MD_tag(a["pos0"], a["mcig"], a["md"], qs2[i], 1, 10 ** 6, 10 ** 7), got[i])
    a = al[3]
    got = lcd.digar_batch([a["pos0"]] * 3, [a["eqx"], a["mcig"], a["eqx"]], [qs[3]] * 3, 1, 10 ** 6, 10 ** 7, cs=[bad_cs["cs"], a["cs"], a["cs"]])
    assert [g["rc"] for g in got] == [-2, got[1]["rc"], got[1]["rc"]] and got[1]["rc"] in (0, -1)
    got = lcd.digar_batch([a["pos0"]] * 3, [a["mcig"], a["eqx"], a["mcig"]], [qs[3]] * 3, 1, 10 ** 6, 10 ** 7, md=[a["md"][:-3], a["md"], a["md"]])
    assert got[0]["rc"] == -2 and got[1]["rc"] == -2 and got[2]["rc"] in (0, -1)     # a tag shorter than its CIGAR; '=' / 'X' next to an MD tag
    for bad, g in ((a["md"][:-3], got[0]), ):
        assert oracle.collect_digar_from_MD_tag(a["pos0"], a["mcig"], bad, qs[3], 1, 10 ** 6, 10 ** 7)["rc"] == -2


def test_reference_comparison(lcd, oracle):
    """lcd_digar_batch_ref (base comparison + CIGAR rewrite on the device, then the same digar kernel) vs collect_digar_from_ref_seq (src/bam_utils.c:1179):
    'M' and '=' / 'X' CIGARs, lower-case reference, 'N' operations, reference windows that end inside the reads (bases stepped over, runs flushed late)"""
    rng = np.random.default_rng(22)
    for pads, n_skip in (((0, 50), False), ((-120, 30), True)):
        al, qs, pal = _tag_inputs(rng, 150, n_skip=n_skip, pads=pads)
        # lcd_digar_batch_ref takes ONE reference window per call (the chunk's): place every read in a common window by giving each call one read group
        for is_ont in (0, 1):
            opt = oracle.digar_opt(is_ont)
            n_win = 0
            for i, a in enumerate(al):
                for cig in ("mcig", "eqx") if i % 4 == 0 else ("mcig",):
                    got = lcd.digar_batch([a["pos0"]], [a[cig]], [qs[i]], 200000, 700000, 1000000, is_ont=is_ont, pal_flags=pal[i:i + 1], seqs=[a["bseq"]],
                                          ref=(a["ref_seq"], a["ref_beg"], a["ref_end"]))[0]
                    _same(oracle.collect_digar_from_ref_seq(a["pos0"], a[cig], a["bseq"], qs[i], a["ref_seq"], a["ref_beg"], a["ref_end"], 200000, 700000, 1000000, opt,
                                                            pal[i] & 1, (pal[i] >> 1) & 1), got)
                    n_win += len(got["noisy"])
            assert n_win > 50


def test_reference_comparison_real_chunk(lcd, oracle):
    """the bundled HG002 chunk: its reads' CIGARs with '=' / 'X' folded into 'M', the 4-bit bases the fixture keeps (only the stretches inside noisy regions
    are stored, the rest is 0 = 'N': tens of thousands of mismatching bases per read, i.e. a dense stress of the rewrite) against the chunk's reference
    slice, ONE launch for all reads; every read against the oracle, and inside the stored stretches the fixture's own digars come back"""
    ch = tc.Chunk()
    cigs = [_cigar_of(d) for d in ch.digars]
    mc = []
    for c in cigs:
        m = []
        for w in c:
            o = int(w) & 0xf; l = int(w) >> 4; o2 = 0 if o in (7, 8) else o
            if m and o2 == 0 and (m[-1] & 0xf) == 0:
                m[-1] += l << 4
            else:
                m.append((l << 4) | o2)
        mc.append(np.array(m, np.uint32))
    pos0 = [int(d[0][0]) - 1 for d in ch.digars]
    o = int(ch.z["ref_beg"]); ref = bytes(b"ACGTN"[b] for b in ch.z["ref"]) if ch.z["ref"].dtype == np.uint8 and ch.z["ref"].max() < 8 else bytes(ch.z["ref"])
    ref_end = o + len(ref) - 1
    quals = [np.full(int(q), 40, np.uint8) for q in ch.qlen]
    got = lcd.digar_batch(pos0, mc, quals, o + 20000, o + 180000, 135086622, seqs=ch.bseq, ref=(ref, o, ref_end))
    n_dig = 0
    for i in range(ch.n_reads):
        exp = oracle.collect_digar_from_ref_seq(pos0[i], mc[i], ch.bseq[i], quals[i], ref, o, ref_end, o + 20000, o + 180000, 135086622)
        _same(exp, got[i])
        n_dig += len(got[i]["digars"])
    assert n_dig > 100000


@pytest.mark.gpu
def test_region_read_slices_batch_equals_oracle(lcd, oracle):
    """SURVEY f2 -> region jobs: collect_noisy_read_info's digar walk for many (region, read) pairs in one launch (lcd_region_read_slices_batch,
    digar_kernel.hip lcd_slice_kernel) == the oracle's walk (oracle lcdo_read_region_slice, src/align.c:1392-1456) -- region ends inside '=' runs, on
    mismatches, inside short and long deletions (the flank rule), at insertions, in clips, outside the read; lists longer than one 64-digar step"""
    import digar_inputs as di
    rng = np.random.default_rng(31)
    digs, qlens, pairs = [], [], []
    for r in range(60):
        ops = di.eqx_ops(rng, noisy=r % 3 == 0, n_skip=False, n_ev=int(rng.integers(3, 400)))
        if r % 7 == 0:
            ops = [(5, 30)] + [o for o in ops if o[0] not in (4, 5)] + [(5, 12)]          # hard clips at both ends
        if r % 5 == 0:
            ops.insert(len(ops) // 2, (2, 40)); ops.insert(len(ops) // 2, (7, 9))      # a deletion longer than the flank
        a = di.build(rng, ops, 5000 + 13 * r)
        d = oracle.collect_digar_from_eqx_cigar(a["pos0"], a["eqx"], np.full(a["qlen"], 40, np.uint8), 1, 10 ** 7, 10 ** 7, oracle.DigarOpt(10, 5, 100, 30, 100, 4.0, 4.0))
        assert d["rc"] == 0 and len(d["digars"]) > 0
        digs.append(d["digars"][:, :4]); qlens.append(a["qlen"])
        lo, hi = int(d["digars"][0][0]), int(d["digars"][-1][0]) + 50
        cand = [int(x[0]) for x in d["digars"]] + [int(x[0]) + int(x[2]) - 1 for x in d["digars"]]   # digar starts and ends: the boundary cases
        for _ in range(25):
            if rng.random() < 0.6:
                b = int(rng.choice(cand)) + int(rng.integers(-1, 2))
            else:
                b = int(rng.integers(lo - 200, hi + 200))
            e = b + int(rng.integers(0, 600)) if rng.random() < 0.7 else int(rng.choice(cand)) + int(rng.integers(-1, 2))
            if e < b:
                b, e = e, b
            pairs.append((r, b, e))
    pr = np.array([p[0] for p in pairs], np.int32); pb = np.array([p[1] for p in pairs], np.int64); pe = np.array([p[2] for p in pairs], np.int64)
    rb, re, cv = lcd.region_read_slices_batch(pr, pb, pe, digs, qlens, flank=10)
    kinds = set()
    for i, (r, b, e) in enumerate(pairs):
        exp = oracle.read_region_slice(digs[r], qlens[r], b, e, 10)
        assert (int(rb[i]), int(re[i]), int(cv[i])) == exp, (i, r, b, e)
        kinds.add(exp[2])
    assert len(kinds) >= 5 and max(len(d) for d in digs) > 64
