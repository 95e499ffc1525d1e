"""Parity of the HIP kernels (through the C ABI) against the CPU oracle on the same seeded inputs.  Bit-exact: integer/byte work."""
import numpy as np
import pytest

from conftest import mutate

pytestmark = pytest.mark.gpu


def _pairs(rng, lens, rates, sv=0.0):
    out = []
    for L in lens:
        for r in rates:
            t = rng.integers(0, 4, L).astype(np.uint8)
            q = mutate(rng, t, r, sv)
            if len(q) == 0:
                q = np.array([0], np.uint8)
            out.append((t, q))
            out.append((t, q[: max(1, int(len(q) * rng.uniform(0.5, 1.0)))].copy()))
    return out


def test_edlib_small_traceback_regime(lcd, oracle):
    """K4 (src/align.c:222-254): distance, xgaps, n_eq, n_xid == oracle (== real edlib) where edlib uses the stored traceback"""
    rng = np.random.default_rng(11)
    pairs = _pairs(rng, [1, 2, 3, 17, 63, 64, 65, 100, 129, 300, 550, 777, 1000, 1500], [0.0, 0.01, 0.05, 0.15, 0.4])
    pairs += [(rng.integers(0, 4, 200).astype(np.uint8), rng.integers(0, 4, 90).astype(np.uint8)),
              (np.zeros(200, np.uint8), np.zeros(193, np.uint8)), (np.tile(np.array([0, 1], np.uint8), 100), np.tile(np.array([0, 1, 1], np.uint8), 66))]
    got = lcd.edlib_batch(pairs)
    for i, (t, q) in enumerate(pairs):
        d, neq, nxid = oracle.edlib_end2end_aln(t, q)
        assert got["dist"][i] == d, i
        assert got["xgaps"][i] == oracle.edlib_xgaps(t, q), (i, len(t), len(q))
        assert (got["n_eq"][i], got["n_xid"][i]) == (neq, nxid), i


def test_edlib_hirschberg_regime(lcd, oracle):
    """K4 above edlib's 1 MiB traceback threshold (edlib.cpp:1188): Hirschberg split rule reproduced, incl. q > 4096 (multi-tile)"""
    rng = np.random.default_rng(12)
    pairs = _pairs(rng, [2100, 3000, 5000], [0.01, 0.08])
    pairs += _pairs(rng, [9000], [0.03])
    got = lcd.edlib_batch(pairs)
    for i, (t, q) in enumerate(pairs):
        assert got["xgaps"][i] == oracle.edlib_xgaps(t, q), (i, len(t), len(q))
        d, neq, nxid = oracle.edlib_end2end_aln(t, q)
        assert (got["dist"][i], got["n_eq"][i], got["n_xid"][i]) == (d, neq, nxid), i


def test_edlib_kernel_matches_reference_golden_vectors(lcd):
    """the HIP K4 kernel itself against tests/golden/edlib_golden.json -- vectors produced by the REFERENCE's own edlib
    (tests/golden/make_edlib_golden.py): distance, the path-dependent xgaps and the =/XID counts, both traceback regimes (edlib.cpp:1188)"""
    import json
    import os
    from conftest import ROOT
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "edlib_golden.json")))["cases"]
    arr = lambda s: np.frombuffer(s.encode(), np.uint8) - ord("0")
    pairs = [(arr(c["target"]), arr(c["query"])) for c in cases]
    got = lcd.edlib_batch(pairs)
    assert len(cases) >= 70
    for i, c in enumerate(cases):
        assert got["dist"][i] == c["dist"] and got["xgaps"][i] == c["xgaps"], i
        assert got["n_eq"][i] == c["n_eq"] and got["n_xid"][i] == c["n_xid"], i


def test_align_h_wrappers(lcd, oracle):
    """the remaining exports of src/align.h: end2end_aln (:610, letters -> codes -> 2-piece WFA CIGAR) and wfa_collect_diff_ins_seq (:463, the longest
    run of large-only columns) == the same composition over the oracle's WFA; edlib_infix_aln == the oracle's (pinned to real edlib); wfa_heuristic_aln exists and fails loudly"""
    import ctypes as C
    from longcalld_amd import _lib
    lib = _lib.load_library()
    rng = np.random.default_rng(31)
    opt = lcd.default_opt()
    u8p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    t = rng.integers(0, 4, 400).astype(np.uint8)
    q = mutate(rng, t, 0.03, 0.004)
    letters = bytes(b"ACGT"[x] for x in t)
    cb = u32p()
    n = lib.lcd_end2end_aln(C.byref(opt), letters, len(letters), q.ctypes.data_as(u8p), len(q), C.byref(cb))
    exp = oracle.wfa_end2end_aln(t, q, gap_aln=opt.gap_aln)
    assert n == len(exp["cigar"]) and (np.ctypeslib.as_array(cb, shape=(n,)) == exp["cigar"]).all()
    libc.free(cb)
    small = rng.integers(0, 4, 300).astype(np.uint8)
    large = np.concatenate([small[:120], rng.integers(0, 4, 75).astype(np.uint8), small[120:200], rng.integers(0, 4, 20).astype(np.uint8), small[200:]])
    ds = u8p()
    n = lib.lcd_wfa_collect_diff_ins_seq(C.byref(opt), large.ctypes.data_as(u8p), len(large), small.ctypes.data_as(u8p), len(small), C.byref(ds))
    e = oracle.wfa_end2end_aln(large, small, gap_aln=opt.gap_aln)
    la, sa = e["pattern_alg"], e["text_alg"]
    best, pos, i = 0, -1, 0
    while i < len(la):
        if sa[i] == 5 and la[i] != 5:
            j = i
            while j < len(la) and sa[j] == 5 and la[j] != 5:
                j += 1
            if j - i > best:
                best, pos = j - i, i
            i = j
        else:
            i += 1
    assert n == best == 75 and (np.ctypeslib.as_array(ds, shape=(n,)) == la[pos:pos + best]).all()
    libc.free(ds)
    a, b = C.c_int(7), C.c_int(7)
    assert lib.lcd_edlib_infix_aln(t.ctypes.data_as(u8p), len(t), q.ctypes.data_as(u8p), len(q), C.byref(a), C.byref(b)) == oracle.edlib_infix_aln(t, q)[0]
    assert (a.value, b.value) == oracle.edlib_infix_aln(t, q)[1:]
    assert lib.lcd_wfa_heuristic_aln(t.ctypes.data_as(u8p), len(t), q.ctypes.data_as(u8p), len(q), 0, 6, 6, 2, 24, 1, C.byref(a), C.byref(b)) == -2


def test_edlib_hw_kernel_matches_reference_golden_vectors(lcd, oracle):
    """edlib_infix_aln (src/align.c:256-275): the HIP kernel in HW (infix) mode against vectors from the REFERENCE's own edlib (EDLIB_MODE_HW + TASK_PATH):
    distance, the target stretch (startLocations[0], endLocations[0]) and the path-dependent counts; then more pairs against the oracle (itself pinned to edlib)"""
    import json
    import os
    from conftest import ROOT
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "edlib_golden.json")))["hw_cases"]
    arr = lambda s: np.frombuffer(s.encode(), np.uint8) - ord("0")
    pairs = [(arr(c["target"]), arr(c["query"])) for c in cases]
    got = lcd.edlib_batch_hw(pairs)
    assert len(cases) >= 50
    for i, c in enumerate(cases):
        assert (got["dist"][i], got["start"][i], got["end"][i]) == (c["dist"], c["start"], c["end"]), i
        assert (got["xgaps"][i], got["n_eq"][i], got["n_xid"][i]) == (c["xgaps"], c["n_eq"], c["n_xid"]), i
    rng = np.random.default_rng(9)
    pairs = []
    for L in [10, 300, 1500, 4200, 6000]:          # > 4 096 query rows: more than one tile of Myers blocks; long ones take the Hirschberg regime
        for rate in [0.01, 0.15]:
            t = rng.integers(0, 4, L + 900).astype(np.uint8)
            a = int(rng.integers(0, 800))
            q = mutate(rng, t[a:a + L], rate)
            pairs.append((t, q))
    got = lcd.edlib_batch_hw(pairs)
    for i, (t, q) in enumerate(pairs):
        d, s0, e0, ops = oracle.edlib_hw(q, t)
        assert (got["dist"][i], got["start"][i], got["end"][i]) == (d, s0, e0), i
        assert (got["n_eq"][i], got["n_xid"][i]) == (int((ops == 0).sum()), int((ops != 0).sum())), i
        assert lcd.edlib_infix_aln(t, q) == oracle.edlib_infix_aln(t, q)


def test_edlib_empty(lcd):
    got = lcd.edlib_batch([(np.zeros(0, np.uint8), np.zeros(5, np.uint8)), (np.zeros(7, np.uint8), np.zeros(0, np.uint8))])
    assert list(got["dist"]) == [5, 7] and list(got["xgaps"]) == [0, 0]  # edlib.cpp:166-173: no alignment is produced


def test_wfa_matches_oracle(lcd, oracle):
    """K3 (src/align.c:374-460): score, CIGAR and both gapped rows == oracle for left- and right-aligned calls"""
    rng = np.random.default_rng(13)
    pairs = []
    for L in [0, 1, 2, 5, 30, 100, 300, 800, 2000]:
        for rate, sv in [(0, 0), (0.01, 0), (0.05, 0.004), (0.2, 0)]:
            t = rng.integers(0, 4, L).astype(np.uint8)
            p = mutate(rng, t, rate, sv)
            pairs += [(p, t), (t, p)]
    pairs += [(np.zeros(50, np.uint8), np.zeros(20, np.uint8)), (np.zeros(0, np.uint8), np.zeros(20, np.uint8))]
    for ga in (1, 2):
        got = lcd.wfa_batch(pairs, gap_aln=ga)
        for i, (p, t) in enumerate(pairs):
            exp = oracle.wfa_end2end_aln(p, t, gap_aln=ga)
            assert got[i]["score"] == exp["score"], (ga, i)
            assert (got[i]["cigar"] == exp["cigar"]).all() and len(got[i]["cigar"]) == len(exp["cigar"]), (ga, i)
            assert (got[i]["pattern_alg"] == exp["pattern_alg"]).all() and (got[i]["text_alg"] == exp["text_alg"]).all(), (ga, i)


def test_wfa_large_gap_retry(lcd, oracle):
    """SV-sized gap: score ~ 24+len forces the arena retry ladder; result still equals the oracle and the Gotoh optimum"""
    rng = np.random.default_rng(14)
    t = rng.integers(0, 4, 1500).astype(np.uint8)
    p = np.concatenate([t[:700], rng.integers(0, 4, 900).astype(np.uint8), t[700:]])
    got = lcd.wfa_batch([(p, t), (t, p)])
    for g, (a, b) in zip(got, [(p, t), (t, p)]):
        exp = oracle.wfa_end2end_aln(a, b)
        assert g["score"] == exp["score"] == oracle.gotoh2p_score(a, b)
        assert (g["cigar"] == exp["cigar"]).all()


def test_wfa_sv_10kb_gap_bounded_memory(lcd, oracle):
    """SURVEY H3 / configs[4]: a 10 kb insertion (score ~ 10 000, 14 000 diagonals).  The retained-wavefront backtrace of the oracle needs ~2 GB;
    the kernel keeps one byte of decisions per diagonal in 16 MB blocks + ring snapshots (checkpoint / recompute) -- same score, CIGAR and rows,
    left- and right-aligned, insertion and deletion, in < 50 MB of arena"""
    rng = np.random.default_rng(140)
    ref = rng.integers(0, 4, 2000).astype(np.uint8)
    ins = rng.integers(0, 4, 10000).astype(np.uint8)
    cons = np.concatenate([ref[:900], ins, ref[900:]])
    cons[400] = (cons[400] + 1) % 4; cons = np.delete(cons, [11500, 11501, 11502])
    pairs = [(ref, cons), (cons, ref)]
    for ga, prs in ((1, pairs), (2, pairs[:1])):   # (the oracle needs ~9 s and 2 GB per alignment of this size)
        got = lcd.wfa_batch(prs, gap_aln=ga)
        for g, (p, t) in zip(got, prs):
            exp = oracle.wfa_end2end_aln(p, t, gap_aln=ga)
            assert g["score"] == exp["score"] and 10000 < g["score"] < 10100
            assert len(g["cigar"]) == len(exp["cigar"]) and (g["cigar"] == exp["cigar"]).all()
            assert (g["pattern_alg"] == exp["pattern_alg"]).all() and (g["text_alg"] == exp["text_alg"]).all()
            assert lcd.wfa_arena_bytes(len(p), len(t), g["score"]) < 50e6
            assert lcd.wfa_arena_bytes(len(p), len(t), 4 * g["score"]) < 50e6     # (the bound a retry would ask for)


def test_wfa_many_small_blocks(lcd, oracle, monkeypatch):
    """the checkpoint / recompute path with tiny blocks (LCD_WFA_BLOCK_KB=8: a few dozen scores per block, tens of blocks per alignment):
    the backtrace crosses block boundaries inside gap chains and between events -- == oracle"""
    monkeypatch.setenv("LCD_WFA_BLOCK_KB", "8")
    rng = np.random.default_rng(141)
    pairs = []
    for L, rate, sv in [(300, 0.1, 0), (800, 0.2, 0), (1500, 0.05, 0.004), (600, 0.0, 0)]:
        t = rng.integers(0, 4, L).astype(np.uint8)
        p = mutate(rng, t, rate, sv)
        pairs += [(p, t), (t, p)]
    t = rng.integers(0, 4, 700).astype(np.uint8)
    pairs += [(np.concatenate([t[:300], rng.integers(0, 4, 500).astype(np.uint8), t[300:]]), t), (t, np.concatenate([t[:100], t[450:]]))]
    for ga in (1, 2):
        got = lcd.wfa_batch(pairs, gap_aln=ga)
        for i, (p, t) in enumerate(pairs):
            exp = oracle.wfa_end2end_aln(p, t, gap_aln=ga)
            assert got[i]["score"] == exp["score"], (ga, i)
            assert len(got[i]["cigar"]) == len(exp["cigar"]) and (got[i]["cigar"] == exp["cigar"]).all(), (ga, i)
            assert (got[i]["pattern_alg"] == exp["pattern_alg"]).all() and (got[i]["text_alg"] == exp["text_alg"]).all(), (ga, i)


def test_wfa_other_penalties(lcd, oracle):
    """ring depths follow the penalties (max(x, o1+e1, o2+e2) + 1 rows of M): a non-default scoring == oracle"""
    rng = np.random.default_rng(142)
    pairs = []
    for L in (50, 400, 900):
        t = rng.integers(0, 4, L).astype(np.uint8)
        pairs += [(mutate(rng, t, 0.08, 0.003), t)]
    for (b, q, e, q2, e2) in [(4, 6, 2, 24, 1), (6, 6, 2, 24, 1), (5, 8, 3, 30, 2), (2, 3, 1, 10, 1)]:
        got = lcd.wfa_batch(pairs, gap_aln=2, b=b, q=q, e=e, q2=q2, e2=e2)
        for g, (p, t) in zip(got, pairs):
            exp = oracle.wfa_end2end_aln(p, t, gap_aln=2, b=b, q=q, e=e, q2=q2, e2=e2)
            assert g["score"] == exp["score"] and len(g["cigar"]) == len(exp["cigar"]) and (g["cigar"] == exp["cigar"]).all()


def test_wfa_per_call_mirror_ownership(lcd, oracle):
    """lcd_wfa_end2end_aln: malloc'd cigar + one-block rows (text row = block + plen+tlen+1), src/align.c:288-291,490"""
    rng = np.random.default_rng(15)
    t = rng.integers(0, 4, 300).astype(np.uint8)
    p = mutate(rng, t, 0.03)
    cigar, pa, ta = lcd.wfa_end2end_aln(p, t)
    exp = oracle.wfa_end2end_aln(p, t)
    assert (cigar == exp["cigar"]).all() and (pa == exp["pattern_alg"]).all() and (ta == exp["text_alg"]).all()


def _poa_chains(rng, n_chains, L, rate, n_reads, two_haps):
    chains = []
    for _ in range(n_chains):
        truth = rng.integers(0, 4, L).astype(np.uint8)
        hap2 = truth.copy()
        if two_haps:
            for p in range(10, L - 10, max(20, L // 6)):
                hap2[p] = (hap2[p] + 1) % 4
        reads = [mutate(rng, truth if (i % 2 == 0 or not two_haps) else hap2, rate) for i in range(n_reads)]
        chains.append(reads)
    return chains


@pytest.mark.parametrize("L,rate", [(30, 0.001), (120, 0.02), (500, 0.001), (500, 0.05), (1500, 0.01)])
def test_poa_k1_matches_oracle(lcd, oracle, L, rate):
    """K1 (src/align.c:762-857), all reads full-cover: consensus and every MSA row == oracle"""
    rng = np.random.default_rng(100 + L)
    chains = _poa_chains(rng, 6, L, rate, 9, False)
    got = lcd.poa_batch([dict(mode=0, reads=r) for r in chains])
    for reads, g in zip(chains, got):
        exp = oracle.poa_partial_aln_msa_cons(reads, [12] * len(reads))
        assert g["status"] == 0 and g["n_cons"] == exp["n_cons"] == 1 and g["msa_len"] == exp["msa_len"]
        assert (g["cons"][0] == exp["cons"][0]).all() and len(g["cons"][0]) == len(exp["cons"][0])
        for a, b in zip(g["msa"], exp["msa"]):
            assert (a == b).all()


@pytest.mark.parametrize("L,rate", [(60, 0.001), (300, 0.02), (700, 0.001), (400, 0.08)])
def test_poa_k2_matches_oracle(lcd, oracle, L, rate):
    """K2 (src/align.c:872-943): unbanded MSA, 2-cluster split, per-cluster consensus == oracle"""
    rng = np.random.default_rng(200 + L)
    chains = _poa_chains(rng, 5, L, rate, 12, True) + _poa_chains(rng, 2, L, rate, 7, False)
    got = lcd.poa_batch([dict(mode=1, reads=r) for r in chains])
    for reads, g in zip(chains, got):
        exp = oracle.poa_aln_msa_cons(reads, 2)
        assert g["status"] == 0 and g["n_cons"] == exp["n_cons"] and g["msa_len"] == exp["msa_len"]
        for c in range(g["n_cons"]):
            assert (g["cons"][c] == exp["cons"][c]).all() and len(g["cons"][c]) == len(exp["cons"][c])
            assert (g["clu"][c] == exp["clu"][c]).all()
        for a, b in zip(g["msa"], exp["msa"]):
            assert (a == b).all()


@pytest.mark.parametrize("match,mismatch", [(2, 6), (31, 32), (40, 50), (1, 33)])
def test_poa_scores_outside_the_lean_rows_score_fields(lcd, oracle, match, mismatch):
    """round 5: the lean rows take a cell's substitution score from a 6-bit field of a per-row scalar word (match and -mismatch in [-32, 31]); scoring outside that
    range makes them decline the read (poa_kernel.hip align_lean "6-bit field") and the windowed / generic rows take it.  K1 and K2 (certified band and full rows) ==
    oracle with the same scoring, at the range's edge and beyond it"""
    from longcalld_amd import align
    from oracle import pyoracle
    opt, oopt = align.default_opt(), pyoracle.default_opt()
    opt.match = oopt.match = match; opt.mismatch = oopt.mismatch = mismatch
    rng = np.random.default_rng(9100 + match)
    k1 = _poa_chains(rng, 3, 300, 0.01, 9, False)
    got = lcd.poa_batch([dict(mode=0, reads=r) for r in k1], opt)
    for reads, g in zip(k1, got):
        exp = oracle.poa_partial_aln_msa_cons(reads, [12] * len(reads), oopt)
        assert g["status"] == 0 and g["n_cons"] == exp["n_cons"] == 1 and g["msa_len"] == exp["msa_len"]
        for a, b in zip(g["msa"], exp["msa"]):
            assert (a == b).all()
    k2 = _poa_chains(rng, 3, 300, 0.004, 12, True)
    got = lcd.poa_batch([dict(mode=1, reads=r) for r in k2], opt)
    for reads, g in zip(k2, got):
        _check_k2(g, oracle.poa_aln_msa_cons(reads, 2, oopt))


def test_poa_lean_rows_window_that_stays_jumps_and_restarts(lcd, oracle):
    """round 5, one cell per lane: the lanes follow the diagonal and the window either moves one column per row or stays (poa_kernel.hip align_lean, "lanes follow the
    DIAGONAL").  Reads with deletions of 1 - 30 bases against the backbone (the band's first column stalls: the window stays), insertions (it jumps: lanes idle on the
    left until a general row re-anchors), both at the very start and end of the read, homopolymer runs, and reads with N bases (score 0 against everything: the
    fifth score field) -- K1 with the adaptive band and K2 with certified intervals and with full rows, all == oracle"""
    rng = np.random.default_rng(9200)
    truth = rng.integers(0, 4, 420).astype(np.uint8)
    truth[200:230] = truth[200]                                    # a homopolymer run: ties between gap placements
    def edit(r, dels=(), ins=()):
        out, i = [], 0
        dels = dict(dels); ins = dict(ins)
        while i < len(r):
            if i in ins: out.extend(rng.integers(0, 4, ins[i]))
            if i in dels: i += dels[i]; continue
            out.append(r[i]); i += 1
        return np.array(out, np.uint8)
    reads = [truth.copy(), edit(truth, dels={0: 7}), edit(truth, dels={50: 1, 120: 30, 300: 12}), edit(truth, ins={0: 9, 210: 25}), edit(truth, ins={60: 14}, dels={400: 15}),
             edit(truth, dels={205: 9}), edit(truth, ins={419: 11}), mutate(rng, truth, 0.02), mutate(rng, truth, 0.02), edit(truth, dels={10: 3, 20: 3, 30: 3, 40: 3, 50: 3, 60: 3})]
    withn = [r.copy() for r in reads]
    for r in withn[1::2]:
        r[rng.integers(0, len(r), 6)] = 4
    for rs in (reads, withn):
        g = lcd.poa_batch([dict(mode=0, reads=rs)])[0]
        exp = oracle.poa_partial_aln_msa_cons(rs, [12] * len(rs))
        assert g["status"] == 0 and g["msa_len"] == exp["msa_len"]
        for a, b in zip(g["msa"], exp["msa"]):
            assert (a == b).all()
    g2 = lcd.poa_batch([dict(mode=1, reads=reads)])[0]
    _check_k2(g2, oracle.poa_aln_msa_cons(reads, 2))


def test_poa_invariants_large(lcd):
    """size-independent properties on a region longer than the oracle is comfortable with: every MSA row de-gaps to its read,
    the consensus row de-gaps to the consensus, clusters partition the reads"""
    rng = np.random.default_rng(300)
    truth = rng.integers(0, 4, 6000).astype(np.uint8)
    reads = [mutate(rng, truth, 0.01) for _ in range(20)]
    g = lcd.poa_batch([dict(mode=0, reads=reads)])[0]
    assert g["status"] == 0 and g["n_cons"] == 1
    for r, row in zip(reads, g["msa"]):
        assert (row[row != 5] == r).all()
    crow = g["msa"][len(reads)]
    assert (crow[crow != 5] == g["cons"][0]).all()
    assert abs(len(g["cons"][0]) - 6000) < 60


def _sv_haps(rng, L):
    """two haplotypes that differ by SNPs, a 40-base deletion and a 30-base insertion: the graph gets long bubbles, i.e. rows whose
    predecessor is far back in topological order (the kernel's spilled rows)"""
    h1 = rng.integers(0, 4, L).astype(np.uint8)
    h2 = h1.copy()
    for p in range(15, L - 15, max(25, L // 8)):
        h2[p] = (h2[p] + 1) % 4
    a, b = L // 3, 2 * L // 3
    h2 = np.concatenate([h2[:a], h2[a + 40:b], rng.integers(0, 4, 30).astype(np.uint8), h2[b:]])
    return h1, h2


def _check_k2(got, exp):
    assert got["status"] == 0 and got["n_cons"] == exp["n_cons"] and got["msa_len"] == exp["msa_len"]
    for c in range(got["n_cons"]):
        assert len(got["cons"][c]) == len(exp["cons"][c]) and (got["cons"][c] == exp["cons"][c]).all()
        assert (got["clu"][c] == exp["clu"][c]).all()
    for a, b in zip(got["msa"], exp["msa"]):
        assert (a == b).all()


@pytest.mark.parametrize("L", [230, 900, 2600])
def test_poa_k2_bubbles_all_workgroup_classes(lcd, oracle, L, monkeypatch):
    """K2 on het-SV regions in the 64-, 256- and 1024-thread classes (the last two run the systolic rows): == oracle.  (LCD_CERT=0: full rows; the
    certified band of the single-wavefront class has its own test below)"""
    monkeypatch.setenv("LCD_CERT", "0")
    rng = np.random.default_rng(400 + L)
    h1, h2 = _sv_haps(rng, L)
    reads = [mutate(rng, h1 if i % 2 == 0 else h2, 0.002) for i in range(10)]
    got = lcd.poa_batch([dict(mode=1, reads=reads)])[0]
    _check_k2(got, oracle.poa_aln_msa_cons(reads, 2))


@pytest.mark.parametrize("L,rate,sv", [(300, 0.002, 12), (900, 0.001, 40), (900, 0.02, 25), (2600, 0.002, 60), (2600, 0.003, 400), (1500, 0.06, 30),
                                       (14000, 0.001, 40)])   # (14 kb: scores of 28 000, the top of what the 16-bit LDS ring of these chains holds)
def test_poa_k2_certified_band(lcd, oracle, L, rate, sv, monkeypatch):
    """K2 with rows restricted to the certified band (single-wavefront class, poa_kernel.hip align_certified) == full rows == oracle: clean and noisy reads,
    small and large het insertions (a 400-base one needs more than the 256-column window: the chain comes back with LCD_ERR_CERT and is re-run with
    full rows -- same result, status 0), reads of very different lengths in one chain"""
    rng = np.random.default_rng(4500 + L + sv)
    h1 = rng.integers(0, 4, L).astype(np.uint8)
    p = L // 3
    h2 = np.concatenate([h1[:p], rng.integers(0, 4, sv).astype(np.uint8), h1[p:]])
    h2[L // 2 + sv] = (h2[L // 2 + sv] + 1) % 4
    reads = [mutate(rng, h1 if i % 2 == 0 else h2, rate) for i in range(12)]
    reads[5] = reads[5][: len(reads[5]) * 2 // 3]       # a read that stops early: forced gap columns at the end
    jobs = [dict(mode=1, reads=reads), dict(mode=1, reads=reads[::-1])]
    monkeypatch.setenv("LCD_CERT", "2")                  # (2: noisy reads as well)
    a = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_CERT_SOLO_LEN", "200")       # the same rows as wavefront 0 of a 256-thread workgroup (off by default)
    a2 = lcd.poa_batch(jobs)
    monkeypatch.delenv("LCD_CERT_SOLO_LEN")
    monkeypatch.setenv("LCD_DBG", "16")                  # a read that outgrows the window ends its chain: the host's re-run with full rows
    a3 = lcd.poa_batch(jobs)
    monkeypatch.delenv("LCD_DBG")
    monkeypatch.setenv("LCD_CERT", "0")
    b = lcd.poa_batch(jobs)
    for x, y in zip(a, a3):
        assert x["status"] == 0 and y["status"] == 0 and x["msa_len"] == y["msa_len"] and all((r == q).all() for r, q in zip(x["msa"], y["msa"]))
    for x, y in zip(a, a2):
        assert x["status"] == 0 and y["status"] == 0 and x["msa_len"] == y["msa_len"] and all((r == q).all() for r, q in zip(x["msa"], y["msa"]))
    for x, y, j in zip(a, b, jobs):
        assert x["status"] == 0 and y["status"] == 0 and x["n_cons"] == y["n_cons"] and x["msa_len"] == y["msa_len"]
        for r, q in zip(x["msa"], y["msa"]):
            assert (r == q).all()
        for c in range(x["n_cons"]):
            assert (x["cons"][c] == y["cons"][c]).all() and (x["clu"][c] == y["clu"][c]).all()
        if L <= 1500:
            _check_k2(x, oracle.poa_aln_msa_cons(j["reads"], 2))


def _same_chains(xs, ys):
    for x, y in zip(xs, ys):
        assert x["status"] == 0 and y["status"] == 0 and x["n_cons"] == y["n_cons"] and x["msa_len"] == y["msa_len"]
        for r, q in zip(x["msa"], y["msa"]):
            assert (r == q).all()
        for c in range(x["n_cons"]):
            assert (x["cons"][c] == y["cons"][c]).all() and (x["clu"][c] == y["clu"][c]).all()


def _het_chain(rng, L, n, rate=0.001, sv=40):
    h1 = rng.integers(0, 4, L).astype(np.uint8)
    h2 = np.concatenate([h1[:L // 3], rng.integers(0, 4, sv).astype(np.uint8), h1[L // 3:]])
    for p in range(L // 2, L - 20, max(200, L // 12)):     # SNP bubbles all the way to the end: rows with two predecessors read the ring where the scores are largest
        h2[p + sv] = (h2[p + sv] + 1) % 4
    return [mutate(rng, h1 if i % 2 == 0 else h2, rate) for i in range(n)]


def test_poa_k2_int16_ring_bonus_overflow(lcd, oracle, monkeypatch):
    """The 16-bit LDS ring's range guard (poa_kernel.hip align_lean "int16 range", lcd_host.cpp chain_caps): every traversed edge adds ilog2(weight), so on a DEEP
    chain the best score is far above qlen * match -- 34 reads x 4.8 kb reach 4 800 x (2 + 5) = 33 600 > int16 although 2 x 4 800 is nowhere near it (round 4's
    guard looked at the matches only and let such a chain's ring saturate).  Default (the host lays the chain out for a 32-bit ring) == LCD_RING16=2 (16-bit pools
    forced: the KERNEL's guard sends the late reads through the generic rows) == LCD_RING16=0 == full rows == oracle"""
    rng = np.random.default_rng(5100)
    jobs = [dict(mode=1, reads=_het_chain(rng, 4800, 34))]
    monkeypatch.setenv("LCD_CERT", "1")
    a = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_RING16", "2")
    b = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_RING16", "0")
    c = lcd.poa_batch(jobs)
    monkeypatch.delenv("LCD_RING16")
    monkeypatch.setenv("LCD_CERT", "0")
    d = lcd.poa_batch(jobs)
    _same_chains(a, b); _same_chains(a, c); _same_chains(a, d)
    _check_k2(a[0], oracle.poa_aln_msa_cons(jobs[0]["reads"], 2))


_LONG_K2 = [(15900, 6), (16100, 6), (30000, 5)]
_long_k2_oracle = {}


def _long_k2_reads(L, n):
    return _het_chain(np.random.default_rng(5200 + L), L, n)


@pytest.mark.parametrize("L,n", _LONG_K2)
def test_poa_k2_int16_ring_guard_long_reads(lcd, L, n, monkeypatch):
    """reads around and beyond the int16 limit of the matches alone (2 x 16 000 = 32 000; regions go to 50 kb, src/call_var_main.h:36): 16-bit pools forced
    (LCD_RING16=2, the kernel's guard decides per read) == default == 32-bit rings == full rows == ORACLE (the three chains' scalar oracle runs --
    1 - 3 minutes and 3 - 11 GB each -- side by side in spawned workers, once for the three cases)"""
    from conftest import oracle_poa_many
    if not _long_k2_oracle:
        res = oracle_poa_many([("k2", _long_k2_reads(l_, n_), 2) for l_, n_ in _LONG_K2])
        for (l_, n_), r in zip(_LONG_K2, res):
            _long_k2_oracle[(l_, n_)] = r
    jobs = [dict(mode=1, reads=_long_k2_reads(L, n))]
    monkeypatch.setenv("LCD_CERT", "1")
    a = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_RING16", "2")
    b = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_RING16", "0")
    c = lcd.poa_batch(jobs)
    monkeypatch.delenv("LCD_RING16")
    monkeypatch.setenv("LCD_CERT", "0")
    d = lcd.poa_batch(jobs)
    _same_chains(a, b); _same_chains(a, c); _same_chains(a, d)
    for r, row in zip(jobs[0]["reads"], a[0]["msa"]):
        assert (row[row != 5] == r).all()
    _check_k2(a[0], _long_k2_oracle[(L, n)])


def test_poa_regions_near_the_50kb_cap_and_a_1000_read_region(lcd):
    """the longest region longcallD hands over is 50 kb (LONGCALLD_NOISY_REG_MAX_LEN, src/call_var_main.h:36) and the deepest 1 000 reads (:42).
    A K1 chain of 48 kb reads (banded: the adaptive band is 10 + 480 columns either side), a K2 chain of 45 kb reads (unbanded in the oracle: 2 x 2 G cells) and a
    1 000-read K1 chain of short reads (every read-set word of an edge in use) == oracle, the oracle's runs side by side in spawned workers"""
    from conftest import oracle_poa_many
    rng = np.random.default_rng(5600)
    h = rng.integers(0, 4, 48000).astype(np.uint8)
    k1_long = [mutate(rng, h, 0.001) for _ in range(5)]
    k2_long = _het_chain(rng, 45000, 3)
    s = rng.integers(0, 4, 150).astype(np.uint8)
    s2 = s.copy(); s2[70] = (s2[70] + 1) % 4
    k1_deep = [mutate(rng, s if i % 3 else s2, 0.002) for i in range(1000)]
    exp = oracle_poa_many([("k2", k2_long, 2), ("k1", k1_long, [12] * len(k1_long)), ("k1", k1_deep, [12] * len(k1_deep))])
    got = lcd.poa_batch([dict(mode=1, reads=k2_long), dict(mode=0, reads=k1_long), dict(mode=0, reads=k1_deep)])
    _check_k2(got[0], exp[0])
    for g, e in zip(got[1:], exp[1:]):
        assert g["status"] == 0 and g["n_cons"] == e["n_cons"] == 1 and g["msa_len"] == e["msa_len"]
        assert len(g["cons"][0]) == len(e["cons"][0]) and (g["cons"][0] == e["cons"][0]).all()
        for a, b in zip(g["msa"], e["msa"]):
            assert (a == b).all()


def test_poa_k2_int16_ring_guard_large_penalties(lcd, oracle, monkeypatch):
    """gap_open2 = 4 000: the WORST-case bound (a gap over all rows plus a gap over all columns: 2 x (4 000 + ...) + ...) leaves the int16 range for a 900-base read, and
    real cells do go below -8 000 at the corners of the intervals.  Forced 16-bit pools == default == 32-bit == full rows == oracle with the same penalties"""
    from longcalld_amd import align
    from oracle import pyoracle
    rng = np.random.default_rng(5300)
    jobs = [dict(mode=1, reads=_het_chain(rng, 900, 12, rate=0.004, sv=25))]
    opt, oopt = align.default_opt(), pyoracle.default_opt()
    opt.gap_open2 = oopt.gap_open2 = 4000
    outs = []
    for cert, r16 in (("1", None), ("1", "2"), ("1", "0"), ("0", None)):
        monkeypatch.setenv("LCD_CERT", cert)
        if r16 is None: monkeypatch.delenv("LCD_RING16", raising=False)
        else: monkeypatch.setenv("LCD_RING16", r16)
        outs.append(lcd.poa_batch(jobs, opt))
    for o in outs[1:]:
        _same_chains(outs[0], o)
    _check_k2(outs[0][0], oracle.poa_aln_msa_cons(jobs[0]["reads"], 2, oopt))


def test_poa_reads_with_n(lcd, oracle):
    """N bases (code 4, score 0 against everything, src/align.c:316) in reads: K2 leaves the systolic rows for the windowed ones"""
    rng = np.random.default_rng(500)
    h1, h2 = _sv_haps(rng, 400)
    reads = [mutate(rng, h1 if i % 2 == 0 else h2, 0.01) for i in range(9)]
    for r in reads[::3]:
        r[rng.integers(5, len(r) - 5, 3)] = 4
    got = lcd.poa_batch([dict(mode=1, reads=reads), dict(mode=0, reads=reads)])
    _check_k2(got[0], oracle.poa_aln_msa_cons(reads, 2))
    exp = oracle.poa_partial_aln_msa_cons(reads, [12] * len(reads))
    assert got[1]["status"] == 0 and got[1]["msa_len"] == exp["msa_len"] and (got[1]["cons"][0] == exp["cons"][0]).all()


def test_poa_generic_rows_agree_with_windowed(lcd, monkeypatch):
    """the fallback for rows wider than the window (values in HBM, value backtrack; LCD_DBG=8 forces it) gives the same chains"""
    rng = np.random.default_rng(600)
    jobs = []
    for L, rate, mode in [(150, 0.01, 0), (600, 0.05, 0), (300, 0.02, 1), (1200, 0.002, 1)]:
        h1, h2 = _sv_haps(rng, L)
        jobs.append(dict(mode=mode, reads=[mutate(rng, h1 if i % 2 == 0 else h2, rate) for i in range(8)]))
    a = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_DBG", "8")
    b = lcd.poa_batch(jobs)
    monkeypatch.delenv("LCD_DBG")
    for x, y in zip(a, b):
        assert x["status"] == 0 and y["status"] == 0 and x["n_cons"] == y["n_cons"] and x["msa_len"] == y["msa_len"]
        for r, q in zip(x["msa"], y["msa"]):
            assert (r == q).all()
        for c in range(x["n_cons"]):
            assert (x["cons"][c] == y["cons"][c]).all() and (x["clu"][c] == y["clu"][c]).all()


def test_poa_k2_longer_than_the_window(lcd):
    """K2 on reads longer than the widest window (4 096 columns): the unbanded rows go through the generic HBM rows.  Too large for the
    oracle in test time; size-independent properties instead: every MSA row de-gaps to its read, consensus rows de-gap to the consensus,
    clusters partition the reads, and the two haplotypes are separated"""
    rng = np.random.default_rng(700)
    h1 = rng.integers(0, 4, 4600).astype(np.uint8)
    h2 = h1.copy()
    for p in range(40, 4560, 300):
        h2[p] = (h2[p] + 1) % 4
    reads = [mutate(rng, h1 if i % 2 == 0 else h2, 0.001) for i in range(10)]
    g = lcd.poa_batch([dict(mode=1, reads=reads)])[0]
    assert g["status"] == 0 and g["n_cons"] == 2
    for r, row in zip(reads, g["msa"]):
        assert (row[row != 5] == r).all()
    for c in range(2):
        crow = g["msa"][len(reads) + c]
        assert (crow[crow != 5] == g["cons"][c]).all()
    members = sorted(int(x) for c in range(2) for x in g["clu"][c])
    assert members == list(range(len(reads)))
    assert {int(x) % 2 for x in g["clu"][0]} in ({0}, {1}) and {int(x) % 2 for x in g["clu"][1]} in ({0}, {1})


def test_poa_k2_column_tiles_agree_with_generic_rows_and_oracle(lcd, oracle, monkeypatch):
    """K2 reads longer than the 4 096-column window are swept in column tiles (align_unbanded: per-row boundary H + F carries through HBM, codes
    row-major over the whole read).  Two and three tiles, clean and noisy reads, an SV between the haplotypes: identical to the generic HBM rows
    (LCD_DBG=8) bit for bit, and -- at the size the scalar oracle still does in seconds -- identical to the oracle"""
    rng = np.random.default_rng(710)
    jobs = []
    for L, rate, n in [(4300, 0.002, 6), (8700, 0.01, 6), (6100, 0.05, 8), (4095, 0.001, 5), (4096, 0.001, 5)]:
        h1 = rng.integers(0, 4, L).astype(np.uint8)
        h2 = np.concatenate([h1[:L // 3], rng.integers(0, 4, 150).astype(np.uint8), h1[L // 3:]])
        h2[L // 2] = (h2[L // 2] + 1) % 4
        jobs.append(dict(mode=1, reads=[mutate(rng, h1 if i % 2 == 0 else h2, rate) for i in range(n)]))
    a = lcd.poa_batch(jobs)
    monkeypatch.setenv("LCD_DBG", "8")
    b = lcd.poa_batch(jobs)
    monkeypatch.delenv("LCD_DBG")
    for x, y in zip(a, b):
        assert x["status"] == 0 and y["status"] == 0 and x["n_cons"] == y["n_cons"] and x["msa_len"] == y["msa_len"]
        for r, q in zip(x["msa"], y["msa"]):
            assert (r == q).all()
        for c in range(x["n_cons"]):
            assert (x["cons"][c] == y["cons"][c]).all() and (x["clu"][c] == y["clu"][c]).all()
    exp = oracle.poa_aln_msa_cons(jobs[0]["reads"])
    assert a[0]["n_cons"] == exp["n_cons"] and a[0]["msa_len"] == exp["msa_len"]
    for c in range(exp["n_cons"]):
        assert (a[0]["cons"][c] == exp["cons"][c]).all()
    for r, q in zip(a[0]["msa"], exp["msa"]):
        assert (r == q).all()


def test_wider_window_after_a_ring_that_outgrew_the_pool_layout(tmp_path):
    """DESIGN 8.7 (the round-3 "hang"): a noisy K1 chain with eight ring slots in a 16 KB pool whose band outgrows 64 and then 128 columns.  The eight slots of 128 columns
    lie over the first-predecessor distances, and the four-cells-per-lane rows that ran next used to follow the overwritten distances in their backtrack -- out of a
    row's band, from a row to itself, for ever.  The chain (tests/golden/ring8_chain.bin: dumped by LCD_DUMP_CHAIN from the failing ONT-shape submission) must now
    come back with the results of the default layout, well inside the watchdog.  (Separate processes: the layout switches are read once per process.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = os.path.join(root, "tests", "golden", "ring8_chain.bin")
    outs = []
    for extra in ({}, {"LCD_RING_K_MAXLDS_KB": "16", "LCD_LDS_CAP_KB": "16"}):
        env = dict(os.environ, LCD_WATCHDOG_S="5", **extra)
        out = str(tmp_path / f"r{len(outs)}.json")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "replay_chain.py"), fx, "--ont", "--json", out, "--oracle"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
        outs.append(json.load(open(out)))
    assert outs[0]["status"] == 0 and outs[0]["rows_degap_to_reads"]
    assert outs[0]["equals_oracle"] and outs[1]["equals_oracle"]   # (against the oracle run with the dump's own anchors, not only layout against layout)
    assert outs[1] == outs[0]
