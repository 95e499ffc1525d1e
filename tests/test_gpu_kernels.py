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
