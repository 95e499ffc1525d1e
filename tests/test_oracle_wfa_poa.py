"""Score-level pins and invariants of the K3 / K1 / K2 oracles (upstream WFA2-lib and abPOA are absent: parity unpinned at byte level)."""
import os

import numpy as np

from conftest import mutate


def test_wfa_score_equals_gotoh_and_cigar_is_valid(oracle):
    rng = np.random.default_rng(2)
    for L in [0, 1, 2, 9, 60, 250, 700]:
        for rate, sv in [(0, 0), (0.02, 0), (0.1, 0.004), (0.3, 0)]:
            t = rng.integers(0, 4, L).astype(np.uint8)
            p = mutate(rng, t, rate, sv)
            for ga in (1, 2):
                for a, b in ((p, t), (t, p)):
                    r = oracle.wfa_end2end_aln(a, b, gap_aln=ga)
                    assert r["score"] == oracle.gotoh2p_score(a, b)                       # optimal under the independent O(nm) DP
                    assert oracle.cigar_score2p(r["cigar"], a, b) == r["score"]           # CIGAR consumes both strings and re-scores to it
                    pa, ta = r["pattern_alg"], r["text_alg"]
                    assert (pa[pa != 5] == a).all() and (ta[ta != 5] == b).all()


def test_wfa_left_alignment_places_gap_leftmost(oracle):
    # deleting one base of a homopolymer run: LEFT_ALN (src/align.c:409-453, reversal trick) must report the first position
    t = np.array([0, 1, 2, 2, 2, 2, 3, 0, 1], np.uint8)
    p = np.array([0, 1, 2, 2, 2, 3, 0, 1], np.uint8)
    left = oracle.wfa_end2end_aln(t, p, gap_aln=1)
    right = oracle.wfa_end2end_aln(t, p, gap_aln=2)
    assert int(np.where(left["text_alg"] == 5)[0][0]) == 2 and int(np.where(right["text_alg"] == 5)[0][0]) == 5


def test_poa_invariants_and_band_score(oracle):
    os.environ["LCDO_POA_CHECK_UNBANDED"] = "1"
    rng = np.random.default_rng(3)
    for L, rate in [(40, 0.001), (150, 0.02), (500, 0.001), (400, 0.08)]:
        truth = rng.integers(0, 4, L).astype(np.uint8)
        reads = [mutate(rng, truth, rate) for _ in range(8)]
        r = oracle.poa_partial_aln_msa_cons(reads, [12] * 8)
        assert r["n_cons"] == 1
        for read, row in zip(reads, r["msa"]):
            assert (row[row != 5] == read).all() and len(row) == r["msa_len"]
        crow = r["msa"][8]
        assert (crow[crow != 5] == r["cons"][0]).all()
        if rate <= 0.001:
            assert abs(len(r["cons"][0]) - L) <= 1
        hap2 = truth.copy()
        for p in range(10, L - 10, max(20, L // 6)):
            hap2[p] = (hap2[p] + 1) % 4
        reads = [mutate(rng, truth if i % 2 == 0 else hap2, rate) for i in range(12)]
        r = oracle.poa_aln_msa_cons(reads, 2)
        assert sorted(np.concatenate(r["clu"]).tolist()) == list(range(12))               # clusters partition the reads
        if rate <= 0.02:
            assert r["n_cons"] == 2 and len({int(i) % 2 for i in r["clu"][0]}) == 1      # and separate the two haplotypes


def test_glue_sort_phase_set_trim(oracle):
    """src/align.c:955 (exchange sort), :1225 (phase-set choice), region driver on a synthetic region"""
    from longcalld_amd import jobs
    rng = np.random.default_rng(4)
    reg = jobs.make_region(rng, jobs.HIFI, length=260, n_reads=16)
    res = oracle.collect_noisy_reg_aln_strs(reg)
    ids = list(res["sorted_ids"])
    covers = {int(i): int(c) for i, c in zip(reg["read_ids"], reg["covers"])}
    lens = {int(i): len(s) for i, s in zip(reg["read_ids"], reg["seqs"])}
    seen_partial = False
    for a, b in zip(ids, ids[1:]):
        if covers[a] != 12:
            seen_partial = True
        assert not (seen_partial and covers[b] == 12)                                    # both-cover reads first
        if covers[a] == covers[b]:
            assert lens[a] >= lens[b]                                                    # then longer first
    assert res["n_cons"] in (0, 1, 2)
    for c in range(res["n_cons"]):
        s0 = res["aln_strs"][c][0]
        assert (s0["target"][s0["target"] != 5] == reg["ref"]).all()                     # ref row of ref<->cons de-gaps to the reference
        n = res["clu_n_seqs"][c]
        for k in range(n):
            s = res["aln_strs"][c][2 * k + 1]
            assert s is not None and len(s["target"]) == s["aln_len"]
            assert not ((s["target"] == 5) & (s["query"] == 5)).any()                    # gap/gap columns dropped (src/align.c:1034-1040)


def test_band_certificate(oracle, monkeypatch):
    """the bound behind the product's certified band for K2 (poa_kernel.hip align_certified), checked inside the oracle's unbanded DP (oracle/poa.c,
    LCDO_CERT_STATS): no cell's H exceeds the prefix half of the bound, every matched cell of the backtrack lies inside its row's interval -- on clean reads
    with het insertions, on noisy reads, on reads of different lengths -- and the intervals are a small fraction of the full rows for clean reads"""
    monkeypatch.setenv("LCDO_CERT_STATS", "1")
    rng = np.random.default_rng(77)
    before = oracle.poa_cert_stats()
    for L, rate, sv in ((400, 0.002, 15), (800, 0.001, 60), (600, 0.05, 20), (700, 0.01, 250)):
        h1 = rng.integers(0, 4, L).astype(np.uint8)
        h2 = np.concatenate([h1[: L // 3], rng.integers(0, 4, sv).astype(np.uint8), h1[L // 3:]])
        reads = [mutate(rng, h1 if i % 2 == 0 else h2, rate) for i in range(10)]
        reads[3] = reads[3][: len(reads[3]) // 2]
        oracle.poa_aln_msa_cons(reads, 2)
    st = oracle.poa_cert_stats()
    d = {k: st[k] - before[k] for k in st if k != "widest"}
    assert d["reads"] >= 36 and d["prefix_violations"] == 0 and d["path_violations"] == 0
    assert d["hull_cells"] * 3 < d["full_cells"]
