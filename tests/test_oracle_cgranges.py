"""Pins the two cgranges behaviours the hot path depends on against the REFERENCE's own src/cgranges.c (compiled from where it lies into
oracle/_ref/libcgranges_ref.so): the order cr_index() leaves the intervals in (restated as lcdo_cr_sorted_order: K5 seeds reads in it, a
read's noisy windows are reported in it) and the order cr_overlap() reports hits in (the oracle's K5 assumes: index order of the sorted array)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(oracle):
    if oracle.ref_cgranges() is None:
        pytest.skip("oracle/_ref/libcgranges_ref.so not built (reference tree absent and no prebuilt copy)")
    return oracle


def test_sorted_order_matches_reference_cgranges(ref):
    rng = np.random.default_rng(11)
    for n in (1, 2, 5, 40, 64, 65, 200, 3000):
        for ties in (False, True):
            st = rng.integers(0, 50 if ties else 1_000_000, n).astype(np.int32)
            en = st + rng.integers(1, 500, n).astype(np.int32)
            assert (ref.cr_sorted_order(st, en) == ref.ref_cr_sorted_order(st, en)).all(), (n, ties)
    # cr_add clamps a negative start to 0 and drops st > en (src/cgranges.c:146-149) -- oracle/digar.c and the digar kernel do the same
    assert ref.ref_cr_sorted_order(np.array([500, -100, 20], np.int32), np.array([510, 30, 25], np.int32)).tolist() == [1, 2, 0]
    assert ref.ref_cr_overlap(np.array([-100], np.int32), np.array([-90], np.int32), -200, 10).tolist() == []
    st = np.sort(rng.integers(0, 10000, 300)).astype(np.int32)   # already sorted: kept as added, ties included
    assert (ref.ref_cr_sorted_order(st, st + 3) == np.arange(300)).all() and (ref.cr_sorted_order(st, st + 3) == np.arange(300)).all()


def test_overlap_reports_in_sorted_index_order(ref):
    rng = np.random.default_rng(12)
    for n in (3, 50, 400):
        st = rng.integers(0, 5000, n).astype(np.int32); en = st + rng.integers(1, 300, n).astype(np.int32)
        order = ref.ref_cr_sorted_order(st, en)
        for _ in range(40):
            q0 = int(rng.integers(0, 5000)); q1 = q0 + int(rng.integers(1, 400))
            hits = ref.ref_cr_overlap(st, en, q0, q1)
            exp = [i for i in order if st[i] < q1 and q0 < en[i]]
            assert hits.tolist() == exp


def test_post_process_noisy_regs_host_glue(ref):
    """post_process_noisy_regs (src/collect_var.c:640): the library's host glue (no device work in it, as in the reference) vs the same control
    flow over the reference's own cgranges -- flank growth along candidate variants, non-candidate categories skipped, touching regions merged"""
    from longcalld_amd import align as lcd
    rng = np.random.default_rng(31)
    for trial in range(40):
        n = int(rng.integers(1, 40))
        st = np.sort(rng.integers(1000, 100000, n)); regs = np.stack([st, st + rng.integers(5, 400, n), rng.integers(6, 300, n)], 1)
        nv = int(rng.integers(0, 200))
        vp = np.sort(rng.integers(900, 101000, nv)); vl = rng.choice([0, 1, 1, 1, 12, 40], nv); vc = rng.choice([0x004, 0x008, 0x080, 0x800, 0x001, 0x002, 0x100], nv)
        for flank in (10, 0, 50):
            exp = ref.ref_post_process_noisy_regs(regs, vp, vl, vc, flank)
            got = lcd.post_process_noisy_regs(regs, vp, vl, vc, flank)
            assert exp.shape == got.shape and (exp == got).all(), (trial, flank)
    assert len(lcd.post_process_noisy_regs(np.zeros((0, 3), np.int64), [], [], [])) == 0


def _golden():
    import json, os
    from conftest import ROOT
    return json.load(open(os.path.join(ROOT, "tests", "golden", "cgranges_golden.json")))


def test_committed_vectors_of_the_reference_cgranges(oracle):
    """tests/golden/cgranges_golden.json (made by tests/golden/make_cgranges_golden.py from the reference's own src/cgranges.c): the pins above, independent of
    oracle/_ref being built -- cr_index order against the oracle's restatement, cr_overlap's hit order against the rule the oracle's K5 assumes, cr_merge with both
    parameter sets of src/collect_var.c and whole pre_ / post_process_noisy_regs cases against the library's host code (touching, nested and tied intervals, negative
    starts, chains that need several passes, label-dependent distances)."""
    from longcalld_amd import align as lcd
    g = _golden()
    for c in g["order"]:
        st, en = np.array(c["st"], np.int32), np.array(c["en"], np.int32)
        if (st < 0).any() or (st > en).any():
            continue        # (cr_add clamps / drops those: the restatement is called on what cr_add keeps -- covered by the merge cases below)
        assert oracle.cr_sorted_order(st, en).tolist() == c["order"]
    for c in g["overlap"]:
        st, en = np.array(c["st"], np.int32), np.array(c["en"], np.int32)
        order = oracle.cr_sorted_order(st, en)
        q0, q1 = c["q"]
        assert [int(i) for i in order if st[i] < q1 and q0 < en[i]] == c["hits"]
    n_dyn = 0
    for c in g["merge"]:
        fixed, dyn, lmin = c["args"]
        ivs = np.stack([c["st"], c["en"], c["label"]], 1)
        got = lcd.cr_merge(ivs, fixed)
        assert got.tolist() == c["merged"], c["args"]
        n_dyn += fixed < 0
    assert n_dyn >= 30
    for c in g["post_process"]:
        got = lcd.post_process_noisy_regs(np.array(c["regs"], np.int64).reshape(-1, 3), c["var_pos"], c["var_ref_len"], c["var_cate"], c["flank"])
        assert got.tolist() == c["out"]
    assert len(g["pre_process"]) >= 20   # (lcd_pre_process_noisy_regs starts the device: those cases run in tests/test_gpu_digar.py)
