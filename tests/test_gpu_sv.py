"""BASELINE configs[4]: 60x noisy reads with 1-10 kb insertions / deletions.  SV regions are where the reference's cost lives
(src/align.c:374-460 at scores ~ 24 + gap length; the read-sampling path :719-728 for regions >= 10 kb): every string of every region == oracle
at the sizes the scalar oracle finishes in seconds, size-independent properties above that."""
import numpy as np
import pytest

from conftest import check_invariants, same_result

pytestmark = pytest.mark.gpu


def _opt(lcd):
    o = lcd.default_opt(); o.is_ont = 1
    return o


def _run(lcd, regs, opt):
    b = lcd.RegionBatch(opt)
    for r in regs:
        b.add_region(r)
    b.upload(); b.run(); b.download()
    out = [b.result(i) for i in range(len(regs))]
    ids = [b.sorted_ids(i) for i in range(len(regs))]
    st, dg = b.stats(), b.digest()
    b.close()
    return out, ids, st, dg


def test_sv_regions_match_oracle(lcd, oracle):
    """insertions and deletions of 1-3 kb in 60x noisy reads, phased and unphased, a third of the reads partial: region by region == oracle"""
    from longcalld_amd import jobs
    rng = np.random.default_rng(4040)
    regs = [jobs.make_sv_region(rng, kind="ins", sv_len=1000, ctx=400), jobs.make_sv_region(rng, kind="del", sv_len=1500, ctx=600),
            jobs.make_sv_region(rng, kind="ins", sv_len=3000, ctx=900, n_reads=40), jobs.make_sv_region(rng, kind="del", sv_len=3000, ctx=800, n_reads=40),
            jobs.make_sv_region(rng, kind="ins", sv_len=1200, ctx=300, phased=False, n_reads=24), jobs.make_sv_region(rng, kind="del", sv_len=1100, ctx=500, phased=False, n_reads=24)]
    got, ids, st, _ = _run(lcd, regs, _opt(lcd))
    n_res = 0
    for r, g, sid in zip(regs, got, ids):
        exp = oracle.collect_noisy_reg_aln_strs(r)
        assert (sid == exp["sorted_ids"]).all()
        same_result(exp, g)
        n_res += g["n_cons"] > 0
    assert n_res >= 4 and st["n_wfa_jobs"] > 0 and st["n_edlib_jobs"] > 0


def test_sv_sampling_region_matches_oracle(lcd, oracle):
    """a 9.6 kb deletion in a 10.4 kb region (>= min_noisy_reg_size_to_sample_reads): reads ordered by error rate (src/align.c:957-968), every
    full read filtered with edlib first (:719-728), partial reads anchored with K4 + K3 at scores in the thousands, ref<->cons gap ~ 9 600 -- == oracle"""
    from longcalld_amd import jobs
    rng = np.random.default_rng(4141)
    reg = jobs.make_sv_region(rng, kind="del", sv_len=9600, ctx=800, n_reads=30, phased=True)
    assert reg["reg_len"] >= 10000
    got, ids, st, _ = _run(lcd, [reg], _opt(lcd))
    exp = oracle.collect_noisy_reg_aln_strs(reg)
    assert (ids[0] == exp["sorted_ids"]).all()
    same_result(exp, got[0])
    assert got[0]["n_cons"] == 2
    assert check_invariants(reg, got[0]) > 0


def test_sv_batch_equals_oracle(lcd, monkeypatch):
    """the configs[4]-shaped batch below (125 regions of a 1 Mb slice: 10 SV regions of 1 - 10 kb among 60x noisy-read regions), EVERY region ==
    oracle -- read order, clusters, every alignment string -- and the digest the same with the 16-bit LDS ring forced / off (item 2c: the SV shape is where int16
    values of a sub-graph alignment leave their range).  The oracle runs on all host cores (conftest.oracle_many)."""
    from conftest import oracle_many
    from longcalld_amd import jobs
    regs = jobs.make_regions(777, 125, jobs.SV)
    exp = oracle_many(777, 125, "sv")
    got, ids, st, dg = _run(lcd, regs, _opt(lcd))
    for k, (e, g, sid) in enumerate(zip(exp, got, ids)):
        assert (sid == e["sorted_ids"]).all(), k
        same_result(e, g)
    assert st["n_regions_resolved"] == sum(e["n_cons"] > 0 for e in exp) and sum("sv" in r for r in regs) == 10
    for r16 in ("2", "0"):
        monkeypatch.setenv("LCD_RING16", r16)
        assert _run(lcd, regs, _opt(lcd))[3] == dg


def test_sv_batch_properties_and_digest(lcd):
    """a configs[4]-shaped batch (125 regions of a 1 Mb slice: 10 SV regions up to 10 kb among 60x noisy-read regions): invariants on every region
    (clusters partition the reads, every row de-gaps to its read / consensus / reference), the SV consensus carries the SV, and the digest does not
    depend on the submission form"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(777, 125, jobs.SV)
    o = _opt(lcd)
    bs = []
    for _ in range(2):
        b = lcd.RegionBatch(o)
        for r in regs:
            b.add_region(r)
        b.upload(); bs.append(b)
    bs[0].run(); bs[0].download()
    d0 = bs[0].digest()
    n_str = n_sv = 0
    for k, r in enumerate(regs):
        res = bs[0].result(k)
        n_str += check_invariants(r, res)
        if "sv" in r and res["n_cons"] == 2:
            kind, sv_len = r["sv"]
            lens = sorted(int((res["aln_strs"][c][0]["query"] != 5).sum()) for c in range(2))
            # one consensus ~ the reference window, the other longer / shorter by the SV
            assert abs((lens[1] - lens[0]) - sv_len) < 0.03 * sv_len + 60, (k, r["sv"], lens)
            n_sv += 1
    assert n_str > 3000 and n_sv >= 5
    lcd.RegionBatch.run_many(bs)
    for b in bs:
        b.download()
    assert bs[0].digest() == d0 and bs[1].digest() == d0
    for b in reversed(bs):
        b.close()
