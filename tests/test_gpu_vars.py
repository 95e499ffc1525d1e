"""SURVEY 8(f) f1 on the GPU: candidate variants + read x variant profile of a region (stage S6, vars_kernel.hip through
lcd_batch_region_vars) vs the oracle's restatement of make_vars_from_msa_cons_aln (src/collect_var.c:2279) run on the oracle's strings --
every field bit-identical, on seeded synthetic regions (HiFi / ONT shapes, K1 and K2 branches) and on the reference's bundled real chunk."""
import numpy as np
import pytest

import testdata_common as tc

pytestmark = pytest.mark.gpu

KEYS = ("pos", "var_type", "ref_len", "alt_len", "cate", "from_cons", "is_homopolymer_indel", "ref_base", "alt_ref_base", "total_cov")


def same_vars(exp, got, ids=None):
    assert exp["n_vars"] == got["n_vars"]
    for k in KEYS:
        assert (exp[k] == got[k]).all(), (k, exp[k], got[k])
    assert (exp["alle_covs"] == got["alle_covs"]).all()
    for a, b in zip(exp["alt_seqs"], got["alt_seqs"]):
        assert a.tobytes() == b.tobytes()
    assert exp["n_rows"] == got["n_rows"]
    assert (exp["prof_start"] == got["prof_start"]).all() and (exp["prof_end"] == got["prof_end"]).all()
    assert (exp["prof_alleles"] == got["prof_alleles"]).all()
    if ids is not None:
        assert (np.concatenate(ids) == got["row_read_ids"]).all()


def _opt(lcd, mode):
    o = lcd.default_opt()
    o.collect_noisy_vars = mode
    return o


@pytest.mark.parametrize("shape,seed,n", [("hifi", 11, 40), ("ont", 12, 12)])
def test_vars_match_oracle_synthetic(lcd, oracle, shape, seed, n):
    from longcalld_amd import jobs
    regs = jobs.make_regions(seed, n, jobs.HIFI if shape == "hifi" else jobs.ONT)
    rng = np.random.default_rng(seed)
    b = lcd.RegionBatch(_opt(lcd, 1))
    for r in regs:
        b.add_region(r)
    b.upload(); b.run(); b.download()
    n_vars = n_hom = 0
    for k, r in enumerate(regs):
        beg = 100000 + 5000 * k
        cref = np.concatenate([r["ref"], rng.integers(0, 4, 8).astype(np.uint8)])   # chunk reference continues past the region
        res = oracle.collect_noisy_reg_aln_strs(r)
        exp = oracle.make_vars_from_msa_cons_aln(res, beg, cref, beg)
        got = b.region_vars(k, beg, cref, beg)
        ids = [res["clu_read_ids"][c] for c in range(res["n_cons"])] if res["n_cons"] else None
        same_vars(exp, got, ids)
        n_vars += got["n_vars"]; n_hom += int((got["cate"] == 0x200).sum())
    assert n_vars > n
    assert b.stats()["ms_vars"] > 0
    b.close()


def test_vars_only_download(lcd, oracle):
    """opt.collect_noisy_vars == 2: the strings stay in HBM, the variants are the same, lcd_batch_region_result refuses"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(21, 12, jobs.HIFI)
    b = lcd.RegionBatch(_opt(lcd, 2))
    for r in regs:
        b.add_region(r)
    b.upload(); b.run(); b.download()
    for k, r in enumerate(regs):
        res = oracle.collect_noisy_reg_aln_strs(r)
        same_vars(oracle.make_vars_from_msa_cons_aln(res, 1000, r["ref"], 1000), b.region_vars(k, 1000, r["ref"], 1000))
    with pytest.raises(Exception):
        b.result(0)
    b.close()


def test_vars_on_real_chunk(lcd, oracle):
    """K5 haplotypes -> K1 -> strings -> variants, on the reference's bundled HG002 reads"""
    ch = tc.Chunk()
    haps, pss = ch.z["exp_haps"], ch.z["exp_phase_sets"]
    views, keep = lcd.make_read_views(ch.digars, ch.bseq, ch.qual, ch.qlen, haps, pss)
    b = lcd.RegionBatch(_opt(lcd, 1))
    for k, (beg, end) in enumerate(ch.regions):
        b.add_region_from_chunk(views, beg, end, ch.reg_reads(k), ch.ref_slice(k))
    b.upload(); b.run(); b.download()
    o = int(ch.z["ref_beg"])
    tot = het = 0
    for k, (beg, end) in enumerate(ch.regions):
        res = oracle.collect_noisy_reg_aln_strs(ch.region_dict(oracle, k, haps, pss))
        exp = oracle.make_vars_from_msa_cons_aln(res, beg, ch.z["ref"], o)
        got = b.region_vars(k, beg, ch.z["ref"], o)
        same_vars(exp, got, [res["clu_read_ids"][c] for c in range(res["n_cons"])])
        tot += got["n_vars"]; het += int((got["cate"] == 0x100).sum())
    assert tot == int(ch.z["exp_n_noisy_vars"]) and het > 0
    b.close()
    del keep


def test_vars_joint_submission(lcd, oracle):
    """lcd_batch_run_many with stage S6: two batches in one set of launches, each gets its own variants"""
    from longcalld_amd import jobs
    sets = [jobs.make_regions(31, 10, jobs.HIFI), jobs.make_regions(32, 7, jobs.HIFI)]
    bs = []
    for regs in sets:
        b = lcd.RegionBatch(_opt(lcd, 1))
        for r in regs:
            b.add_region(r)
        b.upload(); bs.append(b)
    lcd.RegionBatch.run_many(bs)
    for b in bs:
        b.download()
    for b, regs in zip(bs, sets):
        for k, r in enumerate(regs):
            res = oracle.collect_noisy_reg_aln_strs(r)
            same_vars(oracle.make_vars_from_msa_cons_aln(res, 500, r["ref"], 500), b.region_vars(k, 500, r["ref"], 500))
    for b in reversed(bs):
        b.close()
