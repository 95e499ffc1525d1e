"""SURVEY 8d config 1 on the GPU: the reference's bundled real chunk (tests/golden/testdata_chunk.npz) through the HIP path --
K5 on the real read x variant profile, then every noisy region through the chunk-view entry (digar walk + 4-bit unpacking in the library)
with the haplotypes K5 produced -- bit-identical to the oracle and to the committed expected digests."""
import numpy as np
import pytest

import testdata_common as tc
from conftest import same_result

pytestmark = pytest.mark.gpu

KEYS = ("haps", "phase_sets", "n_clean_agree_snps", "n_clean_conflict_snps", "var_phase_set", "hap_to_cons_alle")


def test_k5_real_profile(lcd):
    from longcalld_amd import jobs
    ch = tc.Chunk()
    got = lcd.assign_hap_germline(ch.hap_problem(), jobs.GERMLINE_CLEAN)
    for k in KEYS:
        assert (got[k] == ch.z["exp_" + k]).all(), k


def _run_chunk(lcd, ch, haps, pss, packed=False, digest=False):
    views, keep = lcd.make_read_views(ch.digars, ch.bseq, ch.qual, ch.qlen, haps, pss)
    b = lcd.RegionBatch()
    for k, (beg, end) in enumerate(ch.regions):
        b.add_region_from_chunk(views, beg, end, ch.reg_reads(k), ch.ref_slice(k), packed=packed)
    b.upload(); b.run(); b.download()
    out = [b.result(k) for k in range(len(ch.regions))]
    ids = [b.sorted_ids(k) for k in range(len(ch.regions))]
    dg = b.digest()
    b.close()
    del keep
    return (out, ids, dg) if digest else (out, ids)


def test_packed_read_slices_equal_unpacked(lcd):
    """SURVEY f2, the step into the region jobs: the reads' slices handed over 4-bit packed as they are in the BAM records and unpacked on the device
    (lcd_batch_add_region_from_chunk_packed, digar_kernel.hip lcd_unpack_kernel) give the results of the host's per-base loop -- every region of the bundled
    real chunk, slices starting at even and odd read positions, same digest and same strings"""
    from longcalld_amd import jobs
    ch = tc.Chunk()
    st = lcd.assign_hap_germline(ch.hap_problem(), jobs.GERMLINE_CLEAN)
    a, ida, da = _run_chunk(lcd, ch, st["haps"], st["phase_sets"], packed=False, digest=True)
    p, idp, dp = _run_chunk(lcd, ch, st["haps"], st["phase_sets"], packed=True, digest=True)
    assert da == dp and len(a) == len(p) > 0
    n_res = 0
    for x, y, i1, i2 in zip(a, p, ida, idp):
        assert (i1 == i2).all() and x["n_cons"] == y["n_cons"]
        for c in range(x["n_cons"]):
            for sx, sy in zip(x["aln_strs"][c], y["aln_strs"][c]):
                assert (sx is None) == (sy is None)
                if sx is not None:
                    assert (sx["target"] == sy["target"]).all() and (sx["query"] == sy["query"]).all()
        n_res += x["n_cons"] > 0
    assert n_res > 0


def test_real_regions_with_k5_haplotypes(lcd, oracle):
    from longcalld_amd import jobs
    ch = tc.Chunk()
    st = lcd.assign_hap_germline(ch.hap_problem(), jobs.GERMLINE_CLEAN)     # K5 on the GPU feeds K1, as in the reference's pass loop
    got, ids = _run_chunk(lcd, ch, st["haps"], st["phase_sets"])
    for k, g in enumerate(got):
        exp = oracle.collect_noisy_reg_aln_strs(ch.region_dict(oracle, k, st["haps"], st["phase_sets"]))
        assert (ids[k] == exp["sorted_ids"]).all()
        same_result(exp, g)
        assert tc.result_digest(g) == int(ch.z["exp_region_digest"][k]), k


def test_real_regions_unphased(lcd, oracle):
    """the same regions before any read is phased (first pass on a chunk without clean het variants): K2 de-novo MSA + 2 clusters"""
    ch = tc.Chunk()
    haps = np.zeros(ch.n_reads, np.int32); pss = np.full(ch.n_reads, -1, np.int64)
    got, ids = _run_chunk(lcd, ch, haps, pss)
    n_res = 0
    for k, g in enumerate(got):
        exp = oracle.collect_noisy_reg_aln_strs(ch.region_dict(oracle, k, haps, pss))
        assert (ids[k] == exp["sorted_ids"]).all()
        same_result(exp, g)
        n_res += g["n_cons"] > 0
    assert n_res >= len(ch.regions) // 2


def test_digar_rewrite_on_real_regions(lcd, oracle):
    """SURVEY a13 end to end on the bundled chunk (--refine-aln -b path): regions run with collect_ref_read_aln_str, then every read's digar list is rebuilt
    from its ref<->read string (lcd_update_digars_from_msa1 with the slices the library reports) == the oracle's update_digars_from_msa1 on the same string"""
    import ctypes as C
    from longcalld_amd import _lib, jobs
    from test_digar_rewrite import _call
    ch = tc.Chunk()
    st = lcd.assign_hap_germline(ch.hap_problem(), jobs.GERMLINE_CLEAN)
    views, keep = lcd.make_read_views(ch.digars, ch.bseq, ch.qual, ch.qlen, st["haps"], st["phase_sets"])
    o = lcd.default_opt(); o.collect_ref_read_aln_str = 1
    b = lcd.RegionBatch(o)
    for k, (beg, end) in enumerate(ch.regions):
        b.add_region_from_chunk(views, beg, end, ch.reg_reads(k), ch.ref_slice(k))
    b.upload(); b.run(); b.download()
    prod, orc = C.CDLL(_lib.LIB_PATH), oracle.lib()
    n_new = n_rej = 0
    for k, (beg, end) in enumerate(ch.regions):
        res, sl = b.result(k), b.read_slices(k)
        by_id = {int(r): i for i, r in enumerate(sl["read_ids"])}
        for c in range(res["n_cons"]):
            for j, rid in enumerate(res["clu_read_ids"][c]):
                s = res["aln_strs"][c][2 * j + 2]
                if s is None:
                    continue
                i = by_id[int(rid)]
                digs = [(int(d[0]), int(d[1]), int(d[2]), int(d[3]), 0) for d in ch.digars[int(rid)]]
                args = (digs, int(ch.qlen[int(rid)]), s["target"], s["query"], int(sl["covers"][i]), int(beg), int(end), int(sl["read_beg"][i]), int(sl["read_end"][i]))
                a, e = _call(prod, "lcd_update_digars_from_msa1", *args), _call(orc, "lcdo_update_digars_from_msa1", *args)
                assert a == e, (k, c, j)
                n_new += a[0] == 0; n_rej += a[0] == 1
    b.close()
    del keep
    assert n_new > 200
