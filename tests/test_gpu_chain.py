"""SURVEY 8d config 1, the device components chained on the reference's bundled real chunk -- BAM-record level in, variants out, nothing from
the fixture's pre-computed regions:
    EQX CIGARs --lcd_digar_batch--> digars + per-read noisy windows        (f2, per read)
    chr11 slice --lcd_sdust--> low-complexity intervals                     (chunk->low_comp_cr)
    both --lcd_pre_process_noisy_regs--> merged, read-supported regions     (f2, chunk level)
    regions (+ 10 bp flanks) and the digars --RegionBatch.add_region_from_chunk--> K1..K4 with the K5 haplotypes
    strings --stage S6--> candidate variants + read x variant profile       (f1)
Every stage is compared with its oracle on the same inputs (the reference's own sdust / cgranges where they exist).  The glue between the
stages that is not built yet (post_process_noisy_regs, src/collect_var.c:640) is replaced by fixed flanks -- the point is that the pieces
compose and agree, not that the region set equals longcallD's."""
import numpy as np
import pytest

import testdata_common as tc
from conftest import same_result
from test_gpu_digar import _cigar_of
from test_gpu_vars import same_vars

pytestmark = pytest.mark.gpu


def test_bam_records_to_variants(lcd, oracle):
    if oracle.ref_cgranges() is None:
        pytest.skip("oracle/_ref/libcgranges_ref.so not built")
    ch = tc.Chunk()
    o = int(ch.z["ref_beg"]); ref = ch.z["ref"]
    reg_beg, reg_end = o, o + len(ref) - 1
    cigs = [_cigar_of(d) for d in ch.digars]; pos0 = [int(d[0][0]) - 1 for d in ch.digars]
    quals = [np.full(int(q), 40, np.uint8) for q in ch.qlen]            # (the fixture keeps qualities only inside its own regions)
    # 1. digars + windows
    dg = lcd.digar_batch(pos0, cigs, quals, reg_beg, reg_end, 135086622)
    for i in range(0, ch.n_reads, 7):
        e = oracle.collect_digar_from_eqx_cigar(pos0[i], cigs[i], quals[i], reg_beg, reg_end, 135086622)
        assert (e["digars"] == dg[i]["digars"]).all() and (e["noisy"] == dg[i]["noisy"]).all()
    kept = [i for i in range(ch.n_reads) if dg[i]["rc"] == 0]
    # 2. low-complexity intervals of the chunk reference (0-based half-open -> cr_add(start - 1 .. ) as src/bam_utils.c:1579)
    low = lcd.sdust(ref, 5, 20)
    assert (low == oracle.ref_sdust(ref, 5, 20)).all()
    low_cr = np.stack([o + low[:, 0] - 1, o + low[:, 1] - 1], 1)
    # 3. chunk-level regions
    chunk_noisy = np.concatenate([dg[i]["chunk_noisy"] for i in kept])
    rb = [dg[i]["beg"] for i in kept]; re_ = [dg[i]["end"] for i in kept]; ivs = [dg[i]["noisy"] for i in kept]
    regs = lcd.pre_process_noisy_regs(chunk_noisy, low_cr, rb, re_, ivs)
    assert (regs == oracle.ref_pre_process_noisy_regs(chunk_noisy, low_cr, rb, re_, ivs)).all() and len(regs) >= 10
    # 4. regions through the hot path with the K5 haplotypes of the bundled profile; 5. variants
    from longcalld_amd import jobs
    st = lcd.assign_hap_germline(ch.hap_problem(), jobs.GERMLINE_CLEAN)
    digars4 = [d["digars"][:, :4] for d in dg]
    views, keep = lcd.make_read_views(digars4, ch.bseq, ch.qual, ch.qlen, st["haps"], st["phase_sets"])
    opt = lcd.default_opt(); opt.collect_noisy_vars = 1
    b = lcd.RegionBatch(opt)
    used = []
    for s, e, _ in regs:
        beg, end = int(s) + 1 - 10, int(e) + 10
        if end - beg + 1 > 3000 or beg <= o or end >= reg_end:
            continue
        ids = np.array([i for i in kept if dg[i]["beg"] <= end and dg[i]["end"] >= beg], np.int32)
        if len(ids) < 5:
            continue
        b.add_region_from_chunk(views, beg, end, ids, ref[beg - o:end - o + 1], packed=True)   # (bases stay 4-bit packed; unpacked on the device by upload())
        used.append((beg, end, ids))
    assert len(used) >= 8
    b.upload(); b.run(); b.download()
    n_vars = n_res = 0
    for k, (beg, end, ids) in enumerate(used):
        seqs, qs, covers = [], [], []
        for i in ids:
            rbq, req, cv = oracle.read_region_slice(digars4[i], ch.qlen[i], beg, end, 10)
            seqs.append(ch.bases(i, rbq, req) if req >= rbq else np.zeros(0, np.uint8)); qs.append(ch.qual[i][rbq:req + 1].copy() if req >= rbq else np.zeros(0, np.uint8)); covers.append(cv)
        reg = dict(reg_len=end - beg + 1, read_ids=ids, seqs=seqs, quals=qs, covers=np.array(covers, np.int32), haps=st["haps"][ids], phase_sets=st["phase_sets"][ids],
                   ref=ref[beg - o:end - o + 1])
        exp = oracle.collect_noisy_reg_aln_strs(reg)
        got = b.result(k)
        same_result(exp, got)
        same_vars(oracle.make_vars_from_msa_cons_aln(exp, beg, ref, o), b.region_vars(k, beg, ref, o))
        n_res += got["n_cons"] > 0; n_vars += b.region_vars(k, beg, ref, o)["n_vars"]
    assert n_res >= len(used) // 2 and n_vars > 20
    b.close()
    del keep
