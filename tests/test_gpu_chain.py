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


def test_chunk_handle_keeps_digars_and_bases_on_the_device(lcd, oracle):
    """the same chain through a DEVICE-RESIDENT chunk (lcd_chunk_t): records up once, digars made and kept in HBM, (region, read) slices cut there, the batch's read
    bases unpacked there.  Per-read info, windows, slices and every region result equal the host path and the oracle -- and the copy counters show that no digar
    byte and no read base crossed PCIe after the chunk was created (the host path moves both)."""
    ch = tc.Chunk()
    o = int(ch.z["ref_beg"]); ref = ch.z["ref"]
    reg_beg, reg_end = o, o + len(ref) - 1
    cigs = [_cigar_of(d) for d in ch.digars]; pos0 = [int(d[0][0]) - 1 for d in ch.digars]
    quals = [np.full(int(q), 40, np.uint8) for q in ch.qlen]
    dg = lcd.digar_batch(pos0, cigs, quals, reg_beg, reg_end, 135086622)                   # host path, for comparison
    c0 = lcd.copy_counters()
    assert c0[0] > 0                                                                       # (it downloaded the digars)
    chunk = lcd.DeviceChunk(pos0, cigs, quals, ch.bseq, reg_beg, reg_end, 135086622)
    c1 = lcd.copy_counters()
    assert c1[0] == c0[0] and c1[1] == c0[1] and c1[2] - c0[2] == chunk.packed_bytes       # the records' packed bases went up once; no digar came down
    info = chunk.read_info(); ivs = chunk.intervals()
    for i in range(ch.n_reads):
        assert (info["status"][i], info["beg"][i], info["end"][i], info["n_cand"][i], info["n_digars"][i]) == (dg[i]["rc"], dg[i]["beg"], dg[i]["end"], dg[i]["n_cand"], len(dg[i]["digars"]))
        assert (ivs[i][0] == dg[i]["noisy"]).all() and (ivs[i][0][ivs[i][1]] == dg[i]["chunk_noisy"]).all()
    kept = [i for i in range(ch.n_reads) if info["status"][i] == 0]
    chunk_noisy = np.concatenate([ivs[i][0][ivs[i][1]] for i in kept])
    regs = lcd.pre_process_noisy_regs(chunk_noisy, np.zeros((0, 2), np.int64), [info["beg"][i] for i in kept], [info["end"][i] for i in kept], [ivs[i][0] for i in kept])
    from longcalld_amd import jobs
    st = lcd.assign_hap_germline(ch.hap_problem(), jobs.GERMLINE_CLEAN)
    used, pair_read, pair_beg, pair_end = [], [], [], []
    for s_, e_, _ in regs:
        beg, end = int(s_) + 1 - 10, int(e_) + 10
        if end - beg + 1 > 3000 or beg <= o or end >= reg_end:
            continue
        ids = np.array([i for i in kept if info["beg"][i] <= end and info["end"][i] >= beg], np.int32)
        if len(ids) < 5:
            continue
        used.append((beg, end, ids)); pair_read += list(ids); pair_beg += [beg] * len(ids); pair_end += [end] * len(ids)
    assert len(used) >= 8
    rb, re_, cv = chunk.region_slices(pair_read, pair_beg, pair_end, 10)
    digars4 = [d["digars"][:, :4] for d in dg]
    for k in range(0, len(pair_read), 5):                                                  # slices == the oracle's digar walk
        assert (rb[k], re_[k], cv[k]) == oracle.read_region_slice(digars4[pair_read[k]], ch.qlen[pair_read[k]], pair_beg[k], pair_end[k], 10)
    opt = lcd.default_opt()
    b = lcd.RegionBatch(opt); bh = lcd.RegionBatch(opt)
    views, keep = lcd.make_read_views(digars4, ch.bseq, ch.qual, ch.qlen, st["haps"], st["phase_sets"])
    at = 0
    for beg, end, ids in used:
        n = len(ids)
        chunk.add_region(b, beg, end, ids, rb[at:at + n], re_[at:at + n], cv[at:at + n], st["haps"][ids], st["phase_sets"][ids], ref[beg - o:end - o + 1])
        bh.add_region_from_chunk(views, beg, end, ids, ref[beg - o:end - o + 1], packed=True)
        at += n
    c2 = lcd.copy_counters()
    b.upload(); b.run(); b.download()
    c3 = lcd.copy_counters()
    assert c3[0] == c2[0] and c3[1] == c2[1] and c3[2] == c2[2] and c3[3] == c2[3]         # the device-resident path: nothing of the reads crossed PCIe
    bh.upload(); bh.run(); bh.download()
    assert lcd.copy_counters()[2] > c3[2]                                                  # (the host path uploads the slices' bases)
    assert b.digest() == bh.digest()
    n_res = 0
    for k in range(len(used)):
        g = b.result(k)
        same_result(bh.result(k), g)
        n_res += g["n_cons"] > 0
    assert n_res >= len(used) // 2
    b.close(); bh.close(); chunk.close()
    del keep
