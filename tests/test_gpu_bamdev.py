"""SURVEY 8f f3 on the device, end to end: an indexed BAM file -> (compressed BGZF blocks up, inflated by lcd_inflate_kernel, records found / measured / filtered by
bam_kernel.hip, digars made in HBM) -> lcd_chunk_t -> region jobs.  Checked against the host loader (lcd_bam_load_region_indexed: zlib + the record loop of
collect_ref_seq_bam_main, src/bam_utils.c:1672-1706) feeding the host-array chunk, read by read and region by region, on a seeded BAM with every record kind the
filters see, on the bundled real HiFi chunk written out as a BAM, and on the CG-tag / malformed-record cases of tests/test_io.py."""
import ctypes as C
import struct

import numpy as np
import pytest

import testdata_common as tc
from test_gpu_digar import _cigar_of
from test_io import BamReads, _bgzf, _make_bam, _one_record_bam, _write_bai

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def io():
    from longcalld_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    L.lcd_bam_load_region_indexed.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(BamReads)]
    L.lcd_io_last_error.restype = C.c_char_p
    return L


def _host_reads(io, bam, chrom, beg, end, mq):
    r = BamReads()
    n = io.lcd_bam_load_region_indexed(bam.encode(), (bam + ".bai").encode(), chrom.encode(), beg, end, mq, C.byref(r))
    assert n >= 0, io.lcd_io_last_error()
    out = dict(n=n, pos0=[r.pos0[i] for i in range(n)], end_pos=[r.end_pos[i] for i in range(n)], mapq=[r.mapq[i] for i in range(n)], flag=[r.flag[i] for i in range(n)],
               qlen=[r.qlen[i] for i in range(n)], names=[C.string_at(C.addressof(r.name_pool.contents) + r.name_off[i]).decode() for i in range(n)],
               cig=[np.array([r.cigar_pool[r.cigar_off[i] + k] for k in range(r.n_cigar[i])], np.uint32) for i in range(n)],
               seq=[np.ctypeslib.as_array(r.seq_pool, shape=(int(r.seq_off[i]) + (r.qlen[i] + 1) // 2,))[int(r.seq_off[i]):].copy() for i in range(n)],
               qual=[np.ctypeslib.as_array(r.qual_pool, shape=(int(r.qual_off[i]) + r.qlen[i],))[int(r.qual_off[i]):].copy() for i in range(n)])
    io.lcd_bam_reads_free(C.byref(r))
    return out


def _same_chunk(lcd, dev, host_reads, reg_beg, reg_end, tlen):
    h = host_reads
    m = dev.meta
    assert dev.n == h["n"]
    for k in ("pos0", "end_pos", "mapq", "flag", "qlen"):
        assert list(m[k]) == h[k], k
    assert list(m["n_cigar"]) == [len(c) for c in h["cig"]] and m["names"] == h["names"]
    if h["n"] == 0:
        return None
    ref = lcd.DeviceChunk(h["pos0"], h["cig"], h["qual"], h["seq"], reg_beg, reg_end, tlen)
    a, b = dev.read_info(), ref.read_info()
    for k in a:
        assert (a[k] == b[k]).all(), k
    ia, ib = dev.intervals(), ref.intervals()
    for x, y in zip(ia, ib):
        assert (x[0] == y[0]).all() and (x[1] == y[1]).all()
    return ref


def test_seeded_bam_every_region_equals_the_host_loader(lcd, io, tmp_path):
    """records of two references, unmapped / secondary / supplementary flags, MAPQ cuts, soft clips, X / I / D operations, random BGZF block size: per-read scalars,
    names, digar statistics and noisy windows through the device path == the host loader + host-array chunk; (region, read) slices too"""
    rng = np.random.default_rng(21)
    path = str(tmp_path / "t.bam")
    _make_bam(rng, path, n=400)
    n_checked = 0
    for beg, end, mq in [(1, 2000000, 0), (1, 2000000, 30), (300000, 420000, 30), (300000, 420000, 0), (1100000, 1300000, 10), (5000, 5001, 0), (1500000, 1600000, 0), (1, 900, 0)]:
        h = _host_reads(io, path, "chr11", beg, end, mq)
        dev = lcd.DeviceChunk.from_bam(path, path + ".bai", "chr11", beg, end, min_mapq=mq)
        ref = _same_chunk(lcd, dev, h, beg, end, 2000000)
        if ref is not None:
            info = dev.read_info()
            pr, pb, pe = [], [], []
            for i in range(dev.n):
                if info["status"][i] == 0 and info["end"][i] - info["beg"][i] > 400:
                    mid = int(info["beg"][i] + info["end"][i]) // 2
                    pr.append(i); pb.append(mid - 150); pe.append(mid + 150)
            if pr:
                sa, sb = dev.region_slices(pr, pb, pe, 10), ref.region_slices(pr, pb, pe, 10)
                assert all((x == y).all() for x, y in zip(sa, sb))
                n_checked += len(pr)
            ref.close()
        dev.close()
    assert n_checked > 100
    h = _host_reads(io, path, "chrA", 1, 50000, 0)
    dev = lcd.DeviceChunk.from_bam(path, path + ".bai", "chrA", 1, 50000, min_mapq=0)
    r = _same_chunk(lcd, dev, h, 1, 50000, 50000)
    if r is not None:
        r.close()
    dev.close()
    with pytest.raises(RuntimeError, match="contig"):
        lcd.DeviceChunk.from_bam(path, path + ".bai", "chrZ", 1, 10)


def _write_real_bam(path, ch, block):
    """the bundled chunk's reads (EQX CIGARs from the fixture's digars, 4-bit bases, qualities) as a sorted BAM + .bai"""
    refs = [("chr11", 135086622)]
    hdr = b"@HD\tVN:1.6\tSO:coordinate\n"
    d = b"BAM\x01" + struct.pack("<i", len(hdr)) + hdr + struct.pack("<i", 1) + struct.pack("<i", 6) + b"chr11\0" + struct.pack("<i", refs[0][1])
    order = sorted(range(ch.n_reads), key=lambda i: int(ch.digars[i][0][0]))
    recs = []
    for i in order:
        cig = np.asarray(_cigar_of(ch.digars[i]), "<u4"); pos = int(ch.digars[i][0][0]) - 1; qlen = int(ch.qlen[i])
        name = f"m/{i}/ccs".encode() + b"\0"
        body = struct.pack("<iiBBHHHiiii", 0, pos, len(name), 60, 4680, len(cig), 0, qlen, -1, -1, 0) + name + cig.tobytes() + np.asarray(ch.bseq[i], np.uint8).tobytes()[:(qlen + 1) // 2] + \
            np.asarray(ch.qual[i], np.uint8).tobytes() + b"NMi" + struct.pack("<i", 1)
        u0 = len(d); d += struct.pack("<i", len(body)) + body
        rl = sum(int(c >> 4) for c in cig if int(c & 0xf) in (0, 2, 3, 7, 8))
        recs.append(dict(tid=0, pos=pos, end=pos + max(rl, 1), u0=u0, u1=len(d), idx=i))
    coffs = []
    open(path, "wb").write(_bgzf(d, block=block, offsets=coffs))
    for x in recs:
        x["vbeg"] = (coffs[x["u0"] // block] << 16) | (x["u0"] % block)
        x["vend"] = (coffs[x["u1"] // block] << 16) | (x["u1"] % block) if x["u1"] < len(d) else ((coffs[(len(d) - 1) // block] << 16) | ((len(d) - 1) % block + 1))
    _write_bai(path + ".bai", 1, recs)
    return order


def test_real_chunk_from_a_bam_file_to_region_results(lcd, io, oracle, tmp_path):
    """the bundled HG002 chunk written as a BAM: file -> device inflate -> device records -> digars in HBM -> region jobs whose bases are unpacked from the inflated
    stream.  Region results == the host-array chunk's, digest for digest; no digar and no read base crossed PCIe (copy counters), and the sort order of a long region
    (error rates from qualities that never left HBM) == the host rule's"""
    ch = tc.Chunk()
    path = str(tmp_path / "real.bam")
    order = _write_real_bam(path, ch, 65280)
    o = int(ch.z["ref_beg"]); ref = ch.z["ref"]
    reg_beg, reg_end = o, o + len(ref) - 1
    c0 = lcd.copy_counters()
    dev = lcd.DeviceChunk.from_bam(path, path + ".bai", "chr11", reg_beg, reg_end, min_mapq=30)
    c1 = lcd.copy_counters()
    assert c1 == c0                                                      # nothing of the reads' digars or bases crossed PCIe (the compressed blocks went up)
    h = _host_reads(io, path, "chr11", reg_beg, reg_end, 30)
    assert h["n"] == ch.n_reads == dev.n and dev.meta["names"] == [f"m/{i}/ccs" for i in order]
    hostc = _same_chunk(lcd, dev, h, reg_beg, reg_end, 135086622)
    info = dev.read_info(); ivs = dev.intervals()
    kept = [i for i in range(dev.n) if info["status"][i] == 0]
    chunk_noisy = np.concatenate([ivs[i][0][ivs[i][1]] for i in kept])
    regs = lcd.pre_process_noisy_regs(chunk_noisy, np.zeros((0, 2), np.int64), [info["beg"][i] for i in kept], [info["end"][i] for i in kept], [ivs[i][0] for i in kept])
    used, pr, pb, pe = [], [], [], []
    for s_, e_, _ in regs:
        beg, end = int(s_) + 1 - 10, int(e_) + 10
        if end - beg + 1 > 3000 or beg <= o or end >= reg_end:
            continue
        ids = np.array([i for i in kept if info["beg"][i] <= end and info["end"][i] >= beg], np.int32)
        if len(ids) >= 5:
            used.append((beg, end, ids)); pr += list(ids); pb += [beg] * len(ids); pe += [end] * len(ids)
    # one long region (>= 10 kb: reads are ordered by their error rate, src/align.c:963-985)
    lb = o + 20000; le = lb + 10500
    lids = np.array([i for i in kept if info["beg"][i] <= lb and info["end"][i] >= le], np.int32)[:12]
    assert len(lids) >= 5 and len(used) >= 8
    used.append((lb, le, lids)); pr += list(lids); pb += [lb] * len(lids); pe += [le] * len(lids)
    sa, sb = dev.region_slices(pr, pb, pe, 10), hostc.region_slices(pr, pb, pe, 10)
    assert all((x == y).all() for x, y in zip(sa, sb))
    rb, re_, cv = sa
    opt = lcd.default_opt()
    b1, b2 = lcd.RegionBatch(opt), lcd.RegionBatch(opt)
    zeros = np.zeros(dev.n, np.int32); ps = np.full(dev.n, -1, np.int64)
    at = 0
    for beg, end, ids in used:
        n = len(ids)
        for chunk, b in ((dev, b1), (hostc, b2)):
            chunk.add_region(b, beg, end, ids, rb[at:at + n], re_[at:at + n], cv[at:at + n], zeros[ids], ps[ids], ref[beg - o:end - o + 1])
        at += n
    c2 = lcd.copy_counters()
    b1.upload(); b1.run(); b1.download()
    assert lcd.copy_counters() == c2
    b2.upload(); b2.run(); b2.download()
    assert b1.digest() == b2.digest()
    k = len(used) - 1
    assert (b1.sorted_ids(k) == b2.sorted_ids(k)).all()
    # the host rule on the host's qualities gives the same doubles -> the same order: spot-check that the order is not the trivial one
    assert b1.result(k)["n_cons"] >= 0
    for b in (b1, b2):
        b.close()
    dev.close(); hostc.close()


def test_cg_tag_cigars_and_malformed_records_on_the_device(lcd, io, tmp_path):
    rng = np.random.default_rng(5)
    n_ops = 70001
    ops = np.empty(n_ops, "<u4"); ops[0::2] = (np.uint32(3) << 4) | 7; ops[1::2] = (np.uint32(1) << 4) | 8
    qlen = int((ops >> 4).sum()); rl = qlen
    seq = rng.integers(1, 9, qlen).astype(np.uint8)
    packed = ((np.append(seq, 0)[0:2 * ((qlen + 1) // 2):2] << 4) | np.append(seq, 0)[1:2 * ((qlen + 1) // 2):2]).astype(np.uint8)
    qual = rng.integers(0, 60, qlen).astype(np.uint8)
    name = b"ultralong\0"
    placeholder = np.array([(qlen << 4) | 4, (rl << 4) | 3], "<u4")
    head = struct.pack("<iiBBHHHiiii", 0, 5000, len(name), 60, 4680, 2, 0, qlen, -1, -1, 0) + name + placeholder.tobytes() + packed.tobytes() + qual.tobytes()
    good = head + b"NMi" + struct.pack("<i", 3) + b"ZZZabc\0" + b"CGBI" + struct.pack("<i", n_ops) + ops.tobytes() + b"XXc\x01"
    path = str(tmp_path / "cg.bam")

    def write(bodies):
        _one_record_bam(path, bodies)
        # one record at the start of the data behind the header: its virtual offset from the header's length
        hdr_len = 4 + 4 + len(b"@HD\tVN:1.6\tSO:coordinate\n") + 4 + 4 + 6 + 4
        assert hdr_len < 30000
        recs = [dict(tid=0, pos=5000, end=5000 + rl, vbeg=hdr_len, vend=(1 << 40))]
        _write_bai(path + ".bai", 1, recs)
    write([good])
    h = _host_reads(io, path, "chr11", 1, 2000000, 30)
    assert h["n"] == 1 and len(h["cig"][0]) == n_ops
    dev = lcd.DeviceChunk.from_bam(path, path + ".bai", "chr11", 1, 2000000, min_mapq=30)
    ref = _same_chunk(lcd, dev, h, 1, 2000000, 2000000)
    assert dev.meta["n_cigar"][0] == n_ops and dev.meta["end_pos"][0] == 5000 + rl
    ref.close(); dev.close()
    # htslib's bam_tag2cigar: B,i counts as well; no tag / another type / a shorter array leave the record's own (placeholder) CIGAR in place -- same as the host loader
    for tail, want in ((b"CGBi" + struct.pack("<i", n_ops) + ops.tobytes(), n_ops), (b"NMi" + struct.pack("<i", 3), 2), (b"CGZnot-a-cigar\0", 2),
                       (b"CGBI" + struct.pack("<i", 1) + ops[:1].tobytes(), 2)):
        write([head + tail])
        h = _host_reads(io, path, "chr11", 1, 2000000, 30)
        assert h["n"] == 1 and len(h["cig"][0]) == want
        dev = lcd.DeviceChunk.from_bam(path, path + ".bai", "chr11", 1, 2000000, min_mapq=30)
        assert dev.meta["n_cigar"][0] == want and dev.meta["end_pos"][0] == 5000 + rl
        dev.close()
    short = struct.pack("<iiBBHHHiiii", 0, 5000, len(name), 60, 4680, 1, 0, 100000, -1, -1, 0) + name + struct.pack("<I", (100000 << 4) | 7) + b"\x11" * 50
    write([short])
    with pytest.raises(RuntimeError, match="malformed"):
        lcd.DeviceChunk.from_bam(path, path + ".bai", "chr11", 1, 2000000)
