"""SURVEY 8(f) f3 -- BGZF / BAM / FASTA in front of the hot path, without htslib (liblcd_hotpath.so, lcd_io.cpp; host code, no GPU needed).
The checker is an independent Python writer / reader (zlib + struct): a seeded BAM is written block by block, the library reads the region back;
where the reference's bundled test BAM exists (build container only) it is read too and compared with a plain gzip + struct decoding."""
import ctypes as C
import os
import struct
import zlib

import numpy as np
import pytest

i32p, i64p, u8p, u32p, u64p = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)


class BamReads(C.Structure):
    _fields_ = [("n_reads", C.c_int), ("tid", C.c_int), ("n_targets", C.c_int), ("target_len", C.c_int64), ("pos0", i64p), ("end_pos", i64p), ("mapq", i32p),
                ("flag", i32p), ("n_cigar", i32p), ("qlen", i32p), ("cigar_off", u64p), ("cigar_pool", u32p), ("seq_off", u64p), ("seq_pool", u8p),
                ("qual_off", u64p), ("qual_pool", u8p), ("name_off", u64p), ("name_pool", C.POINTER(C.c_char))]


@pytest.fixture(scope="module")
def lib():
    from longcalld_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    L.lcd_bam_load_region.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(BamReads)]
    L.lcd_bam_load_region_indexed.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(BamReads)]
    L.lcd_fasta_fetch.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.POINTER(u8p)]
    L.lcd_fasta_fetch.restype = C.c_int64
    L.lcd_io_last_error.restype = C.c_char_p
    return L


def _bgzf(data, block=4000, offsets=None):
    out = b""
    for o in list(range(0, len(data), block)) + [None]:
        chunk = b"" if o is None else data[o:o + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        bsize = len(comp) + 25
        if offsets is not None and o is not None:
            offsets.append(len(out))                     # compressed offset of the block that holds uncompressed bytes [o, o + block)
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize) + comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    return out


def _reg2bin(beg, end):                                  # SAM specification 5.3
    end -= 1
    for sh, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> sh == end >> sh:
            return base + (beg >> sh)
    return 0


def _write_bai(path, n_ref, recs_with_voff):
    """a .bai as samtools index writes it (SAM specification 5.2): per reference the bins with their chunks and the 16 kb linear index (unset windows take the
    previous window's offset, as htslib's hts_idx_finish fills them)"""
    out = b"BAI\x01" + struct.pack("<i", n_ref)
    for t in range(n_ref):
        bins, lin = {}, {}
        for x in recs_with_voff:
            if x["tid"] != t:
                continue
            b = _reg2bin(x["pos"], x["end"])
            ch = bins.setdefault(b, [])
            if ch and ch[-1][1] == x["vbeg"]:
                ch[-1][1] = x["vend"]
            else:
                ch.append([x["vbeg"], x["vend"]])
            for w in range(x["pos"] >> 14, ((x["end"] - 1) >> 14) + 1):
                lin[w] = min(lin.get(w, x["vbeg"]), x["vbeg"])
        out += struct.pack("<i", len(bins))
        for b in sorted(bins):
            out += struct.pack("<Ii", b, len(bins[b])) + b"".join(struct.pack("<QQ", c[0], c[1]) for c in bins[b])
        n_intv = max(lin) + 1 if lin else 0
        offs, prev = [], 0
        for w in range(n_intv):
            prev = lin.get(w, prev); offs.append(prev)
        out += struct.pack("<i", n_intv) + b"".join(struct.pack("<Q", o) for o in offs)
    open(path, "wb").write(out)


def _make_bam(rng, path, n=300):
    refs = [("chrA", 50000), ("chr11", 2000000)]
    hdr = b"@HD\tVN:1.6\tSO:coordinate\n"
    d = b"BAM\x01" + struct.pack("<i", len(hdr)) + hdr + struct.pack("<i", len(refs))
    for nm, ln in refs:
        d += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    recs = []
    pos = np.sort(rng.integers(1000, 1200000, n))
    for i in range(n):
        qlen = int(rng.integers(50, 3000))
        ops = []
        left = qlen
        if rng.random() < 0.3:
            c = int(rng.integers(1, 20)); ops.append((4, c)); left -= c
        while left > 0:
            ln = int(min(left, rng.integers(1, 400))); ops.append((7, ln)); left -= ln
            if left > 0 and rng.random() < 0.5:
                op = int(rng.choice([8, 1, 2]))
                ln = int(rng.integers(1, 30))
                if op == 2:
                    ops.append((2, ln))
                else:
                    ln = min(ln, left); ops.append((op, ln)); left -= ln
        flag = int(rng.choice([0, 16, 256, 2048, 4], p=[0.45, 0.4, 0.05, 0.05, 0.05]))
        mapq = int(rng.choice([60, 10, 30, 0], p=[0.7, 0.1, 0.15, 0.05]))
        tid = 1 if i >= 5 else 0
        seq = rng.integers(0, 16, qlen).astype(np.uint8)
        packed = ((np.append(seq, 0)[0:2 * ((qlen + 1) // 2):2] << 4) | np.append(seq, 0)[1:2 * ((qlen + 1) // 2):2]).astype(np.uint8)
        qual = rng.integers(0, 60, qlen).astype(np.uint8)
        name = f"read/{i}/ccs".encode() + b"\0"
        cig = np.array([(ln << 4) | op for op, ln in ops], "<u4")
        body = struct.pack("<iiBBHHHiiii", tid, int(pos[i]), len(name), mapq, 4680, len(cig), flag, qlen, -1, -1, 0) + name + cig.tobytes() + packed.tobytes() + qual.tobytes()
        if rng.random() < 0.5:
            body += b"NMi" + struct.pack("<i", 3)
        u0 = len(d)
        d += struct.pack("<i", len(body)) + body
        rl = sum(ln for op, ln in ops if op in (0, 2, 3, 7, 8))
        recs.append(dict(tid=tid, pos=int(pos[i]), end=int(pos[i]) + max(rl, 1), mapq=mapq, flag=flag, cig=cig, seq=packed, qual=qual, name=name[:-1].decode(), qlen=qlen,
                         u0=u0, u1=len(d)))
    block = int(rng.integers(2000, 60000)); coffs = []
    open(path, "wb").write(_bgzf(d, block=block, offsets=coffs))
    coffs.append(coffs[-1] + 1)        # (never used as a block: an end offset exactly at the data's end is normalised below)
    for x in recs:                     # virtual offsets: compressed offset of the block << 16 | offset inside it; a position at a block's end is the next block's start
        x["vbeg"] = (coffs[x["u0"] // block] << 16) | (x["u0"] % block)
        x["vend"] = (coffs[x["u1"] // block] << 16) | (x["u1"] % block) if x["u1"] < len(d) else ((coffs[(len(d) - 1) // block] << 16) | ((len(d) - 1) % block + 1))
    _write_bai(path + ".bai", len(refs), recs)
    return recs


def _check(lib, path, chrom, tid, recs, beg, end, min_mq):
    r = BamReads()
    n = lib.lcd_bam_load_region(path.encode(), chrom, beg, end, min_mq, 3, C.byref(r))
    assert n >= 0, lib.lcd_io_last_error()
    exp = [x for x in recs if x["tid"] == tid and x["pos"] < end and x["end"] > beg - 1 and not (x["flag"] & (4 | 256 | 2048)) and x["mapq"] >= min_mq]
    assert n == len(exp)
    for i, x in enumerate(exp):
        assert r.pos0[i] == x["pos"] and r.end_pos[i] == x["end"] and r.mapq[i] == x["mapq"] and r.flag[i] == x["flag"] and r.qlen[i] == x["qlen"]
        assert r.n_cigar[i] == len(x["cig"]) and [r.cigar_pool[r.cigar_off[i] + k] for k in range(len(x["cig"]))] == list(x["cig"])
        nb = (x["qlen"] + 1) // 2
        assert bytes(r.seq_pool[r.seq_off[i] + k] for k in range(nb)) == x["seq"].tobytes()
        assert bytes(r.qual_pool[r.qual_off[i] + k] for k in range(0, x["qlen"], 7)) == x["qual"][::7].tobytes()
        assert C.string_at(C.addressof(r.name_pool.contents) + r.name_off[i]).decode() == x["name"]
    lib.lcd_bam_reads_free(C.byref(r))
    if os.path.exists(path + ".bai"):   # the same records through the index
        q = BamReads()
        m = lib.lcd_bam_load_region_indexed(path.encode(), (path + ".bai").encode(), chrom, beg, end, min_mq, C.byref(q))
        assert m == len(exp), lib.lcd_io_last_error()
        for i, x in enumerate(exp):
            assert q.pos0[i] == x["pos"] and q.end_pos[i] == x["end"] and q.flag[i] == x["flag"] and q.qlen[i] == x["qlen"] and q.n_cigar[i] == len(x["cig"])
            assert C.string_at(C.addressof(q.name_pool.contents) + q.name_off[i]).decode() == x["name"]
            nb = (x["qlen"] + 1) // 2
            assert bytes(q.seq_pool[q.seq_off[i] + k] for k in range(0, nb, 5)) == x["seq"].tobytes()[::5]
        lib.lcd_bam_reads_free(C.byref(q))
    return n


def test_bam_region_loader_matches_python_writer(lib, tmp_path):
    rng = np.random.default_rng(11)
    path = str(tmp_path / "t.bam")
    recs = _make_bam(rng, path)
    assert _check(lib, path, b"chr11", 1, recs, 1, 2000000, 30) > 100
    assert _check(lib, path, b"chr11", 1, recs, 300001, 800000, 30) > 20         # a 500 kb chunk (src/bam_utils.h:10): boundary reads on both sides
    _check(lib, path, b"chr11", 1, recs, 300001, 800000, 0)
    _check(lib, path, b"chrA", 0, recs, 1, 50000, 30)
    assert _check(lib, path, b"chr11", 1, recs, 1900000, 2000000, 30) == 0      # empty region
    r = BamReads()
    assert lib.lcd_bam_load_region(path.encode(), b"chrX", 1, 10, 30, 1, C.byref(r)) < 0 and b"contig" in lib.lcd_io_last_error()
    assert lib.lcd_bam_load_region(str(tmp_path / "nope.bam").encode(), b"chr11", 1, 10, 30, 1, C.byref(r)) < 0


def _one_record_bam(path, body_of):
    refs = [("chr11", 2000000)]
    hdr = b"@HD\tVN:1.6\tSO:coordinate\n"
    d = b"BAM\x01" + struct.pack("<i", len(hdr)) + hdr + struct.pack("<i", len(refs))
    for nm, ln in refs:
        d += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    for body in body_of:
        d += struct.pack("<i", len(body)) + body
    open(path, "wb").write(_bgzf(d, block=30000))


def test_long_cigar_in_cg_tag_and_malformed_records(lib, tmp_path):
    """a read with more than 65 535 CIGAR operations keeps `<l_seq>S<ref_len>N` in the record and the operations in CG:B,I (SAM specification 4.2.2); htslib's
    bam_read1, behind the reference's sam_itr_next (src/bam_utils.c:1672), puts the real CIGAR back -- so does the loader.  A record whose fields run past its
    block_size is an error and not an out-of-bounds read; a placeholder without a usable tag keeps its own two operations, as in htslib's bam_tag2cigar."""
    rng = np.random.default_rng(5)
    n_ops = 70001
    ops = np.empty(n_ops, "<u4")
    ops[0::2] = (np.uint32(3) << 4) | 7                       # 3=
    ops[1::2] = (np.uint32(1) << 4) | 8                       # 1X
    qlen = int((ops >> 4).sum()); rl = qlen
    seq = rng.integers(1, 9, qlen).astype(np.uint8)
    packed = ((np.append(seq, 0)[0:2 * ((qlen + 1) // 2):2] << 4) | np.append(seq, 0)[1:2 * ((qlen + 1) // 2):2]).astype(np.uint8)
    qual = rng.integers(0, 60, qlen).astype(np.uint8)
    name = b"ultralong\0"
    placeholder = np.array([(qlen << 4) | 4, (rl << 4) | 3], "<u4")
    head = struct.pack("<iiBBHHHiiii", 0, 5000, len(name), 60, 4680, 2, 0, qlen, -1, -1, 0) + name + placeholder.tobytes() + packed.tobytes() + qual.tobytes()
    good = head + b"NMi" + struct.pack("<i", 3) + b"ZZZabc\0" + b"CGBI" + struct.pack("<i", n_ops) + ops.tobytes() + b"XXc\x01"
    path = str(tmp_path / "cg.bam")
    _one_record_bam(path, [good])
    r = BamReads()
    assert lib.lcd_bam_load_region(path.encode(), b"chr11", 1, 2000000, 30, 1, C.byref(r)) == 1, lib.lcd_io_last_error()
    assert r.n_cigar[0] == n_ops and r.qlen[0] == qlen and r.end_pos[0] == 5000 + rl
    got = np.ctypeslib.as_array(r.cigar_pool, shape=(n_ops,))
    assert (got == ops).all()
    lib.lcd_bam_reads_free(C.byref(r))
    # htslib's bam_tag2cigar: B,i is accepted like B,I
    _one_record_bam(path, [head + b"CGBi" + struct.pack("<i", n_ops) + ops.tobytes()])
    assert lib.lcd_bam_load_region(path.encode(), b"chr11", 1, 2000000, 30, 1, C.byref(r)) == 1 and r.n_cigar[0] == n_ops
    lib.lcd_bam_reads_free(C.byref(r))
    # the placeholder without its tag, with a CG tag of another type, or with a CG array shorter than n_cigar: htslib keeps the record's own CIGAR, silently
    for tail in (b"NMi" + struct.pack("<i", 3), b"CGZnot-a-cigar\0", b"CGBI" + struct.pack("<i", 1) + ops[:1].tobytes()):
        _one_record_bam(path, [head + tail])
        assert lib.lcd_bam_load_region(path.encode(), b"chr11", 1, 2000000, 30, 1, C.byref(r)) == 1, lib.lcd_io_last_error()
        assert r.n_cigar[0] == 2 and r.end_pos[0] == 5000 + rl
        assert (np.ctypeslib.as_array(r.cigar_pool, shape=(2,)) == placeholder).all()
        lib.lcd_bam_reads_free(C.byref(r))
    # l_seq larger than the record holds
    short = struct.pack("<iiBBHHHiiii", 0, 5000, len(name), 60, 4680, 1, 0, 100000, -1, -1, 0) + name + struct.pack("<I", (100000 << 4) | 7) + b"\x11" * 50
    _one_record_bam(path, [short])
    assert lib.lcd_bam_load_region(path.encode(), b"chr11", 1, 2000000, 30, 1, C.byref(r)) < 0 and b"malformed" in lib.lcd_io_last_error()


def test_hostile_bgzf_header_is_an_error_not_an_out_of_bounds_read(lib, tmp_path):
    """a BSIZE smaller than header + trailer (or an extra field that runs past the file) must be refused BEFORE the trailer bytes f[end - 8 ..] are read"""
    r = BamReads()
    for bsize, xlen, tail in ((3, 6, b"\0" * 4), (5, 6, b"\0" * 64), (40, 60000, b"\0" * 64)):
        img = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, xlen, 66, 67, 2, bsize) + tail
        path = str(tmp_path / f"hostile{bsize}.bam")
        open(path, "wb").write(img)
        assert lib.lcd_bam_load_region(path.encode(), b"chr11", 1, 10, 30, 1, C.byref(r)) < 0
        assert b"BGZF" in lib.lcd_io_last_error()


def test_fasta_fetch_matches_python(lib, tmp_path):
    rng = np.random.default_rng(12)
    seqs = {"chrA": "".join(rng.choice(list("ACGTNacgt"), 1234)), "chr11": "".join(rng.choice(list("ACGT"), 9001))}
    fa, lw = str(tmp_path / "r.fa"), 60
    with open(fa, "w") as f, open(fa + ".fai", "w") as fi:
        for nm, s in seqs.items():
            f.write(f">{nm} test\n"); off = f.tell()
            for o in range(0, len(s), lw):
                f.write(s[o:o + lw] + "\n")
            fi.write(f"{nm}\t{len(s)}\t{off}\t{lw}\t{lw + 1}\n")
    code = {c: i for i, c in enumerate("ACGT")}
    for nm, beg, end in [("chr11", 1, 9001), ("chr11", 61, 120), ("chr11", 59, 62), ("chrA", 1000, 5000), ("chrA", 1, 1)]:
        p = u8p()
        n = lib.lcd_fasta_fetch(fa.encode(), nm.encode(), beg, end, C.byref(p))
        e = min(end, len(seqs[nm]))
        assert n == e - beg + 1
        assert [p[i] for i in range(n)] == [code.get(c.upper(), 4) for c in seqs[nm][beg - 1:e]]
    p = u8p()
    assert lib.lcd_fasta_fetch(fa.encode(), b"chrZ", 1, 5, C.byref(p)) < 0


def test_vcf_header_lines(lib):
    names = (C.c_char_p * 2)(b"chr1", b"chr11"); lens = (C.c_int64 * 2)(248956422, 135086622)
    t = C.c_void_p()
    lib.lcd_vcf_header.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_char_p, C.POINTER(C.c_void_p)]
    n = lib.lcd_vcf_header(b"0.0.11", b"longcallD call ref.fa in.bam --hifi", b"20260928", 2, names, lens, b"HG002", C.byref(t))
    text = C.string_at(t).decode().splitlines()
    assert n == len(text) and text[0] == "##fileformat=VCFv4.2" and "##contig=<ID=chr11,length=135086622>" in text
    assert text[-1].split("\t") == ["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT", "HG002"]
    assert sum(l.startswith("##FORMAT=") for l in text) == 8 and sum(l.startswith("##INFO=") for l in text) == 12 and sum(l.startswith("##FILTER=") for l in text) == 4


@pytest.mark.skipif(not os.path.exists("/root/reference/test_data/HG002_chr11_hifi_test.bam"), reason="the reference's bundled test BAM exists in the build container only")
def test_real_test_bam_matches_plain_gzip_decoding(lib):
    """SURVEY 8d config 1 input: every primary chr11 read of the bundled BAM, against a gzip + struct decoding of the same file"""
    import gzip
    path = "/root/reference/test_data/HG002_chr11_hifi_test.bam"
    d = gzip.open(path).read()
    lt, = struct.unpack_from("<i", d, 4); o = 8 + lt
    nref, = struct.unpack_from("<i", d, o); o += 4
    names = []
    for _ in range(nref):
        ln, = struct.unpack_from("<i", d, o); o += 4
        names.append(d[o:o + ln - 1].decode()); o += ln + 4
    recs = []
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o); o += 4
        refid, pos, lname, mapq, _bin, ncig, flag, lseq = struct.unpack_from("<iiBBHHHi", d, o)
        p = o + 32 + lname
        cig = np.frombuffer(d, "<u4", ncig, p); p += 4 * ncig
        seq = np.frombuffer(d, np.uint8, (lseq + 1) // 2, p); p += (lseq + 1) // 2
        qual = np.frombuffer(d, np.uint8, lseq, p)
        rl = int(sum(int(c >> 4) for c in cig if int(c & 15) in (0, 2, 3, 7, 8)))
        recs.append(dict(tid=refid, pos=pos, end=pos + max(rl, 1), mapq=mapq, flag=flag, cig=cig, seq=seq, qual=qual, name=d[o + 32:o + 32 + lname - 1].decode(), qlen=lseq))
        o += bs
    tid = names.index("chr11")
    assert _check(lib, path, b"chr11", tid, recs, 1, 2000000, 30) >= 350
    _check(lib, path, b"chr11", tid, recs, 1236832, 1441808, 30)


def test_indexed_loader_equals_scan_on_the_reference_bam(lib):
    """the reference's bundled HG002 BAM with the .bai samtools wrote for it (build container only: /root/reference is absent elsewhere): every region through
    the index == the same region by scanning the file -- chunk boundaries, linear-index cut, records spanning BGZF blocks, regions without reads"""
    bam = "/root/reference/test_data/HG002_chr11_hifi_test.bam"
    if not os.path.exists(bam):
        pytest.skip("reference test data not present")
    rng = np.random.default_rng(3)
    full = BamReads()
    n_all = lib.lcd_bam_load_region(bam.encode(), b"chr11", 1, 1 << 28, 0, 4, C.byref(full))
    assert n_all > 100
    lo, hi = int(full.pos0[0]), int(max(full.end_pos[i] for i in range(n_all)))
    lib.lcd_bam_reads_free(C.byref(full))
    regs = [(1, 1 << 28), (lo + 1, lo + 1), (hi, hi + 1000), (hi + 1, hi + 5000), (1, lo)] + [(int(b), int(b + w)) for b, w in zip(rng.integers(max(lo - 20000, 1), hi + 20000, 40), rng.integers(1, 120000, 40))]
    n_nonempty = 0
    for beg, end in regs:
        for mq in (0, 30):
            a, b = BamReads(), BamReads()
            na = lib.lcd_bam_load_region(bam.encode(), b"chr11", beg, end, mq, 4, C.byref(a))
            nb = lib.lcd_bam_load_region_indexed(bam.encode(), (bam + ".bai").encode(), b"chr11", beg, end, mq, C.byref(b))
            assert na == nb >= 0, (beg, end, lib.lcd_io_last_error())
            for i in range(na):
                assert a.pos0[i] == b.pos0[i] and a.end_pos[i] == b.end_pos[i] and a.qlen[i] == b.qlen[i] and a.flag[i] == b.flag[i] and a.n_cigar[i] == b.n_cigar[i]
                assert C.string_at(C.addressof(a.name_pool.contents) + a.name_off[i]) == C.string_at(C.addressof(b.name_pool.contents) + b.name_off[i])
            if na:
                k = na - 1; nbytes = (a.qlen[k] + 1) // 2
                assert bytes(a.seq_pool[a.seq_off[k] + j] for j in range(nbytes)) == bytes(b.seq_pool[b.seq_off[k] + j] for j in range(nbytes))
                assert [a.cigar_pool[a.cigar_off[k] + j] for j in range(a.n_cigar[k])] == [b.cigar_pool[b.cigar_off[k] + j] for j in range(b.n_cigar[k])]
            n_nonempty += na > 0
            lib.lcd_bam_reads_free(C.byref(a)); lib.lcd_bam_reads_free(C.byref(b))
    assert n_nonempty > 20
    r = BamReads()
    assert lib.lcd_bam_load_region_indexed(bam.encode(), b"/nonexistent.bai", b"chr11", 1, 10, 0, C.byref(r)) < 0
    assert lib.lcd_bam_load_region_indexed(bam.encode(), bam.encode(), b"chr11", 1, 10, 0, C.byref(r)) < 0 and b".bai" in lib.lcd_io_last_error()
