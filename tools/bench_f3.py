"""f3 measurement: one 500 kb chunk of a 30x HiFi-like BAM -> lcd_chunk_t, through the host loader (lcd_bam_load_region_indexed: zlib + record loop on the calling thread,
then lcd_chunk_create uploads the records) and through the device path (lcd_chunk_create_from_bam: compressed blocks up, inflate + records + digars in HBM).
usage: python tools/bench_f3.py [n_reads, default 1000] [repeats]"""
import ctypes as C
import json
import struct
import sys
import time

import numpy as np

import os  # noqa: E402
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
from longcalld_amd import _lib, align as lcd  # noqa: E402
from test_io import BamReads, _bgzf, _write_bai  # noqa: E402


def make_bam(path, n_reads, rng, span=500000, start=1000000):
    refs = [("chr11", 135086622)]
    hdr = b"@HD\tVN:1.6\tSO:coordinate\n"
    d = bytearray(b"BAM\x01" + struct.pack("<i", len(hdr)) + hdr + struct.pack("<i", 1) + struct.pack("<i", 6) + b"chr11\0" + struct.pack("<i", refs[0][1]))
    pos = np.sort(rng.integers(start - 10000, start + span, n_reads))
    qv = np.array([93, 93, 93, 93, 80, 70, 60, 50, 40, 30, 20, 10], np.uint8)
    recs = []
    for i in range(n_reads):
        qlen = int(rng.integers(10000, 20000))
        ops, left = [], qlen
        while left > 0:
            ln = int(min(left, rng.geometric(1 / 500.0))); ops.append((7, ln)); left -= ln
            if left > 0:
                op = int(rng.choice([8, 1, 2], p=[0.5, 0.25, 0.25]))
                if op == 2:
                    ops.append((2, 1))
                else:
                    ops.append((op, 1)); left -= 1
        cig = np.array([(ln << 4) | op for op, ln in ops], "<u4")
        seq = np.array([1, 2, 4, 8], np.uint8)[rng.integers(0, 4, qlen)]
        packed = ((np.append(seq, 0)[0:2 * ((qlen + 1) // 2):2] << 4) | np.append(seq, 0)[1:2 * ((qlen + 1) // 2):2]).astype(np.uint8)
        qual = qv[np.minimum(rng.geometric(0.45, qlen) - 1, len(qv) - 1)]
        name = b"m64011_190830_220126/%d/ccs\0" % i
        body = struct.pack("<iiBBHHHiiii", 0, int(pos[i]), len(name), 60, 4680, len(cig), 0, qlen, -1, -1, 0) + name + cig.tobytes() + packed.tobytes() + qual.tobytes() + b"NMi" + struct.pack("<i", 3)
        u0 = len(d); d += struct.pack("<i", len(body)) + body
        rl = sum(ln for op, ln in ops if op in (2, 7, 8))
        recs.append(dict(tid=0, pos=int(pos[i]), end=int(pos[i]) + rl, u0=u0, u1=len(d)))
    block, coffs = 65280, []
    d = bytes(d)
    open(path, "wb").write(_bgzf(d, block=block, offsets=coffs))
    for x in recs:
        x["vbeg"] = (coffs[x["u0"] // block] << 16) | (x["u0"] % block)
        x["vend"] = (coffs[x["u1"] // block] << 16) | (x["u1"] % block) if x["u1"] < len(d) else ((coffs[(len(d) - 1) // block] << 16) | ((len(d) - 1) % block + 1))
    _write_bai(path + ".bai", 1, recs)
    return len(d)


def measure(n=1000, reps=5, path="/tmp/f3_bench.bam"):
    """-> dict: one 500 kb chunk of a 30x HiFi-like BAM of n reads -> lcd_chunk_t through the host loader and through the device path (medians of `reps`)"""
    rng = np.random.default_rng(3)
    ubytes = make_bam(path, n, rng)
    L = C.CDLL(_lib.LIB_PATH)
    L.lcd_bam_load_region_indexed.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(BamReads)]
    beg, end = 1000000, 1500000
    res = dict(reads_in_file=n, uncompressed_mb=round(ubytes / 2**20, 1), compressed_mb=round(os.path.getsize(path) / 2**20, 1))
    host, dev = [], []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        q = BamReads()
        m = L.lcd_bam_load_region_indexed(path.encode(), (path + ".bai").encode(), b"chr11", beg, end, 30, C.byref(q))
        t1 = time.perf_counter()
        cig = [np.ctypeslib.as_array(q.cigar_pool, shape=(int(q.cigar_off[i]) + q.n_cigar[i],))[int(q.cigar_off[i]):] for i in range(m)]
        seq = [np.ctypeslib.as_array(q.seq_pool, shape=(int(q.seq_off[i]) + (q.qlen[i] + 1) // 2,))[int(q.seq_off[i]):] for i in range(m)]
        qual = [np.ctypeslib.as_array(q.qual_pool, shape=(int(q.qual_off[i]) + q.qlen[i],))[int(q.qual_off[i]):] for i in range(m)]
        pos0 = [q.pos0[i] for i in range(m)]
        t2 = time.perf_counter()
        ch = lcd.DeviceChunk(pos0, cig, qual, seq, beg, end, 135086622)
        t3 = time.perf_counter()
        info_h = ch.read_info(); ch.close(); L.lcd_bam_reads_free(C.byref(q))
        t4 = time.perf_counter()
        dv = lcd.DeviceChunk.from_bam(path, path + ".bai", "chr11", beg, end, min_mapq=30)
        t5 = time.perf_counter()
        info_d = dv.read_info(); dv.close()
        assert m == dv.n and all((info_h[k] == info_d[k]).all() for k in info_h)
        if r:
            host.append(((t1 - t0) * 1e3, (t3 - t2) * 1e3)); dev.append((t5 - t4) * 1e3)
    host.sort(key=lambda x: x[0] + x[1]); dev.sort()
    res.update(reads_in_region=m, host_load_ms=round(host[len(host) // 2][0], 2), host_chunk_create_ms=round(host[len(host) // 2][1], 2),
               device_path_ms=round(dev[len(dev) // 2], 2), note="host_chunk_create includes the Python wrapper's array packing; host_load is one thread (the reference's per-chunk worker)")
    return res


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 5)))
