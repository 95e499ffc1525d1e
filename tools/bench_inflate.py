"""f3 measurement: BGZF inflate on the device (lcd_bgzf_inflate_dev) against zlib on the host's cores, on a BAM-like stream.
usage: python tools/bench_inflate.py [MB of uncompressed stream, default 256] [repeats]"""
import ctypes as C
import json
import os
import struct
import sys
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longcalld_amd import _lib  # noqa: E402


def bam_like(rng, nbytes):
    """records shaped like HiFi reads: core + name + CIGAR + 4-bit bases (near-random) + qualities (a few common values)"""
    out, tot, i = [], 0, 0
    qv = np.array([93, 93, 93, 93, 80, 70, 60, 50, 40, 30, 20, 10], np.uint8)
    while tot < nbytes:
        qlen = int(rng.integers(8000, 20000))
        name = b"m64011_190830_220126/%d/ccs\0" % i
        ncig = int(rng.integers(5, 60))
        cig = ((rng.integers(1, 2000, ncig).astype("<u4") << 4) | rng.choice(np.array([7, 8, 1, 2], "<u4"), ncig)).tobytes()
        seq = rng.integers(0, 4, qlen).astype(np.uint8)
        code = np.array([1, 2, 4, 8], np.uint8)[seq]
        packed = ((np.append(code, 0)[0:2 * ((qlen + 1) // 2):2] << 4) | np.append(code, 0)[1:2 * ((qlen + 1) // 2):2]).astype(np.uint8).tobytes()
        qual = qv[np.minimum(rng.geometric(0.45, qlen) - 1, len(qv) - 1)].tobytes()
        body = struct.pack("<iiBBHHHiiii", 0, 100 * i, len(name), 60, 4680, ncig, 0, qlen, -1, -1, 0) + name + cig + packed + qual + b"NMi" + struct.pack("<i", 3)
        rec = struct.pack("<i", len(body)) + body
        out.append(rec); tot += len(rec); i += 1
    return b"".join(out)[:nbytes]


def measure(mb=256, reps=3, host_threads=(8, 32, 0)):
    """-> dict: device inflate (kernel / upload / call ms, GB/s) and host zlib on 1 and on `host_threads` threads (0 = all cores) for `mb` MB of BAM-like data"""
    rng = np.random.default_rng(1)
    data = bam_like(rng, mb << 20)
    blocks = [data[o:o + 65280] for o in range(0, len(data), 65280)]

    def member(chunk):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        return struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    with ThreadPoolExecutor(16) as ex:
        members = list(ex.map(member, blocks))
    image = b"".join(members) + member(b"")
    import os
    L = C.CDLL(os.environ.get("LCD_LIB", _lib.LIB_PATH))
    L.lcd_bgzf_inflate_dev.restype = C.c_void_p; L.lcd_bgzf_inflate_dev.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    L.lcd_inflated_size.restype = C.c_size_t; L.lcd_inflated_size.argtypes = [C.c_void_p]
    L.lcd_inflated_kernel_ms.restype = C.c_double; L.lcd_inflated_kernel_ms.argtypes = [C.c_void_p]
    L.lcd_inflated_upload_ms.restype = C.c_double; L.lcd_inflated_upload_ms.argtypes = [C.c_void_p]
    L.lcd_inflated_to_host.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_char_p]
    L.lcd_inflated_free.argtypes = [C.c_void_p]; L.lcd_inflated_free.restype = None
    L.lcd_io_last_error.restype = C.c_char_p
    res = dict(uncompressed_mb=len(data) / 2**20, compressed_mb=len(image) / 2**20, blocks=len(blocks))
    for verify in (1, 0):
        best = None
        for r in range(reps + 1):
            t0 = time.perf_counter()
            h = L.lcd_bgzf_inflate_dev(image, len(image), verify)
            wall = (time.perf_counter() - t0) * 1e3
            assert h, L.lcd_io_last_error()
            km, um = L.lcd_inflated_kernel_ms(h), L.lcd_inflated_upload_ms(h)
            if r == 0:   # first call: check the bytes once
                out = C.create_string_buffer(len(data))
                assert L.lcd_inflated_to_host(h, 0, len(data), out) == 0 and out.raw[:len(data)] == data
            L.lcd_inflated_free(h)
            if r and (best is None or km < best[0]):
                best = (km, um, wall)
        res["device_crc%d" % verify] = dict(kernel_ms=round(best[0], 3), upload_ms=round(best[1], 3), call_ms=round(best[2], 3),
                                            kernel_GBps_out=round(len(data) / best[0] / 1e6, 2), kernel_GBps_in=round(len(image) / best[0] / 1e6, 2))
    raw = [m[18:-8] for m in members]

    def inf(c):
        return len(zlib.decompress(c, -15))
    t0 = time.perf_counter(); n1 = sum(inf(c) for c in raw[:400]); t1 = time.perf_counter() - t0
    res["host_zlib_1_thread_GBps_out"] = round(n1 / t1 / 1e9, 3)
    import os
    for th in sorted({t if t > 0 else (os.cpu_count() or 1) for t in host_threads}):
        with ThreadPoolExecutor(th) as ex:
            t0 = time.perf_counter(); n = sum(ex.map(inf, raw, chunksize=8)); t = time.perf_counter() - t0
        res["host_zlib_%d_threads_GBps_out" % th] = round(n / t / 1e9, 3)
    res["host_cores"] = os.cpu_count()
    return res


def main():
    mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    print(json.dumps(measure(mb, reps)))


if __name__ == "__main__":
    main()
