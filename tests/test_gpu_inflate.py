"""SURVEY 8(f) f3 -- BGZF blocks inflated on the device (inflate_kernel.hip, lcd_bgzf_inflate_dev): the checker is Python's zlib (the writer and the reader of the
expected bytes), on data that exercises every part of RFC 1951 the decoder implements: stored / fixed / dynamic blocks, long codes (15 bits, behind the 11-bit primary
table), matches at distance 1 and at the window's far end, blocks of 1 byte and of 65 280 bytes, several deflate blocks inside one BGZF block, and the container's
own checks (ISIZE, CRC-32, truncation)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
u8p = C.POINTER(C.c_uint8)


@pytest.fixture(scope="module")
def lib():
    from longcalld_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    L.lcd_bgzf_inflate_dev.restype = C.c_void_p
    L.lcd_bgzf_inflate_dev.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
    L.lcd_inflated_size.restype = C.c_size_t
    L.lcd_inflated_size.argtypes = [C.c_void_p]
    L.lcd_inflated_n_blocks.restype = C.c_size_t
    L.lcd_inflated_n_blocks.argtypes = [C.c_void_p]
    L.lcd_inflated_kernel_ms.restype = C.c_double
    L.lcd_inflated_kernel_ms.argtypes = [C.c_void_p]
    L.lcd_inflated_to_host.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_char_p]
    L.lcd_inflated_free.argtypes = [C.c_void_p]
    L.lcd_inflated_free.restype = None
    L.lcd_io_last_error.restype = C.c_char_p
    return L


def _member(chunk, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, crc=None, isize=None):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    if flush_every:     # several deflate blocks inside one BGZF block (Z_FULL_FLUSH ends a block and emits an empty stored one)
        comp = b"".join(co.compress(chunk[o:o + flush_every]) + co.flush(zlib.Z_FULL_FLUSH) for o in range(0, len(chunk), flush_every)) + co.flush()
    else:
        comp = co.compress(chunk) + co.flush()
    assert len(comp) + 25 < 65536
    return (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp
            + struct.pack("<II", (zlib.crc32(chunk) if crc is None else crc) & 0xffffffff, len(chunk) if isize is None else isize))


EOF_BLOCK = _member(b"")


def _inflate(lib, image, verify=1):
    h = lib.lcd_bgzf_inflate_dev(image, len(image), verify)
    if not h:
        return None
    n = lib.lcd_inflated_size(h)
    out = C.create_string_buffer(n)
    assert lib.lcd_inflated_to_host(h, 0, n, out) == 0
    lib.lcd_inflated_free(h)
    return out.raw[:n]


def _payloads(rng):
    text = (b"ACGTTGCAAGGCTTAACCGGTTAACC" * 40 + bytes(rng.integers(33, 74, 600).astype(np.uint8))) * 30
    bam_like = b"".join(struct.pack("<iiBBHHHiiii", 1, int(p), 12, 60, 4680, 3, 0, 150, -1, -1, 0) + b"read/%07d\0" % i + bytes(rng.integers(0, 256, 75).astype(np.uint8))
                        + bytes(rng.integers(20, 45, 150).astype(np.uint8)) for i, p in enumerate(np.sort(rng.integers(0, 1 << 28, 200))))
    skew = bytes(np.minimum(rng.geometric(0.03, 60000), 255).astype(np.uint8))           # a long-tailed alphabet: code lengths up to 15 bits
    far = bytes(rng.integers(0, 256, 400).astype(np.uint8))
    far = far + bytes(rng.integers(0, 4, 32300).astype(np.uint8)) + far                       # a match 32 700 bytes back
    return dict(text=text[:65280], bam=bam_like[:65280], skew=skew, run=b"\x07" * 65280, one=b"Z", random=bytes(rng.integers(0, 256, 50000).astype(np.uint8)), far=far,
                zeros_then_text=b"\0" * 20000 + text[:30000])


def test_every_block_kind_equals_zlib(lib):
    rng = np.random.default_rng(5)
    P = _payloads(rng)
    image, expect = b"", b""
    for name, data in P.items():
        for level, strategy, fe in ((6, zlib.Z_DEFAULT_STRATEGY, 0), (9, zlib.Z_DEFAULT_STRATEGY, 0), (1, zlib.Z_DEFAULT_STRATEGY, 0), (6, zlib.Z_FIXED, 0), (0, zlib.Z_DEFAULT_STRATEGY, 0),
                                    (6, zlib.Z_HUFFMAN_ONLY, 0), (6, zlib.Z_RLE, 0), (6, zlib.Z_DEFAULT_STRATEGY, 7000)):
            d = data[:60000] if level == 0 or strategy == zlib.Z_HUFFMAN_ONLY else data    # (stored / literal-only output must still fit a 64 KB BGZF block)
            if name == "random" and level != 0:
                d = d[:40000]
            image += _member(d, level, strategy, fe); expect += d
    image += EOF_BLOCK
    got = _inflate(lib, image)
    assert got is not None, lib.lcd_io_last_error()
    assert len(got) == len(expect)
    assert got == expect


def test_a_file_of_many_blocks_and_odd_block_sizes(lib):
    rng = np.random.default_rng(11)
    data = b"".join(_payloads(rng)[k] for k in ("bam", "text", "skew", "bam")) * 6
    image, o = b"", 0
    sizes = []
    while o < len(data):
        n = int(rng.choice([1, 2, 3, 15, 16, 17, 255, 4000, 16383, 16384, 16385, 32768, 65280]))
        image += _member(data[o:o + n], int(rng.integers(1, 10))); sizes.append(min(n, len(data) - o)); o += n
    image += EOF_BLOCK
    h = lib.lcd_bgzf_inflate_dev(image, len(image), 1)
    assert h, lib.lcd_io_last_error()
    assert lib.lcd_inflated_n_blocks(h) == len(sizes) and lib.lcd_inflated_size(h) == len(data)
    out = C.create_string_buffer(len(data))
    assert lib.lcd_inflated_to_host(h, 0, len(data), out) == 0
    assert out.raw[:len(data)] == data
    # a range in the middle of the stream
    part = C.create_string_buffer(5000)
    assert lib.lcd_inflated_to_host(h, 123457, 5000, part) == 0 and part.raw[:5000] == data[123457:128457]
    assert lib.lcd_inflated_to_host(h, len(data) - 10, 11, part) < 0
    lib.lcd_inflated_free(h)


def test_container_checks(lib):
    rng = np.random.default_rng(2)
    data = _payloads(rng)["bam"]
    good = _member(data) + EOF_BLOCK
    assert _inflate(lib, good) == data
    # CRC-32 of the trailer does not match the inflated bytes
    bad = _member(data, crc=zlib.crc32(data) ^ 0x10) + EOF_BLOCK
    assert _inflate(lib, bad, 1) is None and b"CRC" in lib.lcd_io_last_error()
    assert _inflate(lib, bad, 0) == data                        # (not asked to check it)
    # ISIZE too small / too large
    assert _inflate(lib, _member(data, isize=len(data) - 1) + EOF_BLOCK) is None and b"ISIZE" in lib.lcd_io_last_error()
    assert _inflate(lib, _member(data, isize=len(data) + 1) + EOF_BLOCK) is None and b"ISIZE" in lib.lcd_io_last_error()
    # a flipped bit inside the deflate stream: some check of the decoder or the CRC catches it
    m = bytearray(_member(data)); m[18 + len(m) // 3] ^= 0x40
    assert _inflate(lib, bytes(m) + EOF_BLOCK) is None
    # not a BGZF block / truncated file
    assert _inflate(lib, b"\x1f\x8b\x08\x00" + good[4:]) is None and b"BGZF" in lib.lcd_io_last_error()
    assert _inflate(lib, good[:len(good) // 2]) is None


def test_truncated_and_crafted_blocks_stop_at_the_blocks_own_bytes(lib):
    """input side of the decoder: a block whose deflate stream is cut short -- with the header's BSIZE and the trailer's ISIZE still claiming the whole
    block -- must end with an error, with and without the CRC check, instead of decoding on into the bytes behind it; a stored block whose LEN reaches behind the
    block likewise.  (The decoder's input windows never start behind the block's last byte + 8: inflate_kernel.hip BitIn::lim.)"""
    rng = np.random.default_rng(9)
    data = bytes(rng.integers(0, 256, 40000).astype(np.uint8))             # incompressible: the stream is about as long as the data
    whole = _member(data)
    comp = whole[18:-8]
    for cut in (1, 7, 300, len(comp) // 2):
        short = comp[:-cut]
        m = (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(short) + 25) + short + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))
        # behind it: another member whose bytes would happily decode as more symbols
        for verify in (0, 1):
            assert _inflate(lib, m + _member(data[:5000]) + EOF_BLOCK, verify) is None, (cut, verify)
    # a stored block whose LEN runs past the member: 1 (final, stored) + LEN 60000 / NLEN, with 100 bytes present
    stored = b"\x01" + struct.pack("<HH", 60000, 60000 ^ 0xffff) + b"x" * 100
    m = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(stored) + 25) + stored + struct.pack("<II", 0, 60000)
    assert _inflate(lib, m + _member(data) + _member(data) + EOF_BLOCK, 0) is None and b"past the block" in lib.lcd_io_last_error()
    # the 1-bit-code worst case of the review: a block of 65 280 equal bytes is ~80 bytes of stream; cut to 20 it still claims ISIZE 65 280
    run = _member(b"\x07" * 65280)
    short = run[18:-8][:20]
    m = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(short) + 25) + short + struct.pack("<II", zlib.crc32(b"\x07" * 65280) & 0xffffffff, 65280)
    for verify in (0, 1):
        assert _inflate(lib, m + EOF_BLOCK, verify) is None
    # hostile container: BSIZE smaller than header + trailer
    bad = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, 5) + b"\0" * 40
    assert _inflate(lib, bad) is None and b"BSIZE" in lib.lcd_io_last_error()


def test_a_bam_file_image_round_trip(lib, tmp_path):
    """the BAM the host loader's tests write (tests/test_io.py): the device's inflated stream == gzip's, i.e. exactly the bytes the record walk reads"""
    import gzip
    from test_io import _make_bam
    rng = np.random.default_rng(77)
    path = str(tmp_path / "x.bam")
    _make_bam(rng, path, n=400)
    image = open(path, "rb").read()
    assert _inflate(lib, image) == gzip.open(path, "rb").read()
