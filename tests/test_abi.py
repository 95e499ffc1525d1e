"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/lcd_hotpath.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "lcd_hotpath.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lcd_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported():
    from longcalld_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)  # no compute, no GPU needed
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lcd_hotpath.h but not exported"
    assert sorted(_lib.EXPORTS) == names


def test_opt_defaults_match_reference():
    """src/call_var_main.c:140-224 / src/call_var_main.h:20-50 / src/align.h:21-26"""
    from longcalld_amd import _lib
    lib = _lib.load_library()
    o = _lib.LcdOpt()
    lib.lcd_opt_default(ctypes.byref(o))
    assert (o.match, o.mismatch, o.gap_open1, o.gap_ext1, o.gap_open2, o.gap_ext2) == (2, 6, 6, 2, 24, 1)
    assert o.gap_aln == 1 and o.min_dp == 5 and abs(o.min_af - 0.2) < 1e-12 and abs(o.partial_aln_ratio - 1.1) < 1e-12
    assert (o.min_noisy_reg_size_to_sample_reads, o.max_noisy_reg_len, o.noisy_reg_flank_len) == (10000, 50000, 10)
    assert (o.min_hap_full_reads, o.min_hap_reads) == (1, 2) and o.min_sv_len == 30   # LONGCALLD_MIN_SV_LEN


def test_no_cpu_fallback_without_gpu():
    """on a box without a GPU the product path must fail loudly, never compute on the host"""
    import numpy as np
    import pytest
    import torch
    from longcalld_amd import align
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        align.edlib_xgaps(np.zeros(10, np.uint8), np.zeros(10, np.uint8))
    with pytest.raises(Exception):   # the digar walk of collect_noisy_read_info for a chunk's pairs runs on the device or not at all
        align.region_read_slices_batch([0], [100], [200], [np.array([[90, 7, 200, 0]], np.int64)], [200])
    with pytest.raises(Exception):
        align.RegionBatch()
    with pytest.raises(Exception):   # the device-resident chunk, HW-mode edlib and the one-block results: the device or nothing
        align.DeviceChunk([0], [np.array([(100 << 4) | 7], np.uint32)], [np.full(100, 30, np.uint8)], [np.zeros(50, np.uint8)], 1, 1000, 100000)
    with pytest.raises(Exception):
        align.edlib_batch_hw([(np.zeros(30, np.uint8), np.zeros(10, np.uint8))])
    assert align.edlib_infix_aln(np.zeros(30, np.uint8), np.zeros(10, np.uint8))[0] == -1
    with pytest.raises(Exception):   # BGZF inflate + record decode of an indexed BAM on the device: no host path behind this entry
        align.DeviceChunk.from_bam("/nonexistent.bam", "/nonexistent.bam.bai", "chr11", 1, 1000)


def test_product_does_not_import_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "longcalld_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                t = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in t and "lcd_oracle.h" not in t and "liblcd_oracle" not in t, f
