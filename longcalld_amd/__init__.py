"""longcalld_amd -- MI355X (gfx950) implementation of longcallD's per-noisy-region alignment/phasing hot path.

The product is the C-ABI shared library ``liblcd_hotpath.so`` (see ``include/lcd_hotpath.h``); this package
only loads it and mirrors the reference's ``src/align.h`` interface for Python callers (tests, bench).
Importing :mod:`longcalld_amd.align` fails loudly if the HIP library is missing -- there is no CPU path.
"""
from ._lib import load_library, LcdError  # noqa: F401

__all__ = ["load_library", "LcdError"]
