#!/usr/bin/env python
"""Resident wavefronts / LDS over time from LCD_CHAIN_TIMES=<file> (lcd_host.cpp): how full the chip is during a submission's chain kernels.
usage: python tools/occupancy.py <file> [bins]"""
import sys
import numpy as np
rows = np.loadtxt(sys.argv[1], dtype=np.int64)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 40
thr, lds, mode, b, e = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3] / 1e5, rows[:, 4] / 1e5   # ms
T = e.max()
edges = np.linspace(0, T, nb + 1)
print(f"{len(rows)} chains, {T:.1f} ms; per bin: mean resident wavefronts (of 4096 = 16 per CU), LDS in use (of 40 MB), by class")
for i in range(nb):
    lo, hi = edges[i], edges[i + 1]
    ov = np.clip(np.minimum(e, hi) - np.maximum(b, lo), 0, None) / (hi - lo)
    waves = (ov * (thr // 64)).sum()
    ldsb = (ov * (lds + np.where(thr == 64, 912, 6096))).sum() / 1e6
    per = {t: (ov[thr == t] * (t // 64)).sum() for t in sorted(set(thr))}
    small = (ov[(thr == 64) & (lds <= 8192)]).sum()
    print(f"{lo:7.1f}-{hi:7.1f} ms  waves {waves:7.0f}  LDS {ldsb:5.1f} MB  8K-chains {small:6.0f}  " + " ".join(f"{t}:{v:.0f}" for t, v in per.items()))
