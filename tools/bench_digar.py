"""f2 (per-read part) throughput: lcd_digar_batch over the reads of many chunks vs the oracle on one core.
usage (GPU box): python tools/bench_digar.py [n_reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from longcalld_amd import align as lcd
from oracle import pyoracle as orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(1)
cigs, quals, pos0 = [], [], []
for i in range(n):   # HiFi shape: 15 kb reads, one difference per ~150 bp
    ops, ql = [], 0
    for _ in range(100):
        l = int(rng.integers(20, 300)); ops.append((l << 4) | 7); ql += l
        x = rng.random()
        if x < 0.5: ops.append((1 << 4) | 8); ql += 1
        elif x < 0.75: l = int(rng.integers(1, 4)); ops.append((l << 4) | 1); ql += l
        else: ops.append((int(rng.integers(1, 4)) << 4) | 2)
    ops.append((50 << 4) | 7); ql += 50
    cigs.append(np.array(ops, np.uint32)); quals.append(np.full(ql, 35, np.uint8)); pos0.append(int(rng.integers(1000, 10_000_000)))
orc.build()
lcd.digar_batch(pos0[:100], cigs[:100], quals[:100], 0, 1 << 40, 1 << 40)   # warm-up
t0 = time.perf_counter(); out = lcd.digar_batch(pos0, cigs, quals, 0, 1 << 40, 1 << 40); t_gpu = time.perf_counter() - t0
m = min(n, 2000)
t0 = time.perf_counter()
for i in range(m):
    orc.collect_digar_from_eqx_cigar(pos0[i], cigs[i], quals[i], 0, 1 << 40, 1 << 40)
t_cpu = (time.perf_counter() - t0) / m * n
nd = sum(len(o["digars"]) for o in out)
print(f"{n} reads, {nd} digars: lcd_digar_batch {t_gpu:.3f} s end to end through the Python mirror (packing, PCIe both ways, unpacking) = {n / t_gpu:.0f} reads/s; "
      f"oracle on one core through ctypes {t_cpu:.3f} s = {n / t_cpu:.0f} reads/s")
