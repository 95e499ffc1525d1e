#!/usr/bin/env python
"""K5 (assign_hap_based_on_germline_het_vars_kmeans, src/assign_hap.c:473) throughput: one wavefront per chunk, chunks batched.
Chunk shape from SURVEY 8: ~1 000 reads and 500-800 candidate variants per 500 kb chunk, a read spans 15-30 variants."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longcalld_amd import align, jobs
from oracle import pyoracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(11)
probs = [jobs.make_hap_problem(rng, n_vars=int(rng.integers(500, 800)), n_reads=1000, span=(15, 30)) for _ in range(n)]
cates = [jobs.GERMLINE_CLEAN] * n
align.assign_hap_batch(probs[:4], cates[:4])            # warm-up
t0 = time.perf_counter(); st = align.assign_hap_batch(probs, cates); t = time.perf_counter() - t0
m = min(n, 20)
c0 = time.perf_counter()
ref = [pyoracle.assign_hap_germline(p, jobs.GERMLINE_CLEAN) for p in probs[:m]]
c = time.perf_counter() - c0
for a, b in zip(st[:m], ref):
    assert (a["haps"] == b["haps"]).all() and (a["phase_sets"] == b["phase_sets"]).all()
span = sum(int((p["end_var_idx"] - p["start_var_idx"] + 1).clip(0).sum()) for p in probs)
print(f"K5: {n} chunks in {t*1e3:.1f} ms (host->device, kernel, device->host) = {n/t:.0f} chunks/s = {n*1000/t/1e6:.2f} M reads/s; "
      f"oracle {m/c:.1f} chunks/s on one core; read-variant cells {span/1e6:.2f} M; parity on the first {m} chunks ok")
