import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from longcalld_amd import align as lcd
from oracle import pyoracle as orc
import testdata_common as tc
orc.build()
ref = tc.Chunk().z["ref"]
rng = np.random.default_rng(1)
big = np.concatenate([ref, rng.integers(0, 4, 500000 - len(ref)).astype(np.uint8)])
for name, s in (("chr11 slice 205 kb", ref), ("500 kb chunk", big)):
    lcd.sdust(s[:1000])
    for T, W in ((5, 20), (20, 64)):
        t0 = time.perf_counter(); g = lcd.sdust(s, T, W); t1 = time.perf_counter(); e = orc.ref_sdust(s, T, W); t2 = time.perf_counter()
        print(f"{name} T={T} W={W}: {len(g)} intervals, GPU {1e3 * (t1 - t0):.1f} ms end to end, reference sdust on one core {1e3 * (t2 - t1):.1f} ms, equal {bool((g == e).all())}")
