#!/usr/bin/env python
"""HBM bytes per step of the chain kernels from tools/gpu_pmc_quick.sh's two passes.  usage: python tools/pmc_quick.py <tag>"""
import sys
sys.path.insert(0, "tools")
from rocprof_summary import timed_half
tag = sys.argv[1]
f = timed_half(f"gpurun_out/pmc_fetch_{tag}/f_results.db", ["FETCH_SIZE"]); w = timed_half(f"gpurun_out/pmc_write_{tag}/w_results.db", ["WRITE_SIZE"])
fe = sum(v[1].get("FETCH_SIZE", 0) for k, v in f.items() if "lcd_poa_chain_kernel" in k); wr = sum(v[1].get("WRITE_SIZE", 0) for k, v in w.items() if "lcd_poa_chain_kernel" in k)
print(f"{tag}: fetch {2 * fe * 1024 / 20 / 1e9:.3f} GB/step (x2 corrected), write {wr * 1024 / 20 / 1e9:.3f} GB/step, total {(2 * fe + wr) * 1024 / 20 / 1e9:.3f} GB/step")
