#!/usr/bin/env python
"""lcd_sdust_batch throughput against the reference's own sdust() (oracle/_ref) on chunk-sized references: one call per chunk vs all chunks in one launch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from longcalld_amd import align  # noqa: E402


def main():
    n_chunks, L = int(sys.argv[1]) if len(sys.argv) > 1 else 40, 500000
    rng = np.random.default_rng(1)
    seqs = []
    for _ in range(n_chunks):
        s = rng.integers(0, 4, L).astype(np.uint8)
        for _ in range(40):   # tandem repeats / homopolymers at the density of a human reference
            p = int(rng.integers(0, L - 400)); ln = int(rng.integers(10, 300)); unit = rng.integers(0, 4, int(rng.integers(1, 6))).astype(np.uint8)
            s[p:p + ln] = np.resize(unit, ln)
        seqs.append(s)
    align.sdust(seqs[0])                                   # start-up
    t0 = time.perf_counter(); one = [align.sdust(s) for s in seqs[:8]]; t_one = (time.perf_counter() - t0) / 8
    align.sdust_batch(seqs)
    t0 = time.perf_counter(); got = align.sdust_batch(seqs); t_b = time.perf_counter() - t0
    assert all((a == b).all() for a, b in zip(one, got[:8]))
    line = f"sdust: {n_chunks} chunks of {L} bp: one call per chunk {t_one * 1e3:.1f} ms/chunk; one batched launch {t_b * 1e3:.1f} ms total = {t_b / n_chunks * 1e3:.2f} ms/chunk"
    try:
        from oracle import pyoracle
        t0 = time.perf_counter(); ref = [pyoracle.ref_sdust(s, 5, 20) for s in seqs[:8]]; t_ref = (time.perf_counter() - t0) / 8
        assert all((a == b).all() for a, b in zip(ref, got[:8]))
        line += f"; reference sdust() on one core {t_ref * 1e3:.1f} ms/chunk"
    except Exception as e:  # noqa
        line += f" (reference sdust not available: {e})"
    print(line)


if __name__ == "__main__":
    main()
