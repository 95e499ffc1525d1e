#!/usr/bin/env python
"""Cost-model study behind DESIGN 6: how unbalanced are contiguous blocks of chunks, what do LPT and one rebalance epoch reach (no GPU needed).
usage: python tools/shard_balance.py [shape=sv] [job_mb=100] [ranks=8]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longcalld_amd import align, jobs, rebalance as rb  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "sv"
    job_mb = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    ranks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    n_chunks = int(round(job_mb / 0.5))
    cost = []
    for i in range(n_chunks):
        regs = jobs.make_regions(20250928 + i, jobs.regions_for_ref_mb(0.5), jobs.SHAPES[shape], poisson_sv=True)
        cost.append(sum(rb.region_cost(r) for r in regs))
    cost = np.array(cost)
    blocks = [list(cost[r * n_chunks // ranks:(r + 1) * n_chunks // ranks]) for r in range(ranks)]
    contig = np.array([sum(b) for b in blocks])
    _, lpt = align.lpt_assign(cost, ranks)
    moves, before, after = rb.plan_moves(blocks, tol=0.02)
    mean = cost.sum() / ranks
    print(f"{shape} shape, {job_mb:g} Mb = {n_chunks} chunks on {ranks} ranks; chunk cost max/median = {cost.max() / np.median(cost):.1f}")
    print(f"  contiguous blocks : max/mean = {contig.max() / mean:.3f}")
    print(f"  LPT over chunks   : max/mean = {lpt.max() / mean:.3f}")
    print(f"  one rebalance epoch from contiguous blocks: {len(moves)} chunks moved, max/mean = {max(after) / mean:.3f}")
    per10 = [cost[i:i + 20].sum() for i in range(0, n_chunks, 20)]
    _, lpt10 = align.lpt_assign(per10, ranks)
    print(f"  LPT over {len(per10)} batches of 10 Mb: max/mean = {lpt10.max() / mean:.3f}")


if __name__ == "__main__":
    main()
