#!/usr/bin/env python
"""HBM account of the POA chain kernels per array group.

usage: python tools/hbm_account.py <bs.out> <steps> <out.txt> [<traffic.json>] [title]
  <bs.out>: stdout of a bench run on the -DLCD_X_BYTESTAT build (tools/ab_build.sh bs -DLCD_X_BYTESTAT): one "[bs] ..." line per chain with the bytes its
  memory instructions asked for, by array group (elements x element size per lane, from the trip counts the phases ran with -- poa_kernel.hip LCD_BS).
  <traffic.json>: profiles/<tag>_traffic.json of tools/rocprof_summary.py (PMC FETCH_SIZE / WRITE_SIZE of the same command on the product build).
"""
import json
import sys
from collections import defaultdict

NAMES = ["direction codes written (1 B / cell)", "row metadata written (rbeg, rend, roff)", "plan arrays read by the rows", "read bases staged",
         "backtrack reads (codes, row metadata, order)", "path (cigar) written + read back", "graph update (node / edge / read-set arrays, path scratch)",
         "plan build: graph arrays read", "plan build: plan arrays written", "full re-sort", "incremental re-sort", "remain by pointer jumping",
         "chain output (MSA rows, consensus)", "certified-band node arrays + intervals", "generic rows (int32 planes)", "per-row extremes (partial-cover reads)"]


def main():
    src, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    traffic = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and sys.argv[4] not in ("", "-") else None
    title = sys.argv[5] if len(sys.argv) > 5 else ""
    tot = [0] * 16
    by = defaultdict(lambda: [0] * 19)  # chains, reads, cells, 16 groups
    n = 0
    # the -DLCD_X_BYTESTAT build prints the process's running totals by (threads, kind) whenever a launch has finished, closed by "[bs-end]": the LAST complete block counts
    blocks, cur = [], []
    for line in open(src, errors="replace"):
        if line.startswith("[bs-end]"):
            if cur:
                blocks.append(cur)
            cur = []
        elif line.startswith("[bs] "):
            f = line.split()
            if any(g[1:4] == f[1:4] for g in cur):   # (a file without the end marks: a class that comes again opens the next block)
                blocks.append(cur); cur = []
            cur.append(f)
    if cur:
        blocks.append(cur)
    best = max(blocks, key=lambda b: sum(int(f[4]) for f in b)) if blocks else []
    for f in best:
        nt, mode, cert, chains, reads, cells = int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5]), int(f[6])
        v = [int(x) for x in f[8:24]]
        k = (nt, "K1" if mode == 0 else ("K2 certified band" if cert else "K2 full rows"))
        b = by[k]
        b[0] += chains; b[1] += reads; b[2] += cells
        for i in range(16):
            b[3 + i] += v[i]; tot[i] += v[i]
        n += chains
    s = sum(tot)
    with open(out, "w") as o:
        o.write(f"# {title}\n")
        o.write("# HBM account of lcd_poa_chain_kernel by array group: bytes the wavefronts' memory instructions ask for (elements x element size per lane; NOT sectors or\n"
                "# cache lines, and without the private-segment traffic of spilled registers), summed over every chain of the submission -- software counters of the\n"
                "# -DLCD_X_BYTESTAT build (poa_kernel.hip LCD_BS: per-phase formulas evaluated with the trip counts the phases ran with; the per-element constants are in the\n"
                "# comments beside each LCD_BS).  LDS traffic is not counted (ring, query cache, re-sort staging).\n")
        o.write(f"# {n} chains, {steps} steps\n")
        o.write(f"{'array group':62s} {'GB / step':>10s} {'share':>7s}\n")
        for i in range(16):
            if tot[i]:
                o.write(f"{NAMES[i]:62s} {tot[i] / steps / 1e9:10.3f} {100.0 * tot[i] / s:6.1f}%\n")
        o.write(f"{'sum (requested bytes)':62s} {s / steps / 1e9:10.3f}\n")
        dp = tot[0] + tot[1] + tot[2] + tot[3]
        o.write(f"# rows (codes, row metadata, plan, read bases): {100.0 * dp / s:.1f} %; per-read graph phases (backtrack .. output): {100.0 * (s - dp) / s:.1f} %\n")
        o.write("\n# by class (threads, kind): chains, reads, GB / step requested, largest groups\n")
        for k in sorted(by):
            b = by[k]
            gs = sorted(range(16), key=lambda i: -b[3 + i])[:4]
            o.write(f"#   {k[0]:5d} {k[1]:18s}: {b[0]:6d} chains {b[1]:7d} reads  {sum(b[3:]) / steps / 1e9:7.3f} GB/step  cells {b[2] / steps / 1e6:8.1f} M/step   "
                    + "; ".join(f"{NAMES[i].split(' (')[0].split(':')[0]} {100.0 * b[3 + i] / max(1, sum(b[3:])):.0f}%" for i in gs) + "\n")
        if traffic:
            per_step = traffic.get("hbm_bytes_per_step")
            if per_step:
                o.write(f"\n# PMC (2 x FETCH_SIZE + WRITE_SIZE, separate passes, product build, same command): {per_step / 1e9:.2f} GB / step for the chain kernels\n")
                o.write(f"# PMC / requested = {per_step / (s / steps):.2f}: what 32-byte sectors and 128-byte lines add to gathers of 1 - 4 byte elements (node -> edge -> node chains\n"
                        "# touch one element per line), plus the private-segment (scratch) traffic of the per-read control functions, which no software counter sees\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
