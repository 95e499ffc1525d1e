#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite) into the per-kernel summary text committed under profiles/."""
import sqlite3
import sys


def main(db_path, out_path, title):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats (durations in us)\n")
        f.write(f"{'kernel':70s} {'calls':>8s} {'total_us':>16s} {'avg_us':>14s} {'pct':>7s}\n")
        for name, calls, tot, avg, pct in rows:
            f.write(f"{name.split('(')[0][:70]:70s} {calls:8d} {tot:16.0f} {avg:14.1f} {pct:7.2f}\n")
        try:
            for r in cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels group by name"):
                f.write(f"# {r[0].split('(')[0]}: grid={r[1]} wg={r[2]} lds={r[3]} vgpr={r[4]} sgpr={r[5]} scratch={r[6]}\n")
        except Exception as e:  # noqa
            f.write(f"# (no dispatch detail: {e})\n")
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rocprofv3 kernel stats")


def pmc_summary(db_path, counter):
    """sum of a PMC counter (rocprofv3 --pmc X) per kernel over all its dispatches"""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    out = {}
    for name, val in cur.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        k = name.split("(")[0]
        out.setdefault(k, [0, 0.0])
        out[k][0] += 1
        out[k][1] += val
    return out
