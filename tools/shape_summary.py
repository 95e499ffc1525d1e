#!/usr/bin/env python
"""summaries of tools/profile_shapes.sh: per shape (ont, sv) kernel stats and PMC HBM traffic of the timed submission(s) -> profiles/<tag>_<shape>_bench_kernel_stats.txt,
profiles/<tag>_<shape>_pmc_hbm_traffic.txt, profiles/<tag>_<shape>_traffic.json.   usage: python tools/shape_summary.py <tag> "<title>" """
import json
import sys
sys.path.insert(0, "tools")
import sqlite3
from rocprof_summary import kernel_stats  # noqa: E402


def all_dispatches(db_path, counter):
    """per kernel: (dispatches, sum of the counter) over EVERY dispatch of the run"""
    cur = sqlite3.connect(db_path).cursor()
    out = {}
    for name, did, val in cur.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name = ? group by name, dispatch_id", (counter,)):
        k = name.split("(")[0]
        n, v = out.get(k, (0, 0.0))
        out[k] = (n + 1, v + val)
    return out


def steps_run(log):
    """warm-up + timed steps of the bench line in a rocprofv3 log"""
    for line in open(log, errors="replace"):
        if line.startswith("{") and '"metric"' in line:
            j = json.loads(line)
            return int(j.get("warmup_steps_run", 0)) + int(j["steps"]), j
    raise RuntimeError("no bench line in " + log)


def main(tag, title):
    for sh, steps, what in (("ont", 48, "configs[2] ONT shape: python bench.py --shape ont --steps 48 --warmup 2"), ("sv", 8, "configs[4] SV shape: python bench.py --shape sv --steps 8 --warmup 2")):
        t = f"{title} -- {what}"
        try:
            kernel_stats(f"gpurun_out/prof_{tag}_{sh}/kt_results.db", f"profiles/{tag}_{sh}_bench_kernel_stats.txt", t + " (kernel trace: warm-up + timed submissions)")
            f = all_dispatches(f"gpurun_out/pmc_fetch_{tag}_{sh}/f_results.db", "FETCH_SIZE")
            w = all_dispatches(f"gpurun_out/pmc_write_{tag}_{sh}/w_results.db", "WRITE_SIZE")
            n_steps, line = steps_run(f"gpurun_out/pmc_fetch_{tag}_{sh}.log")
        except Exception as e:  # noqa
            print(sh, "missing:", e)
            continue
        poa = 0.0
        with open(f"profiles/{tag}_{sh}_pmc_hbm_traffic.txt", "w") as o:
            o.write(f"# {t}\n# HBM traffic from PMC counters (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes), KB per dispatch summed per kernel over EVERY\n"
                    f"# dispatch of the run: {n_steps} steps of the same workload (warm-up on other seeds + the timed steps) -- per step = sum / {n_steps}\n"
                    "# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x 2; WRITE_SIZE as reported\n")
            o.write(f"{'kernel':30s} {'dispatches':>10s} {'FETCH_KB':>14s} {'WRITE_KB':>14s} {'bytes=(2*F+W)*1024':>22s}\n")
            for k in sorted(set(f) | set(w)):
                fk = f.get(k, (0, 0.0)); wk = w.get(k, (0, 0.0))
                bts = (2 * fk[1] + wk[1]) * 1024
                if "lcd_poa_chain_kernel" in k:
                    poa += bts
                o.write(f"{k[:30]:30s} {fk[0]:10d} {fk[1]:14.1f} {wk[1]:14.1f} {bts:22.0f}\n")
            o.write(f"# lcd_poa_chain_kernel (all workgroup classes): {poa:.0f} bytes over {n_steps} steps = {poa / n_steps:.0f} per step\n")
        json.dump({"tag": tag, "shape": sh, "kernel": "lcd_poa_chain_kernel", "hbm_bytes_per_step": poa / n_steps, "steps_in_the_run": n_steps,
                   "source": f"profiles/{tag}_{sh}_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the shape's builder command, every dispatch, 2*FETCH+WRITE)"},
                  open(f"profiles/{tag}_{sh}_traffic.json", "w"), indent=1)
        print(open(f"profiles/{tag}_{sh}_bench_kernel_stats.txt").read())
        print(open(f"profiles/{tag}_{sh}_pmc_hbm_traffic.txt").read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
