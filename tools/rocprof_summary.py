#!/usr/bin/env python
"""Turn the rocprofv3 results of tools/profile_round.sh (rocpd sqlite .db files) into the summaries committed under profiles/.

usage: python tools/rocprof_summary.py <tag> "<title>"
  reads  gpurun_out/prof_<tag>/kt_results.db, gpurun_out/pmc_fetch_<tag>/f_results.db, gpurun_out/pmc_write_<tag>/w_results.db
  writes profiles/<tag>_bench_kernel_stats.txt, profiles/<tag>_pmc_hbm_traffic.txt, profiles/<tag>_traffic.json
"""
import json
import sqlite3
import sys


def kernel_stats(db_path, out_path, title):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w") as f:
        f.write(f"# {title}\n# source: rocprofv3 --kernel-trace --stats (durations in us)\n")
        f.write(f"{'kernel':70s} {'calls':>8s} {'total_us':>16s} {'avg_us':>14s} {'pct':>7s}\n")
        for name, calls, tot, avg, pct in rows:
            f.write(f"{name.split('(')[0][:70]:70s} {calls:8d} {tot:16.0f} {avg:14.1f} {pct:7.2f}\n")
        if any("lcd_gate_kernel" in r[0] for r in rows):
            f.write("# note: lcd_gate_kernel is ONE lane that sleeps until the wide workgroups are resident (launch ordering, lcd_host.cpp launch_poa_grouped);\n"
                    "#       its duration is waiting time on one wavefront slot and overlaps the chain kernels -- the percentages are of summed kernel time, not of wall time\n")
        try:
            for r in cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, scratch_size from kernels group by name"):
                f.write(f"# {r[0].split('(')[0]}: grid={r[1]} wg={r[2]} lds={r[3]} vgpr={r[4]} sgpr={r[5]} scratch={r[6]}\n")
        except Exception as e:  # noqa
            f.write(f"# (no dispatch detail: {e})\n")
    return rows


def pmc_summary(db_path, counter):
    """sum of a PMC counter (rocprofv3 --pmc X) per kernel over all its dispatches"""
    cur = sqlite3.connect(db_path).cursor()
    out = {}
    for name, val in cur.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        k = name.split("(")[0]
        out.setdefault(k, [0, 0.0])
        out[k][0] += 1
        out[k][1] += val
    return out


def main(tag, title):
    import os
    kernel_stats(f"gpurun_out/prof_{tag}/kt_results.db", f"profiles/{tag}_bench_kernel_stats.txt", title)
    for suffix, what in (("_ont", "configs[2] ONT shape: python bench.py --shape ont --steps 48"), ("_sv", "configs[4] SV shape: python bench.py --shape sv --steps 8 --coalesce 4")):
        if os.path.exists(f"gpurun_out/prof_{tag}{suffix}/kt_results.db"):
            kernel_stats(f"gpurun_out/prof_{tag}{suffix}/kt_results.db", f"profiles/{tag}{suffix}_bench_kernel_stats.txt", f"{title} -- {what}")
    f = pmc_summary(f"gpurun_out/pmc_fetch_{tag}/f_results.db", "FETCH_SIZE")
    w = pmc_summary(f"gpurun_out/pmc_write_{tag}/w_results.db", "WRITE_SIZE")
    poa = 0.0
    with open(f"profiles/{tag}_pmc_hbm_traffic.txt", "w") as o:
        o.write(f"# {title}\n# HBM traffic from PMC counters over ONE bench step (python bench.py --steps 1 --warmup 0 --lanes 1), separate passes:\n")
        o.write("#   rocprofv3 --pmc FETCH_SIZE ...   and   rocprofv3 --pmc WRITE_SIZE ...   (KB per dispatch, summed per kernel)\n")
        o.write("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of a wide coalesced read -> x2; WRITE_SIZE as reported\n")
        o.write(f"{'kernel':30s} {'dispatches':>10s} {'FETCH_KB':>14s} {'WRITE_KB':>14s} {'bytes=(2*F+W)*1024':>22s}\n")
        for k in sorted(set(f) | set(w)):
            fk = f.get(k, [0, 0.0]); wk = w.get(k, [0, 0.0])
            b = (2 * fk[1] + wk[1]) * 1024
            if "lcd_poa_chain_kernel" in k:
                poa += b
            o.write(f"{k[:30]:30s} {fk[0]:10d} {fk[1]:14.1f} {wk[1]:14.1f} {b:22.0f}\n")
        o.write(f"# lcd_poa_chain_kernel (all workgroup classes of one step): {poa:.0f} bytes\n")
    try:   # SQ block pass (LDS / issue counters), summed per kernel over the dispatches of the same single step
        cur = sqlite3.connect(f"gpurun_out/pmc_sq_{tag}/s_results.db").cursor()
        acc = {}
        for name, cn, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
            acc.setdefault(name.split("(")[0], {}).setdefault(cn, 0.0)
            acc[name.split("(")[0]][cn] += val
        with open(f"profiles/{tag}_pmc_sq_lds.txt", "w") as o:
            o.write(f"# {title}\n# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES\n")
            o.write("# over ONE bench step (python bench.py --steps 1 --warmup 0 --lanes 1 --coalesce 1); sums over all dispatches of a kernel\n")
            for k in sorted(acc):
                c = acc[k]
                o.write(k[:60] + "\n")
                for cn in sorted(c):
                    o.write(f"    {cn:24s} {c[cn]:18.0f}\n")
                if c.get("SQ_LDS_IDX_ACTIVE"):
                    o.write(f"    LDS bank-conflict cycles / LDS active cycles = {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.4f}\n")
                if c.get("SQ_WAVES"):
                    o.write(f"    VALU / SALU / LDS instructions per wavefront = {c.get('SQ_INSTS_VALU', 0) / c['SQ_WAVES']:.0f} / {c.get('SQ_INSTS_SALU', 0) / c['SQ_WAVES']:.0f} / {c.get('SQ_INSTS_LDS', 0) / c['SQ_WAVES']:.0f}\n")
        poa_valu = sum(c.get("SQ_INSTS_VALU", 0) for k, c in acc.items() if "lcd_poa_chain_kernel" in k)
        poa_salu = sum(c.get("SQ_INSTS_SALU", 0) for k, c in acc.items() if "lcd_poa_chain_kernel" in k)
        poa_lds = sum(c.get("SQ_INSTS_LDS", 0) for k, c in acc.items() if "lcd_poa_chain_kernel" in k)
        json.dump({"tag": tag, "kernel": "lcd_poa_chain_kernel", "valu_wave_insts_per_step": poa_valu, "salu_wave_insts_per_step": poa_salu,
                   "lds_wave_insts_per_step": poa_lds, "source": f"profiles/{tag}_pmc_sq_lds.txt (rocprofv3 --pmc SQ_INSTS_VALU ...)"},
                  open(f"profiles/{tag}_sq.json", "w"), indent=1)
    except Exception as e:  # noqa
        print("no SQ pass:", e)
    json.dump({"tag": tag, "kernel": "lcd_poa_chain_kernel", "hbm_bytes_per_step": poa,
               "source": f"profiles/{tag}_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2*FETCH+WRITE)"},
              open(f"profiles/{tag}_traffic.json", "w"), indent=1)
    print(open(f"profiles/{tag}_bench_kernel_stats.txt").read())
    print(open(f"profiles/{tag}_pmc_hbm_traffic.txt").read())


def timed_half(db_path, counters):
    """per kernel: sums of the counters over the dispatches of the TIMED submission only.  The driver's command runs the same number of launches twice -- one warm-up
    submission (other seeds) and the timed one -- so the timed launches are the second half of a kernel's dispatches in dispatch order."""
    cur = sqlite3.connect(db_path).cursor()
    per = {}
    for name, did, cn, val in cur.execute("select name, dispatch_id, counter_name, sum(counter_value) from pmc_events group by name, dispatch_id, counter_name"):
        per.setdefault(name.split("(")[0], {}).setdefault(did, {})[cn] = val
    out = {}
    for k, d in per.items():
        ids = sorted(d)
        half = ids[len(ids) // 2:]
        acc = {}
        for i in half:
            for cn, v in d[i].items():
                acc[cn] = acc.get(cn, 0.0) + v
        out[k] = (len(half), acc)
    return out


def main_driver(tag, title):
    """summaries of tools/profile_driver.sh: the driver's exact command, counters of the timed 20-batch submission"""
    kernel_stats(f"gpurun_out/prof_{tag}/kt_results.db", f"profiles/{tag}_bench_kernel_stats.txt", title + " (kernel trace: warm-up + timed submission)")
    f = timed_half(f"gpurun_out/pmc_fetch_{tag}/f_results.db", ["FETCH_SIZE"])
    w = timed_half(f"gpurun_out/pmc_write_{tag}/w_results.db", ["WRITE_SIZE"])
    poa = 0.0
    with open(f"profiles/{tag}_pmc_hbm_traffic.txt", "w") as o:
        o.write(f"# {title}\n# HBM traffic from PMC counters over the TIMED submission (20 batches) of the driver's command, separate passes:\n")
        o.write("#   rocprofv3 --pmc FETCH_SIZE -- python bench.py --gpus 1 --steps 20 --warmup 5   and the same with WRITE_SIZE   (KB per dispatch, summed per kernel over the\n")
        o.write("#   second half of its dispatches = the timed submission; the first half is the warm-up submission on other seeds)\n")
        o.write("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of a wide coalesced read -> x2; WRITE_SIZE as reported\n")
        o.write(f"{'kernel':30s} {'dispatches':>10s} {'FETCH_KB':>14s} {'WRITE_KB':>14s} {'bytes=(2*F+W)*1024':>22s}\n")
        for k in sorted(set(f) | set(w)):
            fk = f.get(k, (0, {})); wk = w.get(k, (0, {}))
            fv, wv = fk[1].get("FETCH_SIZE", 0.0), wk[1].get("WRITE_SIZE", 0.0)
            b = (2 * fv + wv) * 1024
            if "lcd_poa_chain_kernel" in k:
                poa += b
            o.write(f"{k[:30]:30s} {fk[0]:10d} {fv:14.1f} {wv:14.1f} {b:22.0f}\n")
        o.write(f"# lcd_poa_chain_kernel (all workgroup classes, the timed submission of 20 steps): {poa:.0f} bytes = {poa / 20:.0f} per step\n")
    json.dump({"tag": tag, "kernel": "lcd_poa_chain_kernel", "hbm_bytes_per_step": poa / 20, "hbm_bytes_per_submission": poa, "steps_per_submission": 20,
               "source": f"profiles/{tag}_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on the driver's command, timed submission, 2*FETCH+WRITE)"},
              open(f"profiles/{tag}_traffic.json", "w"), indent=1)
    sq = timed_half(f"gpurun_out/pmc_sq_{tag}/s_results.db", None)
    with open(f"profiles/{tag}_pmc_sq_lds.txt", "w") as o:
        o.write(f"# {title}\n# rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES\n")
        o.write("# on the driver's command (python bench.py --gpus 1 --steps 20 --warmup 5); sums over the dispatches of the TIMED submission (20 steps)\n")
        for k in sorted(sq):
            n, c = sq[k]
            o.write(f"{k[:60]}  ({n} dispatches)\n")
            for cn in sorted(c):
                o.write(f"    {cn:24s} {c[cn]:18.0f}\n")
            if c.get("SQ_LDS_IDX_ACTIVE"):
                o.write(f"    LDS bank-conflict cycles / LDS active cycles = {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.4f}\n")
            if c.get("SQ_WAVES"):
                o.write(f"    VALU / SALU / LDS instructions per wavefront = {c.get('SQ_INSTS_VALU', 0) / c['SQ_WAVES']:.0f} / {c.get('SQ_INSTS_SALU', 0) / c['SQ_WAVES']:.0f} / {c.get('SQ_INSTS_LDS', 0) / c['SQ_WAVES']:.0f}\n")
    tot = lambda cn: sum(c.get(cn, 0) for k, (n, c) in sq.items() if "lcd_poa_chain_kernel" in k)
    json.dump({"tag": tag, "kernel": "lcd_poa_chain_kernel", "valu_wave_insts_per_step": tot("SQ_INSTS_VALU") / 20, "salu_wave_insts_per_step": tot("SQ_INSTS_SALU") / 20,
               "lds_wave_insts_per_step": tot("SQ_INSTS_LDS") / 20, "steps_per_submission": 20,
               "source": f"profiles/{tag}_pmc_sq_lds.txt (rocprofv3 --pmc SQ_INSTS_VALU ... on the driver's command, timed submission)"},
              open(f"profiles/{tag}_sq.json", "w"), indent=1)
    st = timed_half(f"gpurun_out/pmc_stall_{tag}/s_results.db", None)
    with open(f"profiles/pmc_stall_{tag}.txt", "w") as o:
        o.write(f"# {title}\n# where the wavefronts spend their cycles (fractions of SQ_WAVE_CYCLES; the driver's command, timed submission)\n")
        for k in sorted(st):
            n, d = st[k]
            if "lcd_" not in k:
                continue
            wc = d.get("SQ_WAVE_CYCLES", 0) or 1
            o.write(f"{k:40s} " + " ".join(f"{cn[3:]}={v / wc:.3f}" for cn, v in sorted(d.items()) if cn != "SQ_WAVE_CYCLES") + f" WAVE_CYCLES={wc:.3g}\n")
    for fn in (f"profiles/{tag}_bench_kernel_stats.txt", f"profiles/{tag}_pmc_hbm_traffic.txt", f"profiles/pmc_stall_{tag}.txt"):
        print(open(fn).read())
    print(open(f"profiles/{tag}_sq.json").read())


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "driver":
        main_driver(sys.argv[1], sys.argv[2])
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 summary")
