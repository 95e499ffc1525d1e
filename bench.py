#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its configs[1]: regions/sec (+ POA-aligned bases/sec) of the per-region hot path
on synthetic 30x HiFi-shape region jobs (15 kb reads, 0.1 % error, 10 Mb reference => 1 250 regions per GPU).

One "step" = one pass of the hot path (anchors -> POA chains -> ref/cons WFA -> MSA strings) over the rank's batch with
inputs already resident in HBM.  N > 1: regions are sharded across ranks as independent work items (no data-path
collective); torch.distributed (RCCL) is used for the barrier and the max-over-ranks time only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The library spreads one submission's launch groups over 4 streams (LCD_STREAMS).  Keep the number of hardware queues equal to that:
# with more queues holding runnable kernels the queue scheduler time-slices them (measured: the same chains run 2.6x slower next to
# 8 other active queues); concurrency comes from coalescing batches into one submission (--coalesce), not from more queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def _cpu_worker(a):
    """cpu_baseline leg: one process per host core runs the oracle (oracle/, test infrastructure) on its slice of the workload"""
    seed, n_regions, shape_name, w, n_workers, per, budget_s = a
    from longcalld_amd import jobs
    from oracle import pyoracle
    n_gen = min(n_regions, per)   # (generating the whole workload in every worker would cost more than the timed loop)
    regs = jobs.make_regions(seed + 7919 * w, n_gen, jobs.HIFI if shape_name == "hifi" else jobs.ONT)
    done = 0
    t0 = time.perf_counter()
    while done < per and time.perf_counter() - t0 < budget_s:   # bounded by a time budget: the core count of the box is not known in advance
        pyoracle.collect_noisy_reg_aln_strs(regs[done % n_gen])
        done += 1
    return done, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ref-mb", type=float, default=10.0, help="synthetic reference size per GPU (10 Mb = configs[1])")
    ap.add_argument("--shape", default="hifi", choices=["hifi", "ont"])
    ap.add_argument("--cpu-sample", type=int, default=200, help="cap on regions per CPU worker of the cpu_baseline leg, which is bounded to ~12 s (rank 0, N=1 only; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU worker processes of the cpu_baseline leg (0 = all host cores)")
    ap.add_argument("--seed", type=int, default=20250928)
    ap.add_argument("--lanes", type=int, default=0,
                    help="concurrent submission lanes per GPU (host threads, one lcd_batch_t + HIP stream each) -- the reference's own "
                         "execution model: kt_for runs n_threads chunk workers concurrently (src/call_var_main.c:773); the K steps are "
                         "dealt round-robin to the lanes and ALL of them complete inside the timed region")
    ap.add_argument("--vars", type=int, default=0, choices=[0, 1, 2],
                    help="SURVEY 8(f) f1: also run the candidate-variant stage S6 inside the timed step (1), and leave the alignment strings "
                         "in HBM at download (2); 0 = the headline path exactly as collect_noisy_reg_aln_strs defines it")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="steps (batches) submitted together through lcd_batch_run_many: one set of launches per stage over the chains of "
                         "all of them, so that the GPU's workgroup dispatcher -- not HIP streams -- packs several chunks' chains onto the CUs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    from longcalld_amd import align, jobs, _lib
    lib = _lib.load_library()
    _lib.check(lib.lcd_init(local_rank), lib)

    shape = jobs.HIFI if args.shape == "hifi" else jobs.ONT
    n_regions = jobs.regions_for_ref_mb(args.ref_mb)
    regs = jobs.make_regions(args.seed + 1000 * rank, n_regions, shape)   # weak scaling: same work per GPU, different seed
    bench_opt = align.default_opt()
    bench_opt.collect_noisy_vars = args.vars
    bench_opt.is_ont = 1 if args.shape == "ont" else 0   # the reference's --ont / --hifi switch (src/call_var_main.h:128)
    import threading
    # a step = one batch (the configs[1] workload).  `coalesce` steps are submitted together through lcd_batch_run_many (one set of
    # launches per stage over all their chains), `lanes` host threads keep that many such submissions in flight.
    # defaults (measured): HiFi shape = two lanes of 32 batches per submission when there are enough steps for each lane to
    # have two submissions (one lane's anchor / WFA / string stages and the tail of its chain launches overlap the other lane's chains:
    # 42 500 regions/s against 38 000 for one lane of 32; 2 x 32 batches hold ~180 GB of the 284 GB arena budget; three lanes lose: 38 600),
    # else one lane of up to 32; the ONT shape (noisy reads: 4x graph / WFA estimates, ~8 GB per batch in flight) runs one lane of 24:
    # 13 900 regions/s at 191 GB (16: 13 000; 32: 15 200 but 254 GB of the 284 GB budget with the retry rounds; two lanes of 16: 14 900)
    if args.lanes <= 0:
        args.lanes = 2 if (args.shape == "hifi" and args.coalesce <= 0 and args.steps >= 128) else 1
    if args.coalesce <= 0:
        args.coalesce = 32 if args.shape == "hifi" else 24
    n_co = max(1, min(args.coalesce, args.steps))
    n_lanes = max(1, min(args.lanes, (args.steps + n_co - 1) // n_co))
    groups = []
    t_up = 0.0
    for _ in range(n_lanes):
        grp = []
        for _ in range(n_co):
            bt = align.RegionBatch(bench_opt)
            for r in regs:
                bt.add_region(r)
            t_up0 = time.perf_counter()
            bt.upload()
            t_up = time.perf_counter() - t_up0
            grp.append(bt)
        groups.append(grp)
    batches = [bt for grp in groups for bt in grp]
    batch = batches[0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    acc = {"ms": 0.0, "launches": 0, "st": None, "alg": 0.0, "cells": 0.0}
    lock = threading.Lock()

    lane_errors = []

    def lane_work(grp, sizes, record):
        try:
            _lane_work(grp, sizes, record)
        except Exception as e:  # noqa: surfaced after the join (a thread's traceback alone would leave a half-measured line)
            lane_errors.append(e)

    def _lane_work(grp, sizes, record):
        for k in sizes:
            align.RegionBatch.run_many(grp[:k])
            if record:
                sts = [bt.stats() for bt in grp[:k]]
                with lock:
                    acc["ms"] += sts[0]["ms_poa_kernel"]; acc["launches"] += sts[0]["n_poa_launches"]; acc["st"] = sts[0]
                    acc["alg"] += sum(float(x["poa_alg_bytes"]) for x in sts); acc["cells"] += sum(float(x["poa_cells"]) for x in sts)

    def run_steps(n_steps, record):
        # cut the steps into submissions of `coalesce` and deal those round-robin to the lanes; lanes overlap on the GPU
        subs = [n_co] * (n_steps // n_co) + ([n_steps % n_co] if n_steps % n_co else [])
        share = [subs[i::n_lanes] for i in range(n_lanes)]
        ths = [threading.Thread(target=lane_work, args=(groups[i], share[i], record)) for i in range(n_lanes) if share[i]]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if lane_errors:
            raise lane_errors[0]

    # every lane warms its buffers at least once; the noisy-read shape needs three submissions: the capacity hints (graph, DP region, WFA score
    # bound) rise one level per submission that overflowed, and a timed run that still re-runs overflowed chains measures the learning, not the path
    n_warm = max(args.warmup, (n_lanes * n_co * (3 if args.shape == "ont" else 1)) if args.warmup else 0)
    run_steps(n_warm, False)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    poa_kernel_ms, poa_launches, st = acc["ms"], acc["launches"], acc["st"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([float(st["n_regions"]), float(st["poa_aligned_bases"])], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_regions, tot_bases = float(cnt[0].item()), float(cnt[1].item())
    else:
        tot_regions, tot_bases = float(st["n_regions"]), float(st["poa_aligned_bases"])

    # PCIe-inclusive figure for DESIGN.md (never `value`)
    t_dl0 = time.perf_counter()
    batch.download()
    digest = batch.digest()
    t_dl = time.perf_counter() - t_dl0

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = tot_regions * args.steps / elapsed
        # roofline of the dominant kernel (POA chains): algorithmic bytes (SURVEY 8d B_poa) per launch / mean launch time
        alg_bytes = acc["alg"] / max(poa_launches, 1)     # per launch set: the chains of `coalesce` batches
        mean_launch_s = poa_kernel_ms / max(poa_launches, 1) * 1e-3
        achieved = alg_bytes / mean_launch_s / 1e9 if mean_launch_s > 0 else 0.0
        # HBM bytes of the same launch from the PMC counters: they need rocprofv3 (separate --pmc passes, tools/profile_round.sh), so the
        # figure is the committed measurement of this exact workload (profiles/<tag>_traffic.json), or null for any other workload
        traffic, traffic_src = None, None
        if args.ref_mb == 10 and shape["name"] == "hifi" and world == 1:
            import glob
            import re
            cand = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")),
                          key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])   # r01_v9 < r01_v10
            if cand:
                tj = json.load(open(cand[-1]))
                traffic, traffic_src = float(tj["hbm_bytes_per_step"]) * n_co, tj["source"] + f" x {n_co} coalesced steps"
        # secondary roofline: integer VALU issue.  Wavefront instructions of the POA kernels per step from the committed SQ counter
        # pass (same rule as `traffic`), x 64 lanes, over the live launch time; peak = 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 Tops/s
        valu = None
        if traffic is not None:
            cand2 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_sq.json")),
                           key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
            if cand2 and mean_launch_s > 0:
                sj = json.load(open(cand2[-1]))
                ops = float(sj["valu_wave_insts_per_step"]) * 64 * n_co
                valu = {"achieved": round(ops / mean_launch_s / 1e12, 3), "peak": 78.6, "unit": "Tops/s (int32 lane-ops)", "frac": round(ops / mean_launch_s / 78.6e12, 4),
                        "source": sj["source"]}
                if n_lanes > 1 and world == 1:   # chip-wide rate over the wall clock of the timed region (see roofline.achieved_wall)
                    ops_all = float(sj["valu_wave_insts_per_step"]) * 64 * args.steps
                    valu["achieved_wall"] = round(ops_all / elapsed / 1e12, 3); valu["frac_wall"] = round(ops_all / elapsed / 78.6e12, 4)
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": "lcd_poa_chain_kernel", "ms_per_launch": round(mean_launch_s * 1e3, 4),
                    "alg_bytes_per_launch": alg_bytes, "cells_per_launch": int(acc["cells"] / max(poa_launches, 1)),
                    "gcups": round(acc["cells"] / max(poa_launches, 1) / mean_launch_s / 1e9, 3) if mean_launch_s > 0 else 0.0,
                    "valu": valu}
        if n_lanes > 1 and world == 1:
            # `achieved` prices ONE lane's launch set against its own duration while the other lanes' chains share the CUs; the chip-wide rate is
            # all algorithmic bytes of the timed region over its wall clock (conservative: the non-POA stages are inside that time)
            roofline["concurrent_launch_sets"] = n_lanes
            roofline["achieved_wall"] = round(acc["alg"] / elapsed / 1e9, 3); roofline["frac_wall"] = round(acc["alg"] / elapsed / 1e9 / HBM_PEAK_GBS, 6)
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            # the reference runs this path on kt_for worker threads (src/call_var_main.c:773): time the CPU port the same way, one
            # process per host core (spawned: no fork after HIP start-up), every worker on its own slice of the same regions
            import multiprocessing as mp
            try:
                n_avail = len(os.sched_getaffinity(0))
            except Exception:  # noqa
                n_avail = os.cpu_count() or 1
            try:  # a container's CPU quota (cgroup v2 cpu.max = "quota period") is the real core count
                q, per_us = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if q != "max":
                    n_avail = max(1, min(n_avail, -(-int(q) // int(per_us))))
            except Exception:  # noqa
                pass
            n_workers = args.cpu_threads if args.cpu_threads > 0 else n_avail
            per = max(1, args.cpu_sample)
            ctx = mp.get_context("spawn")
            # one worker alone first (the per-core rate), then all workers at once (what the node delivers: SMT siblings and cgroup CPU
            # quotas make that less than cores x per-core rate)
            one = _cpu_worker((args.seed, n_regions, args.shape, 0, 1, min(per, 120), 6.0))
            c0 = time.perf_counter()
            with ctx.Pool(n_workers) as pool:
                parts = pool.map(_cpu_worker, [(args.seed, n_regions, args.shape, w, n_workers, per, 12.0) for w in range(n_workers)], chunksize=1)
            wall = time.perf_counter() - c0
            done = sum(p_[0] for p_ in parts); busy = max(p_[1] for p_ in parts)
            cpu = {"value": round(done / busy, 2), "unit": "regions/s", "cores": n_workers, "kind": "port",
                   "sample": f"{done} regions of the same workload shape in {busy:.1f} s on {n_workers} worker processes (one per schedulable CPU / cgroup quota; "
                             f"{wall:.1f} s with process start-up), oracle/ C restatement (-O3 scalar, not upstream SIMD abPOA/WFA2: those submodules are "
                             f"absent); one worker alone: {one[0] / one[1]:.1f} regions/s"}
        out = {
            "metric": "regions_per_sec", "value": round(value, 2), "unit": "regions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": n_warm,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic 30x {shape['name']} region jobs over {args.ref_mb:g} Mb reference per GPU "
                                   f"({n_regions} regions/GPU, ~{tot_bases / max(world, 1) / 1e6:.1f} Mbase POA-aligned/GPU)",
                       "regions_per_gpu": n_regions, "sharding": "regions sharded across ranks, no data-path collective",
                       "lanes_per_gpu": n_lanes, "coalesced_steps_per_submission": n_co},
            "poa_aligned_bases_per_sec": round(tot_bases * args.steps / elapsed, 1),
            "regions_resolved": int(st["n_regions_resolved"]),
            "wfa_offsets_per_sec": round(float(st["wfa_offsets"]) * n_co / max((st["ms_wfa"] + st["ms_anchor"]) * 1e-3, 1e-9), 1),
            "edlib_blocks_per_sec": round(float(st["edlib_blocks"]) * n_co / max(st["ms_anchor"] * 1e-3, 1e-9), 1),
            "stage_ms": {k: round(st[k], 3) for k in ("ms_anchor", "ms_poa", "ms_wfa", "ms_strings", "ms_vars", "ms_total", "ms_host", "ms_poa_kernel")},
            "noisy_vars_stage": args.vars,
            "pcie_inclusive": {"upload_s": round(t_up, 4), "download_and_digest_s": round(t_dl, 4),
                               "regions_per_sec": round(tot_regions / max(world, 1) / (t_up + ms_step / 1e3 + t_dl), 2)},
            "digest": f"{digest:016x}",
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    for bt in batches:
        bt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
