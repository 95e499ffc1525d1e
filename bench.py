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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ref-mb", type=float, default=10.0, help="synthetic reference size per GPU (10 Mb = configs[1])")
    ap.add_argument("--shape", default="hifi", choices=["hifi", "ont"])
    ap.add_argument("--cpu-sample", type=int, default=300, help="regions timed on the CPU oracle for cpu_baseline (rank 0, N=1 only)")
    ap.add_argument("--seed", type=int, default=20250928)
    ap.add_argument("--lanes", type=int, default=1,
                    help="concurrent submission lanes per GPU (host threads, one lcd_batch_t + HIP stream each) -- the reference's own "
                         "execution model: kt_for runs n_threads chunk workers concurrently (src/call_var_main.c:773); the K steps are "
                         "dealt round-robin to the lanes and ALL of them complete inside the timed region")
    ap.add_argument("--coalesce", type=int, default=16,
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
    import threading
    # a step = one batch (the configs[1] workload).  `coalesce` steps are submitted together through lcd_batch_run_many (one set of
    # launches per stage over all their chains), `lanes` host threads keep that many such submissions in flight.
    n_co = max(1, min(args.coalesce, args.steps))
    n_lanes = max(1, min(args.lanes, (args.steps + n_co - 1) // n_co))
    groups = []
    t_up = 0.0
    for _ in range(n_lanes):
        grp = []
        for _ in range(n_co):
            bt = align.RegionBatch()
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

    run_steps(max(args.warmup, n_lanes * n_co if args.warmup else 0), False)   # every lane warms its buffers at least once
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
            cand = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*_traffic.json")))
            if cand:
                tj = json.load(open(cand[-1]))
                traffic, traffic_src = float(tj["hbm_bytes_per_step"]) * n_co, tj["source"] + f" x {n_co} coalesced steps"
        roofline = {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": "lcd_poa_chain_kernel", "ms_per_launch": round(mean_launch_s * 1e3, 4),
                    "alg_bytes_per_launch": alg_bytes, "cells_per_launch": int(acc["cells"] / max(poa_launches, 1)),
                    "gcups": round(acc["cells"] / max(poa_launches, 1) / mean_launch_s / 1e9, 3) if mean_launch_s > 0 else 0.0}
        cpu = None
        if world == 1 and args.cpu_sample > 0:
            from oracle import pyoracle
            sample = regs[: min(args.cpu_sample, len(regs))]
            c0 = time.perf_counter()
            for r in sample:
                pyoracle.collect_noisy_reg_aln_strs(r)
            c = time.perf_counter() - c0
            cpu = {"value": round(len(sample) / c, 2), "unit": "regions/s", "cores": 1, "kind": "port",
                   "sample": f"first {len(sample)} of the {len(regs)} regions of the same workload, oracle/ C restatement (-O3 scalar, not upstream SIMD abPOA/WFA2: "
                             f"those submodules are absent), {c:.1f} s on one host core of {os.cpu_count()}"}
        out = {
            "metric": "regions_per_sec", "value": round(value, 2), "unit": "regions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: synthetic 30x {shape['name']} region jobs over {args.ref_mb:g} Mb reference per GPU "
                                   f"({n_regions} regions/GPU, ~{tot_bases / max(world, 1) / 1e6:.1f} Mbase POA-aligned/GPU)",
                       "regions_per_gpu": n_regions, "sharding": "regions sharded across ranks, no data-path collective",
                       "lanes_per_gpu": n_lanes, "coalesced_steps_per_submission": n_co},
            "poa_aligned_bases_per_sec": round(tot_bases * args.steps / elapsed, 1),
            "regions_resolved": int(st["n_regions_resolved"]),
            "stage_ms": {k: round(st[k], 3) for k in ("ms_anchor", "ms_poa", "ms_wfa", "ms_strings", "ms_total", "ms_host", "ms_poa_kernel")},
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
