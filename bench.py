#!/usr/bin/env python
"""bench.py -- BASELINE.json's metric on its configs[1]: regions/sec (+ POA-aligned bases/sec) of the per-region hot path
on synthetic 30x HiFi-shape region jobs (15 kb reads, 0.1 % error, 10 Mb reference => 1 250 regions per step).

One "step" = one pass of the hot path (anchors -> POA chains -> ref/cons WFA -> MSA strings) over one batch (the regions of 10 Mb of
reference) with inputs already resident in HBM.  The timed steps cycle over `--distinct` batches generated from DIFFERENT seeds (default 10 =
100 Mb of distinct regions); the warm-up runs on OTHER seeds, so nothing the library learns (capacity hints) is learned on the timed data.
N > 1: one process per GPU, each rank its own seeds (weak scaling), no data-path collective; torch.distributed (RCCL) carries the
barrier and the max-over-ranks time.  `--job-mb M` instead times ONE job of M Mb (M / 10 distinct batches) sharded over the ranks
(configs[3] / configs[4]: strong scaling; longest-processing-time assignment by the batches' read bases, optional RCCL queue rebalance).
Other shapes: --shape ont (configs[2]), --shape sv (configs[4]: 60x noisy reads, 1-10 kb INS/DEL).
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


def _gen(a):
    seed, n_regions, shape_name = a[:3]
    from longcalld_amd import jobs
    return jobs.make_regions(seed, n_regions, jobs.SHAPES[shape_name], poisson_sv=len(a) > 3 and a[3])


def _cpu_worker(a):
    """cpu_baseline leg: one process per host core runs the oracle (oracle/, test infrastructure) on its slice of the workload"""
    seed, n_regions, shape_name, w, n_workers, per, budget_s = a
    from longcalld_amd import jobs
    from oracle import pyoracle
    n_gen = min(n_regions, per)   # (generating the whole workload in every worker would cost more than the timed loop)
    regs = jobs.make_regions(seed + 7919 * w, n_gen, jobs.SHAPES[shape_name])
    done = 0
    t0 = time.perf_counter()
    while done < per and time.perf_counter() - t0 < budget_s:   # bounded by a time budget: the core count of the box is not known in advance
        pyoracle.collect_noisy_reg_aln_strs(regs[done % n_gen])
        done += 1
    return done, time.perf_counter() - t0


def _host_cores():
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
    return n_avail


def _k4_reference(pairs, n_threads, budget_s=6.0):
    """the reference's OWN edlib (oracle/_ref/libedlib_ref.so = /root/reference/edlib/src/edlib.cpp compiled where it lies) on the K4 job set
    of one step, `n_threads` host threads (ctypes releases the GIL); returns (pairs per second, seconds, pairs done) or None if the .so is absent"""
    import ctypes as C
    import threading
    so = os.path.join(ROOT, "oracle", "_ref", "libedlib_ref.so")
    if not os.path.exists(so) or not pairs:
        return None
    lib = C.CDLL(so)
    u8p = C.POINTER(C.c_uint8)
    lib.ref_edlib_nw_path.argtypes = [u8p, C.c_int, u8p, C.c_int, u8p, C.c_int, C.POINTER(C.c_int)]
    done = [0] * n_threads
    t0 = time.perf_counter()

    def work(w):
        cap = 1 + 2 * max(max(len(t), len(q)) for t, q in pairs)
        buf = (C.c_uint8 * cap)(); al = C.c_int(0)
        i = w
        while time.perf_counter() - t0 < budget_s:
            t, q = pairs[i % len(pairs)]
            lib.ref_edlib_nw_path(q.ctypes.data_as(u8p), len(q), t.ctypes.data_as(u8p), len(t), buf, cap, C.byref(al))   # edlib_xgaps: NW + path (src/align.c:222)
            done[w] += 1; i += n_threads
            if done[w] * n_threads >= 4 * len(pairs):
                break
    ths = [threading.Thread(target=work, args=(w,)) for w in range(n_threads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    el = time.perf_counter() - t0
    return sum(done) / el, el, sum(done)


def job_block(n_chunks, rank, world):
    """--job-mb: the contiguous block of chunks (genome order, src/call_var_main.c:599-621) rank `rank` of `world` generates: [lo, hi)"""
    return rank * n_chunks // world, (rank + 1) * n_chunks // world


def job_batches(costs, per):
    """--job-mb: this rank's chunks (by index into its queue) heaviest first, `per` of them per batch (= one step); returns the list of index lists"""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    return [order[i:i + per] for i in range(0, len(order), per)]


def _inproc(args, align, lib, timed_regs, warm_regs, n_regions, shape, torch):
    """--inproc: the one-process model.  Every device gets --steps batches (weak scaling, the seeds its rank would have had), already uploaded; the timed region is one
    lcd_dispatch_run over all of them -- one submitter thread per device taking `coalesce` batches at a time from the cost-ordered queue -- between two synchronisations
    of every device; results stay on the devices (the same clock as the per-process run)."""
    n_dev = max(1, args.gpus)
    if torch.cuda.device_count() < n_dev:
        raise SystemExit(f"--inproc --gpus {n_dev}: only {torch.cuda.device_count()} device(s) visible")
    opt = align.default_opt(); opt.collect_noisy_vars = args.vars; opt.is_ont = 0 if args.shape == "hifi" else 1
    n_co = args.coalesce if args.coalesce > 0 else (32 if args.shape == "hifi" else 8 if args.shape == "sv" else 24)
    n_co = max(1, min(n_co, max(args.steps, 1)))
    per_dev = len(timed_regs) // n_dev
    slots = []   # (device, batch)
    for d in range(n_dev):
        for q in range(max(args.steps, 1)):
            slots.append((d, align.RegionBatch(opt, device=d)))

    def load(regs_of):
        for i, (d, bt) in enumerate(slots):
            bt.clear()
            for r in regs_of(d, i % max(args.steps, 1)):
                bt.add_region(r)
            bt.upload()

    def sync_all():
        for d in range(n_dev):
            torch.cuda.synchronize(d)
    disp = align.Dispatcher(devices=list(range(n_dev)), coalesce=n_co)
    disp.set_flags(1)
    n_warm = 0
    if args.warmup:
        load(lambda d, q: warm_regs[q % len(warm_regs)])
        disp.run([bt for _, bt in slots]); n_warm = len(slots)
    load(lambda d, q: timed_regs[d * per_dev + q % per_dev])
    n_rep = args.repeats if args.repeats > 0 else (5 if args.steps <= n_co else 1)
    reps = []
    for _ in range(n_rep):
        sync_all(); a0 = lib.lcd_alloc_events(); t0 = time.perf_counter()
        dev_of = disp.run([bt for _, bt in slots])
        sync_all(); el = time.perf_counter() - t0
        busy, nsub = disp.busy()
        reps.append(dict(elapsed=el, allocs=int(lib.lcd_alloc_events() - a0), busy=[round(float(x), 2) for x in busy], nsub=[int(x) for x in nsub]))
    med = sorted(reps, key=lambda r: r["elapsed"])[(n_rep - 1) // 2]
    sts = [bt.stats() for _, bt in slots]
    tot_regions = float(sum(x["n_regions"] for x in sts)); tot_bases = float(sum(x["poa_aligned_bases"] for x in sts))
    assert all(int(dev_of[i]) == slots[i][0] for i in range(len(slots)))   # a batch bound to a device stays there
    steps = max(args.steps, 1)
    out = {"metric": "regions_per_sec", "value": round(tot_regions / med["elapsed"], 2), "unit": "regions/s", "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup,
           "warmup_steps_run": n_warm, "ms_per_step": round(med["elapsed"] / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
           "data": "synthetic",
           "config": {"workload": f"configs[1] shape {shape['name']}: {n_regions} regions/step, {steps} steps per device", "regions_per_step": n_regions,
                      "process_model": "ONE process, lcd_dispatch_run: one submitter thread per device over a cost-ordered queue (the drop-in's kt_for model)",
                      "coalesced_steps_per_submission": n_co, "distinct_batches_per_gpu": per_dev, "sharding": "per-device seeds, no data-path collective"},
           "poa_aligned_bases_per_sec": round(tot_bases / med["elapsed"], 1), "regions_timed": int(tot_regions),
           "repeats": {"n": n_rep, "seconds": [round(r["elapsed"], 4) for r in reps], "reported": "median", "allocations_per_repeat": [r["allocs"] for r in reps]},
           "dispatcher": {"busy_ms_per_device": med["busy"], "submissions_per_device": med["nsub"],
                          "busy_over_wall": [round(x / (med["elapsed"] * 1e3), 3) for x in med["busy"]]},
           "results_downloaded": False,   # (the dispatcher leaves results on the devices: the same clock as the per-process line, whose timed region ends before the download)
           "roofline": None, "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    for _, bt in slots:
        bt.close()
    disp.close()


def launcher_command(n_gpus, argv, port=None, python=None):
    """The command `python bench.py --gpus N ...` turns itself into when it is started WITHOUT a launcher (no RANK in the environment): one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container's hostname may not resolve), exactly what the driver would have typed.  `argv` = the arguments
    after the script name, passed through unchanged."""
    if port is None:
        import socket
        with socket.socket() as sk:   # a free port: two benches on one host must not meet on 29500
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def needs_self_launch(args, environ):
    """N > 1 ranks wanted (one process per GPU, not the --inproc dispatcher) and nobody has launched them: RANK / WORLD_SIZE are not set"""
    return args.gpus > 1 and not args.inproc and "RANK" not in environ and "WORLD_SIZE" not in environ


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--print-launch", type=int, default=0, help="1: print the launcher command `--gpus N` (N > 1, no RANK in the environment) would exec, and exit")
    ap.add_argument("--steps", type=int, default=192)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--ref-mb", type=float, default=10.0, help="synthetic reference size per step (10 Mb = configs[1]: 1 250 regions)")
    ap.add_argument("--job-mb", type=float, default=0.0, help="time ONE job of this many Mb sharded over the ranks (configs[3] / configs[4]): 500 kb chunks "
                    "(the reference's unit of work, src/bam_utils.h:10) in contiguous blocks per rank, rebalanced, then submitted ref-mb at a time; "
                    "--steps is ignored, scaling is strong")
    ap.add_argument("--rebalance", type=int, default=1, help="--job-mb with N > 1: one RCCL epoch of queue rebalancing before the timed run "
                    "(longcalld_amd/rebalance.py: all_gather of the queues, whole packed chunks sent from the deepest to the shallowest); 0 = keep the contiguous blocks")
    ap.add_argument("--distinct", type=int, default=0, help="distinct-seed batches the timed steps cycle over (0 = as many as one submission coalesces, at least 10: "
                    "every batch of a submission is a different 10 Mb of regions)")
    ap.add_argument("--repeats", type=int, default=0, help="how often the timed region (exactly --steps steps, barrier + synchronize on both sides) is run; `value` is the "
                    "MEDIAN repeat, all of them are listed (0 = 5 when the region is a single submission -- the driver's flags: 0.3 s -- else 1)")
    ap.add_argument("--inproc", type=int, default=0, help="1: ONE process drives all --gpus devices through lcd_dispatch_run (submitter threads over one cost-ordered queue: the "
                    "drop-in's model -- kt_for workers of one longcallD process, src/call_var_main.c:773) instead of one process per GPU; not under torchrun")
    ap.add_argument("--depth-profile", type=int, default=-1, help="also time lone submissions of 1 / 4 / 16 batches (what a caller with fewer chunks in flight gets); "
                    "-1 = on for the single-GPU HiFi run")
    ap.add_argument("--shape", default="hifi", choices=["hifi", "ont", "ont60", "sv"])
    ap.add_argument("--cpu-sample", type=int, default=200, help="cap on regions per CPU worker of the cpu_baseline leg, which is bounded to ~12 s (rank 0, N=1 only; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU worker processes of the cpu_baseline leg (0 = all host cores)")
    ap.add_argument("--seed", type=int, default=20250928)
    ap.add_argument("--f3", type=int, default=-1, help="also measure the step in FRONT of the path on the device (SURVEY 8f f3; never part of `value`): BGZF inflate of 128 MB of BAM-like "
                                                       "data against host zlib, and one 500 kb chunk of an indexed BAM -> device-resident chunk against the host loader; -1 = on for the "
                                                       "single-GPU HiFi line, 0 = off")
    ap.add_argument("--lanes", type=int, default=0,
                    help="concurrent submission lanes per GPU (host threads, one leader lcd_batch_t + HIP stream each) -- the reference's own "
                         "execution model: kt_for runs n_threads chunk workers concurrently (src/call_var_main.c:773); the K steps are "
                         "dealt round-robin to the lanes and ALL of them complete inside the timed region")
    ap.add_argument("--vars", type=int, default=0, choices=[0, 1, 2],
                    help="SURVEY 8(f) f1: also run the candidate-variant stage S6 inside the timed step (1), and leave the alignment strings "
                         "in HBM at download (2); 0 = the headline path exactly as collect_noisy_reg_aln_strs defines it")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="steps (batches) submitted together through lcd_batch_run_many: one set of launches per stage over the chains of "
                         "all of them, so that the GPU's workgroup dispatcher -- not HIP streams -- packs several chunks' chains onto the CUs")
    ap.add_argument("--overlap", type=int, default=-1, help="PCIe-inclusive pipeline (never `value`): BENCH_OVL_LANES (3) submissions of the timed region's size take turns on the GPU while a pool of "
                    "host threads downloads and materialises the previous one's results and uploads the next one's inputs; -1 = on for the single-GPU HiFi line, 0 = off")
    ap.add_argument("--e2e", type=int, default=0, help="also time the PCIe-inclusive path with E lanes (host threads) each doing upload -> run -> download -> "
                    "materialisation of every result for its own batches, so that one lane's copies overlap another's kernels; reported under pcie_inclusive")
    args = ap.parse_args()
    if needs_self_launch(args, os.environ):
        # `python bench.py --gpus N` by itself: become N ranks (the first multi-GPU run must not die at init_process_group for want of a launcher)
        cmd = launcher_command(args.gpus, [a for a in sys.argv[1:]])
        if args.print_launch:
            print(json.dumps({"launch": cmd}), flush=True)
            return
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: what the host driver supports; RCCL's peer mappings fail without it)
        env.setdefault("OMP_NUM_THREADS", "1")
        sys.stdout.flush(); sys.stderr.flush()
        os.execve(cmd[0], cmd, env)
    if args.print_launch:
        print(json.dumps({"launch": None, "why": "no launcher needed: --gpus 1, --inproc, or RANK / WORLD_SIZE already set"}), flush=True)
        return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("BENCH_LAUNCH_PROBE") == "1":
        # (tests/test_dist_cpu.py: the launcher path end to end without a GPU -- the ranks the self-launch made meet over gloo, rank 0 says who came, nobody touches HIP)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "pid": os.getpid()})
        if rank == 0:
            print(json.dumps({"probe": True, "world": world, "gpus_arg": args.gpus, "ranks": seen}), flush=True)
        dist.destroy_process_group()
        return
    from longcalld_amd import jobs
    shape = jobs.SHAPES[args.shape]
    n_regions = jobs.regions_for_ref_mb(args.ref_mb)
    job_mode = args.job_mb > 0

    # ---- synthetic region jobs first (worker processes forked before torch / HIP start): timed seeds and, disjoint from them, warm-up seeds ----
    CHUNK_MB = 0.5
    if job_mode:
        # the job's chunks in genome order; this rank generates only its contiguous block, the rebalance epoch below moves whole chunks
        n_chunks = max(1, int(round(args.job_mb / CHUNK_MB)))
        lo, hi = job_block(n_chunks, rank, world)
        timed_seeds = [args.seed + i for i in range(lo, hi)]
        regs_per_unit = jobs.regions_for_ref_mb(CHUNK_MB)
    else:
        if args.distinct <= 0:   # (the submission size is fixed further down from the same arguments)
            co_guess = args.coalesce if args.coalesce > 0 else (32 if args.shape == "hifi" else 8 if args.shape == "sv" else 24)
            args.distinct = max(10, min(co_guess, max(args.steps, 1), 32))
        timed_seeds = [args.seed + 1000 * rank + i for i in range(max(1, args.distinct))]   # weak scaling: same work per GPU, different seeds
        if args.inproc and world == 1 and args.gpus > 1:   # one process, N devices: device d's batches are the seeds rank d would have had
            timed_seeds = [args.seed + 1000 * d + i for d in range(args.gpus) for i in range(max(1, args.distinct))]
        regs_per_unit = n_regions
    n_warm_seeds = (6 if args.shape == "hifi" else 3) if args.warmup else 0   # (more warm-up data: the grow-only buffers see a wider sample of batch sizes before the clock starts)
    warm_seeds = [args.seed + 500000 + 1000 * rank + i for i in range(n_warm_seeds)]
    import multiprocessing as mp
    n_gen_procs = max(1, min(len(timed_seeds) + len(warm_seeds), _host_cores() // max(1, world)))
    with mp.get_context("fork").Pool(n_gen_procs) as pool:
        gen = pool.map(_gen, [(sd, regs_per_unit, args.shape, job_mode) for sd in timed_seeds] + [(sd, n_regions, args.shape) for sd in warm_seeds], chunksize=1)
    timed_regs, warm_regs = gen[:len(timed_seeds)], gen[len(timed_seeds):]

    import torch
    import torch.distributed as dist
    if (args.gpus > 1 and not args.inproc) or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank)

    from longcalld_amd import align, _lib
    lib = _lib.load_library()
    _lib.check(lib.lcd_init(local_rank), lib)
    if args.inproc and world == 1 and not job_mode:
        return _inproc(args, align, lib, timed_regs, warm_regs, n_regions, shape, torch)

    bench_opt = align.default_opt()
    bench_opt.collect_noisy_vars = args.vars
    bench_opt.is_ont = 0 if args.shape == "hifi" else 1   # the reference's --ont / --hifi switch (src/call_var_main.h:128)
    noisy = args.shape != "hifi"
    import threading
    # a step = one batch.  `coalesce` steps are submitted together through lcd_batch_run_many (one set of launches per stage over all their
    # chains), `lanes` host threads keep that many such submissions in flight.
    # defaults (measured): HiFi shape = two lanes of 32 batches per submission when there are enough steps for each lane to
    # have two submissions (one lane's anchor / WFA / string stages and the tail of its chain launches overlap the other lane's chains),
    # else one lane of up to 32; the noisy-read shapes (4x graph estimates, ~8 GB per batch in flight) run one lane of 24 (sv: 8)
    rb_stats = None
    rccl_info = None
    lib_comm = None
    if world > 1:
        # The library's own RCCL communicator (lcd_comm_create over the dlopen'ed librccl): torch.distributed only carries its 128-byte id.  What RCCL reports for
        # it on every rank (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) goes into the line: evidence that N ranks really met over RCCL
        from longcalld_amd import rebalance as rb
        # (every collective sits OUTSIDE the try blocks: a rank whose librccl call fails still takes part in the broadcast / gather the others are waiting in)
        uid = [None]
        if rank == 0:
            try:
                uid = [rb.Comm.unique_id()]
            except Exception as e:  # noqa
                uid = [None]
                print(f"bench.py: ncclGetUniqueId failed on rank 0: {e!r}", file=sys.stderr, flush=True)
        dist.broadcast_object_list(uid, src=0)
        try:
            if uid[0] is None:
                raise RuntimeError("no RCCL unique id from rank 0")
            lib_comm = rb.Comm(world, rank, uid[0], local_rank)
            mine = dict(lib_comm.info(), rank=rank, local_rank=local_rank, hip_device=int(torch.cuda.current_device()))
        except Exception as e:  # noqa  (the run itself does not depend on it: the epoch falls back to torch.distributed's RCCL -- decided together, below)
            lib_comm = None
            mine = {"rank": rank, "error": repr(e)[:200]}
        allinfo = [None] * world
        dist.all_gather_object(allinfo, mine)
        rccl_info = {"library_communicator": "lcd_comm_create (librccl via dlopen), id broadcast by torch.distributed", "per_rank": allinfo,
                     "world_seen_by_rccl": sorted({i.get("nccl_world", -1) for i in allinfo}), "torch_backend": dist.get_backend()}
    if job_mode:
        from longcalld_amd import rebalance as rb
        queue = [(sum(rb.region_cost(r) for r in regs), rb.pack_regions(regs)) for regs in timed_regs]
        if world > 1 and args.rebalance:
            t_rb = time.perf_counter()
            depth_before = len(queue)
            # default: the epoch inside liblcd_hotpath.so (lcd_rebalance_exchange over librccl).  LCD_REBALANCE_VIA=torch (or no library communicator on some
            # rank -- decided together, a rank must not wait in a collective the others never enter): the same plan (lcd_rebalance_plan) with torch.distributed's
            # RCCL point-to-point as the transport
            ok_t = torch.tensor([1 if lib_comm is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            if os.environ.get("LCD_REBALANCE_VIA", "lib") == "lib" and int(ok_t.item()) == 1:
                queue, rb_stats = rb.rebalance_c(lib_comm, queue)
            else:
                queue, rb_stats = rb.rebalance(queue, device=dev)
                rb_stats["transport"] = "torch.distributed (RCCL) point-to-point"
            rb_stats["seconds"] = round(time.perf_counter() - t_rb, 4)
            # every rank's queue depth and load before / after the epoch, so that a scaling run shows what the epoch did
            dl = torch.tensor([depth_before, len(queue), sum(c for c, _ in queue)], dtype=torch.float64, device=dev); al = [torch.zeros_like(dl) for _ in range(world)]
            dist.all_gather(al, dl)
            rb_stats["queue_depth_before_per_rank"] = [int(x[0].item()) for x in al]; rb_stats["queue_depth_after_per_rank"] = [int(x[1].item()) for x in al]
            rb_stats["load_after_per_rank"] = [float(x[2].item()) for x in al]
        else:
            loads = [sum(c for c, _ in queue)]
            if world > 1:
                lt = torch.tensor(loads, dtype=torch.float64, device=dev); al = [torch.zeros_like(lt) for _ in range(world)]
                dist.all_gather(al, lt); loads = [float(x.item()) for x in al]
            rb_stats = {"n_moves": 0, "moved_bytes": 0, "imbalance_before": max(loads) / (sum(loads) / len(loads)) if sum(loads) > 0 else 1.0, "loads_before": loads}
            rb_stats["imbalance_after"] = rb_stats["imbalance_before"]
        # this rank's chunks, heaviest first, ref-mb of them per batch (= one step)
        per = max(1, int(round(args.ref_mb / CHUNK_MB)))
        chunks = [rb.unpack_regions(b) for _, b in queue]
        timed_regs = [[r for i in idxs for r in chunks[i]] for idxs in job_batches([c for c, _ in queue], per)]
        args.steps = len(timed_regs)
    if args.lanes <= 0:
        args.lanes = 2 if (args.shape == "hifi" and args.coalesce <= 0 and args.steps >= 128) else 1
    if args.coalesce <= 0:
        args.coalesce = 32 if args.shape == "hifi" else 8 if args.shape == "sv" else 24
    n_co = max(1, min(args.coalesce, max(args.steps, 1)))
    n_lanes = 1 if job_mode else max(1, min(args.lanes, (args.steps + n_co - 1) // n_co))
    # slots = uploaded batches: the weak-scaling run re-submits lanes x coalesce slots (cycling over the distinct batches); a job owns one slot
    # per batch and submits them `coalesce` at a time, each exactly once
    n_slots = max(args.steps, 1) if job_mode else n_lanes * n_co

    def load_slot(bt, regs):
        bt.clear()
        for r in regs:
            bt.add_region(r)
        t0_ = time.perf_counter()
        bt.upload()
        return time.perf_counter() - t0_

    groups = [[align.RegionBatch(bench_opt) for _ in range(n_slots // n_lanes)] for _ in range(n_lanes)]
    batches = [bt for grp in groups for bt in grp]
    t_up = 0.0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    acc = {"ms": 0.0, "launches": 0, "st": None, "alg": 0.0, "cells": 0.0, "regions": 0.0, "bases": 0.0, "resolved": 0, "wfa_off": 0.0, "ed_blocks": 0.0, "ms_wfa": 0.0, "ms_anchor": 0.0, "cells_computed": 0.0}
    lock = threading.Lock()
    lane_errors = []

    def lane_work(grp, sizes, record):
        try:
            for a, k in sizes:
                align.RegionBatch.run_many(grp[a:a + k])
                if record:
                    sts = [bt.stats() for bt in grp[a:a + k]]
                    with lock:
                        acc["ms"] += sts[0]["ms_poa_kernel"]; acc["launches"] += sts[0]["n_poa_launches"]; acc["st"] = sts[0]
                        acc["ms_wfa"] += sts[0]["ms_wfa"] + sts[0]["ms_anchor"]; acc["ms_anchor"] += sts[0]["ms_anchor"]
                        for x in sts:
                            acc["alg"] += float(x["poa_alg_bytes"]); acc["cells"] += float(x["poa_cells"]); acc["cells_computed"] += float(x["poa_cells_computed"]); acc["regions"] += float(x["n_regions"])
                            acc["bases"] += float(x["poa_aligned_bases"]); acc["resolved"] += int(x["n_regions_resolved"])
                            acc["wfa_off"] += float(x["wfa_offsets"]); acc["ed_blocks"] += float(x["edlib_blocks"])
        except Exception as e:  # noqa: surfaced after the join (a thread's traceback alone would leave a half-measured line)
            lane_errors.append(e)

    def run_steps(n_steps, record):
        # cut the steps into submissions of `coalesce` and deal those round-robin to the lanes; lanes overlap on the GPU
        subs = [n_co] * (n_steps // n_co) + ([n_steps % n_co] if n_steps % n_co else [])
        if job_mode:   # consecutive slots, each batch once (warm-up: the same walk over the warm-up data)
            starts = [sum(subs[:i]) % max(n_slots, 1) for i in range(len(subs))]
            subs = [(a, min(k, n_slots - a)) for a, k in zip(starts, subs)]
        else:
            subs = [(0, k) for k in subs]
        share = [subs[i::n_lanes] for i in range(n_lanes)]
        ths = [threading.Thread(target=lane_work, args=(groups[i], share[i], record)) for i in range(n_lanes) if share[i]]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if lane_errors:
            raise lane_errors[0]

    # ---- warm-up on OTHER seeds: every slot's buffers grow to size, the capacity hints settle (the noisy-read shapes need three submissions:
    # the hints rise one level per submission that overflowed) -- none of it on the data that is timed ----
    n_warm = 0
    if args.warmup and args.steps > 0:
        for q, bt in enumerate(batches):
            load_slot(bt, warm_regs[q % len(warm_regs)])
        n_warm = max(args.warmup, (min(n_slots, n_co) if job_mode else n_slots) * (3 if noisy else 1))
        run_steps(n_warm, False)
    for q, bt in enumerate(batches):
        if args.steps > 0:
            t_up = load_slot(bt, timed_regs[q % len(timed_regs)])
    # the timed region = exactly --steps steps between two (barrier + synchronize); a region that is ONE submission lasts a third of a second, so it is repeated
    # and the MEDIAN repeat is the one reported (value, stage times, statistics all come from that repeat; every repeat's time is listed)
    n_rep = args.repeats if args.repeats > 0 else (5 if (not job_mode and args.steps <= n_co * n_lanes) else 1)
    reps = []
    _rep = 0
    while _rep < n_rep:
        _rep += 1
        for k_ in acc:
            acc[k_] = None if k_ == "st" else 0.0 if isinstance(acc[k_], float) else 0
        barrier()
        allocs0 = lib.lcd_alloc_events()
        t0 = time.perf_counter()
        if args.steps > 0:
            run_steps(args.steps, True)
        barrier()
        el = time.perf_counter() - t0
        rank_el = el
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        reps.append({"elapsed": el, "rank_elapsed": rank_el, "allocs": int(lib.lcd_alloc_events() - allocs0), "acc": dict(acc)})
        # `value` never comes from a repeat that allocated device memory inside its timed region (a buffer of the library that had to grow on the
        # timed seeds: hipFree + hipMalloc synchronise the device).  Such a repeat has grown the buffers for good, so the region is simply timed again (every
        # repeat, allocating or not, is listed in `repeats`); if three extra repeats do not settle it the line is not printed and the exit code says so.
        if _rep == n_rep and not any(r_["allocs"] == 0 for r_ in reps) and n_rep < (args.repeats if args.repeats > 0 else 1) + 3 and args.steps > 0 and world == 1:
            n_rep += 1   # (one rank only: with several ranks every rank must run the same number of barriers)
    if world > 1 and args.steps > 0:   # (every rank must take the same repeat: the allocation counts are agreed first -- MAX over the ranks, per repeat)
        a_t = torch.tensor([r_["allocs"] for r_ in reps], dtype=torch.int64, device=dev)
        dist.all_reduce(a_t, op=dist.ReduceOp.MAX)
        for r_, a_ in zip(reps, a_t.tolist()):
            r_["allocs_any_rank"] = int(a_)
    clean = [i for i in range(n_rep) if reps[i].get("allocs_any_rank", reps[i]["allocs"]) == 0] if args.steps > 0 else list(range(n_rep))
    reported = "median of the repeats without an allocation"
    if world > 1 and not clean:   # (no extra repeats with several ranks: the line says what it is)
        clean = list(range(n_rep))
        reported = "median of all repeats (every repeat allocated device memory on some rank; extra repeats are only run with one rank)"
    if not clean:
        print(f"bench.py: every one of the {n_rep} repeats of the timed region allocated device memory (allocations per repeat: {[r_['allocs'] for r_ in reps]})", file=sys.stderr, flush=True)
        sys.exit(3)
    order = sorted(clean, key=lambda i: reps[i]["elapsed"])
    med = reps[order[(len(order) - 1) // 2]]
    elapsed, rank_elapsed, allocs_timed = med["elapsed"], med["rank_elapsed"], med["allocs"]
    acc.update(med["acc"])
    dev_gb = lib.lcd_device_bytes(local_rank) / 1e9
    # host threads of this rank beside the submitting thread: the library divides the CPUs the process may use (affinity mask, cgroup quota) by the
    # ranks on the host, so N ranks do not each start the teams a lone process starts
    import ctypes as _C
    _ht = [_C.c_int(0) for _ in range(4)]
    lib.lcd_host_threads(*[_C.byref(v) for v in _ht])
    host_info = {"host_threads_per_rank": {"team": _ht[0].value, "arena": _ht[1].value}, "cpus_this_process_may_use": _ht[2].value, "ranks_on_this_host": _ht[3].value,
                 "generator_processes": n_gen_procs}
    poa_kernel_ms, poa_launches, st = acc["ms"], acc["launches"], acc["st"]
    tot_regions, tot_bases = acc["regions"], acc["bases"]
    if world > 1:
        cnt = torch.tensor([tot_regions, tot_bases, rank_elapsed], dtype=torch.float64, device=dev)
        allc = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(allc, cnt)
        tot_regions, tot_bases = float(sum(c[0].item() for c in allc)), float(sum(c[1].item() for c in allc))
        rank_times = [round(float(c[2].item()), 4) for c in allc]
    else:
        rank_times = [round(elapsed, 4)]

    # lone submissions of 1 / 4 / 16 batches: what a caller with fewer chunks in flight than the timed region's gets (kt_for offers n_threads chunks per pass,
    # src/collect_var.c:2952-2969); never `value`
    depth = None
    if (args.depth_profile > 0 or (args.depth_profile < 0 and args.shape == "hifi" and not job_mode)) and world == 1 and args.steps > 0 and not args.vars:
        depth = {}
        for dsz, dn in ((1, 5), (4, 3), (16, 3)):
            if dsz > len(groups[0]):
                continue
            ts = []
            for q in range(dn + 1):   # (the first one is not timed: a submission of another size re-plans the leader's buffers)
                torch.cuda.synchronize(); td0 = time.perf_counter()
                align.RegionBatch.run_many(groups[0][:dsz])
                torch.cuda.synchronize()
                if q:
                    ts.append(time.perf_counter() - td0)
            ts.sort()
            depth[str(dsz)] = {"regions_per_sec": round(dsz * n_regions / ts[len(ts) // 2], 1), "ms_per_submission": round(ts[len(ts) // 2] * 1e3, 2), "submissions": dn}

    # PCIe-inclusive with overlap (never `value`): E lanes, each upload -> run_many -> download -> digest (lcd_batch_digest materialises every
    # malloc()'d aln_str_t / variant record of the batch exactly as a caller would receive them) on its own group of slots
    e2e = None
    if args.e2e > 0 and args.steps > 0 and world == 1:
        E = max(1, min(args.e2e, n_slots))
        per_lane = max(1, min(n_co, n_slots // E))
        egroups = [batches[i * per_lane:(i + 1) * per_lane] for i in range(E)]
        rounds = max(1, min(4, args.steps // (E * per_lane)))
        e_regions = [0.0]
        e_err = []

        def e_lane(grp, n_rounds=None):
            try:
                for _ in range(rounds if n_rounds is None else n_rounds):
                    for bt in grp:
                        bt.upload()
                    align.RegionBatch.run_many(grp)
                    for bt in grp:
                        bt.download(); bt.materialize()
                    with lock:
                        e_regions[0] += sum(float(bt.stats()["n_regions"]) for bt in grp)
            except Exception as e:  # noqa
                e_err.append(e)
        # one untimed round first: every lane's leader sizes its buffers for a submission of this shape (the lanes of the timed region above had other leaders)
        ths = [threading.Thread(target=e_lane, args=(g, 1)) for g in egroups if g]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        e_regions[0] = 0.0
        barrier()
        te0 = time.perf_counter()
        ths = [threading.Thread(target=e_lane, args=(g,)) for g in egroups if g]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        te = time.perf_counter() - te0
        if e_err:
            raise e_err[0]
        e2e = {"lanes": E, "batches_per_submission": per_lane, "rounds": rounds, "seconds": round(te, 4), "regions_per_sec": round(e_regions[0] / te, 1),
               "what": "upload + run + download + materialisation of every result (strings" + (" left in HBM, variants + alleles" if args.vars == 2 else "") + "), lanes overlapped"}

    # PCIe-inclusive PIPELINE (never `value`): what a caller that consumes every result on the host gets when it keeps two submissions in flight --
    # the reference's kt_for workers do exactly that with their chunks (src/collect_var.c:2952-2969: one pass's regions are independent).  Two lanes, each with its
    # OWN batches of the timed region's size: a lane runs its submission (the GPU phases of the two lanes are serialised by a lock: two submissions side by side only
    # time-slice the hardware queues), then a pool of host threads does lcd_batch_download + lcd_batch_region_results_arena of every batch (and uploads the inputs
    # again, as a new pass would) WHILE the other lane's kernels run.
    overlap = None
    if (args.overlap > 0 or (args.overlap < 0 and args.shape == "hifi")) and world == 1 and not job_mode and args.steps > 0 and args.vars != 2 and st is not None:
        from concurrent.futures import ThreadPoolExecutor
        n_sub = min(n_co, len(groups[0]))
        n_ol = max(2, int(os.environ.get("BENCH_OVL_LANES", "3")))   # (three: a lane's cycle is run + results + upload = ~300 ms of which 200 on the GPU -- with two lanes the GPU waits for results a fifth of the time)
        lanes2 = [groups[0][:n_sub]] + [[align.RegionBatch(bench_opt) for _ in range(n_sub)] for _ in range(n_ol - 1)]
        for grp_ in lanes2[1:]:
            for q, bt in enumerate(grp_):
                load_slot(bt, timed_regs[(q + 3) % len(timed_regs)])
        # (measured on the 16-CPU quota of the GPU box: 6 threads x 1 -> 103 k regions/s, 10 x 1 -> 97 k, 8 x 16 (a thread team inside every call) -> 66 k: the
        #  submission's own host threads -- job tables, launches -- need cores too, and a cgroup that runs out of quota stalls all of them)
        n_host = int(os.environ.get("BENCH_OVL_THREADS", "0")) or max(2, min(8, (_host_cores() * 3) // 8))
        arena_env = os.environ.get("LCD_ARENA_THREADS")
        os.environ["LCD_ARENA_THREADS"] = os.environ.get("BENCH_OVL_ARENA_THREADS", "1")   # host threads INSIDE one lcd_batch_region_results_arena call (n_host calls run side by side)
        team_env = os.environ.get("LCD_HOST_TEAM")
        os.environ["LCD_HOST_TEAM"] = os.environ.get("BENCH_OVL_TEAM", "4")   # the submission's own short thread teams, beside n_host busy pool threads on a 16-CPU quota
        pool_ex = ThreadPoolExecutor(n_host)
        gpu_lock = threading.Lock()
        o_err, o_bytes = [], [0]

        o_t = {"wait_results": 0.0, "upload": 0.0, "wait_gpu": 0.0, "run": 0.0, "download_thread_s": 0.0, "arena_thread_s": 0.0}

        def post(bt):
            t1 = time.perf_counter()
            bt.download()
            t2 = time.perf_counter()
            nreg, nbytes = bt.results_arena(parse=False)
            t3 = time.perf_counter()
            with lock:
                o_t["download_thread_s"] = o_t.get("download_thread_s", 0.0) + (t2 - t1); o_t["arena_thread_s"] = o_t.get("arena_thread_s", 0.0) + (t3 - t2)
            return nbytes

        def o_lane(grp, rounds_):
            try:
                pending = None
                for _ in range(rounds_):
                    ta = time.perf_counter()
                    if pending is not None:
                        nb_ = sum(f.result() for f in pending)
                        tb = time.perf_counter()
                        with lock:
                            o_bytes[0] += nb_; o_t["wait_results"] += tb - ta
                        list(pool_ex.map(lambda bt: bt.upload(), grp))   # the next pass's inputs (the same regions: a caller's next chunk has the same shape), once every
                        ta = time.perf_counter()                           # result of the last one is on the host
                        with lock:
                            o_t["upload"] += ta - tb
                    with gpu_lock:
                        tb = time.perf_counter()
                        align.RegionBatch.run_many(grp)
                        tc = time.perf_counter()
                    with lock:
                        o_t["wait_gpu"] += tb - ta; o_t["run"] += tc - tb
                    pending = [pool_ex.submit(post, bt) for bt in grp]
                nb_ = sum(f.result() for f in pending)
                with lock:
                    o_bytes[0] += nb_
            except Exception as e:  # noqa
                o_err.append(e)
        for grp in lanes2:      # untimed: the second lane's buffers grow to size, every batch's host blocks are allocated once
            o_lane(grp, 1)
        o_rounds = int(os.environ.get("BENCH_OVL_ROUNDS", "4"))   # (per lane: the first upload and the last download are not overlapped -- 2 of 6 submissions at 3 rounds, 2 of 10 at 5)
        o_bytes[0] = 0
        for k_ in o_t:
            o_t[k_] = 0.0
        barrier()
        to0 = time.perf_counter()
        ths = [threading.Thread(target=o_lane, args=(g, o_rounds)) for g in lanes2]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        barrier()
        to = time.perf_counter() - to0
        pool_ex.shutdown()
        if arena_env is None:
            os.environ.pop("LCD_ARENA_THREADS", None)
        else:
            os.environ["LCD_ARENA_THREADS"] = arena_env
        if team_env is None:
            os.environ.pop("LCD_HOST_TEAM", None)
        else:
            os.environ["LCD_HOST_TEAM"] = team_env
        if o_err:
            raise o_err[0]
        overlap = {"lanes": n_ol, "batches_per_submission": n_sub, "rounds_per_lane": o_rounds, "host_threads": n_host, "seconds": round(to, 4),
                   "regions_per_sec": round(n_ol * o_rounds * n_sub * n_regions / to, 1), "result_bytes_per_submission": int(o_bytes[0] / (n_ol * o_rounds)),
                   "lane_seconds": {k_: round(v_ / n_ol, 4) for k_, v_ in o_t.items()},
                   "what": "several submissions in flight (`lanes`): lcd_batch_run_many of one while a pool of host threads runs lcd_batch_download + lcd_batch_region_results_arena + "
                           "(then) lcd_batch_upload for every batch of the other; every result byte lands in host memory (one block per batch)"}
        for grp_ in lanes2[1:]:
            for bt in grp_:
                bt.close()

    # PCIe-inclusive figure for DESIGN.md (never `value`)
    digest, t_dl, t_ar, arena_bytes = 0, 0.0, 0.0, 0
    if st is not None:
        t_dl0 = time.perf_counter()
        batches[0].download()
        batches[0].materialize()          # every result as a caller receives it (malloc()'d rows); the digest below also hashes every byte
        t_dl = time.perf_counter() - t_dl0
        t_ar0 = time.perf_counter()
        batches[0].download()
        arena_regions, arena_bytes = batches[0].results_arena(parse=False) if args.vars != 2 else (0, 0)   # the same results in ONE host block (additive entry)
        t_ar = time.perf_counter() - t_ar0
        digest = batches[0].digest()

    if rank == 0:
        steps = max(args.steps, 1)
        ms_step = elapsed / steps * 1e3
        value = tot_regions / elapsed
        # roofline of the dominant kernel (POA chains): algorithmic bytes (SURVEY 8d B_poa) per launch / mean launch time
        alg_bytes = acc["alg"] / max(poa_launches, 1)     # per launch set: the chains of `coalesce` batches
        mean_launch_s = poa_kernel_ms / max(poa_launches, 1) * 1e-3
        achieved = alg_bytes / mean_launch_s / 1e9 if mean_launch_s > 0 else 0.0
        # HBM bytes of the same launch from the PMC counters: they need rocprofv3 (separate --pmc passes, tools/profile_round.sh), so the
        # figure is the committed measurement of this exact workload (profiles/<tag>_traffic.json), or null for any other workload
        traffic, traffic_src = None, None
        import glob
        import re
        if args.ref_mb == 10 and shape["name"] in ("hifi", "ont", "sv") and world == 1 and not job_mode:
            # (the driver's workload: profiles/rNN_vM_traffic.json; the two noisy shapes' builder commands: profiles/rNN_vM_<shape>_traffic.json, tools/profile_shapes.sh)
            pat = r"^r\d+_v\d+_traffic\.json$" if shape["name"] == "hifi" else r"^r\d+_v\d+_" + shape["name"] + r"_traffic\.json$"
            cand = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")) if re.match(pat, os.path.basename(f))),
                          key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])   # r01_v9 < r01_v10 < r02_v1
            if cand:
                tj = json.load(open(cand[-1]))
                traffic, traffic_src = float(tj["hbm_bytes_per_step"]) * n_co, tj["source"] + f" x {n_co} coalesced steps"
        # secondary roofline: integer VALU issue.  Wavefront instructions of the POA kernels per step from the committed SQ counter
        # pass (same rule as `traffic`), x 64 lanes, over the live launch time; peak = 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz = 78.6 Tops/s
        valu = None
        if traffic is not None:
            cand2 = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq.json")), key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
            if cand2 and mean_launch_s > 0 and shape["name"] == "hifi":
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
                    # cells / bytes above are those of the REFERENCE's algorithm (K1 adaptive band, K2 full rows).  K2 chains of clean reads run
                    # over a certified band (same alignments): what the kernels actually computed / streamed is the smaller figure below
                    "cells_computed_per_launch": int(acc["cells_computed"] / max(poa_launches, 1)),
                    "gcups_computed": round(acc["cells_computed"] / max(poa_launches, 1) / mean_launch_s / 1e9, 3) if mean_launch_s > 0 else 0.0,
                    "achieved_computed": round((acc["alg"] - acc["cells"] + acc["cells_computed"]) / max(poa_launches, 1) / mean_launch_s / 1e9, 3) if mean_launch_s > 0 else 0.0,
                    "valu": valu}
        if n_lanes > 1 and world == 1:
            # `achieved` prices ONE lane's launch set against its own duration while the other lanes' chains share the CUs; the chip-wide rate is
            # all algorithmic bytes of the timed region over its wall clock (conservative: the non-POA stages are inside that time)
            roofline["concurrent_launch_sets"] = n_lanes
            roofline["achieved_wall"] = round(acc["alg"] / elapsed / 1e9, 3); roofline["frac_wall"] = round(acc["alg"] / elapsed / 1e9 / HBM_PEAK_GBS, 6)
        cpu = None
        if world == 1 and args.cpu_sample > 0 and st is not None:
            # the reference runs this path on kt_for worker threads (src/call_var_main.c:773): time the CPU port the same way, one
            # process per host core (spawned: no fork after HIP start-up), every worker on its own slice of the same workload shape
            n_avail = _host_cores()
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
                             f"{wall:.1f} s with process start-up), oracle/ C restatement (-O3 scalar, NOT upstream SIMD abPOA/WFA2: those submodules are "
                             f"absent from the reference checkout, so this is a port, not the reference); one worker alone: {one[0] / one[1]:.1f} regions/s"}
            # the one kernel whose REFERENCE implementation exists here: K4 on the reference's own vendored edlib, same job set as one step
            try:
                pairs = batches[0].k4_pairs()
                ref = _k4_reference(pairs, n_workers)
                if ref is not None:
                    g0 = time.perf_counter(); align.edlib_batch(pairs); g1 = time.perf_counter(); align.edlib_batch(pairs); g2 = time.perf_counter()
                    blocks = float(st["edlib_blocks"])
                    cpu["k4_reference"] = {"kind": "reference", "value": round(ref[0], 1), "unit": "edlib_xgaps pairs/s", "cores": n_workers,
                                           "sample": f"{ref[2]} NW+path alignments (the {len(pairs)} K4 jobs of one step, cycled) in {ref[1]:.2f} s on {n_workers} threads of the "
                                                     f"reference's own edlib (oracle/_ref/libedlib_ref.so)",
                                           "gpu_pairs_per_sec_batch_call": round(len(pairs) / max(g2 - g1, 1e-9), 1),
                                           "gpu_note": "lcd_edlib_batch on the same pairs, one call including its allocations and both PCIe copies (second call; the kernel "
                                                       "itself is inside ms_anchor of the step)", "k4_jobs_per_step": len(pairs), "myers_blocks_per_step": blocks}
            except Exception as e:  # noqa
                cpu["k4_reference"] = {"error": str(e)}
        cfg_name = {"hifi": "configs[1]", "ont": "configs[2]", "ont60": "configs[2] at 60x", "sv": "configs[4] (60x noisy reads, 1-10 kb INS/DEL)"}[args.shape]
        if job_mode:
            cfg_name = ("configs[3]" if args.shape == "hifi" else cfg_name) + f": one {args.job_mb:g} Mb job sharded over the ranks"
        out = {
            "metric": "regions_per_sec", "value": round(value, 2), "unit": "regions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "warmup_steps_run": n_warm,
            "ms_per_step": round(ms_step, 3), "higher_is_better": True, "scaling": "strong" if job_mode else "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"{cfg_name}: synthetic {shape['depth']}x {shape['name']} region jobs, {args.ref_mb:g} Mb of reference per step "
                                   f"({n_regions} regions/step, ~{tot_bases / max(world, 1) / steps / 1e6:.1f} Mbase POA-aligned/step)",
                       "regions_per_step": n_regions, "distinct_batches_per_gpu": len(timed_regs), "warmup_on_other_seeds": bool(n_warm),
                       "sharding": "per-rank seeds, no data-path collective",
                       "lanes_per_gpu": n_lanes, "coalesced_steps_per_submission": n_co},
            "poa_aligned_bases_per_sec": round(tot_bases / elapsed, 1),
            "regions_resolved": int(acc["resolved"]),
            "regions_timed": int(tot_regions),
            "wfa_offsets_per_sec": round(acc["wfa_off"] / max(acc["ms_wfa"] * 1e-3, 1e-9), 1),
            "edlib_blocks_per_sec": round(acc["ed_blocks"] / max(acc["ms_anchor"] * 1e-3, 1e-9), 1),
            "stage_ms": {k: round(st[k], 3) for k in ("ms_anchor", "ms_poa", "ms_wfa", "ms_strings", "ms_vars", "ms_total", "ms_host", "ms_poa_kernel")} if st else None,
            "noisy_vars_stage": args.vars,
            "rank_seconds": rank_times,
            "host": host_info,
            "repeats": {"n": n_rep, "seconds": [round(r_["elapsed"], 4) for r_ in reps], "reported": reported, "allocations_per_repeat": [r_.get("allocs_any_rank", r_["allocs"]) for r_ in reps]},
            "depth": depth,
            "device_memory": {"library_buffers_gb": round(dev_gb, 2), "allocations_inside_timed_region": int(allocs_timed)},
            "pcie_inclusive": {"upload_s": round(t_up, 4), "download_and_materialize_s": round(t_dl, 4),
                               # the pipelined rate when it was measured (two submissions in flight, see `pipeline`); else the serial sum of one batch's phases
                               "regions_per_sec": overlap["regions_per_sec"] if overlap else round(tot_regions / max(world, 1) / steps / (t_up + ms_step / 1e3 + t_dl), 2),
                               "regions_per_sec_one_batch_serial": round(tot_regions / max(world, 1) / steps / (t_up + ms_step / 1e3 + t_dl), 2),
                               "pipeline": overlap,
                               "download_and_arena_s": round(t_ar, 4), "arena_bytes": int(arena_bytes),
                               "regions_per_sec_arena": round(tot_regions / max(world, 1) / steps / (t_up + ms_step / 1e3 + t_ar), 2) if t_ar > 0 else None,
                               "what": "one batch: upload + its share of the timed run + download + every result on the host -- one malloc() per row as the reference's contract "
                                       "has it (lcd_batch_region_result), or one block per batch (lcd_batch_region_results_arena)", "overlapped": e2e},
            "digest": f"{digest:016x}",
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        if (args.f3 > 0 or (args.f3 < 0 and world == 1 and not job_mode and args.shape == "hifi" and args.cpu_sample > 0)):
            try:   # (tools/: importable measurements; a failure here never costs the line)
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_f3
                import bench_inflate
                inf = bench_inflate.measure(128, 2, host_threads=(16,))
                ch = bench_f3.measure(600, 3)
                out["f3_device"] = {"what": "in front of the path, never part of `value`: lcd_bgzf_inflate_dev (one wavefront per BGZF block, CRC-32 checked on the device) against zlib on host "
                                            "threads, 128 MB of BAM-like records; lcd_chunk_create_from_bam (compressed blocks up, records + digars in HBM) against lcd_bam_load_region_indexed + "
                                            "lcd_chunk_create for one 500 kb chunk (tools/bench_inflate.py, tools/bench_f3.py)",
                                    "inflate_device_GBps_out": inf["device_crc1"]["kernel_GBps_out"], "inflate_device_call_ms": inf["device_crc1"]["call_ms"],
                                    "inflate_host_zlib_1_thread_GBps_out": inf["host_zlib_1_thread_GBps_out"], "inflate_host_zlib_16_threads_GBps_out": inf.get("host_zlib_16_threads_GBps_out"),
                                    "chunk_reads": ch["reads_in_region"], "chunk_device_ms": ch["device_path_ms"], "chunk_host_ms": round(ch["host_load_ms"] + ch["host_chunk_create_ms"], 2)}
            except Exception as e:  # noqa
                out["f3_device"] = {"error": repr(e)[:300]}
        if job_mode:
            out["queue_rebalance"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in (rb_stats or {}).items() if not k.startswith("loads")}
            out["config"]["sharding"] = (f"{int(round(args.job_mb / CHUNK_MB))} chunks of {CHUNK_MB} Mb in contiguous blocks per rank"
                                         + (", one RCCL rebalance epoch (whole packed chunks moved)" if world > 1 and args.rebalance else "") + ", no data-path collective")
        if rccl_info is not None:
            out["rccl"] = rccl_info
        print(json.dumps(out), flush=True)
    for bt in batches:
        bt.close()
    if lib_comm is not None:
        lib_comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
