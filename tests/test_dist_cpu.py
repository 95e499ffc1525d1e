"""N > 1 path on CPU (gloo, world_size 2): the region sharding is a partition (every region on exactly one rank, no data-path
collective) and the aggregate uses the max-over-ranks time -- the same arithmetic bench.py does with RCCL on the GPUs."""
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from longcalld_amd import jobs
from oracle import pyoracle
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
regs = jobs.make_regions(31, 10)
mine = list(range(len(regs)))[rank::world]                 # static shard, independent work items
done = [pyoracle.collect_noisy_reg_aln_strs(regs[i])["n_cons"] for i in mine]
t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)   # pretend per-rank elapsed
dist.all_reduce(t, op=dist.ReduceOp.MAX)
cnt = torch.tensor([float(len(mine)), float(sum(done))], dtype=torch.float64)
dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
gathered = [None] * world
dist.all_gather_object(gathered, mine)
if rank == 0:
    print(json.dumps({"t": t.item(), "n": cnt[0].item(), "cons": cnt[1].item(), "shards": gathered}))
dist.destroy_process_group()
'''


def test_two_rank_sharding_gloo(tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29517", str(w), ROOT], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n"] == 10 and abs(r["t"] - 0.2) < 1e-9
    flat = sorted(i for s in r["shards"] for i in s)
    assert flat == list(range(10))
