"""Region-queue rebalancing across ranks: the only use of a collective library on this path.

Chunks are independent until stitch_var_main (src/collect_var.c:2983), so no data-path collective exists; what can go wrong across the 8 GPUs of a node
is load: the host enumerates chunks in genome order (src/call_var_main.c:599-621) and hands contiguous blocks to the devices, and SV-heavy chunks
(configs[4]) make blocks unequal.  One epoch here = all_gather of every rank's queue (cost + payload size per job), a deterministic plan computed
identically on every rank, and point-to-point sends of WHOLE packed job buffers from the deepest queues to the shallowest -- torch.distributed, i.e.
RCCL over xGMI with backend "nccl" on the GPU node (device tensors) and gloo in the CPU tests (host tensors).  Job buffers are 15 KB - a few MB, so
ring / per-link bandwidth is irrelevant; what matters is that a job moves at most once per epoch and that every rank agrees on the plan.
"""
import numpy as np

MAGIC = 0x4C434452  # 'LCDR'


def region_cost(reg):
    """DP-cell estimate of one region job (the same shape as lcd_batch_cost, without running the host planner): phased regions run banded K1
    chains (reads x length x band), unphased ones the unbanded K2 chain over their full-cover reads (reads x length^2)"""
    lens = np.array([len(s) for s in reg["seqs"]], np.float64)
    if len(lens) == 0:
        return 0.0
    maxl = float(lens.max())
    if (np.asarray(reg["haps"]) > 0).any():
        return float(lens.sum() * min(maxl + 1, 2 * (10 + maxl / 100) + 1 + 64))
    full = np.asarray(reg["covers"]) == 12
    return float(lens[full].sum() * (maxl + 1))


def pack_regions(regs):
    """region jobs -> one uint8 buffer (what travels between ranks): header, per-region scalars and per-read arrays, then the bases and qualities"""
    head = [np.array([MAGIC, len(regs)], np.int64)]
    body = []
    for r in regs:
        n = len(r["seqs"])
        lens = np.array([len(s) for s in r["seqs"]], np.int32)
        head.append(np.array([int(r["reg_len"]), n, len(r["ref"])], np.int64))
        body += [np.asarray(r["read_ids"], np.int32).view(np.uint8), np.asarray(r["covers"], np.int32).view(np.uint8), np.asarray(r["haps"], np.int32).view(np.uint8),
                 np.asarray(r["phase_sets"], np.int64).view(np.uint8), lens.view(np.uint8), np.asarray(r["ref"], np.uint8)]
        body += [np.asarray(s, np.uint8) for s in r["seqs"]]
        q = r.get("quals")
        body += [np.asarray(x, np.uint8) for x in q] if q is not None else [np.zeros(int(l), np.uint8) for l in lens]
    h = np.concatenate(head).view(np.uint8)
    return np.concatenate([h] + body) if body else h


def unpack_regions(buf):
    buf = np.ascontiguousarray(buf, np.uint8)
    magic, n_regs = (int(x) for x in buf[:16].view(np.int64))
    assert magic == MAGIC, "not a packed region buffer"
    scal = buf[16:16 + 24 * n_regs].view(np.int64).reshape(n_regs, 3)
    o = 16 + 24 * n_regs
    out = []

    def take(nbytes, dt):
        nonlocal o
        a = buf[o:o + nbytes].view(dt).copy(); o += nbytes
        return a
    for reg_len, n, ref_len in scal:
        n = int(n)
        ids, cov, haps = take(4 * n, np.int32), take(4 * n, np.int32), take(4 * n, np.int32)
        ps, lens, ref = take(8 * n, np.int64), take(4 * n, np.int32), take(int(ref_len), np.uint8)
        seqs = [take(int(l), np.uint8) for l in lens]
        quals = [take(int(l), np.uint8) for l in lens]
        out.append(dict(reg_len=int(reg_len), read_ids=ids, seqs=seqs, quals=quals, covers=cov, haps=haps, phase_sets=ps, ref=ref))
    assert o == len(buf)
    return out


def plan_moves(costs_per_rank, tol=0.02, max_moves=None):
    """deterministic greedy plan: repeatedly move, from the most loaded rank to the least loaded one, the job that brings the pair closest to equal
    (a job moves at most once).  -> list of (src_rank, index in src's queue, dst_rank), loads before, loads after"""
    load = [float(sum(c)) for c in costs_per_rank]
    before = list(load)
    moved = [set() for _ in costs_per_rank]
    mean = sum(load) / max(len(load), 1)
    moves = []
    while mean > 0 and (max(load) - mean) / mean > tol and (max_moves is None or len(moves) < max_moves):
        src = max(range(len(load)), key=lambda r: (load[r], -r))
        dst = min(range(len(load)), key=lambda r: (load[r], r))
        gap = load[src] - load[dst]
        best, best_c = -1, 0.0
        for i, c in enumerate(costs_per_rank[src]):        # the job closest to half the gap; anything >= the gap would only swap the roles
            if i in moved[src] or c <= 0 or c >= gap:
                continue
            if best < 0 or abs(c - gap / 2) < abs(best_c - gap / 2):
                best, best_c = i, c
        if best < 0:
            break
        moved[src].add(best); moves.append((src, best, dst)); load[src] -= best_c; load[dst] += best_c
    return moves, before, load


def rebalance(queue, group=None, device=None, tol=0.02):
    """queue: list of (cost, packed uint8 buffer) of THIS rank.  One epoch: all_gather of (cost, nbytes) per job, the common plan, batched point-to-point
    transfers of whole buffers.  Returns (new queue, stats).  torch.distributed must be initialised (nccl = RCCL: pass the rank's device)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = [(float(c), int(len(b))) for c, b in queue]
    allq = [None] * world
    dist.all_gather_object(allq, mine, group=group)
    moves, before, after = plan_moves([[c for c, _ in q] for q in allq], tol)
    ops, recv_bufs, sent = [], [], set()
    # the plan is in group-relative ranks (all_gather_object, get_rank(group)); the positional peer of a P2POp is a GLOBAL rank
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    to_dev = (lambda t: t.to(device)) if device is not None else (lambda t: t)
    keep_alive = []
    for tag, (src, i, dst) in enumerate(moves):
        nbytes = allq[src][i][1]
        if rank == src:
            t = to_dev(torch.from_numpy(np.ascontiguousarray(queue[i][1])))
            keep_alive.append(t); sent.add(i)
            ops.append(dist.P2POp(dist.isend, t, peer(dst), group=group))
        elif rank == dst:
            t = torch.empty(nbytes, dtype=torch.uint8, device=device if device is not None else "cpu")
            recv_bufs.append((allq[src][i][0], t))
            ops.append(dist.P2POp(dist.irecv, t, peer(src), group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if device is not None:
            torch.cuda.synchronize(device)
    new_q = [q for i, q in enumerate(queue) if i not in sent] + [(c, t.cpu().numpy()) for c, t in recv_bufs]
    moved_bytes = sum(allq[s][i][1] for s, i, _ in moves)
    mean = sum(before) / world if world else 0.0
    stats = dict(n_moves=len(moves), moved_bytes=int(moved_bytes), imbalance_before=(max(before) / mean if mean > 0 else 1.0),
                 imbalance_after=(max(after) / mean if mean > 0 else 1.0), loads_before=before, loads_after=after)
    return new_q, stats
