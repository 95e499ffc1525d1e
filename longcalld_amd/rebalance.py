"""Region-queue rebalancing across ranks: the only use of a collective library on this path.

Chunks are independent until stitch_var_main (src/collect_var.c:2983), so no data-path collective exists; what can go wrong across the 8 GPUs of a node
is load: the host enumerates chunks in genome order (src/call_var_main.c:599-621) and hands contiguous blocks to the devices, and SV-heavy chunks
(configs[4]) make blocks unequal.  One epoch here = all_gather of every rank's queue (cost + payload size per job), a deterministic plan computed
identically on every rank, and point-to-point sends of WHOLE packed job buffers from the deepest queues to the shallowest -- torch.distributed, i.e.
RCCL over xGMI with backend "nccl" on the GPU node (device tensors) and gloo in the CPU tests (host tensors).  Job buffers are 15 KB - a few MB, so
ring / per-link bandwidth is irrelevant; what matters is that a job moves at most once per epoch and that every rank agrees on the plan.
"""
import ctypes as C

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


class _RegionJob(C.Structure):
    _fields_ = [("reg_len", C.c_int64), ("n_reads", C.c_int), ("read_ids", C.POINTER(C.c_int)), ("lens", C.POINTER(C.c_int)), ("seqs", C.POINTER(C.POINTER(C.c_uint8))),
                ("quals", C.POINTER(C.POINTER(C.c_uint8))), ("fully_covers", C.POINTER(C.c_int)), ("haps", C.POINTER(C.c_int)), ("phase_sets", C.POINTER(C.c_int64)),
                ("ref_seq", C.POINTER(C.c_uint8)), ("ref_seq_len", C.c_int)]


class _Move(C.Structure):
    _fields_ = [("src", C.c_int), ("index", C.c_int), ("dst", C.c_int)]


def _clib():
    from ._lib import load_library
    lib = load_library()
    if not getattr(lib, "_rb_ready", False):
        lib.lcd_region_job_cost.restype = C.c_double
        lib.lcd_region_job_cost.argtypes = [C.POINTER(_RegionJob)]
        lib.lcd_region_jobs_pack.restype = C.c_uint64
        lib.lcd_region_jobs_pack.argtypes = [C.c_int, C.POINTER(_RegionJob), C.c_void_p]
        lib.lcd_batch_add_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        lib.lcd_rebalance_plan.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(_Move), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        lib.lcd_rebalance_last_error.restype = C.c_char_p
        lib._rb_ready = True
    return lib


def _c_jobs(regs):
    """the regions as lcd_region_job_t records (+ the arrays they point into, which the caller keeps alive)"""
    keep, jobs = [], (_RegionJob * max(len(regs), 1))()
    u8pp = C.POINTER(C.c_uint8)
    for k, r in enumerate(regs):
        n = len(r["seqs"])
        ids = np.ascontiguousarray(r["read_ids"], np.int32); cov = np.ascontiguousarray(r["covers"], np.int32); haps = np.ascontiguousarray(r["haps"], np.int32)
        ps = np.ascontiguousarray(r["phase_sets"], np.int64); lens = np.array([len(x) for x in r["seqs"]], np.int32); ref = np.ascontiguousarray(r["ref"], np.uint8)
        seqs = [np.ascontiguousarray(x, np.uint8) for x in r["seqs"]]
        q = r.get("quals")
        quals = [np.ascontiguousarray(x, np.uint8) for x in q] if q is not None else None
        sp = (u8pp * max(n, 1))(*[x.ctypes.data_as(u8pp) for x in seqs])
        qp = (u8pp * max(n, 1))(*[x.ctypes.data_as(u8pp) for x in quals]) if quals is not None else None
        keep += [ids, cov, haps, ps, lens, ref, seqs, quals, sp, qp]
        j = jobs[k]
        j.reg_len, j.n_reads, j.ref_seq_len = int(r["reg_len"]), n, len(ref)
        j.read_ids = ids.ctypes.data_as(C.POINTER(C.c_int)); j.lens = lens.ctypes.data_as(C.POINTER(C.c_int)); j.seqs = sp
        j.quals = qp if qp is not None else C.POINTER(u8pp)()
        j.fully_covers = cov.ctypes.data_as(C.POINTER(C.c_int)); j.haps = haps.ctypes.data_as(C.POINTER(C.c_int)); j.phase_sets = ps.ctypes.data_as(C.POINTER(C.c_int64))
        j.ref_seq = ref.ctypes.data_as(u8pp)
    return jobs, keep


def region_cost_c(reg):
    """lcd_region_job_cost (liblcd_hotpath.so): the library's own price of a region job"""
    jobs, keep = _c_jobs([reg])
    return float(_clib().lcd_region_job_cost(C.byref(jobs[0])))


def pack_regions_c(regs):
    """lcd_region_jobs_pack (liblcd_hotpath.so): the same bytes as pack_regions"""
    lib = _clib()
    jobs, keep = _c_jobs(regs)
    n = int(lib.lcd_region_jobs_pack(len(regs), jobs, None))
    buf = np.zeros(n, np.uint8)
    assert int(lib.lcd_region_jobs_pack(len(regs), jobs, buf.ctypes.data_as(C.c_void_p))) == n
    return buf


def plan_moves(costs_per_rank, tol=0.02, max_moves=None):
    """lcd_rebalance_plan (liblcd_hotpath.so): the plan every rank computes from the gathered queues.  -> list of (src_rank, index in src's queue, dst_rank),
    loads before, loads after.  (plan_moves_py below is the same rule in Python: the tests hold the two together.)"""
    lib = _clib()
    world = len(costs_per_rank)
    nj = np.array([len(c) for c in costs_per_rank], np.int32)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64) for c in costs_per_rank]) if nj.sum() else np.zeros(1), np.float64)
    moves = (_Move * (int(nj.sum()) + 1))()
    lb, la = np.zeros(max(world, 1), np.float64), np.zeros(max(world, 1), np.float64)
    n = lib.lcd_rebalance_plan(world, nj.ctypes.data_as(C.POINTER(C.c_int)), flat.ctypes.data_as(C.POINTER(C.c_double)), float(tol), -1 if max_moves is None else int(max_moves),
                               moves, lb.ctypes.data_as(C.POINTER(C.c_double)), la.ctypes.data_as(C.POINTER(C.c_double)))
    return [(int(moves[i].src), int(moves[i].index), int(moves[i].dst)) for i in range(n)], [float(x) for x in lb[:world]], [float(x) for x in la[:world]]


def plan_moves_py(costs_per_rank, tol=0.02, max_moves=None):
    """deterministic greedy plan: repeatedly move, from the most loaded rank to the least loaded one, the job that brings the pair closest to equal -- or, when
    every job of the most loaded rank is at least as large as the gap, swap the pair of jobs whose difference does (a job moves at most once).  -> list of (src_rank, index in src's queue, dst_rank), loads before, loads after"""
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
            # nothing of src's is smaller than the gap: a swap -- src's job a for dst's job b with 0 < a - b < gap, the difference closest to half the gap
            ba, bb, bd = -1, -1, 0.0
            if max_moves is None or len(moves) + 2 <= max_moves:
                for i, a in enumerate(costs_per_rank[src]):
                    if i in moved[src] or a <= 0:
                        continue
                    for j, b in enumerate(costs_per_rank[dst]):
                        if j in moved[dst] or b <= 0:
                            continue
                        d = a - b
                        if d <= 0 or d >= gap:
                            continue
                        if ba < 0 or abs(d - gap / 2) < abs(bd - gap / 2):
                            ba, bb, bd = i, j, d
            if ba < 0:
                break
            moved[src].add(ba); moved[dst].add(bb)
            moves.append((src, ba, dst)); moves.append((dst, bb, src)); load[src] -= bd; load[dst] += bd
            continue
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


class _RbStats(C.Structure):
    _fields_ = [("n_moves", C.c_int), ("world", C.c_int), ("moved_bytes", C.c_uint64), ("imbalance_before", C.c_double), ("imbalance_after", C.c_double),
                ("load_before_mine", C.c_double), ("load_after_mine", C.c_double), ("jobs_before_mine", C.c_int), ("jobs_after_mine", C.c_int)]


class Comm:
    """lcd_comm_create: an RCCL communicator owned by the library (librccl opened lazily).  `uid` = the 128 bytes of lcd_rccl_unique_id made on rank 0 and handed
    to the others by the caller (bench.py: torch.distributed's broadcast_object_list)."""

    def __init__(self, world, rank, uid, device):
        lib = _clib()
        lib.lcd_comm_create.restype = C.c_void_p
        lib.lcd_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        lib.lcd_comm_destroy.argtypes = [C.c_void_p]; lib.lcd_comm_destroy.restype = None
        self.lib, self.world, self.rank = lib, world, rank
        self.h = lib.lcd_comm_create(int(world), int(rank), bytes(uid), int(device))
        if not self.h:
            raise RuntimeError("lcd_comm_create: " + lib.lcd_rebalance_last_error().decode())

    @staticmethod
    def unique_id():
        lib = _clib()
        buf = C.create_string_buffer(128)
        lib.lcd_rccl_unique_id.argtypes = [C.c_char_p]
        if lib.lcd_rccl_unique_id(buf) != 0:
            raise RuntimeError("lcd_rccl_unique_id: " + lib.lcd_rebalance_last_error().decode())
        return buf.raw

    def info(self):
        """what RCCL itself reports for this communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)"""
        w, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        self.lib.lcd_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if self.lib.lcd_comm_info(self.h, C.byref(w), C.byref(r), C.byref(d)) != 0:
            raise RuntimeError("lcd_comm_info: " + self.lib.lcd_rebalance_last_error().decode())
        return dict(nccl_world=w.value, nccl_rank=r.value, nccl_device=d.value)

    def close(self):
        if self.h:
            self.lib.lcd_comm_destroy(self.h); self.h = None


def rebalance_c(comm, queue, tol=0.02):
    """One epoch through the LIBRARY (lcd_rebalance_exchange: ncclAllGather of the queue depths and (cost, bytes) tables, lcd_rebalance_plan, ncclSend / ncclRecv of
    whole packed buffers).  queue: list of (cost, packed uint8 buffer) of this rank -> (new queue, stats)."""
    lib = comm.lib
    n = len(queue)
    cost = np.array([c for c, _ in queue] or [0.0], np.float64)
    bufs = [np.ascontiguousarray(b, np.uint8) for _, b in queue]
    nb = np.array([len(b) for b in bufs] or [0], np.uint64)
    u8p = C.POINTER(C.c_uint8)
    bp = (u8p * max(n, 1))(*[b.ctypes.data_as(u8p) for b in bufs])
    n_out = C.c_int(0); co = C.POINTER(C.c_double)(); no = C.POINTER(C.c_uint64)(); bo = C.POINTER(u8p)(); ow = u8p(); st = _RbStats()
    lib.lcd_rebalance_exchange.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(u8p), C.c_double, C.POINTER(C.c_int),
                                           C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(u8p)), C.POINTER(u8p), C.POINTER(_RbStats)]
    rc = lib.lcd_rebalance_exchange(comm.h, n, cost.ctypes.data_as(C.POINTER(C.c_double)), nb.ctypes.data_as(C.POINTER(C.c_uint64)), bp, float(tol), C.byref(n_out),
                                    C.byref(co), C.byref(no), C.byref(bo), C.byref(ow), C.byref(st))
    if rc != 0:
        raise RuntimeError("lcd_rebalance_exchange: " + lib.lcd_rebalance_last_error().decode())
    libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]
    new_q = []
    for i in range(n_out.value):
        b = np.ctypeslib.as_array(bo[i], shape=(max(int(no[i]), 1),))[:int(no[i])].copy()
        new_q.append((float(co[i]), b))
        if ow[i]:
            libc.free(C.cast(bo[i], C.c_void_p))
    for p_ in (co, no, bo, ow):
        libc.free(C.cast(p_, C.c_void_p))
    stats = dict(n_moves=st.n_moves, moved_bytes=int(st.moved_bytes), imbalance_before=st.imbalance_before, imbalance_after=st.imbalance_after,
                 load_before_mine=st.load_before_mine, load_after_mine=st.load_after_mine, jobs_before_mine=st.jobs_before_mine, jobs_after_mine=st.jobs_after_mine,
                 transport="librccl via liblcd_hotpath.so (lcd_rebalance_exchange)")
    return new_q, stats
