import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def lcd():
    """the HIP library through its Python mirror; fails loudly if liblcd_hotpath.so is missing"""
    from longcalld_amd import align
    align.load_library()
    return align


def _oracle_chunk(a):
    """one worker of oracle_many: a slice of a seeded workload through the oracle (spawned process: no HIP state is inherited)"""
    seed, n, shape_name, lo, hi = a
    from longcalld_amd import jobs
    from oracle import pyoracle
    regs = jobs.make_regions(seed, n, jobs.SHAPES[shape_name])
    return [pyoracle.collect_noisy_reg_aln_strs(regs[i]) for i in range(lo, hi)]


def oracle_many(seed, n, shape_name, workers=0):
    """the oracle's result for every region of jobs.make_regions(seed, n, shape) on all host cores: full-size batches of the noisy-read shapes take minutes on one
    core (a 10 kb SV region alone ~10 s).  Regions are dealt out in strides of 8 so that the expensive ones spread over the workers"""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:  # noqa
        cores = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, -(-int(q) // int(per))))
    except Exception:  # noqa
        pass
    w = workers or max(1, min(cores, 16))
    step = 8
    tasks = [(seed, n, shape_name, lo, min(n, lo + step)) for lo in range(0, n, step)]
    with mp.get_context("spawn").Pool(w) as pool:
        parts = pool.map(_oracle_chunk, tasks, chunksize=1)
    return [e for part in parts for e in part]


def _oracle_poa_task(t):
    """worker of oracle_poa_many (spawned: imports the oracle itself)"""
    kind, reads, extra = t
    sys.path.insert(0, ROOT)
    from oracle import pyoracle
    if kind == "k2":
        return pyoracle.poa_aln_msa_cons(reads, extra)
    return pyoracle.poa_partial_aln_msa_cons(reads, extra)


def oracle_poa_many(tasks, workers=0):
    """the oracle's K1 / K2 chains for [(kind, reads, max_n_cons | covers)], each in its own spawned process (a 16 - 45 kb K2 chain is minutes of scalar DP and
    3 - 25 GB of matrices: side by side instead of one after the other, and the memory goes back when the worker ends)"""
    import multiprocessing as mp
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:  # noqa
        cores = os.cpu_count() or 1
    w = workers or max(1, min(cores, len(tasks)))
    with mp.get_context("spawn").Pool(w, maxtasksperchild=1) as pool:
        return pool.map(_oracle_poa_task, tasks, chunksize=1)


def mutate(rng, s, rate, sv=0.0):
    out, i = [], 0
    while i < len(s):
        x = rng.random()
        if x < rate / 3:
            out.append(rng.integers(0, 4))
        elif x < 2 * rate / 3:
            out.append(s[i]); out.append(rng.integers(0, 4))
        elif x < rate:
            pass
        elif x < rate + sv:
            L = int(rng.integers(5, 80))
            if rng.random() < 0.5:
                out.extend(rng.integers(0, 4, L)); out.append(s[i])
            else:
                i += L
        else:
            out.append(s[i])
        i += 1
    return np.array(out, dtype=np.uint8)


def same_result(a, b):
    """compare two region results (oracle vs HIP): n_cons, cluster membership, every alignment string and coordinate"""
    assert a["n_cons"] == b["n_cons"]
    if a["n_cons"] == 0:
        return
    for c in range(a["n_cons"]):
        assert a["clu_n_seqs"][c] == b["clu_n_seqs"][c]
        assert (a["clu_read_ids"][c] == b["clu_read_ids"][c]).all()
        for j, (x, y) in enumerate(zip(a["aln_strs"][c], b["aln_strs"][c])):
            assert (x is None) == (y is None), (c, j)
            if x is None:
                continue
            for k in ("aln_len", "target_beg", "target_end", "query_beg", "query_end"):
                assert x[k] == y[k], (c, j, k, x[k], y[k])
            assert (x["target"] == y["target"]).all() and (x["query"] == y["query"]).all(), (c, j)


def check_invariants(reg, res):
    """size-independent properties of a region result: clusters partition reads, every string de-gaps to its inputs"""
    if res["n_cons"] == 0:
        return 0
    ids = list(reg["read_ids"])
    seen = []
    n_str = 0
    for c in range(res["n_cons"]):
        rc = res["aln_strs"][c][0]
        assert rc is not None and len(rc["target"]) == rc["aln_len"] == len(rc["query"])
        assert (rc["target"][rc["target"] != 5] == reg["ref"]).all()          # ref row of ref<->cons de-gaps to the reference slice
        cons = rc["query"][rc["query"] != 5]
        assert not ((rc["target"] == 5) & (rc["query"] == 5)).any()
        members = res["clu_read_ids"][c]
        assert len(members) == res["clu_n_seqs"][c]
        seen += list(members)
        for j, rid in enumerate(members):
            s = res["aln_strs"][c][2 * j + 1]
            if s is None:
                continue
            n_str += 1
            read = reg["seqs"][ids.index(rid)]
            q = s["query"][s["query"] != 5]
            t = s["target"][s["target"] != 5]
            assert len(s["target"]) == s["aln_len"]
            if reg["covers"][ids.index(rid)] == 12:                            # full cover: the rows are the whole consensus and the whole read
                assert q.tobytes() == read.tobytes() and t.tobytes() == cons.tobytes()
            else:   # partial cover: the row may carry the anchor node's base at either end (the sub-graph alignment labels the edge out of
                    # the anchor node with the read, src/align.c:797-803), everything between is the read
                assert q[1:-1].tobytes() in read.tobytes() and t.tobytes() in cons.tobytes()
    assert len(seen) == len(set(seen)) and set(seen) <= set(ids)
    return n_str
