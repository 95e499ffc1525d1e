"""SURVEY 8e on the GPU box: the library's side of the queue rebalance.  lcd_batch_add_packed (a packed chunk buffer into a batch == the regions added one by one)
and the RCCL epoch lcd_rebalance_exchange with the one rank a single-GPU box has (librccl opened, communicator made, both all-gathers run, no move, the queue comes
back byte for byte).  The N > 1 plan and wire format are covered on CPU (tests/test_dist_cpu.py)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_packed_chunk_into_a_batch_equals_regions_added_one_by_one():
    from longcalld_amd import align, jobs, rebalance as rb
    regs = jobs.make_regions(321, 40, jobs.HIFI) + jobs.make_regions(322, 3, jobs.SV, poisson_sv=False)
    a = align.RegionBatch()
    for r in regs:
        a.add_region(r)
    a.upload(); a.run(); a.download()
    buf = rb.pack_regions_c(regs)
    b = align.RegionBatch()
    n = rb._clib().lcd_batch_add_packed(b.h, buf.ctypes.data_as(C.c_void_p), len(buf))
    assert n == len(regs), rb._clib().lcd_rebalance_last_error()
    b.n_reads = [len(r["seqs"]) for r in regs]
    b.upload(); b.run(); b.download()
    assert a.digest() == b.digest()
    # malformed buffers are refused
    assert rb._clib().lcd_batch_add_packed(b.h, buf.ctypes.data_as(C.c_void_p), len(buf) - 5) < 0
    bad = buf.copy(); bad[0] ^= 1
    assert rb._clib().lcd_batch_add_packed(b.h, bad.ctypes.data_as(C.c_void_p), len(bad)) < 0
    a.close(); b.close()


def test_rccl_epoch_with_one_rank():
    from longcalld_amd import jobs, rebalance as rb
    chunks = [jobs.make_regions(400 + i, 4, jobs.HIFI) for i in range(5)]
    queue = [(sum(rb.region_cost(r) for r in regs), rb.pack_regions_c(regs)) for regs in chunks]
    comm = rb.Comm(1, 0, rb.Comm.unique_id(), 0)
    new_q, st = rb.rebalance_c(comm, queue)
    comm.close()
    assert st["n_moves"] == 0 and st["moved_bytes"] == 0 and st["jobs_before_mine"] == st["jobs_after_mine"] == 5
    assert abs(st["imbalance_before"] - 1.0) < 1e-12 and abs(st["load_before_mine"] - sum(c for c, _ in queue)) < 1e-6
    assert len(new_q) == 5 and all(c0 == c1 and (b0 == b1).all() for (c0, b0), (c1, b1) in zip(queue, new_q))
