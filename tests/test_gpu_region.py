"""Region-level parity: lcd_batch_* (anchors -> POA chains -> ref/cons WFA -> strings) vs the oracle's restatement of
collect_noisy_reg_aln_strs (src/align.c:1760) on seeded synthetic region jobs, HiFi and ONT shapes."""
import numpy as np
import pytest

from conftest import check_invariants as _check_invariants, same_result

pytestmark = pytest.mark.gpu


def _run_batch(lcd, regs, opt=None):
    b = lcd.RegionBatch(opt)
    for r in regs:
        b.add_region(r)
    b.upload(); b.run(); b.download()
    out = [b.result(i) for i in range(len(regs))]
    ids = [b.sorted_ids(i) for i in range(len(regs))]
    st = b.stats(); dg = b.digest()
    b.close()
    return out, ids, st, dg


@pytest.mark.parametrize("shape,seed,n", [("hifi", 1, 48), ("ont", 2, 16)])
def test_regions_match_oracle(lcd, oracle, shape, seed, n):
    from longcalld_amd import jobs
    regs = jobs.make_regions(seed, n, jobs.HIFI if shape == "hifi" else jobs.ONT)
    got, ids, st, _ = _run_batch(lcd, regs)
    n_res = 0
    for r, g, sid in zip(regs, got, ids):
        exp = oracle.collect_noisy_reg_aln_strs(r)
        assert (sid == exp["sorted_ids"]).all()  # the in-place permutation of noisy_reads (src/align.c:1774)
        same_result(exp, g)
        n_res += g["n_cons"] > 0
    assert st["n_regions_resolved"] == n_res and st["poa_aligned_bases"] > 0


def test_edge_regions(lcd, oracle):
    """empty / ragged inputs the reference guards: too few reads, no full-cover read, partial-only haplotype, zero-length slices"""
    from longcalld_amd import jobs
    rng = np.random.default_rng(5)
    base = jobs.make_region(rng, jobs.HIFI, length=200, n_reads=12)
    few = jobs.make_region(rng, jobs.HIFI, length=150, n_reads=5)
    few["haps"][:] = 0; few["phase_sets"][:] = -1
    few["covers"][:2] = 8  # only 3 full-cover reads < min_dp: region skipped (src/align.c:1794)
    nofull = jobs.make_region(rng, jobs.HIFI, length=150, n_reads=8)
    nofull["covers"][:] = 8
    zero = jobs.make_region(rng, jobs.HIFI, length=180, n_reads=10)
    zero["seqs"][3] = np.zeros(0, np.uint8)
    part = jobs.make_region(rng, jobs.HIFI, length=400, n_reads=14)
    for i in range(0, 14, 3):
        s = part["seqs"][i]
        if len(s) > 120:
            part["seqs"][i] = s[:len(s) // 2]; part["covers"][i] = 8
    for i in range(1, 14, 5):
        s = part["seqs"][i]
        if len(s) > 120:
            part["seqs"][i] = s[len(s) // 3:]; part["covers"][i] = 4
    regs = [base, few, nofull, zero, part]
    got, ids, _, _ = _run_batch(lcd, regs)
    for r, g in zip(regs, got):
        same_result(oracle.collect_noisy_reg_aln_strs(r), g)
    assert got[1]["n_cons"] == 0 and got[2]["n_cons"] == 0


def test_long_region_sampling_path(lcd, oracle):
    """regions >= min_noisy_reg_size_to_sample_reads (10 kb): the anchor step samples the full read slices with edlib first
    (collect_partial_aln_beg_end, src/align.c:719-728) -- phased reads, some partially covering: == oracle"""
    from longcalld_amd import jobs
    rng = np.random.default_rng(77)
    reg = jobs.make_region(rng, jobs.HIFI, length=10300, n_reads=10)
    n = len(reg["seqs"])
    reg["haps"][:] = np.arange(n) % 2 + 1
    reg["phase_sets"][:] = 4242
    for i in (2, 5):   # a left-cover and a right-cover read
        s = reg["seqs"][i]
        if i == 2:
            reg["seqs"][i] = s[: len(s) * 2 // 3]; reg["covers"][i] = 8
        else:
            reg["seqs"][i] = s[len(s) // 3:]; reg["covers"][i] = 4
        reg["quals"][i] = np.full(len(reg["seqs"][i]), 30, np.uint8)
    got, ids, _, _ = _run_batch(lcd, [reg])
    exp = oracle.collect_noisy_reg_aln_strs(reg)
    assert (ids[0] == exp["sorted_ids"]).all()
    same_result(exp, got[0])
    assert got[0]["n_cons"] == 2


def test_run_is_idempotent_and_order_independent(lcd):
    """size-independent properties: re-running a batch gives the same digest; a region's result does not depend on its batch mates"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(9, 24)
    b = lcd.RegionBatch()
    for r in regs:
        b.add_region(r)
    b.upload(); b.run(); b.download(); d1 = b.digest()
    b.run(); b.download(); d2 = b.digest()
    one = b.result(5)
    b.close()
    assert d1 == d2 and d1 != 0
    got, _, _, _ = _run_batch(lcd, [regs[5]])
    same_result(one, got[0])


@pytest.mark.parametrize("shape,seed,n", [("hifi", 41, 20), ("ont", 42, 8)])
def test_ref_read_strings_match_oracle(lcd, oracle, shape, seed, n):
    """collect_ref_read_aln_str (--refine-aln -b / -s): make_ref_read_aln_str (src/align.c:1056-1146) composes ref<->cons with cons<->read
    and re-aligns the stretches where both have an insertion against the consensus -- every aln_strs[c][2k+2] == oracle"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(seed, n, jobs.HIFI if shape == "hifi" else jobs.ONT)
    o_dev = lcd.default_opt(); o_dev.collect_ref_read_aln_str = 1
    o_cpu = oracle.default_opt(); o_cpu.collect_ref_read_aln_str = 1
    got, _, _, _ = _run_batch(lcd, regs, o_dev)
    n_rr = 0
    for r, g in zip(regs, got):
        exp = oracle.collect_noisy_reg_aln_strs(r, o_cpu)
        same_result(exp, g)
        for c in range(g["n_cons"]):
            n_rr += sum(1 for j, a in enumerate(g["aln_strs"][c]) if j >= 2 and j % 2 == 0 and a is not None)
    assert n_rr > 0


def test_run_many_equals_separate_runs(lcd):
    """lcd_batch_run_many (one launch set per stage over the chains of several batches) == each batch run on its own"""
    from longcalld_amd import jobs
    sets = [jobs.make_regions(31, 14), jobs.make_regions(32, 9, jobs.ONT), jobs.make_regions(33, 11)]
    alone = []
    for regs in sets:
        b = lcd.RegionBatch()
        for r in regs:
            b.add_region(r)
        b.upload(); b.run(); b.download()
        alone.append((b.digest(), [b.result(i) for i in range(len(regs))]))
        b.close()
    bs = []
    for regs in sets:
        b = lcd.RegionBatch()
        for r in regs:
            b.add_region(r)
        b.upload()
        bs.append(b)
    lcd.RegionBatch.run_many(bs)
    for b in bs:          # (the ref<->cons rows live in the leader's buffers: download everything before the leader runs again or is closed)
        b.download()
    for b, (dig, res) in zip(bs, alone):
        assert b.digest() == dig
        for i, r in enumerate(res):
            same_result(r, b.result(i))
        st = b.stats()
        assert st["n_regions"] > 0 and st["ms_poa_kernel"] > 0
    for b in bs:
        b.close()


def test_results_arena_equals_per_region_results(lcd):
    """lcd_batch_region_results_arena (one host block per batch, interior pointers, filled by host threads) hands out exactly what lcd_batch_region_result
    hands out region by region -- cluster id lists, every aln_str_t slot (set or NULL), rows and coordinates -- on phased (K1) and unphased (K2) regions, with and
    without the ref<->read strings"""
    from longcalld_amd import jobs
    for rr_on in (0, 1):
        opt = lcd.default_opt(); opt.collect_ref_read_aln_str = rr_on
        regs = jobs.make_regions(77 + rr_on, 40)
        b = lcd.RegionBatch(opt)
        for r in regs:
            b.add_region(r)
        b.upload(); b.run(); b.download()
        arena = b.results_arena()
        assert len(arena) == len(regs)
        n_rows = 0
        for i in range(len(regs)):
            same_result(b.result(i), arena[i])
            n_rows += sum(x is not None for c in range(2) for x in arena[i]["aln_strs"][c])
        assert n_rows > 500
        n, nbytes = b.results_arena(parse=False)
        assert n == len(regs) and nbytes > 100000
        b.close()


def test_per_call_mirror(lcd, oracle):
    """lcd_collect_noisy_reg_aln_strs through chunk views: digar walk (src/align.c:1392-1458) + 4-bit base unpacking + in-place permutation"""
    import ctypes as C
    from longcalld_amd import _lib, jobs
    lib = _lib.load_library()
    rng = np.random.default_rng(21)
    reg = jobs.make_region(rng, jobs.HIFI, length=300, n_reads=14)
    # embed every slice in a longer read: 50 flanking bases each side, one '=' digar covering it, region = [1051, 1050+len(ref)]
    n = len(reg["seqs"])
    views = (_lib.LcdReadView * n)()
    keep = []
    reg_beg, reg_end = 1051, 1050 + len(reg["ref"])
    exp_cover = []
    for i, s in enumerate(reg["seqs"]):
        cover = int(reg["covers"][i])
        left = rng.integers(0, 4, 50).astype(np.uint8) if cover & 8 else np.zeros(0, np.uint8)
        right = rng.integers(0, 4, 50).astype(np.uint8) if cover & 4 else np.zeros(0, np.uint8)
        full = np.concatenate([left, s, right])
        # BAM 4-bit packing: A1 C2 G4 T8
        code = np.array([1, 2, 4, 8, 15], np.uint8)[full]
        if len(code) % 2:
            code = np.append(code, 0)
        packed = ((code[0::2] << 4) | code[1::2]).astype(np.uint8)
        qual = np.full(len(full), 30, np.uint8)
        # reference span of the read: cover both ends -> starts 50 before the region; the read's ref length is faked with one '=' op
        # of the right length so that boundaries fall where the slice is
        if cover == 12:
            pos, rlen = reg_beg - 50, 50 + (reg_end - reg_beg + 1) + 50
        elif cover == 8:
            pos, rlen = reg_beg - 50, 50 + len(s)
        else:
            pos, rlen = reg_end - len(s) + 1, len(s) + 50
        # a single EQUAL digar cannot express length differences; build: '=' up to the region start, then an 'I'/'D' free walk is not
        # needed because lcd only uses (pos, len, qi) of the digars that contain the boundaries.
        d = (_lib.LcdDigar1 * 3)()
        if cover == 12:
            d[0] = _lib.LcdDigar1(pos, 7, 50 + 1, 0)                                        # '=' containing reg_beg -> read_beg = 50
            d[1] = _lib.LcdDigar1(reg_beg + 1, 1, max(len(s) - 2, 0), 51)                    # 'I' (body of the slice), ref length 0
            d[2] = _lib.LcdDigar1(reg_end, 7, 51, 50 + len(s) - 1)                          # '=' containing reg_end -> read_end
            nd = 3
        elif cover == 8:
            d[0] = _lib.LcdDigar1(pos, 7, 50 + 1, 0)
            d[1] = _lib.LcdDigar1(reg_beg + 1, 1, len(s) - 1, 51)
            nd = 2
        else:
            d[0] = _lib.LcdDigar1(reg_end - 1, 1, len(s) - 1, 0)
            d[1] = _lib.LcdDigar1(reg_end, 7, 51, len(s) - 1)
            nd = 2
        keep += [packed, qual, d]
        views[i] = _lib.LcdReadView(C.cast(d, C.POINTER(_lib.LcdDigar1)), nd, len(full), packed.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    qual.ctypes.data_as(C.POINTER(C.c_uint8)), int(reg["haps"][i]), int(reg["phase_sets"][i]))
        exp_cover.append(cover)
    opt = lcd.default_opt()
    noisy = np.arange(n, dtype=np.int32)
    m = 1 + 2 * n
    clu_n = (C.c_int * 2)(0, 0)
    clu_ids = (C.POINTER(C.c_int) * 2)()
    a0, a1 = (_lib.LcdAlnStr * m)(), (_lib.LcdAlnStr * m)()
    arr = (C.POINTER(_lib.LcdAlnStr) * 2)(C.cast(a0, C.POINTER(_lib.LcdAlnStr)), C.cast(a1, C.POINTER(_lib.LcdAlnStr)))
    ref = np.ascontiguousarray(reg["ref"])
    nc = lib.lcd_collect_noisy_reg_aln_strs(C.byref(opt), views, reg_beg, reg_end, 0, n, noisy.ctypes.data_as(C.POINTER(C.c_int)),
                                            ref.ctypes.data_as(C.POINTER(C.c_uint8)), len(ref), clu_n, clu_ids, arr)
    assert nc >= 0, lib.lcd_last_error()
    reg2 = dict(reg); reg2["read_ids"] = np.arange(n, dtype=np.int32)
    exp = oracle.collect_noisy_reg_aln_strs(reg2)
    assert nc == exp["n_cons"]
    assert (noisy == exp["sorted_ids"]).all()
    for c in range(nc):
        assert clu_n[c] == exp["clu_n_seqs"][c]
        s = (a0, a1)[c][0]
        t = np.ctypeslib.as_array(s.target_aln, shape=(s.aln_len,))
        assert (t == exp["aln_strs"][c][0]["target"]).all()


def test_full_size_batch_properties(lcd):
    """BASELINE configs[1] at full size (1 250 regions, ~37 000 strings): invariants on every region, and the digest does not depend on
    how the batch is submitted (alone / jointly with a copy of itself)"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(1000, jobs.regions_for_ref_mb(10), jobs.HIFI)
    bs = []
    for _ in range(2):
        b = lcd.RegionBatch()
        for r in regs:
            b.add_region(r)
        b.upload(); bs.append(b)
    bs[0].run(); bs[0].download()
    d0 = bs[0].digest()
    n_str = sum(_check_invariants(r, bs[0].result(k)) for k, r in enumerate(regs))
    assert n_str > 30000 and bs[0].stats()["n_regions_resolved"] > 1200
    lcd.RegionBatch.run_many(bs)
    for b in bs:
        b.download()
    assert bs[0].digest() == d0 and bs[1].digest() == d0
    for b in reversed(bs):
        b.close()


def test_sv_region_ont_60x(lcd, oracle):
    """SURVEY 8d config 5 shape: one 4 kb region, 60 noisy reads, a 1.2 kb insertion on one haplotype and a 700 bp deletion on the other:
    long ref<->cons gaps (K3 scores in the thousands), wide POA bands -- == oracle"""
    from longcalld_amd import jobs
    from conftest import mutate
    rng = np.random.default_rng(505)
    ref = rng.integers(0, 4, 4000).astype(np.uint8)
    ins = rng.integers(0, 4, 1200).astype(np.uint8)
    hap = [np.concatenate([ref[:1500], ins, ref[1500:]]), np.concatenate([ref[:2200], ref[2900:]])]
    seqs, haps, covers = [], [], []
    for i in range(60):
        h = i % 2
        s = mutate(rng, hap[h], 0.05)
        cv = 12
        if i % 11 == 5:
            s, cv = s[: len(s) * 3 // 5], 8
        elif i % 13 == 7:
            s, cv = s[len(s) // 2:], 4
        seqs.append(s); haps.append(h + 1); covers.append(cv)
    n = len(seqs)
    reg = dict(reg_len=len(ref), read_ids=np.arange(n, dtype=np.int32), seqs=seqs, quals=[np.full(len(s), 20, np.uint8) for s in seqs],
               covers=np.array(covers, np.int32), haps=np.array(haps, np.int32), phase_sets=np.full(n, 777, np.int64), ref=ref)
    o = lcd.default_opt(); o.is_ont = 1; o.collect_noisy_vars = 1
    got, ids, _, _ = _run_batch(lcd, [reg], o)
    exp = oracle.collect_noisy_reg_aln_strs(reg)
    assert (ids[0] == exp["sorted_ids"]).all()
    same_result(exp, got[0])
    assert got[0]["n_cons"] == 2
    lens = sorted(len(got[0]["aln_strs"][c][0]["query"][got[0]["aln_strs"][c][0]["query"] != 5]) for c in range(2))
    assert abs(lens[0] - 3300) < 60 and abs(lens[1] - 5200) < 80      # the two consensus sequences carry the deletion / the insertion


def test_full_size_batch_equals_oracle(lcd, oracle):
    """BASELINE configs[1] at full size against the oracle itself, region by region (the scalar oracle needs ~35 s for the 1 250 regions)"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(2000, jobs.regions_for_ref_mb(10), jobs.HIFI)
    got, ids, st, _ = _run_batch(lcd, regs)
    for r, g, sid in zip(regs, got, ids):
        exp = oracle.collect_noisy_reg_aln_strs(r)
        assert (sid == exp["sorted_ids"]).all()
        same_result(exp, g)
    assert st["n_regions"] == len(regs)


def test_ont_batch_at_scale_equals_oracle(lcd, oracle):
    """BASELINE configs[2] shape: 200 regions of 5 %-error reads in one batch, run twice -- the first run meets the overflow / retry rounds
    (capacities start from the clean-read estimates), the second runs with what the library learned; both == oracle region by region"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(3000, 200, jobs.ONT)
    o = lcd.default_opt(); o.is_ont = 1
    exp = [oracle.collect_noisy_reg_aln_strs(r) for r in regs]
    for attempt in range(2):
        got, ids, st, _ = _run_batch(lcd, regs, o)
        for e, g, sid in zip(exp, got, ids):
            assert (sid == e["sorted_ids"]).all()
            same_result(e, g)
        assert st["n_regions_resolved"] == sum(e["n_cons"] > 0 for e in exp)


def test_full_ont_batch_equals_oracle(lcd):
    """ONE FULL configs[2] batch -- the 1 250 regions of a 10 Mb step of 5 %-error reads, exactly what bench.py --shape ont submits per
    step -- region by region == oracle (sorted read order, clusters, every alignment string and coordinate).  This is where the retry / spill / overflow
    paths live (capacities start from the clean-read estimates; the first run meets the overflow rounds), so the batch is run twice.  The oracle runs on all
    host cores (conftest.oracle_many: ~5 CPU-minutes)."""
    from conftest import oracle_many
    from longcalld_amd import jobs
    n = jobs.regions_for_ref_mb(10.0)
    assert n == 1250
    regs = jobs.make_regions(3100, n, jobs.ONT)
    exp = oracle_many(3100, n, "ont")
    o = lcd.default_opt(); o.is_ont = 1
    for attempt in range(2):
        got, ids, st, _ = _run_batch(lcd, regs, o)
        for k, (e, g, sid) in enumerate(zip(exp, got, ids)):
            assert (sid == e["sorted_ids"]).all(), k
            same_result(e, g)
        assert st["n_regions"] == n and st["n_regions_resolved"] == sum(e["n_cons"] > 0 for e in exp)


def test_ont_certified_band_ring16_on_off_equals_oracle(lcd, oracle, monkeypatch):
    """noisy K2 chains through the certified band (LCD_CERT=2: off by default for noisy reads) with the 16-bit LDS ring on (forced), at its
    default and off -- 120 ONT-shape regions, all three == oracle"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(3200, 120, jobs.ONT)
    exp = [oracle.collect_noisy_reg_aln_strs(r) for r in regs]
    o = lcd.default_opt(); o.is_ont = 1
    monkeypatch.setenv("LCD_CERT", "2")
    dgs = []
    for r16 in ("2", None, "0"):
        if r16 is None: monkeypatch.delenv("LCD_RING16", raising=False)
        else: monkeypatch.setenv("LCD_RING16", r16)
        got, ids, st, dg = _run_batch(lcd, regs, o)
        for e, g, sid in zip(exp, got, ids):
            assert (sid == e["sorted_ids"]).all()
            same_result(e, g)
        dgs.append(dg)
    assert dgs[0] == dgs[1] == dgs[2]


def test_concurrent_callers(lcd):
    """SURVEY 8b threading: kt_for workers call into the library concurrently, each on its own chunk; four host threads with their own batches
    (ctypes releases the GIL during the calls) get the digests of the same batches run one after the other"""
    import threading
    from longcalld_amd import jobs
    sets = [jobs.make_regions(700 + t, 24, jobs.HIFI if t % 2 == 0 else jobs.ONT) for t in range(4)]
    serial = [_run_batch(lcd, s)[3] for s in sets]
    got, errs = [None] * 4, []

    def work(t):
        try:
            for _ in range(3):
                got[t] = _run_batch(lcd, sets[t])[3]
        except Exception as e:  # noqa
            errs.append(e)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs and got == serial


def test_device_per_batch_and_per_thread(lcd):
    """one process drives every GPU (SURVEY 8b threading; the reference's kt_for workers are threads): a batch is created ON a device, a worker
    thread selects its device for the per-call mirrors; same results, bad indices fail"""
    import threading
    from longcalld_amd import jobs, _lib
    lib = _lib.load_library()
    n = lib.lcd_device_count()
    assert n >= 1
    regs = jobs.make_regions(55, 8)
    ref = _run_batch(lcd, regs)[3]
    got = {}

    def work(dev):
        assert lib.lcd_set_thread_device(dev) == 0
        b = lcd.RegionBatch(device=dev)
        for r in regs:
            b.add_region(r)
        b.upload(); b.run(); b.download(); got[dev] = b.digest(); b.close()
        t = np.random.default_rng(1).integers(0, 4, 200).astype(np.uint8)
        got[("ed", dev)] = lcd.edlib_xgaps(t, np.delete(t, [7, 8, 90]))
    ths = [threading.Thread(target=work, args=(d,)) for d in range(n)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert all(got[d] == ref for d in range(n)) and all(got[("ed", d)] == 2 for d in range(n))
    assert lib.lcd_set_thread_device(n) != 0
    with pytest.raises(Exception):
        lcd.RegionBatch(device=n)


def test_dispatcher_one_process_all_gpus(lcd):
    """lcd_dispatch_run: job buffers created without a device (LCD_DEVICE_ANY), ordered by estimated work, pulled by one submitter thread per GPU;
    every batch's digest equals its stand-alone run, every batch ran on a device of the dispatcher"""
    from longcalld_amd import jobs
    sets = [jobs.make_regions(900 + i, n, jobs.HIFI if i % 3 else jobs.ONT) for i, n in enumerate([30, 6, 18, 2, 40, 11, 25])]
    alone = [_run_batch(lcd, s)[3] for s in sets]
    d = lcd.Dispatcher(coalesce=3)
    bs = []
    for s in sets:
        b = lcd.RegionBatch(device=lcd.DEVICE_ANY)
        for r in s:
            b.add_region(r)
        bs.append(b)
    costs = [b.cost() for b in bs]
    assert min(costs) > 0 and costs[3] == min(costs) and costs[4] > costs[5] > costs[3]   # (same shape: more regions, more work)
    dev = d.run(bs)
    assert ((dev >= 0) & (dev < d.n_devices)).all()
    assert [b.digest() for b in bs] == alone
    for b in bs:
        b.close()
    d.close()


def test_arena_slots_under_contention(lcd, monkeypatch):
    """arena slots (poa_kernel.hip / run_many_once): with LCD_ARENA_SLOT_CUS=2 a launch group of a full-size batch has 10-50x more chains than slots
    and far more resident workgroups than slots, so workgroups wait for a slot, run on whatever arena they get and release it -- the digest is that
    of private arenas"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(1234, 600, jobs.HIFI)
    ref = _run_batch(lcd, regs)[3]
    monkeypatch.setenv("LCD_ARENA_SLOT_CUS", "2")
    got = _run_batch(lcd, regs)[3]
    assert got == ref and ref != 0


def test_certified_band_equals_full_rows_at_scale(lcd, monkeypatch):
    """600 HiFi-shape regions (the phase-set-less ones run K2: ~90 chains of 20-50 reads) with the certified band (default) and with full rows (LCD_CERT=0):
    the same digest over every consensus, cluster and alignment string, fewer cells computed, the same cells of the reference's algorithm accounted"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(777, 600, jobs.HIFI)
    monkeypatch.setenv("LCD_CERT", "1")
    _, _, st1, d1 = _run_batch(lcd, regs)
    monkeypatch.setenv("LCD_CERT", "0")
    _, _, st0, d0 = _run_batch(lcd, regs)
    assert d1 == d0
    assert st1["poa_cells"] == st0["poa_cells"] == st0["poa_cells_computed"]
    assert st1["poa_cells_computed"] * 2 < st0["poa_cells_computed"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n", [("ont", 300), ("sv", 24)])
def test_certified_band_in_the_systolic_rows_equals_full_rows(lcd, monkeypatch, shape, n):
    """noisy reads: the K2 chains' certified band runs in the multi-wavefront (systolic) rows of the class the reads' length asks for (default; LCD_CERT_SYS=0:
    full rows) -- the same digest over every consensus, cluster and alignment string, the same cells of the reference's algorithm accounted, fewer computed"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(4242, n, jobs.ONT if shape == "ont" else jobs.SV)
    o = lcd.default_opt(); o.is_ont = 1
    monkeypatch.setenv("LCD_CERT_SYS", "1")
    _, _, st1, d1 = _run_batch(lcd, regs, o)
    monkeypatch.setenv("LCD_CERT_SYS", "0")
    _, _, st0, d0 = _run_batch(lcd, regs, o)
    assert d1 == d0
    assert st1["poa_cells"] == st0["poa_cells"]
    assert st1["poa_cells_computed"] < st0["poa_cells_computed"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,n", [("hifi", 400), ("ont", 300), ("sv", 16)])
def test_resort_layouts_give_the_same_order(lcd, oracle, monkeypatch, shape, n):
    """the re-sort's three homes of the graph -- packed words in LDS, the compact copy in LDS (bytes + CSR + a 256-entry FIFO; graphs of noisy reads that miss
    the pool as packed words), packed words in HBM -- are one Kahn FIFO order: same digest with the compact copy never used (LCD_DBG=64) and used wherever it
    fits (LCD_DBG=128), and the default equals the oracle region by region"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(777, n, {"hifi": jobs.HIFI, "ont": jobs.ONT, "sv": jobs.SV}[shape])
    o = lcd.default_opt(); o.is_ont = 0 if shape == "hifi" else 1
    monkeypatch.delenv("LCD_DBG", raising=False)
    res, _, _, d0 = _run_batch(lcd, regs, o)
    monkeypatch.setenv("LCD_DBG", "64")
    _, _, _, d1 = _run_batch(lcd, regs, o)
    monkeypatch.setenv("LCD_DBG", "128")
    _, _, _, d2 = _run_batch(lcd, regs, o)
    monkeypatch.setenv("LCD_DBG", "384")   # ... with a FIFO of three nodes: graphs with bubbles are handed over to the walk in HBM half way
    _, _, _, d3 = _run_batch(lcd, regs, o)
    assert d0 == d1 == d2 == d3
    monkeypatch.delenv("LCD_DBG")
    for r, got in list(zip(regs, res))[:: max(1, n // 20)]:
        same_result(oracle.collect_noisy_reg_aln_strs(r), got)


@pytest.mark.parametrize("shape", ["hifi", "ont", "sv"])
def test_dp_regions_grow_in_place(lcd, oracle, monkeypatch, shape):
    """DP regions sized far too small (test switch LCD_CELL_SHRINK): a chain whose read does not fit takes a larger region from the launch set's spare pool and
    repeats that read (poa_kernel.hip grow_dp_region) -- every class of rows, with and without the certified band; without a pool (LCD_SPARE_GB=0) the same
    chains come back and are re-run by the host.  Both == the unshrunk run's digest, and (hifi, ont) region by region == oracle"""
    from longcalld_amd import jobs
    if shape == "sv":
        regs = jobs.make_regions(41, 24, jobs.SV)     # (wide classes, systolic full rows, long K1 bands; compared with the unshrunk run, which tests/test_gpu_sv.py holds against the oracle)
    else:
        regs = jobs.make_regions(40, 60 if shape == "hifi" else 24, jobs.HIFI if shape == "hifi" else jobs.ONT)
    o = lcd.default_opt(); o.is_ont = 0 if shape == "hifi" else 1
    monkeypatch.setenv("LCD_CERT", "0" if shape == "hifi" else "1")
    _, _, st_ref, d_ref = _run_batch(lcd, regs, o)
    monkeypatch.setenv("LCD_CELL_SHRINK", "24")
    got, ids, st, d = _run_batch(lcd, regs, o)
    assert d == d_ref and st["poa_grown"] > 0 and st["poa_cells"] == st_ref["poa_cells"]
    for r, g, sid in zip(regs, got, ids):
        if shape == "sv":
            break
        exp = oracle.collect_noisy_reg_aln_strs(r)
        assert (sid == exp["sorted_ids"]).all()
        same_result(exp, g)
    monkeypatch.setenv("LCD_SPARE_GB", "0")
    _, _, st0, d0 = _run_batch(lcd, regs, o)
    assert d0 == d_ref and st0["poa_grown"] == 0 and st0["poa_retries"] > 0


def _rows_score2p(t_row, q_row, b=6, q=6, e=2, q2=24, e2=1):
    """penalty of an alignment given as two gapped rows (gap = 5) under the 2-piece affine model of src/align.c:392-395: mismatch b, a gap run of L columns
    min(q + e L, q2 + e2 L); a run is a maximal stretch of gaps in ONE row"""
    pen, i, n = 0, 0, len(t_row)
    while i < n:
        if t_row[i] == 5 or q_row[i] == 5:
            which = 0 if t_row[i] == 5 else 1
            j = i
            while j < n and ((t_row[j] == 5) if which == 0 else (q_row[j] == 5)):
                j += 1
            L = j - i
            pen += min(q + e * L, q2 + e2 * L)
            i = j
        else:
            pen += b if t_row[i] != q_row[i] else 0
            i += 1
    return pen


@pytest.mark.parametrize("shape,n", [("hifi", 1250), ("sv", 60)])
def test_ref_cons_alignments_are_optimal_by_an_independent_gotoh(lcd, oracle, shape, n):
    """K3 against a truth that is NOT the restatement: every ref<->cons alignment of a full-size batch, re-scored from the rows the HIP kernel returned, costs
    exactly what an independent O(nm) 2-piece Gotoh DP (oracle/wfa2p.c lcdo_gotoh2p_score: no wavefronts, no backtrace) says the optimum is -- and the rows
    de-gap to the reference slice and the consensus.  (Tie-breaks between equally optimal alignments are what stays with the restatement.)"""
    from longcalld_amd import jobs
    regs = jobs.make_regions(4242, n, {"hifi": jobs.HIFI, "sv": jobs.SV}[shape])
    o = lcd.default_opt(); o.is_ont = 0 if shape == "hifi" else 1
    got, _, st, _ = _run_batch(lcd, regs, o)
    n_jobs, worst = 0, 0
    for r, g in zip(regs, got):
        for c in range(g["n_cons"]):
            rc = g["aln_strs"][c][0]
            t_row, q_row = rc["target"], rc["query"]
            ref, cons = t_row[t_row != 5], q_row[q_row != 5]
            assert ref.tobytes() == r["ref"].tobytes() and not ((t_row == 5) & (q_row == 5)).any()
            if len(ref) * len(cons) > 40_000_000:        # (keep the O(nm) check to seconds: the longest SV regions are covered by tests/test_gpu_kernels.py)
                continue
            pen = _rows_score2p(t_row, q_row)
            assert pen == oracle.gotoh2p_score(ref, cons), (len(ref), len(cons), pen)
            n_jobs += 1; worst = max(worst, pen)
    assert n_jobs >= (2000 if shape == "hifi" else 60) and worst > 20
    assert st["n_regions_resolved"] >= 0.95 * len(regs)


def test_ont_batch_rows_degap_to_their_reads_at_full_size(lcd):
    """configs[2] at the full 1 250 regions (K1 wide bands, K2 full rows in every workgroup class, anchors): every cons<->read string de-gaps to its read and to the
    consensus, clusters partition the reads -- properties of the OUTPUT, independent of any restatement"""
    from longcalld_amd import jobs
    from conftest import check_invariants
    regs = jobs.make_regions(99, 1250, jobs.ONT)
    o = lcd.default_opt(); o.is_ont = 1
    got, _, st, _ = _run_batch(lcd, regs, o)
    n_str = sum(check_invariants(r, g) for r, g in zip(regs, got))
    assert n_str > 20000 and st["n_regions_resolved"] > 1000


def test_certified_band_and_long_chain_class_digest_at_job_scale(lcd, monkeypatch):
    """a configs[3]-sized sample (5 batches = 6 250 distinct regions in ONE submission): the digest over every consensus, cluster and alignment string is the same
    with the certified band on / off, with the long chains in the 256-thread class or not (LCD_SOLO_RL) with their rows on one or on four wavefronts (LCD_SOLO_MW), with the long K2 chains ahead of the anchor stage or not (LCD_EARLY) and with the host preparation on a helper thread or not -- the joint submission, the launch-group merge
    and the per-submission choice of long chains included"""
    from longcalld_amd import jobs
    batches = [jobs.make_regions(31000 + i, 1250, jobs.HIFI) for i in range(5)]

    def run():
        bs = []
        for regs in batches:
            b = lcd.RegionBatch()
            for r in regs:
                b.add_region(r)
            b.upload(); bs.append(b)
        lcd.RegionBatch.run_many(bs)
        out = []
        for b in bs:
            b.download(); out.append(b.digest())
        for b in bs:        # (only now: the ref<->cons rows of every batch live in the LEADER's buffers until they are downloaded)
            b.close()
        return out
    ref = run()
    monkeypatch.setenv("LCD_CERT", "0")
    assert run() == ref
    monkeypatch.setenv("LCD_CERT", "1"); monkeypatch.setenv("LCD_SOLO_RL", "0")
    assert run() == ref
    monkeypatch.setenv("LCD_SOLO_RL", "30000")
    assert run() == ref
    monkeypatch.setenv("LCD_SOLO_MW", "1")       # the long certified-band chains' rows on all four wavefronts of their workgroup (align_lean_mw)
    assert run() == ref
    monkeypatch.delenv("LCD_SOLO_MW"); monkeypatch.delenv("LCD_SOLO_RL")
    monkeypatch.setenv("LCD_EARLY", "0")         # the long K2 chains start with everybody else instead of ahead of the anchor stage
    assert run() == ref
    monkeypatch.delenv("LCD_EARLY"); monkeypatch.setenv("LCD_NO_PREP_THREAD", "1")   # classes, order and arena layout on the calling thread
    assert run() == ref
    monkeypatch.delenv("LCD_NO_PREP_THREAD")
    # the long chains' certified-band rows as a pipeline over four wavefronts (align_cyc): the default only from 16 000 chains per submission on -- here always / never
    monkeypatch.setenv("LCD_SOLO_CYC_MIN", "0")
    assert run() == ref
    monkeypatch.setenv("LCD_SOLO_RL", "30000")   # ... and with many more chains in that class
    assert run() == ref
    monkeypatch.delenv("LCD_SOLO_RL"); monkeypatch.delenv("LCD_SOLO_CYC_MIN"); monkeypatch.setenv("LCD_SOLO_CYC", "0")
    assert run() == ref
    monkeypatch.delenv("LCD_SOLO_CYC")
    monkeypatch.setenv("LCD_RING16", "0")        # the certified-band chains' LDS ring as 32-bit values (default: 16-bit, saturating)
    assert run() == ref
