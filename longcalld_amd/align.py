"""Host-side mirror of the reference's src/align.h surface over the C ABI (liblcd_hotpath.so).

Function names and argument meaning follow the reference (src/align.c); arrays are numpy uint8 byte codes
A0 C1 G2 T3 N4, gap 5.  Everything here calls the HIP library -- nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from ._lib import LcdAlnStr, LcdBatchStats, LcdDigar, LcdDigar1, LcdDigarOpt, LcdNoisyIv, LcdNoisyVar, LcdOpt, LcdReadView, check, load_library

_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]

u8p, i32p, u64p, u32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)

GAP_LEFT_ALN, GAP_RIGHT_ALN = 1, 2  # src/align.h:28-29
WFA_NO_HEURISTIC, WFA_AFFINE_2P = 0, 1  # src/align.h:35-40
NOISY_RIGHT_GAP, NOISY_LEFT_GAP, NOISY_RIGHT_COVER, NOISY_LEFT_COVER, NOISY_BOTH_COVER = 1, 2, 4, 8, 12  # src/align.h:6-11


def default_opt():
    lib = load_library()
    o = LcdOpt()
    lib.lcd_opt_default(C.byref(o))
    return o


def _p8(a):
    return a.ctypes.data_as(u8p)


def _pool(arrays):
    """pack byte arrays into one pool, 16-byte aligned; returns pool, offsets"""
    offs, tot = [], 0
    for a in arrays:
        offs.append(tot)
        tot += (len(a) + 15) // 16 * 16
    pool = np.full(tot + 16, 4, np.uint8)
    for a, o in zip(arrays, offs):
        pool[o:o + len(a)] = a
    return pool, np.array(offs, np.uint64)


def edlib_batch(pairs):
    """pairs: list of (target, query) uint8 arrays -> dict of int arrays dist/xgaps/n_eq/n_xid  (src/align.c:210-254)."""
    lib = load_library()
    n = len(pairs)
    arrs = [np.ascontiguousarray(x, np.uint8) for p in pairs for x in (p[1], p[0])]  # query, target
    pool, offs = _pool(arrs)
    qo, to = np.ascontiguousarray(offs[0::2]), np.ascontiguousarray(offs[1::2])
    ql = np.array([len(p[1]) for p in pairs], np.int32)
    tl = np.array([len(p[0]) for p in pairs], np.int32)
    out = {k: np.zeros(n, np.int32) for k in ("dist", "xgaps", "n_eq", "n_xid")}
    check(lib.lcd_edlib_batch(n, _p8(pool), pool.size, qo.ctypes.data_as(u64p), ql.ctypes.data_as(i32p), to.ctypes.data_as(u64p),
                              tl.ctypes.data_as(i32p), *[out[k].ctypes.data_as(i32p) for k in ("dist", "xgaps", "n_eq", "n_xid")]), lib)
    return out


def edlib_batch_hw(pairs):
    """HW (infix) mode, edlib_infix_aln (src/align.c:256): pairs of (target, query) -> dict of int arrays dist/xgaps/n_eq/n_xid/start/end"""
    lib = load_library()
    n = len(pairs)
    arrs = [np.ascontiguousarray(x, np.uint8) for p in pairs for x in (p[1], p[0])]  # query, target
    pool, offs = _pool(arrs)
    qo, to = np.ascontiguousarray(offs[0::2]), np.ascontiguousarray(offs[1::2])
    ql = np.array([len(p[1]) for p in pairs], np.int32)
    tl = np.array([len(p[0]) for p in pairs], np.int32)
    keys = ("dist", "xgaps", "n_eq", "n_xid", "start", "end")
    out = {k: np.zeros(n, np.int32) for k in keys}
    check(lib.lcd_edlib_batch_hw(n, _p8(pool), pool.size, qo.ctypes.data_as(u64p), ql.ctypes.data_as(i32p), to.ctypes.data_as(u64p),
                                 tl.ctypes.data_as(i32p), *[out[k].ctypes.data_as(i32p) for k in keys]), lib)
    return out


def edlib_infix_aln(target, query):
    """edlib_infix_aln, src/align.c:256 -> (distance, n_eq, n_xid) through the per-call export"""
    lib = load_library()
    t, q = np.ascontiguousarray(target, np.uint8), np.ascontiguousarray(query, np.uint8)
    a, b = C.c_int(), C.c_int()
    d = lib.lcd_edlib_infix_aln(_p8(t), len(t), _p8(q), len(q), C.byref(a), C.byref(b))
    return d, a.value, b.value


def edlib_xgaps(target, query):
    """edlib_xgaps, src/align.c:222"""
    return int(edlib_batch([(target, query)])["xgaps"][0])


def edlib_end2end_aln(target, query):
    """edlib_end2end_aln, src/align.c:234 -> (distance, n_eq, n_xid)"""
    r = edlib_batch([(target, query)])
    return int(r["dist"][0]), int(r["n_eq"][0]), int(r["n_xid"][0])


def wfa_batch(pairs, gap_aln=GAP_LEFT_ALN, b=6, q=6, e=2, q2=24, e2=1):
    """pairs: list of (pattern, text) -> list of dict(score, cigar[uint32], pattern_alg, text_alg)  (src/align.c:374-460)."""
    lib = load_library()
    n = len(pairs)
    arrs = [np.ascontiguousarray(x, np.uint8) for p in pairs for x in (p[0], p[1])]
    pool, offs = _pool(arrs)
    po, to = np.ascontiguousarray(offs[0::2]), np.ascontiguousarray(offs[1::2])
    pl = np.array([len(p[0]) for p in pairs], np.int32)
    tl = np.array([len(p[1]) for p in pairs], np.int32)
    ga = np.full(n, gap_aln, np.int32) if np.isscalar(gap_aln) else np.asarray(gap_aln, np.int32)
    stride = int((pl + tl).max()) + 1 if n else 1
    score, ncig, alen = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    cig = np.zeros((n, stride), np.uint32)
    rows = np.zeros((n, 2, stride), np.uint8)
    check(lib.lcd_wfa_batch(n, _p8(pool), pool.size, po.ctypes.data_as(u64p), pl.ctypes.data_as(i32p), to.ctypes.data_as(u64p),
                            tl.ctypes.data_as(i32p), ga.ctypes.data_as(i32p), b, q, e, q2, e2, 3, score.ctypes.data_as(i32p),
                            cig.ctypes.data_as(u32p), stride, ncig.ctypes.data_as(i32p), _p8(rows), stride, alen.ctypes.data_as(i32p)), lib)
    return [dict(score=int(score[i]), cigar=cig[i, :ncig[i]].copy(), pattern_alg=rows[i, 0, :alen[i]].copy(), text_alg=rows[i, 1, :alen[i]].copy())
            for i in range(n)]


def wfa_arena_bytes(plen, tlen, score_bound, b=6, q=6, e=2, q2=24, e2=1):
    """device work-arena bytes of one K3 alignment with the given score bound (lcd_wfa_arena_bytes)"""
    return int(load_library().lcd_wfa_arena_bytes(int(plen), int(tlen), int(score_bound), b, q, e, q2, e2))


def wfa_end2end_aln(pattern, text, gap_aln=GAP_LEFT_ALN, b=6, q=6, e=2, q2=24, e2=1):
    """wfa_end2end_aln through the per-call C mirror (exercises the malloc/ownership contract, src/align.c:374)."""
    lib = load_library()
    pattern = np.ascontiguousarray(pattern, np.uint8)
    text = np.ascontiguousarray(text, np.uint8)
    cb, cl, pa, ta, al = u32p(), C.c_int(), u8p(), u8p(), C.c_int()
    check(lib.lcd_wfa_end2end_aln(_p8(pattern), len(pattern), _p8(text), len(text), gap_aln, b, q, e, q2, e2, WFA_NO_HEURISTIC, WFA_AFFINE_2P,
                                  C.byref(cb), C.byref(cl), C.byref(pa), C.byref(ta), C.byref(al)), lib)
    cigar = np.ctypeslib.as_array(cb, shape=(max(cl.value, 1),))[:cl.value].copy()
    p_alg = np.ctypeslib.as_array(pa, shape=(max(al.value, 1),))[:al.value].copy()
    t_alg = np.ctypeslib.as_array(ta, shape=(max(al.value, 1),))[:al.value].copy()
    _libc.free(cb)
    _libc.free(pa)  # one block: only the first pointer is freed (src/align.c:490)
    return cigar, p_alg, t_alg


def poa_batch(chains, opt=None):
    """chains: list of dict(mode, reads=[uint8 arrays], skip=[...], anchors=[(ref_beg, ref_end, read_beg, read_end)]) -> list of dict."""
    lib = load_library()
    opt = opt or default_opt()
    reads = [np.ascontiguousarray(r, np.uint8) for ch in chains for r in ch["reads"]]
    pool, offs = _pool(reads)
    lens = np.array([len(r) for r in reads], np.int32)
    nC = len(chains)
    mode = np.array([ch["mode"] for ch in chains], np.int32)
    nr = np.array([len(ch["reads"]) for ch in chains], np.int32)
    r0 = np.concatenate([[0], np.cumsum(nr)[:-1]]).astype(np.int32)
    skip = np.array([s for ch in chains for s in ch.get("skip", [0] * len(ch["reads"]))], np.int32)
    anch = []
    for ch in chains:
        a = ch.get("anchors")
        for k, r in enumerate(ch["reads"]):
            anch.extend(a[k] if a is not None else (1, len(ch["reads"][0]), 1, len(r)))
    anch = np.array(anch, np.int32)
    max_reads = int(nr.max())
    stride = int(max(sum(len(r) for r in ch["reads"]) for ch in chains)) + 2
    status, n_cons, msa_len = np.zeros(nC, np.int32), np.zeros(nC, np.int32), np.zeros(nC, np.int32)
    cons_len, clu_n = np.zeros((nC, 2), np.int32), np.zeros((nC, 2), np.int32)
    cons = np.zeros((nC, 2, stride), np.uint8)
    msa = np.zeros((nC, max_reads + 2, stride), np.uint8)
    clu_ids = np.zeros((nC, 2, max_reads), np.int32)
    check(lib.lcd_poa_batch(C.byref(opt), nC, mode.ctypes.data_as(i32p), r0.ctypes.data_as(i32p), nr.ctypes.data_as(i32p), len(reads),
                            offs.ctypes.data_as(u64p), lens.ctypes.data_as(i32p), skip.ctypes.data_as(i32p), anch.ctypes.data_as(i32p),
                            _p8(pool), pool.size, status.ctypes.data_as(i32p), n_cons.ctypes.data_as(i32p), cons_len.ctypes.data_as(i32p),
                            msa_len.ctypes.data_as(i32p), clu_n.ctypes.data_as(i32p), _p8(cons), stride, _p8(msa), stride, max_reads,
                            clu_ids.ctypes.data_as(i32p)), lib)
    out = []
    for c in range(nC):
        nc = int(n_cons[c])
        out.append(dict(status=int(status[c]), n_cons=nc, msa_len=int(msa_len[c]),
                        cons=[cons[c, k, :cons_len[c, k]].copy() for k in range(nc)],
                        msa=[msa[c, r, :msa_len[c]].copy() for r in range(int(nr[c]) + nc)],
                        clu=[clu_ids[c, k, :clu_n[c, k]].copy() for k in range(nc)]))
    return out


DEVICE_ANY = -2  # LCD_DEVICE_ANY: a host-only job buffer, bound to a GPU at upload / dispatch time


def lpt_assign(costs, n_bins):
    """lcd_lpt_assign: longest-processing-time assignment -> (bin of every item, load per bin); pure host code (no GPU needed)"""
    lib = load_library()
    c = np.ascontiguousarray(costs, np.float64)
    out = np.zeros(max(len(c), 1), np.int32); load = np.zeros(max(n_bins, 1), np.float64)
    lib.lcd_lpt_assign(len(c), c.ctypes.data_as(C.POINTER(C.c_double)), int(n_bins), out.ctypes.data_as(i32p), load.ctypes.data_as(C.POINTER(C.c_double)))
    return out[:len(c)].copy(), load[:n_bins].copy()


class Dispatcher:
    """lcd_dispatch_*: one process driving every GPU of the node -- per-device submitter threads pulling job buffers from one cost-ordered queue
    (the GPU-side analogue of kt_for, src/kthread.c:24-64)"""

    def __init__(self, devices=None, coalesce=16):
        self.lib = load_library()
        if devices is None:
            self.h = self.lib.lcd_dispatch_create(0, None, int(coalesce))
        else:
            d = np.ascontiguousarray(devices, np.int32)
            self.h = self.lib.lcd_dispatch_create(len(d), d.ctypes.data_as(i32p), int(coalesce))
        if not self.h:
            raise RuntimeError("lcd_dispatch_create failed: " + self.lib.lcd_last_error().decode())

    @property
    def n_devices(self):
        return int(self.lib.lcd_dispatch_n_devices(self.h))

    def run(self, batches):
        """uploads, runs and downloads every batch; returns the device each batch ran on"""
        arr = (C.c_void_p * len(batches))(*[b.h for b in batches])
        dev = np.full(max(len(batches), 1), -1, np.int32)
        check(self.lib.lcd_dispatch_run(self.h, arr, len(batches), dev.ctypes.data_as(i32p)), self.lib)
        return dev[:len(batches)].copy()

    def set_flags(self, flags):
        """bit 0: leave the results on the device (no download after a submission)"""
        self.lib.lcd_dispatch_set_flags.argtypes = [C.c_void_p, C.c_int]
        self.lib.lcd_dispatch_set_flags(self.h, int(flags))

    def busy(self):
        """-> (ms inside submissions, #submissions) per device of the dispatcher, for the last run()"""
        n = self.n_devices
        ms = np.zeros(max(n, 1), np.float64); ns = np.zeros(max(n, 1), np.int32)
        self.lib.lcd_dispatch_busy.argtypes = [C.c_void_p, C.POINTER(C.c_double), i32p]
        self.lib.lcd_dispatch_busy(self.h, ms.ctypes.data_as(C.POINTER(C.c_double)), ns.ctypes.data_as(i32p))
        return ms[:n].copy(), ns[:n].copy()

    def close(self):
        if self.h:
            self.lib.lcd_dispatch_destroy(self.h)
            self.h = None


class RegionBatch:
    """Batched collect_noisy_reg_aln_strs (src/align.c:1760) over many independent regions of one pass (SURVEY CS-2)."""

    def __init__(self, opt=None, device=-1):
        """device: the GPU this batch lives on (lcd_batch_create_on); -1 = the calling thread's / process default device"""
        self.lib = load_library()
        self.opt = opt or default_opt()
        self.h = self.lib.lcd_batch_create_on(C.byref(self.opt), int(device))
        if not self.h:
            raise RuntimeError("lcd_batch_create failed: " + self.lib.lcd_last_error().decode())
        self.n_reads = []
        self._keep = []

    def close(self):
        if self.h:
            self.lib.lcd_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        self.lib.lcd_batch_clear(self.h)
        self.n_reads = []

    def add_region(self, reg):
        """reg: dict(reg_len, read_ids, seqs, quals(optional), covers, haps, phase_sets, ref) -- the outputs of collect_noisy_read_info"""
        n = len(reg["seqs"])
        seqs = [np.ascontiguousarray(s, np.uint8) for s in reg["seqs"]]
        quals = reg.get("quals")
        quals = [np.ascontiguousarray(s, np.uint8) for s in quals] if quals is not None else [np.zeros(max(len(s), 1), np.uint8) for s in seqs]
        sp = (u8p * n)(*[_p8(s) for s in seqs])
        qp = (u8p * n)(*[_p8(s) for s in quals])
        ids = np.asarray(reg["read_ids"], np.int32)
        lens = np.array([len(s) for s in seqs], np.int32)
        cov = np.asarray(reg["covers"], np.int32)
        haps = np.asarray(reg["haps"], np.int32)
        pss = np.asarray(reg["phase_sets"], np.int64)
        ref = np.ascontiguousarray(reg["ref"], np.uint8)
        idx = check(self.lib.lcd_batch_add_region(self.h, int(reg["reg_len"]), n, ids.ctypes.data_as(i32p), lens.ctypes.data_as(i32p), sp, qp,
                                                  cov.ctypes.data_as(i32p), haps.ctypes.data_as(i32p), pss.ctypes.data_as(C.POINTER(C.c_int64)),
                                                  _p8(ref), len(ref)), self.lib)
        self.n_reads.append(n)
        return idx

    def add_region_from_chunk(self, views, reg_beg, reg_end, noisy_reads, ref_slice, packed=False):
        """region [reg_beg, reg_end] (1-based, flanks included) of a chunk whose reads are `views` (make_read_views): the digar walk of
        collect_noisy_read_info (src/align.c:1377-1461) runs inside the library; noisy_reads = chunk read ids overlapping the region.
        packed: the reads' bases stay 4-bit packed on the host and are unpacked on the device by upload() (lcd_batch_add_region_from_chunk_packed)"""
        ids = np.ascontiguousarray(noisy_reads, np.int32)
        ref = np.ascontiguousarray(ref_slice, np.uint8)
        fn = self.lib.lcd_batch_add_region_from_chunk_packed if packed else self.lib.lcd_batch_add_region_from_chunk
        idx = check(fn(self.h, views[0], int(reg_beg), int(reg_end), len(ids), ids.ctypes.data_as(i32p), _p8(ref), len(ref)), self.lib)
        self.n_reads.append(len(ids))
        return idx

    def read_slices(self, region):
        """-> dict(read_ids, covers, read_beg, read_end) of a region's reads in sorted order (lcd_batch_region_read_slices)"""
        n = max(self.n_reads[region], 1)
        ids, cov, rb, re = (np.zeros(n, np.int32) for _ in range(4))
        k = check(self.lib.lcd_batch_region_read_slices(self.h, region, ids.ctypes.data_as(i32p), cov.ctypes.data_as(i32p), rb.ctypes.data_as(i32p), re.ctypes.data_as(i32p)), self.lib)
        return dict(read_ids=ids[:k].copy(), covers=cov[:k].copy(), read_beg=rb[:k].copy(), read_end=re[:k].copy())

    def cost(self):
        """the work estimate queues and shards are ordered by (DP cells of the batch's chains; host only)"""
        return float(self.lib.lcd_batch_cost(self.h))

    def upload(self):
        check(self.lib.lcd_batch_upload(self.h), self.lib)

    def run(self):
        check(self.lib.lcd_batch_run(self.h), self.lib)

    @staticmethod
    def run_many(batches):
        """lcd_batch_run_many: the hot path of several uploaded batches as ONE set of launches per stage (batches[0] leads)"""
        if not batches:
            return
        arr = (C.c_void_p * len(batches))(*[b.h for b in batches])
        check(batches[0].lib.lcd_batch_run_many(arr, len(batches)), batches[0].lib)

    def download(self):
        check(self.lib.lcd_batch_download(self.h), self.lib)

    def k4_pairs(self):
        """the (target, query) pairs of the batch's K4 jobs (lcd_batch_k4_jobs), copied out of the host pool"""
        n = self.lib.lcd_batch_k4_jobs(self.h, 0, None, None, None, None, None, None)
        if n <= 0:
            return []
        to, qo = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        tl, ql = np.zeros(n, np.int32), np.zeros(n, np.int32)
        pool, plen = u8p(), C.c_uint64(0)
        self.lib.lcd_batch_k4_jobs(self.h, n, to.ctypes.data_as(u64p), tl.ctypes.data_as(i32p), qo.ctypes.data_as(u64p), ql.ctypes.data_as(i32p),
                                   C.byref(pool), C.byref(plen))
        h = np.ctypeslib.as_array(pool, shape=(plen.value,))
        return [(h[int(to[i]):int(to[i]) + int(tl[i])].copy(), h[int(qo[i]):int(qo[i]) + int(ql[i])].copy()) for i in range(n)]

    def stats(self):
        st = LcdBatchStats()
        self.lib.lcd_batch_get_stats(self.h, C.byref(st))
        return {f[0]: getattr(st, f[0]) for f in LcdBatchStats._fields_}

    def digest(self):
        return int(self.lib.lcd_batch_digest(self.h))

    def materialize(self):
        """every region's results as the per-call mirror returns them (malloc()'d), freed again -> bytes of alignment rows"""
        return int(self.lib.lcd_batch_materialize(self.h))

    def sorted_ids(self, region):
        out = np.zeros(max(self.n_reads[region], 1), np.int32)
        n = self.lib.lcd_batch_region_sorted_ids(self.h, region, out.ctypes.data_as(i32p))
        return out[:n].copy()

    def region_vars(self, region, noisy_reg_beg, chunk_ref, chunk_ref_beg):
        """make_vars_from_msa_cons_aln (src/collect_var.c:2279) of one region, computed in stage S6 (opt.collect_noisy_vars):
        -> dict(n_vars, per-variant arrays, alle_covs (n, 2), alt_seqs, row_read_ids, prof_start, prof_end, prof_alleles (rows x n))"""
        cref = np.ascontiguousarray(chunk_ref, np.uint8)
        vp = C.POINTER(LcdNoisyVar)(); nrows = C.c_int(0)
        ids, ps, pe, pa = i32p(), i32p(), i32p(), i32p()
        n = check(self.lib.lcd_batch_region_vars(self.h, region, int(noisy_reg_beg), _p8(cref), int(chunk_ref_beg), len(cref), C.byref(vp), C.byref(nrows),
                                                 C.byref(ids), C.byref(ps), C.byref(pe), C.byref(pa)), self.lib)
        rows = nrows.value
        out = dict(n_vars=n, n_rows=rows)
        for k in ("pos", "var_type", "ref_len", "alt_len", "cate", "from_cons", "is_homopolymer_indel", "ref_base", "alt_ref_base", "total_cov"):
            out[k] = np.array([getattr(vp[i], k) for i in range(n)], np.int64)
        out["alle_covs"] = np.array([[vp[i].alle_covs[0], vp[i].alle_covs[1]] for i in range(n)], np.int32).reshape(n, 2)
        out["alt_seqs"] = [np.array([vp[i].alt_seq[k] for k in range(vp[i].alt_len)], np.uint8) for i in range(n)]
        as_arr = lambda p, m: np.ctypeslib.as_array(p, shape=(max(m, 1),))[:m].copy() if p else np.zeros(0, np.int32)
        out["row_read_ids"] = as_arr(ids, rows); out["prof_start"] = as_arr(ps, rows); out["prof_end"] = as_arr(pe, rows)
        out["prof_alleles"] = as_arr(pa, rows * n).reshape(rows, n)
        for i in range(n):
            if vp[i].alt_seq:
                _libc.free(C.cast(vp[i].alt_seq, C.c_void_p))
        for p in (vp, ids, ps, pe, pa):
            if p:
                _libc.free(C.cast(p, C.c_void_p))
        return out

    def result(self, region):
        """-> dict(n_cons, clu_n_seqs, clu_read_ids, aln_strs[c][j] = None | dict(target, query, beg/end...)), freeing the C buffers"""
        n = self.n_reads[region]
        m = 1 + 2 * n
        clu_n = (C.c_int * 2)(0, 0)
        clu_ids = (i32p * 2)()
        a0, a1 = (LcdAlnStr * m)(), (LcdAlnStr * m)()
        arr = (C.POINTER(LcdAlnStr) * 2)(C.cast(a0, C.POINTER(LcdAlnStr)), C.cast(a1, C.POINTER(LcdAlnStr)))
        nc = check(self.lib.lcd_batch_region_result(self.h, region, clu_n, clu_ids, arr), self.lib)
        res = dict(n_cons=nc, clu_n_seqs=[int(clu_n[0]), int(clu_n[1])], clu_read_ids=[], aln_strs=[[], []])
        for c in range(2):
            if clu_ids[c]:
                res["clu_read_ids"].append(np.ctypeslib.as_array(clu_ids[c], shape=(max(clu_n[c], 1),))[:clu_n[c]].copy())
                _libc.free(clu_ids[c])
            else:
                res["clu_read_ids"].append(None)
            for j in range(m):
                s = (a0, a1)[c][j]
                if not s.target_aln:
                    res["aln_strs"][c].append(None)
                    continue
                L = s.aln_len
                t = np.ctypeslib.as_array(s.target_aln, shape=(max(L, 1),))[:L].copy()
                q = np.ctypeslib.as_array(s.query_aln, shape=(max(L, 1),))[:L].copy()
                res["aln_strs"][c].append(dict(target=t, query=q, aln_len=L, target_beg=s.target_beg, target_end=s.target_end,
                                               query_beg=s.query_beg, query_end=s.query_end))
                _libc.free(s.target_aln)
        return res


    def results_arena(self, parse=True):
        """lcd_batch_region_results_arena: every region's results in ONE host block (interior pointers, one free).  parse=True -> list of dicts like result();
        parse=False -> (n_regions, arena bytes) -- what bench.py times"""
        from ._lib import LcdRegionResult
        tab = C.POINTER(LcdRegionResult)(); arena = C.c_void_p(); nbytes = C.c_uint64()
        fn = self.lib.lcd_batch_region_results_arena
        fn.argtypes = [C.c_void_p, C.POINTER(C.POINTER(LcdRegionResult)), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        n = check(fn(self.h, C.byref(tab), C.byref(arena), C.byref(nbytes)), self.lib)
        if not parse:
            _libc.free(arena)
            return n, int(nbytes.value)
        out = []
        for r in range(n):
            rr = tab[r]
            res = dict(n_cons=rr.n_cons, clu_n_seqs=[int(rr.clu_n_seqs[0]), int(rr.clu_n_seqs[1])], clu_read_ids=[], aln_strs=[[], []])
            for c in range(2):
                res["clu_read_ids"].append(np.ctypeslib.as_array(rr.clu_read_ids[c], shape=(max(rr.clu_n_seqs[c], 1),))[:rr.clu_n_seqs[c]].copy() if rr.clu_read_ids[c] else None)
                for j in range(rr.n_aln_strs):
                    s = rr.aln_strs[c][j]
                    if not s.target_aln:
                        res["aln_strs"][c].append(None)
                        continue
                    L = s.aln_len
                    res["aln_strs"][c].append(dict(target=np.ctypeslib.as_array(s.target_aln, shape=(max(L, 1),))[:L].copy(), query=np.ctypeslib.as_array(s.query_aln, shape=(max(L, 1),))[:L].copy(),
                                                   aln_len=L, target_beg=s.target_beg, target_end=s.target_end, query_beg=s.query_beg, query_end=s.query_end))
            out.append(res)
        _libc.free(arena)
        return out


def make_read_views(digars, bseqs, quals, qlens, haps, phase_sets):
    """lcd_read_view_t[] over numpy storage: digars[i] = (n, 4) int64 rows (pos, type, len, qi) as digar1_t (src/bam_utils.h:27-33),
    bseqs[i] = 4-bit BAM-packed bases, quals[i] = phred bytes.  Returns (ctypes array, keep-alive list)."""
    n = len(digars)
    arr = (LcdReadView * n)()
    keep = []
    for i in range(n):
        d = np.asarray(digars[i], np.int64)
        da = (LcdDigar1 * max(len(d), 1))()
        for k in range(len(d)):
            da[k].pos, da[k].type, da[k].len, da[k].qi = int(d[k, 0]), int(d[k, 1]), int(d[k, 2]), int(d[k, 3])
        bs = np.ascontiguousarray(bseqs[i], np.uint8); ql = np.ascontiguousarray(quals[i], np.uint8)
        arr[i].digars = da; arr[i].n_digar = len(d); arr[i].qlen = int(qlens[i])
        arr[i].bseq = _p8(bs); arr[i].qual = _p8(ql); arr[i].hap = int(haps[i]); arr[i].phase_set = int(phase_sets[i])
        keep += [da, bs, ql]
    return arr, keep


def copy_counters():
    """lcd_copy_counters: bytes of [digars D2H, digars H2D, read bases H2D, read bases D2H] since process start"""
    lib = load_library()
    out = (C.c_ulonglong * 4)()
    lib.lcd_copy_counters.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.lcd_copy_counters(out)
    return [int(x) for x in out]


class DeviceChunk:
    """lcd_chunk_t: a chunk's reads uploaded once; digars made and kept in HBM; region slices cut there; a batch's read bases unpacked from there"""

    def __init__(self, pos0, cigars, quals, bseqs, reg_beg, reg_end, whole_ref_len, is_ont=0, pal_flags=None):
        self.lib = lib = load_library()
        opt = LcdDigarOpt(); lib.lcd_digar_opt_default(C.byref(opt), int(is_ont))
        n = self.n = len(cigars)
        cg = [np.ascontiguousarray(c, np.uint32) for c in cigars]; ql = [np.ascontiguousarray(q, np.uint8) for q in quals]; sq = [np.ascontiguousarray(x, np.uint8) for x in bseqs]
        off = lambda xs: np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.uint64)
        coff, qoff, soff = off(cg), off(ql), off(sq)
        cpool = np.concatenate(cg + [np.zeros(1, np.uint32)]); qpool = np.concatenate(ql + [np.zeros(1, np.uint8)]); spool = np.concatenate(sq + [np.zeros(2, np.uint8)])
        ncig = np.array([len(c) for c in cg], np.int32); qlen = np.array([len(q) for q in ql], np.int32)
        p0 = np.ascontiguousarray(pos0, np.int64)
        pf = np.ascontiguousarray(pal_flags if pal_flags is not None else np.zeros(n, np.uint8), np.uint8)
        u64p_, i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
        lib.lcd_chunk_create.restype = C.c_void_p
        lib.lcd_chunk_create.argtypes = [C.POINTER(LcdDigarOpt), C.c_int, i64p, C.POINTER(C.c_uint32), u64p_, i32p, u8p, u64p_, i32p, u8p, u8p, u64p_, C.c_int64, C.c_int64, C.c_int64]
        self.h = lib.lcd_chunk_create(C.byref(opt), n, p0.ctypes.data_as(i64p), cpool.ctypes.data_as(C.POINTER(C.c_uint32)), coff.ctypes.data_as(u64p_), ncig.ctypes.data_as(i32p),
                                      _p8(qpool), qoff.ctypes.data_as(u64p_), qlen.ctypes.data_as(i32p), _p8(pf), _p8(spool), soff.ctypes.data_as(u64p_),
                                      int(reg_beg), int(reg_end), int(whole_ref_len))
        if not self.h:
            raise RuntimeError("lcd_chunk_create failed: " + lib.lcd_last_error().decode())
        self.packed_bytes = int(soff[-1])

    @classmethod
    def from_bam(cls, bam_path, bai_path, chrom, reg_beg, reg_end, min_mapq=30, is_ont=0, verify_crc=1):
        """lcd_chunk_create_from_bam: the region's BGZF blocks inflated on the device, records found / filtered / turned into digars there.
        -> the chunk; .meta = per-read scalars and names (dict of numpy arrays / list)"""
        from ._lib import LcdBamReads
        self = cls.__new__(cls)
        self.lib = lib = load_library()
        opt = LcdDigarOpt(); lib.lcd_digar_opt_default(C.byref(opt), int(is_ont))
        m = LcdBamReads()
        lib.lcd_chunk_create_from_bam.restype = C.c_void_p
        lib.lcd_chunk_create_from_bam.argtypes = [C.POINTER(LcdDigarOpt), C.c_char_p, C.c_char_p, C.c_char_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(LcdBamReads)]
        enc = lambda x: x if isinstance(x, bytes) else str(x).encode()
        self.h = lib.lcd_chunk_create_from_bam(C.byref(opt), enc(bam_path), enc(bai_path), enc(chrom), int(reg_beg), int(reg_end), int(min_mapq), int(verify_crc), C.byref(m))
        if not self.h:
            raise RuntimeError("lcd_chunk_create_from_bam failed: " + lib.lcd_last_error().decode())
        n = self.n = m.n_reads
        arr = lambda p, dt: np.array([p[i] for i in range(n)], dt)
        self.meta = dict(tid=m.tid, n_targets=m.n_targets, target_len=m.target_len, pos0=arr(m.pos0, np.int64), end_pos=arr(m.end_pos, np.int64), mapq=arr(m.mapq, np.int32),
                         flag=arr(m.flag, np.int32), n_cigar=arr(m.n_cigar, np.int32), qlen=arr(m.qlen, np.int32),
                         names=[C.string_at(C.addressof(m.name_pool.contents) + m.name_off[i]).decode() for i in range(n)])
        lib.lcd_bam_reads_free.argtypes = [C.POINTER(LcdBamReads)]
        lib.lcd_bam_reads_free(C.byref(m))
        self.packed_bytes = 0
        return self

    def read_info(self):
        n = self.n
        st = np.zeros(n, np.int32); beg = np.zeros(n, np.int64); end = np.zeros(n, np.int64); nc = np.zeros(n, np.int32); nd = np.zeros(n, np.int32)
        i64p = C.POINTER(C.c_int64)
        self.lib.lcd_chunk_read_info.argtypes = [C.c_void_p, i32p, i64p, i64p, i32p, i32p]
        self.lib.lcd_chunk_read_info(self.h, st.ctypes.data_as(i32p), beg.ctypes.data_as(i64p), end.ctypes.data_as(i64p), nc.ctypes.data_as(i32p), nd.ctypes.data_as(i32p))
        return dict(status=st, beg=beg, end=end, n_cand=nc, n_digars=nd)

    def intervals(self):
        """-> per read (noisy (m, 3) int64 [start, end, label], in_chunk mask)"""
        u64p_ = C.POINTER(C.c_uint64)
        ioff, iv, inc = u64p_(), C.POINTER(LcdNoisyIv)(), u8p()
        self.lib.lcd_chunk_intervals.argtypes = [C.c_void_p, C.POINTER(u64p_), C.POINTER(C.POINTER(LcdNoisyIv)), C.POINTER(u8p)]
        self.lib.lcd_chunk_intervals(self.h, C.byref(ioff), C.byref(iv), C.byref(inc))
        out = []
        for r in range(self.n):
            a, b = int(ioff[r]), int(ioff[r + 1])
            out.append((np.array([[iv[k].start, iv[k].end, iv[k].label] for k in range(a, b)], np.int64).reshape(-1, 3), np.array([inc[k] for k in range(a, b)], bool)))
        return out

    def region_slices(self, pair_read, pair_beg, pair_end, flank=10):
        pr = np.ascontiguousarray(pair_read, np.int32); pb = np.ascontiguousarray(pair_beg, np.int64); pe = np.ascontiguousarray(pair_end, np.int64)
        n = len(pr)
        rb = np.zeros(n, np.int32); re_ = np.zeros(n, np.int32); cv = np.zeros(n, np.int32)
        i64p = C.POINTER(C.c_int64)
        self.lib.lcd_chunk_region_slices.argtypes = [C.c_void_p, C.c_int, i32p, i64p, i64p, C.c_int, i32p, i32p, i32p]
        check(self.lib.lcd_chunk_region_slices(self.h, n, pr.ctypes.data_as(i32p), pb.ctypes.data_as(i64p), pe.ctypes.data_as(i64p), int(flank),
                                               rb.ctypes.data_as(i32p), re_.ctypes.data_as(i32p), cv.ctypes.data_as(i32p)), self.lib)
        return rb, re_, cv

    def add_region(self, batch, reg_beg, reg_end, read_ids, read_beg, read_end, cover, haps, phase_sets, ref):
        ids = np.ascontiguousarray(read_ids, np.int32); rb = np.ascontiguousarray(read_beg, np.int32); re_ = np.ascontiguousarray(read_end, np.int32)
        cv = np.ascontiguousarray(cover, np.int32); hp = np.ascontiguousarray(haps, np.int32); ps = np.ascontiguousarray(phase_sets, np.int64)
        rf = np.ascontiguousarray(ref, np.uint8)
        i64p = C.POINTER(C.c_int64)
        self.lib.lcd_batch_add_region_from_chunk_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, i32p, i32p, i32p, i32p, i32p, i64p, u8p, C.c_int]
        ri = check(self.lib.lcd_batch_add_region_from_chunk_dev(batch.h, self.h, int(reg_beg), int(reg_end), len(ids), ids.ctypes.data_as(i32p), rb.ctypes.data_as(i32p),
                                                                re_.ctypes.data_as(i32p), cv.ctypes.data_as(i32p), hp.ctypes.data_as(i32p), ps.ctypes.data_as(i64p), _p8(rf), len(rf)), self.lib)
        batch.n_reads.append(len(ids))
        return ri

    def close(self):
        if self.h:
            self.lib.lcd_chunk_destroy.argtypes = [C.c_void_p]
            self.lib.lcd_chunk_destroy(self.h)
            self.h = None


def digar_batch(pos0, cigars, quals, reg_beg, reg_end, whole_ref_len, is_ont=0, pal_flags=None, opt=None, cs=None, md=None, seqs=None, ref=None):
    """collect_digar_from_eqx_cigar (src/bam_utils.c:701) for a list of reads on the GPU: cigars[i] = uint32 BAM CIGAR words, quals[i] = phred bytes.
    The reference's three other sources (src/collect_var.c:1072-1079): cs=[bytes, ...] -> collect_digar_from_cs_tag (:844), md=[bytes, ...] ->
    collect_digar_from_MD_tag (:1010), seqs=[4-bit packed bases, ...] + ref=(ref_seq bytes, ref_beg, ref_end) -> collect_digar_from_ref_seq (:1179).
    -> list of dict(rc, digars (n,5), noisy (m,3), chunk_noisy (k,3), beg, end, n_cand), the layout of the oracle's wrapper"""
    lib = load_library()
    if opt is None:
        opt = LcdDigarOpt(); lib.lcd_digar_opt_default(C.byref(opt), int(is_ont))
    n = len(cigars)
    if n == 0:
        return []
    cg = [np.ascontiguousarray(c, np.uint32) for c in cigars]; ql = [np.ascontiguousarray(q, np.uint8) for q in quals]
    coff = np.concatenate([[0], np.cumsum([len(c) for c in cg])]).astype(np.uint64); qoff = np.concatenate([[0], np.cumsum([len(q) for q in ql])]).astype(np.uint64)
    cpool = np.concatenate(cg + [np.zeros(1, np.uint32)]); qpool = np.concatenate(ql + [np.zeros(1, np.uint8)])
    ncig = np.array([len(c) for c in cg], np.int32); qlen = np.array([len(q) for q in ql], np.int32)
    p0 = np.ascontiguousarray(pos0, np.int64)
    pf = np.ascontiguousarray(pal_flags if pal_flags is not None else np.zeros(n, np.uint8), np.uint8)
    status = np.zeros(n, np.int32); beg = np.zeros(n, np.int64); end = np.zeros(n, np.int64); ncand = np.zeros(n, np.int32)
    u64p_, i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    doff, ioff = u64p_(), u64p_(); dg = C.POINTER(LcdDigar)(); iv = C.POINTER(LcdNoisyIv)(); inc = u8p()
    head = (C.byref(opt), n, p0.ctypes.data_as(i64p), cpool.ctypes.data_as(C.POINTER(C.c_uint32)), coff.ctypes.data_as(u64p_), ncig.ctypes.data_as(i32p))
    quals_ = (_p8(qpool), qoff.ctypes.data_as(u64p_), qlen.ctypes.data_as(i32p), _p8(pf))
    tail = (int(reg_beg), int(reg_end), int(whole_ref_len), C.byref(doff), C.byref(dg), C.byref(ioff), C.byref(iv), C.byref(inc), status.ctypes.data_as(i32p),
            beg.ctypes.data_as(i64p), end.ctypes.data_as(i64p), ncand.ctypes.data_as(i32p))
    if cs is not None or md is not None:
        tags = (C.c_char_p * n)(*[bytes(t) for t in (cs if cs is not None else md)])
        check(lib.lcd_digar_batch_tags(head[0], 1 if cs is not None else 2, *head[1:], tags, *quals_, *tail), lib)
    elif seqs is not None:
        sq = [np.ascontiguousarray(x, np.uint8) for x in seqs]
        soff = np.concatenate([[0], np.cumsum([len(x) for x in sq])]).astype(np.uint64); spool = np.concatenate(sq + [np.zeros(1, np.uint8)])
        ref_seq, ref_beg, ref_end = ref
        check(lib.lcd_digar_batch_ref(*head, _p8(spool), soff.ctypes.data_as(u64p_), *quals_, C.c_char_p(bytes(ref_seq)), int(ref_beg), int(ref_end), *tail), lib)
    else:
        check(lib.lcd_digar_batch(*head, *quals_, *tail), lib)
    out = []
    nd_tot, ni_tot = int(doff[n]), int(ioff[n])
    D = np.frombuffer((C.c_char * (24 * max(nd_tot, 1))).from_address(C.addressof(dg.contents)), dtype=np.dtype([("pos", "<i8"), ("type", "<i4"), ("len", "<i4"), ("qi", "<i4"), ("lq", "<i4")]), count=nd_tot).copy() if nd_tot else np.zeros(0, [("pos", "<i8"), ("type", "<i4"), ("len", "<i4"), ("qi", "<i4"), ("lq", "<i4")])
    I = np.frombuffer((C.c_char * (24 * max(ni_tot, 1))).from_address(C.addressof(iv.contents)), dtype=np.dtype([("st", "<i8"), ("en", "<i8"), ("label", "<i4"), ("pad", "<i4")]), count=ni_tot).copy() if ni_tot else np.zeros(0, [("st", "<i8"), ("en", "<i8"), ("label", "<i4"), ("pad", "<i4")])
    INC = np.array([inc[k] for k in range(ni_tot)], np.uint8)
    for r in range(n):
        d = D[int(doff[r]):int(doff[r + 1])]; v = I[int(ioff[r]):int(ioff[r + 1])]; m = INC[int(ioff[r]):int(ioff[r + 1])]
        noisy = np.stack([v["st"], v["en"], v["label"].astype(np.int64)], 1).reshape(-1, 3) if len(v) else np.zeros((0, 3), np.int64)
        out.append(dict(rc=int(status[r]), digars=np.stack([d["pos"], d["type"].astype(np.int64), d["len"].astype(np.int64), d["qi"].astype(np.int64), d["lq"].astype(np.int64)], 1).reshape(-1, 5),
                        noisy=noisy, chunk_noisy=noisy[m.astype(bool)].reshape(-1, 3), beg=int(beg[r]), end=int(end[r]), n_cand=int(ncand[r])))
    for p in (doff, ioff, dg, iv, inc):
        if p:
            _libc.free(C.cast(p, C.c_void_p))
    return out


def region_read_slices_batch(pair_read, pair_reg_beg, pair_reg_end, digars, qlens, flank=10):
    """collect_noisy_read_info's digar walk (src/align.c:1392-1456) for many (region, read) pairs in one launch (lcd_region_read_slices_batch):
    digars[r] = (n, >= 4) rows (pos, type, len, qi) of read r -> (read_beg, read_end, cover) arrays, one entry per pair"""
    lib = load_library()
    u64p_, i64p = C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    n_reads = len(digars)
    off = np.zeros(n_reads + 1, np.uint64)
    for r in range(n_reads):
        off[r + 1] = off[r] + len(digars[r])
    pool = np.zeros((max(int(off[-1]), 1), 3), np.int64)      # lcd_digar_t: int64 pos | int32 type, len | int32 qi, is_low_qual
    v32 = pool.view(np.int32).reshape(len(pool), 6)
    for r in range(n_reads):
        d = np.asarray(digars[r], np.int64)
        if len(d):
            a, b = int(off[r]), int(off[r + 1])
            pool[a:b, 0] = d[:, 0]; v32[a:b, 2] = d[:, 1]; v32[a:b, 3] = d[:, 2]; v32[a:b, 4] = d[:, 3]
    pr = np.ascontiguousarray(pair_read, np.int32); pb = np.ascontiguousarray(pair_reg_beg, np.int64); pe = np.ascontiguousarray(pair_reg_end, np.int64)
    ql = np.ascontiguousarray(qlens, np.int32)
    n = len(pr)
    rb, re, cv = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    check(lib.lcd_region_read_slices_batch(n, pr.ctypes.data_as(i32p), pb.ctypes.data_as(i64p), pe.ctypes.data_as(i64p), n_reads, off.ctypes.data_as(u64p_),
                                           C.cast(pool.ctypes.data, C.POINTER(LcdDigar)), ql.ctypes.data_as(i32p), int(flank), rb.ctypes.data_as(i32p),
                                           re.ctypes.data_as(i32p), cv.ctypes.data_as(i32p)), lib)
    return rb[:n], re[:n], cv[:n]


def pre_process_noisy_regs(chunk_noisy, low_comp, read_beg, read_end, read_ivs, min_alt_dp=2, min_af=0.2):
    """pre_process_noisy_regs (src/collect_var.c:557): chunk_noisy (n,3) in cr_add order, low_comp (m,2), per read beg/end and its own (k,>=2)
    interval array -> surviving regions (r,3)"""
    lib = load_library()
    def ivarr(a):
        a = np.asarray(a, np.int64).reshape(-1, a.shape[1] if hasattr(a, "shape") and a.ndim == 2 else 3) if len(a) else np.zeros((0, 3), np.int64)
        arr = (LcdNoisyIv * max(len(a), 1))()
        for i, row in enumerate(a):
            arr[i].start, arr[i].end, arr[i].label = int(row[0]), int(row[1]), int(row[2]) if len(row) > 2 else 0
        return arr, len(a)
    cn, n_noisy = ivarr(np.asarray(chunk_noisy, np.int64).reshape(-1, 3))
    lc = np.ascontiguousarray(np.asarray(low_comp, np.int64).reshape(-1, 2)); n_low = len(lc)
    if n_low == 0:
        lc = np.zeros((1, 2), np.int64)
    rb = np.ascontiguousarray(read_beg, np.int64); re_ = np.ascontiguousarray(read_end, np.int64)
    n_reads = len(rb)
    off = np.concatenate([[0], np.cumsum([len(x) for x in read_ivs])]).astype(np.uint64) if n_reads else np.zeros(1, np.uint64)
    flat = np.concatenate([np.asarray(x, np.int64).reshape(-1, 3) for x in read_ivs] + [np.zeros((0, 3), np.int64)]) if n_reads else np.zeros((0, 3), np.int64)
    ri, _ = ivarr(flat)
    i64p, u64p_ = C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
    out = C.POINTER(LcdNoisyIv)()
    if n_reads == 0:
        rb = np.zeros(1, np.int64); re_ = np.zeros(1, np.int64)
    n = check(lib.lcd_pre_process_noisy_regs(cn, n_noisy, lc.ctypes.data_as(i64p), n_low, n_reads, rb.ctypes.data_as(i64p), re_.ctypes.data_as(i64p), off.ctypes.data_as(u64p_), ri,
                                             int(min_alt_dp), float(min_af), C.byref(out)), lib)
    res = np.array([[out[i].start, out[i].end, out[i].label] for i in range(n)], np.int64).reshape(-1, 3)
    if out:
        _libc.free(C.cast(out, C.c_void_p))
    return res


def post_process_noisy_regs(regs, var_pos, var_ref_len, var_cate, flank=10):
    """post_process_noisy_regs (src/collect_var.c:640): regs (n,3) -> flank-extended, merged regions (m,3)"""
    lib = load_library()
    r = np.asarray(regs, np.int64).reshape(-1, 3)
    arr = (LcdNoisyIv * max(len(r), 1))()
    for i, row in enumerate(r):
        arr[i].start, arr[i].end, arr[i].label = int(row[0]), int(row[1]), int(row[2])
    vp = np.ascontiguousarray(var_pos, np.int64); vl = np.ascontiguousarray(var_ref_len, np.int32); vc = np.ascontiguousarray(var_cate, np.int32)
    if len(vp) == 0:
        vp = np.zeros(1, np.int64); vl = np.zeros(1, np.int32); vc = np.zeros(1, np.int32); nv = 0
    else:
        nv = len(vp)
    out = C.POINTER(LcdNoisyIv)()
    n = check(lib.lcd_post_process_noisy_regs(arr, len(r), nv, vp.ctypes.data_as(C.POINTER(C.c_int64)), vl.ctypes.data_as(i32p), vc.ctypes.data_as(i32p), int(flank), C.byref(out)), lib)
    res = np.array([[out[i].start, out[i].end, out[i].label] for i in range(n)], np.int64).reshape(-1, 3)
    if out:
        _libc.free(C.cast(out, C.c_void_p))
    return res


def cr_merge(ivs, fixed_merge_win=-1):
    """lcd_cr_merge: cr_merge (src/cgranges.c:289) of (n,3) (start, end, label) intervals -> merged (m,3)"""
    lib = load_library()
    r = np.asarray(ivs, np.int64).reshape(-1, 3)
    arr = (LcdNoisyIv * max(len(r), 1))()
    for i, row in enumerate(r):
        arr[i].start, arr[i].end, arr[i].label = int(row[0]), int(row[1]), int(row[2])
    out = C.POINTER(LcdNoisyIv)()
    lib.lcd_cr_merge.argtypes = [C.POINTER(LcdNoisyIv), C.c_int, C.c_int, C.POINTER(C.POINTER(LcdNoisyIv))]
    n = lib.lcd_cr_merge(arr, len(r), int(fixed_merge_win), C.byref(out))
    res = np.array([[out[i].start, out[i].end, out[i].label] for i in range(n)], np.int64).reshape(-1, 3)
    if out:
        _libc.free(C.cast(out, C.c_void_p))
    return res


def sdust(seq, T=5, W=20):
    """low-complexity intervals (src/sdust.c) of a code / letter sequence on the GPU -> (n, 2) array of (start, finish)"""
    lib = load_library()
    a = np.ascontiguousarray(seq, np.uint8)
    out = C.POINTER(C.c_int64)()
    n = check(lib.lcd_sdust(_p8(a), len(a), int(T), int(W), C.byref(out)), lib)
    res = np.array([out[i] for i in range(2 * n)], np.int64).reshape(-1, 2)
    if out:
        _libc.free(C.cast(out, C.c_void_p))
    return res


def sdust_batch(seqs, T=5, W=20):
    """lcd_sdust_batch: the low-complexity intervals of many sequences in one launch -> list of (n, 2) arrays"""
    lib = load_library()
    arrs = [np.ascontiguousarray(x, np.uint8) for x in seqs]
    n = len(arrs)
    ptrs = (u8p * n)(*[_p8(a) for a in arrs])
    lens = (C.c_int64 * n)(*[len(a) for a in arrs])
    outs = (C.POINTER(C.c_int64) * n)(); cnt = (C.c_int * n)()
    check(lib.lcd_sdust_batch(n, ptrs, lens, int(T), int(W), outs, cnt), lib)
    res = []
    for q in range(n):
        res.append(np.array([outs[q][i] for i in range(2 * cnt[q])], np.int64).reshape(-1, 2))
        if outs[q]:
            _libc.free(C.cast(outs[q], C.c_void_p))
    return res


def _hap_state(prob):
    R, V, TA = prob["n_reads"], prob["n_vars"], int(prob["alle_off"][-1])
    return dict(haps=np.zeros(R, np.int32), phase_sets=np.full(R, -1, np.int64), n_clean_agree_snps=np.zeros(R, np.int32),
                n_clean_conflict_snps=np.zeros(R, np.int32), var_phase_set=np.full(V, -1, np.int64),
                hap_to_cons_alle=np.full(V * 3, -1, np.int32), hap_to_alle_profile=np.zeros(3 * TA, np.int32))


def _fill_hap_struct(S, prob, state, keep):
    i32p, i64p = C.POINTER(C.c_int), C.POINTER(C.c_int64)
    s = S()
    s.n_reads, s.n_vars, s.is_ont, s.n_cr = prob["n_reads"], prob["n_vars"], prob["is_ont"], len(prob["cr_read"])
    for name, ty in (("var_pos", i64p), ("var_type", i32p), ("var_cate", i32p), ("is_homopolymer_indel", i32p), ("total_cov", i32p),
                     ("alle_off", i32p), ("alle_covs", i32p), ("start_var_idx", i32p), ("end_var_idx", i32p), ("allele_off", i32p),
                     ("alleles", i32p), ("ordered_read_ids", i32p), ("cr_read", i32p)):
        a = np.ascontiguousarray(prob[name], np.int64 if ty is i64p else np.int32)
        if a.size == 0:
            a = np.zeros(1, a.dtype)
        keep.append(a)
        setattr(s, name, a.ctypes.data_as(ty))
    sk = np.ascontiguousarray(prob["is_skipped"], np.uint8)
    keep.append(sk)
    s.is_skipped = sk.ctypes.data_as(u8p)
    for name, ty in (("haps", i32p), ("phase_sets", i64p), ("n_clean_agree_snps", i32p), ("n_clean_conflict_snps", i32p), ("var_phase_set", i64p),
                     ("hap_to_cons_alle", i32p), ("hap_to_alle_profile", i32p)):
        setattr(s, name, state[name].ctypes.data_as(ty))
    return s


def assign_hap_germline(prob, target_var_cate, state=None):
    """assign_hap_based_on_germline_het_vars_kmeans (src/assign_hap.c:473) on a flattened chunk; returns the mutated state dict"""
    from ._lib import LcdHapProblem
    lib = load_library()
    state = state or _hap_state(prob)
    keep = []
    s = _fill_hap_struct(LcdHapProblem, prob, state, keep)
    check(lib.lcd_assign_hap_germline(C.byref(s), int(target_var_cate)), lib)
    return state


def assign_hap_batch(probs, target_var_cates, states=None):
    """lcd_assign_hap_batch: K5 on several chunks in one launch (one wavefront per chunk); returns the mutated state dicts"""
    from ._lib import LcdHapProblem
    lib = load_library()
    states = states or [_hap_state(p) for p in probs]
    keep = []
    arr = (LcdHapProblem * len(probs))(*[_fill_hap_struct(LcdHapProblem, p, st, keep) for p, st in zip(probs, states)])
    cates = np.ascontiguousarray(target_var_cates, np.int32)
    check(lib.lcd_assign_hap_batch(len(probs), arr, cates.ctypes.data_as(C.POINTER(C.c_int))), lib)
    return states
