#!/usr/bin/env python
"""Generates tests/golden/cgranges_golden.json from the REFERENCE's own interval index (src/cgranges.c compiled from where it lies into
oracle/_ref/libcgranges_ref.so by oracle/Makefile): cr_index order, cr_overlap hit order, cr_merge with the two parameter sets src/collect_var.c uses
(:552/:568: (-1, noisy_reg_merge_dis, min_sv_len); :657: (0, -1, -1)), and whole pre_ / post_process_noisy_regs cases (control flow restated in
oracle/ref_cgranges_shim.c, every interval operation the reference's).

Run in the build container (where /root/reference exists):  python tests/golden/make_cgranges_golden.py
The fixture is data only: inputs and the outputs the reference's code gave."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

i32p = C.POINTER(C.c_int)


def ref_merge(st, en, lab, fixed, dyn, lmin):
    L = pyoracle.ref_cgranges()
    L.ref_cr_merge.argtypes = [C.c_int, i32p, i32p, i32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int]
    st = np.ascontiguousarray(st, np.int32); en = np.ascontiguousarray(en, np.int32); lab = np.ascontiguousarray(lab, np.int32)
    out = np.zeros(3 * (len(st) + 1), np.int32)
    n = L.ref_cr_merge(len(st), st.ctypes.data_as(i32p), en.ctypes.data_as(i32p), lab.ctypes.data_as(i32p), fixed, dyn, lmin, out.ctypes.data_as(i32p), len(st) + 1)
    return out[:3 * n].reshape(-1, 3).tolist()


def main():
    assert pyoracle.ref_cgranges() is not None, "oracle/_ref/libcgranges_ref.so missing: run make -C oracle in the build container"
    rng = np.random.default_rng(20250930)
    order, overlap, merge, post, pre = [], [], [], [], []
    # ---- cr_index order: ties on the start, > 64 intervals, negative starts (clamped to 0), st > en (dropped), already sorted input ----
    hand = [([5], [9]), ([5, 5, 5], [9, 7, 8]), ([10, 0, 10, 0], [20, 5, 12, 5]), ([500, -100, 20], [510, 30, 25]), ([3, 9, 1], [2, 12, 4]), (list(range(0, 300, 3)), list(range(4, 304, 3)))]
    for n, ties in [(2, True), (40, True), (64, False), (65, True), (200, False), (700, True)]:
        st = rng.integers(0, 40 if ties else 1_000_000, n); hand.append((st.tolist(), (st + rng.integers(1, 500, n)).tolist()))
    for st, en in hand:
        order.append(dict(st=st, en=en, order=pyoracle.ref_cr_sorted_order(st, en).tolist()))
    # ---- cr_overlap: half-open ends, containment, identical intervals, nothing hit, everything hit ----
    sets = [([10, 20, 30], [20, 30, 40]), ([0, 0, 0, 5], [10, 10, 10, 6]), ([100, 50, 75, 60], [200, 300, 80, 61]), ([1], [2])]
    for n in (50, 400):
        st = rng.integers(0, 5000, n); sets.append((st.tolist(), (st + rng.integers(1, 300, n)).tolist()))
    for st, en in sets:
        qs = [(0, 1), (10, 10), (10, 11), (19, 20), (20, 21), (29, 31), (40, 50), (0, 10 ** 6), (55, 56), (199, 200), (200, 201), (300, 301)]
        qs += [(int(a), int(a + b)) for a, b in zip(rng.integers(0, 5000, 12), rng.integers(1, 400, 12))]
        for q0, q1 in qs:
            overlap.append(dict(st=st, en=en, q=[q0, q1], hits=pyoracle.ref_cr_overlap(st, en, q0, q1).tolist()))
    # ---- cr_merge: touching / nested / chains that need several passes; label-dependent distance (min of the two labels, capped by the window; labels below the
    # minimum do not merge by distance) ----
    msets = [([10, 20, 30], [20, 30, 40], [6, 6, 6]), ([10, 21, 32], [20, 31, 42], [6, 6, 6]), ([0, 100, 250, 700], [50, 200, 300, 800], [60, 60, 40, 300]),
             ([0, 1000, 1400, 5000], [900, 1100, 1500, 5100], [600, 35, 700, 29]), ([5, 5, 5], [6, 7, 8], [1, 2, 3]), ([0, 10, 20, 30, 40, 55], [5, 15, 25, 35, 45, 60], [30, 8, 30, 4, 30, 30]),
             ([100, 100], [100, 100], [50, 50]), ([0, 499, 1000], [1, 500, 1001], [500, 500, 500]), ([0, 502, 1004], [1, 503, 1005], [500, 500, 500])]
    for n in (30, 200, 900):
        st = np.sort(rng.integers(0, 40000, n)); msets.append((st.tolist(), (st + rng.integers(1, 300, n)).tolist(), rng.choice([6, 12, 29, 30, 31, 60, 300, 700], n).tolist()))
    for st, en, lab in msets:
        for fixed, dyn, lmin in ((0, -1, -1), (-1, 500, 30), (-1, 100, 30), (-1, 500, 1), (25, -1, -1)):
            merge.append(dict(st=st, en=en, label=lab, args=[fixed, dyn, lmin], merged=ref_merge(st, en, lab, fixed, dyn, lmin)))
    # ---- post_process_noisy_regs: flank growth along candidate variants, merge ----
    for trial in range(25):
        n = int(rng.integers(1, 40))
        st = np.sort(rng.integers(1000, 100000, n)); regs = np.stack([st, st + rng.integers(5, 400, n), rng.integers(6, 300, n)], 1)
        nv = int(rng.integers(0, 200))
        vp = np.sort(rng.integers(900, 101000, nv)); vl = rng.choice([0, 1, 1, 1, 12, 40], nv); vc = rng.choice([0x004, 0x008, 0x080, 0x800, 0x001, 0x002, 0x100], nv)
        for flank in (10, 0, 50):
            post.append(dict(regs=regs.tolist(), var_pos=vp.tolist(), var_ref_len=vl.tolist(), var_cate=vc.tolist(), flank=flank,
                             out=pyoracle.ref_post_process_noisy_regs(regs, vp, vl, vc, flank).tolist()))
    # ---- pre_process_noisy_regs: read support, low-complexity extension, label-dependent merging ----
    for trial in range(8):
        n_reads = int(rng.integers(5, 120))
        centers = rng.integers(1000, 200000, int(rng.integers(3, 40)))
        rb, re_, ivs, cn = [], [], [], []
        for r in range(n_reads):
            b = int(rng.integers(0, 180000)); e = b + int(rng.integers(2000, 30000))
            mine = []
            for c in centers:
                if b < c < e and rng.random() < 0.5:
                    s = int(c + rng.integers(-3, 4) * (trial % 2)); mine.append([s, s + int(rng.integers(5, 400)), int(rng.choice([6, 12, 60, 300, 700]))])
            mine.sort()
            rb.append(b + 1); re_.append(e); ivs.append(mine); cn += mine
        low = np.stack([centers[:5] - 50, centers[:5] + 80], 1).tolist() if trial % 3 == 0 else []
        for min_dp, min_af in ((2, 0.2), (1, 0.0), (5, 0.5)):
            exp = pyoracle.ref_pre_process_noisy_regs(np.array(cn, np.int64).reshape(-1, 3), np.array(low, np.int64).reshape(-1, 2), rb, re_,
                                                      [np.array(x, np.int64).reshape(-1, 3) for x in ivs], min_dp, min_af)
            pre.append(dict(chunk_noisy=cn, low_comp=low, read_beg=rb, read_end=re_, read_ivs=ivs, min_alt_dp=min_dp, min_af=min_af, out=exp.tolist()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cgranges_golden.json")
    json.dump(dict(source="reference src/cgranges.c (+ kalloc.c) via oracle/_ref/libcgranges_ref.so; control flow of pre_/post_process_noisy_regs restated in oracle/ref_cgranges_shim.c",
                   order=order, overlap=overlap, merge=merge, post_process=post, pre_process=pre), open(out, "w"), separators=(",", ":"))
    print(len(order), "order,", len(overlap), "overlap,", len(merge), "merge,", len(post), "post,", len(pre), "pre ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
