#!/usr/bin/env python
"""Generates tests/golden/edlib_golden.json from the REFERENCE's own vendored edlib (oracle/_ref/libedlib_ref.so, built
from /root/reference/edlib/src/edlib.cpp by oracle/Makefile) called the way src/align.c:210-254 calls it.

Run in the build container (where /root/reference exists):  python tests/golden/make_edlib_golden.py
The fixture is data only: seeded inputs (as base-4 strings) and the expected distance / xgaps / n_eq / n_xid / path digest.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import mutate  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    assert pyoracle.ref_edlib() is not None, "oracle/_ref/libedlib_ref.so missing: run make -C oracle in the build container"
    rng = np.random.default_rng(20250928)
    cases = []
    for L in [1, 2, 3, 7, 31, 63, 64, 65, 127, 128, 129, 200, 500, 1000, 1800, 2000, 2600, 4100]:
        for rate in [0.0, 0.01, 0.05, 0.2]:
            t = rng.integers(0, 4, L).astype(np.uint8)
            q = mutate(rng, t, rate)
            if len(q) == 0:
                q = np.array([0], np.uint8)
            if L % 3 == 0:
                q = q[: max(1, int(len(q) * 0.7))]
            d, ops = pyoracle.ref_edlib_nw(q, t)
            cases.append(dict(target="".join(map(str, t)), query="".join(map(str, q)), dist=int(d), xgaps=int(pyoracle.ops_to_xgaps(ops)),
                              n_eq=int((ops == 0).sum()), n_xid=int((ops != 0).sum()), path_sha1=hashlib.sha1(ops.tobytes()).hexdigest()))
    # low-complexity / homopolymer cases (tie-heavy tracebacks)
    for t, q in [(np.zeros(300, np.uint8), np.zeros(290, np.uint8)), (np.tile(np.array([0, 1], np.uint8), 150), np.tile(np.array([0, 1, 1], np.uint8), 100)),
                 (np.tile(np.array([2, 2, 3], np.uint8), 700), np.tile(np.array([2, 3], np.uint8), 1000))]:
        d, ops = pyoracle.ref_edlib_nw(q, t)
        cases.append(dict(target="".join(map(str, t)), query="".join(map(str, q)), dist=int(d), xgaps=int(pyoracle.ops_to_xgaps(ops)),
                          n_eq=int((ops == 0).sum()), n_xid=int((ops != 0).sum()), path_sha1=hashlib.sha1(ops.tobytes()).hexdigest()))
    # HW (infix) mode, edlib_infix_aln (src/align.c:256-275): the query is a mutated stretch of a longer target (free start / end in the target)
    hw_cases = []
    rng2 = np.random.default_rng(20250929)
    for L in [1, 2, 5, 30, 63, 64, 65, 130, 400, 900, 1500, 2600]:
        for rate in [0.0, 0.02, 0.1, 0.3]:
            t = rng2.integers(0, 4, 3 * L + int(rng2.integers(0, 70))).astype(np.uint8)
            a = int(rng2.integers(0, max(1, len(t) - L)))
            q = mutate(rng2, t[a:a + L], rate)
            if len(q) == 0:
                q = np.array([1], np.uint8)
            d, s0, e0, ops = pyoracle.ref_edlib_hw(q, t)
            hw_cases.append(dict(target="".join(map(str, t)), query="".join(map(str, q)), dist=int(d), start=int(s0), end=int(e0), xgaps=int(pyoracle.ops_to_xgaps(ops)),
                                 n_eq=int((ops == 0).sum()), n_xid=int((ops != 0).sum()), path_sha1=hashlib.sha1(ops.tobytes()).hexdigest()))
    for t, q in [(np.zeros(300, np.uint8), np.zeros(40, np.uint8)), (np.tile(np.array([0, 1], np.uint8), 200), np.tile(np.array([0, 1, 1], np.uint8), 30)),
                 (np.tile(np.array([2, 2, 3], np.uint8), 300), np.tile(np.array([2, 3], np.uint8), 100)), (rng2.integers(0, 4, 50).astype(np.uint8), rng2.integers(0, 4, 300).astype(np.uint8))]:
        d, s0, e0, ops = pyoracle.ref_edlib_hw(q, t)
        hw_cases.append(dict(target="".join(map(str, t)), query="".join(map(str, q)), dist=int(d), start=int(s0), end=int(e0), xgaps=int(pyoracle.ops_to_xgaps(ops)),
                             n_eq=int((ops == 0).sum()), n_xid=int((ops != 0).sum()), path_sha1=hashlib.sha1(ops.tobytes()).hexdigest()))
    # round 4: the corners of HW mode -- an empty query, a query longer than its target, a target of one base, and distances on either side of
    # the band doublings of edlibAlign (k = 64, 128, 256: edlib/src/edlib.cpp:185-200 starts at 64 and doubles until the distance fits)
    rng3 = np.random.default_rng(20250931)
    extra = [(rng3.integers(0, 4, 40).astype(np.uint8), np.zeros(0, np.uint8)), (rng3.integers(0, 4, 30).astype(np.uint8), rng3.integers(0, 4, 100).astype(np.uint8)),
             (rng3.integers(0, 4, 64).astype(np.uint8), rng3.integers(0, 4, 65).astype(np.uint8)), (np.array([2], np.uint8), np.array([2], np.uint8)), (np.array([2], np.uint8), np.array([1, 3, 0], np.uint8))]
    for d_want in (62, 63, 64, 65, 66, 126, 127, 128, 129, 130, 255, 256, 257):
        L = 5 * d_want + 40
        t = rng3.integers(0, 4, 3 * L).astype(np.uint8)
        a = int(rng3.integers(10, len(t) - L - 10))
        q = t[a:a + L].copy()
        for ppos in np.linspace(2, L - 3, d_want).astype(int):     # d_want substitutions, evenly spread: the infix distance is d_want (or just below it)
            q[ppos] = (q[ppos] + 1 + int(rng3.integers(0, 3))) % 4
        extra.append((t, q))
    # 64 / 65 / 128 / 129-base queries (one / two / three Myers blocks exactly full) against unrelated targets: distances close to the query length
    for L in (63, 64, 65, 127, 128, 129):
        extra.append((rng3.integers(0, 4, 4 * L).astype(np.uint8), rng3.integers(0, 4, L).astype(np.uint8)))
    for t, q in extra:
        d, s0, e0, ops = pyoracle.ref_edlib_hw(q, t)
        hw_cases.append(dict(target="".join(map(str, t)), query="".join(map(str, q)), dist=int(d), start=int(s0), end=int(e0), xgaps=int(pyoracle.ops_to_xgaps(ops)),
                             n_eq=int((ops == 0).sum()), n_xid=int((ops != 0).sum()), path_sha1=hashlib.sha1(ops.tobytes()).hexdigest()))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "edlib_golden.json")
    json.dump(dict(source="reference edlib (edlib/src/edlib.cpp) via oracle/_ref, NW + TASK_PATH, k=-1; hw_cases: HW + TASK_PATH", cases=cases, hw_cases=hw_cases), open(out, "w"))
    print(len(cases), "NW cases,", len(hw_cases), "HW cases ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
