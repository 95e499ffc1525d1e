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
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "edlib_golden.json")
    json.dump(dict(source="reference edlib (edlib/src/edlib.cpp) via oracle/_ref, NW + TASK_PATH, k=-1", cases=cases), open(out, "w"))
    print(len(cases), "cases ->", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
