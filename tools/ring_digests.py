"""Per-region digests of ONT-shape batches (bench.py's flow: warm-up submission, then the timed batches in one submission), to compare two environments.
usage: python tools/ring_digests.py <out.json> [seed] [n_batches]"""
import sys, json, zlib
import numpy as np
sys.path.insert(0, ".")
from longcalld_amd import align, jobs

out = sys.argv[1]
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20250928
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
n = 1250


def rdig(res):
    h = zlib.crc32(np.array([res["n_cons"]] + res["clu_n_seqs"], np.int32).tobytes())
    for c in range(2):
        if res["clu_read_ids"][c] is not None:
            h = zlib.crc32(res["clu_read_ids"][c].tobytes(), h)
        for s in res["aln_strs"][c]:
            if s is not None:
                h = zlib.crc32(s["target"].tobytes(), h); h = zlib.crc32(s["query"].tobytes(), h)
    return h


def run_many(list_of_regs, want):
    bs = []
    for rs in list_of_regs:
        b = align.RegionBatch()
        for r in rs:
            b.add_region(r)
        b.upload(); bs.append(b)
    align.RegionBatch.run_many(bs)
    digs = []
    for b, rs in zip(bs, list_of_regs):
        b.download()
        digs.append([rdig(b.result(i)) for i in range(len(rs))] if want else [])
    for b in bs:
        b.close()
    return digs


warm = [jobs.make_regions(seed + 500000 + i, n, jobs.ONT) for i in range(3)]
timed = [jobs.make_regions(seed + i, n, jobs.ONT) for i in range(nb)]
run_many(warm, False)
json.dump(run_many(timed, True), open(out, "w"))
print("done", out)
