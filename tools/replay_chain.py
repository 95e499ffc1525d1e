"""Replay a chain the library dumped on a POA kernel error (env LCD_DUMP_CHAIN=<file> while the failing command runs) in isolation through lcd_poa_batch, and
compare with the oracle.  usage: python tools/replay_chain.py <dump> [--oracle]"""
import sys, struct
import numpy as np
sys.path.insert(0, ".")
from longcalld_amd import align

d = open(sys.argv[1], "rb").read()
magic, mode, n_reads, status, threads, wmax, ring_k, lds = struct.unpack_from("<8i", d, 0)
assert magic == 0x4c434443
o = 32
reads, skip, anch = [], [], []
for q in range(n_reads):
    ln, sk, rb, re, qb, qe = struct.unpack_from("<6i", d, o); o += 24
    reads.append(np.frombuffer(d, np.uint8, ln, o).copy()); o += ln
    skip.append(sk); anch.append((rb, re, qb, qe))
print(f"chain: mode {mode}, {n_reads} reads, lens {[len(r) for r in reads]}, skip {skip}, status {status}, threads {threads}, window {wmax}, ring slots {ring_k}, LDS {lds}")
print("anchors", anch)
opt = align.default_opt()
if "--ont" in sys.argv:
    opt.is_ont = 1   # (the ring-slot rule of noisy K1 chains: lcd_host.cpp chain_class)
got = align.poa_batch([dict(mode=mode, reads=reads, skip=skip, anchors=anch)], opt)[0]
print("replay status", got["status"], "n_cons", got["n_cons"], "msa_len", got["msa_len"])
if "--json" in sys.argv:
    import json, zlib
    rows_ok = got["status"] == 0 and all(np.array_equal(row[row != 5], r[a[2] - 1:a[3]] if (mode == 0 and q > 0) else r) for q, (r, row, sk, a) in enumerate(zip(reads, got["msa"], skip, anch)) if not sk)  # (K1: a later read is aligned over its anchored slice)
    crc = 0
    for row in got["msa"] + got["cons"]:
        crc = zlib.crc32(np.ascontiguousarray(row).tobytes(), crc)
    json.dump(dict(status=got["status"], n_cons=got["n_cons"], msa_len=got["msa_len"], rows_degap_to_reads=bool(rows_ok), crc=crc), open(sys.argv[sys.argv.index("--json") + 1], "w"))
if "--oracle" in sys.argv:
    from oracle import pyoracle
    if mode == 1:
        exp = pyoracle.poa_aln_msa_cons([r for r, s in zip(reads, skip) if not s], 2)
    else:
        exp = pyoracle.poa_partial_aln_msa_cons_anchored(reads, anch, skip)
    same = got["status"] == 0 and got["n_cons"] == exp["n_cons"] and got["msa_len"] == exp["msa_len"] and all(np.array_equal(a, b) for a, b in zip(got["msa"], exp["msa"])) \
        and all(np.array_equal(got["cons"][c], exp["cons"][c]) for c in range(exp["n_cons"]))
    print("oracle:", {k: (v if np.isscalar(v) else len(v)) for k, v in exp.items()}, "== replay:", same)
    if "--json" in sys.argv:
        import json
        f = sys.argv[sys.argv.index("--json") + 1]
        j = json.load(open(f)); j["equals_oracle"] = bool(same); json.dump(j, open(f, "w"))
