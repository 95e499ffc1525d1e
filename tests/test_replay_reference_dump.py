"""Replay hook for golden vectors of a REAL longcallD build: `longcallD call -V 3 ... 2> dump.txt` prints, per noisy region, the
reads handed to abPOA, the sub-graph windows, the consensus lengths and every alignment string.  Drop such dumps under tests/golden/v3_dumps/*.txt and
the GPU test below re-runs every K1 / K2 chain through the HIP path and compares the cons<->read and ref<->cons rows byte for byte -- that is what
turns "parity unpinned" (K1 / K2 / K3) into pinned.  No dump can be produced in this project's container (abPOA / WFA2-lib / htslib are absent
from the reference checkout), so the GPU test skips until one exists; the parser itself is tested here on text in the reference's formats."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT
import replay_dump as rd

DUMPS = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "v3_dumps", "*.txt")))


def test_parser_round_trips_the_reference_formats():
    rng = np.random.default_rng(3)
    reads = [dict(name=f"m64011_{i}/ccs", seq=rng.integers(0, 5, 40 + i).astype(np.uint8), cover=12 if i else 8, exc=(2 + i, 45, 0, i % 3)) for i in range(4)]
    chains = [dict(mode=0, ps=1240031, hap=1, reads=reads[:2], cons_len=[41]), dict(mode=0, ps=1240031, hap=2, reads=reads[2:], cons_len=[43])]
    row = lambda n: rng.integers(0, 6, n).astype(np.uint8)
    strings = [dict(kind="ref_cons", target=row(50), query=row(50), tb=0, te=49, qb=0, qe=49), dict(kind="cons_read", target=row(47), query=row(47), tb=0, te=46, qb=3, qe=46)]
    text = rd.format_region("Hap", "chr11", 1240001, 1240050, chains, strings, 4, 3, 2)
    text = ["abPOA noise line >Consensus_sequence", "ACGT-ACGT"] + text + rd.format_region("Skipped", "chr11", 1250001, 1250100, [], [], 3, 1, 0)
    regs = rd.parse(text)
    assert len(regs) == 2 and regs[0]["kind"] == "Hap" and regs[0]["n_cons"] == 2 and regs[1]["kind"] == "Skipped" and not regs[1]["chains"]
    got = regs[0]
    assert [c["hap"] for c in got["chains"]] == [1, 2] and [len(c["reads"]) for c in got["chains"]] == [2, 2] and got["chains"][1]["cons_len"] == [43]
    for c, e in zip(got["chains"], chains):
        for a, b in zip(c["reads"], e["reads"]):
            assert a["name"] == b["name"] and a["cover"] == b["cover"] and (a["seq"] == b["seq"]).all() and a["exc"] == b["exc"]
    assert [s["kind"] for s in got["strings"]] == ["ref_cons", "cons_read"]
    assert (got["strings"][1]["query"] == strings[1]["query"]).all() and got["strings"][1]["qb"] == 3


@pytest.mark.gpu
@pytest.mark.skipif(not DUMPS, reason="no `longcallD call -V 3` dump under tests/golden/v3_dumps/ (the reference binary cannot be built in this container)")
def test_replay_reference_dumps(lcd):
    n_chains = 0
    for path in DUMPS:
        for reg in rd.parse(open(path).read().splitlines()):
            cr = [s for s in reg["strings"] if s["kind"] == "cons_read"]
            k = 0
            for ch in reg["chains"]:
                if not ch["reads"]:
                    continue
                job = dict(mode=ch["mode"], reads=[r["seq"] for r in ch["reads"]])
                if ch["mode"] == 0:   # sub-graph windows as the reference computed them: beg_id = ExcBeg + 1 ... (src/align.c:797-803)
                    job["anchors"] = [(1, len(ch["reads"][0]["seq"]), 1, len(r["seq"])) if r["exc"] is None else
                                      (r["exc"][0] + 1 - 1, r["exc"][1] - 1 - 1, 1 + r["exc"][2], len(r["seq"]) - r["exc"][3]) for r in ch["reads"]]
                g = lcd.poa_batch([job])[0]
                assert g["status"] == 0
                if ch["cons_len"]:
                    assert [len(c) for c in g["cons"][: len(ch["cons_len"])]] == ch["cons_len"], (path, reg["beg"])
                # consensus <-> read rows: the MSA row pair with the columns where both are gaps dropped (make_cons_read_aln_str, src/align.c:1029)
                for c in range(g["n_cons"]):
                    crow = g["msa"][len(ch["reads"]) + c]
                    for r in (g["clu"][c] if ch["mode"] == 1 else range(len(ch["reads"]))):
                        if k >= len(cr):
                            break
                        keep = ~((crow == 5) & (g["msa"][int(r)] == 5))
                        t, q = crow[keep], g["msa"][int(r)][keep]
                        if len(t) == len(cr[k]["target"]):     # (full-cover reads: untrimmed rows)
                            assert (t == cr[k]["target"]).all() and (q == cr[k]["query"]).all(), (path, reg["beg"], k)
                        k += 1
                n_chains += 1
    assert n_chains > 0
