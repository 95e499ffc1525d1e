"""Parser for the per-region debug output of a real `longcallD call -V 3` run (the only built-in way to capture hot-path golden vectors from a
reference binary).  What is parsed is what the reference prints unconditionally at that verbosity, in these formats:

  K1 chains, one per haplotype      "PS: <ps> HAP: <hap> n_reads: <n>"                                  src/align.c:1332
    each read handed to abPOA       ">%s %d %d\\n<bases ACGTN>"   name, length, cover flag               src/align.c:786-791
    its sub-graph window            "ExcBeg: %d, ExcEnd: %d, SeqBegCut: %d, SeqEndCut: %d, FullCover: %d" src/align.c:802
    consensus lengths               "ConsLen: %d"                                                        src/align.c:822,906
  K2 chains                         "For abPOA (max %d cons, min_freq: %.2f): %d"                        src/align.c:892
  alignment strings                 ">target %d:%d-%d\\n<row>\\n>query %d:%d-%d\\n<row>"   ref<->cons        src/align.c:600-606
                                    ">target %d-%d\\n<row>\\n>query %d-%d\\n<row>"         cons<->read       src/align.c:1046-1052
  region summary                    "Hap|NoHap|Skipped <chrom>:<beg>-<end> <len> <n> reads (<full> full) [n_cons: <k>]"  src/align.c:1791-1800

abPOA's own MSA dump (abpoa_output at -V 2) sits between these and is skipped: its format belongs to a library that is absent from the
reference checkout.  Everything here is text handling; tests/test_replay_reference_dump.py feeds the parsed chains to the GPU path.
"""
import re

import numpy as np

_CODE = {c: i for i, c in enumerate("ACGTN-")}


def _row(s):
    return np.array([_CODE[c] for c in s.strip()], np.uint8)


def parse(lines):
    """-> list of regions: dict(kind, chrom, beg, end, n_reads, n_full, n_cons, chains=[dict(mode, ps, hap, reads=[dict(name, len, cover, seq, exc)], cons_len=[...])],
    strings=[dict(kind 'ref_cons'|'cons_read', target, query, tb, te, qb, qe)])"""
    regions, chains, strings = [], [], []
    cur = None
    i, n = 0, len(lines)
    while i < n:
        ln = lines[i].rstrip("\n")
        m = re.match(r"PS: (-?\d+) HAP: (\d) n_reads: (\d+)", ln)
        if m:
            cur = dict(mode=0, ps=int(m.group(1)), hap=int(m.group(2)), n_reads=int(m.group(3)), reads=[], cons_len=[])
            chains.append(cur); i += 1; continue
        m = re.match(r"For abPOA \(max (\d+) cons, min_freq: ([0-9.]+)\): (\d+)", ln)
        if m:
            cur = dict(mode=1, ps=-1, hap=0, n_reads=int(m.group(3)), reads=[], cons_len=[])
            chains.append(cur); i += 1; continue
        m = re.match(r">target (\d+):(-?\d+)-(-?\d+)$", ln)
        if m and i + 3 < n and lines[i + 2].startswith(">query"):
            q = re.match(r">query (\d+):(-?\d+)-(-?\d+)", lines[i + 2])
            strings.append(dict(kind="ref_cons", target=_row(lines[i + 1]), query=_row(lines[i + 3]), tb=int(m.group(2)), te=int(m.group(3)), qb=int(q.group(2)), qe=int(q.group(3))))
            i += 4; continue
        m = re.match(r">target (-?\d+)-(-?\d+)$", ln)
        if m and i + 3 < n and lines[i + 2].startswith(">query"):
            q = re.match(r">query (-?\d+)-(-?\d+)", lines[i + 2])
            strings.append(dict(kind="cons_read", target=_row(lines[i + 1]), query=_row(lines[i + 3]), tb=int(m.group(1)), te=int(m.group(2)), qb=int(q.group(1)), qe=int(q.group(2))))
            i += 4; continue
        m = re.match(r">(\S+) (\d+) (\d+)$", ln)
        if m and cur is not None and i + 1 < n and re.fullmatch(r"[ACGTN]*", lines[i + 1].strip()) and len(lines[i + 1].strip()) == int(m.group(2)):
            cur["reads"].append(dict(name=m.group(1), len=int(m.group(2)), cover=int(m.group(3)), seq=_row(lines[i + 1])[: int(m.group(2))], exc=None))
            i += 2; continue
        m = re.match(r"ExcBeg: (-?\d+), ExcEnd: (-?\d+), SeqBegCut: (\d+), SeqEndCut: (\d+), FullCover: (\d+)", ln)
        if m and cur is not None and cur["reads"]:
            cur["reads"][-1]["exc"] = tuple(int(x) for x in m.groups()[:4]); i += 1; continue
        m = re.match(r"ConsLen: (\d+)", ln)
        if m and cur is not None:
            cur["cons_len"].append(int(m.group(1))); i += 1; continue
        m = re.match(r"(Hap|NoHap|Skipped) (\S+):(\d+)-(\d+) (\d+) (\d+) reads \((\d+) full\)(?: n_cons: (\d+))?", ln)
        if m:
            regions.append(dict(kind=m.group(1), chrom=m.group(2), beg=int(m.group(3)), end=int(m.group(4)), n_reads=int(m.group(6)), n_full=int(m.group(7)),
                                n_cons=int(m.group(8)) if m.group(8) else 0, chains=chains, strings=strings))
            chains, strings, cur = [], [], None
            i += 1; continue
        i += 1
    return regions


def format_region(kind, chrom, beg, end, chains, strings, n_reads, n_full, n_cons):
    """the inverse, in the reference's formats: used to test the parser on this project's own outputs"""
    out = []
    for ch in chains:
        out.append(f"PS: {ch['ps']} HAP: {ch['hap']} n_reads: {len(ch['reads'])}" if ch["mode"] == 0 else f"For abPOA (max 2 cons, min_freq: 0.20): {len(ch['reads'])}")
        for r in ch["reads"]:
            out.append(f">{r['name']} {len(r['seq'])} {r['cover']}")
            out.append("".join("ACGTN"[b] for b in r["seq"]))
            if ch["mode"] == 0 and r.get("exc"):
                out.append("ExcBeg: %d, ExcEnd: %d, SeqBegCut: %d, SeqEndCut: %d, FullCover: %d" % (*r["exc"], r["cover"]))
        for cl in ch["cons_len"]:
            out.append(f"ConsLen: {cl}")
    for s in strings:
        t, q = "".join("ACGTN-"[b] for b in s["target"]), "".join("ACGTN-"[b] for b in s["query"])
        if s["kind"] == "ref_cons":
            out += [f">target {len(t.replace('-', ''))}:{s['tb']}-{s['te']}", t, f">query {len(q.replace('-', ''))}:{s['qb']}-{s['qe']}", q]
        else:
            out += [f">target {s['tb']}-{s['te']}", t, f">query {s['qb']}-{s['qe']}", q]
    out.append(f"{kind} {chrom}:{beg}-{end} {end - beg + 1} {n_reads} reads ({n_full} full)" + (f" n_cons: {n_cons}" if kind != "Skipped" else ""))
    return out
