"""Builds tests/golden/testdata_chunk.npz from the reference's bundled real input:
    /root/reference/test_data/chr11_2M.fa + HG002_chr11_hifi_test.bam   (HG002 HiFi, EQX CIGARs)
Run in the build container only (`python tests/golden/make_testdata_fixture.py`); the GPU box has no /root/reference.

What is DATA from the reference's test input: read positions, CIGAR-derived digars (the layout of digar1_t, src/bam_utils.h:27-33, as
collect_digar_from_eqx_cigar src/bam_utils.c:701 fills it), read bases (4-bit BAM packing) and qualities, the reference slice.
What is NOT the reference's code: the noisy-window detector, the candidate-SNP pile-up and the read x variant profile below are a
plain-Python APPROXIMATION of rows f2/f1 (src/bam_utils.c:161-200, src/collect_var.c) that exists only to produce realistic inputs for
the hot path (regions with real reads; a real read x het-variant profile for K5).  No parity is claimed for those steps.
Expected outputs are the oracle's (oracle/, parity unpinned for K1-K3/K5 -- see DESIGN.md section 2): haplotypes/phase sets of K5 on the
profile, then per region the results of collect_noisy_reg_aln_strs with those haplotypes.  Bases and qualities outside the region
slices are zeroed (the path never reads them) so that the fixture stays small.
"""
import gzip
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TD = "/root/reference/test_data"
OUT = os.path.join(ROOT, "tests", "golden", "testdata_chunk.npz")

CEQ, CX, CI, CD, CS, CH, CN = 7, 8, 1, 2, 4, 5, 3
NT16 = np.array([4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4], np.uint8)  # htslib seq_nt16_int
MIN_BQ, WIN, MAX_XGAPS, FLANK = 10, 100, 5, 10


def read_fasta(path):
    seq = []
    with open(path) as f:
        for line in f:
            if line[0] != ">":
                seq.append(line.strip())
    s = np.frombuffer("".join(seq).upper().encode(), np.uint8)
    code = np.full(256, 4, np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    return code[s]


def read_bam(path):
    d = gzip.open(path).read()
    assert d[:4] == b"BAM\x01"
    lt, = struct.unpack_from("<i", d, 4)
    o = 8 + lt
    nref, = struct.unpack_from("<i", d, o); o += 4
    names = []
    for _ in range(nref):
        ln, = struct.unpack_from("<i", d, o); o += 4
        names.append(d[o:o + ln - 1].decode()); o += ln + 4
    reads = []
    while o < len(d):
        bs, = struct.unpack_from("<i", d, o); o += 4
        refid, pos, lname, mapq, _bin, ncig, flag, lseq = struct.unpack_from("<iiBBHHHi", d, o)
        p = o + 32 + lname
        cig = np.frombuffer(d, "<u4", ncig, p); p += 4 * ncig
        bseq = np.frombuffer(d, np.uint8, (lseq + 1) // 2, p).copy(); p += (lseq + 1) // 2
        qual = np.frombuffer(d, np.uint8, lseq, p).copy()
        o += bs
        if flag & 0x904 or names[refid] != "chr11":
            continue
        reads.append(dict(pos=pos + 1, cigar=cig.copy(), bseq=bseq, qual=qual, qlen=lseq, mapq=mapq))
    return reads


def digars_of(r):
    """digar1_t list of one read (pos 1-based, type, len, qi) + its events for the noisy-window scan"""
    pos, qi, out, ev = r["pos"], 0, [], []
    q = r["qual"]
    for c in r["cigar"]:
        op, ln = int(c & 0xf), int(c >> 4)
        if op == CX:
            for j in range(ln):
                out.append((pos, CX, 1, qi))
                if q[qi] >= MIN_BQ:
                    ev.append((pos, 1, 1))
                pos += 1; qi += 1
        elif op == CEQ:
            out.append((pos, CEQ, ln, qi)); pos += ln; qi += ln
        elif op == CD:
            out.append((pos, CD, ln, qi))
            if (qi == 0 or q[qi - 1] >= MIN_BQ) and q[min(qi, len(q) - 1)] >= MIN_BQ:
                ev.append((pos, ln, ln))
            pos += ln
        elif op == CI:
            out.append((pos, CI, ln, qi))
            if (q[qi:qi + ln] >= MIN_BQ).any():
                ev.append((pos, 0, ln))
            qi += ln
        elif op in (CS, CH):
            out.append((pos, op, ln, qi))
            if op == CS:
                qi += ln
        elif op == CN:
            pos += ln
        else:
            raise ValueError("M op in an EQX BAM")
    r["end"] = pos - 1
    return np.array(out, np.int64), ev


def noisy_windows(ev):
    """> MAX_XGAPS differing bases inside a WIN-bp window -> [first, last] (approximation of push_xid_size_queue_win)"""
    res, i, tot = [], 0, 0
    for j in range(len(ev)):
        tot += ev[j][2]
        while ev[j][0] - ev[i][0] >= WIN:
            tot -= ev[i][2]; i += 1
        if tot > MAX_XGAPS:
            res.append((ev[i][0], ev[j][0] + ev[j][1]))
    return res


def merge(iv, gap=0):
    iv = sorted(iv); out = []
    for b, e in iv:
        if out and b <= out[-1][1] + gap:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([b, e])
    return out


def main():
    from oracle import pyoracle as orc
    from longcalld_amd import jobs
    orc.build()
    ref = read_fasta(os.path.join(TD, "chr11_2M.fa"))
    reads = read_bam(os.path.join(TD, "HG002_chr11_hifi_test.bam"))
    reads.sort(key=lambda r: r["pos"])
    R = len(reads)
    all_win = []
    for r in reads:
        r["digars"], ev = digars_of(r)
        r["win"] = merge(noisy_windows(ev))
        all_win += r["win"]
    chunk_beg, chunk_end = min(r["pos"] for r in reads), max(r["end"] for r in reads)
    # ---- regions: merged windows carried by >= max(2, 20 %) of the spanning reads, + flanks ----
    regs = []
    for b, e in merge(all_win, 5):
        span = [r for r in reads if r["pos"] <= b and r["end"] >= e]
        sup = sum(any(wb <= e and we >= b for wb, we in r["win"]) for r in span)
        if len(span) >= 5 and sup >= max(2, 0.2 * len(span)):
            regs.append([b - FLANK, e + FLANK])
    regs = [(b, e) for b, e in merge(regs) if e - b + 1 <= 50000 and b > chunk_beg and e < chunk_end]
    in_reg = np.zeros(len(ref) + 2, bool)
    for b, e in regs:
        in_reg[b:e + 1] = True
    # ---- candidate variants outside the regions: SNPs and short indels piled up from the digars ----
    depth = np.zeros(len(ref) + 2, np.int32)
    for r in reads:
        depth[r["pos"]:r["end"] + 1] += 1
    pile = {}
    for ri, r in enumerate(reads):
        q = r["qual"]
        for pos, op, ln, qi in r["digars"]:
            if op not in (CX, CI, CD) or in_reg[pos]:
                continue
            if op == CX:
                if q[qi] < MIN_BQ:
                    continue
                base = int(NT16[(r["bseq"][qi >> 1] >> ((~qi & 1) << 2)) & 0xf])
                key = (int(pos), CX, 1, base)
            else:
                key = (int(pos), int(op), int(ln), 0)
            pile.setdefault(key, []).append(ri)
    cand = []
    for key, rl in sorted(pile.items()):
        dp = int(depth[key[0]])
        af = len(rl) / max(dp, 1)
        if dp < 5 or len(rl) < 2 or af < 0.2:
            continue
        if key[1] == CX:
            cate = jobs.CLEAN_HET_SNP if af <= 0.8 else jobs.CLEAN_HOM_VAR
        else:
            cate = jobs.CLEAN_HET_INDEL if af <= 0.8 else jobs.CLEAN_HOM_VAR
        cand.append((key, set(rl), cate))
    V = len(cand)
    var_pos = np.array([c[0][0] for c in cand], np.int64)
    var_type = np.array([c[0][1] for c in cand], np.int32)
    var_cate = np.array([c[2] for c in cand], np.int32)
    is_hp = np.zeros(V, np.int32)
    for v, (key, _, _) in enumerate(cand):
        if key[1] != CX:  # indel inside a run of >= 4 equal reference bases
            p = key[0]
            is_hp[v] = int(len(set(ref[p:p + 4].tolist())) == 1)
    alle_off = np.arange(V + 1, dtype=np.int32) * 2
    alle_covs = np.zeros(2 * V, np.int32)
    start_var, end_var, alleles, allele_off = [], [], [], [0]
    for ri, r in enumerate(reads):
        lo = int(np.searchsorted(var_pos, r["pos"], "left")); hi = int(np.searchsorted(var_pos, r["end"], "right")) - 1
        if hi < lo:
            start_var.append(-1); end_var.append(-2); allele_off.append(allele_off[-1]); continue
        lowq = {int(p) for p, op, ln, qi in r["digars"] if op == CX and r["qual"][qi] < MIN_BQ}
        a = []
        for v in range(lo, hi + 1):
            if ri in cand[v][1]:
                a.append(1)
            elif int(var_pos[v]) in lowq:
                a.append(-2)
            else:
                a.append(0)
            if a[-1] >= 0:
                alle_covs[2 * v + a[-1]] += 1
        start_var.append(lo); end_var.append(hi); alleles += a; allele_off.append(allele_off[-1] + len(a))
    start_var = np.array(start_var, np.int32); end_var = np.array(end_var, np.int32)
    total_cov = (alle_covs[0::2] + alle_covs[1::2]).astype(np.int32)
    is_skipped = np.array([r["mapq"] < 5 for r in reads], np.uint8)
    keep = np.array([i for i in range(R) if start_var[i] >= 0 and not is_skipped[i]], np.int32)
    order = orc.cr_sorted_order(start_var[keep], end_var[keep] + 1)
    prob = dict(n_reads=R, n_vars=V, is_ont=0, var_pos=var_pos, var_type=var_type, var_cate=var_cate, is_homopolymer_indel=is_hp,
                total_cov=total_cov, alle_off=alle_off, alle_covs=alle_covs, start_var_idx=start_var, end_var_idx=end_var,
                allele_off=np.array(allele_off, np.int32), alleles=np.array(alleles, np.int32), ordered_read_ids=np.arange(R, dtype=np.int32),
                is_skipped=is_skipped, cr_read=keep[order])
    st = orc.assign_hap_germline(prob, jobs.GERMLINE_CLEAN)
    haps, pss = st["haps"], st["phase_sets"]
    print(f"{R} reads, chunk {chunk_beg}-{chunk_end}, {V} candidate vars, {len(regs)} regions; "
          f"haps 0/1/2 = {[(haps == k).sum() for k in range(3)]}, phase sets {sorted(set(pss.tolist()))}")
    # ---- which bases the path reads: the slice of every read in every region (collect_noisy_read_info) ----
    used = [np.zeros(r["qlen"], bool) for r in reads]
    reg_reads = []
    for b, e in regs:
        ids = [i for i, r in enumerate(reads) if r["pos"] <= e and r["end"] >= b and not is_skipped[i]]
        reg_reads.append(np.array(ids, np.int32))
        for i in ids:
            rb, re, _ = orc.read_region_slice(reads[i]["digars"], reads[i]["qlen"], b, e, FLANK)
            if re >= rb:
                used[i][rb:re + 1] = True
    for i, r in enumerate(reads):
        u = used[i]
        r["qual"] = np.where(u, r["qual"], 0).astype(np.uint8)
        u2 = np.zeros(2 * len(r["bseq"]), bool); u2[:r["qlen"]] = u
        r["bseq"] = (np.where(u2[0::2], r["bseq"] & 0xf0, 0) | np.where(u2[1::2], r["bseq"] & 0x0f, 0)).astype(np.uint8)
    lo, hi = chunk_beg - 1, chunk_end + 1
    out = dict(ref_beg=np.int64(lo), ref=ref[lo - 1:hi].astype(np.uint8),   # ref[k] = base at 1-based position ref_beg + k
               regions=np.array(regs, np.int64), reg_read_off=np.cumsum([0] + [len(x) for x in reg_reads]).astype(np.int32),
               reg_reads=np.concatenate(reg_reads).astype(np.int32),
               read_pos=np.array([r["pos"] for r in reads], np.int64), read_qlen=np.array([r["qlen"] for r in reads], np.int32),
               digar_off=np.cumsum([0] + [len(r["digars"]) for r in reads]).astype(np.int64),
               digars=np.concatenate([r["digars"] for r in reads]).astype(np.int64),
               bseq_off=np.cumsum([0] + [len(r["bseq"]) for r in reads]).astype(np.int64), bseq=np.concatenate([r["bseq"] for r in reads]),
               qual_off=np.cumsum([0] + [r["qlen"] for r in reads]).astype(np.int64), qual=np.concatenate([r["qual"] for r in reads]),
               exp_haps=haps.astype(np.int32), exp_phase_sets=pss.astype(np.int64))
    for k, v in prob.items():
        out["hap_" + k] = np.asarray(v)
    for k in ("n_clean_agree_snps", "n_clean_conflict_snps", "var_phase_set", "hap_to_cons_alle", "hap_to_alle_profile"):
        out["exp_" + k] = st[k]
    # ---- expected region results (oracle) as one digest per region + the shapes a reader can eyeball ----
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import testdata_common as tc
    chunk = tc.Chunk(out)
    digs, ncons, mlen, nvar = [], [], [], 0
    for k in range(len(regs)):
        reg = chunk.region_dict(orc, k, haps, pss)
        res = orc.collect_noisy_reg_aln_strs(reg)
        digs.append(tc.result_digest(res)); ncons.append(res["n_cons"])
        mlen.append(res["aln_strs"][0][0]["aln_len"] if res["n_cons"] else 0)
        nvar += orc.make_vars_from_msa_cons_aln(res, regs[k][0], out["ref"], int(out["ref_beg"]))["n_vars"]
    out["exp_n_noisy_vars"] = np.int64(nvar)
    out["exp_region_digest"] = np.array(digs, np.uint64); out["exp_n_cons"] = np.array(ncons, np.int32); out["exp_ref_cons_len"] = np.array(mlen, np.int32)
    np.savez_compressed(OUT, **out)
    print("regions (len, reads, n_cons):", [(int(e - b + 1), len(x), n) for (b, e), x, n in zip(regs, reg_reads, ncons)])
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
