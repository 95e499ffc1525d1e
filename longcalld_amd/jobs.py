"""Synthetic region-job generator in the shape of BASELINE.json's configs.

A *region job* is what collect_noisy_reg_aln_strs sees after collect_noisy_read_info (src/align.c:1377):
the reference bytes of the noisy region (with 10-bp flanks) and, per read, its slice of bases, cover flag,
haplotype tag and phase set.  Generated directly (bypassing BAM) from a seeded diploid truth so the CPU
oracle and the GPU consume identical bytes.
"""
import numpy as np

BOTH, LEFT, RIGHT = 12, 8, 4

HIFI = dict(name="hifi", err=0.001, hp_frac=0.7, depth=30, median_len=500.0, sigma=0.75, min_len=67, max_len=3995,
            partial_frac=0.03, untagged_frac=0.12, no_ps_frac=0.15, sv_frac=0.05)
ONT = dict(name="ont", err=0.05, hp_frac=0.5, depth=30, median_len=500.0, sigma=0.75, min_len=67, max_len=3995,
           partial_frac=0.04, untagged_frac=0.2, no_ps_frac=0.2, sv_frac=0.05)


def _mutate(rng, seq, err, hp_frac):
    n = len(seq)
    k = rng.binomial(n, err) if n else 0
    if k == 0:
        return seq.copy()
    pos = np.sort(rng.choice(n, size=min(k, n), replace=False))
    out, last = [], 0
    for p in pos:
        out.append(seq[last:p])
        x = rng.random()
        if x < hp_frac:  # homopolymer-style indel: drop or duplicate the base
            if rng.random() < 0.5:
                pass
            else:
                out.append(seq[p:p + 1]); out.append(seq[p:p + 1])
        else:
            y = rng.random()
            if y < 1 / 3:
                out.append(np.array([(seq[p] + 1 + rng.integers(0, 3)) % 4], np.uint8))
            elif y < 2 / 3:
                out.append(seq[p:p + 1]); out.append(rng.integers(0, 4, 1).astype(np.uint8))
        last = p + 1
    out.append(seq[last:])
    return np.concatenate(out).astype(np.uint8)


def _make_hap(rng, ref, flank, want_var, sv_frac):
    """one haplotype of the region: ref with 0-3 variants inside the flanks"""
    if not want_var:
        return ref.copy()
    L = len(ref)
    pieces, last = [], 0
    nvar = 1 + rng.integers(0, 3)
    pos = np.sort(rng.integers(flank + 2, max(flank + 3, L - flank - 2), nvar))
    for p in pos:
        if p <= last:
            continue
        pieces.append(ref[last:p])
        x = rng.random()
        if x < 0.35:
            pieces.append(np.array([(ref[p] + 1 + rng.integers(0, 3)) % 4], np.uint8)); last = p + 1   # SNP
        elif x < 0.65:
            ins = int(rng.integers(6, 30)) if rng.random() > sv_frac else int(rng.integers(30, 120))
            pieces.append(rng.integers(0, 4, ins).astype(np.uint8)); last = p                                # insertion
        else:
            dl = int(rng.integers(6, 30)) if rng.random() > sv_frac else int(rng.integers(30, 120))
            last = min(p + dl, L - flank - 1)                                                               # deletion
    pieces.append(ref[last:])
    return np.concatenate(pieces).astype(np.uint8)


def make_region(rng, shape=HIFI, length=None, n_reads=None, flank=10):
    L = int(length) if length else int(np.clip(np.exp(rng.normal(np.log(shape["median_len"]), shape["sigma"])), shape["min_len"], shape["max_len"]))
    ref = rng.integers(0, 4, L).astype(np.uint8)
    if rng.random() < 0.3 and L > 60:  # tandem-repeat / homopolymer block
        unit = rng.integers(0, 4, int(rng.integers(1, 5))).astype(np.uint8)
        rl = int(rng.integers(8, min(40, L - 2 * flank - 4)))
        s = int(rng.integers(flank + 1, L - flank - rl))
        ref[s:s + rl] = np.resize(unit, rl)
    hap_seq = [_make_hap(rng, ref, flank, True, shape["sv_frac"]), _make_hap(rng, ref, flank, rng.random() < 0.5, shape["sv_frac"])]
    n = int(n_reads) if n_reads else int(np.clip(rng.poisson(shape["depth"]), 5, 60))
    no_ps = rng.random() < shape["no_ps_frac"]
    ps = int(rng.integers(1000, 10_000_000))
    seqs, covers, haps, pss, quals = [], [], [], [], []
    for i in range(n):
        h = int(rng.integers(0, 2))
        s = _mutate(rng, hap_seq[h], shape["err"], shape["hp_frac"])
        cover = BOTH
        if rng.random() < shape["partial_frac"] and len(s) > 40:
            cut = int(rng.integers(20, len(s) - 10))
            if rng.random() < 0.5:
                s, cover = s[:cut], LEFT
            else:
                s, cover = s[cut:], RIGHT
        tagged = (not no_ps) and rng.random() > shape["untagged_frac"]
        seqs.append(s); covers.append(cover)
        haps.append(h + 1 if tagged else 0)
        pss.append(ps if tagged else -1)
        quals.append(np.full(len(s), 30, np.uint8))
    return dict(reg_len=L, read_ids=np.arange(n, dtype=np.int32) + 100, seqs=seqs, quals=quals, covers=np.array(covers, np.int32),
                haps=np.array(haps, np.int32), phase_sets=np.array(pss, np.int64), ref=ref)


# BASELINE configs[4]: 60x ONT with injected 1-10 kb INS/DEL at 1 per 100 kb => per 10 Mb ~100 SV regions next to the
# ordinary noisy regions.  An insertion's region is a short reference window whose carrier reads are sv_len longer; a deletion's region spans the
# deleted bases (1-12 kb of reference; >= 10 kb takes the read-sampling path, src/align.c:719-728,1773) and its carrier reads are short.
ONT60 = dict(ONT, name="ont60", depth=60)
SV = dict(name="sv", err=0.05, hp_frac=0.5, depth=60, sv_min=1000, sv_max=10000, ctx_min=150, ctx_max=2000, read_len=20000.0,
          untagged_frac=0.2, no_ps_frac=0.1, sv_per_10mb=100, base=ONT60)


def make_sv_region(rng, shape=SV, kind=None, sv_len=None, ctx=None, n_reads=None, flank=10, phased=None):
    """one SV region: kind 'ins' | 'del', sv_len bases inserted into / deleted from haplotype 1 of a reference window"""
    kind = kind or ("ins" if rng.random() < 0.5 else "del")
    sv_len = int(sv_len) if sv_len else int(np.exp(rng.uniform(np.log(shape["sv_min"]), np.log(shape["sv_max"]))))
    ctx = int(ctx) if ctx else int(rng.integers(shape["ctx_min"], shape["ctx_max"]))
    L = ctx if kind == "ins" else ctx + sv_len
    ref = rng.integers(0, 4, L).astype(np.uint8)
    p = int(rng.integers(flank + 20, ctx - flank - 20)) if ctx > 2 * flank + 60 else ctx // 2
    if kind == "ins":
        if rng.random() < 0.3:   # tandem duplication of the bases before the breakpoint (as many SV insertions are)
            unit = ref[max(0, p - min(p, 400)):p]
            ins = np.resize(unit, sv_len).astype(np.uint8) if len(unit) else rng.integers(0, 4, sv_len).astype(np.uint8)
        else:
            ins = rng.integers(0, 4, sv_len).astype(np.uint8)
        hap1 = np.concatenate([ref[:p], ins, ref[p:]])
    else:
        hap1 = np.concatenate([ref[:p], ref[p + sv_len:]])
    hap2 = ref.copy()
    for hp in (0, 1):   # a few small heterozygous variants next to the SV
        if rng.random() < 0.5:
            h = hap1 if hp == 0 else hap2
            q = int(rng.integers(flank + 2, max(flank + 3, min(len(h), ctx) - flank - 2)))
            h[q] = (h[q] + 1 + rng.integers(0, 3)) % 4
    hap_seq = [hap1.astype(np.uint8), hap2.astype(np.uint8)]
    n = int(n_reads) if n_reads else int(np.clip(rng.poisson(shape["depth"]), 20, 120))
    no_ps = (rng.random() < shape["no_ps_frac"]) if phased is None else (not phased)
    ps = int(rng.integers(1000, 10_000_000))
    seqs, covers, haps, pss, quals = [], [], [], [], []
    for i in range(n):
        h = int(rng.integers(0, 2))
        s = _mutate(rng, hap_seq[h], shape["err"], shape["hp_frac"])
        cover = BOTH
        # a read of ~20 kb ends inside a region of length R with probability ~R / read_len
        if rng.random() < min(0.5, 0.04 + len(hap_seq[h]) / shape["read_len"]) and len(s) > 60:
            cut = int(rng.integers(30, len(s) - 20))
            if rng.random() < 0.5:
                s, cover = s[:cut], LEFT
            else:
                s, cover = s[cut:], RIGHT
        tagged = (not no_ps) and rng.random() > shape["untagged_frac"]
        seqs.append(s); covers.append(cover)
        haps.append(h + 1 if tagged else 0)
        pss.append(ps if tagged else -1)
        q0 = int(rng.integers(12, 28))   # reads differ in quality: the sampling path orders by error rate (src/align.c:957-968)
        quals.append(np.clip(q0 + rng.integers(-3, 4, len(s)), 2, 60).astype(np.uint8))
    return dict(reg_len=L, read_ids=np.arange(n, dtype=np.int32) + 100, seqs=seqs, quals=quals, covers=np.array(covers, np.int32),
                haps=np.array(haps, np.int32), phase_sets=np.array(pss, np.int64), ref=ref, sv=(kind, sv_len))


def make_regions(seed, n_regions, shape=HIFI, poisson_sv=False):
    """n_regions region jobs of one shape; the SV shape mixes sv_per_10mb SV regions per 1 250 into ordinary 60x noisy-read regions (configs[4]);
    poisson_sv: the number of SV regions is Poisson-distributed (a 500 kb chunk holds 5 on average, some chunks none, some a dozen)"""
    rng = np.random.default_rng(seed)
    if shape.get("name") != "sv":
        return [make_region(rng, shape) for _ in range(n_regions)]
    n_sv = max(1, int(round(n_regions * shape["sv_per_10mb"] / 1250.0)))
    if poisson_sv:
        n_sv = int(min(n_regions, rng.poisson(n_regions * shape["sv_per_10mb"] / 1250.0)))
    is_sv = np.zeros(n_regions, bool)
    is_sv[rng.choice(n_regions, n_sv, replace=False)] = True
    return [make_sv_region(rng, shape) if f else make_region(rng, shape["base"]) for f in is_sv]


SHAPES = {"hifi": HIFI, "ont": ONT, "ont60": ONT60, "sv": SV}


def regions_for_ref_mb(ref_mb):
    """~1 250 regions per 10 Mb of 30x HiFi"""
    return int(round(125 * ref_mb))


# ---------------- K5: synthetic haplotype-assignment problems (src/assign_hap.c:473) ----------------
CLEAN_HET_SNP, CLEAN_HET_INDEL, CLEAN_HOM_VAR, NOISY_HET, NOISY_HOM, NON_VAR = 0x004, 0x008, 0x080, 0x100, 0x200, 0x800
GERMLINE_CLEAN = CLEAN_HET_SNP | CLEAN_HET_INDEL | CLEAN_HOM_VAR
GERMLINE_ALL = GERMLINE_CLEAN | NOISY_HET | NOISY_HOM


def make_hap_problem(rng, n_vars=300, n_reads=400, span=(8, 30), err=0.02, is_ont=0, gap_every=0):
    """one chunk's read x variant profile in the flattened form of lcd_hap_problem_t; returns dict of numpy arrays + truth"""
    pos = np.sort(rng.choice(np.arange(1000, 1000 + n_vars * 400), n_vars, replace=False)).astype(np.int64)
    cate = rng.choice([CLEAN_HET_SNP, CLEAN_HET_INDEL, CLEAN_HOM_VAR, NOISY_HET, NOISY_HOM, NON_VAR], n_vars, p=[0.55, 0.1, 0.1, 0.12, 0.05, 0.08]).astype(np.int32)
    vtype = np.where(np.isin(cate, [CLEAN_HET_SNP]), 8, rng.choice([8, 1, 2], n_vars)).astype(np.int32)
    vtype[cate == CLEAN_HET_INDEL] = rng.choice([1, 2], int((cate == CLEAN_HET_INDEL).sum()))
    is_hp = ((vtype != 8) & (rng.random(n_vars) < 0.15)).astype(np.int32)
    hom = np.isin(cate, [CLEAN_HOM_VAR, NOISY_HOM])
    h1 = rng.integers(0, 2, n_vars)
    truth = np.stack([np.where(hom, 1, h1), np.where(hom, 1, 1 - h1)])  # allele of hap 1 / hap 2
    n_alle = np.where(rng.random(n_vars) < 0.05, 3, 2).astype(np.int32)
    alle_off = np.concatenate([[0], np.cumsum(n_alle)]).astype(np.int32)
    starts = np.sort(rng.integers(0, max(1, n_vars - span[0]), n_reads))
    read_hap = rng.integers(1, 3, n_reads)
    start_var, end_var, alleles, allele_off = [], [], [], [0]
    for r in range(n_reads):
        s = int(starts[r]); e = min(n_vars - 1, s + int(rng.integers(span[0], span[1])))
        if gap_every and (s // gap_every) != (e // gap_every):
            e = (s // gap_every + 1) * gap_every - 1          # reads never bridge a block boundary -> several phase sets
        if rng.random() < 0.03:
            start_var.append(-1); end_var.append(-2); allele_off.append(allele_off[-1]); continue
        a = truth[read_hap[r] - 1, s:e + 1].copy()
        flip = rng.random(e - s + 1) < err
        a = np.where(flip, 1 - a, a)
        lowq = rng.random(e - s + 1)
        a = np.where(lowq < 0.02, -1, np.where(lowq < 0.03, -2, a))
        start_var.append(s); end_var.append(e); alleles.extend(a.tolist()); allele_off.append(allele_off[-1] + len(a))
    start_var = np.array(start_var, np.int32); end_var = np.array(end_var, np.int32)
    alleles = np.array(alleles, np.int32); allele_off = np.array(allele_off, np.int32)
    is_skipped = (rng.random(n_reads) < 0.03).astype(np.uint8)
    alle_covs = np.zeros(alle_off[-1], np.int32)
    for r in range(n_reads):
        if start_var[r] < 0 or is_skipped[r]:
            continue
        for k, v in enumerate(range(start_var[r], end_var[r] + 1)):
            al = alleles[allele_off[r] + k]
            if al >= 0:
                alle_covs[alle_off[v] + al] += 1
    total_cov = np.array([alle_covs[alle_off[v]:alle_off[v + 1]].sum() for v in range(n_vars)], np.int32)
    ordered = np.arange(n_reads, dtype=np.int32)
    cr_read = np.array([r for r in ordered if start_var[r] >= 0 and not is_skipped[r]], np.int32)  # cr_add order == sorted by start
    return dict(n_reads=n_reads, n_vars=n_vars, is_ont=is_ont, var_pos=pos, var_type=vtype, var_cate=cate, is_homopolymer_indel=is_hp,
                total_cov=total_cov, alle_off=alle_off, alle_covs=alle_covs, start_var_idx=start_var, end_var_idx=end_var, allele_off=allele_off,
                alleles=alleles, ordered_read_ids=ordered, is_skipped=is_skipped, cr_read=cr_read, read_hap_truth=read_hap)
