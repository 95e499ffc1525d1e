"""Seeded alignments in the four shapes the reference builds digars from (src/collect_var.c:1072-1079): an EQX CIGAR, a cs:Z tag (short and long
form), an MD:Z tag next to an 'M' CIGAR, and the plain 'M' CIGAR + read bases + reference window.  All derived from ONE alignment, so the four
collect_digar_from_* functions have to agree on it wherever their code does (tests/test_digar_tags.py, tests/test_gpu_digar.py)."""
import numpy as np

BASES = b"ACGT"


def pack4(nt):
    """0..3 (4 = N) -> BAM 4-bit packed bases (bam_get_seq)"""
    code = np.array([1, 2, 4, 8, 15], np.uint8)[np.asarray(nt, np.uint8)]
    if len(code) & 1:
        code = np.concatenate([code, np.zeros(1, np.uint8)])
    return ((code[0::2] << 4) | code[1::2]).astype(np.uint8)


def eqx_ops(rng, noisy=False, clips=True, n_skip=True, n_ev=None):
    ops = []
    if clips and rng.random() < 0.5:
        if rng.random() < 0.3:
            ops.append((5, int(rng.integers(1, 200))))
        else:
            ops.append((4, int(rng.integers(1, 120))))
    for _ in range(n_ev if n_ev is not None else int(rng.integers(20, 300))):
        ops.append((7, int(rng.integers(1, 300 if not noisy else 30))))
        x = rng.random()
        if x < 0.45:
            ops.append((8, int(rng.integers(1, 4))))
        elif x < 0.7:
            ops.append((1, int(rng.integers(1, 60 if rng.random() < 0.1 else 5))))
        elif x < 0.95 or not n_skip:
            ops.append((2, int(rng.integers(1, 80 if rng.random() < 0.1 else 5))))
        else:
            ops.append((3, int(rng.integers(10, 2000))))
    ops.append((7, int(rng.integers(5, 100))))
    if clips and rng.random() < 0.5:
        ops.append((4 if rng.random() < 0.7 else 5, int(rng.integers(1, 150))))
    return ops


def build(rng, ops, pos0, ref_pad=(0, 0), lower=False):
    """ops: [(op, len)] EQX operations.  -> dict with every input shape; the reference window is [pos0 + 1 - ref_pad[0], end + ref_pad[1]] (negative pads cut
    into the read: the 'read exceeds the reference region' case of collect_digar_from_ref_seq)"""
    rlen = sum(l for o, l in ops if o in (7, 8, 2, 3))
    ref = rng.integers(0, 4, rlen).astype(np.uint8)
    read, cs_s, cs_l, md = [], [], [], []
    md_cnt = 0
    rp = 0
    for o, l in ops:
        if o == 7:
            read.append(ref[rp:rp + l]); cs_s.append(b":%d" % l); cs_l.append(b"=" + bytes(BASES[b] for b in ref[rp:rp + l])); md_cnt += l; rp += l
        elif o == 8:
            for k in range(l):
                r = int(ref[rp + k]); q = (r + int(rng.integers(1, 4))) & 3
                read.append(np.array([q], np.uint8))
                e = b"*" + bytes([BASES[r] | 32, BASES[q] | 32]); cs_s.append(e); cs_l.append(e)
                md.append(b"%d" % md_cnt + bytes([BASES[r]])); md_cnt = 0
            rp += l
        elif o == 1:
            ins = rng.integers(0, 4, l).astype(np.uint8); read.append(ins)
            e = b"+" + bytes(BASES[b] | 32 for b in ins); cs_s.append(e); cs_l.append(e)
        elif o == 2:
            e = b"-" + bytes(BASES[b] | 32 for b in ref[rp:rp + l]); cs_s.append(e); cs_l.append(e)
            md.append(b"%d^" % md_cnt + bytes(BASES[b] for b in ref[rp:rp + l])); md_cnt = 0; rp += l
        elif o == 3:
            rp += l     # (a cs tag would carry "~gt<len>ag" here; the cs function does not move pos over it -- only used where the shapes are compared separately)
        elif o == 4:
            read.append(rng.integers(0, 4, l).astype(np.uint8))
    md.append(b"%d" % md_cnt)
    read = np.concatenate(read) if read else np.zeros(0, np.uint8)
    eqx = np.array([(l << 4) | o for o, l in ops], np.uint32)
    m = []
    for o, l in ops:
        o2 = 0 if o in (7, 8) else o
        if m and o2 == 0 and (m[-1] & 0xf) == 0:
            m[-1] += l << 4
        else:
            m.append((l << 4) | o2)
    refc = bytes(BASES[b] | (32 if lower and (i // 50) % 3 == 0 else 0) for i, b in enumerate(ref))
    lp, rpd = ref_pad
    ref_beg = pos0 + 1 - lp; ref_end = pos0 + rlen + rpd
    left = bytes(BASES[b] for b in rng.integers(0, 4, max(lp, 0))); right = bytes(BASES[b] for b in rng.integers(0, 4, max(rpd, 0)))
    core = refc[max(-lp, 0):rlen - max(-rpd, 0)]
    return dict(pos0=pos0, eqx=eqx, mcig=np.array(m, np.uint32), cs=b"".join(cs_s), cs_long=b"".join(cs_l), md=b"".join(md), bseq=pack4(read), qlen=len(read),
                ref_seq=left + core + right, ref_beg=ref_beg, ref_end=ref_end, has_n=any(o == 3 for o, _ in ops))


def quals(rng, qlen):
    return rng.choice([2, 9, 10, 25, 40], qlen, p=[0.03, 0.04, 0.08, 0.35, 0.5]).astype(np.uint8)
