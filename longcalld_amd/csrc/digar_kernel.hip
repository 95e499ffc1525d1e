// digar_kernel.hip -- SURVEY 8(f) row f2, first part, on gfx950: EQX CIGARs -> digar lists + each read's noisy windows
// (collect_digar_from_eqx_cigar, src/bam_utils.c:701-842; push_xid_size_queue_win :161-200).
// One wavefront per read.  The CIGAR operations are taken 64 at a time: reference / query advances and the number of digars each operation
// expands to ('X' runs: one per base) are wave prefix sums, so every lane knows where its operation's digars go and writes them itself.
// The sliding-window detector is a queue automaton over the read's events (mismatches and gaps above the base-quality floor, in order): lane 0
// walks the digars it needs (a few hundred per HiFi read) and keeps the queue in HBM scratch.  Interval ordering (cr_index), the skip rule
// and the overlap with the chunk region are a few compares per read on the host side of lcd_digar_batch.
// Byte streaming: reads 4 B per CIGAR operation + the qualities under X / I / D, writes 24 B per digar.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
__device__ __forceinline__ int wave_incl_scan(int v, const int lane) {
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}
__device__ __forceinline__ long long wave_incl_scan64(long long v, const int lane) {
    for (int d = 1; d < 64; d <<= 1) { const long long o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}
}

__global__ void __launch_bounds__(64) lcd_digar_kernel(const DigarJob *jobs, DigarOut *outs, DigarOpt opt, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int lane = threadIdx.x;
    const DigarJob jb = jobs[jid];
    const unsigned *cig = (const unsigned *)jb.cigar_off;
    const uint8_t *qual = (const uint8_t *)jb.qual_off;
    DigarRec *dg = (DigarRec *)jb.digar_off;
    long long pos = jb.pos0 + 1; // 1-based reference position of the next operation
    int qi = 0, nd = 0, n_cand = 0, bad = 0;
    long long rlen = 0;
    for (int c0 = 0; c0 < jb.n_cigar; c0 += 64) {
        const int i = c0 + lane;
        int op = -1, len = 0;
        if (i < jb.n_cigar) { const unsigned c = cig[i]; op = (int)(c & 0xf); len = (int)(c >> 4); }
        // op 9 (LCD_OP_SKIP) never comes from a BAM: the converters in front of this kernel use it for bases the reference steps over without a
        // digar (collect_digar_from_ref_seq outside [ref_beg, ref_end], src/bam_utils.c:1206-1212)
        const int radv = (op == 7 || op == 8 || op == 2 || op == 3 || op == 0 || op == 9) ? len : 0;
        const int qadv = (op == 7 || op == 8 || op == 1 || op == 4 || op == 9) ? len : 0;
        const int ndig = op == 8 ? len : (op == 7 || op == 2 || op == 1 || op == 4 || op == 5) ? 1 : 0;
        bad |= op == 0;
        const long long rinc = wave_incl_scan64(radv, lane);
        const int qinc = wave_incl_scan(qadv, lane), dinc = wave_incl_scan(ndig, lane);
        const long long p = pos + rinc - radv; const int q = qi + qinc - qadv; int w = nd + dinc - ndig;
        if (op == 8) {
            for (int j = 0; j < len; ++j, ++w) if (w < jb.digar_cap) { DigarRec r; r.pos = p + j; r.type = 8; r.len = 1; r.qi = q + j; r.is_low_qual = qual[q + j] < opt.min_bq; dg[w] = r; }
            n_cand += len;
        } else if (ndig && w < jb.digar_cap) {
            DigarRec r; r.pos = p; r.type = op; r.len = len; r.qi = q; r.is_low_qual = 0;
            if (op == 2) { const int qr = q < jb.qlen ? q : jb.qlen - 1; r.is_low_qual = !((q == 0 || qual[q - 1] >= opt.min_bq) && qual[qr] >= opt.min_bq); ++n_cand; }
            else if (op == 1) { int low = 1; for (int k = 0; k < len; ++k) if (qual[q + k] >= opt.min_bq) { low = 0; break; } r.is_low_qual = low; ++n_cand; }
            else if (op == 4 || op == 5) { if ((i == 0 && jb.left_pal) || (i != 0 && jb.right_pal)) r.type = 5; }
            dg[w] = r;
        }
        pos += __shfl(rinc, 63); qi += __shfl(qinc, 63); nd += __shfl(dinc, 63);
        rlen += __shfl(rinc, 63);
    }
    for (int d = 32; d >= 1; d >>= 1) { n_cand += __shfl_xor(n_cand, d); bad |= __shfl_xor(bad, d); }
    __syncthreads();
    // ---- the window automaton (lane 0): events in order, queue (pos, len, count) in scratch ----
    int n_iv = 0;
    if (lane == 0) {
        long long *qpos = (long long *)jb.ev_off; int *qlen_ = (int *)(qpos + jb.ev_cap), *qcnt = qlen_ + jb.ev_cap;
        IvRec *iv = (IvRec *)jb.iv_off;
        int front = 0, rear = -1, count = 0, q_s = -1, q_e = -1;
        long long cur_s = -1, cur_e = -1;
        const int ndc = nd < jb.digar_cap ? nd : jb.digar_cap;
        auto add_iv = [&](long long st, long long en, int label) { if (st < 0) st = 0; if (st > en) return; /* cr_add, src/cgranges.c:145-149 */ if (n_iv < jb.iv_cap) { IvRec r; r.st = st; r.en = en; r.label = label; r.pad = 0; iv[n_iv] = r; } ++n_iv; };
        for (int k = 0; k < ndc; ++k) {
            const DigarRec r = dg[k];
            if (r.type == 4 || r.type == 5) { // clipping: a long one marks the flank next to it (src/bam_utils.c:772-787)
                const unsigned c = k == 0 ? cig[0] : cig[jb.n_cigar - 1];
                const bool first = k == 0 && ((c & 0xf) == 4 || (c & 0xf) == 5);
                if (jb.clip_rule == 1) { // collect_digar_from_cs_tag's variant (src/bam_utils.c:884-888, :969-972): no outer position test, the palindrome first
                    if (r.len > opt.end_clip_reg && !(first ? jb.left_pal : jb.right_pal)) {
                        if (first) { if (r.pos > 10) add_iv(r.pos - 1, r.pos + opt.end_clip_flank, 0); }
                        else if (r.pos < opt.whole_ref_len - 10) add_iv(r.pos - 1 - opt.end_clip_flank, r.pos, 0);
                        ++n_cand;
                    }
                    continue;
                }
                if ((first && r.pos > 10) || (!first && r.pos < opt.whole_ref_len - 10)) {
                    if (r.len > opt.end_clip_reg) {
                        if (first && !jb.left_pal) { if (r.pos > 1) add_iv(r.pos - 1, r.pos + opt.end_clip_flank, 0); ++n_cand; }
                        else if (!first && !jb.right_pal) { if (r.pos < opt.whole_ref_len) add_iv(r.pos - 1 - opt.end_clip_flank, r.pos, 0); ++n_cand; }
                    }
                }
                continue;
            }
            if (r.is_low_qual || r.type == 7) continue;
            const int len = r.type == 8 ? 1 : r.type == 2 ? r.len : 0, cnt = r.type == 8 ? 1 : r.len;
            if (rear + 1 >= jb.ev_cap) { bad |= 2; break; }
            ++rear; qpos[rear] = r.pos; qlen_[rear] = len; qcnt[rear] = cnt; count += cnt;
            while (qpos[front] + qlen_[front] - 1 <= r.pos - opt.win) { count -= qcnt[front]; ++front; }
            if (count > opt.max_xgaps) {
                const long long ns = qpos[front], ne = r.pos + len;
                if (cur_s == -1) { cur_s = ns; cur_e = ne; q_s = front; q_e = rear; }
                else if (ns <= cur_e) { cur_e = ne; q_e = rear; }
                else {
                    int vs = 0; for (int t = q_s; t <= q_e; ++t) vs += qcnt[t];
                    if (vs < (int)(cur_e - cur_s + 1)) vs = (int)(cur_e - cur_s + 1);
                    add_iv(cur_s - 1, cur_e, vs);
                    cur_s = ns; cur_e = ne; q_s = front; q_e = rear;
                }
            }
        }
        if (cur_s != -1) {
            int vs = 0; for (int t = q_s; t <= q_e; ++t) vs += qcnt[t];
            if (vs < (int)(cur_e - cur_s + 1)) vs = (int)(cur_e - cur_s + 1);
            add_iv(cur_s - 1, cur_e, vs);
        }
        DigarOut o; o.status = (bad & 1) ? -2 : (nd > jb.digar_cap || n_iv > jb.iv_cap || (bad & 2)) ? -3 : 0; o.n_digar = nd; o.n_iv = n_iv; o.n_cand = n_cand; o.rlen = (int)rlen;
        outs[jid] = o;
    }
}

// ---- collect_digar_from_ref_seq's base comparison (src/bam_utils.c:1201-1235) as a CIGAR rewrite: M / = / X operations -> '=' runs, one 'X' per
// differing base and LCD_OP_SKIP for bases outside the loaded reference window, everything else passed through; the result feeds lcd_digar_kernel.
// One wavefront per read, 64 bases per step: the three base classes are ballots, every mismatching lane finds the run in front of it with popcounts
// of the ballots and writes its own operations ([skip] [=] X; the reference flushes a pending '=' run at pos - eq_len, i.e. AFTER the stepped-over
// bases, hence the order).  Two passes of the same code: COUNT sizes the output (operations, digars, window events), EMIT writes it.
// Bytes per base: 1 (reference char) + 1/2 (BAM 4-bit base), per pass.
namespace {
__device__ __forceinline__ int nt4_of_char(const unsigned char c) { // nst_nt4_table, src/seq.c:14-31
    if (c < 4) return c;
    if (c == '-') return 5;
    const unsigned char u = c | 32;
    return u == 'a' ? 0 : u == 'c' ? 1 : u == 'g' ? 2 : u == 't' ? 3 : 4;
}
__device__ __forceinline__ int nt4_of_bam4(const int b) { return b == 1 ? 0 : b == 2 ? 1 : b == 4 ? 2 : b == 8 ? 3 : 4; } // seq_nt16_int (htslib)
}
template <bool EMIT>
__global__ void __launch_bounds__(64) lcd_refcmp_kernel(const RefCmpJob *jobs, RefCmpOut *outs, const char *ref, long long ref_beg, long long ref_end, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int lane = threadIdx.x;
    const RefCmpJob jb = jobs[jid];
    const unsigned *cig = (const unsigned *)jb.cigar_off;
    const uint8_t *seq = (const uint8_t *)jb.seq_off;
    unsigned *out = (unsigned *)jb.out_off;
    long long pos = jb.pos0 + 1;
    int qi = 0, w = 0, nd = 0, nev = 0;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int c0 = 0; c0 < jb.n_cigar; c0 += 64) {
        const unsigned cw = c0 + lane < jb.n_cigar ? cig[c0 + lane] : 0u;
        const int nc = jb.n_cigar - c0 < 64 ? jb.n_cigar - c0 : 64;
        for (int ci = 0; ci < nc; ++ci) {
            const unsigned c = __shfl(cw, ci);
            const int op = (int)(c & 0xf), len = (int)(c >> 4);
            if (op == 0 || op == 7 || op == 8) {
                int e_carry = 0, s_carry = 0;
                for (int b0 = 0; b0 < len; b0 += 64) {
                    const int j = b0 + lane;
                    int cls = 3;
                    if (j < len) {
                        const long long p = pos + j; const int q = qi + j;
                        if (p < ref_beg || p > ref_end) cls = 2;
                        else cls = nt4_of_char((unsigned char)ref[p - ref_beg]) != nt4_of_bam4((seq[q >> 1] >> ((~q & 1) << 2)) & 0xf);
                    }
                    const unsigned long long mm = __ballot(cls == 1), eqm = __ballot(cls == 0), skm = __ballot(cls == 2);
                    int e = 0, s = 0, nops = 0;
                    if (cls == 1) {
                        const unsigned long long prev = mm & below;
                        const int lo = prev ? 64 - __clzll(prev) : 0;
                        const unsigned long long between = below & ~((1ull << lo) - 1);
                        e = __popcll(eqm & between) + (prev ? 0 : e_carry); s = __popcll(skm & between) + (prev ? 0 : s_carry);
                        nops = 1 + (e > 0) + (s > 0);
                    }
                    const int incl = wave_incl_scan(nops, lane);
                    if (EMIT && cls == 1) {
                        int t = w + incl - nops;
                        if (s > 0) out[t++] = ((unsigned)s << 4) | 9u;
                        if (e > 0) out[t++] = ((unsigned)e << 4) | 7u;
                        out[t] = (1u << 4) | 8u;
                    }
                    w += __shfl(incl, 63);
                    const int nmm = __popcll(mm);
                    nd += nmm + __popcll(__ballot(cls == 1 && e > 0)); nev += nmm;
                    if (mm) {
                        const int hi = 64 - __clzll(mm);
                        const unsigned long long tail = hi == 64 ? 0ull : ~((1ull << hi) - 1);
                        e_carry = __popcll(eqm & tail); s_carry = __popcll(skm & tail);
                    } else { e_carry += __popcll(eqm); s_carry += __popcll(skm); }
                }
                if (EMIT && lane == 0) { int t = w; if (s_carry > 0) out[t++] = ((unsigned)s_carry << 4) | 9u; if (e_carry > 0) out[t] = ((unsigned)e_carry << 4) | 7u; }
                w += (s_carry > 0) + (e_carry > 0); nd += e_carry > 0;
                pos += len; qi += len;
            } else {
                if (EMIT && lane == 0) out[w] = c;
                ++w;
                if (op == 2 || op == 3) pos += len;
                if (op == 1 || op == 4) qi += len;
                if (op == 1 || op == 2 || op == 4 || op == 5) ++nd;
                if (op == 1 || op == 2) ++nev;
            }
        }
    }
    if (!EMIT && lane == 0) { RefCmpOut o; o.n_ops = w; o.nd = nd; o.nev = nev; o.pad = 0; outs[jid] = o; }
}
void lcd_launch_refcmp(bool emit, const RefCmpJob *jobs, RefCmpOut *outs, const char *ref, long long ref_beg, long long ref_end, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return;
    if (emit) hipLaunchKernelGGL(lcd_refcmp_kernel<true>, dim3(n_jobs), dim3(64), 0, stream, jobs, outs, ref, ref_beg, ref_end, n_jobs);
    else hipLaunchKernelGGL(lcd_refcmp_kernel<false>, dim3(n_jobs), dim3(64), 0, stream, jobs, outs, ref, ref_beg, ref_end, n_jobs);
}

// pre_process_noisy_regs, the read-support part (src/collect_var.c:584-602): per merged region, the reads spanning it and, among them, those
// with a noisy window of their own inside it.  One wavefront per region, lanes over the chunk's reads (cr_overlap semantics: half-open
// intervals, a.st < b.en && b.st < a.en), wave reduction of the two counters.
__global__ void __launch_bounds__(64) lcd_region_support_kernel(const IvRec *regs, int n_regs, const long long *read_beg, const long long *read_end,
                                                                const unsigned long long *iv_off, const IvRec *ivs, int n_reads, int *total, int *noisy) {
    const int ri = blockIdx.x;
    if (ri >= n_regs) return;
    const int lane = threadIdx.x;
    const long long rs = regs[ri].st, re = regs[ri].en;
    int tot = 0, nz = 0;
    for (int r = lane; r < n_reads; r += 64) {
        const long long qb = read_beg[r] - 1, qe = read_end[r];
        if (!(rs < qe && qb < re)) continue;
        ++tot;
        int hit = 0;
        for (unsigned long long k = iv_off[r]; k < iv_off[r + 1] && !hit; ++k) hit = ivs[k].st < re && rs < ivs[k].en;
        nz += hit;
    }
    for (int d = 32; d >= 1; d >>= 1) { tot += __shfl_xor(tot, d); nz += __shfl_xor(nz, d); }
    if (lane == 0) { total[ri] = tot; noisy[ri] = nz; }
}
void lcd_launch_region_support(const IvRec *regs, int n_regs, const long long *read_beg, const long long *read_end, const unsigned long long *iv_off,
                               const IvRec *ivs, int n_reads, int *total, int *noisy, hipStream_t stream) {
    if (n_regs > 0) hipLaunchKernelGGL(lcd_region_support_kernel, dim3(n_regs), dim3(64), 0, stream, regs, n_regs, read_beg, read_end, iv_off, ivs, n_reads, total, noisy);
}

// collect_noisy_read_info's digar walk (src/align.c:1392-1441), one wavefront per (region, read) pair: which query interval of the read lies over the region,
// and does the read cover the region's ends (a deletion longer than the flank at an end counts as a gap).  The reference walks the list in order and lets a
// later digar overwrite what an earlier one set -- an insertion AT the region's first position is followed by the '=' run that starts there, so the run wins --
// and stops at the first digar that begins behind the region.  Here 64 digars are looked at per step: the stop is the first lane whose digar begins behind the
// region, "later overwrites earlier" is the highest matching lane below it (and of the steps so far), the deletion flags are set-only in the reference and so an OR.
__global__ void __launch_bounds__(64) lcd_slice_kernel(const SliceJob *jobs, SliceOut *outs, const DigarRec *digars, const int flank, const int n_jobs) {
    const int ji = blockIdx.x, lane = threadIdx.x;
    if (ji >= n_jobs) return;
    const SliceJob j = jobs[ji];
    const DigarRec *d = digars + j.digar_off;
    const int nd = j.n_digar;
    int rb = 0, re = j.qlen - 1;
    if (nd > 0) { if (d[0].type == 5) rb = d[0].len; if (d[nd - 1].type == 5) re = d[nd - 1].qi - 1; }
    bool hit_b = false, hit_e = false; int beg_del = 0, end_del = 0;
    for (int base = 0; base < nd; base += 64) {
        const int k = base + lane;
        bool cand = false, stop = false; long long db = 0, de = 0; int op = 0, len = 0, qi = 0;
        if (k < nd) {
            const DigarRec r = d[k];
            op = r.type; len = r.len; qi = r.qi; db = r.pos;
            if (op != 4 && op != 5) {
                de = (op == 8 || op == 7 || op == 2) ? db + len - 1 : db;
                stop = db > j.reg_end;
                cand = !stop && de >= j.reg_beg;
            }
        }
        const unsigned long long sm = __ballot(stop);
        const unsigned long long below = sm ? ((1ull << __builtin_ctzll(sm)) - 1ull) : ~0ull; // lanes before the first stop
        const unsigned long long mb = __ballot(cand && db <= j.reg_beg && de >= j.reg_beg) & below;
        const unsigned long long me = __ballot(cand && db <= j.reg_end && de >= j.reg_end) & below;
        if (mb) {
            const int src = 63 - __builtin_clzll(mb);
            const int sop = __shfl(op, src), sqi = __shfl(qi, src); const long long sdb = __shfl(db, src);
            rb = sop == 2 ? sqi : sqi + (int)(j.reg_beg - sdb); hit_b = true;
            beg_del |= (__ballot(op == 2 && len > flank) & mb) != 0;
        }
        if (me) {
            const int src = 63 - __builtin_clzll(me);
            const int sop = __shfl(op, src), sqi = __shfl(qi, src); const long long sdb = __shfl(db, src);
            re = sop == 2 ? sqi - 1 : sqi + (int)(j.reg_end - sdb); hit_e = true;
            end_del |= (__ballot(op == 2 && len > flank) & me) != 0;
        }
        if (sm) break;
    }
    if (lane == 0) {
        int cover = 0; // LONGCALLD_NOISY_{LEFT,RIGHT}_{COVER,GAP} (src/align.c:1442-1456; lcd_types.h LCD_LEFT_COVER ...)
        if (hit_b && hit_e) cover = (beg_del ? LCD_LEFT_GAP : LCD_LEFT_COVER) | (end_del ? LCD_RIGHT_GAP : LCD_RIGHT_COVER);
        else if (hit_b) cover = beg_del ? LCD_LEFT_GAP : LCD_LEFT_COVER;
        else if (hit_e) cover = end_del ? LCD_RIGHT_GAP : LCD_RIGHT_COVER;
        SliceOut o; o.read_beg = rb; o.read_end = re; o.cover = cover; o.pad = 0;
        outs[ji] = o;
    }
}
void lcd_launch_slices(const SliceJob *jobs, SliceOut *outs, const DigarRec *digars, int flank, int n_jobs, hipStream_t stream) {
    if (n_jobs > 0) hipLaunchKernelGGL(lcd_slice_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, outs, digars, flank, n_jobs);
}

// Read slices of noisy regions, 4-bit packed -> 1 B/base codes in the batch's input pool (the per-base loop of collect_noisy_read_info, src/align.c:1445-1448:
// seq_nt16_int[bam_seqi(bseq, j)]).  One workgroup per slice, four bases per lane and step: two or three packed bytes in, one 32-bit store out (the pool's
// slices start 16-byte aligned).  HBM-bound and small: 0.5 B read + 1 B written per base, ~20 MB per configs[1] batch.
__global__ void __launch_bounds__(256) lcd_unpack_kernel(const UnpackJob *jobs, const uint8_t *packed, uint8_t *pool) {
    const UnpackJob j = jobs[blockIdx.x];
    const uint8_t *s = packed + j.src; uint8_t *d = pool + j.dst;
    auto code = [](unsigned c) -> unsigned { return c == 1 ? 0u : c == 2 ? 1u : c == 4 ? 2u : c == 8 ? 3u : 4u; }; // htslib seq_nt16_int: A C G T, everything else 4
    const int n4 = j.len & ~3;
    for (int k = threadIdx.x * 4; k < n4; k += blockDim.x * 4) {
        const int q = j.first + k; // nibble index of the first of four bases
        const unsigned b0 = s[q >> 1], b1 = s[(q >> 1) + 1], b2 = (q & 1) ? s[(q >> 1) + 2] : 0u;
        const unsigned w = (b0 << 16) | (b1 << 8) | b2; // nibbles 0..5, high first
        const int sh = (q & 1) ? 16 : 20;                // first base: nibble 1 or 0 of w's 24 bits
        const unsigned o = code((w >> sh) & 15) | (code((w >> (sh - 4)) & 15) << 8) | (code((w >> (sh - 8)) & 15) << 16) | (code((w >> (sh - 12)) & 15) << 24);
        *(unsigned *)(d + k) = o;
    }
    for (int k = n4 + threadIdx.x; k < j.len; k += blockDim.x) { const int q = j.first + k; d[k] = (uint8_t)code((s[q >> 1] >> ((~q & 1) << 2)) & 15); }
}
void lcd_launch_unpack(const UnpackJob *jobs, int n_jobs, const uint8_t *packed, uint8_t *pool, hipStream_t stream) {
    if (n_jobs > 0) hipLaunchKernelGGL(lcd_unpack_kernel, dim3(n_jobs), dim3(256), 0, stream, jobs, packed, pool);
}

void lcd_launch_digar(const DigarJob *jobs, DigarOut *outs, DigarOpt opt, int n_jobs, hipStream_t stream) {
    if (n_jobs > 0) hipLaunchKernelGGL(lcd_digar_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, outs, opt, n_jobs);
}
