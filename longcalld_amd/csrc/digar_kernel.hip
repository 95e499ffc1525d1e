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
        const int radv = (op == 7 || op == 8 || op == 2 || op == 3 || op == 0) ? len : 0;
        const int qadv = (op == 7 || op == 8 || op == 1 || op == 4) ? len : 0;
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

void lcd_launch_digar(const DigarJob *jobs, DigarOut *outs, DigarOpt opt, int n_jobs, hipStream_t stream) {
    if (n_jobs > 0) hipLaunchKernelGGL(lcd_digar_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, outs, opt, n_jobs);
}
