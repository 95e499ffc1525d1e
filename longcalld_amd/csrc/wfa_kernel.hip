// wfa_kernel.hip -- K3: wavefront gap-affine-2p end-to-end alignment on gfx950 (CDNA4, wave64).
//
// Replaces what src/align.c:374-460 (wfa_end2end_aln, heuristic none, affine-2p, memory high) asks of
// WFA2-lib.  One 64-lane wavefront owns one alignment; lanes sweep the diagonals of the score-s wavefront
// (five int32 offsets per diagonal: M, I1, I2, D1, D2), the extend step is a per-lane match run, the end
// test is a wave ballot.  All wavefronts are retained in the job's HBM arena (4 B x 5 per diagonal, written
// once -- the algorithmic bytes of SURVEY 8d) so the backtrace is the direct one.
// gap_aln == LEFT (src/align.c:409-453) is done by index reversal, not by copying: P(v) = pat[plen-1-v].
// Semantics are defined by oracle/wfa2p.c; this file must match it bit for bit.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

#define WF_NULL (-(1 << 29))

namespace {

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

struct Wf {
    int *lo, *hi;          // per score
    unsigned int *off;     // per score: index of M[lo] in `data`; components are width apart
    int *data;
    int n;                 // number of scores computed (s+1)
};

__device__ __forceinline__ int wf_get(const Wf &W, int comp, int s, int k) {
    if (s < 0 || s >= W.n) return WF_NULL;
    const int lo = W.lo[s], hi = W.hi[s];
    if (lo > hi || k < lo || k > hi) return WF_NULL;
    return W.data[(size_t)W.off[s] + (size_t)comp * (hi - lo + 1) + (k - lo)];
}
__device__ __forceinline__ bool wf_exists(const Wf &W, int s) { return s >= 0 && s < W.n && W.lo[s] <= W.hi[s]; }

} // namespace

__global__ void __launch_bounds__(64) lcd_wfa_kernel(const WfaJob *jobs, const uint8_t *pool, uint8_t *arena, uint8_t *outpool,
                                                     WfaOut *outs, LcdScoring sc, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int lane = threadIdx.x;
    const WfaJob jb = jobs[jid];
    const uint8_t *pat = pool + jb.p_off, *txt = pool + jb.t_off;
    const int plen = jb.plen, tlen = jb.tlen;
    const bool rev = jb.gap_aln == 1;
    const int x = sc.mismatch, o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2;
#define PAT(v) (rev ? pat[plen - 1 - (v)] : pat[(v)])
#define TXT(h) (rev ? txt[tlen - 1 - (h)] : txt[(h)])
    // arena: lo[s_cap+1] hi[s_cap+1] off[s_cap+1] | ops[plen+tlen+2] | data...
    uint8_t *ws = arena + jb.ws_off;
    const int s_cap = jb.s_cap;
    Wf W;
    W.lo = (int *)ws; W.hi = W.lo + (s_cap + 1); W.off = (unsigned int *)(W.hi + (s_cap + 1));
    uint8_t *ops = (uint8_t *)(W.off + (s_cap + 1));
    const uint64_t hdr = lcd_align_up((uint64_t)3 * (s_cap + 1) * 4 + (uint64_t)(plen + tlen + 2), 16);
    W.data = (int *)(ws + hdr);
    const uint64_t data_cap = (jb.ws_bytes - hdr) / 4;
    uint64_t used = 0;
    WfaOut out; out.status = LCD_OK; out.score = -1; out.n_cigar = 0; out.aln_len = 0; out.offsets = 0;
    const int k_end = tlen - plen;
    int s = 0;
    if (lane == 0) { W.lo[0] = 0; W.hi[0] = 0; W.off[0] = 0; W.data[0] = 0; W.data[1] = W.data[2] = W.data[3] = W.data[4] = WF_NULL; }
    used = 5; W.n = 1;
    __syncthreads();
    unsigned long long n_off = 5;
    for (;;) {
        const int lo = W.lo[s], hi = W.hi[s];
        if (lo <= hi) {
            int *m = W.data + W.off[s];
            int done = 0;
            for (int k0 = lo; k0 <= hi; k0 += 64) {
                const int k = k0 + lane;
                if (k <= hi) {
                    int h = m[k - lo];
                    if (h >= 0) {
                        int v = h - k;
                        while (v < plen && h < tlen && PAT(v) == TXT(h)) { ++v; ++h; }
                        m[k - lo] = h;
                        if (k == k_end && h >= tlen) done = 1;
                    }
                }
            }
            __syncthreads();
            if (__any(done)) break;
        }
        ++s;
        if (s > s_cap) { out.status = LCD_ERR_WF; break; }
        W.n = s + 1;
        int nlo = 1 << 30, nhi = -(1 << 30), any = 0;
        const int src[5] = {s - x, s - o1 - e1, s - o2 - e2, s - e1, s - e2};
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if (wf_exists(W, src[i])) { any = 1; nlo = imin(nlo, W.lo[src[i]]); nhi = imax(nhi, W.hi[src[i]]); }
        if (!any) { if (lane == 0) { W.lo[s] = 1; W.hi[s] = 0; W.off[s] = (unsigned int)used; } __syncthreads(); continue; }
        nlo -= 1; nhi += 1;
        const int width = nhi - nlo + 1;
        if (used + (uint64_t)5 * width > data_cap || used + (uint64_t)5 * width > 0xffffffffull) { out.status = LCD_ERR_WF; break; }
        if (lane == 0) { W.lo[s] = nlo; W.hi[s] = nhi; W.off[s] = (unsigned int)used; }
        int *d = W.data + used;
        for (int k0 = nlo; k0 <= nhi; k0 += 64) {
            const int k = k0 + lane;
            if (k <= nhi) {
                int i1 = imax(wf_get(W, 0, s - o1 - e1, k - 1), wf_get(W, 1, s - e1, k - 1)) + 1;
                int i2 = imax(wf_get(W, 0, s - o2 - e2, k - 1), wf_get(W, 2, s - e2, k - 1)) + 1;
                int d1 = imax(wf_get(W, 0, s - o1 - e1, k + 1), wf_get(W, 3, s - e1, k + 1));
                int d2 = imax(wf_get(W, 0, s - o2 - e2, k + 1), wf_get(W, 4, s - e2, k + 1));
                int mm = wf_get(W, 0, s - x, k) + 1;
                int mv = imax(imax(mm, imax(i1, i2)), imax(d1, d2));
                if (i1 < 0) i1 = WF_NULL;
                if (i2 < 0) i2 = WF_NULL;
                if (d1 < 0) d1 = WF_NULL;
                if (d2 < 0) d2 = WF_NULL;
                if (mv < 0 || mv > tlen || mv - k > plen || mv - k < 0) mv = WF_NULL;
                const int j = k - nlo;
                d[j] = mv; d[width + j] = i1; d[2 * width + j] = i2; d[3 * width + j] = d1; d[4 * width + j] = d2;
            }
        }
        used += (uint64_t)5 * width; n_off += (unsigned long long)5 * width;
        __syncthreads();
    }
    out.offsets = n_off;
    if (out.status == LCD_OK) {
        out.score = s;
        int n = 0;
        if (lane == 0) {
            // backtrace (oracle/wfa2p.c), ops emitted end->start of the (possibly reversed) pair
            int k = k_end, off = tlen, type = 0;
            int h = off, v = off - k;
            while (v > 0 && h > 0 && s > 0) {
                const int mism = s - x, go1 = s - o1 - e1, ge1 = s - e1, go2 = s - o2 - e2, ge2 = s - e2;
                long long best = (long long)WF_NULL * 16, c;
#define CAND(val, ty) do { int vv_ = (val); c = (long long)vv_ * 16 + (ty); if (vv_ >= 0 && c > best) best = c; } while (0)
                if (type == 0) {
                    CAND(wf_get(W, 0, mism, k) + 1, 9);
                    CAND(wf_get(W, 0, go1, k - 1) + 1, 1); CAND(wf_get(W, 1, ge1, k - 1) + 1, 2);
                    CAND(wf_get(W, 0, go2, k - 1) + 1, 3); CAND(wf_get(W, 2, ge2, k - 1) + 1, 4);
                    CAND(wf_get(W, 0, go1, k + 1), 5); CAND(wf_get(W, 3, ge1, k + 1), 6);
                    CAND(wf_get(W, 0, go2, k + 1), 7); CAND(wf_get(W, 4, ge2, k + 1), 8);
                } else if (type == 1) {
                    CAND(wf_get(W, 0, go1, k - 1) + 1, 1); CAND(wf_get(W, 1, ge1, k - 1) + 1, 2);
                } else if (type == 2) {
                    CAND(wf_get(W, 0, go2, k - 1) + 1, 3); CAND(wf_get(W, 2, ge2, k - 1) + 1, 4);
                } else if (type == 3) {
                    CAND(wf_get(W, 0, go1, k + 1), 5); CAND(wf_get(W, 3, ge1, k + 1), 6);
                } else {
                    CAND(wf_get(W, 0, go2, k + 1), 7); CAND(wf_get(W, 4, ge2, k + 1), 8);
                }
#undef CAND
                if (best < 0) break;
                const int boff = (int)(best / 16), bty = (int)(best % 16);
                if (type == 0) {
                    int nm = off - boff;
                    for (int i = 0; i < nm; ++i) ops[n++] = 'M';
                    off = boff; h = off; v = off - k;
                    if (v <= 0 || h <= 0) break;
                }
                switch (bty) {
                case 9: s = mism; type = 0; ops[n++] = 'X'; --off; break;
                case 1: s = go1; type = 0; ops[n++] = 'I'; --k; --off; break;
                case 2: s = ge1; type = 1; ops[n++] = 'I'; --k; --off; break;
                case 3: s = go2; type = 0; ops[n++] = 'I'; --k; --off; break;
                case 4: s = ge2; type = 2; ops[n++] = 'I'; --k; --off; break;
                case 5: s = go1; type = 0; ops[n++] = 'D'; ++k; break;
                case 6: s = ge1; type = 3; ops[n++] = 'D'; ++k; break;
                case 7: s = go2; type = 0; ops[n++] = 'D'; ++k; break;
                case 8: s = ge2; type = 4; ops[n++] = 'D'; ++k; break;
                }
                h = off; v = off - k;
            }
            if (v > 0 && h > 0) {
                int nm = v < h ? v : h;
                for (int i = 0; i < nm; ++i) ops[n++] = 'M';
                v -= nm; h -= nm;
            }
            while (v > 0) { ops[n++] = 'D'; --v; }
            while (h > 0) { ops[n++] = 'I'; --h; }
            // emission order is end->start of the aligned pair: for rev (left-aligned) that IS start->end of the
            // original pair (src/align.c:433-453 reverses back); otherwise read it backwards.
            uint8_t *ob = outpool + jb.out_off;
            const int maxl = plen + tlen + 1;
            uint64_t o = 0;
            if (jb.want & 1) {
                uint32_t *cig = (uint32_t *)ob;
                int nc = 0;
                for (int i = 0; i < n;) {
                    const uint8_t op = rev ? ops[i] : ops[n - 1 - i];
                    int j = i;
                    while (j < n && (rev ? ops[j] : ops[n - 1 - j]) == op) ++j;
                    const uint32_t bop = op == 'M' ? 7u : op == 'X' ? 8u : op == 'I' ? 1u : 2u;
                    cig[nc++] = ((uint32_t)(j - i) << 4) | bop;
                    i = j;
                }
                out.n_cigar = nc;
                o = lcd_align_up((uint64_t)maxl * 4, 16);
            }
            if (jb.want & 2) {
                uint8_t *pa = ob + o, *ta = pa + maxl;
                int pp = 0, tp = 0;
                for (int i = 0; i < n; ++i) {
                    const uint8_t op = rev ? ops[i] : ops[n - 1 - i];
                    if (op == 'M' || op == 'X') { pa[i] = pat[pp++]; ta[i] = txt[tp++]; }
                    else if (op == 'I') { pa[i] = LCD_GAP; ta[i] = txt[tp++]; }
                    else { pa[i] = pat[pp++]; ta[i] = LCD_GAP; }
                }
                out.aln_len = n;
            }
        }
    }
    if (lane == 0) outs[jid] = out;
#undef PAT
#undef TXT
}

void lcd_launch_wfa(const WfaJob *jobs, const uint8_t *pool, uint8_t *arena, uint8_t *outpool, WfaOut *outs, LcdScoring sc,
                    int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lcd_wfa_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, pool, arena, outpool, outs, sc, n_jobs);
}
