// strings_kernel.hip -- make_cons_read_aln_str (src/align.c:1029-1054) + wfa_trim_aln_str (:496-562) on gfx950.
// One wavefront per (cluster, read): ordered ballot compaction of the MSA columns that are not gap/gap,
// then the cover-dependent trim as two wave reductions.  Pure byte streaming: reads 2 rows, writes 2 rows.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
__device__ __forceinline__ int wave_max(int v) { for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d); v = v > o ? v : o; } return v; }
__device__ __forceinline__ int wave_min(int v) { for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d); v = v < o ? v : o; } return v; }
}

__global__ void __launch_bounds__(64) lcd_strings_kernel(const StrJob *jobs, uint8_t *pool, StrOut *outs, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int lane = threadIdx.x;
    const StrJob jb = jobs[jid];
    const uint8_t *cons = pool + jb.cons_off;
    const uint8_t *read = pool + (jb.member_addr ? jb.row0 + (uint64_t)(*(const int *)(pool + jb.member_addr)) * (uint64_t)jb.row_stride : jb.read_off);
    uint8_t *t = pool + jb.out_off, *q = t + jb.msa_len;
    int aln = 0;
    for (int c0 = 0; c0 < jb.msa_len; c0 += 64) {
        const int c = c0 + lane;
        uint8_t cb = 5, rb = 5;
        if (c < jb.msa_len) { cb = cons[c]; rb = read[c]; }
        const int keep = rb != 5 || cb != 5;
        const unsigned long long m = __ballot(keep);
        if (keep) { const int p = aln + __popcll(m & ((1ull << lane) - 1)); t[p] = cb; q[p] = rb; }
        aln += __popcll(m);
    }
    __syncthreads();
    StrOut o; o.aln_len = aln; o.target_beg = 0; o.target_end = aln - 1; o.query_beg = 0; o.query_end = aln - 1; o.shift = 0;
    const int fc = jb.full_cover;
    const bool none = LCD_IS_NOT_COVER(fc) || LCD_IS_BOTH_COVER(fc) || (LCD_IS_LEFT_COVER(fc) && LCD_IS_RIGHT_GAP(fc)) ||
                      (LCD_IS_RIGHT_COVER(fc) && LCD_IS_LEFT_GAP(fc));
    if (!none) {
        if (LCD_IS_LEFT_COVER(fc)) {
            int te = -1, qe = -1;
            for (int i = lane; i < aln; i += 64) {
                const uint8_t tb = t[i], qb = q[i];
                if (tb != 5) te = i;
                if (qb != 5 && tb == qb) qe = i;
            }
            te = wave_max(te); qe = wave_max(qe);
            if (qe == -1) qe = te;
            o.aln_len = te + 1; o.target_beg = 0; o.target_end = te; o.query_beg = 0; o.query_end = qe;
            for (int i = qe + 1 + lane; i < o.aln_len; i += 64) q[i] = 5;
        } else if (LCD_IS_RIGHT_COVER(fc)) {
            int ts = 1 << 30, qs = 1 << 30;
            for (int i = aln - 1 - lane; i >= 0; i -= 64) { // descending so the last hit per lane is its smallest index
                const uint8_t tb = t[i], qb = q[i];
                if (tb != 5) ts = i;
                if (qb != 5 && tb == qb) qs = i;
            }
            ts = wave_min(ts); qs = wave_min(qs);
            if (ts == (1 << 30)) ts = 0;
            if (qs == (1 << 30)) qs = ts;
            o.shift = ts; o.aln_len = aln - ts;
            o.target_beg = 0; o.target_end = o.aln_len - 1; o.query_beg = qs - ts; o.query_end = o.aln_len - 1;
            for (int i = lane; i < o.query_beg; i += 64) q[ts + i] = 5;
        }
    }
    if (lane == 0) outs[jid] = o;
}

void lcd_launch_strings(const StrJob *jobs, uint8_t *pool, StrOut *outs, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lcd_strings_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, pool, outs, n_jobs);
}
