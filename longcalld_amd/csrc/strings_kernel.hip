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


// ---- make_ref_read_aln_str (src/align.c:1056-1146): one lane per (cluster, read); the walk is inherently serial per pair ----
// emit == 0: scan pass (records the both-gap segments, counts the rest of the output); emit == 1: writes the rows.
__global__ void __launch_bounds__(64) lcd_compose_kernel(const CmpJob *jobs, CmpOut *outs, const CmpSeg *segs, int n_jobs, int emit) {
    const int jid = blockIdx.x * 64 + threadIdx.x;
    if (jid >= n_jobs) return;
    const CmpJob jb = jobs[jid];
    const uint8_t *rt = (const uint8_t *)(uintptr_t)jb.rc_t, *rq = (const uint8_t *)(uintptr_t)jb.rc_q;
    const uint8_t *ct = (const uint8_t *)(uintptr_t)jb.cr_t, *cq = (const uint8_t *)(uintptr_t)jb.cr_q;
    int4 *seg = (int4 *)(uintptr_t)jb.seg_off;
    uint8_t *ot = (uint8_t *)(uintptr_t)jb.out_off, *oq = ot + (jb.rc_len + jb.cr_len);
    int i = 0, j = 0, n = 0, nseg = 0;
    while (i < jb.rc_len && j < jb.cr_len) {
        const bool g1 = rq[i] == 5, g2 = ct[j] == 5;
        if (g1 && g2) { // both are gaps on the consensus: the reference aligns the ref segment with the read segment (:1065-1090)
            int rd = 1, qd = 1;
            while (i + rd < jb.rc_len && rq[i + rd] == 5) ++rd;
            while (j + qd < jb.cr_len && ct[j + qd] == 5) ++qd;
            if (!emit) { if (nseg < jb.seg_cap) seg[nseg] = make_int4(i, rd, j, qd); }
            else {
                const CmpSeg sg = segs[jb.seg_first + nseg];
                const uint8_t *pr = (const uint8_t *)(uintptr_t)sg.rows_off, *tr = pr + sg.row_stride;
                for (int k = 0; k < sg.aln_len; ++k) { ot[n + k] = pr[k]; oq[n + k] = tr[k]; }
                n += sg.aln_len;
            }
            ++nseg; i += rd; j += qd;
        } else if (!g1 && !g2) { if (emit) { ot[n] = rt[i]; oq[n] = cq[j]; } ++n; ++i; ++j; }
        else if (g1) { if (emit) { ot[n] = rt[i]; oq[n] = 5; } ++n; ++i; }
        else { if (emit) { ot[n] = 5; oq[n] = cq[j]; } ++n; ++j; }
    }
    for (; i < jb.rc_len; ++i, ++n) if (emit) { ot[n] = rt[i]; oq[n] = 5; }
    for (; j < jb.cr_len; ++j, ++n) if (emit) { ot[n] = 5; oq[n] = cq[j]; }
    outs[jid].n_seg = nseg; outs[jid].aln_len = n;
}

void lcd_launch_compose(const CmpJob *jobs, CmpOut *outs, const CmpSeg *segs, int n_jobs, int emit, hipStream_t stream) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lcd_compose_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, stream, jobs, outs, segs, n_jobs, emit);
}


// ---- gather: many small device segments -> one contiguous staging block (lcd_batch_download: the ref<->cons rows and the K2 cluster lists of a batch are a few
// thousand pieces of ~1 KB scattered over the WFA / chain-output buffers; one copy per piece cost 10 - 20 us each through the runtime) ----
__global__ void __launch_bounds__(64) lcd_gather_kernel(const GatherJob *jobs, int n_jobs) {
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const GatherJob g = jobs[j];
    const uint8_t *src = (const uint8_t *)(uintptr_t)g.src; uint8_t *dst = (uint8_t *)(uintptr_t)g.dst;
    const unsigned n = g.bytes;
    if ((((uintptr_t)src | (uintptr_t)dst) & 3) == 0) {
        const unsigned nw = n >> 2;
        for (unsigned i = threadIdx.x; i < nw; i += 64) ((unsigned *)dst)[i] = ((const unsigned *)src)[i];
        for (unsigned i = (nw << 2) + threadIdx.x; i < n; i += 64) dst[i] = src[i];
    } else
        for (unsigned i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
}
void lcd_launch_gather(const GatherJob *jobs, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lcd_gather_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, n_jobs);
}


// collect_aln_beg_end (src/align.c:630-663) for the anchor jobs of a submission: one lane per K3b alignment walks its CIGAR once (tens of operations: the pair is two
// read ends of ~1.1x the shorter one's length) and returns the prefix sums through the last '=' run and the suffix sums from the first one.  The host used to fetch every
// CIGAR -- one copy of the whole output span, 48 MB for 17 000 jobs -- and scan them between the anchor kernels and the first chain launch.
__global__ void __launch_bounds__(64) lcd_anchor_ends_kernel(const AnchorEndsJob *jobs, AnchorEndsOut *outs, const int n_jobs) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_jobs) return;
    const AnchorEndsJob j = jobs[i];
    const uint32_t *c = (const uint32_t *)(uintptr_t)j.cigar;
    int tr = 0, tq = 0, pre_r = 0, pre_q = 0, first_r = -1, first_q = -1, has = 0;
    for (int k = 0; k < j.n_cigar; ++k) {
        const uint32_t w = c[k]; const int op = (int)(w & 0xf), len = (int)(w >> 4);
        if (op == 7 || op == 0) { if (first_r < 0) { first_r = tr; first_q = tq; } tr += len; tq += len; pre_r = tr; pre_q = tq; has = 1; }
        else if (op == 8) { tr += len; tq += len; }
        else if (op == 2) tr += len;
        else if (op == 1) tq += len;
    }
    AnchorEndsOut o; o.has_eq = has; o.pre_r = pre_r; o.pre_q = pre_q; o.suf_r = has ? tr - first_r : 0; o.suf_q = has ? tq - first_q : 0;
    outs[i] = o;
}
void lcd_launch_anchor_ends(const AnchorEndsJob *jobs, AnchorEndsOut *outs, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lcd_anchor_ends_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, stream, jobs, outs, n_jobs);
}

// the anchored reads' entries of the submission's read table, replaced after the anchor stage (40 bytes per anchored read instead of the whole table a second time)
__global__ void __launch_bounds__(64) lcd_patch_reads_kernel(PoaRead *tab, const ReadPatch *patches, const int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const ReadPatch p = patches[i];
    tab[p.idx] = p.r;
}
void lcd_launch_patch_reads(PoaRead *tab, const ReadPatch *patches, int n, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(lcd_patch_reads_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, tab, patches, n);
}
