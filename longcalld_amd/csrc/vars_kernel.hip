// vars_kernel.hip -- SURVEY 8(f) row f1 on gfx950: the alignment strings of a noisy region (still in HBM after stages S3/S4) ->
// candidate variants + the read x variant allele profile, so that only variants and alleles have to cross PCIe.
//   scan kernel     make_cand_vars_from_msa (src/collect_var.c:1784-1873): one wavefront per ref<->cons string.  Ordered ballot
//                   compaction of the columns that are not gap/gap, then every lane classifies one column from its two neighbours
//                   (a variant starts where a mismatch is not followed by a gap, or where an insertion / deletion run begins);
//                   reference offsets and variant slots are running ballot prefix counts, so records come out in column order.
//   profile kernel  make_vars_from_msa_cons_aln / update_cand_var_profile_from_cons_aln_str{,1,21,2} (:2133-2347): one workgroup per
//                   region.  The two consensus lists are merged in exact_comp_var_site order (:1878) by one lane (tens of entries),
//                   then the (read, variant) pairs are spread over the lanes: each walks its cons<->read string once
//                   (is_match_aln_str :1960, is_match_aln_str_del :1999, is_cover_aln_str :2072, get_full_cover_from_ref_cons_aln_str
//                   :2101) and adds to the coverage counters with LDS-free integer atomics (order-independent sums).
// HBM-bound byte streaming; nothing here is a contraction.  TE / TSD annotation of >= min_sv_len gaps (collect_te_info_from_cons, SURVEY
// a14) and the homopolymer flag (needs chunk reference bases beyond the region) stay on the host, as documented in include/lcd_hotpath.h.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
constexpr uint8_t GAPC = 5;
__device__ __forceinline__ const uint8_t *gp(uint64_t a) { return (const uint8_t *)a; }

struct AlnView { const uint8_t *t, *q; int len, lo, hi; }; // window: columns in [lo, hi] are inside (max of begs, min of ends)

// is_match_aln_str, src/collect_var.c:1960-1997
__device__ int match_str(const AlnView &s, int tp, int len, int *full) {
    int cur = -1, n_eq = 0, n_x = 0, cs = 0, ce = 0;
    const int sp = tp < 0 ? 0 : tp, ep = tp < 0 ? len - 1 : tp + len - 1, stop = tp + len;
    for (int i = 0; i < s.len; ++i) {
        const uint8_t tb = s.t[i];
        cur += tb != GAPC;
        if (cur == stop) break;
        if (i < s.lo) continue;
        if (i > s.hi) break;
        cs |= cur == sp; ce |= cur == ep;
        if (cur >= tp) { if (s.q[i] == tb) ++n_eq; else ++n_x; }
    }
    *full = cs & ce;
    const bool ok = len >= 10 ? ((float)n_eq >= (float)len * 0.9f) : (n_eq == len && n_x == 0);
    return ok ? 1 : (*full ? 0 : -1);
}
// is_match_aln_str_del, :1999-2037
__device__ int match_del(const AlnView &s, int dl, int dr, int *full) {
    int cur = -1, started = 0, non_del = 0, cs = 0, ce = 0;
    const int sp = dl < 0 ? 0 : dl;
    for (int i = 0; i < s.len; ++i) {
        cur += s.t[i] != GAPC;
        if (cur > dr) break;
        if (i < s.lo) continue;
        if (i > s.hi) break;
        cs |= cur == sp; ce |= cur == dr;
        if (cur >= dl && cur < dr) { if (!started) started = 1; else non_del += s.q[i] != GAPC; }
    }
    *full = cs & ce;
    return *full ? (non_del == 0) : -1;
}
// is_cover_aln_str, :2072-2093
__device__ int cover_str(const AlnView &s, int tp, int len) {
    int cur = -1, cs = 0, ce = 0;
    const int sp = tp < 0 ? 0 : tp, ep = tp < 0 ? len - 1 : tp + len - 1;
    for (int i = 0; i < s.len; ++i) {
        cur += s.t[i] != GAPC;
        if (i < s.lo) continue;
        if (i > s.hi) break;
        cs |= cur == sp; ce |= cur == ep;
        if (cs & ce) return 1;
    }
    return 0;
}
// get_full_cover_from_ref_cons_aln_str, :2101-2128
__device__ int cover_via_ref(const AlnView &cr, const AlnView &rc, int beg_ref, int end_ref) {
    int cur_r = -1, cur_c = -1, bc = -1, ec = -1, reach = 0;
    for (int i = 0; i < rc.len; ++i) {
        const uint8_t qb = rc.q[i];
        cur_r += rc.t[i] != GAPC; cur_c += qb != GAPC;
        if (i < rc.lo) continue;
        if (i > rc.hi) break;
        if (cur_r == beg_ref && bc == -1) bc = cur_c;
        reach |= cur_r == end_ref;
        if (reach && qb != GAPC) { ec = cur_c; break; }
    }
    return cover_str(cr, bc, ec - bc + 1);
}
} // namespace

__global__ void __launch_bounds__(64) lcd_vars_scan_kernel(const VarScanJob *jobs, VarScanOut *outs, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int lane = threadIdx.x;
    const VarScanJob jb = jobs[jid];
    const uint8_t *rt = gp(jb.rc_t), *rq = gp(jb.rc_q);
    uint8_t *R = (uint8_t *)jb.work_off, *C = R + jb.row_cap;
    VarRec *recs = (VarRec *)jb.rec_off;
    const unsigned long long below = (1ull << lane) - 1;
    int L = 0;
    for (int c0 = 0; c0 < jb.rc_len; c0 += 64) { // columns that are not gap/gap, in order (:1863-1868)
        const int c = c0 + lane;
        uint8_t a = GAPC, b = GAPC;
        if (c < jb.rc_len) { a = rt[c]; b = rq[c]; }
        const int keep = a != GAPC || b != GAPC;
        const unsigned long long m = __ballot(keep);
        if (keep) { const int p = L + __popcll(m & below); R[p] = a; C[p] = b; }
        L += __popcll(m);
    }
    __syncthreads();
    int n_ref = 0, n_var = 0;
    for (int c0 = 0; c0 < L; c0 += 64) {
        const int i = c0 + lane;
        int cls = 0, start = 0, is_ref = 0; // cls 0 equal, 1 mismatch, 2 insertion column, 3 deletion column
        uint8_t a = GAPC, b = GAPC, pa = 0, pb = 0;
        if (i < L) {
            a = R[i]; b = C[i];
            is_ref = a != GAPC;
            cls = a == b ? 0 : (a == GAPC ? 2 : (b == GAPC ? 3 : 1));
            if (cls == 1) start = (i + 1 == L) || (R[i + 1] != GAPC && C[i + 1] != GAPC);
            else if (cls >= 2) {
                int pcls = 0;
                if (i > 0) { pa = R[i - 1]; pb = C[i - 1]; pcls = pa == pb ? 0 : (pa == GAPC ? 2 : (pb == GAPC ? 3 : 1)); }
                start = pcls != cls;
            }
        }
        const unsigned long long mr = __ballot(is_ref), ms = __ballot(start);
        if (start) {
            const int slot = n_var + __popcll(ms & below);
            VarRec v;
            v.ref_off = n_ref + __popcll(mr & below); v.col = i; v.from_cons = 0; v.cate = 0; v.total_cov = 0; v.alle_cov0 = 0; v.alle_cov1 = 0; v.delta0 = 0; v.delta1 = 0; v.src = 0; v.alt_off = 0;
            if (cls == 1) { v.type = 8; v.ref_len = 1; v.alt_len = 1; v.ref_base = a; v.alt_ref_base = 0; }
            else {
                int g = 1;
                if (cls == 2) { while (i + g < L && R[i + g] == GAPC && C[i + g] != GAPC) ++g; }
                else { while (i + g < L && R[i + g] != GAPC && C[i + g] == GAPC) ++g; }
                v.type = cls == 2 ? 1 : 2; v.ref_len = cls == 2 ? 0 : g; v.alt_len = cls == 2 ? g : 0; v.ref_base = 0; v.alt_ref_base = i >= 1 ? pb : 4;
            }
            if (slot < jb.rec_cap) recs[slot] = v;
        }
        n_ref += __popcll(mr); n_var += __popcll(ms);
    }
    if (lane == 0) { outs[jid].n_vars = n_var; outs[jid].n_cols = L; }
}

__device__ __forceinline__ int site_cmp(const VarRec &a, const uint8_t *ca, const VarRec &b, const uint8_t *cb) { // exact_comp_var_site, :1878-1898
    const int pa = a.type == 8 ? a.ref_off : a.ref_off - 1, pb = b.type == 8 ? b.ref_off : b.ref_off - 1;
    if (pa != pb) return pa < pb ? -1 : 1;
    if (a.type != b.type) return a.type < b.type ? -1 : 1;
    if (a.ref_len != b.ref_len) return a.ref_len < b.ref_len ? -1 : 1;
    if (a.alt_len != b.alt_len) return a.alt_len < b.alt_len ? -1 : 1;
    if (a.type == 8 || a.type == 1)
        for (int k = 0; k < a.alt_len; ++k) { const int d = (int)ca[a.col + k] - (int)cb[b.col + k]; if (d) return d; }
    return 0;
}

__global__ void __launch_bounds__(256) lcd_vars_profile_kernel(const VarRegJob *jobs, VarRegOut *outs, const StrJob *sjobs, const StrOut *souts, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int tid = threadIdx.x;
    const VarRegJob jb = jobs[jid];
    VarRec *mv = (VarRec *)jb.out_rec;
    __shared__ int s_n;
    if (tid == 0) { // merge (update_cand_var_profile_from_cons_aln_str2, :2206-2238) + per-cluster running ref/alt length difference
        const VarRec *h0 = (const VarRec *)jb.rec[0], *h1 = (const VarRec *)jb.rec[1];
        const uint8_t *c0 = gp(jb.cons[0]), *c1 = gp(jb.cons[1]);
        int n = 0, d0 = 0, d1 = 0, i1 = 0, i2 = 0, ao = 0;
        const int n0 = jb.n_rec[0], n1 = jb.n_cons == 2 ? jb.n_rec[1] : 0;
        while (i1 < n0 || i2 < n1) {
            int r = i1 >= n0 ? 1 : (i2 >= n1 ? -1 : site_cmp(h0[i1], c0, h1[i2], c1));
            VarRec v;
            if (r < 0) { v = h0[i1++]; v.src = 0; v.from_cons = 1; v.cate = 0x100; }
            else if (r > 0) { v = h1[i2++]; v.src = 1; v.from_cons = 2; v.cate = 0x100; }
            else { v = h0[i1++]; ++i2; v.src = 0; v.from_cons = 3; v.cate = 0x200; }
            if (jb.n_cons == 1) { v.from_cons = 1; v.cate = 0x200; } // make_cand_vars_from_baln0 :1870, update_..._str1 :2168
            v.delta0 = d0; v.delta1 = d1; v.alt_off = ao; ao += v.alt_len;
            const int dd = v.type == 1 ? -v.alt_len : (v.type == 2 ? v.ref_len : 0);
            if (v.from_cons & 1) d0 += dd;
            if (v.from_cons & 2) d1 += dd;
            mv[n++] = v;
        }
        s_n = n;
        outs[jid].n_vars = n; outs[jid].alt_bytes = ao;
    }
    __syncthreads();
    const int n = s_n;
    const int rows0 = jb.n_rows[0], rows = rows0 + (jb.n_cons == 2 ? jb.n_rows[1] : 0);
    int8_t *prof = (int8_t *)jb.out_prof;
    int *pse = (int *)jb.out_se;
    {   // alt_seq of every merged variant -> the region's alt pool
        uint8_t *pool = (uint8_t *)jb.out_alt;
        for (int vi = tid; vi < n; vi += blockDim.x) {
            const VarRec v = mv[vi];
            const uint8_t *src = gp(jb.cons[v.src]) + v.col;
            for (int k = 0; k < v.alt_len; ++k) pool[v.alt_off + k] = src[k];
        }
    }
    if (n == 0) { for (int r = tid; r < rows; r += blockDim.x) { pse[2 * r] = -1; pse[2 * r + 1] = -2; } return; }
    for (int p = tid; p < rows * n; p += blockDim.x) {
        const int row = p / n, vi = p - row * n;
        const int c = row >= rows0, j = c ? row - rows0 : row;
        const int si = jb.str_first[c] + j;
        const StrJob sj = sjobs[si]; const StrOut so = souts[si];
        AlnView cr;
        cr.t = gp(sj.out_off + so.shift); cr.q = gp(sj.out_off + sj.msa_len + so.shift); cr.len = so.aln_len;
        cr.lo = so.query_beg > so.target_beg ? so.query_beg : so.target_beg; cr.hi = so.query_end < so.target_end ? so.query_end : so.target_end;
        const VarRec v = mv[vi];
        const int clu_idx = c + 1;
        const int mine = (v.from_cons & clu_idx) != 0;
        const int delta = c ? v.delta1 : v.delta0;
        const int vb = v.ref_off, ve = v.type == 1 ? vb : vb + v.ref_len - 1;
        int full = 0, al = 0;
        if (mine) {
            if (v.type == 8) al = match_str(cr, vb - delta, 1, &full);
            else if (v.type == 1) al = match_str(cr, vb - delta, v.alt_len, &full);
            else al = match_del(cr, vb - delta - 1, vb - delta, &full);
        } else if (v.type == 8) full = cover_str(cr, vb - delta, 1);
        else if (v.type == 1) full = cover_str(cr, vb - delta, v.ref_len + 1);
        else {
            AlnView rc; rc.t = gp(jb.rc_t[c]); rc.q = gp(jb.rc_q[c]); rc.len = jb.rc_len[c]; rc.lo = 0; rc.hi = jb.rc_len[c] - 1;
            full = cover_via_ref(cr, rc, vb - 1, ve + 1);
        }
        if (full) {
            atomicAdd(&mv[vi].total_cov, 1);
            if (al == 0) atomicAdd(&mv[vi].alle_cov0, 1); else if (al == 1) atomicAdd(&mv[vi].alle_cov1, 1);
        }
        prof[p] = full ? (int8_t)al : (int8_t)-2;
    }
    __syncthreads();
    for (int r = tid; r < rows; r += blockDim.x) { // update_read_var_profile_with_allele, src/bam_utils.c:248: first / last fully covered variant
        int s = -1, e = -2;
        for (int vi = 0; vi < n; ++vi) if (prof[(size_t)r * n + vi] != -2) { if (s == -1) s = vi; e = vi; }
        pse[2 * r] = s; pse[2 * r + 1] = e;
    }
}

void lcd_launch_vars_scan(const VarScanJob *jobs, VarScanOut *outs, int n_jobs, hipStream_t stream) {
    if (n_jobs > 0) hipLaunchKernelGGL(lcd_vars_scan_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, outs, n_jobs);
}
void lcd_launch_vars_profile(const VarRegJob *jobs, VarRegOut *outs, const StrJob *sjobs, const StrOut *souts, int n_jobs, hipStream_t stream) {
    if (n_jobs > 0) hipLaunchKernelGGL(lcd_vars_profile_kernel, dim3(n_jobs), dim3(256), 0, stream, jobs, outs, sjobs, souts, n_jobs);
}
