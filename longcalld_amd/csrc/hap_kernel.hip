// hap_kernel.hip -- K5: read <-> haplotype assignment + phase sets (src/assign_hap.c:473-547) on gfx950.
//
// One 64-lane wavefront per chunk problem.  What is order-dependent in the reference stays ordered:
//   * the seeding pass (src/assign_hap.c:497-527) visits variants outward from the seed and reads in cgranges order; a read's
//     score over its variants and the profile update are wave-parallel (one lane per variant of the read), reads are sequential;
//   * the phase-set walk (:388-419) is a serial prefix over the valid variants.
// What is not order-dependent is data-parallel: adjacent-het agree/conflict counts (one lane per read, integer atomics),
// the re-assignment of every read (:436-447; the "fill the missing consensus with 1-other" side effect of
// read_to_cons_allele_score :141-142 is idempotent, so lanes may apply it concurrently), the per-variant arg-max (:449-453).
// Semantics are defined by oracle/assign_hap.c; integer results must match it bit for bit.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
constexpr int CLEAN_HET_SNP = 0x004, CLEAN_HET_INDEL = 0x008, CLEAN_HOM_VAR = 0x080, NOISY_HET = 0x100, NOISY_HOM = 0x200;
constexpr int CLEAN_CATE = CLEAN_HET_SNP | CLEAN_HET_INDEL | CLEAN_HOM_VAR;
constexpr int CDIFF = 8;

__device__ __forceinline__ int wsum(int v) { for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d); return v; }
__device__ __forceinline__ int wmax(int v) { for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d); v = v > o ? v : o; } return v; }
__device__ __forceinline__ int wmin(int v) { for (int d = 32; d >= 1; d >>= 1) { int o = __shfl_xor(v, d); v = v < o ? v : o; } return v; }

#define CONS(v, h) P.cons[(v) * 3 + (h)]
#define PROF(h, v, a) P.prof[(size_t)(h) * P.total_alle + P.alle_off[v] + (a)]

// read_to_cons_allele_score (:127-147) for both haplotypes of one variant, incl. the fill-in side effect
__device__ __forceinline__ void score_var(const HapProb &P, int v, int cate, int al, int &s1, int &s2) {
    const int var_score = (cate == CLEAN_HET_SNP || cate == CLEAN_HET_INDEL) ? 2 : 1;
    int c1 = CONS(v, 1), c2 = CONS(v, 2);
    if (c1 == -1 && c2 == -1) { s1 = 0; s2 = 0; return; }
    if (c1 == -1) { c1 = 1 - c2; CONS(v, 1) = c1; }
    if (c2 == -1) { c2 = 1 - c1; CONS(v, 2) = c2; }
    s1 = c1 == al ? var_score : -var_score;
    s2 = c2 == al ? var_score : -var_score;
}

// update_var_hap_to_cons_alle (:244-268)
__device__ __forceinline__ void update_cons(const HapProb &P, int v, int hap) {
    int max_cov = 0, mi = -1, total = 0;
    const int na = P.alle_off[v + 1] - P.alle_off[v];
    for (int i = 0; i < na; ++i) { const int c = PROF(hap, v, i); total += c; if (c > max_cov) { max_cov = c; mi = i; } }
    if (P.is_ont && P.is_hp[v] == 1 && max_cov < total * 0.67) mi = -1;
    CONS(v, hap) = mi;
}

// init_assign_read_hap_based_on_cons_alle (:151-198) for read r, variants spread over the lanes; result is wave-uniform
__device__ int assign_read_wave(const HapProb &P, int r, int lane) {
    int hs1 = 0, hs2 = 0, u1 = 0, u2 = 0, ag1 = 0, ag2 = 0, cf1 = 0, cf2 = 0;
    const int sv = P.start_var[r], ev = P.end_var[r];
    for (int v0 = sv; v0 <= ev; v0 += 64) {
        const int v = v0 + lane;
        if (v <= ev) {
            const int cate = P.var_cate[v];
            if ((cate & P.target) && !(P.is_hp[v] == 1 || cate == NOISY_HOM)) {
                const int al = P.alleles[P.allele_off[r] + (v - sv)];
                if (al >= 0) {
                    int s1, s2; score_var(P, v, cate, al, s1, s2);
                    const bool clean_snp = (cate & CLEAN_CATE) > 0 && P.var_type[v] == CDIFF;
                    if (s1 != 0) { if (cate != CLEAN_HOM_VAR) u1++; if (clean_snp) { if (s1 > 0) ag1++; else cf1++; } }
                    if (s2 != 0) { if (cate != CLEAN_HOM_VAR) u2++; if (clean_snp) { if (s2 > 0) ag2++; else cf2++; } }
                    if (cate != CLEAN_HOM_VAR) { hs1 += s1; hs2 += s2; }
                }
            }
        }
    }
    hs1 = wsum(hs1); hs2 = wsum(hs2); u1 = wsum(u1); u2 = wsum(u2); ag1 = wsum(ag1); ag2 = wsum(ag2); cf1 = wsum(cf1); cf2 = wsum(cf2);
    int max_hap = 0, max_score = 0, min_hap = 0, min_score = 0;
    if (hs1 > max_score) { max_hap = 1; max_score = hs1; } else if (hs1 < min_score) { min_hap = 1; min_score = hs1; }
    if (hs2 > max_score) { max_hap = 2; max_score = hs2; } else if (hs2 < min_score) { min_hap = 2; min_score = hs2; }
    int hap, a = 0, c = 0;
    if (u1 == 0 && u2 == 0) hap = -1;
    else if (max_score == 0 && min_score == 0) hap = 0;
    else if (max_score > 0) { hap = max_hap; a = max_hap == 1 ? ag1 : ag2; c = max_hap == 1 ? cf1 : cf2; }
    else hap = 3 - min_hap;
    if (lane == 0) { P.n_agree_snps[r] = a; P.n_conflict_snps[r] = c; }
    return hap;
}

// same scoring with one lane per read (re-assignment pass)
__device__ int assign_read_lane(const HapProb &P, int r) {
    int hs1 = 0, hs2 = 0, u1 = 0, u2 = 0, ag1 = 0, ag2 = 0, cf1 = 0, cf2 = 0;
    const int sv = P.start_var[r], ev = P.end_var[r];
    for (int v = sv; v <= ev; ++v) {
        const int cate = P.var_cate[v];
        if ((cate & P.target) == 0) continue;
        if (P.is_hp[v] == 1 || cate == NOISY_HOM) continue;
        const int al = P.alleles[P.allele_off[r] + (v - sv)];
        if (al < 0) continue;
        int s1, s2; score_var(P, v, cate, al, s1, s2);
        const bool clean_snp = (cate & CLEAN_CATE) > 0 && P.var_type[v] == CDIFF;
        if (s1 != 0) { if (cate != CLEAN_HOM_VAR) u1++; if (clean_snp) { if (s1 > 0) ag1++; else cf1++; } }
        if (s2 != 0) { if (cate != CLEAN_HOM_VAR) u2++; if (clean_snp) { if (s2 > 0) ag2++; else cf2++; } }
        if (cate != CLEAN_HOM_VAR) { hs1 += s1; hs2 += s2; }
    }
    int max_hap = 0, max_score = 0, min_hap = 0, min_score = 0;
    if (hs1 > max_score) { max_hap = 1; max_score = hs1; } else if (hs1 < min_score) { min_hap = 1; min_score = hs1; }
    if (hs2 > max_score) { max_hap = 2; max_score = hs2; } else if (hs2 < min_score) { min_hap = 2; min_score = hs2; }
    int hap, a = 0, c = 0;
    if (u1 == 0 && u2 == 0) hap = -1;
    else if (max_score == 0 && min_score == 0) hap = 0;
    else if (max_score > 0) { hap = max_hap; a = max_hap == 1 ? ag1 : ag2; c = max_hap == 1 ? cf1 : cf2; }
    else hap = 3 - min_hap;
    P.n_agree_snps[r] = a; P.n_conflict_snps[r] = c;
    return hap;
}
} // namespace

__global__ void __launch_bounds__(64) lcd_hap_kernel(const HapProb *probs, int n_probs) {
    if ((int)blockIdx.x >= n_probs) return;
    const HapProb P = probs[blockIdx.x];
    const int lane = threadIdx.x;
    // valid variants (ordered compaction)
    int n = 0;
    for (int v0 = 0; v0 < P.n_vars; v0 += 64) {
        const int v = v0 + lane, ok = v < P.n_vars && (P.var_cate[v] & P.target);
        const unsigned long long m = __ballot(ok);
        if (ok) P.valid[n + __popcll(m & ((1ull << lane) - 1))] = v;
        n += __popcll(m);
    }
    if (n == 0) return; // src/assign_hap.c:482-485: nothing is touched
    __syncthreads();
    for (int r = lane; r < P.n_reads; r += 64) { P.haps[r] = 0; P.phase_sets[r] = -1; }
    for (int i = lane; i < n; i += 64) { // var_init_hap_profile_cons_allele :39-63
        const int v = P.valid[i], na = P.alle_off[v + 1] - P.alle_off[v];
        for (int h = 1; h <= 2; ++h) for (int a = 0; a < na; ++a) PROF(h, v, a) = 0;
        int mi = -1;
        if (!(P.is_ont == 1 && P.is_hp[v])) { int mc = 0; for (int a = 0; a < na; ++a) { const int c = P.alle_covs[P.alle_off[v] + a]; if (c > mc) { mc = c; mi = a; } } }
        CONS(v, 0) = mi;
        const int hom = P.var_cate[v] == NOISY_HOM || P.var_cate[v] == CLEAN_HOM_VAR;
        CONS(v, 1) = CONS(v, 2) = hom ? 1 : -1;
    }
    __syncthreads();
    // select_init_var :94-125: first variant of maximal depth per class, classes by priority
    int init = -1;
    {
        int best[4] = {-1, -1, -1, -1}, bi[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30};
        for (int i = lane; i < n; i += 64) {
            const int v = P.valid[i], cate = P.var_cate[v], cov = P.total_cov[v];
            int cls = -1;
            if (cate == CLEAN_HET_SNP) cls = 0; else if (cate == CLEAN_HET_INDEL) cls = 1;
            else if (cate == NOISY_HET) { if (P.var_type[v] == CDIFF) cls = 2; else if (P.is_hp[v] == 0) cls = 3; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (cls == k && cov > best[k]) { best[k] = cov; bi[k] = i; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the reference keeps the first index among equal depths (strict '<' at :101) -- and a depth-0 first hit counts
            const int gb = wmax(best[k]);
            const int gi = wmin(best[k] == gb ? bi[k] : (1 << 30));
            if (init == -1 && gb >= 0 && gi < (1 << 30)) init = gi;
        }
    }
    // ---- seeding pass :496-527 ----
    if (init != -1) {
        for (int k = 0; k < n; ++k) {
            const int vi = k == 0 ? init : (k <= init ? init - k : k);
            const int v = P.valid[vi];
            if (P.var_cate[v] == NOISY_HOM || P.var_cate[v] == CLEAN_HOM_VAR) continue;
            for (int c0 = 0; c0 < P.n_cr; c0 += 64) {
                const int c = c0 + lane;
                int cand = 0;
                if (c < P.n_cr) { const int r = P.cr_read[c]; cand = P.start_var[r] < v + 1 && v < P.end_var[r] + 1 && !P.is_skipped[r]; }
                unsigned long long m = __ballot(cand);
                while (m) {
                    const int b = __ffsll((long long)m) - 1; m &= m - 1;
                    const int r = P.cr_read[c0 + b];
                    if (P.haps[r] != 0) continue;
                    int hap = assign_read_wave(P, r, lane);
                    if (hap == -1) hap = 1;
                    __syncthreads();
                    if (lane == 0) P.haps[r] = hap;
                    // update_var_hap_profile_cons_alle_based_on_read_hap :270-290 (one lane per variant of the read)
                    const int sv = P.start_var[r], ev = P.end_var[r];
                    for (int v0 = sv; v0 <= ev; v0 += 64) {
                        const int u = v0 + lane;
                        if (u <= ev && (P.var_cate[u] & P.target)) {
                            const int al = P.alleles[P.allele_off[r] + (u - sv)];
                            if (al >= 0) {
                                if (hap == 0) { for (int h = 1; h <= 2; ++h) { PROF(h, u, al) += 1; update_cons(P, u, h); } }
                                else { PROF(hap, u, al) += 1; update_cons(P, u, hap); }
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }
    __syncthreads();
    // ---- iterations :530-542 ----
    for (int it = 0; it < 10; ++it) {
        // A: iter_update_var_hap_cons_phase_set :345-422
        int n_het = 0;
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane; int h = 0;
            if (i < n) { const int v = P.valid[i]; h = CONS(v, 1) != -1 && CONS(v, 2) != -1 && CONS(v, 1) != CONS(v, 2) && P.is_hp[v] == 0; P.is_het[i] = h; P.n_agree[i] = 0; P.n_conflict[i] = 0; }
            const unsigned long long m = __ballot(h);
            if (h) P.het[n_het + __popcll(m & ((1ull << lane) - 1))] = i;
            n_het += __popcll(m);
        }
        __syncthreads();
        // agree / conflict of every read over the adjacent het pairs inside its span (check_agree_haps :307-320), one lane per read
        for (int c = lane; c < P.n_cr; c += 64) {
            const int r = P.cr_read[c];
            if (P.is_skipped[r]) continue;
            const int hap = P.haps[r];
            if (hap == 0) continue;
            const int sv = P.start_var[r], ev = P.end_var[r];
            // first het whose variant index >= sv (het[] holds positions in valid[], ascending)
            int lo = 0, hi = n_het;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (P.valid[P.het[mid]] < sv) lo = mid + 1; else hi = mid; }
            for (int k = lo + 1; k < n_het; ++k) {
                const int i = P.het[k], v2 = P.valid[i], v1 = P.valid[P.het[k - 1]];
                if (v2 > ev) break;
                const int a1 = P.alleles[P.allele_off[r] + (v1 - sv)], a2 = P.alleles[P.allele_off[r] + (v2 - sv)];
                if (a1 < 0 || a2 < 0) continue;
                const bool agree = CONS(v1, hap) == a1 && CONS(v2, hap) == a2;
                const bool conflict = CONS(v1, hap) == a1 && CONS(v2, 3 - hap) == a2;
                if (agree) atomicAdd(&P.n_agree[i], 1); else if (conflict) atomicAdd(&P.n_conflict[i], 1);
            }
        }
        __syncthreads();
        int changed1 = 0;
        if (lane == 0) { // serial prefix over the valid variants :388-419
            int flip = 0; long long ps = -1;
            for (int i = 0; i < n; ++i) {
                const int v = P.valid[i];
                if (i == 0) { ps = P.var_type[v] == CDIFF ? P.var_pos[v] : P.var_pos[v] - 1; P.var_ps[v] = ps; continue; }
                if (P.is_het[i] == 1) {
                    if (P.n_agree[i] < 2 && P.n_conflict[i] < 2) ps = P.var_type[v] == CDIFF ? P.var_pos[v] : P.var_pos[v] - 1;
                    else if (P.n_conflict[i] > P.n_agree[i]) flip ^= 1;
                    if (flip == 1) changed1 = 1; // the reference swaps hap 1<->2 twice (:406-411): the alleles end up unchanged
                }
                P.var_ps[v] = ps;
            }
        }
        changed1 = __shfl(changed1, 0);
        __syncthreads();
        // B: iter_update_var_hap_to_cons_alle :425-467
        for (int i = lane; i < n; i += 64) {
            const int v = P.valid[i], na = P.alle_off[v + 1] - P.alle_off[v];
            P.cur_cons[i * 2] = CONS(v, 1); P.cur_cons[i * 2 + 1] = CONS(v, 2);
            for (int h = 0; h <= 2; ++h) for (int a = 0; a < na; ++a) PROF(h, v, a) = 0;
        }
        __syncthreads();
        for (int r = lane; r < P.n_reads; r += 64) { // order-independent: see the file header
            if (P.is_skipped[r]) continue;
            int hap = assign_read_lane(P, r);
            if (hap == -1) hap = 0;
            P.haps[r] = hap;
            const int sv = P.start_var[r], ev = P.end_var[r];
            for (int v = sv; v <= ev; ++v) {
                if ((P.var_cate[v] & P.target) == 0) continue;
                const int al = P.alleles[P.allele_off[r] + (v - sv)];
                if (al < 0) continue;
                if (hap == 0) { atomicAdd(&PROF(1, v, al), 1); atomicAdd(&PROF(2, v, al), 1); } else atomicAdd(&PROF(hap, v, al), 1);
            }
        }
        __syncthreads();
        int changed2 = 0;
        for (int i = lane; i < n; i += 64) {
            const int v = P.valid[i];
            update_cons(P, v, 1); update_cons(P, v, 2);
            if (CONS(v, 1) != P.cur_cons[i * 2] || CONS(v, 2) != P.cur_cons[i * 2 + 1]) changed2 = 1;
        }
        changed2 = __any(changed2);
        __syncthreads();
        if (changed1 == 0 && changed2 == 0) break;
    }
    // update_read_phase_set :322-339
    for (int r = lane; r < P.n_reads; r += 64) {
        if (P.is_skipped[r] || P.start_var[r] == -1) continue;
        long long ps = -1;
        for (int v = P.start_var[r]; v <= P.end_var[r]; ++v) {
            if ((P.var_cate[v] & P.target) == 0) continue;
            if (CONS(v, 1) != -1 && CONS(v, 2) != -1 && CONS(v, 1) != CONS(v, 2)) ps = P.var_ps[v];
            if (ps != -1) break;
        }
        P.phase_sets[r] = ps;
    }
}

void lcd_launch_hap(const HapProb *probs, int n, hipStream_t stream) {
    if (n <= 0) return;
    hipLaunchKernelGGL(lcd_hap_kernel, dim3(n), dim3(64), 0, stream, probs, n);
}
