// sdust_kernel.hip -- symmetric DUST low-complexity intervals of a chunk's reference on gfx950 (chunk->low_comp_cr, src/bam_utils.c:1573-1581;
// the algorithm is sdust, src/sdust.c:78-163 -- Morgulis et al. 2006 as implemented by H. Li -- a sequential automaton over the sequence).
// What makes it parallel: everything the automaton keeps (the window of the last W - 2 triplets with their counts, the longest suffix in which no
// triplet is over-represented, the "perfect intervals" that start inside the window) is a function of the last W bases only, and a perfect interval
// is never influenced by intervals that start before it.  So the sequence is cut into segments, one LANE per segment: the lane starts the automaton
// early enough for its state to be exact W bases before the segment (2W + 4 words back, counted by the host), runs 2W + 8 bases past its segment so that every interval starting inside has left the window, and
// reports only the intervals that START in its segment, unmerged; the host chains the reports in order with sdust's own merge rule (adjacent or
// overlapping intervals are joined).  Checked byte for byte against the reference's sdust.c itself (oracle/_ref), tests/test_gpu_digar.py.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
struct SdPerf { int start, finish, r, l; };
__device__ __forceinline__ int sd_code(unsigned char c) { // seq_nt4_table, src/sdust.c:22-39: raw codes 0..3 and the letters ACGT / acgt
    if (c < 4) return c;
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
}

__global__ void __launch_bounds__(64) lcd_sdust_kernel(const unsigned char *seq, int len, int T, int W, int seg, int n_seg, int cap, const int *seg_from, int *n_out, int2 *out, int4 *pbuf, int pcap) {
    const int sid = blockIdx.x * 64 + threadIdx.x;
    if (sid >= n_seg) return;
    const int a = sid * seg, b = min(len, a + seg);
    // seg_from[sid]: where the automaton has to start so that its state is exact W bases before the segment (the host counts 2W + 4 triplet words
    // back from there: the window is made of words, and words on both sides of a run of N share it); i == len is the end-of-sequence flush
    const int from = seg_from[sid], to = min(len, b + 2 * W + 8);
    // per-lane tables in LDS, lane-interleaved (entry k of lane t at [k * 64 + t]): window ring, the two triplet counters, find_perfect's copy
    extern __shared__ int sd_lds[];
    int *const wq = sd_lds + threadIdx.x, *const cv = wq + 64 * 64, *const cw = cv + 64 * 64, *const c = cw + 64 * 64;
#define SDX(k) ((k) * 64)
    // the perfect intervals of the current window: at most one per (start inside the window, step it was found at) = W x W entries, in an HBM slab
    SdPerf *P = (SdPerf *)(pbuf + (size_t)sid * pcap);
    int qfront = 0, qcount = 0, pn = 0, rv = 0, rw = 0, L = 0, l = 0, nout = 0, bad = 0;
    unsigned t = 0;
    for (int k = 0; k < 64; ++k) { cv[SDX(k)] = 0; cw[SDX(k)] = 0; }
    int2 *mine = out + (size_t)sid * cap;
    auto at = [&](int i) { return wq[SDX((qfront + i) & 63)]; };
    auto save = [&](int start) { // save_masked_regions :91-106, minus the merge into the previous result (done by the host over all segments)
        if (pn == 0 || P[pn - 1].start >= start) return;
        const SdPerf p = P[pn - 1];
        if (p.start >= a && p.start < b) { if (nout < cap) mine[nout] = make_int2(p.start, p.finish); ++nout; }
        int i = pn - 1;
        while (i >= 0 && P[i].start < start) --i;
        pn = i + 1;
    };
    for (int i = from; i <= to; ++i) {
        if (i == to && to < len) break;
        const int bcode = i < len ? sd_code(seq[i]) : 4;
        if (bcode < 4) {
            ++l; t = (t << 2 | (unsigned)bcode) & 63u;
            if (l >= 3) {
                const int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
                save(start);
                { // shift_window :68-89
                    if (qcount >= W - 3 + 1) { const int s = wq[SDX(qfront)]; qfront = (qfront + 1) & 63; --qcount; rw -= --cw[SDX(s)]; if (L > qcount) { --L; rv -= --cv[SDX(s)]; } }
                    wq[SDX((qfront + qcount) & 63)] = (int)t; ++qcount;
                    ++L; rw += cw[SDX(t)]++; rv += cv[SDX(t)]++;
                    if (cv[SDX(t)] * 10 > T << 1) { int s; do { s = at(qcount - L); rv -= --cv[SDX(s)]; --L; } while (s != (int)t); }
                }
                if (rw * 10 > L * T) { // find_perfect :108-135
                    for (int k = 0; k < 64; ++k) c[SDX(k)] = cv[SDX(k)];
                    int r = rv, max_r = 0, max_l = 0;
                    for (int x = qcount - L - 1; x >= 0; --x) {
                        const int tt = at(x);
                        r += c[SDX(tt)]++;
                        const int new_r = r, new_l = qcount - x - 1;
                        if (new_r * 10 > T * new_l) {
                            int j = 0;
                            for (; j < pn && P[j].start >= x + start; ++j) if (max_r == 0 || P[j].r * max_l > max_r * P[j].l) { max_r = P[j].r; max_l = P[j].l; }
                            if (max_r == 0 || new_r * max_l >= max_r * new_l) {
                                max_r = new_r; max_l = new_l;
                                if (pn >= pcap) { bad = 1; break; }
                                for (int m = pn; m > j; --m) P[m] = P[m - 1];
                                ++pn;
                                P[j].start = x + start; P[j].finish = qcount + 2 + start; P[j].r = new_r; P[j].l = new_l;
                            }
                        }
                    }
                }
            }
        } else { // N or the end of the sequence: independent pieces (:152-156)
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
            while (pn) save(start++);
            l = 0; t = 0;
            // (the reference keeps the window and its counters across an N: the next piece's first words see them; kept here as well)
        }
    }
#undef SDX
    n_out[sid] = bad ? -1 : nout;
}

void lcd_launch_sdust(const unsigned char *seq, int len, int T, int W, int seg, int n_seg, int cap, const int *seg_from, int *n_out, int2 *out, int4 *pbuf, int pcap, hipStream_t stream) {
    if (n_seg > 0) hipLaunchKernelGGL(lcd_sdust_kernel, dim3((n_seg + 63) / 64), dim3(64), 4 * 64 * 64 * sizeof(int), stream, seq, len, T, W, seg, n_seg, cap, seg_from, n_out, out, pbuf, pcap);
}
