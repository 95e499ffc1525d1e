// sdust_kernel.hip -- symmetric DUST low-complexity intervals of a chunk's reference on gfx950 (chunk->low_comp_cr, src/bam_utils.c:1573-1581;
// the algorithm is sdust, src/sdust.c:78-163 -- Morgulis et al. 2006 as implemented by H. Li -- a sequential automaton over the sequence).
// What makes it parallel: everything the automaton keeps (the window of the last W - 2 triplets with their counts, the longest suffix in which no
// triplet is over-represented, the "perfect intervals" that start inside the window) is a function of the last W bases only, and a perfect interval
// is never influenced by intervals that start before it.  So the sequence is cut into segments, one LANE per segment: the lane starts the automaton
// early enough for its state to be exact W bases before the segment (2W + 4 words back, counted by the host), runs 2W + 8 bases past its segment so that every interval starting inside has left the window, and
// reports only the intervals that START in its segment, unmerged; the host chains the reports in order with sdust's own merge rule (adjacent or
// overlapping intervals are joined).  Checked byte for byte against the reference's sdust.c itself (oracle/_ref), tests/test_gpu_digar.py.
#include <hip/hip_runtime.h>
#include <mutex>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
constexpr int SD_LANES = 16; // lanes (segments) per workgroup.  The kernel is latency-bound per lane, so what counts is lanes resident per CU: the window ring and the
                             // triplet counters are BYTES (a count never exceeds W <= 64), 256 B per lane, and the first 32 perfect intervals sit in LDS (512 B per
                             // lane; segments inside a tandem repeat spill the rest of their list to the HBM slab): 12 KB per workgroup = 13 workgroups per CU
struct SdPerf { int start, finish, r, l; };
__device__ __forceinline__ int sd_code(unsigned char c) { // seq_nt4_table, src/sdust.c:22-39: raw codes 0..3 and the letters ACGT / acgt
    if (c < 4) return c;
    switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
}

__global__ void __launch_bounds__(SD_LANES) lcd_sdust_kernel(const unsigned char *pool, const SdSeg *segs, int T, int W, int seg, int n_seg, int cap, int *n_out, int2 *out, int4 *pbuf, int pcap, int SD_PL) {
    const int sid = blockIdx.x * SD_LANES + threadIdx.x;
    if (sid >= n_seg) return;
    // one launch serves the segments of MANY sequences (a lane is latency-bound on its own automaton: throughput comes from lanes, i.e. from chunks in flight)
    const SdSeg sg = segs[sid];
    const unsigned char *seq = pool + sg.seq_off;
    const int len = sg.len, a = sg.a, b = min(len, a + seg);
    // sg.from: where the automaton has to start so that its state is exact W bases before the segment (the host counts 2W + 4 triplet words
    // back from there: the window is made of words, and words on both sides of a run of N share it); i == len is the end-of-sequence flush
    const int from = sg.from, to = min(len, b + 2 * W + 8);
    // per-lane tables in LDS, lane-interleaved (entry k of lane t at [k * SD_LANES + t]): window ring, the two triplet counters, find_perfect's copy
    extern __shared__ int sd_lds[];
    unsigned char *const wq = (unsigned char *)sd_lds + threadIdx.x, *const cv = wq + 64 * SD_LANES, *const cw = cv + 64 * SD_LANES, *const c = cw + 64 * SD_LANES;
#define SDX(k) ((k) * SD_LANES)
    // the perfect intervals of the current window (descending start): at most one per (start inside the window, step it was found at) = W x W
    // entries; the first SD_PL of them (all of them for W <= 22) in LDS, lane-interleaved, the rest in an HBM slab
    int4 *const Pl = (int4 *)((unsigned char *)sd_lds + 4 * 64 * SD_LANES) + threadIdx.x; int4 *const Pg = pbuf + (size_t)sid * pcap;
    // (macros, not lambdas: by-reference captures would put the whole automaton state into scratch memory -- measured 30x slower)
#define Pget(j) ((j) < SD_PL ? Pl[(j) * SD_LANES] : Pg[(j)])   /* (x, y, z, w) = (start, finish, r, l) */
#define Pset(j, v) do { const int4 v_ = (v); if ((j) < SD_PL) Pl[(j) * SD_LANES] = v_; else Pg[(j)] = v_; } while (0)
    int qfront = 0, qcount = 0, pn = 0, rv = 0, rw = 0, L = 0, l = 0, nout = 0, bad = 0;
    unsigned t = 0;
    for (int k = 0; k < 64; ++k) { cv[SDX(k)] = 0; cw[SDX(k)] = 0; }
    int2 *mine = out + (size_t)sid * cap;
#define AT(i) (wq[SDX((qfront + (i)) & 63)])
    /* save_masked_regions :91-106, minus the merge into the previous result (done by the host over all segments) */
#define SD_SAVE(start_)                                                                                                    \
    do {                                                                                                                   \
        if (pn == 0) break;                                                                                                \
        const int4 p_ = Pget(pn - 1);                                                                                      \
        if (p_.x >= (start_)) break;                                                                                       \
        if (p_.x >= a && p_.x < b) { if (nout < cap) mine[nout] = make_int2(p_.x, p_.y); ++nout; }                         \
        int i_ = pn - 2;                                                                                                   \
        while (i_ >= 0 && Pget(i_).x < (start_)) --i_;                                                                     \
        pn = i_ + 1;                                                                                                       \
    } while (0)
    for (int i = from; i <= to; ++i) {
        if (i == to && to < len) break;
        const int bcode = i < len ? sd_code(seq[i]) : 4;
        if (bcode < 4) {
            ++l; t = (t << 2 | (unsigned)bcode) & 63u;
            if (l >= 3) {
                const int start = (l - W > 0 ? l - W : 0) + (i + 1 - l);
                SD_SAVE(start);
                { // shift_window :68-89
                    if (qcount >= W - 3 + 1) { const int s = wq[SDX(qfront)]; qfront = (qfront + 1) & 63; --qcount; rw -= --cw[SDX(s)]; if (L > qcount) { --L; rv -= --cv[SDX(s)]; } }
                    wq[SDX((qfront + qcount) & 63)] = (unsigned char)t; ++qcount;
                    ++L; rw += cw[SDX(t)]++; rv += cv[SDX(t)]++;
                    if (cv[SDX(t)] * 10 > T << 1) { int s; do { s = AT(qcount - L); rv -= --cv[SDX(s)]; --L; } while (s != (int)t); }
                }
                if (rw * 10 > L * T) { // find_perfect :108-135
                    for (int k = 0; k < 64; ++k) c[SDX(k)] = cv[SDX(k)];
                    int r = rv, max_r = 0, max_l = 0;
                    for (int x = qcount - L - 1; x >= 0; --x) {
                        const int tt = AT(x);
                        r += c[SDX(tt)]++;
                        const int new_r = r, new_l = qcount - x - 1;
                        if (new_r * 10 > T * new_l) {
                            int j = 0;
                            for (; j < pn; ++j) { const int4 q = Pget(j); if (q.x < x + start) break; if (max_r == 0 || q.z * max_l > max_r * q.w) { max_r = q.z; max_l = q.w; } }
                            if (max_r == 0 || new_r * max_l >= max_r * new_l) {
                                max_r = new_r; max_l = new_l;
                                if (pn >= pcap) { bad = 1; break; }
                                for (int m = pn; m > j; --m) Pset(m, Pget(m - 1));
                                ++pn;
                                Pset(j, make_int4(x + start, qcount + 2 + start, new_r, new_l));
                            }
                        }
                    }
                }
            }
        } else { // N or the end of the sequence: independent pieces (:152-156)
            int start = (l - W + 1 > 0 ? l - W + 1 : 0) + (i + 1 - l);
            while (pn) { SD_SAVE(start); ++start; }
            l = 0; t = 0;
            // (the reference keeps the window and its counters across an N: the next piece's first words see them; kept here as well)
        }
    }
#undef SDX
#undef AT
#undef Pget
#undef Pset
#undef SD_SAVE
    n_out[sid] = bad ? -1 : nout;
}

void lcd_launch_sdust(const unsigned char *pool, const SdSeg *segs, int T, int W, int seg, int n_seg, int cap, int *n_out, int2 *out, int4 *pbuf, int pcap, hipStream_t stream) {
    static std::once_flag attr_once[16]; // (function attributes are per device)
    int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    std::call_once(attr_once[dev], [] { (void)hipFuncSetAttribute((const void *)lcd_sdust_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
    const int pl = pcap < 32 ? pcap : 32;
    const size_t lds = (size_t)SD_LANES * (4 * 64 + (size_t)pl * sizeof(int4));
    if (n_seg > 0) hipLaunchKernelGGL(lcd_sdust_kernel, dim3((n_seg + SD_LANES - 1) / SD_LANES), dim3(SD_LANES), lds, stream, pool, segs, T, W, seg, n_seg, cap, n_out, out, pbuf, pcap, pl);
}
