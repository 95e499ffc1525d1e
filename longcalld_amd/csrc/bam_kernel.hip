// SURVEY 8(f) f3 on the device, behind the inflate: BAM records are found, measured and their CIGARs collected in the inflated stream in HBM -- what bam_read1 and
// the record loop of collect_ref_seq_bam_main (src/bam_utils.c:1672-1706) do on the calling thread for the reference, one record at a time.
//   lcd_bam_walk_kernel   one lane per range of the stream (a .bai chunk): hops from block_size to block_size (the only serial dependency of the format: ~1 000
//                         records of ~25 KB per 500 kb HiFi chunk, one dependent load each) and leaves a 40-byte descriptor per record;
//   lcd_bam_stat_kernel   one wavefront per record of the wanted reference: reference span (bam_endpos), digar / window-event capacities of its CIGAR (the counts
//                         lcd_digar_batch's host pass makes), the CG:B,I tag behind the placeholder CIGAR of a read with more than 65 535 operations;
//   lcd_bam_cigar_kernel  the kept records' CIGAR words -> a 4-byte aligned pool (records sit at any byte offset of the stream);
//   lcd_errrate_kernel    calc_read_error_rate (src/seq.c:429-436) of a read slice on the qualities in HBM: the same table values added in the same order as the
//                         host loop, so the doubles are the host's.
// Bases and qualities are not moved at all: the digar kernel and the unpack kernel read them where the inflate left them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
// four bytes at any alignment from two aligned words (the stream buffer is padded: the second word may lie behind its end)
__device__ __forceinline__ unsigned ld32u(const uint8_t *p) {
    const uintptr_t a = (uintptr_t)p;
    const unsigned *q = (const unsigned *)(a & ~(uintptr_t)3);
    const unsigned lo = q[0], hi = q[1];
    return __builtin_amdgcn_alignbyte(hi, lo, (unsigned)(a & 3));
}
__device__ __forceinline__ long long wave_sum(long long v) {
    for (int d = 32; d >= 1; d >>= 1) {
        const int lo = __shfl_xor((int)(unsigned)v, d, 64), hi = __shfl_xor((int)(v >> 32), d, 64);
        v += (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
    }
    return v;
}
} // namespace

__global__ void __launch_bounds__(64) lcd_bam_walk_kernel(const BamWalkJob *jobs, BamWalkOut *outs) {
    if (threadIdx.x) return;
    const BamWalkJob j = jobs[blockIdx.x];
    const uint8_t *s = (const uint8_t *)(uintptr_t)j.stream;
    BamRecDesc *d = (BamRecDesc *)(uintptr_t)j.descs;
    uint64_t o = j.ubeg;
    int n = 0, status = 0;
    while (o < j.uend) {
        if (o + 4 > j.usize) break;                                  // end of the file
        const int bs = (int)ld32u(s + o);
        if (bs < 32 || o + 4 + (uint64_t)bs > j.usize) { status = 1; break; } // truncated record
        const uint8_t *r = s + o + 4;
        const unsigned w2 = ld32u(r + 8), w3 = ld32u(r + 12);
        BamRecDesc x;
        x.off = o + 4; x.bs = bs; x.refid = (int)ld32u(r); x.pos = (int)ld32u(r + 4); x.lseq = (int)ld32u(r + 16);
        x.lname = (uint8_t)(w2 & 0xff); x.mapq = (uint8_t)((w2 >> 8) & 0xff); x.nc = (uint16_t)(w3 & 0xffff); x.flag = (uint16_t)(w3 >> 16); x.pad = 0; x.pad2 = 0;
        if (x.lseq < 0 || 32ull + x.lname + 4ull * x.nc + ((uint64_t)x.lseq + 1) / 2 + (uint64_t)x.lseq > (uint64_t)bs) { status = 2; break; } // a field runs past the record
        if (n >= j.cap) { status = 3; break; }
        d[n++] = x;
        if (x.refid == j.tid && (long long)x.pos >= j.reg_end) { status = 4; break; } // sorted input: nothing further overlaps (this record is kept in the list: the loader's checks see it)
        o += 4 + (uint64_t)bs;
    }
    BamWalkOut w; w.n = n; w.status = status; w.next = o; outs[blockIdx.x] = w;
}

// the operations of the record's CIGAR, or of its CG tag when the 16-bit field holds the placeholder `<l_seq>S<ref_len>N`
__global__ void __launch_bounds__(64) lcd_bam_stat_kernel(const BamStatJob *jobs, BamStatOut *outs, const int n_jobs) {
    const int jb = blockIdx.x;
    if (jb >= n_jobs) return;
    const BamStatJob j = jobs[jb];
    const int lane = threadIdx.x;
    const uint8_t *r = (const uint8_t *)(uintptr_t)j.rec;
    const uint8_t *cg = r + 32 + j.lname;
    int nc = j.nc, kind = 0;
    // htslib's bam_tag2cigar (behind the reference's sam_itr_next): a record with a reference and a position whose FIRST operation is `<l_seq>S` may carry its real
    // operations in a CG:B,I (or B,i) tag with at least n_cigar and fewer than 2^29 entries; anything else -- no tag, another type, a shorter array, a broken
    // auxiliary field -- leaves the record's own CIGAR in place, silently
    if (nc >= 1 && (int)ld32u(r) >= 0 && (int)ld32u(r + 4) >= 0) {
        const unsigned c0 = ld32u(cg);
        if ((c0 & 0xf) == 4 && (int)(c0 >> 4) == j.lseq) {
            unsigned long long found = 0; unsigned cnt_found = 0;
            if (lane == 0) { // the auxiliary fields, one after the other
                const uint8_t *aux = cg + 4 * (size_t)nc + ((size_t)j.lseq + 1) / 2 + (size_t)j.lseq, *end = r + j.bs;
                while (aux + 3 <= end) {
                    const uint8_t t0 = aux[0], t1 = aux[1], ty = aux[2]; aux += 3;
                    size_t sz = 0; bool bad = false;
                    if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
                    else if (ty == 's' || ty == 'S') sz = 2;
                    else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
                    else if (ty == 'Z' || ty == 'H') { const uint8_t *q = aux; while (q < end && *q) ++q; if (q >= end) bad = true; else sz = (size_t)(q - aux) + 1; }
                    else if (ty == 'B') {
                        if (aux + 5 > end) bad = true;
                        else {
                            const uint8_t sub = aux[0]; const unsigned cnt = ld32u(aux + 1);
                            const size_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
                            if (!es || (size_t)(end - (aux + 5)) < (size_t)cnt * es) bad = true;
                            else if (t0 == 'C' && t1 == 'G') { if ((sub == 'I' || sub == 'i') && cnt >= (unsigned)nc && cnt < (1u << 29)) { found = (unsigned long long)(uintptr_t)(aux + 5); cnt_found = cnt; } break; } // (bam_aux_get: the first CG tag decides)
                            else sz = 5 + (size_t)cnt * es;
                        }
                    } else bad = true;
                    if (t0 == 'C' && t1 == 'G') break; // a CG tag of another type: not a CIGAR
                    if (bad || (size_t)(end - aux) < sz) break;
                    aux += sz;
                }
            }
            found = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(found >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)found);
            cnt_found = (unsigned)__builtin_amdgcn_readfirstlane((int)cnt_found);
            if (found) { kind = 1; cg = (const uint8_t *)(uintptr_t)found; nc = (int)cnt_found; }
        }
    }
    long long rl = 0, nd = 0, nev = 0, nid = 0;
    if (kind >= 0)
        for (int k = lane; k < nc; k += 64) {
            const unsigned c = ld32u(cg + 4 * (size_t)k); const int op = (int)(c & 0xf); const long long len = (long long)(c >> 4);
            if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) rl += len;            // bam_cigar2rlen
            if (op == 8) { nd += len; nev += len; } else if (op != 3 && op != 9) { ++nd; if (op == 1 || op == 2) { ++nev; ++nid; } } // (lcd_digar_batch's capacity pass)
        }
    rl = wave_sum(rl); nd = wave_sum(nd); nev = wave_sum(nev); nid = wave_sum(nid);
    if (lane == 0) { BamStatOut o; o.rl = rl; o.nd = nd; o.nev = nev; o.nid = nid; o.cig_src = (uint64_t)(uintptr_t)cg; o.nc = nc; o.kind = kind; outs[jb] = o; }
}

__global__ void __launch_bounds__(64) lcd_bam_cigar_kernel(const GatherJob *jobs, const int n_jobs) { // bytes = 4 x operations; dst 4-byte aligned, src anywhere
    const int jb = blockIdx.x;
    if (jb >= n_jobs) return;
    const GatherJob g = jobs[jb];
    const uint8_t *src = (const uint8_t *)(uintptr_t)g.src; unsigned *dst = (unsigned *)(uintptr_t)g.dst;
    const unsigned nw = g.bytes >> 2;
    for (unsigned k = threadIdx.x; k < nw; k += 64) dst[k] = ld32u(src + 4 * (size_t)k);
}

// e = sum over the slice of 10^(-q / 10), in slice order, divided by the length: tab[q] is the host's pow(10.0, -q / 10.0)
__global__ void __launch_bounds__(64) lcd_errrate_kernel(const ErrJob *jobs, const double *tab, double *out, const int n_jobs) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_jobs) return;
    const ErrJob j = jobs[i];
    if (j.len <= 0) { out[i] = 0.0; return; }
    const uint8_t *q = (const uint8_t *)(uintptr_t)j.qual;
    double e = 0.0;
    int k = 0;
    for (; k < j.len && (((uintptr_t)(q + k)) & 3); ++k) e += tab[q[k]];
    for (; k + 4 <= j.len; k += 4) { const unsigned w = *(const unsigned *)(q + k); e += tab[w & 0xff]; e += tab[(w >> 8) & 0xff]; e += tab[(w >> 16) & 0xff]; e += tab[w >> 24]; }
    for (; k < j.len; ++k) e += tab[q[k]];
    out[i] = e / j.len;
}

void lcd_launch_bam_walk(const BamWalkJob *jobs, BamWalkOut *outs, int n_jobs, hipStream_t st) { if (n_jobs > 0) hipLaunchKernelGGL(lcd_bam_walk_kernel, dim3(n_jobs), dim3(64), 0, st, jobs, outs); }
void lcd_launch_bam_stat(const BamStatJob *jobs, BamStatOut *outs, int n_jobs, hipStream_t st) { if (n_jobs > 0) hipLaunchKernelGGL(lcd_bam_stat_kernel, dim3(n_jobs), dim3(64), 0, st, jobs, outs, n_jobs); }
void lcd_launch_bam_cigar(const GatherJob *jobs, int n_jobs, hipStream_t st) { if (n_jobs > 0) hipLaunchKernelGGL(lcd_bam_cigar_kernel, dim3(n_jobs), dim3(64), 0, st, jobs, n_jobs); }
void lcd_launch_errrate(const ErrJob *jobs, const double *tab, double *out, int n_jobs, hipStream_t st) { if (n_jobs > 0) hipLaunchKernelGGL(lcd_errrate_kernel, dim3((n_jobs + 63) / 64), dim3(64), 0, st, jobs, tab, out, n_jobs); }
