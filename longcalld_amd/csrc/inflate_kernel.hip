// inflate_kernel.hip -- SURVEY 8(f) f3: BGZF blocks inflated on the device (gfx950), one wavefront per block.
//
// The reference reads its BAM through htslib's bgzf reader (src/bam_utils.c:1659-1716 -> sam_itr_next -> bgzf_read_block -> zlib inflate, one block at a
// time on the calling thread).  A BGZF block is an independent raw-deflate stream of <= 64 KB of output (SAM specification 4.1), so a file is thousands of
// independent decode jobs; each is a serial bit stream, so a job is ONE wavefront whose lanes all follow the same symbol (uniform control flow) and split
// the work that is parallel inside a block:
//   * the compressed bytes come through a 256-byte lane window (one coalesced load per 256 bytes, the next one already in flight); the bit buffer takes
//     its next 32 bits by v_readlane -- no byte loads on the symbol path;
//   * literal/length and distance codes are looked up in 10- / 9-bit tables in LDS (u16: symbol << 4 | length); the few codes longer than that walk the
//     canonical code bit by bit (count / first-code arrays).  Tables are built by all lanes: lengths -> per-length ranks by ballot -> bit-reversed codes
//     -> strided fills;
//   * a match is copied by the lanes 64 bytes at a time (overlapping matches: source index modulo the distance), finished halves of the output ring go
//     to HBM as coalesced 16-byte stores;
//   * CRC-32 and ISIZE of every block are checked on the device: each lane takes a contiguous slice of the block, the slices' CRCs are combined by
//     multiplication with x^(8 * bytes behind the slice) mod P (the arithmetic of zlib's crc32_combine).
//   * the 32 KB history: see the ring below (most recent bytes in LDS, older ones read back from the block's own output in HBM).
// No zlib source was consulted for the decoder: it follows RFC 1951 (deflate) / RFC 1952 (gzip CRC) and the SAM specification's BGZF section.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(1))) v4u_u; // (a block's place in the inflated stream is wherever the blocks before it end)
typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) unsigned short lds_u16;
typedef __attribute__((address_space(3))) unsigned lds_u32;

#ifndef LCD_INFLATE_LBITS
#define LCD_INFLATE_LBITS 10
#endif
#ifndef LCD_INFLATE_DBITS
#define LCD_INFLATE_DBITS 9
#endif
constexpr int LBITS = LCD_INFLATE_LBITS, DBITS = LCD_INFLATE_DBITS;       // primary table widths
// The most recent output lives in a ring in LDS; finished halves go to HBM at once, and a match that reaches further back than the ring reads the block's
// own output in HBM (written by this wavefront: program order after s_waitcnt vmcnt(0)).  A 32 KB ring (the whole deflate window) would leave room for four
// blocks per CU, one per SIMD, and the decode is serial scalar code -- a chain of dependent LDS round trips (code -> distance code -> source bytes) and
// taken branches, ~1 100 cycles per symbol for a wavefront that has its SIMD to itself: nothing hides them.  Measured on 512 MB of BAM-like data (kernel,
// GB/s of output): 32 KB ring + 11/10-bit tables, 4 blocks per CU: 7.3; 8 KB: 18.1; 4 KB: 22.3; 4 KB + 10/9-bit tables: 24.7; 2 KB + 10/9 bits (6 KB of
// LDS, 26 blocks per CU): 28.1 -- a wavefront is 1.4x slower there (1 580 cycles per symbol: the far matches' trips to HBM, the shared issue slots), the
// chip 3.9x faster.
#ifndef LCD_INFLATE_RING
#define LCD_INFLATE_RING 2048
#endif
constexpr int WINB = LCD_INFLATE_RING, WINM = WINB - 1; // ring bytes
constexpr int HALF = WINB / 2;
constexpr int NEAR = WINB - 512;                        // a match whose source starts within this many bytes is entirely in the ring (a match writes <= 258 bytes ahead)

// dynamic LDS layout (bytes)
constexpr int O_WIN = 0;
constexpr int O_LTAB = O_WIN + WINB;                 // u16[2048]
constexpr int O_DTAB = O_LTAB + (2 << LBITS);        // u16[1024]
constexpr int O_LSYM = O_DTAB + (2 << DBITS);        // u16[288]: litlen symbols sorted by (length, symbol)
constexpr int O_DSYM = O_LSYM + 2 * 288;             // u16[32]
constexpr int O_LENS = O_DSYM + 2 * 32;              // u8[320 + 16]: code lengths of the block's two alphabets
constexpr int O_LCNT = O_LENS + 336;                 // u16[16] x 2: codes per length
constexpr int O_CRCT = O_LTAB;                       // u32[256]: CRC-32 byte table (after the last symbol: over the literal/length table)
constexpr int O_END = O_LCNT + 64;                   // 6 160 B: 26 blocks per CU

__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned sgpr(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ unsigned long long sgpr(unsigned long long v) { return ((unsigned long long)sgpr((unsigned)(v >> 32)) << 32) | (unsigned long long)sgpr((unsigned)v); }
template <typename T> __device__ __forceinline__ T *sgpr(T *p) { return (T *)sgpr((unsigned long long)(uintptr_t)p); }

__constant__ unsigned short c_lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ unsigned char c_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ unsigned short c_dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ unsigned char c_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ unsigned char c_clord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// the compressed stream: dword k of the current 256-byte window sits in lane k
struct BitIn {
    const unsigned *base;  // 256-byte aligned window address (global)
    unsigned cur, nxt;     // this lane's dword of the current / the next window
    int widx;              // next dword of `cur` to take (0..64)
    unsigned long long bb; // bit buffer (uniform)
    int bc;                // valid bits in bb
    const unsigned *lim;   // no window of the stream starts at or behind this address: the job's compressed bytes (+ the gzip trailer) end before it
    int over;              // the stream asked for one: a truncated or crafted block (the decode stops with a status; nothing is read beyond lim + 512 bytes)
};
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ unsigned bi_word(BitIn &s, const int lane) { // the stream's next aligned dword
    if (s.widx == 64) {
        if (s.base + 64 >= s.lim) { s.over = 1; s.widx = 0; } // (the window is kept: whatever is decoded from here on is discarded with the error)
        else { s.base = sgpr(s.base + 64); s.cur = s.nxt; s.nxt = __builtin_nontemporal_load(s.base + 64 + lane); s.widx = 0; }
    }
    const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)s.cur, sgpr(s.widx));
    s.widx = sgpr(s.widx + 1);
    return w;
}
__device__ __forceinline__ void bi_start(BitIn &s, const uint8_t *p, const int lane) { // the stream continues at byte p
    const uintptr_t a = (uintptr_t)p;
    s.base = sgpr((const unsigned *)(a & ~(uintptr_t)255));
    s.cur = __builtin_nontemporal_load(s.base + lane);
    s.nxt = __builtin_nontemporal_load(s.base + 64 + lane);
    s.widx = sgpr((int)(a & 255) >> 2);
    const int sk = (int)(a & 3); // the first dword is taken from its byte `sk` on
    const unsigned w = bi_word(s, lane);
    s.bb = sgpr((unsigned long long)(w >> (8 * sk))); s.bc = sgpr(32 - 8 * sk);
}
__device__ __forceinline__ void bi_fill(BitIn &s, const int lane) { // >= 32 valid bits afterwards
    if (s.bc < 32) { s.bb = sgpr(s.bb | ((unsigned long long)bi_word(s, lane) << s.bc)); s.bc = sgpr(s.bc + 32); }
}
__device__ __forceinline__ unsigned bi_take(BitIn &s, const int n) { // n <= 32 bits already in the buffer
    const unsigned v = sgpr((unsigned)(s.bb & ((1ull << n) - 1ull)));
    s.bb = sgpr(s.bb >> n); s.bc = sgpr(s.bc - n);
    return v;
}
// address of the next unread BYTE (after dropping the bits up to the next byte boundary)
__device__ __forceinline__ const uint8_t *bi_byte_pos(BitIn &s) {
    const int drop = s.bc & 7; s.bb = sgpr(s.bb >> drop); s.bc = sgpr(s.bc - drop);
    return (const uint8_t *)(s.base + s.widx) - (s.bc >> 3);
}

// canonical Huffman tables of one alphabet from its code lengths (LDS `lens`, n symbols): primary table of TB bits, symbols sorted by (length, symbol) and
// the per-length counts for the bit-serial walk of longer codes.  Returns 0, or -1 for an over-subscribed set of lengths.
template <int TB>
__device__ int build_table(const unsigned lds, const unsigned o_lens, const int n, const unsigned o_tab, const unsigned o_sym, const unsigned o_cnt, const int lane) {
    lds_u16 *tab = (lds_u16 *)(uintptr_t)(lds + o_tab), *symt = (lds_u16 *)(uintptr_t)(lds + o_sym), *cnt = (lds_u16 *)(uintptr_t)(lds + o_cnt);
    const lds_u8 *lens = (const lds_u8 *)(uintptr_t)(lds + o_lens);
    for (int k = lane; k < (1 << TB); k += 64) tab[k] = 0;
    // codes per length; a symbol's rank among the symbols of its length
    int run[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) run[l] = 0;
    int myrank[5], mylen[5]; // n <= 320
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int sym = r * 64 + lane;
        const int L = (r * 64 < n && sym < n) ? lens[sym] : 0;
        mylen[r] = L; myrank[r] = 0;
        if (r * 64 < n) {
#pragma unroll
            for (int l = 1; l < 16; ++l) {
                const unsigned long long m = __ballot(L == l);
                if (L == l) myrank[r] = run[l] + __popcll(m & ((1ull << lane) - 1ull));
                run[l] += __popcll(m);
            }
        }
    }
    // first code and first sorted position of every length
    int first[16], offs[16];
    int code = 0, o = 0, left = 1;
    first[0] = 0; offs[0] = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        code = (code + run[l - 1]) << 1;
        first[l] = code; offs[l] = o; o += run[l];
        left = (left << 1) - run[l];
        if (left < 0) return -1;
    }
    if (lane < 16) { int c = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) c = lane == l ? run[l] : c;
        cnt[lane] = (unsigned short)c; }
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int L = mylen[r];
        if (L == 0) continue;
        const int sym = r * 64 + lane;
        int fc = 0, of = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) { fc = L == l ? first[l] : fc; of = L == l ? offs[l] : of; }
        symt[of + myrank[r]] = (unsigned short)sym;
        if (L <= TB) {
            const unsigned rev = __brev((unsigned)(fc + myrank[r])) >> (32 - L);
            const unsigned short e = (unsigned short)((sym << 4) | L);
            for (unsigned k = rev; k < (1u << TB); k += 1u << L) tab[k] = e;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): the table is complete before the first lookup (one wavefront: no barrier needed)
    return 0;
}
// a code longer than the primary table: canonical walk, one bit at a time (RFC 1951 3.2.2); -1: no such code
__device__ __forceinline__ int slow_symbol(BitIn &s, const unsigned lds, const unsigned o_sym, const unsigned o_cnt) {
    const lds_u16 *symt = (const lds_u16 *)(uintptr_t)(lds + o_sym), *cnt = (const lds_u16 *)(uintptr_t)(lds + o_cnt);
    int code = 0, first = 0, index = 0;
    for (int l = 1; l <= 15; ++l) {
        if (s.bc < 1) return -1;
        code |= (int)(s.bb & 1); s.bb = sgpr(s.bb >> 1); s.bc = sgpr(s.bc - 1);
        const int c = sgpr((int)cnt[l]);
        if (code - c < first) return sgpr((int)symt[index + (code - first)]);
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

__device__ __forceinline__ void flush_half(const unsigned lds, uint8_t *out, const int from, const int n, const int lane) { // out[from .. from + n) <- ring; from is a multiple of 16
    const lds_u8 *win = (const lds_u8 *)(uintptr_t)(lds + O_WIN);
    const int n16 = n & ~15;
    for (int k = lane * 16; k < n16; k += 64 * 16) {
        const v4u v = *(const __attribute__((address_space(3))) v4u *)(uintptr_t)(lds + O_WIN + ((from + k) & WINM));
        *(v4u_u *)(out + from + k) = v;
    }
    for (int k = n16 + lane; k < n; k += 64) out[from + k] = win[(from + k) & WINM];
}

__device__ __forceinline__ unsigned gf2_mulmod(unsigned a, unsigned b) { // a * b mod P, reflected CRC-32 polynomial
    unsigned m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}
} // namespace

struct InflateJob { unsigned long long src, dst; unsigned clen, ulen, crc, pad_; }; // src: first byte of the raw deflate stream; dst: its place in the inflated stream
struct InflateOut { int status; unsigned crc; unsigned ulen; unsigned n_sym; unsigned long long t_total, t_tables, t_flush, t_match; }; // (t_*: s_memtime ticks, profiling aid)

// x^(2^k) mod P, k = 0..31 (bits, reflected representation): filled by the host once
__constant__ unsigned c_x2n[32];

template <bool PROF> __device__ __forceinline__ long long tick() { if constexpr (PROF) return clock64(); else return 0; }
template <bool PROF>
__global__ void __launch_bounds__(64) lcd_inflate_kernel(const InflateJob *jobs, InflateOut *outs, const int n_jobs, const int verify) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];
    const unsigned lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)dyn_lds;
    const int lane = threadIdx.x;
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const InflateJob job = sgpr(jobs)[j];
    const uint8_t *src = (const uint8_t *)(((unsigned long long)sgpr((unsigned)(job.src >> 32)) << 32) | (unsigned long long)sgpr((unsigned)(job.src & 0xffffffffu)));
    uint8_t *out = (uint8_t *)(((unsigned long long)sgpr((unsigned)(job.dst >> 32)) << 32) | (unsigned long long)sgpr((unsigned)(job.dst & 0xffffffffu)));
    const int ulen = sgpr((int)job.ulen);
    lds_u8 *win = (lds_u8 *)(uintptr_t)(lds + O_WIN);
    lds_u8 *lens = (lds_u8 *)(uintptr_t)(lds + O_LENS);
    const lds_u16 *ltab = (const lds_u16 *)(uintptr_t)(lds + O_LTAB), *dtab = (const lds_u16 *)(uintptr_t)(lds + O_DTAB);
    // base value | extra bits << 16 of the length / distance symbols: lane = symbol, taken by v_readlane (a constant-memory load is ~200 cycles on the symbol's critical path)
    const unsigned lane_l = lane < 29 ? (unsigned)c_lbase[lane] | ((unsigned)c_lext[lane] << 16) : 0u;
    const unsigned lane_d = lane < 30 ? (unsigned)c_dbase[lane] | ((unsigned)c_dext[lane] << 16) : 0u;
    BitIn s;
    const unsigned clen = sgpr(job.clen);
    s.lim = sgpr((const unsigned *)((((uintptr_t)src + clen + 8) + 255) & ~(uintptr_t)255)); s.over = 0;
    bi_start(s, src, lane);
    int pos = 0, flushed = 0, status = 0;
    bool last = false;
    unsigned n_sym = 0; unsigned long long t_tables = 0, t_flush = 0, t_match = 0;
    const long long t_begin = tick<PROF>();
    while (!last && status == 0) {
        bi_fill(s, lane);
        last = bi_take(s, 1) != 0;
        const int type = (int)bi_take(s, 2);
        if (type == 0) { // stored
            const uint8_t *p = bi_byte_pos(s);
            if (s.over || p + 4 > src + clen) { status = 14; break; }
            // LEN / NLEN straight from the stream (the bytes may straddle the windows: byte loads, uniform address)
            const unsigned len = sgpr((unsigned)(p[0] | (p[1] << 8))), nlen = sgpr((unsigned)(p[2] | (p[3] << 8))); // (loads are per-lane values to the compiler: everything the loop's control depends on is made uniform explicitly)
            if ((len ^ nlen) != 0xffffu) { status = 2; break; }
            if (pos + (int)len > ulen) { status = 3; break; }
            p += 4;
            if (p + len > src + clen) { status = 14; break; } // the stored bytes would come from behind the block
            for (int done = 0; done < (int)len;) {
                const int room = imin((int)len - done, HALF - (pos & (HALF - 1)));
                for (int k = lane; k < room; k += 64) win[(pos + k) & WINM] = p[done + k];
                pos = sgpr(pos + room); done = sgpr(done + room);
                if ((pos & (HALF - 1)) == 0) { __builtin_amdgcn_s_waitcnt(0xc07f); flush_half(lds, out, flushed, pos - flushed, lane); flushed = sgpr(pos); }
            }
            bi_start(s, p + len, lane);
            continue;
        }
        if (type == 3) { status = 4; break; }
        const long long tt0 = tick<PROF>();
        if (type == 1) { // fixed codes
            for (int k = lane; k < 288; k += 64) lens[k] = (uint8_t)(k < 144 ? 8 : k < 256 ? 9 : k < 280 ? 7 : 8);
            if (lane < 32) lens[288 + lane] = 5;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            build_table<LBITS>(lds, O_LENS, 288, O_LTAB, O_LSYM, O_LCNT, lane);
            build_table<DBITS>(lds, O_LENS + 288, 32, O_DTAB, O_DSYM, O_LCNT + 32, lane);
        } else { // dynamic codes
            bi_fill(s, lane);
            const int hlit = (int)bi_take(s, 5) + 257, hdist = (int)bi_take(s, 5) + 1, hclen = (int)bi_take(s, 4) + 4;
            if (hlit > 286 || hdist > 30) { status = 5; break; }
            // the code-length alphabet (19 symbols, <= 7 bits): its table goes where the distance table will be
            if (lane < 19) lens[lane] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            for (int k = 0; k < hclen; ++k) { bi_fill(s, lane); const unsigned v = bi_take(s, 3); if (lane == 0) lens[c_clord[k]] = (uint8_t)v; }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (sgpr(build_table<7>(lds, O_LENS, 19, O_DTAB, O_DSYM, O_LCNT + 32, lane))) { status = 6; break; }
            // the two alphabets' lengths, serially (run-length codes 16 / 17 / 18); kept in a lane window and written out every 64 entries
            const int total = hlit + hdist;
            int n = 0, prev = 0;
            int pend = 0, pbase = 0; // this lane's entry of lens[pbase .. pbase + 64)
            while (n < total && status == 0) {
                bi_fill(s, lane);
                const unsigned e = sgpr((unsigned)dtab[s.bb & 127]);
                if (e == 0) { status = 6; break; }
                s.bb = sgpr(s.bb >> (e & 15)); s.bc = sgpr(s.bc - (int)(e & 15));
                const int sym = (int)(e >> 4);
                int rep = 1, val = sym;
                if (sym == 16) { if (n == 0) { status = 6; break; } rep = 3 + (int)bi_take(s, 2); val = prev; }
                else if (sym == 17) { rep = 3 + (int)bi_take(s, 3); val = 0; }
                else if (sym == 18) { rep = 11 + (int)bi_take(s, 7); val = 0; }
                if (n + rep > total) { status = 6; break; }
                // entries n .. n + rep - 1 <- val
                while (rep > 0) {
                    const int k = n - pbase;                 // first lane to write
                    const int m = imin(rep, 64 - k);
                    if (lane >= k && lane < k + m) pend = val;
                    n += m; rep -= m;
                    if (n - pbase == 64) { lens[pbase + lane] = (uint8_t)pend; pbase += 64; }
                }
                prev = val;
            }
            if (status) break;
            if (lane < n - pbase) lens[pbase + lane] = (uint8_t)pend;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (sgpr((int)lens[256]) == 0) { status = 7; break; } // no end-of-block code
            // the distance lengths follow the literal/length ones directly: move them to a fixed place (lens + 288.. is free: hlit <= 286)
            { const int d = lane < hdist ? lens[hlit + lane] : 0; __builtin_amdgcn_s_waitcnt(0xc07f); if (lane < 32) lens[288 + lane] = (uint8_t)d; }
            for (int k = hlit + lane; k < 288; k += 64) lens[k] = 0;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (sgpr(build_table<LBITS>(lds, O_LENS, 288, O_LTAB, O_LSYM, O_LCNT, lane))) { status = 8; break; }
            if (sgpr(build_table<DBITS>(lds, O_LENS + 288, 32, O_DTAB, O_DSYM, O_LCNT + 32, lane))) { status = 8; break; }
        }
        t_tables += (unsigned long long)(tick<PROF>() - tt0);
        // ---- the block's symbols ----
        for (;;) {
            ++n_sym;
            if (s.over) { status = 14; break; }
            bi_fill(s, lane);
            unsigned e = sgpr((unsigned)ltab[s.bb & ((1u << LBITS) - 1)]);
            int sym;
            if (e) { s.bb = sgpr(s.bb >> (e & 15)); s.bc = sgpr(s.bc - (int)(e & 15)); sym = (int)(e >> 4); }
            else { sym = slow_symbol(s, lds, O_LSYM, O_LCNT); if (sym < 0) { status = 9; break; } }
            if (sym < 256) {
                if (pos >= ulen) { status = 3; break; }
                if (lane == 0) win[pos & WINM] = (uint8_t)sym;
                pos = sgpr(pos + 1);
            } else if (sym == 256) break;
            else {
                sym -= 257;
                if (sym >= 29) { status = 10; break; }
                bi_fill(s, lane);
                const unsigned le = (unsigned)__builtin_amdgcn_readlane((int)lane_l, sym);
                const int len = (int)(le & 0xffffu) + (int)bi_take(s, (int)(le >> 16));
                bi_fill(s, lane);
                e = sgpr((unsigned)dtab[s.bb & ((1u << DBITS) - 1)]);
                int ds;
                if (e) { s.bb = sgpr(s.bb >> (e & 15)); s.bc = sgpr(s.bc - (int)(e & 15)); ds = (int)(e >> 4); }
                else { ds = slow_symbol(s, lds, O_DSYM, O_LCNT + 32); if (ds < 0) { status = 9; break; } }
                if (ds >= 30) { status = 10; break; }
                bi_fill(s, lane);
                const unsigned de = (unsigned)__builtin_amdgcn_readlane((int)lane_d, ds);
                const int dist = (int)(de & 0xffffu) + (int)bi_take(s, (int)(de >> 16));
                if (dist > pos) { status = 11; break; }
                if (pos + len > ulen) { status = 3; break; }
                const long long tm0 = tick<PROF>();
                const int from = pos - dist; // (LDS operations of one wavefront complete in order: the literals before this match are in the ring)
                if (dist > NEAR) { // behind the ring: the block's own output in HBM (flushed: from + len <= pos + 516 - WINB <= flushed)
                    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0): the flushes have landed
                    for (int k = lane; k < len; k += 64) win[(pos + k) & WINM] = out[from + k];
                } else if (dist >= len) {
                    for (int k = lane; k < len; k += 64) win[(pos + k) & WINM] = win[(from + k) & WINM];
                } else if (dist == 1) {
                    const uint8_t b = win[from & WINM];
                    for (int k = lane; k < len; k += 64) win[(pos + k) & WINM] = b;
                } else {
                    for (int k = lane; k < len; k += 64) win[(pos + k) & WINM] = win[(from + (k % dist)) & WINM];
                }
                pos = sgpr(pos + len);
                t_match += (unsigned long long)(tick<PROF>() - tm0);
            }
            if (pos - flushed >= HALF) { // a finished half of the ring goes to HBM
                const long long tf0 = tick<PROF>();
                __builtin_amdgcn_s_waitcnt(0xc07f);
                flush_half(lds, out, flushed, HALF, lane); flushed = sgpr(flushed + HALF);
                t_flush += (unsigned long long)(tick<PROF>() - tf0);
            }
        }
    }
    if (status == 0 && (s.over || bi_byte_pos(s) > src + clen)) status = 14; // the stream ran over the end of the block's compressed bytes (input side of a truncated / crafted block)
    if (status == 0 && pos != ulen) status = 12;
    if (status == 0) { __builtin_amdgcn_s_waitcnt(0xc07f); flush_half(lds, out, flushed, pos - flushed, lane); }
    unsigned crc = 0;
    if (status == 0 && verify) {
        // CRC-32 of the block: byte table in LDS, a contiguous slice per lane, slices combined by x^(8 * bytes behind) mod P
        lds_u32 *ct = (lds_u32 *)(uintptr_t)(lds + O_CRCT);
        for (int k = lane; k < 256; k += 64) { unsigned c = (unsigned)k; for (int b = 0; b < 8; ++b) c = (c & 1) ? (c >> 1) ^ 0xedb88320u : c >> 1; ct[k] = c; }
        __builtin_amdgcn_s_waitcnt(0x0070); // vmcnt(0) lgkmcnt(0): the block's stores have landed, the table is written
        const int per = ((ulen + 63) / 64 + 15) & ~15; // slice length: a multiple of 16 bytes
        const int b0 = imin(lane * per, ulen), b1 = imin(b0 + per, ulen);
        unsigned c = 0xffffffffu;
        int k = b0;
        for (; k + 16 <= b1; k += 16) {
            const v4u v = *(const v4u_u *)(out + k);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned x = w[q];
#pragma unroll
                for (int b = 0; b < 4; ++b) { c = ct[(c ^ x) & 255] ^ (c >> 8); x >>= 8; }
            }
        }
        for (; k < b1; ++k) c = ct[(c ^ out[k]) & 255] ^ (c >> 8);
        c = b1 > b0 ? ~c : 0;
        // shift by the bytes behind this slice: c * x^(8 n) mod P
        unsigned n = (unsigned)(ulen - b1), p = 1u << 31; int kk = 3;
        while (n) { if (n & 1) p = gf2_mulmod(c_x2n[kk & 31], p); n >>= 1; ++kk; }
        c = b1 > b0 ? gf2_mulmod(p, c) : 0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) c ^= __shfl_xor(c, o);
        crc = c;
        if (crc != job.crc) status = 13;
    }
    if (lane == 0) { outs[j].status = status; outs[j].crc = crc; outs[j].ulen = (unsigned)pos; outs[j].n_sym = n_sym;
                     outs[j].t_total = (unsigned long long)(tick<PROF>() - t_begin); outs[j].t_tables = t_tables; outs[j].t_flush = t_flush; outs[j].t_match = t_match; }
}

void lcd_inflate_set_x2n(const unsigned *t32, hipStream_t st) { (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(c_x2n), t32, 32 * sizeof(unsigned), 0, hipMemcpyHostToDevice, st); }
void lcd_launch_inflate(const void *jobs, void *outs, int n_jobs, int verify, int timers, hipStream_t stream) { // timers: the per-phase tick counters of InflateOut are filled (a profiling build of the same kernel)
    if (n_jobs <= 0) return;
    if (timers) hipLaunchKernelGGL(lcd_inflate_kernel<true>, dim3(n_jobs), dim3(64), O_END, stream, (const InflateJob *)jobs, (InflateOut *)outs, n_jobs, verify);
    else hipLaunchKernelGGL(lcd_inflate_kernel<false>, dim3(n_jobs), dim3(64), O_END, stream, (const InflateJob *)jobs, (InflateOut *)outs, n_jobs, verify);
}
