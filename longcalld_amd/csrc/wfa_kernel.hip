// wfa_kernel.hip -- K3: wavefront gap-affine-2p end-to-end alignment on gfx950 (CDNA4, wave64), bounded memory.
//
// Replaces what src/align.c:374-460 (wfa_end2end_aln, heuristic none, affine-2p) asks of WFA2-lib.  Semantics are defined by
// oracle/wfa2p.c (every wavefront retained, direct backtrace); this file must match it bit for bit while keeping only
//   * a RING of wavefront values: M of the last max(x, o1+e1, o2+e2) + 1 scores, I1/D1 of the last e1 + 1, I2/D2 of the last e2 + 1
//     (36 rows at the default penalties) -- in LDS for the small jobs (one wavefront per job), in HBM/L2 for wide fronts (256 threads per job);
//   * ONE BYTE of backtrace decisions per (score, diagonal): the source of M in the oracle's priority order (a nibble) and the
//     open / extend choice of I1, I2, D1, D2 (a bit each).  The oracle's backtrace compares (offset << 4 | type) of the sources of the cell it
//     stands on; those sources and that comparison are exactly what the forward step has in registers, so the decision is recorded there and the
//     backtrace follows bytes instead of re-reading five retained offsets per step (20 B -> 1 B per diagonal);
//   * the decision bytes of ONE block of scores, plus a snapshot of the value ring at every block start: when the backtrace walks into an
//     earlier block, that block is re-computed from its snapshot (checkpoint / recompute, at most 2x the forward work).  A 10 kb insertion
//     (score ~ 10 000, 14 000 diagonals) needs ~40 MB instead of the ~2 GB of retained wavefronts.
// Match runs are not stored anywhere: the backtrace yields the list of mismatches / gap runs, and a forward replay from (0, 0) re-extends the
// matches between them (wave-cooperative 64-base compares + ballot), which is the extension the forward pass performed at those cells.
// The extend step of the forward pass compares 8 packed bases per load; the end test is a workgroup OR; gap chains of the backtrace are followed
// 64 steps at a time (ballot over the extend bits).  gap_aln == LEFT (src/align.c:409-453) is done by index reversal, not by copying.
#include <hip/hip_runtime.h>
#include <mutex>
#include "lcd_types.h"
#include "lcd_kernels.h"

#define WF_NULL (-(1 << 29))

namespace {

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

enum { OP_M = 0, OP_X = 1, OP_I = 2, OP_D = 3 };

struct Seq { const uint8_t *pat, *txt; int plen, tlen; bool rev; };

// matching bases from (v, h) in working coordinates (reversed when left-aligning), 8 packed bases per compare
__device__ __forceinline__ int ext_run(const Seq &q, int v, int h) {
    const int n = imin(q.plen - v, q.tlen - h);
    int m = 0;
    if (!q.rev) {
        const uint8_t *p = q.pat + v, *t = q.txt + h;
        if (n <= 0 || p[0] != t[0]) return 0; // (most diagonals off the alignment path stop at once)
        while (m + 8 <= n) {
            unsigned long long a, b;
            __builtin_memcpy(&a, p + m, 8); __builtin_memcpy(&b, t + m, 8);
            const unsigned long long d = a ^ b;
            if (d) return m + (__builtin_ctzll(d) >> 3);
            m += 8;
        }
        while (m < n && p[m] == t[m]) ++m;
    } else {
        const uint8_t *p = q.pat + (q.plen - 1 - v), *t = q.txt + (q.tlen - 1 - h); // p[-j] is base v + j
        if (n <= 0 || p[0] != t[0]) return 0;
        while (m + 8 <= n) {
            unsigned long long a, b;
            __builtin_memcpy(&a, p - m - 7, 8); __builtin_memcpy(&b, t - m - 7, 8);
            const unsigned long long d = a ^ b;
            if (d) return m + (__builtin_clzll(d) >> 3);
            m += 8;
        }
        while (m < n && p[-m] == t[-m]) ++m;
    }
    return m;
}
// the same by one wavefront: 64 bases per step, first mismatch by ballot (forward replay of the backtrace)
__device__ __forceinline__ int coop_ext(const Seq &q, int v, int h, int lane) {
    const int n = imin(q.plen - v, q.tlen - h);
    int m = 0;
    for (;;) {
        const int j = m + lane;
        bool mis = true;
        if (j < n) mis = q.rev ? q.pat[q.plen - 1 - v - j] != q.txt[q.tlen - 1 - h - j] : q.pat[v + j] != q.txt[h + j];
        const unsigned long long b = __ballot(mis);
        if (b) return m + __builtin_ctzll(b);
        m += 64;
    }
}

} // namespace

template <int NT, bool LDSR, bool WIDE = false>
__global__ void __launch_bounds__(NT) lcd_wfa_kernel(const WfaJob *jobs, const uint8_t *pool, uint8_t *arena, uint8_t *outpool, WfaOut *outs,
                                                     LcdScoring sc, int n_jobs) {
    extern __shared__ int lds_ring[];
    __shared__ int sh_s, sh_k, sh_type, sh_nrec, sh_runop, sh_runlen, sh_err;
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const WfaJob jb = jobs[jid];
    Seq q;
    q.pat = pool + jb.p_off; q.txt = pool + jb.t_off; q.plen = jb.plen; q.tlen = jb.tlen; q.rev = jb.gap_aln == 1;
    const int plen = jb.plen, tlen = jb.tlen;
    const int x = sc.mismatch, o1 = sc.o1, e1 = sc.e1, o2 = sc.o2, e2 = sc.e2;
    const int s_cap = jb.s_cap, B = jb.blk_rows;
    const WfaLayout L = wfa_layout(plen, tlen, s_cap, B, jb.n_ckpt, LDSR ? 1 : 0, x, o1, e1, o2, e2);
    uint8_t *ws = arena + jb.ws_off;
    int *ring;
    if constexpr (LDSR) ring = lds_ring; else ring = (int *)(ws + L.ring);
    int2 *rec = (int2 *)(ws + L.rec), *runs = (int2 *)(ws + L.runs);
    uint8_t *ch = ws + L.choice;
    const int w_cap = L.w_cap, rm = L.rm, r1 = L.r1, r2 = L.r2;
    const int cbase = imin(s_cap, plen) + 1; // ring column of diagonal k = k + cbase (one guard column each side stays null)
    const int k_end = tlen - plen;
    const int ring_n = L.rows * w_cap;
    auto wid = [&](int s) { return imin(s, plen) + imin(s, tlen) + 1; };
    auto slot = [&](int base, int depth, int s) { int r = s % depth; if (r < 0) r += depth; return ring + (size_t)(base + r) * w_cap; };

    // one score: every diagonal of the row from the ring, the decision byte, the extension; returns 1 on the thread that reached the end
    auto compute_row = [&](int s, uint8_t *chrow) -> int {
        const int klo = -imin(s, plen), khi = imin(s, tlen);
        int *Mw = slot(0, rm, s), *I1w = slot(rm, r1, s), *D1w = slot(rm + r1, r1, s), *I2w = slot(rm + 2 * r1, r2, s), *D2w = slot(rm + 2 * r1 + r2, r2, s);
        int done = 0;
        if (s == 0) {
            if (tid == 0) {
                const int h = ext_run(q, 0, 0);
                Mw[cbase] = h; I1w[cbase] = I2w[cbase] = D1w[cbase] = D2w[cbase] = WF_NULL;
                chrow[0] = 0;
                done = k_end == 0 && h >= tlen;
            }
            return done;
        }
        const int *Mx = slot(0, rm, s - x), *Mo1 = slot(0, rm, s - o1 - e1), *Mo2 = slot(0, rm, s - o2 - e2);
        const int *I1e = slot(rm, r1, s - e1), *D1e = slot(rm + r1, r1, s - e1), *I2e = slot(rm + 2 * r1, r2, s - e2), *D2e = slot(rm + 2 * r1 + r2, r2, s - e2);
        // WIDE (the launch of an SV-size job class, lcd_host.cpp run_wfa_stage): U diagonals per thread IN FLIGHT on rows of thousands of diagonals (round 6).  The nine ring
        // values of all U are fetched before anything is stored -- the row being written and the rows being read are different slots of the same ring, which the compiler
        // cannot know, so the plain loop below is one memory round trip per diagonal: a 20 000-diagonal front of a 10 kb gap was 78 such trips per row and thread.  The
        // first base of every extension is probed the same way (most diagonals off the alignment's path stop there).  A separate instantiation: inside the common
        // kernel the batched form cost the clean-read shapes 1 - 2 % (registers shared with the narrow rows), profiles/NOTES_r06.md 9.
        constexpr int U = 4;
        if (WIDE && khi - klo + 1 >= 2 * U * NT) {
        for (int k0 = klo + tid; k0 <= khi; k0 += U * NT) {
            int vmx[U], vi1o[U], vi1x[U], vi2o[U], vi2x[U], vd1o[U], vd1x[U], vd2o[U], vd2x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = imin(k0 + u * NT, khi), c = k + cbase; // (past the row's end: the last diagonal once more, not stored)
                vmx[u] = Mx[c] + 1;
                vi1o[u] = Mo1[c - 1] + 1; vi1x[u] = I1e[c - 1] + 1; vi2o[u] = Mo2[c - 1] + 1; vi2x[u] = I2e[c - 1] + 1;
                vd1o[u] = Mo1[c + 1]; vd1x[u] = D1e[c + 1]; vd2o[u] = Mo2[c + 1]; vd2x[u] = D2e[c + 1];
            }
            int vmv[U], vi1[U], vi2[U], vd1[U], vd2[U], vcode[U]; bool live[U];
            uint8_t pf[U], tf[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = imin(k0 + u * NT, khi);
                const int mx = vmx[u], i1o = vi1o[u], i1x = vi1x[u], i2o = vi2o[u], i2x = vi2x[u], d1o = vd1o[u], d1x = vd1x[u], d2o = vd2o[u], d2x = vd2x[u];
                int i1 = imax(i1o, i1x), i2 = imax(i2o, i2x), d1 = imax(d1o, d1x), d2 = imax(d2o, d2x);
                int mv = imax(imax(mx, imax(i1, i2)), imax(d1, d2));
                int code = mx == mv ? 9 : d2x == mv ? 8 : d2o == mv ? 7 : d1x == mv ? 6 : d1o == mv ? 5 : i2x == mv ? 4 : i2o == mv ? 3 : i1x == mv ? 2 : 1; // (as in the plain loop below)
                code |= (i1x >= i1o ? 16 : 0) | (i2x >= i2o ? 32 : 0) | (d1x >= d1o ? 64 : 0) | (d2x >= d2o ? 128 : 0);
                if (i1 < 0) i1 = WF_NULL;
                if (i2 < 0) i2 = WF_NULL;
                if (d1 < 0) d1 = WF_NULL;
                if (d2 < 0) d2 = WF_NULL;
                live[u] = !(mv < 0 || mv > tlen || mv - k > plen || mv - k < 0);
                if (!live[u]) mv = WF_NULL;
                vmv[u] = mv; vi1[u] = i1; vi2[u] = i2; vd1[u] = d1; vd2[u] = d2; vcode[u] = code;
                pf[u] = 0; tf[u] = 1;
                if (live[u]) { // the first base of the extension from (mv - k, mv)
                    const int v = mv - k, h = mv;
                    if (v < plen && h < tlen) { pf[u] = q.rev ? q.pat[plen - 1 - v] : q.pat[v]; tf[u] = q.rev ? q.txt[tlen - 1 - h] : q.txt[h]; }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * NT;
                if (k > khi) continue;
                const int c = k + cbase;
                int mv = vmv[u];
                if (live[u]) {
                    if (pf[u] == tf[u]) mv += ext_run(q, mv - k, mv);
                    if (k == k_end && mv >= tlen) done = 1;
                }
                Mw[c] = mv; I1w[c] = vi1[u]; I2w[c] = vi2[u]; D1w[c] = vd1[u]; D2w[c] = vd2[u];
                chrow[k - klo] = (uint8_t)vcode[u];
            }
        }
        } else
        for (int k = klo + tid; k <= khi; k += NT) {
            const int c = k + cbase;
            const int mx = Mx[c] + 1;
            const int i1o = Mo1[c - 1] + 1, i1x = I1e[c - 1] + 1, i2o = Mo2[c - 1] + 1, i2x = I2e[c - 1] + 1;
            const int d1o = Mo1[c + 1], d1x = D1e[c + 1], d2o = Mo2[c + 1], d2x = D2e[c + 1];
            int i1 = imax(i1o, i1x), i2 = imax(i2o, i2x), d1 = imax(d1o, d1x), d2 = imax(d2o, d2x);
            int mv = imax(imax(mx, imax(i1, i2)), imax(d1, d2));
            // the oracle's backtrace at this cell: the largest (offset << 4 | type) among the sources, i.e. on equal offsets the higher type:
            // mismatch 9 > D2 ext 8 > D2 open 7 > D1 ext 6 > D1 open 5 > I2 ext 4 > I2 open 3 > I1 ext 2 > I1 open 1
            int code = mx == mv ? 9 : d2x == mv ? 8 : d2o == mv ? 7 : d1x == mv ? 6 : d1o == mv ? 5 : i2x == mv ? 4 : i2o == mv ? 3 : i1x == mv ? 2 : 1;
            // ... and in a gap state: extend wins a tie with open
            code |= (i1x >= i1o ? 16 : 0) | (i2x >= i2o ? 32 : 0) | (d1x >= d1o ? 64 : 0) | (d2x >= d2o ? 128 : 0);
            if (i1 < 0) i1 = WF_NULL;
            if (i2 < 0) i2 = WF_NULL;
            if (d1 < 0) d1 = WF_NULL;
            if (d2 < 0) d2 = WF_NULL;
            if (mv < 0 || mv > tlen || mv - k > plen || mv - k < 0) mv = WF_NULL; // out of the DP matrix (the maximum is nulled, as WFA2 does)
            else {
                mv += ext_run(q, mv - k, mv);
                if (k == k_end && mv >= tlen) done = 1;
            }
            Mw[c] = mv; I1w[c] = i1; I2w[c] = i2; D1w[c] = d1; D2w[c] = d2;
            chrow[k - klo] = (uint8_t)code;
        }
        return done;
    };

    WfaOut out; out.status = LCD_OK; out.score = -1; out.n_cigar = 0; out.aln_len = 0; out.offsets = 0;
    unsigned long long n_off = 0;
    for (int i = tid; i < ring_n; i += NT) ring[i] = WF_NULL;
    __syncthreads();
    int s = 0;
    uint64_t choff = 0;
    for (;;) {
        if (s > 0 && s % B == 0) { // block start: snapshot of the value ring (rows < s), decision bytes restart
            int *cp = (int *)(ws + L.ckpt + (uint64_t)(s / B - 1) * L.ring_bytes);
            for (int i = tid; i < ring_n; i += NT) cp[i] = ring[i];
            choff = 0;
            __syncthreads();
        }
        int done = compute_row(s, ch + choff);
        n_off += 5ull * wid(s);
        done = __syncthreads_or(done);
        if (done) break;
        choff += wid(s);
        ++s;
        if (s > s_cap) { out.status = LCD_ERR_WF; break; }
    }
    const int S = s;
    if (out.status == LCD_OK) {
        // ---- backtrace over the decision bytes (wave 0), re-computing earlier blocks from their snapshots (all threads) ----
        int blk_s0 = (S / B) * B;
        if (tid == 0) { sh_s = S; sh_k = k_end; sh_type = 0; sh_nrec = 0; sh_runop = 0; sh_runlen = 0; sh_err = 0; }
        __syncthreads();
        for (;;) {
            if (wave == 0) {
                int ws_ = sh_s, wk = sh_k, wt = sh_type, nrec = sh_nrec, runop = sh_runop, runlen = sh_runlen, err = 0;
                const uint64_t cum0 = wfa_cum(blk_s0, plen, tlen);
                auto cell = [&](int s2, int k2) -> int { return ch[wfa_cum(s2, plen, tlen) - cum0 + (uint64_t)(k2 + imin(s2, plen))]; };
                auto push = [&](int op, int len) { if (nrec >= L.ev_cap) { err = 1; return; } if (lane == 0) rec[nrec] = make_int2(op, len); ++nrec; };
                while (ws_ > 0 && ws_ >= blk_s0 && !err) {
                    if (wt == 0) {
                        const int code = cell(ws_, wk) & 15;
                        if (code == 0) { err = 1; break; }
                        if (code == 9) { push(OP_X, 1); ws_ -= x; continue; }
                        const bool ins = code <= 4, p2 = ((code - 1) & 2) != 0, isx = (code & 1) == 0;
                        runop = ins ? OP_I : OP_D; runlen = 1;
                        wk += ins ? -1 : 1;
                        ws_ -= isx ? (p2 ? e2 : e1) : (p2 ? o2 + e2 : o1 + e1);
                        if (isx) wt = (ins ? 0 : 2) + (p2 ? 2 : 1); else push(runop, runlen);
                    } else { // inside a gap: up to 64 extend steps at once
                        const int e = (wt & 1) ? e1 : e2, o = (wt & 1) ? o1 : o2, dk = wt <= 2 ? -1 : 1;
                        const int bit = wt == 1 ? 16 : wt == 2 ? 32 : wt == 3 ? 64 : 128;
                        const int sj = ws_ - lane * e, kj = wk + lane * dk;
                        const bool valid = sj >= blk_s0 && sj > 0;
                        const bool isx = valid && (cell(sj, kj) & bit) != 0;
                        const unsigned long long bx = __ballot(isx);
                        const int n_valid = __popcll(__ballot(valid));
                        const int n_ext = bx == ~0ull ? 64 : __builtin_ctzll(~bx);
                        if (n_ext < n_valid) { runlen += n_ext + 1; ws_ -= n_ext * e + o + e; wk += (n_ext + 1) * dk; push(runop, runlen); wt = 0; }
                        else { runlen += n_valid; ws_ -= n_valid * e; wk += n_valid * dk; }
                    }
                }
                if (ws_ <= 0 && (ws_ < 0 || wt != 0 || wk != 0)) err = 1; // the path ends on M of score 0
                if (lane == 0) { sh_s = ws_; sh_k = wk; sh_type = wt; sh_nrec = nrec; sh_runop = runop; sh_runlen = runlen; sh_err = err; }
            }
            __syncthreads();
            if (sh_err || sh_s <= 0) break;
            // the path continues in the block before: restore the ring as it was at that block's start and re-compute its decision bytes
            const int b = blk_s0 / B - 1;
            if (b < 0) { if (tid == 0) sh_err = 1; __syncthreads(); break; }
            if (b == 0) { for (int i = tid; i < ring_n; i += NT) ring[i] = WF_NULL; }
            else { const int *cp = (const int *)(ws + L.ckpt + (uint64_t)(b - 1) * L.ring_bytes); for (int i = tid; i < ring_n; i += NT) ring[i] = cp[i]; }
            __syncthreads();
            uint64_t co = 0;
            for (int s2 = b * B; s2 < b * B + B; ++s2) { compute_row(s2, ch + co); co += wid(s2); n_off += 5ull * wid(s2); __syncthreads(); }
            blk_s0 = b * B;
        }
        if (sh_err) out.status = LCD_ERR_BACKTRACK;
    }
    if (out.status == LCD_OK && wave == 0) {
        out.score = S;
        // ---- forward replay: the matches between the recorded events are the extensions of the forward pass ----
        const int nrec = sh_nrec;
        int v = 0, h = 0, nruns = 0, curop = OP_M, curlen = 0;
        auto add = [&](int op, int len) {
            if (len <= 0) return;
            if (op == curop) { curlen += len; return; }
            if (curlen > 0) { if (lane == 0) runs[nruns] = make_int2(curop, curlen); ++nruns; }
            curop = op; curlen = len;
        };
        { const int m = coop_ext(q, v, h, lane); v += m; h += m; add(OP_M, m); }
        for (int r0 = nrec - 1; r0 >= 0; r0 -= 64) {
            const int ri = r0 - lane;
            const int2 mine = ri >= 0 ? rec[ri] : make_int2(0, 0);
            const int cnt = imin(64, r0 + 1);
            for (int j = 0; j < cnt; ++j) {
                const int op = __shfl(mine.x, j), len = __shfl(mine.y, j);
                if (op == OP_X) { ++v; ++h; } else if (op == OP_I) h += len; else v += len;
                add(op, len);
                const int m = coop_ext(q, v, h, lane); v += m; h += m; add(OP_M, m);
            }
        }
        if (curlen > 0) { if (lane == 0) runs[nruns] = make_int2(curop, curlen); ++nruns; }
        if (v != plen || h != tlen) out.status = LCD_ERR_BACKTRACK;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (out.status == LCD_OK) {
            // runs are in working order = start->end of the (possibly reversed) pair; left alignment reads them backwards (src/align.c:433-453)
            uint8_t *ob = outpool + jb.out_off;
            const int maxl = plen + tlen + 1;
            uint64_t o = 0;
            const bool rev = q.rev;
            if (jb.want & 1) {
                uint32_t *cig = (uint32_t *)ob;
                for (int i = lane; i < nruns; i += 64) {
                    const int2 r = runs[rev ? nruns - 1 - i : i];
                    cig[i] = ((uint32_t)r.y << 4) | (r.x == OP_M ? 7u : r.x == OP_X ? 8u : r.x == OP_I ? 1u : 2u);
                }
                out.n_cigar = nruns;
                o = lcd_align_up((uint64_t)maxl * 4, 16);
            }
            if (jb.want & 2) {
                uint8_t *pa = ob + o, *ta = pa + maxl;
                int col = 0, pp = 0, tp = 0;
                for (int i0 = 0; i0 < nruns; i0 += 64) {
                    const int i = i0 + lane;
                    const int2 r = i < nruns ? runs[rev ? nruns - 1 - i : i] : make_int2(0, 0);
                    const int cnt = imin(64, nruns - i0);
                    for (int j = 0; j < cnt; ++j) {
                        const int op = __shfl(r.x, j), len = __shfl(r.y, j);
                        for (int c = lane; c < len; c += 64) {
                            pa[col + c] = op == OP_I ? (uint8_t)LCD_GAP : q.pat[pp + c];
                            ta[col + c] = op == OP_D ? (uint8_t)LCD_GAP : q.txt[tp + c];
                        }
                        col += len; if (op != OP_I) pp += len; if (op != OP_D) tp += len;
                    }
                }
                out.aln_len = col;
            }
        }
    }
    out.offsets = n_off;
    if (tid == 0) outs[jid] = out;
}

void lcd_launch_wfa(const WfaJob *jobs, const uint8_t *pool, uint8_t *arena, uint8_t *outpool, WfaOut *outs, LcdScoring sc, int n_jobs, int lds_bytes,
                    hipStream_t stream, int wide) {
    if (n_jobs <= 0) return;
    if (lds_bytes > 0) {
        static std::once_flag attr_once[16]; // (function attributes are per device)
        int dev = 0; if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        std::call_once(attr_once[dev], [] { hipFuncSetAttribute((const void *)lcd_wfa_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); });
        hipLaunchKernelGGL((lcd_wfa_kernel<64, true>), dim3(n_jobs), dim3(64), lds_bytes, stream, jobs, pool, arena, outpool, outs, sc, n_jobs);
    } else if (wide)
        hipLaunchKernelGGL((lcd_wfa_kernel<256, false, true>), dim3(n_jobs), dim3(256), 0, stream, jobs, pool, arena, outpool, outs, sc, n_jobs);
    else
        hipLaunchKernelGGL((lcd_wfa_kernel<256, false>), dim3(n_jobs), dim3(256), 0, stream, jobs, pool, arena, outpool, outs, sc, n_jobs);
}
