// edlib_kernel.hip -- K4: Myers bit-vector NW edit distance + edlib's exact optimal path on gfx950.
//
// Replaces edlibAlign(query, target, {k=-1, NW, TASK_PATH}) as called from src/align.c:210-254
// (edlib_xgaps / edlib_end2end_aln / edlib_edit_distance).  One 64-lane wavefront owns one pair.
// Lanes are the 64-row blocks of the query (Myers/Hyyro word = uint64, edlib/src/edlib.cpp:412-447);
// a column sweep is skewed so lane b works on target column (step - b) and receives its horizontal
// delta from lane b-1 by a wave shift -- the vertical state Pv/Mv never leaves registers.
// The path is edlib's: stored-column traceback with precedence Up -> Left -> Diagonal when
// 20*ceil(q/64)*t + 8*t < 1 MiB (edlib.cpp:1188-1190, :942-1141), else Hirschberg splits at t/2 taking the
// first query row whose left+right scores meet the optimum (edlib.cpp:1231-1396).  The Ukkonen band and
// k-doubling of edlib are omitted on purpose: every cell on an optimal path is exact inside the band, so
// the unbanded matrix yields the same path (proved in oracle/edlib_nw.c, which is pinned to real edlib).
// Only the counters longcallD consumes are produced: distance, #mismatch + #gap-runs (xgaps), n_eq, n_xid.
#include <hip/hip_runtime.h>
#include "lcd_types.h"
#include "lcd_kernels.h"

namespace {

typedef unsigned long long Word;

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
// lane l <- lane l - 1 (lane 0: 0) as a DPP wave shift: the horizontal delta of the block above is on every step's dependency chain, and __shfl_up is a trip
// through the LDS crossbar (ds_bpermute, ~120 clk) where the DPP form is one VALU instruction.  All 64 lanes must be active.
__device__ __forceinline__ int wave_shr1(const int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }

__device__ __forceinline__ int calc_block(Word Pv, Word Mv, Word Eq, int hin, Word &PvOut, Word &MvOut) {
    const Word hinIsNeg = (Word)(hin < 0);
    const Word Xv = Eq | Mv;
    Eq |= hinIsNeg;
    const Word Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
    Word Ph = Mv | ~(Xh | Pv);
    Word Mh = Pv & Xh;
    int hout = (int)(Ph >> 63) - (int)(Mh >> 63);
    Ph <<= 1; Mh <<= 1;
    Mh |= hinIsNeg;
    Ph |= (Word)(hin > 0);
    PvOut = Mh | ~(Xv | Ph);
    MvOut = Ph & Xv;
    return hout;
}

struct Sub { // sub-problem in the coordinates of the job's query/target
    const uint8_t *q, *t;
    int qlen, tlen;
};

// One NW pass over `ncols` target columns.  rev: walk both strings from their ends.
// STORE: keep P/M/score of every (column, block) for the traceback (caller guarantees it fits).
// COLS: write the scores of the last column for every query row to col_out.
// (compile-time flags, not nullable pointers: hipcc folded the `if (P)` form of this test away.)
// Returns (lane-uniform) the score at (qlen-1, ncols-1).
template <bool STORE, bool COLS>
__device__ __forceinline__ int myers_pass(const Sub &sp, int ncols, bool rev, int lane, Word *P, Word *M, int *S, int *col_out, signed char *hcarry,
                                          unsigned long long *blocks_acc) {
    const int qlen = sp.qlen, nb = (qlen + 63) >> 6;
    int result = 0;
    __shared__ uint8_t tbuf[128];
    for (int tile0 = 0; tile0 < nb; tile0 += 64) {
        const int b = tile0 + lane;
        const bool act = b < nb;
        Word peq[5] = {0, 0, 0, 0, 0};
        if (act) {
            const int base = b << 6, lim = imin(64, qlen - base);
            for (int r = 0; r < lim; ++r) {
                const int qi = base + r;
                const uint8_t c = rev ? sp.q[qlen - 1 - qi] : sp.q[qi];
                const Word bit = 1ull << r;
                if (c == 0) peq[0] |= bit; else if (c == 1) peq[1] |= bit; else if (c == 2) peq[2] |= bit;
                else if (c == 3) peq[3] |= bit; else peq[4] |= bit;
            }
        }
        Word Pv = ~0ull, Mv = 0;
        int score = (b + 1) << 6, hout = 0;
        const bool more = tile0 + 64 < nb;
        const int nsteps = ncols + 63;
        // the target's characters come through a 128-byte ring in LDS, 64 columns ahead: read from HBM inside the step they were one dependent ~1 us load per
        // column -- 0.5 ms per pass of a 400-column pair, which is what the anchor stage's K4 kernel lasted (three passes per pair, 17 000 pairs in two rounds).
        // Round 6: the 64 characters (and, for the second and later 64-block tiles of a long query, the 64 horizontal deltas the tile above left in HBM -- they were
        // one dependent load per STEP on lane 0) of the NEXT block of columns are fetched a block ahead into a register: the refill is an LDS write, not a round trip
        auto tchar = [&](const int cc) -> uint8_t { return cc < ncols ? (rev ? sp.t[sp.tlen - 1 - cc] : sp.t[cc]) : (uint8_t)4; };
        uint8_t tnext = tchar(lane);
        int hcnext = (tile0 != 0 && lane < ncols) ? (int)hcarry[lane] : 0, hcreg = 0;
        for (int step = 0; step < nsteps; ++step) {
            if ((step & 63) == 0) {
                const int cc = step + lane;
                __syncthreads();
                tbuf[cc & 127] = tnext;
                hcreg = hcnext;
                __syncthreads();
                tnext = tchar(cc + 64);
                if (tile0 != 0) hcnext = cc + 64 < ncols ? (int)hcarry[cc + 64] : 0;
            }
            const int hleft = wave_shr1(hout);
            const int c = step - lane;
            const int hc0 = __builtin_amdgcn_readlane(hcreg, step & 63); // (lane 0's column is `step`: the delta it needs sits in lane step & 63 of this block's register)
            if (act && c >= 0 && c < ncols) {
                const int hin = lane == 0 ? (tile0 == 0 ? 1 : hc0) : hleft;
                const uint8_t tc = tbuf[c & 127];
                const Word Eq = tc == 0 ? peq[0] : tc == 1 ? peq[1] : tc == 2 ? peq[2] : tc == 3 ? peq[3] : peq[4];
                hout = calc_block(Pv, Mv, Eq, hin, Pv, Mv);
                score += hout;
                if (STORE) { const size_t o = (size_t)c * nb + b; P[o] = Pv; M[o] = Mv; S[o] = score; }
                if (more && lane == 63) hcarry[ncols + c] = (signed char)hout;
            }
        }
        __syncthreads();
        if (more) { // lane 63's horizontal deltas (written to the upper half during the sweep) feed the next tile
            for (int c = lane; c < ncols; c += 64) hcarry[c] = hcarry[ncols + c];
            __syncthreads();
        }
        *blocks_acc += (unsigned long long)ncols * (unsigned long long)(imin(64, nb - tile0));
        // last-column scores of this tile's rows
        if (act) {
            const int base = b << 6, lim = imin(64, qlen - base);
            if (COLS)
                for (int pos = 0; pos < lim; ++pos) {
                    int v = score;
                    if (pos < 63) v += -__popcll(Pv >> (pos + 1)) + __popcll(Mv >> (pos + 1));
                    col_out[base + pos] = v;
                }
        }
        const int lastb = nb - 1;
        if (lastb >= tile0 && lastb < tile0 + 64) {
            const int pos = (qlen - 1) & 63;
            int v = score;
            if (pos < 63) v += -__popcll(Pv >> (pos + 1)) + __popcll(Mv >> (pos + 1));
            result = __shfl(v, lastb - tile0);
        }
        __syncthreads();
    }
    return result;
}

// One semi-global pass (distance only) over `ncols` target columns: the score of the LAST query row at every column, reduced to its minimum and the first / last
// column that reaches it.  HIN0 = 0: free start in the target (edlib HW, first DP row all zero); HIN0 = 1: the target stretch is anchored at its first
// character (edlib SHW, first row 0, 1, 2, ...: the reverse pass that finds an alignment's start, edlib.cpp:230-262).  rev as in myers_pass.
template <int HIN0>
__device__ __forceinline__ void sg_pass(const Sub &sp, int ncols, bool rev, int lane, signed char *hcarry, unsigned long long *blocks_acc, int *best_out, int *first_out, int *last_out) {
    const int qlen = sp.qlen, nb = (qlen + 63) >> 6;
    int best = 1 << 30, first_c = -1, last_c = -1;
    __shared__ uint8_t tbuf2[128];
    for (int tile0 = 0; tile0 < nb; tile0 += 64) {
        const int b = tile0 + lane;
        const bool act = b < nb;
        Word peq[5] = {0, 0, 0, 0, 0};
        if (act) {
            const int base = b << 6, lim = imin(64, qlen - base);
            for (int r = 0; r < lim; ++r) {
                const int qi = base + r;
                const uint8_t c = rev ? sp.q[qlen - 1 - qi] : sp.q[qi];
                const Word bit = 1ull << r;
                if (c == 0) peq[0] |= bit; else if (c == 1) peq[1] |= bit; else if (c == 2) peq[2] |= bit;
                else if (c == 3) peq[3] |= bit; else peq[4] |= bit;
            }
        }
        Word Pv = ~0ull, Mv = 0;
        int score = (b + 1) << 6, hout = 0;
        const bool more = tile0 + 64 < nb;
        const bool is_last = b == nb - 1;
        const int pos = (qlen - 1) & 63;
        const int nsteps = ncols + 63;
        auto tchar = [&](const int cc) -> uint8_t { return cc < ncols ? (rev ? sp.t[sp.tlen - 1 - cc] : sp.t[cc]) : (uint8_t)4; }; // (a block ahead, as in myers_pass)
        uint8_t tnext = tchar(lane);
        int hcnext = (tile0 != 0 && lane < ncols) ? (int)hcarry[lane] : 0, hcreg = 0;
        for (int step = 0; step < nsteps; ++step) {
            if ((step & 63) == 0) { // (the target through a 128-byte LDS ring, 64 columns ahead: myers_pass)
                const int cc = step + lane;
                __syncthreads();
                tbuf2[cc & 127] = tnext;
                hcreg = hcnext;
                __syncthreads();
                tnext = tchar(cc + 64);
                if (tile0 != 0) hcnext = cc + 64 < ncols ? (int)hcarry[cc + 64] : 0;
            }
            const int hleft = wave_shr1(hout);
            const int c = step - lane;
            const int hc0 = __builtin_amdgcn_readlane(hcreg, step & 63);
            if (act && c >= 0 && c < ncols) {
                const int hin = lane == 0 ? (tile0 == 0 ? HIN0 : hc0) : hleft;
                const uint8_t tc = tbuf2[c & 127];
                const Word Eq = tc == 0 ? peq[0] : tc == 1 ? peq[1] : tc == 2 ? peq[2] : tc == 3 ? peq[3] : peq[4];
                hout = calc_block(Pv, Mv, Eq, hin, Pv, Mv);
                score += hout;
                if (more && lane == 63) hcarry[ncols + c] = (signed char)hout;
                if (is_last) { // the last query row lives in this block: its score at this column
                    int v = score;
                    if (pos < 63) v += -__popcll(Pv >> (pos + 1)) + __popcll(Mv >> (pos + 1));
                    if (v < best) { best = v; first_c = c; last_c = c; } else if (v == best) last_c = c;
                }
            }
        }
        __syncthreads();
        if (more) {
            for (int c = lane; c < ncols; c += 64) hcarry[c] = hcarry[ncols + c];
            __syncthreads();
        }
        *blocks_acc += (unsigned long long)ncols * (unsigned long long)(imin(64, nb - tile0));
        const int lastb = nb - 1;
        if (lastb >= tile0 && lastb < tile0 + 64) { best = __shfl(best, lastb - tile0); first_c = __shfl(first_c, lastb - tile0); last_c = __shfl(last_c, lastb - tile0); }
        __syncthreads();
    }
    *best_out = best; *first_out = first_c; *last_out = last_c;
}

struct Tally {
    int n_mis, n_eq, n_ins, n_del, runs;
    int last_op; // forward-order last op of everything tallied so far (-1 none)
};

} // namespace

__global__ void __launch_bounds__(64) lcd_edlib_kernel(const EdJob *jobs, const uint8_t *pool, uint8_t *arena, EdOut *outs, int n_jobs) {
    const int jid = blockIdx.x;
    if (jid >= n_jobs) return;
    const int lane = threadIdx.x;
    const EdJob jb = jobs[jid];
    EdOut out; out.status = LCD_OK; out.dist = -1; out.xgaps = 0; out.n_eq = 0; out.n_xid = 0; out.blocks = 0; out.start = 0; out.end = jb.tlen - 1;
    const uint8_t *q = pool + jb.q_off, *t = pool + jb.t_off;
    const int qlen = jb.qlen, tlen = jb.tlen;
    if (qlen == 0 || tlen == 0) { // edlib.cpp:166-177: distance only, no alignment.  NW: the longer side; HW: the query's length (an empty query matches the empty infix), end -1
        out.dist = jb.mode == 1 ? qlen : (qlen > tlen ? qlen : tlen);
        if (jb.mode == 1) { out.start = -1; out.end = -1; }
        if (lane == 0) outs[jb.pad_] = out;
        return;
    }
    // arena: P[TB_CAP] M[TB_CAP] (u64) S[TB_CAP] (i32) | colL[qlen] colR[qlen] | hcarry[2*tlen]
    const size_t TB_CAP = (size_t)ed_tb_cap(qlen, tlen); // (the pair's own blocks x columns, at most edlib's 1 MiB rule: lcd_types.h)
    uint8_t *ws = arena + jb.ws_off;
    Word *P = (Word *)ws, *M = P + TB_CAP;
    int *S = (int *)(M + TB_CAP);
    int *colL = S + TB_CAP, *colR = colL + qlen;
    signed char *hcarry = (signed char *)(colR + qlen);
    unsigned long long blocks = 0;
    Sub top = {q, t, qlen, tlen};
    int dist, t0 = 0, tl0 = tlen;
    const bool top_leaf = jb.mode != 1 && 20ll * ((qlen + 63) >> 6) * tlen + 8ll * tlen < 1024 * 1024;
    if (jb.mode == 1) {
        // HW (infix, edlib.cpp:146-280): best score over all end positions with a free start; the FIRST end position; its start from the reversed problem on the
        // prefix that ends there (anchored, the LAST position with the best score = the longest stretch); the path is then the NW path on that stretch
        int best, endf, endl, best2, sf, sl;
        sg_pass<0>(top, tlen, false, lane, hcarry, &blocks, &best, &endf, &endl);
        if (best >= qlen) { // no end position beats "the whole query in front of the target": edlib reports end -1 (the padded last block sees that position first,
            out.dist = qlen; out.start = 0; out.end = -1;            // edlib.cpp:659-691), takes 0 as its start (:236) and aligns against an empty target: qlen insertions
            out.xgaps = 1; out.n_eq = 0; out.n_xid = qlen; out.blocks = blocks;
            if (lane == 0) outs[jb.pad_] = out;
            return;
        }
        Sub pre = {q, t, qlen, endf + 1};
        sg_pass<1>(pre, endf + 1, true, lane, hcarry, &blocks, &best2, &sf, &sl);
        if (best2 != best || endf < 0 || sl < 0) { out.status = LCD_ERR_BACKTRACK; if (lane == 0) outs[jb.pad_] = out; return; }
        dist = best; t0 = endf - sl; tl0 = sl + 1;
        out.start = t0; out.end = endf;
    } else if (top_leaf)
        dist = -1; // (the pair is one leaf of the path search: its stored-column pass below gives the distance as well -- a distance-only pass in front of it was half the kernel)
    else
        dist = myers_pass<false, false>(top, tlen, false, lane, nullptr, nullptr, nullptr, nullptr, hcarry, &blocks);
    out.dist = dist;
    __shared__ int stk[64][5]; // qoff, qlen, toff, tlen, best
    int sp = 0;
    if (lane == 0) { stk[0][0] = 0; stk[0][1] = qlen; stk[0][2] = t0; stk[0][3] = tl0; stk[0][4] = dist; }
    sp = 1;
    __syncthreads();
    Tally T = {0, 0, 0, 0, 0, -1};
    while (sp > 0 && out.status == LCD_OK) {
        --sp;
        const int qo = stk[sp][0], ql = stk[sp][1], to = stk[sp][2], tl = stk[sp][3], best = stk[sp][4];
        __syncthreads();
        if (ql == 0 || tl == 0) { // edlib.cpp:1169-1176
            const int n = ql + tl;
            if (n > 0) {
                const int op = ql == 0 ? 2 : 1;
                if (op == 2) T.n_del += n; else T.n_ins += n;
                if (T.last_op != op) T.runs++;
                T.last_op = op;
            }
            continue;
        }
        Sub s = {q + qo, t + to, ql, tl};
        const long long nb = (ql + 63) >> 6;
        const long long data_size = 20ll * nb * tl + 8ll * tl;
        if (data_size < 1024 * 1024) {
            const int leaf_score = myers_pass<true, false>(s, tl, false, lane, P, M, S, nullptr, hcarry, &blocks);
            if (top_leaf) out.dist = leaf_score;
            __syncthreads();
            // Traceback, wave-uniform (every lane walks the same path on the same scalars).  The stored columns are in HBM, and a step needs up to three cells of
            // two adjacent columns: walked by one lane with dependent loads this was ~2 us per step, 1 - 3 ms per pair -- the anchor stage's K4 kernel (9.4 ms for 17 000
            // pairs) was its slowest pair's walk.  Now the 64 columns ending at the current one are CACHED IN REGISTERS, lane l holding column c0 - l for the two row
            // blocks the walk can be in (the current one and the one above: 64 diagonal steps move 64 rows); a cell is three v_readlane's away.  A step that leaves
            // the cached columns or blocks refreshes the cache (one round trip per ~64 steps); a cell outside it (a long vertical run) is read from HBM as before.
            int r_mis = 0, r_eq = 0, r_ins = 0, r_del = 0, r_runs = 0, leaf_first = -1, leaf_last = -1;
            {
                const int NB = (int)nb;
                int c0 = -1, bA = -1;                       // cache: columns c0 - 63 .. c0, blocks bA and bA - 1
                Word cPa = 0, cMa = 0, cPb = 0, cMb = 0; int cSa = 0, cSb = 0;
                auto refresh = [&](const int r, const int c) {
                    c0 = c; bA = r >> 6;
                    const int cl = c - lane;
                    if (cl >= 0) {
                        const size_t oa = (size_t)cl * NB + bA;
                        cPa = P[oa]; cMa = M[oa]; cSa = S[oa];
                        if (bA > 0) { cPb = P[oa - 1]; cMb = M[oa - 1]; cSb = S[oa - 1]; }
                    }
                };
                auto rl64 = [&](const Word v, const int l) { return (Word)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l) | ((Word)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32); };
                auto val = [&](const int r, const int c) -> int {
                    if (r < 0) return c + 1;
                    if (c < 0) return r + 1;
                    const int b = r >> 6, l = __builtin_amdgcn_readfirstlane(c0 - c), pos = r & 63;
                    Word pv, mv; int sv;
                    if (l >= 0 && l < 64 && b == bA) { pv = rl64(cPa, l); mv = rl64(cMa, l); sv = __builtin_amdgcn_readlane(cSa, l); }
                    else if (l >= 0 && l < 64 && b == bA - 1) { pv = rl64(cPb, l); mv = rl64(cMb, l); sv = __builtin_amdgcn_readlane(cSb, l); }
                    else { const size_t o = (size_t)c * NB + b; pv = P[o]; mv = M[o]; sv = S[o]; }
                    return sv + (pos < 63 ? (-__popcll(pv >> (pos + 1)) + __popcll(mv >> (pos + 1))) : 0);
                };
                int r = __builtin_amdgcn_readfirstlane(ql - 1), c = __builtin_amdgcn_readfirstlane(tl - 1), prev = -1;
                refresh(r, c);
                int cur = val(r, c);
                // the vertical step D[r][c] - D[r-1][c] of a stored column IS its P / M bit at row r (row -1 is the boundary c + 1): the Up test and the diagonal
                // cell (= the left cell minus ITS vertical step) need one 32-bit half of a word instead of a full cell value (score + two popcounts)
                auto vbit = [&](const Word wa, const Word wb, const Word *G, const int r, const int c) -> int { // bit (r & 63) of the word at (column c, block r >> 6)
                    const int b = r >> 6, l = __builtin_amdgcn_readfirstlane(c0 - c), pos = r & 63;
                    unsigned h;
                    if (l >= 0 && l < 64 && (b == bA || b == bA - 1)) { const Word w = b == bA ? wa : wb; h = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(pos < 32 ? w : w >> 32), l); }
                    else { const Word w = G[(size_t)c * NB + b]; h = (unsigned)(pos < 32 ? w : w >> 32); }
                    return (int)((h >> (pos & 31)) & 1u);
                };
                while (r >= 0 && c >= 0) {
                    if (c - 1 < c0 - 63 || (r > 0 && ((r - 1) >> 6) < bA - 1)) refresh(r, c);
                    int op;
                    if (vbit(cPa, cPb, P, r, c)) { op = 1; --r; cur -= 1; }   // Up: D[r-1][c] + 1 == D[r][c]
                    else {
                        const int left = val(r, c - 1);
                        if (left + 1 == cur) { op = 2; --c; cur = left; }
                        else {
                            const int d = c == 0 ? val(r - 1, c - 1) : left - vbit(cPa, cPb, P, r, c - 1) + vbit(cMa, cMb, M, r, c - 1);
                            op = d == cur ? 0 : 3; --r; --c; cur = d;
                        }
                    }
                    r = __builtin_amdgcn_readfirstlane(r); c = __builtin_amdgcn_readfirstlane(c); cur = __builtin_amdgcn_readfirstlane(cur);
                    if (op == 0) r_eq++; else if (op == 3) r_mis++; else { if (op == 1) r_ins++; else r_del++; if (op != prev) r_runs++; }
                    if (leaf_last < 0) leaf_last = op;
                    leaf_first = op; prev = op;
                }
                if (c >= 0) { const int n = c + 1; r_del += n; if (prev != 2) r_runs++; if (leaf_last < 0) leaf_last = 2; leaf_first = 2; }
                if (r >= 0) { const int n = r + 1; r_ins += n; if (prev != 1) r_runs++; if (leaf_last < 0) leaf_last = 1; leaf_first = 1; }
            }
            T.n_mis += r_mis; T.n_eq += r_eq; T.n_ins += r_ins; T.n_del += r_del; T.runs += r_runs;
            if (leaf_first >= 0) {
                if ((leaf_first == 1 || leaf_first == 2) && T.last_op == leaf_first) T.runs--; // run continues across leaves
                T.last_op = leaf_last;
            }
            __syncthreads();
        } else {
            const int left_w = tl / 2, right_w = tl - left_w;
            myers_pass<false, true>(s, left_w, false, lane, nullptr, nullptr, nullptr, colL, hcarry, &blocks);
            myers_pass<false, true>(s, right_w, true, lane, nullptr, nullptr, nullptr, colR, hcarry, &blocks);
            __syncthreads();
            // first row i in [0, ql-2] with colL[i] + right[i+1] == best, right[idx] = colR[ql-1-idx]
            int cand = 1 << 30;
            for (int i = lane; i <= ql - 2; i += 64)
                if (colL[i] + colR[ql - 1 - (i + 1)] == best) { cand = i; break; }
            for (int d = 32; d >= 1; d >>= 1) cand = imin(cand, __shfl_xor(cand, d));
            int q_idx, ls, rs;
            if (cand < (1 << 30)) { q_idx = cand; ls = colL[cand]; rs = colR[ql - 1 - (cand + 1)]; }
            else if (left_w + colR[ql - 1] == best) { q_idx = -1; ls = left_w; rs = colR[ql - 1]; }
            else if (colL[ql - 1] + right_w == best) { q_idx = ql - 1; ls = colL[ql - 1]; rs = right_w; }
            else { out.status = LCD_ERR_BACKTRACK; break; }
            const int ul_h = q_idx + 1, lr_h = ql - ul_h;
            if (sp + 2 > 64) { out.status = LCD_ERR_BACKTRACK; break; }
            __syncthreads();
            if (lane == 0) { // push lower-right first so the upper-left is processed first (forward order)
                stk[sp][0] = qo + ul_h; stk[sp][1] = lr_h; stk[sp][2] = to + left_w; stk[sp][3] = right_w; stk[sp][4] = rs;
                stk[sp + 1][0] = qo; stk[sp + 1][1] = ul_h; stk[sp + 1][2] = to; stk[sp + 1][3] = left_w; stk[sp + 1][4] = ls;
            }
            sp += 2;
            __syncthreads();
        }
    }
    out.xgaps = T.n_mis + T.runs;
    out.n_eq = T.n_eq;
    out.n_xid = T.n_mis + T.n_ins + T.n_del;
    out.blocks = blocks;
    if (lane == 0) outs[jb.pad_] = out;
}

void lcd_launch_edlib(const EdJob *jobs, const uint8_t *pool, uint8_t *arena, EdOut *outs, int n_jobs, hipStream_t stream) {
    if (n_jobs <= 0) return;
    hipLaunchKernelGGL(lcd_edlib_kernel, dim3(n_jobs), dim3(64), 0, stream, jobs, pool, arena, outs, n_jobs);
}
