// lcd_types.h -- POD descriptors shared by the host orchestration (lcd_host.cpp) and the gfx950 kernels.
// Names follow the reference's domain (regions, reads, chains = one abPOA graph build, cons, msa).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LCD_HD __host__ __device__
#else
#define LCD_HD
#endif

#define LCD_NEG (-(1 << 29))
#define LCD_GAP 5

// cover flags, src/align.h:6-18
#define LCD_RIGHT_GAP 0x1
#define LCD_LEFT_GAP 0x2
#define LCD_RIGHT_COVER 0x4
#define LCD_LEFT_COVER 0x8
#define LCD_IS_BOTH_COVER(c) (((c)&LCD_LEFT_COVER) && ((c)&LCD_RIGHT_COVER))
#define LCD_IS_LEFT_COVER(c) (((c)&LCD_LEFT_COVER) && ((c)&LCD_RIGHT_COVER) == 0)
#define LCD_IS_LEFT_GAP(c) ((c)&LCD_LEFT_GAP)
#define LCD_IS_RIGHT_COVER(c) (((c)&LCD_LEFT_COVER) == 0 && ((c)&LCD_RIGHT_COVER))
#define LCD_IS_RIGHT_GAP(c) ((c)&LCD_RIGHT_GAP)
#define LCD_IS_NOT_COVER(c) (((c)&LCD_LEFT_COVER) == 0 && ((c)&LCD_RIGHT_COVER) == 0)

// scoring, src/align.h:21-26 via call_var_opt_t
struct LcdScoring {
    int match, mismatch, o1, e1, o2, e2;
    int wd_s; // watchdog of the POA chain kernel: a chain that has not finished after this many seconds ends with LCD_ERR_WATCHDOG (0: 30 s; env LCD_WATCHDOG_S)
    int dbg; // test switches (env LCD_DBG; 0 in production): 8 = force the generic rows of the POA kernel, 16 = a certified-band read that outgrows the window ends its
             // chain (LCD_ERR_CERT, re-run with full rows) instead of taking the generic rows (tests/test_gpu_kernels.py)
};

// status codes written by kernels (0 = ok). Anything else makes the host fail loudly or retry with a bigger arena.
enum { LCD_OK = 0, LCD_ERR_CELLS = 1, LCD_ERR_NODES = 2, LCD_ERR_EDGES = 3, LCD_ERR_BACKTRACK = 4, LCD_ERR_WF = 5, LCD_ERR_TOPO = 6, LCD_ERR_SYNC = 7, LCD_ERR_LDS = 8,
       LCD_ERR_CERT = 9 /* a K2 chain's certified band did not fit the single-wavefront window: the host re-runs the chain with full rows */,
       LCD_ERR_WATCHDOG = 10 /* a POA chain ran past its deadline (LcdScoring.wd_s): a loop that does not end is a bug, and this turns it into a loud error instead of a hung GPU */, LCD_FALLBACK = 100 /* internal: take the generic rows */ };

// ---------------- POA chain (one graph build = one abpoa_t life, src/align.c:762 / :872) ----------------
struct PoaRead {
    uint64_t seq_off;  // into the device pool (1 B/base codes 0-4)
    int len;
    int skip;          // 1: rejected by the anchor step (src/align.c:796 `continue`)
    int ref_beg, ref_end;   // 1-based anchors on the backbone read (collect_partial_aln_beg_end); beg_id=ref_beg+1
    int read_beg, read_end; // 1-based anchors on this read
};

// a read-table entry the anchor stage narrowed (run_many_once: the table goes up once, before the anchor stage; the anchored reads are patched after it)
struct ReadPatch { uint32_t idx, pad_; PoaRead r; };

struct PoaChain {
    int n_reads;
    int read0;        // first PoaRead of this chain
    int mode;         // 0: K1 sub-graph incremental, wb=10 wf=0.01, 1 consensus; 1: K2 unbanded, <=2 consensus
    int node_cap, edge_cap, rid_words, max_len;
    int threads;      // workgroup size class: 64 / 256 / 1024 (by DP row width)
    int wmax;         // columns per LDS ring slot
    int lds_words;    // dynamic LDS of the launch this chain is in (ring + query cache, re-used by the re-sort)
    int spill_x;      // DP region = cell_cap x (1 code plane + ord_x ordinal planes + spill_x of spilled value rows) bytes; spill_x 2 -> ord_x 1 (clean
                      // reads: one row in four has >= 2 usable predecessors), spill_x > 2 -> ord_x 4 (K2 chains of noisy reads: nearly every row has)
    uint32_t min_w;   // (int)(n*min_af) clipped below at 2 (cluster threshold)
    uint64_t cell_cap;
    uint64_t ws_off;  // byte offset of this chain's arena -- or, with slot_flags != 0, of its pool of arena SLOTS
    uint64_t out_off; // byte offset in the chain-output pool: cons[2][node_cap] + msa[(n_reads+2)][node_cap] + clu[2][n_reads] ints
    // Arena slots (lcd_host.cpp run_many_once): a chain's work arena is only live while its workgroup is resident, and a CU holds at most per_cu
    // workgroups of a launch.  Launches with more chains than the chip can hold share a pool of n_slots = (CUs x per_cu) slots of slot_bytes each; a
    // starting workgroup claims one with a compare-and-swap on flags[] (its own CU's per_cu slots first) and releases it when the chain is done.
    uint64_t slot_flags;  // 0: private arena at ws_off; else device address of int flags[n_slots] (0 free, 1 taken)
    uint64_t slot_bytes;
    uint64_t cu_rank;     // device address of int[4096]: raw (XCC, SE, SH, CU) id -> compact CU index, -1 unknown; 0: none
    int n_slots, per_cu;
    int solo;             // 1: a long single-wavefront chain in a 256-thread workgroup -- wavefront 0 runs the rows (align_lean), all four the per-read graph phases
    int ring16, pad3_;    // ring16: the LDS ring of this chain holds 16-bit values (certified-band K2 chains of the single-wavefront class: half the pool, lcd_host.cpp chain_class)
    int cert, ring_k;     // ring_k: ring slots of the single-wavefront class's windowed rows (a power of two >= 2, 0 = 2; the other classes: 2), lcd_host.cpp chain_class;  cert 1: K2 chain in the single-wavefront class, rows restricted to the certified band (poa_kernel.hip align_certified)
};

struct PoaChainOut {
    int status;
    int n_cons;
    int cons_len[2];
    int msa_len;
    int clu_n[2];
    int n_node, n_edge;
    int n_aligned_reads;
    unsigned long long cells;          // DP cells computed
    unsigned long long cells_alg;      // DP cells of the reference's algorithm (K1: the adaptive band, K2: full rows) -- SURVEY 8d's unit; == cells unless a certified band was used
    unsigned long long aligned_bases;  // POA-aligned bases (BASELINE metric)
    unsigned long long t_total, t_dp, t_bt, t_graph, t_out, t_sub; // shader-clock ticks per phase (profiling aid)
    unsigned long long rt_begin, rt_end; unsigned hw_id, xcc_id;    // placement probe: s_memrealtime (100 MHz) at start / end, HW_ID, XCC_ID
    unsigned long long t_plan, t_poll;                             // inside t_dp: plan-window refreshes / mailbox polls of one wavefront (unbanded rows)
    unsigned long long t_bp, t_add, t_sort, t_setup;               // row plan; graph update; re-sort (inside t_graph, with t_add); query staging + first row of the lean rows (profiling aid)
};

// Spare DP memory of a launch set (lcd_host.cpp round loop): a chain whose DP region is too small for the read at hand (LCD_ERR_CELLS) takes a region four
// times larger from here and repeats THAT read (poa_kernel.hip grow_dp_region) instead of ending; only when the pool is exhausted does the chain come back to
// the host to be re-run from its first read with larger estimates.  Bump allocation, reset by the host between rounds.
struct PoaSpare { unsigned long long used, cap, base; unsigned n_grown, n_refused; };

// arena layout (byte offsets relative to ws_off); identical on host and device
struct PoaLayout {
    uint64_t H, E1, E2;                                  // DP region of 4*cell_cap bytes (codes | ordinals | spilled rows, or H/E1/E2 planes)
    uint64_t rbeg, rend, roff, ooff, spoff;              // int32/int32/uint32 x3 [node_cap], by topological index
    uint64_t mpl, mpr;                                   // int32[node_cap]: leftmost/rightmost row-max column, by topological index
    uint64_t idx2node, node2idx, remain, deg, queue;     // int32[node_cap]
    uint64_t n_out_head, n_out_tail, n_in_head, n_in_tail, n_nin, n_aligned;  // int32[node_cap]
    uint64_t e_from, e_to, e_w, e_next_out, e_next_in;   // int32[edge_cap]
    uint64_t rid;                                        // uint64[edge_cap*rid_words]
    uint64_t cig_node, cig_qpos;                         // int32[max_len+1]
    uint64_t n_base, imap;                               // uint8[node_cap]
    uint64_t het, clu, nclu, prof;                       // int32[node_cap], int32[n_reads] x2, uint8[2*node_cap]
    uint64_t aa_node, aa_flag, aa_eid;                   // int32[max_len+2] x3: per-cigar-entry scratch of the parallel graph update
    uint64_t tb;                                         // int32[4*node_cap]: column-tile boundaries of the unbanded rows (poa_kernel.hip align_unbanded)
    uint64_t cert;                                       // int32[7*node_cap]: path-length ranges / bonus sums per node and the rows' certified column intervals (cert chains only)
    uint64_t pl_start, pl_pidx, pl_bonus, pl_rem, pl_base; // row plan: int32[node_cap+4], int32[edge_cap] x2, int32[node_cap], u8[node_cap]
    uint64_t e_slot;                                     // int32[edge_cap]: the row-plan entry an edge has (-1: none): a read that only adds weight patches the plan instead of rebuilding it
    uint64_t total;
};

static inline LCD_HD uint64_t lcd_align_up(uint64_t x, uint64_t a) { return (x + a - 1) / a * a; }

static inline LCD_HD PoaLayout poa_layout(int node_cap, int edge_cap, int rid_words, int max_len, uint64_t cell_cap, int n_reads, int spill_x = 2, int cert = 0) {
    PoaLayout L;
    uint64_t o = 0;
#define LCD_TAKE(field, bytes) do { L.field = o; o = lcd_align_up(o + (uint64_t)(bytes), 16); } while (0)
    LCD_TAKE(H, lcd_align_up(cell_cap, 16) * (uint64_t)(spill_x > 2 ? 1 + 4 + spill_x : 4) + 64); L.E1 = L.E2 = L.H; // DP region, partitioned by the kernel (poa_kernel.hip prologue)
    LCD_TAKE(rbeg, (uint64_t)node_cap * 4); LCD_TAKE(rend, (uint64_t)node_cap * 4); LCD_TAKE(roff, (uint64_t)node_cap * 4);
    LCD_TAKE(ooff, (uint64_t)node_cap * 4); LCD_TAKE(spoff, (uint64_t)node_cap * 4);
    LCD_TAKE(mpl, (uint64_t)node_cap * 4); LCD_TAKE(mpr, (uint64_t)node_cap * 4);
    LCD_TAKE(idx2node, (uint64_t)node_cap * 4); LCD_TAKE(node2idx, (uint64_t)node_cap * 4); LCD_TAKE(remain, (uint64_t)node_cap * 4);
    LCD_TAKE(deg, (uint64_t)node_cap * 4); LCD_TAKE(queue, (uint64_t)node_cap * 4);
    LCD_TAKE(n_out_head, (uint64_t)node_cap * 4); LCD_TAKE(n_out_tail, (uint64_t)node_cap * 4);
    LCD_TAKE(n_in_head, (uint64_t)node_cap * 4); LCD_TAKE(n_in_tail, (uint64_t)node_cap * 4);
    LCD_TAKE(n_nin, (uint64_t)node_cap * 4); LCD_TAKE(n_aligned, (uint64_t)node_cap * 4);
    LCD_TAKE(e_from, (uint64_t)edge_cap * 4); LCD_TAKE(e_to, (uint64_t)edge_cap * 4); LCD_TAKE(e_w, (uint64_t)edge_cap * 4);
    LCD_TAKE(e_next_out, (uint64_t)edge_cap * 4); LCD_TAKE(e_next_in, (uint64_t)edge_cap * 4);
    LCD_TAKE(rid, (uint64_t)edge_cap * rid_words * 8);
    LCD_TAKE(cig_node, (uint64_t)(max_len + 1) * 4); LCD_TAKE(cig_qpos, (uint64_t)(max_len + 1) * 4);
    LCD_TAKE(n_base, (uint64_t)node_cap); LCD_TAKE(imap, (uint64_t)node_cap);
    LCD_TAKE(het, (uint64_t)node_cap * 4); LCD_TAKE(clu, (uint64_t)n_reads * 4); LCD_TAKE(nclu, (uint64_t)n_reads * 4);
    LCD_TAKE(prof, (uint64_t)node_cap * 2);
    LCD_TAKE(pl_start, (uint64_t)(node_cap + 4) * 4); LCD_TAKE(pl_pidx, (uint64_t)edge_cap * 4); LCD_TAKE(pl_bonus, (uint64_t)edge_cap * 4);
    LCD_TAKE(pl_rem, (uint64_t)node_cap * 4); LCD_TAKE(pl_base, (uint64_t)node_cap);
    LCD_TAKE(e_slot, (uint64_t)edge_cap * 4);
    LCD_TAKE(aa_node, (uint64_t)(max_len + 2) * 4); LCD_TAKE(aa_flag, (uint64_t)(max_len + 2) * 4); LCD_TAKE(aa_eid, (uint64_t)(max_len + 2) * 4);
    LCD_TAKE(tb, max_len + 2 > 4096 && cert != 1 ? (uint64_t)node_cap * 16 : 16); // (only reads longer than one 4 096-column tile use it)
    LCD_TAKE(cert, cert ? (uint64_t)node_cap * 28 : 16);
#undef LCD_TAKE
    L.total = lcd_align_up(o, 256);
    return L;
}

// chain-output pool layout: cons[2][node_cap] u8, msa[(n_reads+2)][node_cap] u8, clu_ids[2][n_reads] i32
static inline LCD_HD uint64_t poa_out_bytes(int node_cap, int n_reads) {
    return lcd_align_up((uint64_t)(n_reads + 4) * node_cap, 16) + lcd_align_up((uint64_t)2 * n_reads * 4, 16);
}

// ---------------- WFA job (K3, src/align.c:374-460, heuristic none, affine-2p) ----------------
// Bounded-memory wavefront alignment (wfa_kernel.hip).  Wavefront VALUES live in a ring of the last max(x, o1+e1, o2+e2) + 1 scores (LDS for the
// small jobs, HBM for wide fronts); what is kept per (score, diagonal) is ONE byte of backtrace decisions.  Scores are cut into blocks of
// `blk_rows`; only the current block's decision bytes are kept, plus a snapshot of the value ring at every block start (`n_ckpt` of them), so the
// backtrace re-computes a block when it walks into it: memory ~ blk_rows x width + n_ckpt x ring instead of 20 B x score^2.
struct WfaJob {
    uint64_t p_off, t_off; // pattern / text byte offsets in the device pool (inputs, or a chain's consensus)
    int plen, tlen;
    int gap_aln;           // 1 = left (reverse both, reverse outputs), 2 = right
    int want;              // bit0: cigar, bit1: aligned strings
    int s_cap;             // max score the arena can hold = blk_rows * (n_ckpt + 1) - 1
    int blk_rows, n_ckpt;  // decision-byte block (scores per block), number of ring snapshots
    int lds;               // 1: value ring in LDS (single block, one wavefront per job); 0: ring in HBM (256 threads per job)
    uint64_t ws_off;       // work arena (byte offset), laid out by wfa_layout
    uint64_t ws_bytes;
    uint64_t out_off;      // output: cigar uint32[plen+tlen+1] | or rows uint8[2*(plen+tlen+1)]
};
struct WfaOut {
    int status, score, n_cigar, aln_len;
    unsigned long long offsets; // wavefront offsets computed (K3 algorithmic unit), re-computed blocks included
};
// decision bytes of scores [0, s): the row of score s' holds the diagonals [-min(s', plen), min(s', tlen)] (|k| <= s' because every gap column costs >= 1)
static inline LCD_HD uint64_t wfa_cum_half(long long s, long long L) { return s - 1 <= L ? (uint64_t)(s * (s - 1) / 2) : (uint64_t)(L * (L + 1) / 2 + (s - 1 - L) * L); }
static inline LCD_HD uint64_t wfa_cum(long long s, long long plen, long long tlen) { return s <= 0 ? 0 : wfa_cum_half(s, plen) + wfa_cum_half(s, tlen) + (uint64_t)s; }
struct WfaLayout {
    int rm, r1, r2, rows;      // ring depth of M, of I1/D1, of I2/D2; rows = rm + 2*r1 + 2*r2
    int w_cap, ev_cap;         // ring columns (widest row + 2 guard columns); capacity of the backtrace record list
    uint64_t ring_bytes, blk_bytes;
    uint64_t rec, runs, ring, ckpt, choice, total; // byte offsets relative to ws_off
};
static inline LCD_HD WfaLayout wfa_layout(int plen, int tlen, int s_cap, int blk_rows, int n_ckpt, int lds, int x, int o1, int e1, int o2, int e2) {
    WfaLayout L;
    int d = x; if (o1 + e1 > d) d = o1 + e1; if (o2 + e2 > d) d = o2 + e2;
    L.rm = d + 1; L.r1 = e1 + 1; L.r2 = e2 + 1; L.rows = L.rm + 2 * L.r1 + 2 * L.r2;
    const long long wl = s_cap < plen ? s_cap : plen, wr = s_cap < tlen ? s_cap : tlen;
    L.w_cap = (int)(wl + wr + 3);
    int cmin = x; if (o1 + e1 < cmin) cmin = o1 + e1; if (o2 + e2 < cmin) cmin = o2 + e2; if (cmin < 1) cmin = 1;
    long long ev = (long long)s_cap / cmin + 4; if (ev > (long long)plen + tlen + 4) ev = (long long)plen + tlen + 4;
    L.ev_cap = (int)ev;
    L.ring_bytes = lcd_align_up((uint64_t)L.rows * L.w_cap * 4, 16);
    L.blk_bytes = n_ckpt == 0 ? wfa_cum((long long)s_cap + 1, plen, tlen) : (uint64_t)blk_rows * (uint64_t)(L.w_cap - 2);
    uint64_t o = 0;
    L.rec = o; o = lcd_align_up(o + (uint64_t)L.ev_cap * 8, 16);
    L.runs = o; o = lcd_align_up(o + ((uint64_t)2 * L.ev_cap + 4) * 8, 16);
    L.ring = o; if (!lds) o += L.ring_bytes;
    L.ckpt = o; o += (uint64_t)n_ckpt * L.ring_bytes;
    L.choice = o; o = lcd_align_up(o + L.blk_bytes + 16, 256);
    L.total = o;
    return L;
}

// ---------------- edlib NW job (K4, src/align.c:222-232) ----------------
// K4's stored columns (P, M: u64, score: i32 per (column, 64-row block)): edlib keeps them when 20 * blocks * columns + 8 * columns < 1 MiB (edlib.cpp:1188) and splits the
// problem otherwise (Hirschberg); a sub-problem is never larger than the pair, so the pair's own blocks x columns -- capped by that rule -- is all the room a job needs
// (until round 4 every pair took the cap: 1 MiB x 17 000 pairs = 18 GB of workspace for a 20-batch submission, 56 KB per pair were used)
static inline LCD_HD uint64_t ed_tb_cap(int qlen, int tlen) {
    const uint64_t need = (uint64_t)((qlen + 63) >> 6) * (uint64_t)(tlen > 0 ? tlen : 0) + 16;
    return (need < 52432 ? need : 52432) + 3 & ~(uint64_t)3;
}
struct GatherJob { uint64_t src, dst; uint32_t bytes, pad_; };   // strings_kernel.hip lcd_gather_kernel
// collect_aln_beg_end (src/align.c:630-663) on the device: reference / query bases consumed up to the END of the last '=' run (a left-to-right anchor) and from the
// START of the first '=' run to the end (a right-to-left one) of a BAM-style CIGAR -- 20 bytes per anchor job come back instead of the CIGARs themselves
struct AnchorEndsJob { uint64_t cigar; int n_cigar, pad_; };
struct AnchorEndsOut { int has_eq, pre_r, pre_q, suf_r, suf_q; };
// ---------------- BAM records in the inflated stream (bam_kernel.hip; SURVEY 8f f3 on the device) ----------------
struct BamWalkJob { uint64_t stream, ubeg, uend, usize, descs; long long reg_end; int tid, cap; };   // [ubeg, uend): a .bai chunk as offsets of the inflated stream; usize: its length
struct BamRecDesc { uint64_t off; int bs, refid, pos, lseq; uint16_t flag, nc; uint8_t lname, mapq; uint16_t pad; uint32_t pad2; uint32_t pad3; }; // off: the record behind its block_size word
struct BamWalkOut { int n, status; uint64_t next; };   // status 0 range done, 1 truncated record, 2 a field runs past the record, 3 descriptor capacity, 4 stopped at the first record at / behind reg_end
struct BamStatJob { uint64_t rec; int bs, lname, nc, lseq; };
struct BamStatOut { long long rl, nd, nev, nid; uint64_t cig_src; int nc, kind; };   // kind 0 the record's own CIGAR, 1 the CG tag's, -2 placeholder without its tag; cig_src: where the operations are
struct ErrJob { uint64_t qual; int len, pad; };
struct EdJob {
    uint64_t q_off, t_off;
    int qlen, tlen;
    uint64_t ws_off, ws_bytes;
    int mode, pad_;   // pad_: where the result goes (index in the caller's order; run_edlib_stage launches the longest pairs first).  mode 0: NW (edlib_xgaps / edlib_end2end_aln / edlib_edit_distance), 1: HW = infix (edlib_infix_aln, src/align.c:256)
};
struct EdOut {
    int status, dist, xgaps, n_eq, n_xid;
    unsigned long long blocks; // Myers block-columns computed
    int start, end;            // HW mode: the target stretch [start, end] (0-based, inclusive) the path was taken on -- edlib's startLocations[0] / endLocations[0]; NW: 0 / tlen - 1
};

// ---------------- MSA row -> cons/read pairwise string (src/align.c:1029-1054 + wfa_trim_aln_str :496-562) ----------------
struct StrJob {
    uint64_t cons_off, read_off; // MSA rows in the device pool
    int msa_len, full_cover;
    uint64_t out_off;            // target row at out_off, query row at out_off + msa_len (src/align.c:1032-1033)
    uint64_t member_addr;        // != 0: the read row is row0 + (*(int *)member_addr) * row_stride (K2: k-th member of a cluster)
    uint64_t row0;
    int row_stride;
};
struct StrOut {
    int aln_len, target_beg, target_end, query_beg, query_end;
    int shift;                   // right-cover trim: rows start at +shift (the reference re-allocates, :541-549)
};

// ---------------- ref<->read string by composing ref<->cons with cons<->read (make_ref_read_aln_str, src/align.c:1056-1146) ----------------
// Two passes around one WFA stage: the scan pass finds the stretches where BOTH alignments have a gap on the consensus side (the reference
// aligns the two inserted segments with wfa_end2end_aln, :1083) and records them; the emit pass writes the rows, splicing in the WFA rows.
struct CmpJob {
    uint64_t rc_t, rc_q;   // ref<->cons rows (ref row, cons row), absolute device addresses
    uint64_t cr_t, cr_q;   // cons<->read rows (cons row, read row)
    int rc_len, cr_len;
    uint64_t seg_off;      // int4 per segment: (i, ref_len, j, read_len)
    int seg_cap, seg_first; // capacity of the segment list; index of this job's first segment in the WFA stage (emit pass)
    uint64_t out_off;      // target row at out_off, query row at out_off + rc_len + cr_len (:1060-1061)
};
struct CmpOut { int n_seg, aln_len; };
struct CmpSeg { uint64_t rows_off; int aln_len, row_stride; }; // WFA rows of one segment: pattern row at rows_off, text row at + row_stride

// ---------------- alignment strings -> candidate variants + read x variant profile (src/collect_var.c:1784-2347) ----------------
struct VarRec {
    int ref_off;                 // cand_var_t.pos - noisy_reg_beg
    int col;                     // column of the variant in the compacted cons row (alt_seq = cons[col .. col+alt_len))
    int type, ref_len, alt_len;  // BAM_CDIFF 8 / BAM_CINS 1 / BAM_CDEL 2
    int ref_base, alt_ref_base;
    int from_cons, cate, src;    // var_from_cons_idx (1 | 2 | 3), LONGCALLD_NOISY_CAND_HET/HOM_VAR, cluster whose cons row holds alt_seq
    int total_cov, alle_cov0, alle_cov1;
    int delta0, delta1;          // delta_ref_alt seen by the reads of cluster 0 / 1 when they reach this variant (:2180, :2194-2203)
    int alt_off;                 // alt_seq in the region's alt pool
};
struct VarScanJob {
    uint64_t rc_t, rc_q;         // ref row, cons row of the ref<->cons string
    int rc_len, row_cap;
    uint64_t work_off;           // compacted rows: ref at work_off, cons at work_off + row_cap
    uint64_t rec_off; int rec_cap, pad;
};
struct VarScanOut { int n_vars, n_cols; };
struct VarRegJob {
    int n_cons, pad;
    uint64_t rec[2], cons[2];    // per consensus: VarRec list, compacted cons row
    int n_rec[2], n_rows[2], str_first[2]; // reads of each cluster = StrJob/StrOut [str_first, str_first + n_rows)
    uint64_t rc_t[2], rc_q[2]; int rc_len[2];
    uint64_t out_rec, out_alt, out_prof, out_se; // merged VarRec[], alt pool, int8 profile rows x n_vars (-2 = not covered), int2 (start, end) per row
};
struct VarRegOut { int n_vars, alt_bytes; };

// ---------------- SURVEY 8(f) f2 (first part): EQX CIGAR -> digars + per-read noisy windows (src/bam_utils.c:701-842) ----------------
struct DigarRec { long long pos; int type, len, qi, is_low_qual; };   // digar1_t without the alt_seq copy
struct IvRec { long long st, en; int label, pad; };                    // one cr_add(): [st, en), label
struct DigarJob {
    uint64_t cigar_off, qual_off;   // absolute device addresses: uint32[n_cigar], uint8[qlen]
    int n_cigar, qlen;
    long long pos0;                 // bam1_core_t.pos (0-based)
    int left_pal, right_pal;        // is_ont_palindrome_clip for the left / right clip (caller-supplied)
    uint64_t digar_off, iv_off, ev_off; // outputs DigarRec[digar_cap], IvRec[iv_cap]; scratch: the event queue (16 B x ev_cap)
    int digar_cap, iv_cap, ev_cap, clip_rule; // clip_rule 0: the EQX / MD / reference paths (src/bam_utils.c:772-787), 1: the cs path (:884-888, :969-972)
};
struct RefCmpJob { uint64_t cigar_off, seq_off, out_off; int n_cigar, pad; long long pos0; }; // BAM CIGAR words, 4-bit bases, rewritten words
struct RefCmpOut { int n_ops, nd, nev, pad; };                                               // rewritten operations, digars, window events
struct DigarOut { int status, n_digar, n_iv, n_cand, rlen; };
// a read's slice of a noisy region, still 4-bit packed as in its BAM record (bam_get_seq), to be written as 1 B/base codes 0-4 (seq_nt16_int) into a batch's
// input pool on the device: lcd_host.cpp lcd_batch_add_region_from_chunk_packed / digar_kernel.hip lcd_unpack_kernel
// one (region, read) pair of collect_noisy_read_info's digar walk (digar_kernel.hip lcd_slice_kernel)
struct SliceJob { uint64_t digar_off; int n_digar, qlen; long long reg_beg, reg_end; }; // digar_off: index of the read's first DigarRec
struct SliceOut { int read_beg, read_end, cover, pad; };
struct UnpackJob { uint64_t src, dst; int first, len; }; // src: byte offset in the packed staging pool; first: 0 / 1 = the slice starts at the high / low nibble of that byte
struct DigarOpt { int min_bq, max_xgaps, win, end_clip_reg, end_clip_flank, pad; long long whole_ref_len; };

// ---------------- sdust segments (sdust_kernel.hip): one lane per segment, segments of many sequences per launch ----------------
struct SdSeg { uint64_t seq_off; int len, a, from, pad; }; // sequence (offset in the pool, length), segment start, where the lane's automaton starts

// ---------------- K5: haplotype assignment (src/assign_hap.c:473-547) ----------------
// bam_chunk_t / cand_var_t / read_var_profile_t flattened; every pointer is an absolute device address.
struct HapProb {
    int n_reads, n_vars, is_ont, n_cr, total_alle, target;
    const long long *var_pos;
    const int *var_type, *var_cate, *is_hp, *total_cov, *alle_off, *alle_covs;
    const int *start_var, *end_var, *allele_off, *alleles, *ordered, *cr_read;
    const uint8_t *is_skipped;
    int *haps; long long *phase_sets; int *n_agree_snps, *n_conflict_snps;
    long long *var_ps; int *cons; int *prof;
    // scratch
    int *valid, *vii, *het, *is_het, *n_agree, *n_conflict, *cur_cons, *flags;
};
