/* oracle/te_info.c -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
 * retrotransposon annotation of SV-size gaps, restated from the reference's text:
 *   collect_te_info            src/align.c:32-83
 *   collect_te_info_from_var   src/align.c:87-131   (through lcdo_collect_te_info_from_cons with the alt_seq as the row)
 *   collect_te_info_from_cons  src/align.c:139-163
 *   not_simple_kmer / collect_kmer / collect_rev_kmer / make_kmer_hash_tables / collect_query_kmer / collect_kmer_hist / check_te_seq
 *                              src/kmer.c:16-24, :52-75, :27-50, :78-117, :153-176, :178-192, :218-253
 *   get_bseq1, nst_nt4_table   src/seq.c:101-104, :14-31
 * The reference keeps each TE sequence's k-mers in a khash; here they are sorted arrays searched by bisection (only membership is ever asked).
 * Parity: unpinned against a reference build (src/kmer.c pulls in htslib-typed headers: unbuildable here); pinned by the hand-derived known answers of
 * tests/test_te_info.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

static int nt4(unsigned char c) {
    if (c < 4) return c;
    if (c == 'A' || c == 'a') return 0;
    if (c == 'C' || c == 'c') return 1;
    if (c == 'G' || c == 'g') return 2;
    if (c == 'T' || c == 't') return 3;
    return 4; /* ('-' is 5 in the table: equally "not a base") */
}
/* src/kmer.c:16-24, literally */
static int not_simple_kmer(uint32_t kmer, int k) {
    for (int i = 0; i < k; ++i) {
        if ((kmer & 3) != (kmer & 3) << 2 * i) return 1;
        kmer >>= 2;
    }
    return 0;
}
struct lcdo_te_lib { int k, n; uint32_t **fw, **rv; int *nfw, *nrv; };
static int cmp_u32(const void *a, const void *b) { uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }
static int uniq(uint32_t *a, int n) { if (n == 0) return 0; qsort(a, n, 4, cmp_u32); int m = 1; for (int i = 1; i < n; ++i) if (a[i] != a[m - 1]) a[m++] = a[i]; return m; }
static int has(const uint32_t *a, int n, uint32_t x) { int lo = 0, hi = n; while (lo < hi) { int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; } return lo < n && a[lo] == x; }

lcdo_te_lib_t *lcdo_te_lib_create(int n_seqs, const char *const *seqs, const int *lens, int k) {
    lcdo_te_lib_t *L = (lcdo_te_lib_t *)calloc(1, sizeof(*L));
    L->k = k; L->n = n_seqs;
    L->fw = (uint32_t **)calloc(n_seqs + 1, sizeof(uint32_t *)); L->rv = (uint32_t **)calloc(n_seqs + 1, sizeof(uint32_t *));
    L->nfw = (int *)calloc(n_seqs + 1, sizeof(int)); L->nrv = (int *)calloc(n_seqs + 1, sizeof(int));
    for (int s = 0; s < n_seqs; ++s) {
        const int len = lens[s];
        uint32_t *f = (uint32_t *)malloc((len + 1) * 4), *r = (uint32_t *)malloc((len + 1) * 4);
        int nf = 0, nr = 0;
        /* forward: every window of k valid bases, 2 bits per base, first base highest (:52-75) */
        for (int i = 0; i + k <= len; ++i) {
            uint32_t w = 0; int ok = 1;
            for (int j = 0; j < k; ++j) { int c = nt4((unsigned char)seqs[s][i + j]); if (c > 3) { ok = 0; break; } w = (w << 2) | (uint32_t)c; }
            if (!ok) continue;
            if (not_simple_kmer(w, k)) f[nf++] = w;
            /* reverse complement of the same window (:27-50 builds it incrementally: last base, complemented, highest) */
            uint32_t rc = 0;
            for (int j = k - 1; j >= 0; --j) rc = (rc << 2) | (uint32_t)(3 - nt4((unsigned char)seqs[s][i + j]));
            if (not_simple_kmer(rc, k)) r[nr++] = rc;
        }
        L->nfw[s] = uniq(f, nf); L->nrv[s] = uniq(r, nr); L->fw[s] = f; L->rv[s] = r;
    }
    return L;
}
void lcdo_te_lib_destroy(lcdo_te_lib_t *L) {
    if (!L) return;
    for (int s = 0; s < L->n; ++s) { free(L->fw[s]); free(L->rv[s]); }
    free(L->fw); free(L->rv); free(L->nfw); free(L->nrv); free(L);
}

int lcdo_check_te_seq(const lcdo_te_lib_t *L, const uint8_t *seq, int len, int *is_rev) {
    const int k = L->k;
    uint32_t *q = (uint32_t *)malloc((len / k + 2) * 4); int nq = 0;
    /* :153-176: a counter of valid bases since the last k-mer or invalid base; at k the word is taken and the counter starts again */
    uint32_t key = 0; int l = 0;
    for (int i = 0; i < len; ++i) {
        int c = nt4(seq[i]);
        if (c < 4) {
            key = (key << 2) | (uint32_t)c; l++;
            if (l == k) { uint32_t x = key & ((1u << 2 * k) - 1); if (not_simple_kmer(x, k)) q[nq++] = x; l = 0; }
        } else l = 0;
    }
    if (nq <= 0) { free(q); return -1; }
    int max_for = 0, max_rev = 0, fi = -1, ri = -1;
    for (int s = 0; s < L->n; ++s) {
        int fc = 0, rc = 0;
        for (int i = 0; i < nq; ++i) { fc += has(L->fw[s], L->nfw[s], q[i]); rc += has(L->rv[s], L->nrv[s], q[i]); }
        if (fc > max_for) max_for = fc, fi = s;
        if (rc > max_rev) max_rev = rc, ri = s;
    }
    free(q);
    if (max_for > max_rev) { *is_rev = 0; return max_for >= 3 ? fi : -1; }
    else { *is_rev = 1; return max_rev >= 3 ? ri : -1; }
}

int lcdo_collect_te_info(int min_tsd_len, int max_tsd_len, int min_polya_len, float min_polya_ratio, const lcdo_te_lib_t *lib, int var_type,
                         const uint8_t *gap_seq, const uint8_t *flank_ref_seq, int gap_len, int64_t gap_pos, uint8_t *tsd_seq, int64_t *tsd_pos1,
                         int64_t *tsd_pos2, int *tsd_polya_len, int *te_seq_i, int *te_is_rev) {
    *tsd_pos1 = -1; *tsd_pos2 = -1; *tsd_polya_len = -1; *te_seq_i = -1; *te_is_rev = 0;
    int tsd_len = 0, n_mis = 0, max_allow_mis = 1;
    for (int i = 0; i < gap_len; i++) {
        uint8_t base1 = gap_seq[i], base2 = flank_ref_seq[i];
        if (base1 == base2) tsd_len = i + 1;
        else { n_mis++; if (n_mis > max_allow_mis) break; }
        if (tsd_len > max_tsd_len) break;
    }
    int has_tsd = 0, has_polya = 0, max_search_polya_len = 20;
    if (tsd_len >= min_tsd_len && tsd_len <= max_tsd_len) {
        has_tsd = 1;
        for (int polya_len = 0, polya = 0, i = gap_len - 1; i >= 0; i--) {
            polya_len++;
            if (gap_seq[i] == 0) { polya++; if (polya_len >= min_polya_len && polya >= min_polya_ratio * polya_len) { has_polya = 1; *tsd_polya_len = polya_len; } }
            else if (polya_len > max_search_polya_len) break;
        }
        if (has_polya == 0)
            for (int polyt_len = 0, polyt = 0, i = tsd_len; i < gap_len; i++) {
                polyt_len++;
                if (gap_seq[i] == 3) { polyt++; if (polyt_len >= min_polya_len && polyt >= min_polya_ratio * polyt_len) { has_polya = 1; *tsd_polya_len = -polyt_len; } }
                else if (polyt_len > max_search_polya_len) break;
            }
    }
    if (has_tsd && has_polya) {
        if (lib && lib->n > 0) *te_seq_i = lcdo_check_te_seq(lib, gap_seq, gap_len, te_is_rev);
        for (int i = 0; i < tsd_len; i++) tsd_seq[i] = flank_ref_seq[i];
        *tsd_pos1 = gap_pos;
        *tsd_pos2 = var_type == 2 ? gap_pos + gap_len : -1;
        return tsd_len;
    }
    return 0;
}

int lcdo_collect_te_info_from_cons(int min_tsd_len, int max_tsd_len, int min_polya_len, float min_polya_ratio, const lcdo_te_lib_t *lib, const char *ref_seq,
                                   int64_t ref_beg, int64_t ref_end, int64_t gap_ref_start, int msa_gap_start, int var_type, int gap_len,
                                   const uint8_t *cons_msa_seq, uint8_t *tsd_seq, int64_t *tsd_pos1, int64_t *tsd_pos2, int *tsd_polya_len, int *te_seq_i,
                                   int *te_is_rev) {
    uint8_t *gap = (uint8_t *)malloc(gap_len + 1), *flank = (uint8_t *)malloc(gap_len + 1);
    for (int i = 0; i < gap_len; ++i) {
        int64_t p = gap_ref_start + i, p2 = gap_ref_start + i + gap_len;
        uint8_t r1 = (p < ref_beg || p > ref_end) ? 4 : (uint8_t)nt4((unsigned char)ref_seq[p - ref_beg]);
        if (var_type == 1) { gap[i] = cons_msa_seq[msa_gap_start + i]; flank[i] = r1; }
        else { gap[i] = r1; flank[i] = (p2 < ref_beg || p2 > ref_end) ? 4 : (uint8_t)nt4((unsigned char)ref_seq[p2 - ref_beg]); }
    }
    int r = lcdo_collect_te_info(min_tsd_len, max_tsd_len, min_polya_len, min_polya_ratio, lib, var_type, gap, flank, gap_len, gap_ref_start, tsd_seq, tsd_pos1,
                                 tsd_pos2, tsd_polya_len, te_seq_i, te_is_rev);
    free(gap); free(flank);
    return r;
}
