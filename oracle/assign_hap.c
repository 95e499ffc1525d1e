/*
 * assign_hap.c -- ORACLE (test infrastructure only; see lcd_oracle.h).
 *
 * Source-level restatement of the germline part of src/assign_hap.c (lines 16-547): two-haplotype read assignment
 * and phase-set construction (K5).  All of that source is present in /root/reference, so this follows it statement by
 * statement on flattened arrays; "parity unpinned" only because the reference binary cannot be built here.
 * cr_overlap() results are reproduced with the sorted-interval order of cgranges (src/cgranges.c:449-502 reports
 * overlapping intervals in index order of the sorted array).
 */
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

#define CLEAN_HET_SNP 0x004
#define CLEAN_HET_INDEL 0x008
#define CLEAN_HOM_VAR 0x080
#define NOISY_CAND_HET_VAR 0x100
#define NOISY_CAND_HOM_VAR 0x200
#define GERMLINE_CLEAN_CATE (CLEAN_HET_SNP | CLEAN_HET_INDEL | CLEAN_HOM_VAR)

typedef lcdo_hap_problem_t P;
#define NALLE(p, v) ((p)->alle_off[(v) + 1] - (p)->alle_off[v])
#define PROF(p, h, v, a) (p)->hap_to_alle_profile[(size_t)(h) * (p)->alle_off[(p)->n_vars] + (p)->alle_off[v] + (a)]
#define CONS(p, v, h) (p)->hap_to_cons_alle[(v) * 3 + (h)]
#define ALLELE(p, r, v) (p)->alleles[(p)->allele_off[r] + ((v) - (p)->start_var_idx[r])]

/* src/assign_hap.c:23-36 */
static int init_max_cov_allele(const P *p, int v) {
    if (p->is_ont == 1 && p->is_homopolymer_indel[v]) return -1;
    int max_cov = 0, mi = -1;
    for (int i = 0; i < NALLE(p, v); ++i)
        if (p->alle_covs[p->alle_off[v] + i] > max_cov) { max_cov = p->alle_covs[p->alle_off[v] + i]; mi = i; }
    return mi;
}

/* src/assign_hap.c:127-147 (mutates hap_to_cons_alle) */
static int read_to_cons_allele_score(P *p, int hap, int v, int cate, int allele_i) {
    int var_score = 1;
    if (cate == CLEAN_HET_SNP || cate == CLEAN_HET_INDEL) var_score = 2;
    if (CONS(p, v, hap) == -1 && CONS(p, v, 3 - hap) == -1) return 0;
    else {
        if (CONS(p, v, hap) == -1) CONS(p, v, hap) = 1 - CONS(p, v, 3 - hap);
        if (CONS(p, v, 3 - hap) == -1) CONS(p, v, 3 - hap) = 1 - CONS(p, v, hap);
    }
    if (CONS(p, v, hap) == allele_i) return var_score;
    else if (CONS(p, v, hap) == -1) return 0;
    else return -var_score;
}

/* src/assign_hap.c:151-198 */
static int init_assign_read_hap(P *p, int r, int target) {
    int hap_scores[3] = {0, 0, 0}, n_used[3] = {0, 0, 0}, agree[3] = {0, 0, 0}, conflict[3] = {0, 0, 0};
    p->n_clean_agree_snps[r] = p->n_clean_conflict_snps[r] = 0;
    for (int v = p->start_var_idx[r]; v <= p->end_var_idx[r]; ++v) {
        const int cate = p->var_cate[v];
        if ((cate & target) == 0) continue;
        if (p->is_homopolymer_indel[v] == 1 || cate == NOISY_CAND_HOM_VAR) continue;
        const int al = ALLELE(p, r, v);
        if (al < 0) continue;
        for (int hap = 1; hap <= 2; ++hap) {
            int s0 = read_to_cons_allele_score(p, hap, v, cate, al);
            if (s0 != 0) {
                if (cate != CLEAN_HOM_VAR) n_used[hap]++;
                if ((cate & GERMLINE_CLEAN_CATE) > 0 && p->var_type[v] == LCDO_CDIFF) { if (s0 > 0) agree[hap]++; else conflict[hap]++; }
            }
            if (cate != CLEAN_HOM_VAR) hap_scores[hap] += s0;
        }
    }
    int max_hap = 0, max_score = 0, min_hap = 0, min_score = 0;
    for (int hap = 1; hap <= 2; ++hap) {
        if (hap_scores[hap] > max_score) { max_hap = hap; max_score = hap_scores[hap]; }
        else if (hap_scores[hap] < min_score) { min_hap = hap; min_score = hap_scores[hap]; }
    }
    if (n_used[1] == 0 && n_used[2] == 0) return -1;
    else if (max_score == 0 && min_score == 0) return 0;
    else if (max_score > 0) { p->n_clean_agree_snps[r] = agree[max_hap]; p->n_clean_conflict_snps[r] = conflict[max_hap]; return max_hap; }
    else return 3 - min_hap;
}

/* src/assign_hap.c:244-268 */
static void update_var_hap_to_cons_alle(P *p, int v, int hap) {
    if (hap == 0) return;
    int max_cov = 0, mi = -1, total = 0;
    for (int i = 0; i < NALLE(p, v); ++i) {
        total += PROF(p, hap, v, i);
        if (PROF(p, hap, v, i) > max_cov) { max_cov = PROF(p, hap, v, i); mi = i; }
    }
    if (p->is_ont && p->is_homopolymer_indel[v] == 1 && max_cov < total * 0.67) mi = -1;
    CONS(p, v, hap) = mi;
}

/* src/assign_hap.c:270-290 */
static void update_profile_cons_by_read(P *p, int r, int hap, int target) {
    for (int v = p->start_var_idx[r]; v <= p->end_var_idx[r]; ++v) {
        if ((p->var_cate[v] & target) == 0) continue;
        const int al = ALLELE(p, r, v);
        if (al < 0) continue;
        if (hap == 0) { for (int i = 1; i <= 2; ++i) { PROF(p, i, v, al) += 1; update_var_hap_to_cons_alle(p, v, i); } }
        else { PROF(p, hap, v, al) += 1; update_var_hap_to_cons_alle(p, v, hap); }
    }
}

/* src/assign_hap.c:292-305 */
static void update_profile_by_read(P *p, int r, int hap, int target) {
    for (int v = p->start_var_idx[r]; v <= p->end_var_idx[r]; ++v) {
        if ((p->var_cate[v] & target) == 0) continue;
        const int al = ALLELE(p, r, v);
        if (al < 0) continue;
        if (hap == 0) { PROF(p, 1, v, al) += 1; PROF(p, 2, v, al) += 1; }
        else PROF(p, hap, v, al) += 1;
    }
}

/* src/assign_hap.c:307-320 */
static int check_agree_haps(const P *p, int r, int hap, int var1, int var2) {
    if (var1 < p->start_var_idx[r] || var2 > p->end_var_idx[r]) return -1;
    if (hap == 0) return -1;
    const int a1 = ALLELE(p, r, var1), a2 = ALLELE(p, r, var2);
    if (a1 < 0 || a2 < 0) return -1;
    int agree = 0, conflict = 0;
    if (CONS(p, var1, hap) == a1 && CONS(p, var2, hap) == a2) agree = 1;
    if (CONS(p, var1, hap) == a1 && CONS(p, var2, 3 - hap) == a2) conflict = 1;
    if (agree) return 1; else if (conflict) return 0; else return -1;
}

/* src/assign_hap.c:345-422 */
static int iter_update_phase_set(P *p, const int *var_idx, int n) {
    int *het = (int *)malloc((n > 0 ? n : 1) * sizeof(int)), n_het = 0;
    int *is_het = (int *)calloc(n > 0 ? n : 1, sizeof(int));
    for (int i = 0; i < n; ++i) {
        int v = var_idx[i];
        if (CONS(p, v, 1) != -1 && CONS(p, v, 2) != -1 && CONS(p, v, 1) != CONS(p, v, 2) && p->is_homopolymer_indel[v] == 0) { is_het[i] = 1; het[n_het++] = i; }
    }
    int *n_agree = (int *)calloc(n > 0 ? n : 1, sizeof(int)), *n_conflict = (int *)calloc(n > 0 ? n : 1, sizeof(int));
    for (int k = 1; k < n_het; ++k) {
        const int i = het[k], v = var_idx[i], pv = var_idx[het[k - 1]];
        /* cr_overlap(read_var_cr, pv, v+1): intervals [start_var_idx, end_var_idx+1) with start < v+1 and pv < end */
        for (int c = 0; c < p->n_cr; ++c) {
            const int r = p->cr_read[c];
            if (!(p->start_var_idx[r] < v + 1 && pv < p->end_var_idx[r] + 1)) continue;
            if (p->is_skipped[r]) continue;
            int a = check_agree_haps(p, r, p->haps[r], pv, v);
            if (a > 0) n_agree[i]++; else if (a == 0) n_conflict[i]++;
        }
    }
    int changed = 0, flip = 0; int64_t phase_set = -1;
    for (int i = 0; i < n; ++i) {
        const int v = var_idx[i];
        if (i == 0) { phase_set = p->var_type[v] == LCDO_CDIFF ? p->var_pos[v] : p->var_pos[v] - 1; p->var_phase_set[v] = phase_set; continue; }
        if (is_het[i] == 1) {
            if (n_agree[i] < 2 && n_conflict[i] < 2) phase_set = p->var_type[v] == LCDO_CDIFF ? p->var_pos[v] : p->var_pos[v] - 1;
            else if (n_conflict[i] > n_agree[i]) flip ^= 1;
            if (flip == 1) {
                changed = 1;
                for (int hap = 1; hap <= 2; ++hap) { int t = CONS(p, v, hap); CONS(p, v, hap) = CONS(p, v, 3 - hap); CONS(p, v, 3 - hap) = t; } /* two swaps == identity, :406-411 */
            }
        }
        p->var_phase_set[v] = phase_set;
    }
    free(het); free(is_het); free(n_agree); free(n_conflict);
    return changed;
}

/* src/assign_hap.c:425-467 */
static int iter_update_cons_alle(P *p, const int *var_idx, int n, int target) {
    int *cur = (int *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(int));
    for (int i = 0; i < n; ++i) for (int h = 1; h <= 2; ++h) cur[i * 3 + h] = CONS(p, var_idx[i], h);
    for (int i = 0; i < n; ++i) { /* var_init_hap_to_alle_profile :80-92 zeroes all three planes */
        const int v = var_idx[i];
        for (int h = 0; h <= 2; ++h) for (int a = 0; a < NALLE(p, v); ++a) PROF(p, h, v, a) = 0;
    }
    for (int i = 0; i < p->n_reads; ++i) {
        const int r = p->ordered_read_ids[i];
        if (p->is_skipped[r]) continue;
        int hap = (p->start_var_idx[r] < 0) ? -1 : init_assign_read_hap(p, r, target);
        if (p->start_var_idx[r] < 0) { p->n_clean_agree_snps[r] = p->n_clean_conflict_snps[r] = 0; }
        if (hap == -1) hap = 0;
        p->haps[r] = hap;
        if (p->start_var_idx[r] >= 0) update_profile_by_read(p, r, hap, target);
    }
    for (int i = 0; i < n; ++i) for (int h = 1; h <= 2; ++h) update_var_hap_to_cons_alle(p, var_idx[i], h);
    int changed = 0;
    for (int i = 0; i < n; ++i) for (int h = 1; h <= 2; ++h) if (CONS(p, var_idx[i], h) != cur[i * 3 + h]) changed = 1;
    free(cur);
    return changed;
}

/* src/assign_hap.c:473-547 */
int lcdo_assign_hap_germline(lcdo_hap_problem_t *p, int target) {
    int n = 0, *valid = (int *)malloc((p->n_vars > 0 ? p->n_vars : 1) * sizeof(int));
    uint8_t *is_valid = (uint8_t *)calloc(p->n_vars > 0 ? p->n_vars : 1, 1);
    for (int i = 0; i < p->n_vars; ++i) if (p->var_cate[i] & target) { valid[n++] = i; is_valid[i] = 1; }
    if (n == 0) { free(valid); free(is_valid); return 0; }
    for (int r = 0; r < p->n_reads; ++r) { p->haps[r] = 0; p->phase_sets[r] = -1; }            /* :16-20 */
    for (int i = 0; i < n; ++i) {                                                                /* :39-63 */
        const int v = valid[i];
        for (int h = 1; h <= 2; ++h) for (int a = 0; a < NALLE(p, v); ++a) PROF(p, h, v, a) = 0;
        CONS(p, v, 0) = init_max_cov_allele(p, v);
        const int hom = p->var_cate[v] == NOISY_CAND_HOM_VAR || p->var_cate[v] == CLEAN_HOM_VAR;
        CONS(p, v, 1) = CONS(p, v, 2) = hom ? 1 : -1;
    }
    /* select_init_var :94-125 */
    int init = -1;
    {
        int ci = -1, ii = -1, ns = -1, ni = -1, cd = 0, id = 0, nsd = 0, nid = 0;
        for (int i = 0; i < n; ++i) {
            const int v = valid[i], cate = p->var_cate[v], cov = p->total_cov[v];
            if (cate == CLEAN_HET_SNP) { if (ci == -1 || cd < cov) { ci = i; cd = cov; } }
            else if (cate == CLEAN_HET_INDEL) { if (ii == -1 || id < cov) { ii = i; id = cov; } }
            else if (cate == NOISY_CAND_HET_VAR) {
                if (p->var_type[v] == LCDO_CDIFF) { if (ns == -1 || nsd < cov) { ns = i; nsd = cov; } }
                else if (p->is_homopolymer_indel[v] == 0) { if (ni == -1 || nid < cov) { ni = i; nid = cov; } }
            }
        }
        init = ci != -1 ? ci : ii != -1 ? ii : ns != -1 ? ns : ni;
    }
    if (init != -1) {
        int *vii = (int *)malloc(n * sizeof(int));
        vii[0] = init;
        for (int k = init - 1; k >= 0; --k) vii[init - k] = k;
        for (int k = init + 1; k < n; ++k) vii[k] = k;
        for (int k = 0; k < n; ++k) {
            const int v = valid[vii[k]];
            if (p->var_cate[v] == NOISY_CAND_HOM_VAR || p->var_cate[v] == CLEAN_HOM_VAR) continue;
            for (int c = 0; c < p->n_cr; ++c) { /* cr_overlap(v, v+1) */
                const int r = p->cr_read[c];
                if (!(p->start_var_idx[r] < v + 1 && v < p->end_var_idx[r] + 1)) continue;
                if (p->is_skipped[r] || p->haps[r] != 0) continue;
                int hap = init_assign_read_hap(p, r, target);
                if (hap == -1) hap = 1;
                p->haps[r] = hap;
                update_profile_cons_by_read(p, r, hap, target);
            }
        }
        free(vii);
    }
    for (int it = 0; it < 10; ++it) {
        int c1 = iter_update_phase_set(p, valid, n);
        int c2 = iter_update_cons_alle(p, valid, n, target);
        if (c1 == 0 && c2 == 0) break;
    }
    for (int i = 0; i < p->n_reads; ++i) { /* update_read_phase_set :322-339 */
        const int r = p->ordered_read_ids[i];
        if (p->is_skipped[r]) continue;
        if (p->start_var_idx[r] == -1) continue;
        int64_t ps = -1;
        for (int v = p->start_var_idx[r]; v <= p->end_var_idx[r]; ++v) {
            if (!is_valid[v]) continue;
            if (CONS(p, v, 1) != -1 && CONS(p, v, 2) != -1 && CONS(p, v, 1) != CONS(p, v, 2)) ps = p->var_phase_set[v];
            if (ps != -1) break;
        }
        p->phase_sets[r] = ps;
    }
    free(valid); free(is_valid);
    return 0;
}

/* ---- cgranges order: cr_is_sorted() ? as added : radix_sort_cr_intv (src/cgranges.c:13-86; RS_MIN_SIZE 64, RS_MAX_BITS 8, key = x) ---- */
typedef struct { uint64_t x; int y, label; } intv_t;
static void rs_insertsort(intv_t *beg, intv_t *end) {
    for (intv_t *i = beg + 1; i < end; ++i)
        if (i->x < (i - 1)->x) { intv_t *j, tmp = *i; for (j = i; j > beg && tmp.x < (j - 1)->x; --j) *j = *(j - 1); *j = tmp; }
}
static void rs_sort(intv_t *beg, intv_t *end, int n_bits, int s) {
    typedef struct { intv_t *b, *e; } bucket_t;
    int size = 1 << n_bits, m = size - 1;
    bucket_t b[256], *k, *be = b + size;
    for (k = b; k != be; ++k) k->b = k->e = beg;
    for (intv_t *i = beg; i != end; ++i) ++b[i->x >> s & m].e;
    for (k = b + 1; k != be; ++k) k->e += (k - 1)->e - beg, k->b = (k - 1)->e;
    for (k = b; k != be;) {
        if (k->b != k->e) {
            bucket_t *l;
            if ((l = b + (k->b->x >> s & m)) != k) {
                intv_t tmp = *k->b, swap;
                do { swap = tmp; tmp = *l->b; *l->b++ = swap; l = b + (tmp.x >> s & m); } while (l != k);
                *k->b++ = tmp;
            } else ++k->b;
        } else ++k;
    }
    for (b->b = beg, k = b + 1; k != be; ++k) k->b = (k - 1)->e;
    if (s) {
        s = s > n_bits ? s - n_bits : 0;
        for (k = b; k != be; ++k)
            if (k->e - k->b > 64) rs_sort(k->b, k->e, n_bits, s);
            else if (k->e - k->b > 1) rs_insertsort(k->b, k->e);
    }
}
void lcdo_cr_sorted_order(int n, const int *st, const int *en, int *order_out) {
    intv_t *a = (intv_t *)malloc((n > 0 ? n : 1) * sizeof(intv_t));
    int sorted = 1;
    for (int i = 0; i < n; ++i) { a[i].x = (uint64_t)(uint32_t)st[i]; a[i].y = en[i]; a[i].label = i; if (i && a[i - 1].x > a[i].x) sorted = 0; }
    if (!sorted) { if (n <= 64) rs_insertsort(a, a + n); else rs_sort(a, a + n, 8, 7 * 8); }
    for (int i = 0; i < n; ++i) order_out[i] = a[i].label;
    free(a);
}
