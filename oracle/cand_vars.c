/* oracle/cand_vars.c -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
 *
 * candidate variants of one noisy region from the alignment strings of collect_noisy_reg_aln_strs, and the
 * read x variant allele profile.  Restates, function by function:
 *   make_cand_vars_from_msa / make_cand_vars_from_baln0      src/collect_var.c:1784-1873
 *   exact_comp_var_site                                      src/collect_var.c:1878-1898
 *   is_match_aln_str, is_match_aln_str_del                   src/collect_var.c:1960-2037
 *   get_var_allele_i_from_cons_aln_str                       src/collect_var.c:2058-2070
 *   is_cover_aln_str, get_full_cover_from_{cons,ref_cons}_aln_str   :2072-2128
 *   update_cand_var_profile_from_cons_aln_str{,1,21,2}       src/collect_var.c:2133-2276
 *   make_vars_from_msa_cons_aln                              src/collect_var.c:2279-2347
 *   var_is_homopolymer_indel                                 src/collect_var.c:1720-1744
 *   update_read_var_profile_with_allele                      src/bam_utils.c:248-255
 * Not restated (stays on the host in the reference as well): collect_te_info_from_cons for gaps >= min_sv_len
 * (TSD / polyA / TE annotation); such variants carry is_homopolymer_indel = 0 and no TE fields here.
 * Parity: UNPINNED (the reference binary cannot be built in this image and holds no golden vectors for this path); every source
 * line of the functions above is present under /root/reference/src and was followed.
 */
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

#define GAPC 5
#define CDIFF 8
#define CINS 1
#define CDEL 2
#define CATE_NOISY_HET 0x100
#define CATE_NOISY_HOM 0x200

static int hp_indel(const uint8_t *cref, int64_t cref_beg, int64_t cref_len, int64_t ref_pos, int type, int ref_len, int alt_len,
                    const uint8_t *alt) {
    int64_t o = ref_pos - cref_beg;
    if (type == CDIFF) return 0;
    if (o < 0 || o + 5 > cref_len || (type == CDEL && o + ref_len > cref_len)) return 0; /* the reference reads past nothing: chunk ref covers it */
    if (type == CINS) {
        for (int i = 1; i < alt_len; ++i) if (alt[i] != alt[0]) return 0;
        for (int i = 0; i < 5; ++i) if (cref[o + i] != alt[0]) return 0;
        return 1;
    }
    for (int i = 1; i < ref_len; ++i) if (cref[o + i] != cref[o]) return 0;
    for (int i = 0; i < 5; ++i) if (cref[o + i] != cref[o]) return 0;
    return 1;
}

typedef struct { lcdo_noisy_var_t v; const uint8_t *alt; } var1_t;

/* one consensus: ref row / cons row of the ref<->cons string -> variant list (no_end_var = 0, the germline path) */
static int vars_of_cons(int min_sv_len, const uint8_t *cref, int64_t cref_beg, int64_t cref_len, int64_t reg_beg, const uint8_t *ref_row,
                        const uint8_t *cons_row, int aln_len, uint8_t **keep_cons, var1_t **out) {
    uint8_t *R = (uint8_t *)malloc(aln_len + 1), *C = (uint8_t *)malloc(aln_len + 1);
    int L = 0;
    for (int i = 0; i < aln_len; ++i)
        if (ref_row[i] != GAPC || cons_row[i] != GAPC) { R[L] = ref_row[i]; C[L++] = cons_row[i]; }
    var1_t *vs = (var1_t *)calloc(L + 1, sizeof(var1_t));
    int n = 0, i = 0;
    int64_t pos = reg_beg;
    while (i < L) {
        if (R[i] == C[i]) { ++i; ++pos; continue; }
        if (R[i] != GAPC && C[i] != GAPC) {
            const int next_ok = (i + 1 == L) || (R[i + 1] != GAPC && C[i + 1] != GAPC);
            if (next_ok) {
                var1_t *v = &vs[n++];
                v->v.pos = pos; v->v.var_type = CDIFF; v->v.ref_len = 1; v->v.alt_len = 1; v->v.ref_base = R[i]; v->v.alt_ref_base = 0; v->alt = C + i;
            }
            ++i; ++pos;
        } else if (R[i] == GAPC) {
            int g = 1;
            while (i + g < L && R[i + g] == GAPC && C[i + g] != GAPC) ++g;
            var1_t *v = &vs[n++];
            v->v.pos = pos; v->v.var_type = CINS; v->v.ref_len = 0; v->v.alt_len = g; v->alt = C + i;
            v->v.alt_ref_base = i >= 1 ? C[i - 1] : 4;
            v->v.is_homopolymer_indel = g >= min_sv_len ? 0 : hp_indel(cref, cref_beg, cref_len, pos, CINS, 0, g, C + i);
            i += g;
        } else {
            int g = 1;
            while (i + g < L && R[i + g] != GAPC && C[i + g] == GAPC) ++g;
            var1_t *v = &vs[n++];
            v->v.pos = pos; v->v.var_type = CDEL; v->v.ref_len = g; v->v.alt_len = 0; v->alt = NULL;
            v->v.alt_ref_base = i >= 1 ? C[i - 1] : 4;
            v->v.is_homopolymer_indel = g >= min_sv_len ? 0 : hp_indel(cref, cref_beg, cref_len, pos, CDEL, g, 0, NULL);
            i += g; pos += g;
        }
    }
    free(R);
    *keep_cons = C; *out = vs;
    return n;
}

static int site_cmp(const var1_t *a, const var1_t *b) {
    const int64_t pa = a->v.var_type == CDIFF ? a->v.pos : a->v.pos - 1, pb = b->v.var_type == CDIFF ? b->v.pos : b->v.pos - 1;
    if (pa != pb) return pa < pb ? -1 : 1;
    if (a->v.var_type != b->v.var_type) return a->v.var_type < b->v.var_type ? -1 : 1;
    if (a->v.ref_len != b->v.ref_len) return a->v.ref_len < b->v.ref_len ? -1 : 1;
    if (a->v.alt_len != b->v.alt_len) return a->v.alt_len < b->v.alt_len ? -1 : 1;
    if (a->v.var_type == CDIFF || a->v.var_type == CINS) return memcmp(a->alt, b->alt, a->v.alt_len);
    return 0;
}

static int in_window(const lcdo_aln_str_t *s, int i) { /* 0 skip, 1 inside, 2 stop */
    if (i < s->query_beg || i < s->target_beg) return 0;
    if (i > s->query_end || i > s->target_end) return 2;
    return 1;
}

static int match_str(const lcdo_aln_str_t *s, int tp, int len, float thres, int *full) {
    int cur = -1, n_eq = 0, n_x = 0, cs = 0, ce = 0;
    const int sp = tp < 0 ? 0 : tp, ep = tp < 0 ? len - 1 : tp + len - 1;
    for (int i = 0; i < s->aln_len; ++i) {
        if (s->target_aln[i] != GAPC) ++cur;
        if (cur == tp + len) break;
        const int w = in_window(s, i);
        if (w == 0) continue;
        if (w == 2) break;
        if (cur == sp) cs = 1;
        if (cur == ep) ce = 1;
        if (cur >= tp) { if (s->query_aln[i] == s->target_aln[i]) ++n_eq; else ++n_x; }
    }
    *full = cs && ce;
    const int ok = len >= 10 ? (n_eq >= (len * thres)) : (n_eq == len && n_x == 0);
    return ok ? 1 : (*full ? 0 : -1);
}

static int match_del(const lcdo_aln_str_t *s, int dl, int dr, int *full) {
    int cur = -1, started = 0, non_del = 0, cs = 0, ce = 0;
    const int sp = dl < 0 ? 0 : dl, ep = dr;
    for (int i = 0; i < s->aln_len; ++i) {
        if (s->target_aln[i] != GAPC) ++cur;
        if (cur > dr) break;
        const int w = in_window(s, i);
        if (w == 0) continue;
        if (w == 2) break;
        if (cur == sp) cs = 1;
        if (cur == ep) ce = 1;
        if (cur >= dl && cur < dr) { if (!started) started = 1; else if (s->query_aln[i] != GAPC) ++non_del; }
    }
    *full = cs && ce;
    return *full ? (non_del == 0) : -1;
}

static int cover_str(const lcdo_aln_str_t *s, int tp, int len) {
    int cur = -1, cs = 0, ce = 0;
    const int sp = tp < 0 ? 0 : tp, ep = tp < 0 ? len - 1 : tp + len - 1;
    for (int i = 0; i < s->aln_len; ++i) {
        if (s->target_aln[i] != GAPC) ++cur;
        const int w = in_window(s, i);
        if (w == 0) continue;
        if (w == 2) break;
        if (cur == sp) cs = 1;
        if (cur == ep) ce = 1;
        if (cs && ce) return 1;
    }
    return 0;
}

static int cover_via_ref(const lcdo_aln_str_t *cr, const lcdo_aln_str_t *rc, int beg_ref, int end_ref) {
    int cur_r = -1, cur_c = -1, bc = -1, ec = -1, reach = 0;
    for (int i = 0; i < rc->aln_len; ++i) {
        if (rc->target_aln[i] != GAPC) ++cur_r;
        if (rc->query_aln[i] != GAPC) ++cur_c;
        const int w = in_window(rc, i);
        if (w == 0) continue;
        if (w == 2) break;
        if (cur_r == beg_ref && bc == -1) bc = cur_c;
        if (cur_r == end_ref) reach = 1;
        if (reach && rc->query_aln[i] != GAPC) { ec = cur_c; break; }
    }
    return cover_str(cr, bc, ec - bc + 1);
}

static int allele_of(const lcdo_aln_str_t *cr, int type, int alt_pos, int alt_len, int *full) {
    *full = 0;
    if (type == CDIFF) return match_str(cr, alt_pos, 1, 0.9f, full);
    if (type == CINS) return match_str(cr, alt_pos, alt_len, 0.9f, full);
    if (type == CDEL) return match_del(cr, alt_pos - 1, alt_pos, full);
    return -1;
}

int lcdo_make_vars_from_msa_cons_aln(int min_sv_len, const uint8_t *chunk_ref, int64_t chunk_ref_beg, int64_t chunk_ref_len,
                                     int64_t noisy_reg_beg, int n_cons, const int *clu_n_seqs, lcdo_aln_str_t **aln_strs,
                                     lcdo_noisy_var_t **vars_out, uint8_t **alt_pool_out, int **prof_start, int **prof_end,
                                     int **prof_alleles) {
    *vars_out = NULL; *alt_pool_out = NULL; *prof_start = *prof_end = *prof_alleles = NULL;
    if (n_cons == 0) return 0;
    var1_t *hv[2] = {NULL, NULL}; uint8_t *keep[2] = {NULL, NULL}; int nh[2] = {0, 0};
    for (int c = 0; c < n_cons; ++c) {
        const lcdo_aln_str_t *rc = &aln_strs[c][0];
        nh[c] = vars_of_cons(min_sv_len, chunk_ref, chunk_ref_beg, chunk_ref_len, noisy_reg_beg, rc->target_aln, rc->query_aln, rc->aln_len,
                             &keep[c], &hv[c]);
    }
    const int cap = nh[0] + nh[1];
    var1_t *mv = (var1_t *)calloc(cap + 1, sizeof(var1_t));
    int n = 0;
    if (n_cons == 1) {
        for (int i = 0; i < nh[0]; ++i) { mv[n] = hv[0][i]; mv[n].v.cate = CATE_NOISY_HOM; mv[n].v.from_cons = 1; ++n; }
    } else {
        int i1 = 0, i2 = 0;
        while (i1 < nh[0] && i2 < nh[1]) {
            const int r = site_cmp(&hv[0][i1], &hv[1][i2]);
            if (r < 0) { mv[n] = hv[0][i1++]; mv[n].v.cate = CATE_NOISY_HET; mv[n].v.from_cons = 1; ++n; }
            else if (r > 0) { mv[n] = hv[1][i2++]; mv[n].v.cate = CATE_NOISY_HET; mv[n].v.from_cons = 2; ++n; }
            else { mv[n] = hv[0][i1++]; ++i2; mv[n].v.cate = CATE_NOISY_HOM; mv[n].v.from_cons = 3; ++n; }
        }
        for (; i1 < nh[0]; ++i1) { mv[n] = hv[0][i1]; mv[n].v.cate = CATE_NOISY_HET; mv[n].v.from_cons = 1; ++n; }
        for (; i2 < nh[1]; ++i2) { mv[n] = hv[1][i2]; mv[n].v.cate = CATE_NOISY_HET; mv[n].v.from_cons = 2; ++n; }
    }
    int n_reads = 0;
    for (int c = 0; c < n_cons; ++c) n_reads += clu_n_seqs[c];
    if (n > 0) {
        int *ps = (int *)malloc(sizeof(int) * (n_reads + 1)), *pe = (int *)malloc(sizeof(int) * (n_reads + 1));
        int *pa = (int *)malloc(sizeof(int) * ((size_t)n_reads * n + 1));
        for (int r = 0; r < n_reads; ++r) { ps[r] = -1; pe[r] = -2; }
        for (size_t k = 0; k < (size_t)n_reads * n; ++k) pa[k] = -1;
        int row = 0;
        for (int c = 0; c < n_cons; ++c) {
            const int clu_idx = c + 1;
            const lcdo_aln_str_t *rc = &aln_strs[c][0];
            for (int j = 0; j < clu_n_seqs[c]; ++j, ++row) {
                const lcdo_aln_str_t *cr = &aln_strs[c][2 * j + 1];
                int delta = 0;
                for (int i = 0; i < n; ++i) {
                    lcdo_noisy_var_t *v = &mv[i].v;
                    const int vb = (int)(v->pos - noisy_reg_beg), ve = v->var_type == CINS ? vb : vb + v->ref_len - 1;
                    const int mine = n_cons == 1 ? 1 : (v->from_cons & clu_idx) != 0;
                    int full = 0, al;
                    if (mine) al = allele_of(cr, v->var_type, vb - delta, v->alt_len, &full);
                    else {
                        if (v->var_type == CDIFF) full = cover_str(cr, vb - delta, 1);
                        else if (v->var_type == CINS) full = cover_str(cr, vb - delta, v->ref_len + 1);
                        else full = cover_via_ref(cr, rc, vb - 1, ve + 1);
                        al = 0;
                    }
                    if (full) {
                        v->total_cov++;
                        if (al != -1) v->alle_covs[al]++;
                        if (ps[row] == -1) ps[row] = i;                 /* update_read_var_profile_with_allele */
                        pe[row] = i;
                        pa[(size_t)row * n + i] = al;
                    }
                    if (mine) { if (v->var_type == CINS) delta -= v->alt_len; else if (v->var_type == CDEL) delta += v->ref_len; }
                }
            }
        }
        *prof_start = ps; *prof_end = pe; *prof_alleles = pa;
    }
    /* flatten */
    size_t alt_tot = 0;
    for (int i = 0; i < n; ++i) alt_tot += mv[i].v.alt_len;
    lcdo_noisy_var_t *vo = (lcdo_noisy_var_t *)malloc(sizeof(lcdo_noisy_var_t) * (n + 1));
    uint8_t *pool = (uint8_t *)malloc(alt_tot + 1);
    size_t o = 0;
    for (int i = 0; i < n; ++i) {
        vo[i] = mv[i].v; vo[i].alt_off = (int)o;
        if (mv[i].v.alt_len) { memcpy(pool + o, mv[i].alt, mv[i].v.alt_len); o += mv[i].v.alt_len; }
    }
    *vars_out = vo; *alt_pool_out = pool;
    for (int c = 0; c < 2; ++c) { free(hv[c]); free(keep[c]); }
    free(mv);
    return n;
}
