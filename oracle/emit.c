/*
 * emit.c -- ORACLE (test infrastructure only; see lcd_oracle.h).  PARITY UNPINNED (no reference binary can be built here).
 *
 * SURVEY 8(f) f4, restated from the reference's source function by function on flattened arrays:
 *   cal_sample_GQ / cal_var_QUAL1      src/collect_var.c:1435-1459
 *   make_variants                      src/collect_var.c:1465-1601   (germline fields; TSD / TE / somatic members left out)
 *   update_chunk_read/var_hap_phase_set1, flip_variant_hap   src/collect_var.c:1566-1680
 *   write_var_to_vcf                   src/vcf_utils.c:97-268        (text of the body lines)
 * var1_t's layout quirk is kept on purpose: `int DP, AD[2]; uint8_t GT[2];` (src/call_var_main.h:118) and the loop at :1567 stores every allele's
 * coverage through AD[j], so a third allele's count overwrites the GT bytes (x86-64 layout, little endian) -- the struct below has the same layout.
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

static const unsigned char nt4_of(unsigned char c) { /* nst_nt4_table, src/seq.c:14-31 (the byte codes 0..3 map to themselves) */
    switch (c) { case 0: case 'A': case 'a': return 0; case 1: case 'C': case 'c': return 1; case 2: case 'G': case 'g': return 2; case 3: case 'T': case 't': return 3; case '-': return 5; default: return 4; }
}

typedef struct { /* the tail of var1_t, same member order */
    int DP, AD[2]; uint8_t GT[2];
    int QUAL, GQ;
} lcdo_var_tail_t;

static int cal_sample_GQ(int ref_depth, int alt_depth, double logp, double log1p, double log2, int max_gq) {
    int PL[3];
    PL[0] = (int)(-10 * (ref_depth * log1p + alt_depth * logp));
    PL[1] = (int)(10 * (ref_depth + alt_depth) * log2);
    PL[2] = (int)(-10 * (ref_depth * logp + alt_depth * log1p));
    int min_pl = INT_MAX, sec_min_pl = INT_MAX;
    for (int i = 0; i < 3; ++i) {
        if (PL[i] < min_pl) { sec_min_pl = min_pl; min_pl = PL[i]; }
        else if (PL[i] < sec_min_pl) sec_min_pl = PL[i];
    }
    int GQ = sec_min_pl - min_pl;
    return max_gq < GQ ? max_gq : GQ;
}
static int cal_var_QUAL1(int ref_depth, int alt_depth, double logp, double log1p, int max_qual) {
    int q = (int)(-10 * (ref_depth * log1p + alt_depth * logp));
    return max_qual < q ? max_qual : q;
}

int lcdo_make_variants(const lcdo_call_opt_t *opt, const lcdo_hap_problem_t *p, const int *var_ref_len, const int *var_alt_len, const uint64_t *alt_off,
                       const uint8_t *alt_pool, const uint8_t *alt_ref_base, const char *ref_seq, int64_t ref_beg, int64_t reg_beg, int64_t reg_end,
                       lcdo_var1_t **vars_out) {
    int n_cand_vars = p->n_vars;
    *vars_out = NULL;
    if (n_cand_vars <= 0) return 0;
    lcdo_var1_t *vars = (lcdo_var1_t *)calloc((size_t)n_cand_vars + 1, sizeof(lcdo_var1_t));
    int i = 0, is_hom, hom_alt_is_set, hom_alle, hap1_alle, hap2_alle;
    int target_var_cate = 0x004 | 0x008 | 0x080 | 0x100 | 0x200;
    for (int cand_i = 0; cand_i < n_cand_vars; ++cand_i) {
        if ((p->var_cate[cand_i] & target_var_cate) == 0) continue;
        lcdo_var1_t *v = vars + i;
        memset(v, 0, sizeof(*v));
        v->cand_i = cand_i; v->te_seq_i = -1; v->tsd_pos1 = -1; v->tsd_pos2 = -1;
        if (p->var_type[cand_i] == 2 || p->var_type[cand_i] == 1) { v->pos = p->var_pos[cand_i] - 1; v->ref_len = var_ref_len[cand_i] + 1; }
        else { v->pos = p->var_pos[cand_i]; v->ref_len = var_ref_len[cand_i]; }
        if (v->pos < reg_beg || v->pos > reg_end) continue;
        hom_alle = p->hap_to_cons_alle[cand_i * 3 + 0];
        hap1_alle = p->hap_to_cons_alle[cand_i * 3 + 1];
        hap2_alle = p->hap_to_cons_alle[cand_i * 3 + 2];
        is_hom = 0; hom_alt_is_set = 0;
        if (hap1_alle == -1 && hap2_alle == -1) { is_hom = 1; hap1_alle = hap2_alle = hom_alle; }
        else if (hap1_alle == hap2_alle) is_hom = 1;
        if (hap1_alle == -1) hap1_alle = 0;
        if (hap2_alle == -1) hap2_alle = 0;
        v->type = p->var_type[cand_i];
        v->PS = p->var_phase_set[cand_i];
        v->ref_bases = (uint8_t *)malloc((size_t)v->ref_len + 1);
        for (int j = 0; j < v->ref_len; ++j) v->ref_bases[j] = nt4_of((unsigned char)ref_seq[v->pos - ref_beg + j]);
        v->n_alt_allele = 0; v->is_sv = 0;
        v->is_clean = (p->var_cate[cand_i] & (0x004 | 0x008 | 0x080)) != 0;
        lcdo_var_tail_t tail; memset(&tail, 0, sizeof(tail));
        for (int hap = 1; hap <= 2; ++hap) {
            int hap_alle = hap == 1 ? hap1_alle : hap2_alle;
            if (hap_alle != 0) {
                if (is_hom && hom_alt_is_set) { tail.GT[hap - 1] = (uint8_t)v->n_alt_allele; continue; }
                int alt_len = var_alt_len[cand_i];
                uint8_t *ab = (uint8_t *)malloc((size_t)alt_len + 2);
                if (v->type == 2 || v->type == 1) {
                    alt_len += 1;
                    if (alt_ref_base[cand_i] != 4) ab[0] = alt_ref_base[cand_i];
                    else ab[0] = nt4_of((unsigned char)ref_seq[v->pos - ref_beg]);
                    for (int j = 1; j < alt_len; ++j) ab[j] = alt_pool[alt_off[cand_i] + j - 1];
                } else {
                    for (int j = 0; j < alt_len; ++j) ab[j] = alt_pool[alt_off[cand_i] + j];
                }
                v->alt_bases[v->n_alt_allele] = ab;
                v->alt_len[v->n_alt_allele] = alt_len;
                if (abs(alt_len - v->ref_len) >= opt->min_sv_len) v->is_sv = 1;
                tail.GT[hap - 1] = (uint8_t)(++v->n_alt_allele);
                if (is_hom) hom_alt_is_set = 1;
            } else tail.GT[hap - 1] = 0;
        }
        tail.DP = p->total_cov[cand_i];
        {   /* for (j < n_uniq_alles) AD[j] = alle_covs[j]: the store runs past AD[2] into GT (and QUAL, assigned below) exactly as in var1_t */
            int n_uniq = p->alle_off[cand_i + 1] - p->alle_off[cand_i];
            int *ad = tail.AD;
            unsigned char *raw = (unsigned char *)ad;
            for (int j = 0; j < n_uniq && j < 4; ++j) memcpy(raw + 4 * j, &p->alle_covs[p->alle_off[cand_i] + j], 4);
        }
        if (tail.AD[1] > 0) {
            v->alt_read_i = (int *)malloc((size_t)tail.AD[1] * sizeof(int));
            int idx = 0;
            for (int k = 0; k < p->n_reads; ++k) {
                int read_i = p->ordered_read_ids[k];
                if (p->is_skipped[read_i]) continue;
                if (p->start_var_idx[read_i] < 0 || p->end_var_idx[read_i] < 0) continue;
                if (cand_i < p->start_var_idx[read_i] || cand_i > p->end_var_idx[read_i]) continue;
                int allele = p->alleles[p->allele_off[read_i] + cand_i - p->start_var_idx[read_i]];
                if (allele == 1) {
                    if (idx >= tail.AD[1]) return -6; /* _err_error_exit */
                    v->alt_read_i[idx++] = read_i;
                }
            }
            if (idx != tail.AD[1]) tail.AD[1] = idx;
            v->n_alt_reads = idx;
        } else v->alt_read_i = NULL;
        tail.QUAL = cal_var_QUAL1(tail.AD[0], tail.AD[1], opt->log_p, opt->log_1p, opt->max_qual);
        tail.GQ = cal_sample_GQ(tail.AD[0], tail.AD[1], opt->log_p, opt->log_1p, opt->log_2, opt->max_gq);
        v->DP = tail.DP; v->AD[0] = tail.AD[0]; v->AD[1] = tail.AD[1]; v->GT[0] = tail.GT[0]; v->GT[1] = tail.GT[1]; v->QUAL = tail.QUAL; v->GQ = tail.GQ;
        {   /* AD[2] as var1_t's memory reads: the bytes behind AD[1] are GT[0], GT[1] and two bytes of padding (taken as zero: the reference mallocs the records) --
             * or the third allele's coverage, which the store above wrote over all four */
            int n_uniq = p->alle_off[cand_i + 1] - p->alle_off[cand_i];
            v->AD[2] = n_uniq > 2 ? p->alle_covs[p->alle_off[cand_i] + 2] : ((int)tail.GT[0] | ((int)tail.GT[1] << 8));
        }
        i++;
    }
    *vars_out = vars;
    return i;
}

void lcdo_free_variants(lcdo_var1_t *v, int n) {
    if (!v) return;
    for (int i = 0; i < n; ++i) { free(v[i].ref_bases); free(v[i].alt_bases[0]); free(v[i].alt_bases[1]); free(v[i].alt_read_i); free(v[i].tsd_seq); }
    free(v);
}

int lcdo_flip_variant_hap(lcdo_chunk_phase_t *pre_chunk, lcdo_chunk_phase_t *cur_chunk, int out_aln) {
    if (cur_chunk->tid != pre_chunk->tid) return 0;
    int n_cur_ovlp_reads = cur_chunk->n_up_ovlp, n_pre_ovlp_reads = pre_chunk->n_down_ovlp;
    if (n_cur_ovlp_reads != n_pre_ovlp_reads) return -6;
    if (n_cur_ovlp_reads <= 0) return 0;
    if (pre_chunk->n_vars <= 0 || cur_chunk->n_vars <= 0) return 0;
    int flip_hap_score = 0; int64_t max_pre_read_PS = -1, min_cur_read_PS = INT64_MAX;
    for (int j = 0; j < cur_chunk->n_up_ovlp; ++j) {
        int cur_read_i = cur_chunk->up_ovlp_read_i[j];
        int pre_read_i = pre_chunk->down_ovlp_read_i[j];
        if (pre_chunk->is_skipped[pre_read_i] || pre_chunk->haps[pre_read_i] == 0 || cur_chunk->is_skipped[cur_read_i] || cur_chunk->haps[cur_read_i] == 0) continue;
        int pre_read_hap = pre_chunk->haps[pre_read_i]; int64_t pre_read_PS = pre_chunk->phase_sets[pre_read_i];
        int cur_read_hap = cur_chunk->haps[cur_read_i]; int64_t cur_read_PS = cur_chunk->phase_sets[cur_read_i];
        if (pre_read_hap == cur_read_hap) flip_hap_score -= 1; else flip_hap_score += 1;
        if (max_pre_read_PS < pre_read_PS) max_pre_read_PS = pre_read_PS;
        if (min_cur_read_PS > cur_read_PS) min_cur_read_PS = cur_read_PS;
    }
    if (flip_hap_score == 0) return 0;
    cur_chunk->flip_pre_PS = max_pre_read_PS;
    cur_chunk->flip_cur_PS = min_cur_read_PS;
    cur_chunk->flip_hap = flip_hap_score > 0 ? 1 : 0;
    /* update_chunk_var_hap_phase_set1 */
    if (cur_chunk->flip_hap && cur_chunk->flip_cur_PS != -1) {
        for (int i = 0; i < cur_chunk->n_vars; ++i) {
            if (cur_chunk->var_phase_set[i] == cur_chunk->flip_cur_PS) {
                int tmp = cur_chunk->hap_to_cons_alle[i * 3 + 1];
                cur_chunk->hap_to_cons_alle[i * 3 + 1] = cur_chunk->hap_to_cons_alle[i * 3 + 2];
                cur_chunk->hap_to_cons_alle[i * 3 + 2] = tmp;
            }
        }
    }
    if (cur_chunk->flip_pre_PS != -1 && cur_chunk->flip_cur_PS != INT64_MAX) {
        for (int i = 0; i < cur_chunk->n_vars; ++i) {
            if (cur_chunk->var_phase_set[i] == -1) continue;
            if (cur_chunk->var_phase_set[i] == cur_chunk->flip_cur_PS) cur_chunk->var_phase_set[i] = cur_chunk->flip_pre_PS;
        }
    }
    if (out_aln) { /* update_chunk_read_hap_phase_set1 */
        if (cur_chunk->flip_hap && cur_chunk->flip_cur_PS != -1) {
            for (int i = 0; i < cur_chunk->n_reads; ++i) {
                int read_i = cur_chunk->ordered_read_ids[i];
                if (cur_chunk->haps[read_i] == 0) continue;
                if (cur_chunk->phase_sets[read_i] == cur_chunk->flip_cur_PS) cur_chunk->haps[read_i] = 3 - cur_chunk->haps[read_i];
            }
        }
        if (cur_chunk->flip_pre_PS != -1 && cur_chunk->flip_cur_PS != INT64_MAX) {
            for (int i = 0; i < cur_chunk->n_reads; ++i) {
                int read_i = cur_chunk->ordered_read_ids[i];
                if (cur_chunk->phase_sets[read_i] == -1) continue;
                if (cur_chunk->phase_sets[read_i] == cur_chunk->flip_cur_PS) cur_chunk->phase_sets[read_i] = cur_chunk->flip_pre_PS;
            }
        }
    }
    return 0;
}

int lcdo_format_vcf(const lcdo_call_opt_t *opt, const char *chrom, const lcdo_var1_t *vars, int n_vars, char **text_out) {
    return lcdo_format_vcf_te(opt, chrom, vars, n_vars, NULL, text_out);
}
int lcdo_format_vcf_te(const lcdo_call_opt_t *opt, const char *chrom, const lcdo_var1_t *vars, int n_vars, const char *const *te_names, char **text_out) {
    size_t cap = 1 << 16, tot = 0; char *text = (char *)malloc(cap);
    int n_output_vars = 0;
    int buf_m = 50000; char *buffer = (char *)malloc(buf_m);
    int len = 0;
    for (int var_i = 0; var_i < n_vars; var_i++) {
        const lcdo_var1_t var = vars[var_i];
        if (var.n_alt_allele == 0) continue;
        if (var.DP < opt->min_dp) continue;
        if (var.AD[1] < opt->min_alt_dp) continue;
        if (opt->out_amb_base == 0) {
            uint8_t skip = 0;
            for (int j = 0; j < var.ref_len; j++) if (var.ref_bases[j] >= 4) { skip = 1; break; }
            if (skip) continue;
            for (int j = 0; j < var.n_alt_allele; j++) {
                for (int k = 0; k < var.alt_len[j]; k++) if (var.alt_bases[j][k] >= 4) { skip = 1; break; }
                if (skip) break;
            }
            if (skip) continue;
        }
        int base_len = var.ref_len;
        for (int j = 0; j < var.n_alt_allele; j++) base_len += var.alt_len[j];
        if (base_len + 2048 > buf_m) { buf_m = base_len + 10000; buffer = (char *)realloc(buffer, buf_m); }
        len = snprintf(buffer, buf_m, "%s\t%lld\t.\t", chrom, (long long)var.pos);
        for (int j = 0; j < var.ref_len; j++) len += snprintf(buffer + len, buf_m - len, "%c", "ACGTN"[var.ref_bases[j]]);
        len += snprintf(buffer + len, buf_m - len, "\t");
        for (int j = 0; j < var.n_alt_allele; j++) {
            for (int k = 0; k < var.alt_len[j]; k++) len += snprintf(buffer + len, buf_m - len, "%c", "ACGTN"[var.alt_bases[j][k]]);
            if (j < var.n_alt_allele - 1) len += snprintf(buffer + len, buf_m - len, ",");
        }
        int k = 0;
        char SVLEN[1024] = "SVLEN=", tmp[1024];
        char SVTYPE[1024] = "SVTYPE=";
        for (int i = 0; i < var.n_alt_allele; i++) {
            if (var.is_sv) {
                if (k > 0) { strcat(SVLEN, ","); strcat(SVTYPE, ","); }
                sprintf(tmp, "%d", var.alt_len[i] - var.ref_len); strcat(SVLEN, tmp);
                sprintf(tmp, "%s", var.alt_len[i] > var.ref_len ? "INS" : "DEL"); strcat(SVTYPE, tmp);
                k++;
            }
        }
        len += snprintf(buffer + len, buf_m - len, "\t%d\tPASS\t", var.QUAL);
        if (var.is_clean) len += snprintf(buffer + len, buf_m - len, "CLEAN;");
        if (var.te_seq_i >= 0 && te_names) len += snprintf(buffer + len, buf_m - len, "MEI;"); /* (MEI and REPNAME go together, src/vcf_utils.c:184,194) */
        len += snprintf(buffer + len, buf_m - len, "END=%lld", (long long)(var.pos + var.ref_len - 1));
        if (var.is_sv) {
            len += snprintf(buffer + len, buf_m - len, ";%s;%s", SVTYPE, SVLEN);
            if (var.tsd_len > 0) {
                len += snprintf(buffer + len, buf_m - len, ";TSD=");
                for (int i = 0; i < var.tsd_len; ++i) len += snprintf(buffer + len, buf_m - len, "%c", "ACGTN"[var.tsd_seq[i]]);
                len += snprintf(buffer + len, buf_m - len, ";TSDLEN=%d;POLYALEN=%d;TSDPOS1=%lld", var.tsd_len, var.polya_len, (long long)var.tsd_pos1);
                if (var.tsd_pos2 > 0) len += snprintf(buffer + len, buf_m - len, ";TSDPOS2=%lld", (long long)var.tsd_pos2);
            }
            if (var.te_seq_i >= 0 && te_names) len += snprintf(buffer + len, buf_m - len, ";REPNAME=%c%s", "+-"[var.te_is_rev], te_names[var.te_seq_i]);
        }
        len += snprintf(buffer + len, buf_m - len, "\t");
        int gt1 = var.GT[0], gt2 = var.GT[1];
        int is_hom = gt1 == gt2; int gt_seperator = '|';
        if (var.PS == 0) { gt_seperator = '/'; if (gt1 > gt2) { int t2 = gt1; gt1 = gt2; gt2 = t2; } }
        len += snprintf(buffer + len, buf_m - len, "GT:DP:AD:VAF:GQ");
        if (is_hom == 0 && var.PS != 0) len += snprintf(buffer + len, buf_m - len, ":PS");
        len += snprintf(buffer + len, buf_m - len, "\t");
        len += snprintf(buffer + len, buf_m - len, "%d%c%d:%d:", gt1, gt_seperator, gt2, var.DP);
        for (int j = 0; j < 1 + var.n_alt_allele; j++) {
            if (j > 0) len += snprintf(buffer + len, buf_m - len, ",");
            len += snprintf(buffer + len, buf_m - len, "%d", var.AD[j]);
        }
        for (int j = 0; j < var.n_alt_allele; j++) {
            if (j == 0) len += snprintf(buffer + len, buf_m - len, ":");
            if (j > 0) len += snprintf(buffer + len, buf_m - len, ",");
            float vaf = (float)var.AD[j + 1] / var.DP;
            len += snprintf(buffer + len, buf_m - len, "%.3f", vaf);
        }
        len += snprintf(buffer + len, buf_m - len, ":%d", var.GQ);
        if (is_hom == 0 && var.PS != 0) len += snprintf(buffer + len, buf_m - len, ":%lld", (long long)var.PS);
        len += snprintf(buffer + len, buf_m - len, "\n");
        if (tot + (size_t)len + 1 > cap) { while (tot + (size_t)len + 1 > cap) cap *= 2; text = (char *)realloc(text, cap); }
        memcpy(text + tot, buffer, (size_t)len); tot += (size_t)len;
        n_output_vars++;
    }
    text[tot] = 0;
    free(buffer);
    *text_out = text;
    return n_output_vars;
}

/* the candidate behind a record: position pos + 1 (the record starts at the anchor base), an insertion's bases alt_bases[0] + 1; annotated when the gap has at
 * least min_sv_len bases (src/collect_var.c:1817,1834), values into the record as make_cand_vars0 / make_variants do (:1765-1777, :1504-1520) */
int lcdo_annotate_te(const lcdo_call_opt_t *opt, int min_tsd_len, int max_tsd_len, int min_polya_len, float min_polya_ratio, const struct lcdo_te_lib *te_lib,
                     const char *ref_seq, int64_t ref_beg, int64_t ref_end, lcdo_var1_t *vars, int n_vars) {
    int n = 0;
    uint8_t *buf = (uint8_t *)malloc((size_t)(max_tsd_len > 0 ? max_tsd_len : 1) + 1);
    for (int i = 0; i < n_vars; ++i) {
        lcdo_var1_t *v = vars + i;
        if (v->n_alt_allele < 1) continue;
        if (v->is_clean) continue; /* only candidates made from a noisy region's consensus carry TE fields (src/collect_var.c:1817,1834) */
        int gap_len;
        if (v->type == 1) gap_len = v->alt_len[0] - 1; else if (v->type == 2) gap_len = v->ref_len - 1; else continue;
        if (gap_len < opt->min_sv_len) continue;
        int64_t p1 = -1, p2 = -1; int polya = 0, te_i = -1, te_rev = 0;
        int tsd_len = lcdo_collect_te_info_from_cons(min_tsd_len, max_tsd_len, min_polya_len, min_polya_ratio, te_lib, ref_seq, ref_beg, ref_end, v->pos + 1, 0, v->type,
                                                     gap_len, v->type == 1 ? v->alt_bases[0] + 1 : NULL, buf, &p1, &p2, &polya, &te_i, &te_rev);
        free(v->tsd_seq); v->tsd_seq = NULL;
        if (tsd_len > 0) {
            v->tsd_len = tsd_len; v->polya_len = polya; v->tsd_pos1 = p1; v->tsd_pos2 = p2;
            v->tsd_seq = (uint8_t *)malloc(tsd_len); memcpy(v->tsd_seq, buf, tsd_len);
            n++;
        } else { v->tsd_len = 0; v->polya_len = 0; v->tsd_pos1 = -1; v->tsd_pos2 = -1; }
        if (te_i >= 0) { v->te_seq_i = te_i; v->te_is_rev = te_rev; } else { v->te_seq_i = -1; v->te_is_rev = 0; }
    }
    free(buf);
    return n;
}
