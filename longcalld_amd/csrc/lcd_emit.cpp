// lcd_emit.cpp -- SURVEY 8(f) f4: what turns the hot path's per-chunk results into the caller-visible output: cross-chunk stitching of phase sets
// (flip_variant_hap, src/collect_var.c:1618-1680), genotype records (make_variants :1465-1601, cal_sample_GQ :1435, cal_var_QUAL1 :1455), the VCF body
// lines (write_var_to_vcf, src/vcf_utils.c:97-268) and the HP / PS tag policy (src/bam_utils.c:1955-2006).  Host code, as in the reference: serial,
// tens of microseconds per chunk; it lives here so that a caller of the library gets from K5 + noisy-region variants to VCF text without the reference's
// htslib-typed structs.  Germline fields only (somatic mode a20 is out of scope); the retrotransposon annotation of SV-size gaps (a14) is at the end of the file.
#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_set>
#include <vector>
#include "../../include/lcd_hotpath.h"

namespace {
enum { CLEAN_HET_SNP = 0x004, CLEAN_HET_INDEL = 0x008, CLEAN_HOM = 0x080, NOISY_HET = 0x100, NOISY_HOM = 0x200 };
const int kOutCate = CLEAN_HET_SNP | CLEAN_HET_INDEL | CLEAN_HOM | NOISY_HET | NOISY_HOM; // src/collect_var.c:1479
const int kCleanCate = CLEAN_HET_SNP | CLEAN_HET_INDEL | CLEAN_HOM;                           // LONGCALLD_CAND_GERMLINE_CLEAN_VAR_CATE
inline uint8_t nt4(unsigned char c) { return c < 4 ? c : (c == 'A' || c == 'a') ? 0 : (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : c == '-' ? 5 : 4; }

void join_var_phase(lcd_chunk_phase_t &c) { // update_chunk_var_hap_phase_set1, src/collect_var.c:1589-1611
    if (c.flip_hap && c.flip_cur_PS != -1)
        for (int i = 0; i < c.n_vars; ++i)
            if (c.var_phase_set[i] == c.flip_cur_PS) std::swap(c.hap_to_cons_alle[3 * i + 1], c.hap_to_cons_alle[3 * i + 2]);
    if (c.flip_pre_PS != -1 && c.flip_cur_PS != INT64_MAX)
        for (int i = 0; i < c.n_vars; ++i)
            if (c.var_phase_set[i] != -1 && c.var_phase_set[i] == c.flip_cur_PS) c.var_phase_set[i] = c.flip_pre_PS;
}
void join_read_phase(lcd_chunk_phase_t &c) { // update_chunk_read_hap_phase_set1, :1566-1587
    if (c.flip_hap && c.flip_cur_PS != -1)
        for (int i = 0; i < c.n_reads; ++i) {
            const int r = c.ordered_read_ids[i];
            if (c.haps[r] != 0 && c.phase_sets[r] == c.flip_cur_PS) c.haps[r] = 3 - c.haps[r];
        }
    if (c.flip_pre_PS != -1 && c.flip_cur_PS != INT64_MAX)
        for (int i = 0; i < c.n_reads; ++i) {
            const int r = c.ordered_read_ids[i];
            if (c.phase_sets[r] != -1 && c.phase_sets[r] == c.flip_cur_PS) c.phase_sets[r] = c.flip_pre_PS;
        }
}
void free_members(lcd_var1_t *vars, int n) { for (int i = 0; i < n; ++i) { free(vars[i].ref_bases); free(vars[i].alt_bases[0]); free(vars[i].alt_bases[1]); free(vars[i].alt_read_i); free(vars[i].tsd_seq); } }
} // namespace

extern "C" {

int lcd_flip_variant_hap(lcd_chunk_phase_t *pre, lcd_chunk_phase_t *cur, int update_reads) {
    if (cur->tid != pre->tid) return 0;
    if (cur->n_up_ovlp != pre->n_down_ovlp) return -6; // the reference exits here (:1627-1630)
    if (cur->n_up_ovlp <= 0 || pre->n_vars <= 0 || cur->n_vars <= 0) return 0;
    int score = 0; int64_t max_pre = -1, min_cur = INT64_MAX;
    for (int j = 0; j < cur->n_up_ovlp; ++j) {
        const int cr = cur->up_ovlp_read_i[j], pr = pre->down_ovlp_read_i[j];
        if (pre->is_skipped[pr] || pre->haps[pr] == 0 || cur->is_skipped[cr] || cur->haps[cr] == 0) continue;
        score += pre->haps[pr] == cur->haps[cr] ? -1 : 1;
        max_pre = std::max(max_pre, pre->phase_sets[pr]); min_cur = std::min(min_cur, cur->phase_sets[cr]);
    }
    if (score == 0) return 0;
    cur->flip_pre_PS = max_pre; cur->flip_cur_PS = min_cur; cur->flip_hap = score > 0;
    join_var_phase(*cur);
    if (update_reads) join_read_phase(*cur);
    return 0;
}
int lcd_stitch_chunks(lcd_chunk_phase_t *chunks, int n, int update_reads) {
    for (int i = 1; i < n; ++i) { const int rc = lcd_flip_variant_hap(chunks + i - 1, chunks + i, update_reads); if (rc) return rc; }
    return 0;
}

void lcd_call_opt_default(lcd_call_opt_t *o) {
    o->log_p = -3.0; o->log_1p = -0.00043451177401769168 /* log10(1 - 0.001) */; o->log_2 = 0.301023; o->max_gq = 60; o->max_qual = 60;
    o->min_sv_len = 30; o->min_dp = 5; o->min_alt_dp = 2; o->out_amb_base = 0;
}

int lcd_make_variants(const lcd_call_opt_t *opt, const lcd_hap_problem_t *p, const int *var_ref_len, const int *var_alt_len, const uint64_t *alt_off,
                      const uint8_t *alt_pool, const uint8_t *alt_ref_base, const char *ref_seq, int64_t ref_beg, int64_t reg_beg, int64_t reg_end,
                      lcd_var1_t **vars_out) {
    *vars_out = nullptr;
    const int V = p->n_vars;
    if (V <= 0) return 0;
    std::vector<lcd_var1_t> out;
    for (int ci = 0; ci < V; ++ci) {
        if ((p->var_cate[ci] & kOutCate) == 0) continue;
        lcd_var1_t v; memset(&v, 0, sizeof(v));
        v.cand_i = ci; v.te_seq_i = -1; v.tsd_pos1 = v.tsd_pos2 = -1; // (retrotransposon members: unset until lcd_annotate_te, src/collect_var.c:1504-1520)
        const bool gap = p->var_type[ci] == 2 || p->var_type[ci] == 1; // BAM_CDEL / BAM_CINS: the VCF record starts one base earlier
        v.pos = p->var_pos[ci] - (gap ? 1 : 0); v.ref_len = var_ref_len[ci] + (gap ? 1 : 0);
        if (v.pos < reg_beg || v.pos > reg_end) continue;
        int a1 = p->hap_to_cons_alle[3 * ci + 1], a2 = p->hap_to_cons_alle[3 * ci + 2];
        const int hom = p->hap_to_cons_alle[3 * ci];
        bool is_hom = false;
        if (a1 == -1 && a2 == -1) { is_hom = true; a1 = a2 = hom; } else if (a1 == a2) is_hom = true;
        if (a1 == -1) a1 = 0;
        if (a2 == -1) a2 = 0;
        v.type = p->var_type[ci]; v.PS = p->var_phase_set[ci];
        v.ref_bases = (uint8_t *)malloc((size_t)v.ref_len + 1);
        for (int j = 0; j < v.ref_len; ++j) v.ref_bases[j] = nt4((unsigned char)ref_seq[v.pos - ref_beg + j]);
        v.is_clean = (p->var_cate[ci] & kCleanCate) != 0;
        bool hom_set = false;
        for (int hap = 1; hap <= 2; ++hap) {
            const int al = hap == 1 ? a1 : a2;
            if (al == 0) { v.GT[hap - 1] = 0; continue; }
            if (is_hom && hom_set) { v.GT[hap - 1] = v.n_alt_allele; continue; }
            int alen = var_alt_len[ci];
            uint8_t *ab = (uint8_t *)malloc((size_t)alen + 2);
            const uint8_t *as = alt_pool + alt_off[ci];
            if (gap) {
                ab[0] = alt_ref_base[ci] != 4 ? alt_ref_base[ci] : nt4((unsigned char)ref_seq[v.pos - ref_beg]);
                for (int j = 0; j < alen; ++j) ab[1 + j] = as[j];
                alen += 1;
            } else
                for (int j = 0; j < alen; ++j) ab[j] = as[j];
            v.alt_bases[v.n_alt_allele] = ab; v.alt_len[v.n_alt_allele] = alen;
            if (std::abs(alen - v.ref_len) >= opt->min_sv_len) v.is_sv = 1;
            v.GT[hap - 1] = ++v.n_alt_allele;
            if (is_hom) hom_set = true;
        }
        v.DP = p->total_cov[ci];
        const int na = p->alle_off[ci + 1] - p->alle_off[ci];
        const int *cov = p->alle_covs + p->alle_off[ci];
        v.AD[0] = na > 0 ? cov[0] : 0; v.AD[1] = na > 1 ? cov[1] : 0;
        // var1_t declares `int DP, AD[2]; uint8_t GT[2];` and the reference stores EVERY allele's coverage through AD[j] (:1567): a third allele
        // (the "minor alt") lands on the two GT bytes, which follow AD in the struct -- reproduced, since it is what the reference writes to the VCF
        // (a fourth one lands on QUAL, which is assigned afterwards)
        if (na > 2) { v.GT[0] = cov[2] & 0xff; v.GT[1] = (cov[2] >> 8) & 0xff; }
        v.AD[2] = na > 2 ? cov[2] : ((v.GT[0] & 0xff) | ((v.GT[1] & 0xff) << 8)); // (what `AD[2]` reads in var1_t: the two GT bytes and the padding behind them, zero here)
        if (v.AD[1] > 0) {
            v.alt_read_i = (int *)malloc((size_t)v.AD[1] * sizeof(int));
            int k2 = 0;
            for (int k = 0; k < p->n_reads; ++k) {
                const int r = p->ordered_read_ids[k];
                if (p->is_skipped[r]) continue;
                const int s = p->start_var_idx[r], e = p->end_var_idx[r];
                if (s < 0 || e < 0 || ci < s || ci > e) continue;
                if (p->alleles[p->allele_off[r] + (ci - s)] == 1) { if (k2 >= v.AD[1]) { free_members(out.data(), (int)out.size()); free_members(&v, 1); return -6; /* the reference exits, :1579 */ } v.alt_read_i[k2++] = r; }
            }
            v.AD[1] = k2; v.n_alt_reads = k2;
        }
        v.QUAL = std::min(opt->max_qual, (int)(-10 * (v.AD[0] * opt->log_1p + v.AD[1] * opt->log_p)));
        {
            const int PL[3] = {(int)(-10 * (v.AD[0] * opt->log_1p + v.AD[1] * opt->log_p)), (int)(10 * (v.AD[0] + v.AD[1]) * opt->log_2), (int)(-10 * (v.AD[0] * opt->log_p + v.AD[1] * opt->log_1p))};
            int lo = INT_MAX, sec = INT_MAX;
            for (int x : PL) { if (x < lo) { sec = lo; lo = x; } else if (x < sec) sec = x; }
            v.GQ = std::min(opt->max_gq, sec - lo);
        }
        out.push_back(v);
    }
    lcd_var1_t *res = (lcd_var1_t *)malloc((out.size() + 1) * sizeof(lcd_var1_t));
    memcpy(res, out.data(), out.size() * sizeof(lcd_var1_t));
    *vars_out = res;
    return (int)out.size();
}
void lcd_free_variants(lcd_var1_t *vars, int n) {
    if (!vars) return;
    free_members(vars, n);
    free(vars);
}

int lcd_format_vcf(const lcd_call_opt_t *opt, const char *chrom, const lcd_var1_t *vars, int n, char **text_out) { return lcd_format_vcf_te(opt, chrom, vars, n, nullptr, text_out); }
int lcd_format_vcf_te(const lcd_call_opt_t *opt, const char *chrom, const lcd_var1_t *vars, int n, const char *const *te_names, char **text_out) {
    std::string t; int n_out = 0; char num[64];
    for (int i = 0; i < n; ++i) {
        const lcd_var1_t &v = vars[i];
        if (v.n_alt_allele == 0 || v.DP < opt->min_dp || v.AD[1] < opt->min_alt_dp) continue;
        if (!opt->out_amb_base) {
            bool bad = false;
            for (int j = 0; j < v.ref_len && !bad; ++j) bad = v.ref_bases[j] >= 4;
            for (int a = 0; a < v.n_alt_allele && !bad; ++a) for (int k = 0; k < v.alt_len[a] && !bad; ++k) bad = v.alt_bases[a][k] >= 4;
            if (bad) continue;
        }
        t += chrom; t += '\t'; t += std::to_string((long long)v.pos); t += "\t.\t";
        for (int j = 0; j < v.ref_len; ++j) t += "ACGTN"[v.ref_bases[j]];
        t += '\t';
        for (int a = 0; a < v.n_alt_allele; ++a) { for (int k = 0; k < v.alt_len[a]; ++k) t += "ACGTN"[v.alt_bases[a][k]]; if (a + 1 < v.n_alt_allele) t += ','; }
        std::string svlen = "SVLEN=", svtype = "SVTYPE=";
        if (v.is_sv) for (int a = 0; a < v.n_alt_allele; ++a) { if (a) { svlen += ','; svtype += ','; } svlen += std::to_string(v.alt_len[a] - v.ref_len); svtype += v.alt_len[a] > v.ref_len ? "INS" : "DEL"; }
        t += '\t'; t += std::to_string(v.QUAL); t += "\tPASS\t";
        if (v.is_clean) t += "CLEAN;";
        if (v.te_seq_i >= 0 && te_names) t += "MEI;"; // src/vcf_utils.c:184 (the reference always writes MEI and REPNAME together: without the TE names neither is written)
        t += "END="; t += std::to_string((long long)(v.pos + v.ref_len - 1));
        if (v.is_sv) {
            t += ';'; t += svtype; t += ';'; t += svlen;
            if (v.tsd_len > 0) { // :188-193
                t += ";TSD="; for (int k = 0; k < v.tsd_len; ++k) t += "ACGTN"[v.tsd_seq[k]];
                t += ";TSDLEN="; t += std::to_string(v.tsd_len); t += ";POLYALEN="; t += std::to_string(v.polya_len); t += ";TSDPOS1="; t += std::to_string((long long)v.tsd_pos1);
                if (v.tsd_pos2 > 0) { t += ";TSDPOS2="; t += std::to_string((long long)v.tsd_pos2); }
            }
            if (v.te_seq_i >= 0 && te_names) { t += ";REPNAME="; t += "+-"[v.te_is_rev ? 1 : 0]; t += te_names[v.te_seq_i]; } // :194
        }
        t += '\t';
        int g1 = v.GT[0], g2 = v.GT[1]; const bool hom = g1 == g2; char sep = '|';
        if (v.PS == 0) { sep = '/'; if (g1 > g2) std::swap(g1, g2); }
        t += "GT:DP:AD:VAF:GQ";
        if (!hom && v.PS != 0) t += ":PS";
        t += '\t'; t += std::to_string(g1); t += sep; t += std::to_string(g2); t += ':'; t += std::to_string(v.DP); t += ':';
        for (int a = 0; a < 1 + v.n_alt_allele; ++a) { if (a) t += ','; t += std::to_string(v.AD[a]); }
        for (int a = 0; a < v.n_alt_allele; ++a) { t += a ? ',' : ':'; snprintf(num, sizeof(num), "%.3f", (float)v.AD[a + 1] / v.DP); t += num; }
        t += ':'; t += std::to_string(v.GQ);
        if (!hom && v.PS != 0) { t += ':'; t += std::to_string((long long)v.PS); }
        t += '\n'; ++n_out;
    }
    char *o = (char *)malloc(t.size() + 1);
    memcpy(o, t.c_str(), t.size() + 1);
    *text_out = o;
    return n_out;
}

// ---- update_digars_from_msa1 (src/align.c:1701-1743) ----
// A read's digar list is rebuilt around one noisy region from its ref<->read alignment string: the old digars left of the region
// (collect_left_digars :1463), per-column digars of the string (collect_full / left / right_msa_digars :1543-1699), the old digars right of it
// (collect_right_digars :1500), merged by the reference's push rule (same_digar1, src/bam_utils.c:557: consecutive '=', I or D of equal
// is_low_qual join; X never does).  alt_seq is not materialised: for X and I it is the read's bases [qi, qi + len) by construction.
namespace {
struct DG { long long pos; int type, len, qi, lq; };
void push(std::vector<DG> &v, DG d) {
    if (d.len <= 0) return;
    if (!v.empty()) { const DG &l = v.back(); if ((l.type == 7 || l.type == 1 || l.type == 2) && l.type == d.type && l.lq == d.lq) { v.back().len += d.len; return; } }
    v.push_back(d);
}
void msa_cols(std::vector<DG> &v, const uint8_t *ref, const uint8_t *read, int from, int to, int read_pos, long long ref_pos) {
    for (int i = from; i <= to; ++i) {
        const bool r = read[i] != 5, f = ref[i] != 5;
        if (!r && !f) continue;
        if (r && f) { push(v, {ref_pos, read[i] == ref[i] ? 7 : 8, 1, read_pos, 0}); ++read_pos; ++ref_pos; }
        else if (r) { push(v, {ref_pos, 1, 1, read_pos, 0}); ++read_pos; }
        else { push(v, {ref_pos, 2, 1, read_pos, 0}); ++ref_pos; }
    }
}
} // namespace

int lcd_update_digars_from_msa1(const lcd_digar_t *dg, int n_digar, int qlen, int msa_len, const uint8_t *ref_str, const uint8_t *read_str, int full_cover,
                                int64_t noisy_reg_beg, int64_t noisy_reg_end, int read_beg, int read_end, lcd_digar_t **out, int *n_out) {
    *out = nullptr; *n_out = 0;
    const bool lc = (full_cover & 8) != 0, rc = (full_cover & 4) != 0, lg = (full_cover & 2) != 0, rg = (full_cover & 1) != 0;
    if (!lc && !rc) return 2; // LONGCALLD_NOISY_IS_NOT_COVER: untouched
    const bool whole = (lc && rc) || (lc && !rc && rg) || (!lc && rc && lg);
    std::vector<DG> left, right, mid, nu;
    auto qend = [](const lcd_digar_t &d) { return (d.type == 8 || d.type == 7 || d.type == 1) ? d.qi + d.len - 1 : d.qi; };
    auto rend = [](const lcd_digar_t &d) { return (d.type == 8 || d.type == 7 || d.type == 2) ? d.pos + d.len - 1 : d.pos; };
    if (whole || lc) // collect_left_digars
        for (int i = 0; i < n_digar; ++i) {
            const lcd_digar_t &d = dg[i];
            if (i == 0 && (d.type == 4 || d.type == 5)) { push(left, {d.pos, d.type, d.len, d.qi, d.is_low_qual}); continue; }
            if (d.qi >= read_beg && d.pos >= noisy_reg_beg) break;
            if (qend(d) < read_beg && rend(d) < noisy_reg_beg) push(left, {d.pos, d.type, d.len, d.qi, d.is_low_qual});
            else {
                if (d.type == 1 || d.type == 7 || d.type == 8) push(left, {d.pos, d.type, read_beg - d.qi, d.qi, d.is_low_qual});
                else if (d.type == 2) push(left, {d.pos, d.type, (int)(noisy_reg_beg - d.pos), d.qi, d.is_low_qual});
                break;
            }
        }
    if (whole || (!lc && rc)) // collect_right_digars
        for (int i = 0; i < n_digar; ++i) {
            const lcd_digar_t &d = dg[i];
            if (i == n_digar - 1 && (d.type == 4 || d.type == 5)) { push(right, {d.pos, d.type, d.len, d.qi, d.is_low_qual}); continue; }
            if (qend(d) <= read_end && rend(d) <= noisy_reg_end) continue;
            if (d.qi > read_end && d.pos > noisy_reg_end) push(right, {d.pos, d.type, d.len, d.qi, d.is_low_qual});
            else if (d.type == 1 || d.type == 7 || d.type == 8) push(right, {d.type == 1 ? d.pos : noisy_reg_end + 1, d.type, qend(d) - read_end, read_end + 1, d.is_low_qual});
            else if (d.type == 2) push(right, {noisy_reg_end + 1, d.type, (int)(rend(d) - noisy_reg_end), d.qi, d.is_low_qual});
        }
    if (msa_len > 0) {
        if (whole) msa_cols(mid, ref_str, read_str, 0, msa_len - 1, read_beg, noisy_reg_beg);
        else if (lc) { // collect_left_msa_digars: columns up to the last read base that still faces the reference; the rest of the read becomes a soft clip
            int last = msa_len - 1, skipped = 0, end_pos = read_beg - 1; bool cov = false;
            for (int i = msa_len - 1; i >= 0; --i) { if (ref_str[i] != 5) cov = true; if (cov && read_str[i] != 5) { last = i; break; } else if (!cov && read_str[i] != 5) ++skipped; }
            for (int i = 0; i < msa_len; ++i) end_pos += read_str[i] != 5;
            msa_cols(mid, ref_str, read_str, 0, last, read_beg, noisy_reg_beg);
            if (end_pos < qlen - 1 || skipped > 0) {
                long long ref_pos = noisy_reg_beg; // the reference position after ALL columns (the loop at :1592-1619 advances it outside the window too)
                for (int i = 0; i < msa_len; ++i) ref_pos += (ref_str[i] != 5);
                push(mid, {ref_pos, 4, qlen - 1 - end_pos + skipped, end_pos + 1, 0});
            }
        } else { // collect_right_msa_digars: a leading soft clip, then the columns from the first read base that faces the reference
            int first = 0, skipped = 0, read_pos = read_end + 1; long long rp = noisy_reg_end + 1, ref_pos = noisy_reg_beg; bool cov = false;
            for (int i = 0; i < msa_len; ++i) { if (ref_str[i] != 5) cov = true; if (cov && read_str[i] != 5) { first = i; break; } else if (!cov && read_str[i] != 5) ++skipped; }
            for (int i = msa_len - 1; i >= 0; --i) { if (ref_str[i] != 5) --rp; if (read_str[i] != 5) { --read_pos; ref_pos = rp; } }
            if (read_pos > 0 || skipped > 0) push(mid, {ref_pos, 4, read_pos + skipped, 0, 0});
            msa_cols(mid, ref_str, read_str, first, msa_len - 1, read_pos + skipped, ref_pos);
        }
    }
    for (const DG &d : left) push(nu, d);
    for (const DG &d : mid) push(nu, d);
    for (const DG &d : right) push(nu, d);
    // double_check_digar (src/bam_utils.h:102-120): query offsets must chain; otherwise the read keeps its old digars
    for (size_t i = nu.size(); i-- > 1;) {
        const DG &l = nu[i - 1];
        const int q = (l.type == 7 || l.type == 0 || l.type == 8 || l.type == 1 || l.type == 4 || l.type == 5) ? l.qi + l.len : l.qi;
        if (q != nu[i].qi) return 1; // rejected: *out stays NULL, keep the old list
    }
    lcd_digar_t *o = (lcd_digar_t *)malloc((nu.size() + 1) * sizeof(lcd_digar_t));
    for (size_t i = 0; i < nu.size(); ++i) { o[i].pos = nu[i].pos; o[i].type = nu[i].type; o[i].len = nu[i].len; o[i].qi = nu[i].qi; o[i].is_low_qual = nu[i].lq; }
    *out = o; *n_out = (int)nu.size();
    return 0;
}

void lcd_read_tags(int n, const int *haps, const int64_t *ps, uint8_t *has_hp, int *hp, uint8_t *has_ps, int64_t *ps_out) {
    for (int i = 0; i < n; ++i) { has_hp[i] = haps[i] != 0; hp[i] = haps[i]; has_ps[i] = ps[i] > 0; ps_out[i] = ps[i]; }
}
}

// ---- retrotransposon annotation of SV-size gaps (host code in the reference and here) ----
// collect_te_info (src/align.c:32-83): a gap (INS: the inserted bases, DEL: the deleted ones) is a candidate insertion of a transposable element when the bases
// that follow it on the reference repeat its first bases (target-site duplication, one mismatch allowed) and it ends in poly-A or starts, behind the
// duplication, with poly-T.  check_te_seq (src/kmer.c:218-253): which of the user's TE sequences (make_te_kmer_idx :120-150: all overlapping k-mers of each,
// forward and reverse-complemented, as two sets per sequence) shares the most of the gap's non-overlapping k-mers.
struct lcd_te_lib_t { int k; std::vector<std::unordered_set<uint32_t>> fwd, rev; };

static inline int te_nt4(uint8_t c) { // nst_nt4_table, src/seq.c:14-31: codes 0-3 and the letters ACGT / acgt; everything else ends a k-mer
    switch (c) { case 0: case 'A': case 'a': return 0; case 1: case 'C': case 'c': return 1; case 2: case 'G': case 'g': return 2; case 3: case 'T': case 't': return 3; default: return 4; }
}
// not_simple_kmer (src/kmer.c:16-24) as written there: the word is shifted down two bits per step and its lowest base compared with itself shifted UP by 2i bits,
// which for i >= 1 differs exactly when that base is not A.  So a k-mer is "simple" -- and left out -- only if every base above its last one is A.
static inline bool te_kmer_kept(uint32_t kmer, int k) {
    for (int i = 0; i < k; ++i) { if ((kmer & 3u) != ((kmer & 3u) << (2 * i))) return true; kmer >>= 2; }
    return false;
}
lcd_te_lib_t *lcd_te_lib_create(int n_seqs, const char *const *seqs, const int *lens, int kmer_len) {
    if (n_seqs < 0 || kmer_len < 1 || kmer_len > 15) return nullptr; // (the reference's mask (1 << 2k) - 1 is an int: k <= 15; its default is 15)
    lcd_te_lib_t *L = new lcd_te_lib_t(); L->k = kmer_len; L->fwd.resize(n_seqs); L->rev.resize(n_seqs);
    const uint32_t mask = (1u << (2 * kmer_len)) - 1u; const int shift1 = 2 * (kmer_len - 1);
    for (int s = 0; s < n_seqs; ++s) {
        uint32_t f = 0, r = 0; int l = 0;
        for (int i = 0; i < lens[s]; ++i) {
            const int c = te_nt4((uint8_t)seqs[s][i]);
            if (c >= 4) { l = 0; continue; }
            f = (f << 2) | (uint32_t)c; r = (r >> 2) | ((uint32_t)(c ^ 3) << shift1); // collect_kmer :52-75, collect_rev_kmer :27-50
            if (++l >= kmer_len) {
                if (te_kmer_kept(f & mask, kmer_len)) L->fwd[s].insert(f & mask);
                if (te_kmer_kept(r, kmer_len)) L->rev[s].insert(r);
            }
        }
    }
    return L;
}
void lcd_te_lib_destroy(lcd_te_lib_t *lib) { delete lib; }
int lcd_te_lib_n_seqs(const lcd_te_lib_t *lib) { return lib ? (int)lib->fwd.size() : 0; }

int lcd_check_te_seq(const lcd_te_lib_t *lib, const uint8_t *seq, int len, int *is_rev) {
    if (!lib) return -1;
    const int k = lib->k; const uint32_t mask = (1u << (2 * k)) - 1u;
    std::vector<uint32_t> q; // collect_query_kmer :153-176: k-mers at 0, k, 2k, ... of every stretch of valid bases
    { uint32_t h = 0; int l = 0;
      for (int i = 0; i < len; ++i) { const int c = te_nt4(seq[i]); if (c >= 4) { l = 0; continue; } h = (h << 2) | (uint32_t)c; if (++l == k) { if (te_kmer_kept(h & mask, k)) q.push_back(h & mask); l = 0; } } }
    if (q.empty()) return -1;
    int max_for = 0, max_rev = 0, for_i = -1, rev_i = -1;
    for (size_t s = 0; s < lib->fwd.size(); ++s) {
        int fc = 0, rc = 0;
        for (uint32_t x : q) { fc += lib->fwd[s].count(x) != 0; rc += lib->rev[s].count(x) != 0; }
        if (fc > max_for) { max_for = fc; for_i = (int)s; }
        if (rc > max_rev) { max_rev = rc; rev_i = (int)s; }
    }
    const int min_count = 3;
    if (max_for > max_rev) { *is_rev = 0; return max_for >= min_count ? for_i : -1; }
    *is_rev = 1; return max_rev >= min_count ? rev_i : -1; // (a tie, 0 : 0 included, reports the reverse strand, as the reference does)
}

void lcd_te_opt_default(lcd_te_opt_t *o) { o->min_tsd_len = 2; o->max_tsd_len = 100; o->min_polya_len = 10; o->min_polya_ratio = 0.8f; } // src/call_var_main.h:55-58

int lcd_collect_te_info(const lcd_te_opt_t *opt, const lcd_te_lib_t *lib, int var_type, const uint8_t *gap_seq, const uint8_t *flank_ref_seq, int gap_len,
                        int64_t gap_pos, uint8_t *tsd_seq, int64_t *tsd_pos1, int64_t *tsd_pos2, int *tsd_polya_len, int *te_seq_i, int *te_is_rev) {
    *tsd_pos1 = -1; *tsd_pos2 = -1; *tsd_polya_len = -1; *te_seq_i = -1; *te_is_rev = 0;
    int tsd_len = 0, n_mis = 0;
    for (int i = 0; i < gap_len; ++i) { // target-site duplication: the gap's first bases against the reference behind the gap, one mismatch allowed
        if (gap_seq[i] == flank_ref_seq[i]) tsd_len = i + 1;
        else if (++n_mis > 1) break;
        if (tsd_len > opt->max_tsd_len) break;
    }
    if (tsd_len < opt->min_tsd_len || tsd_len > opt->max_tsd_len) return 0;
    bool has_polya = false; const int max_search = 20;
    for (int n = 0, a = 0, i = gap_len - 1; i >= 0; --i) { // poly-A at the gap's end: the longest suffix (searched while A's keep coming within 20 bases) that is >= 80 % A
        ++n;
        if (gap_seq[i] == 0) { if (++a, n >= opt->min_polya_len && (float)a >= opt->min_polya_ratio * (float)n) { has_polya = true; *tsd_polya_len = n; } }
        else if (n > max_search) break;
    }
    if (!has_polya)
        for (int n = 0, t = 0, i = tsd_len; i < gap_len; ++i) { // poly-T behind the duplication: reported as a negative length
            ++n;
            if (gap_seq[i] == 3) { if (++t, n >= opt->min_polya_len && (float)t >= opt->min_polya_ratio * (float)n) { has_polya = true; *tsd_polya_len = -n; } }
            else if (n > max_search) break;
        }
    if (!has_polya) return 0;
    if (lib && !lib->fwd.empty()) *te_seq_i = lcd_check_te_seq(lib, gap_seq, gap_len, te_is_rev);
    for (int i = 0; i < tsd_len; ++i) tsd_seq[i] = flank_ref_seq[i];
    *tsd_pos1 = gap_pos; *tsd_pos2 = var_type == 2 /* BAM_CDEL */ ? gap_pos + gap_len : -1;
    return tsd_len;
}

// collect_te_info_from_cons (src/align.c:139-163; with cons_msa_seq = the variant's alt_seq and msa_gap_start = 0 also collect_te_info_from_var :87-131)
int lcd_collect_te_info_from_cons(const lcd_te_opt_t *opt, const lcd_te_lib_t *lib, const char *ref_seq, int64_t ref_beg, int64_t ref_end, int64_t gap_ref_start,
                                  int msa_gap_start, int var_type, int gap_len, const uint8_t *cons_msa_seq, uint8_t *tsd_seq, int64_t *tsd_pos1, int64_t *tsd_pos2,
                                  int *tsd_polya_len, int *te_seq_i, int *te_is_rev) {
    if (var_type != 1 && var_type != 2) return -4; // (the reference asserts INS / DEL)
    auto bseq1 = [&](int64_t pos) -> uint8_t { return pos < ref_beg || pos > ref_end ? 4 : (uint8_t)te_nt4((uint8_t)ref_seq[pos - ref_beg]); }; // get_bseq1, src/seq.c:101
    std::vector<uint8_t> gap(gap_len > 0 ? gap_len : 0), flank(gap_len > 0 ? gap_len : 0);
    for (int i = 0; i < gap_len; ++i) {
        if (var_type == 1) { gap[i] = cons_msa_seq[msa_gap_start + i]; flank[i] = bseq1(gap_ref_start + i); }
        else { gap[i] = bseq1(gap_ref_start + i); flank[i] = bseq1(gap_ref_start + i + gap_len); }
    }
    return lcd_collect_te_info(opt, lib, var_type, gap.data(), flank.data(), gap_len, gap_ref_start, tsd_seq, tsd_pos1, tsd_pos2, tsd_polya_len, te_seq_i, te_is_rev);
}

// the annotation in the output records: see include/lcd_hotpath.h.  The record of a gap starts one base before it (the anchor): gap position = pos + 1, an
// insertion's bases = alt_bases[0] + 1.
int lcd_annotate_te(const lcd_call_opt_t *opt, const lcd_te_opt_t *te_opt, const lcd_te_lib_t *te_lib, const char *ref_seq, int64_t ref_beg, int64_t ref_end,
                    lcd_var1_t *vars, int n_vars) {
    int n_tsd = 0;
    std::vector<uint8_t> tsd((size_t)std::max(te_opt->max_tsd_len, 1) + 1);
    for (int i = 0; i < n_vars; ++i) {
        lcd_var1_t &v = vars[i];
        if ((v.type != 1 && v.type != 2) || v.n_alt_allele < 1) continue;
        if (v.is_clean) continue; // (the reference computes TSD / poly-A / TE only for candidates made from a noisy region's consensus, src/collect_var.c:1817,1834 -- never for clean-region ones)
        const int gap_len = v.type == 1 ? v.alt_len[0] - 1 : v.ref_len - 1;
        if (gap_len < opt->min_sv_len) continue;
        int64_t p1, p2; int pa, ti, tr;
        const int tl = lcd_collect_te_info_from_cons(te_opt, te_lib, ref_seq, ref_beg, ref_end, v.pos + 1, 0, v.type, gap_len, v.type == 1 ? v.alt_bases[0] + 1 : nullptr,
                                                     tsd.data(), &p1, &p2, &pa, &ti, &tr);
        free(v.tsd_seq); v.tsd_seq = nullptr; v.tsd_len = 0; v.polya_len = 0; v.tsd_pos1 = v.tsd_pos2 = -1; v.te_seq_i = -1; v.te_is_rev = 0;
        if (tl > 0) { // make_cand_vars0, src/collect_var.c:1765-1777
            v.tsd_len = tl; v.polya_len = pa; v.tsd_pos1 = p1; v.tsd_pos2 = p2;
            v.tsd_seq = (uint8_t *)malloc((size_t)tl); memcpy(v.tsd_seq, tsd.data(), (size_t)tl);
            if (ti >= 0) { v.te_seq_i = ti; v.te_is_rev = tr; }
            ++n_tsd;
        }
    }
    return n_tsd;
}
