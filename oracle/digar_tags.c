/* oracle/digar_tags.c -- TEST INFRASTRUCTURE (CPU restatement; never linked into the product).
 *
 * SURVEY 8(f) row f2, the three other digar sources (src/collect_var.c:1072-1079 picks per read: EQX CIGAR, else cs, else MD, else reference bases):
 *   collect_digar_from_cs_tag        src/bam_utils.c:844-1008
 *   collect_digar_from_MD_tag        src/bam_utils.c:1010-1177
 *   collect_digar_from_ref_seq       src/bam_utils.c:1179-1328
 * Each is restated whole, statement by statement, so that the differences between them stay visible: the cs function takes its clips from the
 * first / last CIGAR operation and has its own clip rule; the MD function walks the MD string with a (base pointer, index) pair that strtol moves;
 * the reference-comparison function flushes '=' runs at pos - eq_len even after it stepped over bases outside the loaded reference window.
 * The shared tail (last window, cr_index, skip rule, overlap with the chunk region) is the one of oracle/digar.c.
 * Where the reference calls _err_error_exit the restatement returns -2.
 * Parity: UNPINNED (no golden vectors for these functions in the reference; its binary cannot be built here).
 */
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include "lcd_oracle.h"
#include "digar_priv.h"

typedef struct {
    lcdo_digar_t *d; int nd, cap;
    xidq_t q; ivlist_t cr;
    int64_t cur_s, cur_e; int q_s, q_e;
    int n_cand;
} dstate_t;

static void st_init(dstate_t *s, const lcdo_digar_opt_t *opt, int rlen) {
    s->cap = 16; s->nd = 0; s->d = (lcdo_digar_t *)malloc(sizeof(lcdo_digar_t) * s->cap);
    s->q.cap = rlen > 16 ? rlen : 16; s->q.pos = (int64_t *)malloc(sizeof(int64_t) * s->q.cap); s->q.lens = (int *)malloc(sizeof(int) * s->q.cap);
    s->q.counts = (int *)malloc(sizeof(int) * s->q.cap); s->q.front = 0; s->q.rear = -1; s->q.count = 0; s->q.max_s = opt->noisy_reg_max_xgaps; s->q.win = opt->noisy_reg_slide_win;
    s->cr.v = NULL; s->cr.n = s->cr.cap = 0;
    s->cur_s = s->cur_e = -1; s->q_s = s->q_e = -1; s->n_cand = 0;
}
static void push_d(dstate_t *s, int64_t pos, int type, int len, int qi, int lq) {
    if (s->nd == s->cap) { s->cap *= 2; s->d = (lcdo_digar_t *)realloc(s->d, sizeof(lcdo_digar_t) * s->cap); }
    s->d[s->nd].pos = pos; s->d[s->nd].type = type; s->d[s->nd].len = len; s->d[s->nd].qi = qi; s->d[s->nd].is_low_qual = lq; ++s->nd;
}
#define QPUSH(S, P, L, C) q_push(&(S)->q, (P), (L), (C), &(S)->cr, &(S)->cur_s, &(S)->cur_e, &(S)->q_s, &(S)->q_e)

/* the common tail: src/bam_utils.c:978-1007 == :1146-1176 == :1298-1327 */
static int st_finish(dstate_t *s, const lcdo_digar_opt_t *opt, int64_t read_pos0, int rlen, int64_t reg_beg, int64_t reg_end, int bad, lcdo_digar_t **digars_out,
                     int *n_digar_out, int64_t **noisy_out, int *n_noisy_out, int64_t **chunk_noisy_out, int *n_chunk_noisy_out, int64_t *beg_out, int64_t *end_out,
                     int *n_cand_out) {
    if (s->cur_s != -1) {
        int vs = 0;
        for (int i = s->q_s; i <= s->q_e; ++i) vs += s->q.counts[i];
        if (vs < (int)(s->cur_e - s->cur_s + 1)) vs = (int)(s->cur_e - s->cur_s + 1);
        iv_add(&s->cr, s->cur_s - 1, s->cur_e, vs);
    }
    const int n = s->cr.n;
    int *order = (int *)malloc(sizeof(int) * (n + 1)), *st = (int *)malloc(sizeof(int) * (n + 1)), *en = (int *)malloc(sizeof(int) * (n + 1));
    for (int i = 0; i < n; ++i) { st[i] = (int)s->cr.v[i].st; en[i] = (int)s->cr.v[i].en; }
    lcdo_cr_sorted_order(n, st, en, order);
    int64_t *noisy = (int64_t *)malloc(sizeof(int64_t) * 3 * (n + 1)), *cn = (int64_t *)malloc(sizeof(int64_t) * 3 * (n + 1));
    int total = 0, ncn = 0;
    for (int i = 0; i < n; ++i) { const iv_t *v = &s->cr.v[order[i]]; noisy[3 * i] = v->st; noisy[3 * i + 1] = v->en; noisy[3 * i + 2] = v->label; total += (int)(v->en - v->st + 1); }
    const int64_t beg = read_pos0 + 1, end = read_pos0 + rlen;
    const int mapped = (int)(end - beg + 1);
    int skip = 0;
    if (total > mapped * opt->max_noisy_frac_per_read || s->n_cand > mapped * opt->max_var_ratio_per_read) skip = 1;
    else
        for (int i = 0; i < n; ++i)
            if (!(noisy[3 * i] + 1 > reg_end || noisy[3 * i + 1] < reg_beg)) { cn[3 * ncn] = noisy[3 * i]; cn[3 * ncn + 1] = noisy[3 * i + 1]; cn[3 * ncn + 2] = noisy[3 * i + 2]; ++ncn; }
    *digars_out = s->d; *n_digar_out = s->nd; *noisy_out = noisy; *n_noisy_out = n; *chunk_noisy_out = cn; *n_chunk_noisy_out = ncn;
    *beg_out = beg; *end_out = end; *n_cand_out = s->n_cand;
    free(order); free(st); free(en); free(s->cr.v); free(s->q.pos); free(s->q.lens); free(s->q.counts);
    if (bad) return -2;
    return skip ? -1 : 0;
}
static int cigar_rlen(const uint32_t *cigar, int n_cigar) {
    int rlen = 0;
    for (int i = 0; i < n_cigar; ++i) { const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4); if (op == CMATCH || op == CDEL || op == CREF_SKIP || op == CEQUAL || op == CDIFF) rlen += len; }
    return rlen;
}
static int ins_low_qual(const uint8_t *qual, int qi, int len, int min_bq) {
    for (int k = 0; k < len; ++k) if (qual[qi + k] >= min_bq) return 0;
    return 1;
}

/* ---- src/bam_utils.c:844-1008 ---- */
int lcdo_collect_digar_from_cs_tag(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const char *cs, const uint8_t *qual, int qlen,
                                   int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome, int right_clip_is_palindrome,
                                   lcdo_digar_t **digars_out, int *n_digar_out, int64_t **noisy_out, int *n_noisy_out, int64_t **chunk_noisy_out,
                                   int *n_chunk_noisy_out, int64_t *beg_out, int64_t *end_out, int *n_cand_out) {
    int64_t pos = read_pos0 + 1; int qi = 0, bad = 0;
    const int rlen = cigar_rlen(cigar, n_cigar); const int64_t tlen = whole_ref_len;
    dstate_t S; st_init(&S, opt, rlen);
    /* left-end clipping (:877-891) */
    if ((cigar[0] & 0xf) == CSOFT || (cigar[0] & 0xf) == CHARD) {
        const int len = (int)(cigar[0] >> 4);
        push_d(&S, pos, left_clip_is_palindrome ? CHARD : (int)(cigar[0] & 0xf), len, qi, 0);
        if (len > opt->end_clip_reg && !left_clip_is_palindrome) {
            if (pos > 10) iv_add(&S.cr, pos - 1, pos + opt->end_clip_reg_flank_win, 0);
            S.n_cand++;
        }
        if ((cigar[0] & 0xf) == CSOFT) qi += len;
    }
    while (*cs) {
        if (*cs == ':') {
            char *e; const int len = (int)strtol(cs + 1, &e, 10);
            if (e == cs + 1) { bad = 1; break; }       /* (strtol without digits would leave the reference spinning on ':'; reported as an error here) */
            cs = e;
            push_d(&S, pos, CEQUAL, len, qi, 0);
            pos += len; qi += len;
        } else if (*cs == '=') {
            cs++; int len = 0;
            while (isalpha((unsigned char)*cs)) { len++; cs++; }
            push_d(&S, pos, CEQUAL, len, qi, 0);
            pos += len; qi += len;
        } else if (*cs == '*') {
            if (!cs[1] || !cs[2]) { bad = 1; break; }
            if (qual[qi] >= opt->min_bq) { QPUSH(&S, pos, 1, 1); push_d(&S, pos, CDIFF, 1, qi, 0); }
            else push_d(&S, pos, CDIFF, 1, qi, 1);
            pos++; qi++; cs += 3;
            S.n_cand++;
        } else if (*cs == '+') {
            cs++; int len = 0;
            while (isalpha((unsigned char)*cs)) { len++; cs++; }
            const int low = ins_low_qual(qual, qi, len, opt->min_bq);
            if (!low) QPUSH(&S, pos, 0, len);
            push_d(&S, pos, CINS, len, qi, low);
            qi += len;
            S.n_cand++;
        } else if (*cs == '-') {
            cs++; int len = 0;
            while (isalpha((unsigned char)*cs)) { len++; cs++; }
            if ((qi == 0 || qual[qi - 1] >= opt->min_bq) && qual[qi < qlen ? qi : qlen - 1] >= opt->min_bq) { QPUSH(&S, pos, len, len); push_d(&S, pos, CDEL, len, qi, 0); }
            else push_d(&S, pos, CDEL, len, qi, 1);
            pos += len;
            S.n_cand++;
        } else if (*cs == '~') {
            cs++;
            while (isalpha((unsigned char)*cs) || isdigit((unsigned char)*cs)) cs++;
        } else { bad = 1; break; }
    }
    /* right-end clipping (:962-976) */
    if (!bad && ((cigar[n_cigar - 1] & 0xf) == CSOFT || (cigar[n_cigar - 1] & 0xf) == CHARD)) {
        const int len = (int)(cigar[n_cigar - 1] >> 4);
        push_d(&S, pos, right_clip_is_palindrome ? CHARD : (int)(cigar[n_cigar - 1] & 0xf), len, qi, 0);
        if (len > opt->end_clip_reg && !right_clip_is_palindrome) {
            if (pos < tlen - 10) iv_add(&S.cr, pos - 1 - opt->end_clip_reg_flank_win, pos, 0);
            S.n_cand++;
        }
        if ((cigar[n_cigar - 1] & 0xf) == CSOFT) qi += len;
    }
    return st_finish(&S, opt, read_pos0, rlen, reg_beg, reg_end, bad, digars_out, n_digar_out, noisy_out, n_noisy_out, chunk_noisy_out, n_chunk_noisy_out, beg_out, end_out, n_cand_out);
}

/* the clip branch the MD and reference-comparison functions share with the EQX one (:1117-1133 == :1268-1288) */
static void clip_rule_a(dstate_t *S, const lcdo_digar_opt_t *opt, int i, int op, int len, int64_t pos, int qi, int64_t tlen, int lp, int rp) {
    if ((i == 0 && lp) || (i != 0 && rp)) push_d(S, pos, CHARD, len, qi, 0);
    else push_d(S, pos, op, len, qi, 0);
    if ((i == 0 && pos > 10) || (i != 0 && pos < tlen - 10)) {
        if (len > opt->end_clip_reg) {
            if (i == 0 && !lp) { if (pos > 1) iv_add(&S->cr, pos - 1, pos + opt->end_clip_reg_flank_win, 0); S->n_cand++; }
            else if (i != 0 && !rp) { if (pos < tlen) iv_add(&S->cr, pos - 1 - opt->end_clip_reg_flank_win, pos, 0); S->n_cand++; }
        }
    }
}

/* ---- src/bam_utils.c:1010-1177 ---- */
int lcdo_collect_digar_from_MD_tag(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const char *md_tag, const uint8_t *qual, int qlen,
                                   int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len, int left_clip_is_palindrome, int right_clip_is_palindrome,
                                   lcdo_digar_t **digars_out, int *n_digar_out, int64_t **noisy_out, int *n_noisy_out, int64_t **chunk_noisy_out,
                                   int *n_chunk_noisy_out, int64_t *beg_out, int64_t *end_out, int *n_cand_out) {
    int64_t pos = read_pos0 + 1; int qi = 0, bad = 0;
    const int rlen = cigar_rlen(cigar, n_cigar); const int64_t tlen = whole_ref_len;
    dstate_t S; st_init(&S, opt, rlen);
    /* a private copy padded with NULs, so the reads the reference makes one or two characters past a malformed tag's end stay defined */
    const size_t mdl = strlen(md_tag);
    char *buf = (char *)calloc(mdl + 8, 1); memcpy(buf, md_tag, mdl);
    char *md = buf; int md_i = 0;
    int last_eq_len = 0;
    for (int i = 0; i < n_cigar && !bad; ++i) {
        const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4);
        if (op == CMATCH) {
            int m_len = len, eq_len;
            while (1) {
                if (md + md_i > buf + mdl) { bad = 1; break; }
                if (last_eq_len > 0) {
                    if (last_eq_len >= m_len) {
                        eq_len = m_len;
                        push_d(&S, pos, CEQUAL, eq_len, qi, 0);
                        pos += eq_len; qi += eq_len;
                        last_eq_len -= m_len; m_len = 0;
                    } else {
                        push_d(&S, pos, CEQUAL, last_eq_len, qi, 0);
                        pos += last_eq_len; qi += last_eq_len;
                        m_len -= last_eq_len; md_i = 0;
                        last_eq_len = 0;
                    }
                } else if (isdigit((unsigned char)md[md_i])) {
                    eq_len = (int)strtol(&md[md_i], &md, 10);
                    if (eq_len > m_len) { last_eq_len = eq_len - m_len; eq_len = m_len; }
                    else if (eq_len == 0) { md_i = 0; continue; }
                    push_d(&S, pos, CEQUAL, eq_len, qi, 0);
                    pos += eq_len; qi += eq_len;
                    m_len -= eq_len; md_i = 0;
                } else if (isalpha((unsigned char)md[md_i])) {
                    if (qual[qi] >= opt->min_bq) { QPUSH(&S, pos, 1, 1); push_d(&S, pos, CDIFF, 1, qi, 0); }
                    else push_d(&S, pos, CDIFF, 1, qi, 1);
                    S.n_cand++;
                    pos++; qi++; m_len -= 1;
                    if (md[md_i + 1] == '\0' || md[md_i + 1] != '0') md_i++;
                    else md_i += 2;
                } else { bad = 1; break; }
                if (m_len <= 0) break;
            }
        } else if (op == CDEL) {
            if ((qi == 0 || qual[qi - 1] >= opt->min_bq) && qual[qi < qlen ? qi : qlen - 1] >= opt->min_bq) { QPUSH(&S, pos, len, len); push_d(&S, pos, CDEL, len, qi, 0); }
            else push_d(&S, pos, CDEL, len, qi, 1);
            S.n_cand++;
            pos += len;
            if (md + md_i <= buf + mdl) {
                md_i++;
                while (md[md_i] && isalpha((unsigned char)md[md_i])) md_i++;
                if (md[md_i] == '0') md_i++;
            }
        } else if (op == CINS) {
            const int low = ins_low_qual(qual, qi, len, opt->min_bq);
            if (!low) QPUSH(&S, pos, 0, len);
            push_d(&S, pos, CINS, len, qi, low);
            S.n_cand++;
            qi += len;
        } else if (op == CSOFT || op == CHARD) {
            clip_rule_a(&S, opt, i, op, len, pos, qi, tlen, left_clip_is_palindrome, right_clip_is_palindrome);
            if (op == CSOFT) qi += len;
        } else if (op == CREF_SKIP) pos += len;
        else if (op == CEQUAL || op == CDIFF) bad = 1;
    }
    free(buf);
    return st_finish(&S, opt, read_pos0, rlen, reg_beg, reg_end, bad, digars_out, n_digar_out, noisy_out, n_noisy_out, chunk_noisy_out, n_chunk_noisy_out, beg_out, end_out, n_cand_out);
}

/* ---- src/bam_utils.c:1179-1328 ---- */
static int lcdo_nt4(unsigned char c) { /* nst_nt4_table, src/seq.c:14-31 */
    switch (c) {
    case 0: case 'A': case 'a': return 0;
    case 1: case 'C': case 'c': return 1;
    case 2: case 'G': case 'g': return 2;
    case 3: case 'T': case 't': return 3;
    case '-': return 5;
    default: return 4;
    }
}
static const unsigned char nt16_int[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4}; /* htslib seq_nt16_int */
int lcdo_collect_digar_from_ref_seq(const lcdo_digar_opt_t *opt, int64_t read_pos0, const uint32_t *cigar, int n_cigar, const uint8_t *bseq, const uint8_t *qual, int qlen,
                                    const char *ref_seq, int64_t ref_beg, int64_t ref_end, int64_t reg_beg, int64_t reg_end, int64_t whole_ref_len,
                                    int left_clip_is_palindrome, int right_clip_is_palindrome, lcdo_digar_t **digars_out, int *n_digar_out, int64_t **noisy_out,
                                    int *n_noisy_out, int64_t **chunk_noisy_out, int *n_chunk_noisy_out, int64_t *beg_out, int64_t *end_out, int *n_cand_out) {
    int64_t pos = read_pos0 + 1; int qi = 0;
    const int rlen = cigar_rlen(cigar, n_cigar); const int64_t tlen = whole_ref_len;
    dstate_t S; st_init(&S, opt, rlen);
    for (int i = 0; i < n_cigar; ++i) {
        const int op = cigar[i] & 0xf, len = (int)(cigar[i] >> 4);
        if (op == CMATCH || op == CDIFF || op == CEQUAL) {
            int eq_len = 0;
            for (int j = 0; j < len; ++j) {
                if (pos < ref_beg || pos > ref_end) { pos++; qi++; continue; }
                const int ref_base = lcdo_nt4((unsigned char)ref_seq[pos - ref_beg]);
                const int read_base = nt16_int[(bseq[qi >> 1] >> ((~qi & 1) << 2)) & 0xf];
                if (ref_base != read_base) {
                    if (eq_len > 0) { push_d(&S, pos - eq_len, CEQUAL, eq_len, qi - eq_len, 0); eq_len = 0; }
                    if (qual[qi] >= opt->min_bq) { QPUSH(&S, pos, 1, 1); push_d(&S, pos, CDIFF, 1, qi, 0); }
                    else push_d(&S, pos, CDIFF, 1, qi, 1);
                    S.n_cand++;
                } else eq_len++;
                pos++; qi++;
            }
            if (eq_len > 0) { push_d(&S, pos - eq_len, CEQUAL, eq_len, qi - eq_len, 0); eq_len = 0; }
        } else if (op == CDEL) {
            if ((qi == 0 || qual[qi - 1] >= opt->min_bq) && qual[qi < qlen ? qi : qlen - 1] >= opt->min_bq) { QPUSH(&S, pos, len, len); push_d(&S, pos, CDEL, len, qi, 0); }
            else push_d(&S, pos, CDEL, len, qi, 1);
            S.n_cand++;
            pos += len;
        } else if (op == CINS) {
            const int low = ins_low_qual(qual, qi, len, opt->min_bq);
            if (!low) QPUSH(&S, pos, 0, len);
            push_d(&S, pos, CINS, len, qi, low);
            S.n_cand++;
            qi += len;
        } else if (op == CSOFT || op == CHARD) {
            clip_rule_a(&S, opt, i, op, len, pos, qi, tlen, left_clip_is_palindrome, right_clip_is_palindrome);
            if (op == CSOFT) qi += len;
        } else if (op == CREF_SKIP) pos += len;
    }
    return st_finish(&S, opt, read_pos0, rlen, reg_beg, reg_end, 0, digars_out, n_digar_out, noisy_out, n_noisy_out, chunk_noisy_out, n_chunk_noisy_out, beg_out, end_out, n_cand_out);
}
