/*
 * digar_rewrite.c -- ORACLE (test infrastructure only; see lcd_oracle.h).  PARITY UNPINNED (no reference binary can be built here).
 *
 * SURVEY a13, restated function by function: update_digars_from_msa1 (src/align.c:1701-1743) with collect_left_digars (:1463), collect_right_digars
 * (:1500), collect_full_msa_digars (:1543), collect_left_msa_digars (:1583), collect_right_msa_digars (:1642), the push rule of push_digar0 /
 * push_digar_alt_seq + same_digar1 (src/bam_utils.c:557-614) and double_check_digar (src/bam_utils.h:102-120).
 * alt_seq is left out on both sides of the comparison: it is the read's bases [qi, qi + len) by construction (see lcd_hotpath.h).
 */
#include <stdlib.h>
#include <string.h>
#include "lcd_oracle.h"

#define CEQUAL 7
#define CDIFF 8
#define CINS 1
#define CDEL 2
#define CSOFT 4
#define CHARD 5
#define CMATCH 0

typedef lcdo_digar_t dg1;
static int same_digar1(dg1 d1, dg1 d2) {
    if (d1.type != CEQUAL && d1.type != CINS && d1.type != CDEL) return 0;
    if (d1.type == d2.type && d1.is_low_qual == d2.is_low_qual) return 1;
    return 0;
}
static dg1 *push_digar0(dg1 *digar, int *n_digar, int *m_digar, dg1 d) {
    if (d.len <= 0) return digar;
    if (*n_digar == 0 || same_digar1(digar[*n_digar - 1], d) == 0) {
        if (*n_digar == *m_digar) { *m_digar = *m_digar ? (*m_digar << 1) : 4; digar = (dg1 *)realloc(digar, *m_digar * sizeof(dg1)); }
        digar[*n_digar] = d;
        (*n_digar)++;
    } else digar[*n_digar - 1].len += d.len;
    return digar;
}

static dg1 *collect_left_digars(const dg1 *digars, int n_digar, int read_noisy_beg, int64_t ref_noisy_beg, int *n_left_digars) {
    dg1 *left_digars = NULL;
    *n_left_digars = 0; int m_left_digar = 0;
    for (int i = 0; i < n_digar; ++i) {
        int op = digars[i].type, qi = digars[i].qi, digar_qi_end;
        int64_t digar_ref_beg = digars[i].pos, digar_ref_end;
        if (i == 0 && (op == CSOFT || op == CHARD)) { left_digars = push_digar0(left_digars, n_left_digars, &m_left_digar, digars[i]); continue; }
        if (op == CDIFF || op == CEQUAL || op == CINS) digar_qi_end = qi + digars[i].len - 1; else digar_qi_end = qi;
        if (op == CDIFF || op == CEQUAL || op == CDEL) digar_ref_end = digars[i].pos + digars[i].len - 1; else digar_ref_end = digars[i].pos;
        if (qi >= read_noisy_beg && digar_ref_beg >= ref_noisy_beg) break;
        if (digar_qi_end < read_noisy_beg && digar_ref_end < ref_noisy_beg)
            left_digars = push_digar0(left_digars, n_left_digars, &m_left_digar, digars[i]);
        else if (digar_qi_end >= read_noisy_beg || digar_ref_end >= ref_noisy_beg) {
            if (op == CINS || op == CEQUAL || op == CDIFF) { dg1 d = digars[i]; d.len = read_noisy_beg - d.qi; left_digars = push_digar0(left_digars, n_left_digars, &m_left_digar, d); }
            else if (op == CDEL) { dg1 d = digars[i]; d.len = (int)(ref_noisy_beg - d.pos); left_digars = push_digar0(left_digars, n_left_digars, &m_left_digar, d); }
            break;
        }
    }
    return left_digars;
}

static dg1 *collect_right_digars(const dg1 *digars, int n_digar, int read_noisy_end, int64_t ref_noisy_end, int *n_right_digars) {
    dg1 *right_digars = NULL; *n_right_digars = 0; int m_right_digar = 0;
    for (int i = 0; i < n_digar; ++i) {
        int qi = digars[i].qi, op = digars[i].type, digar_qi_end;
        int64_t digar_ref_beg = digars[i].pos, digar_ref_end;
        if (i == n_digar - 1 && (op == CSOFT || op == CHARD)) { right_digars = push_digar0(right_digars, n_right_digars, &m_right_digar, digars[i]); continue; }
        if (op == CDIFF || op == CEQUAL || op == CINS) digar_qi_end = qi + digars[i].len - 1; else digar_qi_end = qi;
        if (op == CDIFF || op == CEQUAL || op == CDEL) digar_ref_end = digars[i].pos + digars[i].len - 1; else digar_ref_end = digars[i].pos;
        if (digar_qi_end <= read_noisy_end && digar_ref_end <= ref_noisy_end) continue;
        if (qi > read_noisy_end && digar_ref_beg > ref_noisy_end)
            right_digars = push_digar0(right_digars, n_right_digars, &m_right_digar, digars[i]);
        else if (qi <= read_noisy_end || digar_ref_beg <= ref_noisy_end) {
            if (op == CINS || op == CEQUAL || op == CDIFF) {
                dg1 d = digars[i]; d.len = digar_qi_end - read_noisy_end;
                d.qi = read_noisy_end + 1;
                if (op != CINS) d.pos = ref_noisy_end + 1;
                right_digars = push_digar0(right_digars, n_right_digars, &m_right_digar, d);
            } else if (op == CDEL) {
                dg1 d = digars[i]; d.len = (int)(digar_ref_end - ref_noisy_end);
                d.pos = ref_noisy_end + 1;
                right_digars = push_digar0(right_digars, n_right_digars, &m_right_digar, d);
            }
        }
    }
    return right_digars;
}

static dg1 mk(int64_t pos, int qi, int type, int len) { dg1 d; d.pos = pos; d.qi = qi; d.type = type; d.len = len; d.is_low_qual = 0; return d; }

/* the shared column loop of the three collect_*_msa_digars: columns outside [left_read_start, right_read_end] advance the positions only */
static dg1 *msa_loop(dg1 *msa_digars, int *n, int *m, int i0, int i1, int left_read_start, int right_read_end, const uint8_t *read_str, const uint8_t *ref_str,
                     int *read_pos_io, int64_t *ref_pos_io) {
    int read_pos = *read_pos_io; int64_t ref_pos = *ref_pos_io;
    for (int i = i0; i <= i1; ++i) {
        if (read_str[i] == 5 && ref_str[i] == 5) continue;
        if (read_str[i] != 5 && ref_str[i] != 5) {
            if (i >= left_read_start && i <= right_read_end) msa_digars = push_digar0(msa_digars, n, m, mk(ref_pos, read_pos, read_str[i] == ref_str[i] ? CEQUAL : CDIFF, 1));
            read_pos++; ref_pos++;
        } else if (read_str[i] != 5) {
            if (i >= left_read_start && i <= right_read_end) msa_digars = push_digar0(msa_digars, n, m, mk(ref_pos, read_pos, CINS, 1));
            read_pos++;
        } else {
            if (i >= left_read_start && i <= right_read_end) msa_digars = push_digar0(msa_digars, n, m, mk(ref_pos, read_pos, CDEL, 1));
            ref_pos++;
        }
    }
    *read_pos_io = read_pos; *ref_pos_io = ref_pos;
    return msa_digars;
}

static dg1 *collect_full_msa_digars(int read_beg, int64_t ref_beg, int msa_len, const uint8_t *read_str, const uint8_t *ref_str, int *n_msa_digars) {
    dg1 *msa_digars = NULL; *n_msa_digars = 0; int m = 0;
    if (msa_len <= 0) return NULL;
    int read_pos = read_beg; int64_t ref_pos = ref_beg;
    return msa_loop(msa_digars, n_msa_digars, &m, 0, msa_len - 1, 0, msa_len - 1, read_str, ref_str, &read_pos, &ref_pos);
}
static dg1 *collect_left_msa_digars(int read_beg, int qlen, int64_t ref_beg, int msa_len, const uint8_t *read_str, const uint8_t *ref_str, int *n_msa_digars) {
    dg1 *msa_digars = NULL; *n_msa_digars = 0; int m = 0;
    if (msa_len <= 0) return NULL;
    int read_pos = read_beg, read_end_pos; int64_t ref_pos = ref_beg;
    int right_read_end = msa_len - 1, right_skipped_read_base = 0;
    read_end_pos = read_pos - 1; int is_covered_by_ref = 0;
    for (int i = msa_len - 1; i >= 0; --i) {
        if (ref_str[i] != 5) is_covered_by_ref = 1;
        if (is_covered_by_ref && read_str[i] != 5) { right_read_end = i; break; }
        else if (!is_covered_by_ref && read_str[i] != 5) right_skipped_read_base++;
    }
    for (int i = 0; i < msa_len; ++i) if (read_str[i] != 5) read_end_pos++;
    msa_digars = msa_loop(msa_digars, n_msa_digars, &m, 0, msa_len - 1, 0, right_read_end, read_str, ref_str, &read_pos, &ref_pos);
    if (read_end_pos < qlen - 1 || right_skipped_read_base > 0)
        msa_digars = push_digar0(msa_digars, n_msa_digars, &m, mk(ref_pos, read_end_pos + 1, CSOFT, qlen - 1 - read_end_pos + right_skipped_read_base));
    return msa_digars;
}
static dg1 *collect_right_msa_digars(int read_end, int64_t ref_beg, int64_t ref_end, int msa_len, const uint8_t *read_str, const uint8_t *ref_str, int *n_msa_digars) {
    dg1 *msa_digars = NULL; *n_msa_digars = 0; int m = 0;
    if (msa_len <= 0) return NULL;
    int read_pos; int64_t _ref_pos, ref_pos;
    int left_read_start = 0, right_read_end = msa_len - 1, left_skipped_read_base = 0;
    read_pos = read_end + 1; _ref_pos = ref_end + 1; ref_pos = ref_beg;
    int is_covered_by_ref = 0;
    for (int i = 0; i < msa_len; ++i) {
        if (ref_str[i] != 5) is_covered_by_ref = 1;
        if (is_covered_by_ref && read_str[i] != 5) { left_read_start = i; break; }
        else if (!is_covered_by_ref && read_str[i] != 5) left_skipped_read_base++;
    }
    for (int i = msa_len - 1; i >= 0; --i) {
        if (ref_str[i] != 5) _ref_pos--;
        if (read_str[i] != 5) { read_pos--; ref_pos = _ref_pos; }
    }
    if (read_pos > 0 || left_skipped_read_base > 0) msa_digars = push_digar0(msa_digars, n_msa_digars, &m, mk(ref_pos, 0, CSOFT, read_pos + left_skipped_read_base));
    read_pos += left_skipped_read_base;
    return msa_loop(msa_digars, n_msa_digars, &m, left_read_start, right_read_end, left_read_start, right_read_end, read_str, ref_str, &read_pos, &ref_pos);
}

static int double_check_digar(const dg1 *digars, int n_digar) {
    if (n_digar == 0) return 0;
    for (int i = n_digar - 1; i > 0; --i) {
        int qi, last_i = i - 1; int last_qi = digars[last_i].qi;
        if (digars[last_i].type == CEQUAL || digars[last_i].type == CMATCH || digars[last_i].type == CDIFF || digars[last_i].type == CINS ||
            digars[last_i].type == CSOFT || digars[last_i].type == CHARD) qi = last_qi + digars[last_i].len;
        else qi = last_qi;
        if (qi != digars[i].qi) return 1;
    }
    return 0;
}

int lcdo_update_digars_from_msa1(const lcdo_digar_t *digars, int n_digar, int qlen, int msa_len, const uint8_t *ref_str, const uint8_t *read_str, int full_cover,
                                 int64_t noisy_reg_beg, int64_t noisy_reg_end, int read_beg, int read_end, lcdo_digar_t **out, int *n_out) {
    int new_n_digars = 0, new_m_digar = 0; dg1 *new_digars = NULL;
    int old_left_n_digars = 0, old_right_n_digars = 0, msa_n_digars = 0;
    dg1 *old_left_digars = NULL, *old_right_digars = NULL, *msa_digars = NULL;
    *out = NULL; *n_out = 0;
    int left_cover = (full_cover & 8) && !(full_cover & 4), right_cover = !(full_cover & 8) && (full_cover & 4), both = (full_cover & 8) && (full_cover & 4);
    if (!(full_cover & 8) && !(full_cover & 4)) return 2;
    else if (both || (left_cover && (full_cover & 1)) || (right_cover && (full_cover & 2))) {
        old_left_digars = collect_left_digars(digars, n_digar, read_beg, noisy_reg_beg, &old_left_n_digars);
        old_right_digars = collect_right_digars(digars, n_digar, read_end, noisy_reg_end, &old_right_n_digars);
        msa_digars = collect_full_msa_digars(read_beg, noisy_reg_beg, msa_len, read_str, ref_str, &msa_n_digars);
        for (int i = 0; i < old_left_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, old_left_digars[i]);
        for (int i = 0; i < msa_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, msa_digars[i]);
        for (int i = 0; i < old_right_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, old_right_digars[i]);
    } else if (left_cover) {
        old_left_digars = collect_left_digars(digars, n_digar, read_beg, noisy_reg_beg, &old_left_n_digars);
        msa_digars = collect_left_msa_digars(read_beg, qlen, noisy_reg_beg, msa_len, read_str, ref_str, &msa_n_digars);
        for (int i = 0; i < old_left_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, old_left_digars[i]);
        for (int i = 0; i < msa_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, msa_digars[i]);
    } else if (right_cover) {
        old_right_digars = collect_right_digars(digars, n_digar, read_end, noisy_reg_end, &old_right_n_digars);
        msa_digars = collect_right_msa_digars(read_end, noisy_reg_beg, noisy_reg_end, msa_len, read_str, ref_str, &msa_n_digars);
        for (int i = 0; i < msa_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, msa_digars[i]);
        for (int i = 0; i < old_right_n_digars; ++i) new_digars = push_digar0(new_digars, &new_n_digars, &new_m_digar, old_right_digars[i]);
    }
    free(old_left_digars); free(old_right_digars); free(msa_digars);
    if (double_check_digar(new_digars, new_n_digars)) { free(new_digars); return 1; }
    *out = new_digars; *n_out = new_n_digars;
    return 0;
}
