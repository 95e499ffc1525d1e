/* oracle/digar_priv.h -- TEST INFRASTRUCTURE: the pieces the four collect_digar_from_* restatements share (digar.c, digar_tags.c):
 * BAM operation codes, the xid queue and push_xid_size_queue_win (src/bam_utils.c:123-200), cr_add's clamping (src/cgranges.c:145-149). */
#ifndef LCDO_DIGAR_PRIV_H
#define LCDO_DIGAR_PRIV_H
#define CMATCH 0
#define CINS 1
#define CDEL 2
#define CREF_SKIP 3
#define CSOFT 4
#define CHARD 5
#define CEQUAL 7
#define CDIFF 8

typedef struct { int64_t *pos; int *lens, *counts; int front, rear, count, max_s, win, cap; } xidq_t;
typedef struct { int64_t st, en; int label; } iv_t;
typedef struct { iv_t *v; int n, cap; } ivlist_t;

static void iv_add(ivlist_t *l, int64_t st, int64_t en, int label) { /* cr_add, src/cgranges.c:145-149: a negative start is clamped, st > en is dropped */
    if (st < 0) st = 0;
    if (st > en) return;
    if (l->n == l->cap) { l->cap = l->cap ? l->cap * 2 : 8; l->v = (iv_t *)realloc(l->v, sizeof(iv_t) * l->cap); }
    l->v[l->n].st = st; l->v[l->n].en = en; l->v[l->n].label = label; l->n++;
}

static void q_push(xidq_t *q, int64_t pos, int len, int count, ivlist_t *cr, int64_t *cur_s, int64_t *cur_e, int *q_s, int *q_e) {
    if (q->rear + 1 >= q->cap) {
        q->cap *= 2;
        q->pos = (int64_t *)realloc(q->pos, sizeof(int64_t) * q->cap); q->lens = (int *)realloc(q->lens, sizeof(int) * q->cap);
        q->counts = (int *)realloc(q->counts, sizeof(int) * q->cap);
    }
    ++q->rear;
    q->pos[q->rear] = pos; q->lens[q->rear] = len; q->counts[q->rear] = count;
    q->count += count;
    while (q->pos[q->front] + q->lens[q->front] - 1 <= pos - q->win) { q->count -= q->counts[q->front]; q->front++; }
    if (count > 0 && q->count > q->max_s) {
        const int64_t ns = q->pos[q->front], ne = q->pos[q->rear] + q->lens[q->rear];
        if (*cur_s == -1) { *cur_s = ns; *cur_e = ne; *q_s = q->front; *q_e = q->rear; }
        else if (ns <= *cur_e) { *cur_e = ne; *q_e = q->rear; }
        else {
            int vs = 0;
            for (int i = *q_s; i <= *q_e; ++i) vs += q->counts[i];
            if (vs < (int)(*cur_e - *cur_s + 1)) vs = (int)(*cur_e - *cur_s + 1);
            iv_add(cr, *cur_s - 1, *cur_e, vs);
            *cur_s = ns; *cur_e = ne; *q_s = q->front; *q_e = q->rear;
        }
    }
}

#endif
