/*
 * ref_cgranges_shim.c -- thin C entry points over the REFERENCE's own interval index (src/cgranges.c, present in the reference tree).
 *
 * Built only where /root/reference exists (this container), by oracle/Makefile, together with /root/reference/src/cgranges.c compiled from
 * where it lies, into oracle/_ref/libcgranges_ref.so (git-ignored, travels with gpurun).  No reference source is copied into this repo.
 * It pins the two places where results of the hot path depend on cgranges' behaviour: the order in which cr_index() leaves the intervals
 * (K5 seeds reads in that order, src/assign_hap.c:512-526; a read's noisy windows are reported in it, src/bam_utils.c:806) and the order
 * in which cr_overlap() reports hits (K5, src/assign_hap.c:345-422).
 */
#include <stdlib.h>
#include "cgranges.h"

/* labels (= insertion indices) of the n intervals in the order cr_index() leaves them */
void ref_cr_sorted_labels(int n, const int *st, const int *en, int *labels_out) {
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n; ++i) cr_add(cr, "cr", st[i], en[i], i);
    cr_index(cr);
    for (int64_t i = 0; i < cr->n_r; ++i) labels_out[i] = cr_label(cr, i);
    cr_destroy(cr);
}

/* labels of the intervals overlapping [qst, qen), in the order cr_overlap() returns them; returns their number */
int ref_cr_overlap_labels(int n, const int *st, const int *en, int qst, int qen, int *labels_out, int cap) {
    cgranges_t *cr = cr_init();
    for (int i = 0; i < n; ++i) cr_add(cr, "cr", st[i], en[i], i);
    cr_index(cr);
    int64_t *b = 0, m = 0;
    const int64_t k = cr_overlap(cr, "cr", qst, qen, &b, &m);
    for (int64_t i = 0; i < k && i < cap; ++i) labels_out[i] = cr_label(cr, b[i]);
    free(b);
    cr_destroy(cr);
    return (int)k;
}
