/*
 * ref_edlib_shim.cpp -- thin C entry points over the REFERENCE's own vendored edlib.
 *
 * Built only where /root/reference exists (this container), by oracle/Makefile, together with
 * /root/reference/edlib/src/edlib.cpp compiled from where it lies, into
 * oracle/_ref/libedlib_ref.so (git-ignored, travels with gpurun).  No reference source is copied
 * into this repo.  The wrappers below only call edlibAlign the way src/align.c:210-254 does and
 * hand back the raw alignment so the restatement in edlib_nw.c can be compared byte for byte.
 */
#include <cstring>
#include <cstdlib>
#include "edlib.h"

extern "C" {

/* returns edit distance (or -1); copies the alignment (ops 0=,1I,2D,3X) into aln_out if it fits */
int ref_edlib_nw_path(const unsigned char *query, int qlen, const unsigned char *target, int tlen,
                      unsigned char *aln_out, int aln_cap, int *aln_len) {
    EdlibAlignResult r = edlibAlign((const char *)query, qlen, (const char *)target, tlen,
                                    edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_PATH, NULL, 0));
    if (r.status != EDLIB_STATUS_OK) { edlibFreeAlignResult(r); return -1; }
    int d = r.editDistance;
    *aln_len = r.alignmentLength;
    if (aln_out && r.alignmentLength <= aln_cap && r.alignment) memcpy(aln_out, r.alignment, r.alignmentLength);
    edlibFreeAlignResult(r);
    return d;
}

/* HW (infix) mode + TASK_PATH, as src/align.c:256-275 calls it: distance; start / end locations [0]; the alignment */
int ref_edlib_hw_path(const unsigned char *query, int qlen, const unsigned char *target, int tlen,
                      unsigned char *aln_out, int aln_cap, int *aln_len, int *start0, int *end0, int *n_locations) {
    EdlibAlignResult r = edlibAlign((const char *)query, qlen, (const char *)target, tlen,
                                    edlibNewAlignConfig(-1, EDLIB_MODE_HW, EDLIB_TASK_PATH, NULL, 0));
    if (r.status != EDLIB_STATUS_OK) { edlibFreeAlignResult(r); return -1; }
    int d = r.editDistance;
    *aln_len = r.alignmentLength;
    *n_locations = r.numLocations;
    *start0 = r.numLocations > 0 && r.startLocations ? r.startLocations[0] : -1;
    *end0 = r.numLocations > 0 && r.endLocations ? r.endLocations[0] : -1;
    if (aln_out && r.alignmentLength <= aln_cap && r.alignment) memcpy(aln_out, r.alignment, r.alignmentLength);
    edlibFreeAlignResult(r);
    return d;
}

int ref_edlib_distance(const unsigned char *query, int qlen, const unsigned char *target, int tlen) {
    EdlibAlignResult r = edlibAlign((const char *)query, qlen, (const char *)target, tlen,
                                    edlibNewAlignConfig(-1, EDLIB_MODE_NW, EDLIB_TASK_DISTANCE, NULL, 0));
    int d = r.status == EDLIB_STATUS_OK ? r.editDistance : -1;
    edlibFreeAlignResult(r);
    return d;
}
}
