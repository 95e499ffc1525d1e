/* opt.c -- ORACLE (test infrastructure only). Defaults of the option fields the path reads. */
#include "lcd_oracle.h"
/* src/call_var_main.c:140-224 with src/align.h:21-26 and src/call_var_main.h:36-50 */
void lcdo_opt_default(lcdo_opt_t *o) {
    o->match = 2; o->mismatch = 6; o->gap_open1 = 6; o->gap_ext1 = 2; o->gap_open2 = 24; o->gap_ext2 = 1;
    o->gap_aln = LCDO_GAP_LEFT_ALN;
    o->min_af = 0.20; o->min_dp = 5;
    o->partial_aln_ratio = 1.1;
    o->min_noisy_reg_size_to_sample_reads = 10000;
    o->max_noisy_reg_len = 50000;
    o->noisy_reg_flank_len = 10;
    o->min_hap_full_reads = 1; o->min_hap_reads = 2;
    o->collect_ref_read_aln_str = 0;
    o->is_ont = 0;
}
