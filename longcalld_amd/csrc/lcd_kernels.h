// lcd_kernels.h -- launchers of the gfx950 kernels (internal to liblcd_hotpath.so)
#pragma once
#include <hip/hip_runtime.h>
#include "lcd_types.h"

void lcd_launch_poa(const PoaChain *chains, const PoaRead *reads, const uint8_t *pool, uint8_t *arena, uint8_t *outpool,
                    PoaChainOut *outs, LcdScoring sc, int n_chains, int threads, int lds_bytes, hipStream_t stream, int *gate = nullptr, PoaSpare *spare = nullptr);
void lcd_launch_cu_probe(int *seen /* int[4096], zeroed */, hipStream_t stream);
void lcd_launch_gate(int *ctr, int target0, int target1, hipStream_t stream);
// lds_bytes > 0: the jobs keep their value ring in that much dynamic LDS (one wavefront per job); 0: ring in HBM, 256 threads per job
void lcd_launch_wfa(const WfaJob *jobs, const uint8_t *pool, uint8_t *arena, uint8_t *outpool, WfaOut *outs, LcdScoring sc,
                    int n_jobs, int lds_bytes, hipStream_t stream, int wide = 0); // wide: the HBM-ring class's SV-size jobs (diagonals in flight, wfa_kernel.hip)
void lcd_launch_edlib(const EdJob *jobs, const uint8_t *pool, uint8_t *arena, EdOut *outs, int n_jobs, hipStream_t stream);
void lcd_launch_strings(const StrJob *jobs, uint8_t *pool, StrOut *outs, int n_jobs, hipStream_t stream);
void lcd_launch_gather(const GatherJob *jobs, int n_jobs, hipStream_t stream);
void lcd_launch_patch_reads(PoaRead *tab, const ReadPatch *patches, int n, hipStream_t stream);
void lcd_launch_anchor_ends(const AnchorEndsJob *jobs, AnchorEndsOut *outs, int n_jobs, hipStream_t stream);
void lcd_launch_bam_walk(const BamWalkJob *jobs, BamWalkOut *outs, int n_jobs, hipStream_t st);
void lcd_launch_bam_stat(const BamStatJob *jobs, BamStatOut *outs, int n_jobs, hipStream_t st);
void lcd_launch_bam_cigar(const GatherJob *jobs, int n_jobs, hipStream_t st);
void lcd_launch_errrate(const ErrJob *jobs, const double *tab, double *out, int n_jobs, hipStream_t st);
void lcd_launch_compose(const CmpJob *jobs, CmpOut *outs, const CmpSeg *segs, int n_jobs, int emit, hipStream_t stream);
void lcd_launch_vars_scan(const VarScanJob *jobs, VarScanOut *outs, int n_jobs, hipStream_t stream);
void lcd_launch_vars_profile(const VarRegJob *jobs, VarRegOut *outs, const StrJob *sjobs, const StrOut *souts, int n_jobs, hipStream_t stream);
void lcd_launch_digar(const DigarJob *jobs, DigarOut *outs, DigarOpt opt, int n_jobs, hipStream_t stream);
void lcd_launch_slices(const SliceJob *jobs, SliceOut *outs, const DigarRec *digars, int flank, int n_jobs, hipStream_t stream);
void lcd_launch_unpack(const UnpackJob *jobs, int n_jobs, const uint8_t *packed, uint8_t *pool, hipStream_t stream);
void lcd_launch_refcmp(bool emit, const RefCmpJob *jobs, RefCmpOut *outs, const char *ref, long long ref_beg, long long ref_end, int n_jobs, hipStream_t stream);
void lcd_launch_region_support(const IvRec *regs, int n_regs, const long long *read_beg, const long long *read_end, const unsigned long long *iv_off,
                               const IvRec *ivs, int n_reads, int *total, int *noisy, hipStream_t stream);
void lcd_launch_sdust(const unsigned char *pool, const SdSeg *segs, int T, int W, int seg, int n_seg, int cap, int *n_out, int2 *out, int4 *pbuf, int pcap, hipStream_t stream);
void lcd_launch_hap(const HapProb *probs, int n, hipStream_t stream);
