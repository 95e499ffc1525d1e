#!/bin/bash
# (experiment) per-read phase ticks of the single-wavefront chains: exp/lib_pt.so (-DLCD_X_PHASESTAT) on the driver's submission.  Usage: bash tools/gpu_pt.sh <tag>
tag=${1:-pt}
mkdir -p gpurun_out
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
cp exp/lib_pt.so longcalld_amd/liblcd_hotpath.so
timeout 600 python bench.py --steps 20 --warmup 2 --cpu-sample 0 --repeats 1 --depth-profile 0 --f3 0 --overlap 0 > gpurun_out/${tag}.out 2> gpurun_out/${tag}.err
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
for m in 0 1; do
grep "^\[pt\]" gpurun_out/${tag}.out | awk -v m=$m '$5==m {n++; reads+=$7; nodes+=$9; tot+=$11; for(i=13;i<=NF;i++) if($i!="|"){k++; s[i]+=$i}} END {printf "mode %d: %d chains, %d reads, nodes/chain %.0f, total ticks %.4g\n", m, n, reads, nodes/n, tot; for(i=13;i<=NF;i++) if(s[i]>0) printf "  col %d: %.3g (%.2f%%)  per read %.0f\n", i, s[i], 100*s[i]/tot, s[i]/reads}'
done
