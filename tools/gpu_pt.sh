#!/bin/bash
# (experiment) per-read phase ticks of the single-wavefront chains: exp/lib_pt.so (-DLCD_X_PHASESTAT).  Usage: bash tools/gpu_pt.sh <tag> <bench flags>
tag=$1; shift
mkdir -p gpurun_out
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
cp exp/lib_pt.so longcalld_amd/liblcd_hotpath.so
timeout 600 python bench.py "$@" --warmup 1 --cpu-sample 0 --repeats 1 --depth-profile 0 --f3 0 --overlap 0 > gpurun_out/${tag}.out 2> gpurun_out/${tag}.err
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
for m in 0 1; do
grep "^\[pt\]" gpurun_out/${tag}.out | awk -v m=$m 'BEGIN{split("upd1.loads upd1.scan upd.sync upd2 upd3.loads upd3.edges upd3.scan upd4 | plan.init plan.L1 plan.L2 plan.L3 plan.L4 plan.rest plan.scan | inc.A0 inc.walk inc.apply inc.remain pd.valid remainblock.after.inc fullsort.fallback inc.total gathers", nm, " ")} $5==m {n++; reads+=$7; nodes+=$9; tot+=$11; for(i=13;i<=NF;i++) if($i!="|"){s[i]+=$i}} END {printf "mode %d: %d chains, %d reads, nodes/chain %.0f, total ticks %.4g\n", m, n, reads, nodes/(n+1e-9), tot; for(i=13;i<=NF;i++) if(s[i]>0) printf "  %-24s %.3g (%.2f%%)  per read %.0f\n", nm[i-12], s[i], 100*s[i]/tot, s[i]/reads}'
done
