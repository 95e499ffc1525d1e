#!/bin/bash
# the driver's command under rocprofv3 (kernel stats, HBM PMC, SQ, stalls: tools/profile_driver.sh) and the software byte counters of the -DLCD_X_BYTESTAT build.
# Usage: bash tools/gpu_account.sh <tag>     then here: python tools/rocprof_summary.py <tag> "<title>" driver; python tools/hbm_account.py gpurun_out/<tag>_bs.out 20 profiles/<tag>_hbm_account.txt profiles/<tag>_traffic.json "<title>"
tag=$1
bash tools/profile_driver.sh $tag
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
cp exp/lib_bs.so longcalld_amd/liblcd_hotpath.so
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 0 --cpu-sample 0 --repeats 1 --depth-profile 0 --f3 0 --overlap 0 > gpurun_out/${tag}_bs.out 2> gpurun_out/${tag}_bs.err
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
grep -c "^\[bs\]" gpurun_out/${tag}_bs.out
# keep only what the summaries need (the merge back is capped at 64 MiB)
grep "^\[bs\]\|^\[bs-end\]\|^{" gpurun_out/${tag}_bs.out > gpurun_out/${tag}_bs.tmp; mv gpurun_out/${tag}_bs.tmp gpurun_out/${tag}_bs.out
du -sh gpurun_out/* | sort -h | tail -8
