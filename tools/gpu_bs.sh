#!/bin/bash
# software byte counters of the -DLCD_X_BYTESTAT build on a bench command (default: the driver's).  Usage: bash tools/gpu_bs.sh <tag> [bench flags]
tag=$1; shift
flags=${@:---gpus 1 --steps 20}
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
cp exp/lib_bs.so longcalld_amd/liblcd_hotpath.so
timeout 900 python bench.py $flags --warmup 0 --cpu-sample 0 --repeats 1 --depth-profile 0 --f3 0 --overlap 0 > gpurun_out/${tag}_bs.out 2> gpurun_out/${tag}_bs.err
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
grep -c "^\[bs-end\]" gpurun_out/${tag}_bs.out
