#!/bin/bash
# the two HBM PMC passes of the driver's command only (tools/profile_driver.sh runs all five).  Usage: bash tools/gpu_pmc_quick.sh <tag>; then python tools/pmc_quick.py <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
PMC="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --repeats 1 --depth-profile 0 --overlap 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$tag -o f -- $PMC > gpurun_out/pmc_fetch_$tag.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$tag -o w -- $PMC > gpurun_out/pmc_write_$tag.log 2>&1
