#!/bin/bash
# rocprofv3 runs of the two noisy shapes' builder commands (configs[2] ONT: --shape ont --steps 48; configs[4] SV: --shape sv --steps 8): kernel trace + stats and the two
# HBM PMC passes, each in its own run.  Usage (GPU box, repo root): bash tools/profile_shapes.sh <tag>   then here: python tools/shape_summary.py <tag> "<title>"
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for sh in ont sv; do
  if [ $sh = ont ]; then CMD="python bench.py --gpus 1 --shape ont --steps 48 --warmup 2 --cpu-sample 0 --f3 0 --overlap 0 --depth-profile 0 --repeats 1"; else CMD="python bench.py --gpus 1 --shape sv --steps 8 --warmup 2 --cpu-sample 0 --f3 0 --overlap 0 --depth-profile 0 --repeats 1"; fi
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_$sh -o kt -- $CMD > gpurun_out/prof_${tag}_$sh.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_${tag}_$sh -o f -- $CMD > gpurun_out/pmc_fetch_${tag}_$sh.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_${tag}_$sh -o w -- $CMD > gpurun_out/pmc_write_${tag}_$sh.log 2>&1
  for f in prof pmc_fetch pmc_write; do grep -h '"metric"' gpurun_out/${f}_${tag}_$sh.log | tail -1 | cut -c1-160; done
done
find gpurun_out -name "*_results.db" | grep $tag | xargs du -sh
