#!/bin/bash
# rocprofv3 runs behind profiles/: kernel trace + stats of the default bench command, then the two HBM PMC passes (FETCH_SIZE and
# WRITE_SIZE cannot share a pass on gfx950) on ONE step.  Usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CMD="python bench.py --cpu-sample 0"   # the default bench command (two lanes of 32 batches, 192 steps)
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o kt -- $CMD > gpurun_out/prof_$tag.log 2>&1
# the other shapes: configs[2] (ONT) and configs[4] (SV), kernel trace only
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_ont -o kt -- python bench.py --shape ont --steps 48 --cpu-sample 0 > gpurun_out/prof_${tag}_ont.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_sv -o kt -- python bench.py --shape sv --steps 8 --coalesce 4 --cpu-sample 0 > gpurun_out/prof_${tag}_sv.log 2>&1
CMD1="python bench.py --steps 1 --warmup 0 --lanes 1 --coalesce 1 --cpu-sample 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$tag -o f -- $CMD1 > gpurun_out/pmc_fetch_$tag.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$tag -o w -- $CMD1 > gpurun_out/pmc_write_$tag.log 2>&1
grep -h '"metric"' gpurun_out/prof_$tag.log | tail -1 | cut -c1-200
ls -R gpurun_out | grep -c results.db
# LDS / issue counters of the same single step (SQ block, its own pass)
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d gpurun_out/pmc_sq_$tag -o s -- $CMD1 > gpurun_out/pmc_sq_$tag.log 2>&1
ls gpurun_out/pmc_sq_$tag 2>/dev/null | head -3; tail -3 gpurun_out/pmc_sq_$tag.log
