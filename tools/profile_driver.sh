#!/bin/bash
# rocprofv3 runs of the DRIVER's exact bench command (one submission of 20 batches): kernel trace + stats, the two HBM PMC passes (FETCH_SIZE and WRITE_SIZE
# cannot share a pass on gfx950), the SQ issue / LDS counters and the SQ wait counters -- each in its own run (a counter run carries nothing else).
# Usage (on the GPU box, from the repo root): bash tools/profile_driver.sh <tag>     then: python tools/rocprof_summary.py <tag> "<title>" driver
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0"
# the counter passes take ONE timed repeat and no depth profile (the driver's line reports the median of five repeats of the same submission and adds lone submissions of
# 1 / 4 / 16 batches, and the PCIe-inclusive pipeline of --overlap runs more submissions): warm-up submission + timed submission, which is what tools/rocprof_summary.py's "second half of the dispatches" rule assumes
PMC="$CMD --repeats 1 --depth-profile 0 --overlap 0"
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o kt -- $PMC > gpurun_out/prof_$tag.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$tag -o f -- $PMC > gpurun_out/pmc_fetch_$tag.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$tag -o w -- $PMC > gpurun_out/pmc_write_$tag.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d gpurun_out/pmc_sq_$tag -o s -- $PMC > gpurun_out/pmc_sq_$tag.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d gpurun_out/pmc_stall_$tag -o s -- $PMC > gpurun_out/pmc_stall_$tag.log 2>&1
for f in prof pmc_fetch pmc_write pmc_sq pmc_stall; do grep -h '"metric"' gpurun_out/${f}_$tag.log | tail -1 | cut -c1-120; done
find gpurun_out -name "*_results.db" | grep $tag
