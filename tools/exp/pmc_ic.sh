cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for co in 1 8; do
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH -d gpurun_out/pmc_ic$co -o ic -- python bench.py --steps $co --warmup 0 --lanes 1 --coalesce $co --cpu-sample 0 > gpurun_out/pmc_ic$co.log 2>&1
tail -3 gpurun_out/pmc_ic$co.log | cut -c1-200
ls gpurun_out/pmc_ic$co
done
