cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum -d gpurun_out/pmc_tcc -o t -- python bench.py --steps 1 --warmup 0 --lanes 1 --coalesce 1 --cpu-sample 0 > gpurun_out/pmc_tcc.log 2>&1
tail -2 gpurun_out/pmc_tcc.log | cut -c1-200
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum -d gpurun_out/pmc_tcp -o t -- python bench.py --steps 1 --warmup 0 --lanes 1 --coalesce 1 --cpu-sample 0 > gpurun_out/pmc_tcp.log 2>&1
tail -2 gpurun_out/pmc_tcp.log | cut -c1-200
