echo "== default"; python bench.py --cpu-sample 0 --steps 16 --warmup 16 2>&1 | tail -1 | cut -c1-120
echo "== HSA_SCRATCH_SINGLE_LIMIT=4GB"; HSA_SCRATCH_SINGLE_LIMIT=4000000000 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=8000000000 python bench.py --cpu-sample 0 --steps 16 --warmup 16 2>&1 | tail -1 | cut -c1-120
echo "== HSA_SCRATCH_SINGLE_LIMIT=32MB"; HSA_SCRATCH_SINGLE_LIMIT=32000000 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=32000000 python bench.py --cpu-sample 0 --steps 16 --warmup 16 2>&1 | tail -1 | cut -c1-120
