cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d gpurun_out/trace_co -o tc -- python bench.py --steps 16 --warmup 16 --cpu-sample 0 > gpurun_out/trace_co.log 2>&1
tail -1 gpurun_out/trace_co.log | cut -c1-120
