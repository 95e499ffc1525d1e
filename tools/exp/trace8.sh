cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
GPU_MAX_HW_QUEUES=16 rocprofv3 --kernel-trace -d gpurun_out/trace8 -o t8 -- python bench.py --steps 24 --warmup 8 --lanes 8 --cpu-sample 0 > gpurun_out/trace8.log 2>&1
tail -1 gpurun_out/trace8.log | cut -c1-120
ls -la gpurun_out/trace8
