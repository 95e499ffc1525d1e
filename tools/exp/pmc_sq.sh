cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/pmc_sq -o sq -- python bench.py --steps 1 --warmup 0 --lanes 1 --cpu-sample 0 > gpurun_out/pmc_sq.log 2>&1
ls -R gpurun_out/pmc_sq | head
