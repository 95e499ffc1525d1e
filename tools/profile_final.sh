cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/lines
tag=${1:-r05_v3}
CMD="python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0"
PMC="$CMD --repeats 1 --depth-profile 0 --overlap 0"
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o kt -- $PMC > gpurun_out/prof_$tag.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch_$tag -o f -- $PMC > gpurun_out/pmc_fetch_$tag.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write_$tag -o w -- $PMC > gpurun_out/pmc_write_$tag.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES -d gpurun_out/pmc_sq_$tag -o s -- $PMC > gpurun_out/pmc_sq_$tag.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d gpurun_out/pmc_stall_$tag -o s -- $PMC > gpurun_out/pmc_stall_$tag.log 2>&1; echo stall_rc=$?
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/lines/${tag}_bench_driver_flags.json 2> gpurun_out/lines/${tag}_driver.err; echo driver_rc=$?
tail -c 400 gpurun_out/lines/${tag}_bench_driver_flags.json | head -c 200; echo
find gpurun_out -name "*_results.db" | grep $tag
