#!/bin/bash
# quick GPU check of the POA kernel: the POA kernel tests, then the driver's bench command with the per-phase chain profile.  Usage: bash tools/gpu_quick.sh <tag>
tag=${1:-q}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -k "poa" 2>&1 | tail -5
LCD_PROFILE_CHAINS=1 python bench.py --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
python -c "
import json; j=json.load(open('gpurun_out/${tag}.json')); print(j['value'], j['digest'], j['stage_ms'])"
grep -E "class|total|tail" gpurun_out/${tag}.err | tail -14
