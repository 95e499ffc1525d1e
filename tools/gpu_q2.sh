#!/bin/bash
# quick check of a POA kernel change: the certified-band kernel tests, then the driver's submission once with the per-phase chain profile.  Usage: bash tools/gpu_q2.sh [tag]
tag=${1:-q}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "certified_band or solo" 2>&1 | tail -3
LCD_PROFILE_CHAINS=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --repeats 1 --depth-profile 0 > gpurun_out/$tag.json 2> gpurun_out/$tag.err
grep -E "total .* CU-s|class  256|class   64|slowest|row plan" gpurun_out/$tag.err | tail -9
python -c "import json; j=json.load(open('gpurun_out/$tag.json')); print(j['value'], j['digest'], j['stage_ms'])"
