#!/bin/bash
# builder lines of the two noisy shapes (configs[2] ONT, configs[4] SV), with the per-kind chain profile.  Usage: bash tools/gpu_shapes.sh <tag> [LCD_DBG value]
tag=${1:-shapes}; dbg=${2:-0}
mkdir -p gpurun_out
LCD_DBG=$dbg LCD_PROFILE_CHAINS=1 timeout 900 python bench.py --shape ont --steps 48 --cpu-sample 0 --f3 0 --overlap 0 --depth-profile 0 > gpurun_out/${tag}_ont.json 2> gpurun_out/${tag}_ont.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/${tag}_ont.json') if l.startswith('{')][-1]); print('ont', j['value'], j['digest'], j['stage_ms'])"
grep -E "^\[kind\]" gpurun_out/${tag}_ont.err | tail -6
LCD_DBG=$dbg LCD_PROFILE_CHAINS=1 timeout 900 python bench.py --shape sv --steps 8 --coalesce 4 --cpu-sample 0 --f3 0 --overlap 0 --depth-profile 0 > gpurun_out/${tag}_sv.json 2> gpurun_out/${tag}_sv.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/${tag}_sv.json') if l.startswith('{')][-1]); print('sv', j['value'], j['digest'], j['stage_ms'])"
grep -E "^\[kind\]" gpurun_out/${tag}_sv.err | tail -6
