#!/bin/bash
# round-6 check of the per-read graph phases: (1) the verifying build (every incremental re-sort compared with the full one on the device) on the POA kernel tests and one
# short bench submission, (2) the product build: region / chain tests, then the bench line with the per-phase chain profile.  Usage: bash tools/gpu_inc.sh <tag> [more pytest args]
tag=${1:-inc}
mkdir -p gpurun_out
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
if [ -f exp/lib_vinc.so ]; then
  cp exp/lib_vinc.so longcalld_amd/liblcd_hotpath.so
  timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "poa" > gpurun_out/${tag}_vtest.log 2>&1
  timeout 600 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --repeats 1 --depth-profile 0 --f3 0 --overlap 0 > gpurun_out/${tag}_v.json 2> gpurun_out/${tag}_v.err
  grep -c "verify-inc" gpurun_out/${tag}_v.json gpurun_out/${tag}_vtest.log; grep "verify-inc" gpurun_out/${tag}_v.json | head -5; grep -E "passed|failed" gpurun_out/${tag}_vtest.log
  grep "\[inc\]" gpurun_out/${tag}_v.json | awk '{for(i=1;i<=NF;i++){if($i=="pre"){pre+=$(i+1)} if($i=="ml"){ml+=$(i+1)} if($i=="new"){nw+=$(i+1)} if($i=="cnt"){cnt+=$(i+1)} if($i=="cut"){cut+=$(i+1)} if($i=="cap"){cap+=$(i+1)} if($i=="q"){q+=$(i+1)} if($i=="dry"){dry+=$(i+1)} if($i=="el"){el+=$(i+1)} if($i=="left"){left+=$(i+1)} if($i=="ok"){ok+=$(i+1)} if($i=="walked"){wk+=$(i+1)} if($i=="pieces"){pc+=$(i+1)}}} END {print "inc stat: pre",pre,"ml",ml,"new",nw,"cnt",cnt,"cut",cut,"cap",cap,"q",q,"dry",dry,"el",el,"left",left,"| ok",ok,"walked",wk,"pieces",pc}'
  python -c "import json; j=json.loads([l for l in open('gpurun_out/${tag}_v.json') if l.startswith('{')][-1]); print('verify build', j['value'], j['digest'])"
fi
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_region.py -x -q -k "poa or region or batch" 2>&1 | tail -3
LCD_PROFILE_CHAINS=1 timeout 600 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --f3 0 --overlap 0 > gpurun_out/${tag}.json 2> gpurun_out/${tag}.err
python -c "import json; j=json.loads([l for l in open('gpurun_out/${tag}.json') if l.startswith('{')][-1]); print('product', j['value'], j['digest'], j['stage_ms'], {k:v['ms_per_submission'] for k,v in (j.get('depth') or {}).items()})"
grep -E "^\[kind\]" gpurun_out/${tag}.err | tail -3
