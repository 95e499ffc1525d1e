#!/bin/bash
# same-box A/B of library variants with the stage times: tools/ab_stage.sh <out-prefix> <variant> ...   (exp/lib_<variant>.so; "cur" = the library in the tree).  Extra bench flags: AB_FLAGS
out=$1; shift
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_cur.so
for v in "$@"; do
  if [ "$v" = cur ]; then cp /tmp/lib_cur.so longcalld_amd/liblcd_hotpath.so; else cp exp/lib_$v.so longcalld_amd/liblcd_hotpath.so; fi
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --f3 0 --overlap 0 ${AB_FLAGS} > gpurun_out/${out}_$v.json 2> gpurun_out/${out}_$v.err
  python - "$v" gpurun_out/${out}_$v.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    dp=d.get('depth') or {}
    print(sys.argv[1], 'value', d['value'], d['digest'], 'stage', {k:round(v,1) for k,v in d['stage_ms'].items() if v}, 'depth', {k:v['ms_per_submission'] for k,v in dp.items()}, 'reps', d['repeats']['seconds'], flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[2].replace('.json','.err')).read()[-1500:])
PY
done
cp /tmp/lib_cur.so longcalld_amd/liblcd_hotpath.so
