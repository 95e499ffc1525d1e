#!/bin/bash
# on the GPU box: tools/ab_run.sh <out-prefix> <variant> [<variant> ...]   -- runs the driver's bench line with exp/lib_<variant>.so in place of the library
# (extra bench flags: AB_FLAGS; the library on disk is restored afterwards)
out=$1; shift
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_orig.so
for v in "$@"; do
  cp exp/lib_$v.so longcalld_amd/liblcd_hotpath.so
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --f3 0 --overlap 0 ${AB_FLAGS} > gpurun_out/${out}_$v.json 2> gpurun_out/${out}_$v.err
  python - "$v" gpurun_out/${out}_$v.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    dp=d.get('depth') or {}
    print(sys.argv[1], 'value', d['value'], 'ms_poa_kernel', d['roofline']['ms_per_launch'], 'depth', {k:v['ms_per_submission'] for k,v in dp.items()}, 'digest', d['digest'], 'reps', d['repeats']['seconds'], flush=True)
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[2].replace('.json','.err')).read()[-1500:])
PY
done
cp /tmp/lib_orig.so longcalld_amd/liblcd_hotpath.so
