#!/bin/bash
# the builder's bench lines of a round's final build -> gpurun_out/<tag>_bench_*.json (copied to profiles/ afterwards).  Usage: bash tools/gpu_lines.sh <tag>
tag=$1
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2> gpurun_out/${tag}_bench_driver_flags.err
python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python bench.py --shape ont --steps 48 --cpu-sample 0 --f3 0 --overlap 0 > gpurun_out/${tag}_bench_ont.json 2> gpurun_out/${tag}_bench_ont.err
python bench.py --shape sv --steps 8 --cpu-sample 0 --f3 0 --overlap 0 > gpurun_out/${tag}_bench_sv.json 2> gpurun_out/${tag}_bench_sv.err
python bench.py --gpus 1 --steps 20 --warmup 5 --inproc 1 --cpu-sample 0 --f3 0 --overlap 0 > gpurun_out/${tag}_bench_inproc.json 2> gpurun_out/${tag}_bench_inproc.err
python bench.py --job-mb 100 --steps 1 --warmup 1 --cpu-sample 0 --f3 0 --overlap 0 > gpurun_out/${tag}_bench_job100mb.json 2> gpurun_out/${tag}_bench_job100mb.err
LCD_CERT=0 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --f3 0 --overlap 0 --depth-profile 0 > gpurun_out/${tag}_bench_driver_flags_cert_off.json 2> gpurun_out/${tag}_bench_driver_flags_cert_off.err
for f in driver_flags default ont sv inproc job100mb driver_flags_cert_off; do python - "$f" gpurun_out/${tag}_bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1])
    print(sys.argv[1], d['value'], d['digest'], d.get('ms_per_step'), (d.get('roofline') or {}).get('frac'), (d.get('device_memory') or {}).get('library_buffers_gb'), {k:v['ms_per_submission'] for k,v in (d.get('depth') or {}).items()})
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
