#!/bin/bash
# the bench lines of a round (one MI355X): driver's flags, default, ONT, SV, one 100 Mb job, PCIe-inclusive overlapped.  Usage: bash tools/round_numbers.sh <tag>
tag=${1:-rXX}
mkdir -p gpurun_out/lines
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/lines/${tag}_bench_driver_flags.json 2> gpurun_out/lines/${tag}_driver.err
python bench.py --cpu-sample 0 > gpurun_out/lines/${tag}_bench_default.json 2>/dev/null
python bench.py --shape ont --steps 48 --cpu-sample 0 > gpurun_out/lines/${tag}_bench_ont.json 2>/dev/null
python bench.py --shape sv --steps 8 --cpu-sample 0 > gpurun_out/lines/${tag}_bench_sv.json 2>/dev/null
python bench.py --job-mb 100 --cpu-sample 0 > gpurun_out/lines/${tag}_bench_job100mb.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --cpu-sample 0 --e2e 3 --coalesce 5 > gpurun_out/lines/${tag}_bench_e2e3.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --cpu-sample 0 --e2e 3 --coalesce 5 --vars 2 > gpurun_out/lines/${tag}_bench_e2e3_vars2.json 2>/dev/null
python bench.py --gpus 1 --inproc 1 --steps 20 --warmup 5 > gpurun_out/lines/${tag}_bench_inproc.json 2>/dev/null
LCD_CERT=0 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > gpurun_out/lines/${tag}_bench_driver_flags_cert_off.json 2>/dev/null
for f in gpurun_out/lines/${tag}_*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], j["value"], j.get("digest"), (j.get("stage_ms") or {}).get("ms_total"), (j.get("pcie_inclusive") or {}).get("overlapped"), j.get("depth"), (j.get("device_memory") or {}))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
