#!/bin/bash
# where the wavefronts of one bench step spend their cycles (SQ block, one pass): parked at s_waitcnt / barrier (SQ_WAIT_ANY), issue-stalled
# (SQ_WAIT_INST_ANY), issuing (SQ_ACTIVE_INST_*).  Usage (GPU box, repo root): bash tools/pmc_stall.sh <tag> [bench flags]
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
CMD1="python bench.py --steps 1 --warmup 0 --lanes 1 --coalesce 1 --cpu-sample 0 $*"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d gpurun_out/pmc_stall_$tag -o s -- $CMD1 > gpurun_out/pmc_stall_$tag.log 2>&1
tail -2 gpurun_out/pmc_stall_$tag.log
python - <<PY
import sqlite3, glob, collections
db = glob.glob("gpurun_out/pmc_stall_$tag/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]; info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"select s.kernel_name, i.name, sum(e.value) from {pmc} e join {info} i on e.pmc_id = i.id join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by 1, 2"
acc = collections.defaultdict(dict)
for k, n, v in c.execute(q): acc[k.split('(')[0]][n] = v
with open("gpurun_out/pmc_stall_$tag.txt", "w") as f:
    for k, d in sorted(acc.items()):
        if "lcd_" not in k: continue
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        line = f"{k:40s} " + " ".join(f"{n[3:]}={v / wc:.3f}" for n, v in sorted(d.items()) if n != "SQ_WAVE_CYCLES") + f" WAVE_CYCLES={wc:.3g}"
        print(line); f.write(line + "\n")
PY
