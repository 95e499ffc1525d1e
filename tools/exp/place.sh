LCD_PLACEMENT=1 python bench.py --cpu-sample 0 --steps 32 --warmup 32 2> gpurun_out/place.log | tail -1 | cut -c1-80
python - <<'PY'
import re,collections
rows=[l for l in open('gpurun_out/place.log') if l.startswith('[place]')]
# last submission only: take the last N rows where N = rows per submission (two submissions: warm-up and timed)
n=len(rows)//2; rows=rows[-n:]
cls=collections.defaultdict(list)
for l in rows:
    m=re.search(r'thr (\d+) xcc (\d+) se (\d+) sh (\d+) cu (\d+) simd (\d+)  t ([\d.]+)\.\.([\d.]+) ms',l)
    if m: cls[int(m.group(1))].append((float(m.group(7)),float(m.group(8)),tuple(int(m.group(i)) for i in (2,3,4,5))))
for c,v in sorted(cls.items()):
    t0=min(x[0] for x in cls[min(cls)]) if False else min(min(x[0] for x in vv) for vv in cls.values())
    st=[x[0]-t0 for x in v]; en=[x[1]-t0 for x in v]
    cus=len(set(x[2] for x in v))
    print(f"class {c}: {len(v)} chains on {cus} distinct CUs; first start {min(st):.0f} ms, last start {max(st):.0f} ms, last end {max(en):.0f} ms; mean chain {sum(e-s for s,e in zip(st,en))/len(v):.0f} ms")
PY
