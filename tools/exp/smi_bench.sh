( for i in $(seq 1 40); do sleep 0.3; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics" | sed 's/.*: //' | tr '\n' ' '; echo; done ) > gpurun_out/smi_samples.txt &
python bench.py --steps 24 --warmup 8 --lanes 1 --coalesce 8 --cpu-sample 0 2>&1 | tail -1 | cut -c1-100
wait
sort gpurun_out/smi_samples.txt | uniq -c | sort -rn | head -12
