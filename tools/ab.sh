# A/B of two builds of the library on the same box: gpurun_out/libs/lib_<tag>.so copied over the in-tree one
for tag in "$@"; do
  cp gpurun_out/libs/lib_$tag.so longcalld_amd/liblcd_hotpath.so
  for c in 16 8; do
    echo "== $tag cap $c"
    LCD_LDS_CAP_KB=$c LCD_PROFILE_CHAINS=1 python bench.py --cpu-sample 0 --steps 16 --warmup 16 2>&1 | grep -E "group thr   64|\] total|class   64|\"value\"" | tail -6 | cut -c1-230
  done
done
