# A/B of library builds on the same box: plain bench value per build (tools/exp/libs/lib_<tag>.so)
cp longcalld_amd/liblcd_hotpath.so /tmp/lib_keep.so
for tag in "$@"; do
  cp tools/exp/libs/lib_$tag.so longcalld_amd/liblcd_hotpath.so
  for r in 1 2; do echo "== $tag: $(python bench.py --cpu-sample 0 2>&1 | tail -1 | cut -c1-62)"; done
done
cp /tmp/lib_keep.so longcalld_amd/liblcd_hotpath.so
