# A/B of library builds on the same box: tools/ablibs/lib_<tag>.so copied over the in-tree one, plain bench value per build
for tag in "$@"; do
  cp tools/ablibs/lib_$tag.so longcalld_amd/liblcd_hotpath.so
  for rep in 1 2; do
    echo "== $tag: $(python bench.py --cpu-sample 0 --steps 64 --lanes 1 --coalesce 32 2>/dev/null | tail -1 | python -c 'import sys,json; j=json.load(sys.stdin); print(j["value"], j["stage_ms"]["ms_poa_kernel"])')"
  done
done
