# A/B of library variants on the GPU box: for each build/libsdvgn_<tag>.so (and the product build as `base`), the in-loop trace of
# k_ef_linearize (bench.measure_inloop_kernel) and the headline at K = 60.   usage: bash tools/exp_variants.sh [tags...]
cp sdv-loam_amd/libsdvgn.so /tmp/libsdvgn_base.so
for tag in base "$@"; do
  if [ "$tag" = base ]; then cp /tmp/libsdvgn_base.so sdv-loam_amd/libsdvgn.so; else cp build/libsdvgn_$tag.so sdv-loam_amd/libsdvgn.so; fi
  echo "== $tag"
  timeout 300 python - <<'PY'
import json, bench
for a in (0, 1):
    r = bench.measure_inloop_kernel(arith=a)
    print("in-loop k_ef_linearize arith=%d:" % a, json.dumps(r))
    print(open("gpurun_out/inloop_trace_summary_arith%d.txt" % a).read()[:1400])
PY
  timeout 300 python bench.py --steps 60 --warmup 12 --no-cpu --quick 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', round(d['value']), 'it/s  body', d['iteration_us'], ' b2b', d['roofline'].get('back_to_back_ms'))"
done
cp /tmp/libsdvgn_base.so sdv-loam_amd/libsdvgn.so
