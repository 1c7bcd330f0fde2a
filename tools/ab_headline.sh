# A/B of the headline between library builds on one box: bash tools/ab_headline.sh tagA tagB ... (tools/ab_libs/libsdvgn_<tag>.so), 3 alternating rounds
cp sdv-loam_amd/libsdvgn.so /tmp/libsdvgn_keep.so
for round in 1 2 3; do
  for tag in "$@"; do
    cp tools/ab_libs/libsdvgn_$tag.so sdv-loam_amd/libsdvgn.so
    timeout 200 python bench.py --no-cpu --quick 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', round(d['value']), 'it/s  ms/step %.5f' % d['ms_per_step'], ' body', d['iteration_us']['median_us'])"
  done
done
cp /tmp/libsdvgn_keep.so sdv-loam_amd/libsdvgn.so
