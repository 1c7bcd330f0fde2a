# A/B of the headline between environment settings on one box: bash tools/ab_env.sh "VAR=1" ""  (3 alternating rounds; "" = default)
for round in 1 2 3; do
  for setting in "$@"; do
    env $setting timeout 200 python bench.py --no-cpu --quick 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$setting]', round(d['value']), 'it/s  ms/step %.5f' % d['ms_per_step'], ' body', d['iteration_us']['median_us'])"
  done
done
