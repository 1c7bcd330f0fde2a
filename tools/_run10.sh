export AMD_LOG_LEVEL=0
python - <<'PY' 2>&1 | grep -a "dropin\]\|it/s" | tail -12
import json, sys, os
sys.path.insert(0, '.')
os.environ["SDVGN_DROPIN_TIMING"] = "1"
from tools import bench_legs, exp_keyframe_update
W9 = exp_keyframe_update.world()
d = bench_legs.dropin_legs(W9, want_cpu=False)
for k, v in d.items():
    if isinstance(v, dict):
        print(k, round(v['its_per_s'], 1), 'it/s', round(v['ms_per_optimize_call'], 3), 'ms', {a: round(b, 1) for a, b in v.get('gpu_window', {}).items() if a.startswith('us_')})
PY
