export AMD_LOG_LEVEL=0
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_backend_gpu.py tests/test_window_update_gpu.py tests/test_marginalize_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -p no:cacheprovider > gpurun_out/r05/pytest_part.log 2>&1; grep -a "passed\|failed" gpurun_out/r05/pytest_part.log
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
from tools import bench_legs, exp_keyframe_update
W9 = exp_keyframe_update.world()
d = bench_legs.dropin_legs(W9, want_cpu=False)
print(json.dumps(d)[:1500])
PY
