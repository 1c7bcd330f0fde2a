export AMD_LOG_LEVEL=0
mkdir -p gpurun_out/r05
timeout 600 python tools/exp_keyframe_update.py 8 gpurun_out/r05/keyframe_update.json > gpurun_out/r05/kf_exp.log 2>&1; grep -a "sdvgn\|Error\|value_" gpurun_out/r05/kf_exp.log | cut -c1-1100
python - <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
from sdv_loam_amd import api
L = api.load_library()
L.sdvgn_debug_copy_rate.restype = C.c_double; L.sdvgn_debug_copy_rate.argtypes = [C.c_size_t, C.c_int]
for sz in (1 << 28, 1 << 30, 1 << 31):
    print("copy kernel best of 4 shapes, %d MiB each way: %.0f GB/s" % (sz >> 20, L.sdvgn_debug_copy_rate(sz, 8)))
PY
timeout 900 python -m pytest tests/test_backend_gpu.py tests/test_window_update_gpu.py tests/test_marginalize_gpu.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -3
