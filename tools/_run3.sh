export AMD_LOG_LEVEL=0
mkdir -p gpurun_out/r05
REF_GLUE_TRACE=1 timeout 1500 python -m pytest tests/test_dropin_gpu.py -x -q -m gpu -p no:cacheprovider -k "make_keyframe" 2>&1 | grep -v "python3(" | tail -40
timeout 900 python tools/exp_keyframe_update.py 8 gpurun_out/r05/keyframe_update.json 2>&1 | tail -5
