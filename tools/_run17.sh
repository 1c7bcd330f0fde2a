export AMD_LOG_LEVEL=0
mkdir -p gpurun_out/r05
run() { name=$1; shift; ( "$@" ) > gpurun_out/r05/fence3_$name.log 2>&1; echo "== $name rc=$?"; grep -aE "passed|failed|Memory access fault|^FAILED|^ERROR|Aborted" gpurun_out/r05/fence3_$name.log | head -4; }
SEL="tests/test_immature_gpu.py tests/test_window_update_gpu.py tests/test_dropin_gpu.py tests/test_backend_gpu.py tests/test_marginalize_gpu.py"
SDVGN_GUARD=1 SDVGN_GUARD_PAD_MB=2 SDVGN_GUARD_QUARANTINE=1 SDVGN_FENCE_EXTERNAL=1 run back_a16 timeout 900 python -X faulthandler -m pytest $SEL -q -m gpu -x -p no:cacheprovider
SDVGN_ALLOC_FILL=255 SDVGN_FREE_POISON=1 run fill_poison timeout 900 python -X faulthandler -m pytest $SEL -q -m gpu -x -p no:cacheprovider
bash tools/final_run.sh 2>&1 | grep -v "^+"
