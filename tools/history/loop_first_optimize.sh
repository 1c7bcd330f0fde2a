#!/bin/bash
# N fresh processes, each: the first tests of tests/test_backend_gpu.py up to the first optimize call with rejected steps
N=${1:-20}
mkdir -p gpurun_out/loop
fails=0
for i in $(seq 1 $N); do
    python -m pytest tests/test_backend_gpu.py -x -q -m gpu --tb=line -k "${K:-test_linearize_apply_solve_parity or test_images_built or test_nullspace or test_edge_cases or test_full_size_cfg3 or test_full_size_shard or test_optimize_loop_parity}" > gpurun_out/loop/first_$i.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "run $i rc=$rc"; grep -n "sdvgn\]" gpurun_out/loop/first_$i.log | head -8; fi
done
echo "runs=$N failed=$fails"
