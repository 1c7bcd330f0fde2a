#!/bin/bash
# A/B of two BUILDS on one box: gpu_ab/A.so and gpu_ab/B.so are copied over sdv-loam_amd/libsdvgn.so in turn (the caller leaves the current build in B).  usage: tools/ab_libs.sh ROUNDS A B
R=${1:-4}; A=${2:-prev}; B=${3:-new}
export AMD_LOG_LEVEL=0
for r in $(seq 1 $R); do
    for v in $A $B; do
        cp gpu_ab/$v.so sdv-loam_amd/libsdvgn.so
        out=$(timeout 300 python bench.py --worker --steps 20 --warmup 5 --no-cpu --quick 2>/dev/null | tail -1)
        echo "$r [$v] $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(round(d['value']), d['ms_per_step'], round(d['roofline']['back_to_back_ms']*1e3,2))" "$out" 2>/dev/null)"
    done
done
cp gpu_ab/$B.so sdv-loam_amd/libsdvgn.so
