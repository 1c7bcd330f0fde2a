#!/bin/bash
# A/B on ONE box (boxes differ by ~2 %): bench workers under different environments, interleaved.  usage: tools/ab_bench.sh ROUNDS "ENV1" "ENV2" ...   ("-" = no variables)
R=${1:-3}; shift
export AMD_LOG_LEVEL=0
for r in $(seq 1 $R); do
    for v in "$@"; do
        vv=$v; [ "$v" = "-" ] && vv=""
        out=$(env $vv timeout 300 python bench.py --worker --steps 20 --warmup 5 --no-cpu --quick 2>/dev/null | tail -1)
        echo "$r [$v] $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(round(d['value']), d['ms_per_step'])" "$out" 2>/dev/null)"
    done
done
