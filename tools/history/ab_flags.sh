#!/bin/bash
# usage: tools/ab_flags.sh "0 8 0 8"   -- runs bench.py --quick --no-cpu once per SDVGN_DEBUG_FLAGS value and prints the key numbers
for f in $1; do
  SDVGN_DEBUG_FLAGS=$f python bench.py --quick --no-cpu 2>/dev/null | F=$f python -c '
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith("{")][-1])
k = d["kernel_ms"]
print("flags", os.environ["F"], "it/s", round(d["value"]), "median_us", round(d["iteration_us"]["median_us"], 1), "lin_us", round(k["k_ef_linearize_back_to_back"] * 1000, 2), "acc_us", round(k["accumulate(fused point+top+sc, reduce)"] * 1000, 2))'
done
