"""bench.py's tracker leg alone (no back end, no CPU legs, no PMC passes): python tools/exp_tracker_extras.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SDVGN_BENCH_NO_PMC", "1")
import torch  # noqa: E402
import bench  # noqa: E402

import oracle  # noqa: E402
out = bench.tracker_extras(torch, 0, 4096, oracle, False)
print(json.dumps(out, indent=1))
