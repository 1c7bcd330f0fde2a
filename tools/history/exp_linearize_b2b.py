#!/usr/bin/env python3
"""k_ef_linearize alone, back to back, under the SDVGN_DEBUG_FLAGS of the environment: average launch duration (one HIP event pair around
50 launches on the library's stream).  usage (GPU box): for f in 0 128; do SDVGN_DEBUG_FLAGS=$f python tools/exp_linearize_b2b.py; done"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from sdv_loam_amd import backend_api, synthetic as syn  # noqa: E402

W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)
G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
ext = torch.cuda.ExternalStream(G.stream())
G.launch_linearize_only(5)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(ext)
    G.launch_linearize_only(50)
    b.record(ext)
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 50)
print("SDVGN_DEBUG_FLAGS=%s  k_ef_linearize back to back: %.2f us per launch (best of 5 x 50)" % (os.environ.get("SDVGN_DEBUG_FLAGS", "0"), 1e3 * best))


def b2b(tag):
    best, worst = 1e9, 0
    for rep in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ext)
        G.launch_linearize_only(50)
        b.record(ext)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 50)
        worst = max(worst, a.elapsed_time(b) / 50)
    print("%-60s %.2f us (best of 5 x 50; worst %.2f)" % (tag, 1e3 * best, 1e3 * worst))


if len(sys.argv) > 1 and sys.argv[1] == "stepped":   # the same measurement on the state the optimize loop leaves behind (is the in-loop time a matter of the data?)
    import bench
    W2 = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **bench.HEAD_KW)
    G.load(W2)
    b2b("headline window, fresh")
    for k in range(3):
        G.optimize(6, fixed_its=True, want_trace=False)
        b2b("headline window, after %d x optimize(6)" % (k + 1))
