"""Profiling experiment (not product): time k_ef_linearize under the SDVGN_DEBUG_FLAGS variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sdv_loam_amd import backend_api, synthetic as syn

W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)
for flags in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]:
    os.environ["SDVGN_DEBUG_FLAGS"] = str(flags)
    G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP, device=0).load(W)
    ext = torch.cuda.ExternalStream(G.stream(), device=torch.device("cuda", 0))
    for _ in range(10):
        G.linearizeAll(want_energy=False)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        with torch.cuda.stream(ext):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record(ext)
            for _ in range(50):
                G.linearizeAll(want_energy=False)
            b.record(ext)
        b.synchronize()
        best = min(best, a.elapsed_time(b) / 50)
    print("flags=%d  k_ef_linearize %.2f us/launch" % (flags, best * 1e3), flush=True)
    del G
