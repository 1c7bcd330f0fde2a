"""B windows of the named size: sequential sdvgn_ef_optimize calls vs sdvgn_ef_optimize_batch as host threads vs the lock-step launch sequence."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sdv_loam_amd import backend_api as api, synthetic as syn

def main():
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, state_sigma=1e-3, idepth_sigma=0.01)
    Bs = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8,16").split(",")]
    reps = 6
    modes = (sys.argv[2] if len(sys.argv) > 2 else "sequential,threads,lockstep").split(",")
    for B in Bs:
        hs = [api.EnergyFunctional(W.w, W.h, max_points=W.nP, stream=api.EnergyFunctional.STREAM_OWN).load(W) for _ in range(B)]
        res = {m: float("nan") for m in ("sequential", "threads", "lockstep")}
        for mode in modes:
            if mode == "threads":
                os.environ["SDVGN_BATCH_THREADS"] = "1"
            else:
                os.environ.pop("SDVGN_BATCH_THREADS", None)
            tt = 0.0
            for r in range(reps + 1):
                for h in hs:
                    h.load(W)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if mode == "sequential":
                    for h in hs:
                        h.optimize(6, fixed_its=True, want_trace=False)
                elif mode == "threads":
                    api.optimize_batch(hs, 6, fixed_its=True)
                else:
                    api.optimize_lockstep(hs, 6, fixed_its=True, want_trace=False)
                torch.cuda.synchronize()
                if r:
                    tt += time.perf_counter() - t0
            res[mode] = 6 * B * reps / tt
        print("B=%d  sequential %.0f  threads %.0f  lockstep %.0f it/s   lockstep/sequential %.2fx" % (B, res["sequential"], res["threads"], res["lockstep"], res["lockstep"] / res["sequential"]), flush=True)
        del hs

if __name__ == "__main__":
    main()
