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
    body_s = {}
    for B in Bs:
        hs = [api.EnergyFunctional(W.w, W.h, max_points=W.nP, stream=api.EnergyFunctional.STREAM_OWN).load(W) for _ in range(B)]
        res = {m: float("nan") for m in ("sequential", "threads", "lockstep")}
        res.update({m: float("nan") for m in modes})
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
                elif mode.startswith("staggered"):
                    # two lock-step calls at a time (the library keeps two launch-sequence pools per device), the second `frac` of a body behind
                    import threading
                    frac = float(mode.split(":")[1]) if ":" in mode else 0.5
                    delay = frac * body_s.get(B // 2, 100e-6)
                    def second():
                        t1 = time.perf_counter() + delay
                        while time.perf_counter() < t1:
                            pass
                        api.optimize_lockstep(hs[B // 2:], 6, fixed_its=True, want_trace=False)
                    th = threading.Thread(target=second)
                    th.start()
                    api.optimize_lockstep(hs[:B // 2], 6, fixed_its=True, want_trace=False)
                    th.join()
                else:
                    api.optimize_lockstep(hs, 6, fixed_its=True, want_trace=False)
                torch.cuda.synchronize()
                if r:
                    tt += time.perf_counter() - t0
            res[mode] = 6 * B * reps / tt
            if mode == "lockstep":
                body_s[B] = tt / reps / 7.0          # ~7 linearise-equivalents per optimize(6) call
        print("B=%d  sequential %.0f  threads %.0f  lockstep %.0f it/s   lockstep/sequential %.2fx" % (B, res["sequential"], res["threads"], res["lockstep"], res["lockstep"] / res["sequential"]),
              "  ".join("%s %.0f (%.2fx)" % (m, res[m], res[m] / res["sequential"]) for m in modes if m.startswith("staggered")), flush=True)
        del hs

if __name__ == "__main__":
    main()
