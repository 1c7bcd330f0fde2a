"""What does one key-frame cost when the window STAYS on the device?  

A 9-frame synthetic world at the named shape (BASELINE.json configs[2]: 1241x376, 2000 points per key-frame); the window holds 8 of the 9 frames.
Every step is one FullSystem::makeKeyFrame's worth of graph edits followed by the optimisation the key-frame triggers:
    removePoint x 2000 (the points of the oldest frame), marginalizeFrame (Schur complement on HM / bM), insertFrame (the missing frame: its raw
    image, 1.87 MB, pyramid level 0 built on the device), insertPoint x 2000, insertResidual x 28 000 (every surviving point -> the new frame, the
    new points -> the 7 other frames), makeIDX (the commit: device-side re-pack), setAdjointsF, setPrecalcValues, optimize(6 bodies), the tail's
    linearizeAll(true).
`reload` does the same steps through the whole-plane setters (what r04's value_window_upload_inclusive timed): every table and all 8 images again.
usage (GPU box): python tools/exp_keyframe_update.py [steps] [json-out]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdv_loam_amd import backend_api as api, synthetic as syn    # noqa: E402


def world(w=1241, h=376, nF=9, pts=2000, seed=0, calib=None):
    W = syn.make_window(w=w, h=h, nF=nF, pts_per_kf=pts, seed=seed, calib=calib or syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02)
    W.pts_of = [np.nonzero(W.host == f)[0] for f in range(nF)]
    return W


def residual_rows(W, pts, target):
    """index into W's residual arrays of the residuals (p, target) for p in pts (W is point-major, targets ascending without the host)"""
    nF = W.nF
    host = W.host[pts]
    rank = target - (target > host)
    return pts * (nF - 1) + rank


class ResidentWindow:
    def __init__(self, W, frames):
        self.W, self.win = W, list(frames)
        pts = np.concatenate([W.pts_of[f] for f in frames])
        n = 4 + 6 * len(frames)
        S = syn.subwindow(W, frames, pts, HM=W.HM[:n, :n], bM=W.bM[:n])
        self.G = api.EnergyFunctional(W.w, W.h, max_points=len(pts) + 4096).load(S, raw_images=True)
        self.ids = {f: np.arange(len(W.pts_of[0])) + k * len(W.pts_of[0]) for k, f in enumerate(frames)}     # ids of a setter-loaded window = dense indices
        self.S = S

    def step(self, its=6):
        W, G = self.W, self.G
        old = self.win[0]
        new = [f for f in range(W.nF) if f not in self.win][0]
        G.removePoints(self.ids.pop(old))
        G.removeFrame(0)
        self.win = self.win[1:] + [new]
        k = G.insertFrame(W.evalPT[new], W.state[new], W.state_zero[new], int(W.frameID[new]), 1.0, W.frameEnergyTH[new], image=W.images[new])
        pn = W.pts_of[new]
        ids_new = G.insertPoints(np.full(len(pn), k, np.int32), W.u[pn], W.v[pn], W.idepth[pn], W.idepth_zero[pn], W.color[pn], W.weights[pn],
                                 W.hasDepthPrior[pn], W.isFromSensor[pn])
        # every surviving point towards the new frame
        surv = self.win[:-1]
        ps = np.concatenate([W.pts_of[f] for f in surv])
        rr = residual_rows(W, ps, new)
        G.insertResiduals(np.concatenate([self.ids[f] for f in surv]), np.full(len(ps), k, np.int32), hasMatcher=W.r_hasMatcher[rr], matcher=W.r_matcher[rr])
        # the new points towards the other frames
        for t, f in enumerate(surv):
            rr = residual_rows(W, pn, f)
            G.insertResiduals(ids_new, np.full(len(pn), t, np.int32), hasMatcher=W.r_hasMatcher[rr], matcher=W.r_matcher[rr])
        self.ids[new] = ids_new
        G.makeIDX()
        G.setAdjointsF(); G.setPrecalcValues()
        tr = G.optimize(its, want_trace=False, fixed_its=True)
        G.optimize_finish()
        return tr


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W = world()
    RW = ResidentWindow(W, list(range(8)))
    RW.G.optimize(6, want_trace=False, fixed_its=True); RW.G.optimize_finish()
    for _ in range(2):
        RW.step()                                   # warm-up: scratch planes, staging buffers
    import torch
    torch.cuda.synchronize()
    t = []
    for _ in range(steps):
        t0 = time.perf_counter()
        RW.step()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    # the same key-frame through the whole-plane setters: load() of the window + optimize + tail
    n = 4 + 6 * 8
    S = syn.subwindow(W, RW.win, np.concatenate([W.pts_of[f] for f in RW.win]), HM=W.HM[:n, :n], bM=W.bM[:n])
    G2 = api.EnergyFunctional(W.w, W.h, max_points=S.nP + 4096)
    tl = []
    for _ in range(steps + 1):
        t0 = time.perf_counter()
        G2.load(S, raw_images=True); G2.optimize(6, want_trace=False, fixed_its=True); G2.optimize_finish()
        tl.append(time.perf_counter() - t0)
    tl = np.array(tl[1:]) * 1e3
    out = dict(steps=steps, its_per_step=6,
               keyframe_update_ms=dict(median=float(np.median(t)), min=float(t.min()), max=float(t.max())),
               value_keyframe_update_inclusive=float(6e3 / np.median(t)),
               reload_ms=dict(median=float(np.median(tl)), min=float(tl.min()), max=float(tl.max())),
               value_window_reload_inclusive=float(6e3 / np.median(tl)))
    print(json.dumps(out))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
