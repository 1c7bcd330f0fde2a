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
    # the points of every frame as contiguous arrays (what the host loop holds when it activates them)
    W.pt_arrays = [tuple(np.ascontiguousarray(getattr(W, k)[W.pts_of[f]]) for k in ("u", "v", "idepth", "idepth_zero", "color", "weights", "hasDepthPrior", "isFromSensor"))
                   for f in range(nF)]
    try:     # the key-frame images in pinned host memory, like a host loop that feeds a GPU keeps them: the upload of insertFrame is then really asynchronous
        import torch
        W.images = [torch.from_numpy(np.ascontiguousarray(im, np.float32)).pin_memory().numpy() for im in W.images]
    except Exception:  # noqa: BLE001
        pass
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
        self.r_target, self.r_hm, self.r_m = {}, {}, {}
        self.host_k = {k: np.full(len(W.pts_of[0]), k, np.int32) for k in range(W.nF)}
        # the prior that is left when the oldest frame goes: in the real cycle marginalizePointsF has filled the frame's block of HM before
        # marginalizeFrame takes its Schur complement (a frame with an empty block would make that singular); here a fixed SPD prior
        n1 = n - 6
        self.HM1, self.bM1 = np.ascontiguousarray(W.HM[:n1, :n1]), np.ascontiguousarray(W.bM[:n1])

    def prepare(self):
        """what the host loop has at hand when the key-frame arrives (its own residual objects): target index, hasMatcher and matcher pixel of the
        28 000 residuals the step inserts, in the order the step sends them -- built OUTSIDE the timed region"""
        W = self.W
        new = [f for f in range(W.nF) if f not in self.win][0]
        surv = self.win[1:]
        key = (new, tuple(surv))
        if key in self.r_m:
            return
        k = len(surv)
        ps = np.concatenate([W.pts_of[f] for f in surv])
        pn = W.pts_of[new]
        rr = [residual_rows(W, ps, new)] + [residual_rows(W, pn, f) for f in surv]
        rr = np.concatenate(rr)
        self.r_target[new] = np.concatenate([np.full(len(ps), k, np.int32)] + [np.full(len(pn), t, np.int32) for t in range(k)])
        self.r_hm[key] = np.ascontiguousarray(W.r_hasMatcher[rr])
        self.r_m[key] = np.ascontiguousarray(W.r_matcher[rr])

    def step(self, its=6, prof=None):
        """prof: dict name -> list of host seconds per call (diagnostics: where a step's time goes; the calls are asynchronous, so a
        phase's GPU work shows up in the first later phase that waits for it)"""
        W, G = self.W, self.G
        t = [time.perf_counter()]

        def lap(name):
            if prof is not None:
                t.append(time.perf_counter())
                prof.setdefault(name, []).append(t[-1] - t[-2])
        old = self.win[0]
        new = [f for f in range(W.nF) if f not in self.win][0]
        G.removePoints(self.ids.pop(old))
        G.removeFrame(0, self.HM1, self.bM1)
        lap("removePoints + removeFrame")
        self.win = self.win[1:] + [new]
        k = G.insertFrame(W.evalPT[new], W.state[new], W.state_zero[new], int(W.frameID[new]), 1.0, W.frameEnergyTH[new], image=W.images[new])
        lap("insertFrame (1.87 MB image, async)")
        pn = W.pts_of[new]
        ids_new = G.insertPoints(self.host_k[k], *W.pt_arrays[new])
        lap("insertPoints x 2000")
        # insertResidual: every surviving point towards the new frame, the new points towards the 7 other frames -- one call
        surv = self.win[:-1]
        n_new = len(pn)
        pid = np.concatenate([self.ids[f] for f in surv] + [np.tile(ids_new, len(surv))])
        G.insertResiduals(pid, self.r_target[new], hasMatcher=self.r_hm[(new, tuple(surv))], matcher=self.r_m[(new, tuple(surv))])
        self.ids[new] = ids_new
        lap("insertResiduals x 28000 (incl. numpy gathers of the bench)")
        G.makeIDX()
        lap("makeIDX")
        G.setAdjointsF(); G.setPrecalcValues()
        lap("setAdjointsF + setPrecalcValues")
        tr = G.optimize(its, want_trace=False, fixed_its=True)
        lap("optimize(6)")
        G.optimize_finish()
        lap("optimize_finish")
        return tr


class CHostLoop:
    """the same steps from tools/kf_host_loop.c (plain C on include/sdvgn.h): what the cycle costs without a Python interpreter between the calls"""

    def __init__(self, RW):
        import ctypes as C
        self.C = C
        here = os.path.dirname(os.path.abspath(__file__))
        self.lib = C.CDLL(os.path.join(here, "libkfloop.so"))
        W = self.W = RW.W
        self.RW = RW
        F, P = W.nF, len(W.pts_of[0])
        assert all(len(p) == P for p in W.pts_of)
        f32, f64 = np.float32, np.float64
        keep = self.keep = {}
        keep["evalPT"] = np.ascontiguousarray(W.evalPT, f64); keep["state"] = np.ascontiguousarray(W.state, f64); keep["state_zero"] = np.ascontiguousarray(W.state_zero, f64)
        keep["frameID"] = np.ascontiguousarray(W.frameID, np.int32); keep["th"] = np.ascontiguousarray(W.frameEnergyTH, f32)
        for k, name in (("u", "u"), ("v", "v"), ("id", "idepth"), ("idz", "idepth_zero"), ("color", "color"), ("weights", "weights"), ("prior", "hasDepthPrior"),
                        ("sensor", "isFromSensor")):
            keep[k] = np.ascontiguousarray(np.concatenate([getattr(W, name)[W.pts_of[f]] for f in range(F)]))
        keep["HM1"], keep["bM1"] = np.ascontiguousarray(RW.HM1, f64), np.ascontiguousarray(RW.bM1, f64)
        img = (C.c_void_p * F)(*[im.ctypes.data for im in W.images])
        keep["img"] = img

        class KW(C.Structure):
            _fields_ = [("F", C.c_int), ("P", C.c_int), ("w", C.c_int), ("h", C.c_int)] + [(n, C.c_void_p) for n in (
                "evalPT7", "state10", "state_zero10", "frameID", "frameTH", "image", "u", "v", "idepth", "idepth_zero", "color8", "weights8", "prior", "sensor", "HM1", "bM1")]

        class KS(C.Structure):
            _fields_ = [("r_target", C.c_void_p), ("r_hasMatcher", C.c_void_p), ("r_matcher", C.c_void_p)]

        class KSt(C.Structure):
            _fields_ = [("win", C.c_int * 16), ("ids", C.c_void_p)]
        self.KS = KS
        p = lambda a: a.ctypes.data   # noqa: E731
        self.kw = KW(F, P, W.w, W.h, p(keep["evalPT"]), p(keep["state"]), p(keep["state_zero"]), p(keep["frameID"]), p(keep["th"]), C.cast(img, C.c_void_p),
                     p(keep["u"]), p(keep["v"]), p(keep["id"]), p(keep["idz"]), p(keep["color"]), p(keep["weights"]), p(keep["prior"]), p(keep["sensor"]),
                     p(keep["HM1"]), p(keep["bM1"]))
        self.ids = np.full((F, P), -1, np.int32)
        for f, a in RW.ids.items():
            self.ids[f] = a
        self.st = KSt()
        for k, f in enumerate(RW.win):
            self.st.win[k] = f
        self.st.ids = self.ids.ctypes.data
        self.lib.kf_host_loop.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]

    def run(self, n_steps, its=6):
        C, RW, W = self.C, self.RW, self.W
        win = [self.st.win[k] for k in range(W.nF - 1)]
        new_frames, steps, keep = [], (self.KS * n_steps)(), []
        for s in range(n_steps):
            RW.win = list(win)
            RW.prepare()
            new = [f for f in range(W.nF) if f not in win][0]
            key = (new, tuple(win[1:]))
            a, b, c = RW.r_target[new], RW.r_hm[key], RW.r_m[key]
            keep.append((a, b, c))
            steps[s] = self.KS(a.ctypes.data, b.ctypes.data, c.ctypes.data)
            new_frames.append(new)
            win = win[1:] + [new]
        nf = np.array(new_frames, np.int32)
        sec = np.zeros(n_steps)
        rc = self.lib.kf_host_loop(RW.G.h_, C.byref(self.kw), C.byref(self.st), n_steps, nf.ctypes.data, C.cast(steps, C.c_void_p), its, sec.ctypes.data)
        if rc:
            raise RuntimeError("kf_host_loop: sdvgn error %d" % rc)
        RW.win = win
        RW.ids = {f: self.ids[f].copy() for f in win}
        return sec


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    W = world()
    RW = ResidentWindow(W, list(range(8)))
    RW.G.optimize(6, want_trace=False, fixed_its=True); RW.G.optimize_finish()
    for _ in range(2):
        RW.prepare(); RW.step()                     # warm-up: scratch planes, staging buffers
    import torch
    torch.cuda.synchronize()
    t = []
    for _ in range(steps):
        RW.prepare()
        t0 = time.perf_counter()
        RW.step()
        t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e3
    prof = {}
    for _ in range(steps):
        RW.prepare()
        RW.step(prof=prof)
    phases = {k: float(np.median(v)) * 1e6 for k, v in prof.items()}
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
    ch = CHostLoop(RW)
    ch.run(2)
    tc = ch.run(steps) * 1e3
    out = dict(steps=steps, its_per_step=6, phases_host_us=phases,
               c_host_loop_ms=dict(median=float(np.median(tc)), min=float(tc.min()), max=float(tc.max())), value_keyframe_update_inclusive_c_host_loop=float(6e3 / np.median(tc)),
               keyframe_update_ms=dict(median=float(np.median(t)), min=float(t.min()), max=float(t.max())),
               value_keyframe_update_inclusive=float(6e3 / np.median(t)),
               reload_ms=dict(median=float(np.median(tl)), min=float(tl.min()), max=float(tl.max())),
               value_window_reload_inclusive=float(6e3 / np.median(tl)))
    print(json.dumps(out))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
