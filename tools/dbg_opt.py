import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, 'tests')
import numpy as np
from sdv_loam_amd import synthetic as syn, backend_api as api
from oracle.backend import OracleEF
import oracle; oracle.build()
def low(W):
    W = copy.copy(W); W.frameEnergyTH = np.concatenate([np.linspace(150, 200, W.nF - 1), [300]]).astype(np.float32); return W
for lowth in (False, True):
    W = syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=2, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))
    if lowth: W = low(W)
    G = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W); O = OracleEF(W.w, W.h).load(W)
    tg, to = G.optimize(6), O.optimize(6)
    n = G.dim
    print("len", len(tg), len(to))
    m = min(len(tg), len(to))
    print("ctl g", tg[:m, [0,1,2,6]].tolist()); print("ctl o", to[:m, [0,1,2,6]].tolist())
    print("E g", tg[:m,3:6].tolist()); print("E o", to[:m,3:6].tolist())
    print("TH g", tg[:m, 7+n].tolist()); print("TH o", to[:m, 7+n].tolist())
    print("th final", G.frame_energy_th(), O.frame_energy_th())
    rg, ro = G.residual_state(), O.residual_state()
    print("state diff", (rg["state"] != ro["state"]).sum(), "active diff", (rg["active"] != ro["active"]).sum())
    e, rb, ng, rm = G.optimize_finish(); eo, rbo, ngo, rmo = O.optimize_finish()
    d = rb != rbo
    print("finish", e, eo, d.sum(), np.abs(rb - rbo).max(), rb[d][:5], rbo[d][:5])
