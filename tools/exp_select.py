import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdv_loam_amd import synthetic as syn, backend_api
W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)
G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
for _ in range(60):
    G.linearizeAll(want_energy=False)
G.state()
