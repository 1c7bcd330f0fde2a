"""Host-only microbenchmark of the fp64 stitch + solve (sdvgn_ef_stitch_solve_host on a host-only handle, no GPU needed).
SDVGN_PROFILE=1 prints the per-phase split (sdvgn_debug_phase_report)."""
import os, sys, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from sdv_loam_amd import api, parallel, synthetic as syn
W = syn.make_window(w=320, h=160, nF=8, pts_per_kf=200, seed=7, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))
L = api.load_library()
h = C.c_void_p()
assert L.sdvgn_ef_create(C.byref(h), -1, W.w, W.h, W.nP, None) == 0
c = np.ascontiguousarray
assert L.sdvgn_ef_set_calib(h, c(W.value_scaled, np.float64), c(W.value_minus_value_zero, np.float64)) == 0
assert L.sdvgn_ef_set_frames(h, W.nF, c(W.evalPT, np.float64).reshape(-1), c(W.state, np.float64).reshape(-1), c(W.state_zero, np.float64).reshape(-1),
                             c(W.frameID, np.int32), c(W.ab_exposure, np.float32), c(W.frameEnergyTH, np.float32)) == 0
assert L.sdvgn_ef_set_marg_prior(h, c(W.HM, np.float64).reshape(-1), c(W.bM, np.float64)) == 0
assert L.sdvgn_ef_set_adjoints(h) == 0 and L.sdvgn_ef_set_precalc(h) == 0
n = L.sdvgn_ef_accumulator_count(h)
rng = np.random.default_rng(0)
# PSD-ish accumulators: random but diagonal-dominant enough for LDLT not to blow up (timing only)
acc = rng.standard_normal(n) * 10.0
x = np.zeros(4 + 6 * W.nF)
xp = x.ctypes.data_as(C.c_void_p)
for _ in range(200): L.sdvgn_ef_stitch_solve_host(h, acc, 3, 0.1, xp)
N = 3000
t0 = time.perf_counter()
for _ in range(N): L.sdvgn_ef_stitch_solve_host(h, acc, 3, 0.1, xp)
dt = time.perf_counter() - t0
print("stitch+solve %.1f us/call" % (1e6 * dt / N), "x checksum", float(np.sum(x)))
L.sdvgn_debug_phase_report(N + 200)
