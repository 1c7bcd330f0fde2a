"""The oracle reproduces the committed golden vectors (regression pin; CPU only)."""
import numpy as np

from golden_util import load_marginalize, load_tracker, load_window, setup_tracker


def test_tracker_oracle_matches_golden(orc):
    g = load_tracker()
    T = setup_tracker(orc.OracleTracker(160, 120, 3), g)
    for l in range(3):
        assert np.array_equal(T.get_pyr(l)[1:-1], g["pyr%d" % l][1:-1])
        r = T.calcRes(l, g["start"], 0.02, 1.0, 20.0)
        assert np.allclose(r, g["res%d" % l], rtol=1e-12, equal_nan=True)
        assert np.array_equal(T.warped(), g["warped%d" % l])
        H, b = T.calcGS(l, 0.02, 1.0)
        assert np.allclose(H, g["H%d" % l], rtol=1e-12) and np.allclose(b, g["b%d" % l], rtol=1e-12)
    ok, pose, aff, last_res, flow, trace = T.trackNewestCoarse(g["start"], (0.0, 0.0), 2)
    assert ok == bool(g["track_ok"])
    assert np.allclose(pose, g["track_pose"], rtol=1e-10, atol=1e-12) and np.allclose(aff, g["track_aff"], rtol=1e-10)
    assert np.allclose(trace, g["track_trace"], rtol=1e-9, atol=1e-12)


def test_backend_oracle_matches_golden(orc):
    from oracle.backend import OracleEF
    W, g = load_window()
    E = OracleEF(W.w, W.h).load(W)
    assert np.isclose(E.linearizeAll(), float(g["energy"]), rtol=1e-12)
    st = E.residual_state()
    assert np.array_equal(st["new_state"], g["new_state"]) and np.array_equal(st["new_energy"], g["new_energy"])
    assert np.array_equal(E.residual_J(0), g["Jnew"])
    E.applyRes()
    E.solveSystemF(0, 0.1)
    s = E.system()
    for k in ("HA", "bA", "Hsc", "bsc", "HFinal", "bFinal", "x"):
        assert np.allclose(s[k], g[k], rtol=1e-10, atol=1e-300), k
    assert np.array_equal(E.points(), g["points"])
    E2 = OracleEF(W.w, W.h).load(W)
    assert np.allclose(E2.optimize(6), g["opt_trace"], rtol=1e-9, atol=1e-12)


def test_struct_pose_oracle_matches_golden(orc):
    from golden_util import load_struct_pose
    g, calib, args = load_struct_pose()
    T = orc.OracleTracker(int(g["w"]), int(g["h"]), 3)
    T.makeK(**calib)
    H, b, e, n = T.structResHb(orc.se3_inverse(g["init"]), *args)
    assert n == int(g["num"]) and np.allclose(H, g["H"], rtol=1e-12) and np.allclose(b, g["b"], rtol=1e-12)
    assert np.isclose(e, float(g["energy"]), rtol=1e-12)
    pose, trace, fr = T.structPoseEstimation(g["init"], *args)
    assert np.allclose(pose, g["pose"], rtol=1e-12, atol=1e-14) and np.allclose(trace, g["trace"], rtol=1e-10, atol=1e-14)
    assert np.isclose(fr, float(g["final_res"]), rtol=1e-12)


def test_reproject_oracle_matches_golden(orc):
    from golden_util import load_reproject
    from oracle.reproject import OracleReprojector
    g, setup = load_reproject()
    O = setup(OracleReprojector(int(g["w"]), int(g["h"]), int(g["levels"])))
    a = (g["u"], g["v"], g["idepth"], g["host_idx"])
    px0, cell, q = O.project(*a)
    assert np.array_equal(px0, g["px0"]) and np.array_equal(cell, g["cell"]) and np.array_equal(q, g["quality"])
    ok, pm, lvl = O.find_match(*a, g["ref_idx"], g["type"], px0)
    assert np.array_equal(ok, g["success"]) and np.array_equal(lvl, g["level"]) and np.array_equal(pm[ok], g["px"][g["success"]])


def _trace_problem_from_golden():
    import os
    from golden_util import HERE
    from sdv_loam_amd import synthetic as syn
    g = np.load(os.path.join(HERE, "trace_small.npz"))

    class P:
        pass
    for k in ("u", "v", "energyTH", "gradH", "color", "weights", "host_idx", "KRKi", "Kt", "aff"):
        setattr(P, k, g[k])
    P.w, P.h, P.image = int(g["w"]), int(g["h"]), g["I"]
    P.dI = syn.pyramid_numpy(g["I"], 1)[0]
    n = len(P.u)
    init = dict(idepth_min=np.zeros(n, np.float32), idepth_max=np.full(n, np.nan, np.float32), quality=np.full(n, 10000, np.float32),
                status=np.full(n, 5, np.int32))
    return g, P, init


def test_trace_oracle_matches_golden(orc):
    from oracle.trace import trace_on
    g, P, init = _trace_problem_from_golden()
    s1 = trace_on(P, P.dI, init["idepth_min"], init["idepth_max"], init["quality"], init["status"])
    s2 = trace_on(P, P.dI, s1["idepth_min"], s1["idepth_max"], s1["quality"], s1["status"])
    for k in s1:
        assert np.array_equal(s1[k], g["s1_" + k], equal_nan=True) and np.array_equal(s2[k], g["s2_" + k], equal_nan=True), k


def _coarse_depth_golden():
    import os
    from golden_util import HERE
    return np.load(os.path.join(HERE, "coarse_depth_small.npz"))


def _check_template(T, g):
    T.makeK(60., 60., int(g["w"]) / 2 - 0.5, int(g["h"]) / 2 - 0.5)
    T.set_new_image(g["I"], 1.0)
    T.makeCoarseDepth(g["u"], g["v"], g["idepth"], g["weight"])
    for l in range(int(g["levels"])):
        r = T.get_ref(l)
        for k in r:
            assert np.array_equal(r[k], g["pc%d_%s" % (l, k)]), (l, k)


def test_coarse_depth_oracle_matches_golden(orc):
    g = _coarse_depth_golden()
    _check_template(orc.OracleTracker(int(g["w"]), int(g["h"]), int(g["levels"])), g)


def _immature_golden():
    import os
    from golden_util import HERE, load_window
    W, _ = load_window()
    g = np.load(os.path.join(HERE, "immature_small.npz"))
    args = (W.host, W.u, W.v, g["idepth_min"], g["idepth_max"], g["energyTH"], W.color, W.weights, W.isFromSensor, int(g["minObs"]))
    return W, g, args


def test_immature_oracle_matches_golden(orc):
    from oracle.backend import OracleEF
    W, g, args = _immature_golden()
    r = OracleEF(W.w, W.h).load(W).optimizeImmature(*args)
    assert np.array_equal(r[0], g["result"]) and np.array_equal(r[1], g["idepth"], equal_nan=True) and np.array_equal(r[2], g["res_state"])


def test_marginalize_oracle_matches_golden(orc):
    from oracle.backend import OracleEF
    W, _ = load_window()
    g = load_marginalize()
    W.idepth_zero = g["idepth_zero"]
    E = OracleEF(W.w, W.h).load(W)
    E.linearizeAll(); E.applyRes()
    E.fixLinearization(g["marg"])
    r2z, lin = E.res_toZero()
    assert np.array_equal(r2z, g["res_toZero"]) and np.array_equal(lin, g["isLinearized"])
    E.marginalizePoints(g["marg"], g["drop"])
    HM, bM = E.marg_prior()
    assert np.allclose(HM, g["HM"], rtol=1e-10, atol=1e-300) and np.allclose(bM, g["bM"], rtol=1e-10, atol=1e-300)
    for i in range(W.nF):
        Hf, bf = E.marginalizeFrame(i)
        assert np.allclose(Hf, g["HM_frame"][i], rtol=1e-9, atol=1e-12) and np.allclose(bf, g["bM_frame"][i], rtol=1e-9, atol=1e-12)
    E.solveSystemF(0, 0.1)
    assert E.resInA() == int(g["resInA_after"]) and np.allclose(E.system()["x"], g["x_after"], rtol=1e-9, atol=1e-14)
