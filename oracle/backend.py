"""ctypes front-end of the back-end oracle (oracle/orc_backend.cpp) -- TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import _Prefixed, f32p, f64p, i32p, lib

u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
vp = C.c_void_p
_bound = set()


def _L(raw=None, prefix="orc_"):
    L = _Prefixed(lib() if raw is None else raw, prefix)
    if True:
        key = (prefix, id(raw))        # one binding per library object: libref.so and libref_dropin.so share the prefix ref_
        if key in _bound:
            return L
        _bound.add(key)
        L.orc_ef_create.restype = vp
        L.orc_ef_create.argtypes = [C.c_int, C.c_int]
        L.orc_ef_destroy.argtypes = [vp]
        L.orc_ef_set_calib.argtypes = [vp, f64p, f64p]
        L.orc_ef_set_frames.argtypes = [vp, C.c_int, f64p, f64p, f64p, i32p, f32p, f32p]
        L.orc_ef_set_frame_state.argtypes = [vp, C.c_int, f64p]
        L.orc_ef_set_frame_image.argtypes = [vp, C.c_int, f32p]
        L.orc_ef_set_points.argtypes = [vp, C.c_int, i32p, f32p, f32p, f32p, f32p, f32p, f32p, u8p, u8p]
        L.orc_ef_set_point_idepth.argtypes = [vp, f32p, f32p]
        L.orc_ef_set_residuals.argtypes = [vp, C.c_int, i32p, i32p, i32p, u8p, f64p, u8p, u8p]
        L.orc_ef_set_marg_prior.argtypes = [vp, f64p, f64p]
        L.orc_ef_set_nullspaces.argtypes = [vp, C.c_int, f64p]
        L.orc_ef_compute_nullspaces.argtypes = [vp, vp]
        L.orc_ef_compute_nullspaces.restype = C.c_int
        L.orc_ef_fix_linearization.argtypes = [vp, u8p]
        L.orc_ef_fix_linearization.restype = None
        L.orc_ef_reset_oob.argtypes = [vp, vp]
        L.orc_ef_reset_oob.restype = None
        L.orc_ef_marginalize_points.argtypes = [vp, u8p, u8p]
        L.orc_ef_marginalize_points.restype = None
        L.orc_ef_marginalize_frame.argtypes = [vp, C.c_int, f64p, f64p]
        L.orc_ef_marginalize_frame.restype = None
        L.orc_ef_get_marg_prior.argtypes = [vp, f64p, f64p]
        L.orc_ef_get_marg_prior.restype = None
        L.orc_ef_get_res_toZero.argtypes = [vp, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), u8p]
        L.orc_ef_get_res_toZero.restype = None
        L.orc_ef_get_adHTdeltaF.argtypes = [vp, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")]
        L.orc_ef_get_adHTdeltaF.restype = None
        L.orc_ef_get_frame_prior.argtypes = [vp, C.c_int, f64p, f64p]
        L.orc_ef_get_frame_prior.restype = None
        L.orc_ef_set_fixed_its.argtypes = [vp, C.c_int]
        L.orc_ef_set_fixed_its.restype = None
        L.orc_ef_set_threads.argtypes = [vp, C.c_int]
        L.orc_ef_set_threads.restype = None
        L.orc_ef_set_precalc.argtypes = [vp]
        L.orc_ef_set_adjoints.argtypes = [vp]
        L.orc_ef_linearize_all.argtypes = [vp]
        L.orc_ef_linearize_all.restype = C.c_double
        L.orc_ef_apply_res.argtypes = [vp]
        L.orc_ef_solve_system.argtypes = [vp, C.c_int, C.c_double]
        L.orc_ef_dim.argtypes = [vp]
        L.orc_ef_dim.restype = C.c_int
        L.orc_ef_get_system.argtypes = [vp] + [vp] * 7
        L.orc_ef_get_residual_J.argtypes = [vp, C.c_int, f32p]
        L.orc_ef_get_residual_state.argtypes = [vp, vp, vp, vp, vp, vp]
        L.orc_ef_get_points.argtypes = [vp, f32p]
        L.orc_ef_get_frame_steps.argtypes = [vp, f64p, f64p]
        L.orc_ef_get_top_acc.argtypes = [vp, f32p]
        L.orc_ef_optimize_immature.argtypes = [vp, C.c_int, i32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, u8p, C.c_int, i32p, f32p, i32p]
        L.orc_ef_get_precalc.argtypes = [vp, C.c_int, C.c_int, f32p]
        L.orc_ef_get_adjoints.argtypes = [vp, f64p, f64p]
        L.orc_ef_res_in_A.argtypes = [vp]
        L.orc_ef_res_in_A.restype = C.c_int
        L.orc_ef_get_sc_acc.argtypes = [vp, f32p, f32p, f32p, f32p, f32p]
        L.orc_ef_optimize.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int]
        L.orc_ef_optimize.restype = C.c_int
        L.orc_ef_calc_L_energy.argtypes = [vp]
        L.orc_ef_calc_L_energy.restype = C.c_double
        L.orc_ef_calc_M_energy.argtypes = [vp]
        L.orc_ef_calc_M_energy.restype = C.c_double
        L.orc_ef_get_state.argtypes = [vp, f64p, f64p, f32p]
        L.orc_ef_optimize_finish.argtypes = [vp, f32p, i32p, u8p]
        L.orc_ef_optimize_finish.restype = C.c_double
        L.orc_ef_get_frame_energy_th.argtypes = [vp, f32p]
        L.orc_ef_get_frame_energy_th.restype = None
        L.orc_ef_get_evalPT.argtypes = [vp, C.c_int, f64p, f64p]
        L.orc_ef_get_evalPT.restype = None
    return L


class OracleEF:
    """Flattened EnergyFunctional window on the CPU oracle; method names follow the reference
    (EnergyFunctional.h:51-72, FullSystemOptimize.cpp)."""
    _raw, _prefix = None, "orc_"

    def __init__(self, w, h):
        self.L = _L(self._raw_lib(), self._prefix)
        self.w, self.h = w, h
        self.h_ = self.L.orc_ef_create(w, h)

    def __del__(self):
        try:
            self.L.orc_ef_destroy(self.h_)
        except Exception:
            pass

    @classmethod
    def _raw_lib(cls):
        return None

    def load(self, W):
        c = np.ascontiguousarray
        self.nF, self.nP, self.nR = W.nF, W.nP, W.nR
        self.L.orc_ef_set_calib(self.h_, c(W.value_scaled, np.float64), c(W.value_minus_value_zero, np.float64))
        self.L.orc_ef_set_frames(self.h_, W.nF, c(W.evalPT, np.float64).reshape(-1), c(W.state, np.float64).reshape(-1),
                                 c(W.state_zero, np.float64).reshape(-1), c(W.frameID, np.int32), c(W.ab_exposure, np.float32),
                                 c(W.frameEnergyTH, np.float32))
        for k in range(W.nF):
            self.L.orc_ef_set_frame_image(self.h_, k, c(W.pyr0[k], np.float32).reshape(-1))
        self.L.orc_ef_set_points(self.h_, W.nP, c(W.host, np.int32), c(W.u, np.float32), c(W.v, np.float32), c(W.idepth, np.float32),
                                 c(W.idepth_zero, np.float32), c(W.color, np.float32).reshape(-1), c(W.weights, np.float32).reshape(-1),
                                 c(W.hasDepthPrior, np.uint8), c(W.isFromSensor, np.uint8))
        self.L.orc_ef_set_residuals(self.h_, W.nR, c(W.r_point, np.int32), c(W.r_target, np.int32), c(W.r_state, np.int32),
                                    c(W.r_hasMatcher, np.uint8), c(W.r_matcher, np.float64).reshape(-1), c(W.r_isLinearized, np.uint8),
                                    c(W.r_isActive, np.uint8))
        self.L.orc_ef_set_marg_prior(self.h_, c(W.HM, np.float64).reshape(-1), c(W.bM, np.float64))
        if getattr(W, "nullspaces", None) is not None:
            ns = c(W.nullspaces, np.float64)
            self.L.orc_ef_set_nullspaces(self.h_, ns.shape[0], ns.reshape(-1))
        self.setAdjointsF()
        self.setPrecalcValues()
        return self

    def compute_nullspaces(self):
        """FullSystem::getNullspaces for the loaded frames: installs and returns the 6 pose + 1 scale vectors [7][4+6nF]."""
        out = np.zeros((7, 4 + 6 * self.nF))
        self.L.orc_ef_compute_nullspaces(self.h_, out.ctypes.data_as(C.c_void_p))
        return out

    def setPrecalcValues(self):
        self.L.orc_ef_set_precalc(self.h_)

    def setAdjointsF(self):
        self.L.orc_ef_set_adjoints(self.h_)

    def set_frame_state(self, idx, state10):
        self.L.orc_ef_set_frame_state(self.h_, idx, np.ascontiguousarray(state10, np.float64))

    def set_point_idepth(self, idepth, idepth_zero):
        self.L.orc_ef_set_point_idepth(self.h_, np.ascontiguousarray(idepth, np.float32), np.ascontiguousarray(idepth_zero, np.float32))

    def linearizeAll(self):
        return self.L.orc_ef_linearize_all(self.h_)

    def applyRes(self):
        self.L.orc_ef_apply_res(self.h_)

    def solveSystemF(self, iteration, lam):
        self.L.orc_ef_solve_system(self.h_, iteration, lam)

    @property
    def dim(self):
        return self.L.orc_ef_dim(self.h_)

    def system(self):
        n = self.dim
        out = dict(HA=np.zeros((n, n)), bA=np.zeros(n), Hsc=np.zeros((n, n)), bsc=np.zeros(n), HFinal=np.zeros((n, n)),
                   bFinal=np.zeros(n), x=np.zeros(n))
        self.L.orc_ef_get_system(self.h_, *[out[k].ctypes.data_as(vp) for k in ("HA", "bA", "Hsc", "bsc", "HFinal", "bFinal", "x")])
        return out

    def residual_J(self, which):
        out = np.zeros((self.nR, 24), np.float32)
        self.L.orc_ef_get_residual_J(self.h_, which, out.reshape(-1))
        return out

    def residual_state(self):
        ss = np.zeros(self.nR, np.int32)
        sn = np.zeros(self.nR, np.int32)
        en = np.zeros(self.nR, np.float64)
        eo = np.zeros(self.nR, np.float64)
        ac = np.zeros(self.nR, np.uint8)
        self.L.orc_ef_get_residual_state(self.h_, ss.ctypes.data_as(vp), sn.ctypes.data_as(vp), en.ctypes.data_as(vp),
                                         eo.ctypes.data_as(vp), ac.ctypes.data_as(vp))
        return dict(state=ss, new_state=sn, new_energy=en, energy_with_outlier=eo, active=ac)

    def points(self):
        out = np.zeros((self.nP, 9), np.float32)
        self.L.orc_ef_get_points(self.h_, out.reshape(-1))
        return out

    def frame_steps(self):
        s = np.zeros(6 * self.nF)
        c4 = np.zeros(4)
        self.L.orc_ef_get_frame_steps(self.h_, s, c4)
        return s.reshape(self.nF, 6), c4

    def optimizeImmature(self, host, u, v, idepth_min, idepth_max, energyTH, color, weights, isFromSensor, minObs=1):
        host = np.ascontiguousarray(host, np.int32)
        n = len(host)
        f = lambda a: np.ascontiguousarray(a, np.float32).reshape(-1)   # noqa: E731
        result = np.zeros(n, np.int32)
        idepth = np.zeros(n, np.float32)
        rs = np.zeros((n, self.nF), np.int32)
        self.L.orc_ef_optimize_immature(self.h_, n, host, f(u), f(v), f(idepth_min), f(idepth_max), f(energyTH), f(color), f(weights),
                                        np.ascontiguousarray(isFromSensor, np.uint8), minObs, result, idepth, rs.reshape(-1))
        return result, idepth, rs

    def top_acc(self):
        out = np.zeros((self.nF * self.nF, 13, 13), np.float32)
        self.L.orc_ef_get_top_acc(self.h_, out.reshape(-1))
        return out

    def precalc(self, h, t):
        out = np.zeros(27, np.float32)
        self.L.orc_ef_get_precalc(self.h_, h, t, out)
        return out

    def adjoints(self):
        a = np.zeros(self.nF * self.nF * 36)
        b = np.zeros(self.nF * self.nF * 36)
        self.L.orc_ef_get_adjoints(self.h_, a, b)
        return a.reshape(-1, 6, 6), b.reshape(-1, 6, 6)

    def resInA(self):
        return self.L.orc_ef_res_in_A(self.h_)

    # ---- marginalisation (EnergyFunctionalStructs.cpp:45-55, EnergyFunctional.cpp:434-576) ----
    def fixLinearization(self, mask):
        self.L.orc_ef_fix_linearization(self.h_, np.ascontiguousarray(mask, np.uint8))

    def resetOOB(self, mask=None):
        self.L.orc_ef_reset_oob(self.h_, None if mask is None else np.ascontiguousarray(mask, np.uint8).ctypes.data_as(vp))

    def marginalizePoints(self, marg, drop=None):
        marg = np.ascontiguousarray(marg, np.uint8)
        drop = np.zeros_like(marg) if drop is None else np.ascontiguousarray(drop, np.uint8)
        self.L.orc_ef_marginalize_points(self.h_, marg, drop)

    def marginalizeFrame(self, idx):
        n = self.dim - 6
        HM, bM = np.zeros((n, n)), np.zeros(n)
        self.L.orc_ef_marginalize_frame(self.h_, int(idx), HM.reshape(-1), bM)
        return HM, bM

    def marg_prior(self):
        n = self.dim
        HM, bM = np.zeros((n, n)), np.zeros(n)
        self.L.orc_ef_get_marg_prior(self.h_, HM.reshape(-1), bM)
        return HM, bM

    def res_toZero(self):
        out = np.zeros((self.nR, 2), np.float32)
        lin = np.zeros(self.nR, np.uint8)
        self.L.orc_ef_get_res_toZero(self.h_, out.reshape(-1), lin)
        return out, lin

    def adHTdeltaF(self):
        out = np.zeros((self.nF * self.nF, 6), np.float32)
        self.L.orc_ef_get_adHTdeltaF(self.h_, out.reshape(-1))
        return out

    def frame_prior(self, idx):
        pr, dp = np.zeros(6), np.zeros(6)
        self.L.orc_ef_get_frame_prior(self.h_, int(idx), pr, dp)
        return pr, dp

    def set_threads(self, n):
        """n > 1: the reference's multiThreading=true paths (IndexThreadReduce, NUM_THREADS=6 in the reference) -- timing baseline
        and tolerance-level results; 1 (default) is the reference's default and the parity configuration."""
        self.L.orc_ef_set_threads(self.h_, int(n))

    def optimize(self, its=6, cap=128, fixed_its=False):
        """fixed_its: exactly `its` loop bodies (bench; like flags bit0 of sdvgn_ef_optimize)."""
        self.L.orc_ef_set_fixed_its(self.h_, 1 if fixed_its else 0)
        stride = 8 + self.dim   # ..., x[dim], frameEnergyTH of the newest frame after the trial linearizeAll
        trace = np.zeros((cap, stride))
        n = self.L.orc_ef_optimize(self.h_, its, trace.ctypes.data_as(vp), stride, cap)
        return trace[:n]

    def optimize_finish(self):
        """Tail of FullSystem::optimize (FullSystemOptimize.cpp:460-470): setEvalPT on the newest frame, adjoints, precalc,
        linearizeAll(true).  Returns (lastEnergy[0], relbs_max[nP], ngood_inc[nP], removed[nR])."""
        rb = np.zeros(self.nP, np.float32)
        ng = np.zeros(self.nP, np.int32)
        rm = np.zeros(self.nR, np.uint8)
        e = self.L.orc_ef_optimize_finish(self.h_, rb, ng, rm)
        return e, rb, ng, rm

    def frame_energy_th(self):
        th = np.zeros(self.nF, np.float32)
        self.L.orc_ef_get_frame_energy_th(self.h_, th)
        return th

    def evalPT(self, idx):
        p, z = np.zeros(7), np.zeros(10)
        self.L.orc_ef_get_evalPT(self.h_, int(idx), p, z)
        return p, z

    def calcLEnergy(self):
        return self.L.orc_ef_calc_L_energy(self.h_)

    def calcMEnergy(self):
        return self.L.orc_ef_calc_M_energy(self.h_)

    def state(self):
        vs = np.zeros(4)
        st = np.zeros(10 * self.nF)
        idp = np.zeros(self.nP, np.float32)
        self.L.orc_ef_get_state(self.h_, vs, st, idp)
        return vs, st.reshape(self.nF, 10), idp

    def sc_acc(self):
        nF = self.nF
        accE = np.zeros((nF * nF, 8, 4), np.float32)
        accEB = np.zeros((nF * nF, 8), np.float32)
        accD = np.zeros((nF ** 3, 8, 8), np.float32)
        Hcc = np.zeros((4, 4), np.float32)
        bc = np.zeros(4, np.float32)
        self.L.orc_ef_get_sc_acc(self.h_, accE.reshape(-1), accEB.reshape(-1), accD.reshape(-1), Hcc.reshape(-1), bc)
        return accE, accEB, accD, Hcc, bc


class RefEF(OracleEF):
    """The same window interface on the REFERENCE'S OWN code: oracle/_ref/libref.so = the reference's FullSystem / EnergyFunctional /
    AccumulatedTop- and SCHessian / Residuals / HessianBlocks translation units compiled unmodified (oracle/Makefile target `ref`), driven by
    oracle/ref_glue_ef.cpp.  `optimize_full` is the reference's FullSystem::optimize itself (loop + tail)."""
    _prefix = "ref_"

    @classmethod
    def _raw_lib(cls):
        from . import refpin
        L = refpin.ref_lib()
        if L is None:
            raise RuntimeError("oracle/_ref/libref.so has not been built (needs /root/reference; `make -C oracle ref`)")
        return L

    def __init__(self, w, h):
        super().__init__(w, h)
        R = self.L._L
        R.ref_ef_optimize_full.argtypes = [vp, C.c_int]
        R.ref_ef_optimize_full.restype = C.c_double
        R.ref_ef_last_log.argtypes = [vp, C.c_char_p, C.c_int]
        R.ref_ef_last_log.restype = C.c_int
        R.ref_ef_get_removed.argtypes = [vp, u8p]
        R.ref_ef_get_point_stats.argtypes = [vp, f32p, i32p]
        R.ref_ef_load_report.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        R.ref_ef_get_center_projected.argtypes = [vp, f32p]

    def optimize_full(self, its=6, min_its=None):
        """FullSystem::optimize(its) of the reference.  Returns (rmse, [(accepted, iteration, energy)...] parsed from its console output,
        removed[nR], log).  min_its: value of setting_minOptIterations during the call (= its: exactly `its` loop bodies);
        self.last_seconds = wall time of the optimize() call alone."""
        import re
        R = self.L._L
        R.ref_ef_last_seconds.restype = C.c_double
        R.ref_ef_set_min_its(-1 if min_its is None else int(min_its))
        rmse = R.ref_ef_optimize_full(self.h_, int(its))
        R.ref_ef_set_min_its(-1)
        self.last_seconds = R.ref_ef_last_seconds()
        n = R.ref_ef_last_log(self.h_, None, 0)
        buf = C.create_string_buffer(n + 1)
        R.ref_ef_last_log(self.h_, buf, n + 1)
        log = buf.value.decode(errors="replace")
        steps = [(m.group(1) == "ACCEPT", int(m.group(2)), float(m.group(3)))
                 for m in re.finditer(r"(ACCEPT|REJECT) (\d+) \(L [^)]*\): \tA\(([-0-9.einfa]+)\)", log)]
        removed = np.zeros(self.nR, np.uint8)
        R.ref_ef_get_removed(self.h_, removed)
        return rmse, steps, removed, log

    # ---- the key-frame cycle around optimize, driven through the reference's own members (oracle/ref_glue_ef.cpp) ----
    def set_levels(self, levels):
        """pyramid levels of this world, BEFORE load (default 1); keyframe_tail's setCoarseTrackingRef needs >= 2"""
        self.L._L.ref_ef_set_levels.argtypes = [vp, C.c_int]
        self.L._L.ref_ef_set_levels(self.h_, int(levels))
        return self

    def keyframe_tail(self, its=6, flag_frames=None, min_its=None):
        """FullSystem::makeKeyFrame from its optimize call on (FullSystem.cpp:1133-1178): optimize, removeOutliers, setCoarseTrackingRef of the
        next key-frame's tracker, flagPointsForRemoval, dropPointsF, getNullspaces, marginalizePointsF, marginalizeFrame of the frames flagged
        in `flag_frames` (indices into the window).  Returns what optimize_full returns; afterwards nF is the window that is left, per-point
        getters give NaN for points that left."""
        import re
        R = self.L._L
        R.ref_ef_keyframe_tail.argtypes = [vp, C.c_int, u8p]
        R.ref_ef_keyframe_tail.restype = C.c_double
        R.ref_ef_window_frames.argtypes = [vp]
        R.ref_ef_last_seconds.restype = C.c_double
        fl = np.zeros(self.nF, np.uint8)
        if flag_frames is not None:
            fl[list(flag_frames)] = 1
        R.ref_ef_last_optimize_seconds.restype = C.c_double
        R.ref_ef_set_min_its(-1 if min_its is None else int(min_its))
        rmse = R.ref_ef_keyframe_tail(self.h_, int(its), fl)
        R.ref_ef_set_min_its(-1)
        self.last_seconds = R.ref_ef_last_seconds()
        self.last_optimize_seconds = R.ref_ef_last_optimize_seconds()
        self.nF = R.ref_ef_window_frames(self.h_)
        n = R.ref_ef_last_log(self.h_, None, 0)
        buf = C.create_string_buffer(n + 1)
        R.ref_ef_last_log(self.h_, buf, n + 1)
        log = buf.value.decode(errors="replace")
        steps = [(m.group(1) == "ACCEPT", int(m.group(2)), float(m.group(3)))
                 for m in re.finditer(r"(ACCEPT|REJECT) (\d+) \(L [^)]*\): \tA\(([-0-9.einfa]+)\)", log)]
        removed = np.zeros(self.nR, np.uint8)
        R.ref_ef_get_removed(self.h_, removed)
        return rmse, steps, removed, log

    def point_hosts(self):
        """index of every point's host frame in the window as it is now (-1: the point has left)"""
        R = self.L._L
        R.ref_ef_point_hosts.argtypes = [vp, i32p]
        out = np.zeros(self.nP, np.int32)
        R.ref_ef_point_hosts(self.h_, out)
        return out

    def tracking_ref(self, lvl=0):
        """the template setCoarseTrackingRef built for the next key-frame (CoarseTracker::makeCoarseDepthL0): (pc_n[5], u, v, idepth, color) of level lvl"""
        R = self.L._L
        R.ref_ef_tracking_ref.argtypes = [vp, C.c_int, i32p, f32p, f32p, f32p, f32p]
        n5 = np.zeros(5, np.int32)
        cap = self.w * self.h
        u, v, d, c = (np.zeros(cap, np.float32) for _ in range(4))
        R.ref_ef_tracking_ref(self.h_, int(lvl), n5, u, v, d, c)
        n = int(n5[lvl])
        return n5, u[:n], v[:n], d[:n], c[:n]

    def append_frame(self, evalPT7, state10, state_zero10, frameID, ab_exposure, frameEnergyTH, pyr0):
        """a new key-frame at the end of the window (makeKeyFrame's insertFrame, FullSystem.cpp:1071-1077)"""
        c = np.ascontiguousarray
        self.L.orc_ef_set_frames(self.h_, 1, c(evalPT7, np.float64).reshape(-1), c(state10, np.float64).reshape(-1), c(state_zero10, np.float64).reshape(-1),
                                 c([frameID], np.int32), c([ab_exposure], np.float32), c([frameEnergyTH], np.float32))
        self.L.orc_ef_set_frame_image(self.h_, self.nF, c(pyr0, np.float32).reshape(-1))
        self.nF += 1
        return self.nF - 1

    def append_points(self, host, u, v, idepth, idepth_zero, color, weights, hasDepthPrior, isFromSensor):
        """new points (activatePointsMT's insertPoint); host = index in the window as it is now.  Returns their indices (over all points ever set)."""
        c = np.ascontiguousarray
        n = len(host)
        self.L.orc_ef_set_points(self.h_, n, c(host, np.int32), c(u, np.float32), c(v, np.float32), c(idepth, np.float32), c(idepth_zero, np.float32),
                                 c(color, np.float32).reshape(-1), c(weights, np.float32).reshape(-1), c(hasDepthPrior, np.uint8), c(isFromSensor, np.uint8))
        self.nP += n
        return np.arange(self.nP - n, self.nP)

    def append_residuals(self, point, target, hasMatcher, matcher, state=None):
        """new residuals (insertResidual); point = index over all points ever set, target = index in the window as it is now"""
        c = np.ascontiguousarray
        n = len(point)
        st = np.zeros(n, np.int32) if state is None else c(state, np.int32)
        z = np.zeros(n, np.uint8)
        self.L.orc_ef_set_residuals(self.h_, n, c(point, np.int32), c(target, np.int32), st, c(hasMatcher, np.uint8), c(matcher, np.float64).reshape(-1), z, z)
        self.nR += n

    # ---- the per-frame rows around the window (oracle/ref_glue_ef.cpp): immature points, traceNewCoarse, activatePointsMT ----
    def add_immature(self, host, u, v, idepth_min, idepth_max, my_type=None, quality=None, status=None, interval=None, isFromSensor=None):
        """immature points by the reference's own constructor (FullSystem::makeNewTraces), with a given trace state; returns the number kept"""
        R = self.L._L
        c = np.ascontiguousarray
        n = len(host)
        R.ref_ef_add_immature.argtypes = [vp, C.c_int, i32p, i32p, i32p, f32p, f32p, f32p, f32p, i32p, f32p, u8p]
        R.ref_ef_add_immature.restype = C.c_int
        d = lambda a, v_, t: c(np.full(n, v_, t) if a is None else a, t)      # noqa: E731
        return int(R.ref_ef_add_immature(self.h_, n, c(host, np.int32), c(u, np.int32), c(v, np.int32), d(my_type, 1.0, np.float32), c(idepth_min, np.float32),
                                         c(idepth_max, np.float32), d(quality, 10000.0, np.float32), d(status, 5, np.int32), d(interval, 0.0, np.float32),
                                         d(isFromSensor, 0, np.uint8)))

    def trace_new_frame(self, image, camToWorld7, exposure=1.0, a=0.0, b=0.0):
        """FullSystem::traceNewCoarse (FullSystem.cpp:519-553) of every immature point of the window on a new frame"""
        R = self.L._L
        R.ref_ef_trace_new_frame.argtypes = [vp, f32p, f64p, C.c_float, C.c_double, C.c_double]
        R.ref_ef_trace_new_frame(self.h_, np.ascontiguousarray(image, np.float32).reshape(-1), np.ascontiguousarray(camToWorld7, np.float64), exposure, a, b)
        R.ref_ef_last_seconds.restype = C.c_double
        self.last_seconds = R.ref_ef_last_seconds()          # of the traceNewCoarse call alone

    def immature(self):
        """dict of the immature points' state in frameHessians / immaturePoints order (status -1: a slot the reference has emptied)"""
        R = self.L._L
        R.ref_ef_get_immature.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        R.ref_ef_get_immature.restype = C.c_int
        n = int(R.ref_ef_get_immature(self.h_, None, None, None, None, None, None, None, None, None))
        host, st = np.zeros(n, np.int32), np.zeros(n, np.int32)
        u, v, imin, imax, q, iv = (np.zeros(n, np.float32) for _ in range(6))
        uv = np.zeros((n, 2), np.float32)
        p = lambda a: a.ctypes.data_as(vp)      # noqa: E731
        if n:
            R.ref_ef_get_immature(self.h_, p(host), p(u), p(v), p(imin), p(imax), p(q), p(st), p(uv), p(iv))
        return dict(host=host, u=u, v=v, idepth_min=imin, idepth_max=imax, quality=q, status=st, lastTraceUV=uv, interval=iv)

    def activate_points(self):
        """FullSystem::activatePointsMT (FullSystem.cpp:569-717) + makeIDX; the new points join the point list.  Returns (number activated, u, v, host, target bit masks)"""
        R = self.L._L
        R.ref_ef_activate_points.argtypes = [vp]
        R.ref_ef_activate_points.restype = C.c_int
        R.ref_ef_get_new_points.argtypes = [vp, C.c_int, f32p, f32p, i32p, vp]
        R.ref_ef_num_points.argtypes = [vp]; R.ref_ef_num_residuals.argtypes = [vp]
        n = int(R.ref_ef_activate_points(self.h_))
        R.ref_ef_last_seconds.restype = C.c_double
        self.last_seconds = R.ref_ef_last_seconds()          # of the activatePointsMT call alone
        u, v, host, tg = np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.float32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.uint32)
        if n:
            R.ref_ef_get_new_points(self.h_, n, u, v, host, tg.ctypes.data_as(vp))
        self.nP = int(R.ref_ef_num_points(self.h_)); self.nR = int(R.ref_ef_num_residuals(self.h_))
        return n, u[:n], v[:n], host[:n], tg[:n]

    def point_stats(self):
        rb = np.zeros(self.nP, np.float32)
        ng = np.zeros(self.nP, np.int32)
        self.L._L.ref_ef_get_point_stats(self.h_, rb, ng)
        return rb, ng

    def load_report(self):
        cm, gd = C.c_int(0), C.c_double(0)
        self.L._L.ref_ef_load_report(self.h_, C.byref(cm), C.byref(gd))
        return cm.value, gd.value

    def center_projected(self):
        out = np.zeros((self.nR, 3), np.float32)
        self.L._L.ref_ef_get_center_projected(self.h_, out.reshape(-1))
        return out
