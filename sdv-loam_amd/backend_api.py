"""ctypes binding of the sdvgn_ef_* entry points (include/sdvgn.h) + a Python mirror of the EnergyFunctional surface."""
import ctypes as C

import numpy as np

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
vp = C.c_void_p

PROTOTYPES = [
    ("sdvgn_ef_create", C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    ("sdvgn_ef_destroy", None, [vp]),
    ("sdvgn_ef_stream", vp, [vp]),
    ("sdvgn_ef_set_calib", C.c_int, [vp, f64p, f64p]),
    ("sdvgn_ef_set_frames", C.c_int, [vp, C.c_int, f64p, f64p, f64p, i32p, f32p, f32p]),
    ("sdvgn_ef_set_frame_image", C.c_int, [vp, C.c_int, f32p]),
    ("sdvgn_ef_set_frame_image_raw", C.c_int, [vp, C.c_int, f32p]),
    ("sdvgn_ef_set_points", C.c_int, [vp, C.c_int, i32p, f32p, f32p, f32p, f32p, f32p, f32p, u8p, u8p]),
    ("sdvgn_ef_set_residuals", C.c_int, [vp, C.c_int, i32p, i32p, i32p, u8p, f64p, u8p, u8p]),
    ("sdvgn_ef_set_residual_jacobians", C.c_int, [vp, C.c_int, f32p, vp]),
    ("sdvgn_ef_set_marg_prior", C.c_int, [vp, f64p, f64p]),
    ("sdvgn_ef_set_nullspaces", C.c_int, [vp, C.c_int, f64p]),
    ("sdvgn_ef_compute_nullspaces", C.c_int, [vp]),
    ("sdvgn_ef_get_nullspaces", C.c_int, [vp, vp, C.c_int]),
    ("sdvgn_ef_set_precalc", C.c_int, [vp]),
    ("sdvgn_ef_make_resident", C.c_int, [vp]),
    ("sdvgn_ef_get_accepted_steps", C.c_int, [vp]),
    ("sdvgn_debug_launch_pattern", C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
    ("sdvgn_ef_set_adjoints", C.c_int, [vp]),
    ("sdvgn_ef_linearize_all", C.c_int, [vp, vp]),
    ("sdvgn_ef_apply_res", C.c_int, [vp]),
    ("sdvgn_ef_solve_system", C.c_int, [vp, C.c_int, C.c_double, vp]),
    ("sdvgn_ef_point_step", C.c_int, [vp, C.c_int, C.c_float]),
    ("sdvgn_ef_set_frame_states", C.c_int, [vp, f64p]),
    ("sdvgn_ef_dim", C.c_int, [vp]),
    ("sdvgn_ef_get_system", C.c_int, [vp, vp, vp, vp, vp, vp, vp]),
    ("sdvgn_ef_get_residual_J", C.c_int, [vp, C.c_int, f32p]),
    ("sdvgn_ef_get_residual_state", C.c_int, [vp, vp, vp, vp, vp, vp]),
    ("sdvgn_ef_get_points", C.c_int, [vp, f32p]),
    ("sdvgn_ef_get_point_nogood", C.c_int, [vp, vp]),
    ("sdvgn_ef_clear_error", C.c_int, [vp]),
    ("sdvgn_ef_get_top_acc", C.c_int, [vp, f64p, vp]),
    ("sdvgn_ef_get_iteration_times", C.c_int, [vp, vp, C.c_int]),
    ("sdvgn_debug_phase_report", C.c_int, [C.c_int]),
    ("sdvgn_ef_fix_linearization", C.c_int, [vp, vp]),
    ("sdvgn_ef_reset_oob", C.c_int, [vp, vp]),
    ("sdvgn_ef_marginalize_points", C.c_int, [vp, vp, vp]),
    ("sdvgn_ef_get_marg_prior", C.c_int, [vp, vp, vp]),
    ("sdvgn_ef_marginalize_frame", C.c_int, [vp, C.c_int, vp, vp]),
    ("sdvgn_ef_get_res_toZero", C.c_int, [vp, vp, vp]),
    ("sdvgn_debug_read_stamps", C.c_int, [vp, vp, C.c_int]),
    ("sdvgn_debug_loop_counters", C.c_int, [vp, vp]),
    ("sdvgn_debug_solve_stamps", C.c_int, [vp, vp]),
    ("sdvgn_ef_get_linearize_times", C.c_int, [vp, vp, C.c_int]),
    ("sdvgn_debug_launch_linearize", C.c_int, [vp, C.c_int]),
    ("sdvgn_ef_get_solve_status", C.c_int, [vp]),
    ("sdvgn_ef_set_arith", C.c_int, [vp, C.c_int]),
    ("sdvgn_ef_collective_stride", C.c_int, [vp]),
    ("sdvgn_ef_set_collective_buffer", C.c_int, [vp, vp, C.c_int]),
    ("sdvgn_ef_collective_count", C.c_ulonglong, [vp]),
    ("sdvgn_ef_optimize_batch", C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp]),
    ("sdvgn_ef_optimize_lockstep", C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int]),
    ("sdvgn_ef_frame_image_dev", vp, [vp, C.c_int]),
    ("sdvgn_rccl_unique_id", C.c_int, [vp]),
    ("sdvgn_ef_init_rccl", C.c_int, [vp, vp, C.c_int, C.c_int]),
    ("sdvgn_rccl_comm_alive", C.c_int, [C.c_char_p]),
    ("sdvgn_ef_rccl_ranks", C.c_int, [vp]),
    ("sdvgn_ef_optimize_immature", C.c_int, [vp, C.c_int, i32p, f32p, f32p, f32p, f32p, f32p, f32p, f32p, u8p, C.c_int, i32p, f32p, i32p]),
    ("sdvgn_ef_accumulators_dev", C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int)]),
    ("sdvgn_ef_accumulate", C.c_int, [vp]),
    ("sdvgn_ef_finish_solve", C.c_int, [vp, C.c_int, C.c_double, vp]),
    ("sdvgn_ef_set_host_range", C.c_int, [vp, C.c_int, C.c_int]),
    ("sdvgn_ef_stitch_solve_host", C.c_int, [vp, f64p, C.c_int, C.c_double, vp]),
    ("sdvgn_ef_accumulator_count", C.c_int, [vp]),
    ("sdvgn_ef_set_external_buffers", C.c_int, [vp, vp, C.c_int, vp, C.c_int]),
    ("sdvgn_ef_get_frame_energy_th", C.c_int, [vp, f32p]),
    ("sdvgn_ef_optimize_finish", C.c_int, [vp, vp, vp, vp, vp]),
    ("sdvgn_ef_set_allreduce", C.c_int, [vp, vp, vp]),
    ("sdvgn_ef_optimize", C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int]),
    ("sdvgn_ef_get_state", C.c_int, [vp, vp, vp, vp]),
    ("sdvgn_ef_insert_frame", C.c_int, [vp, f64p, f64p, f64p, C.c_int, C.c_float, C.c_float, vp, vp]),
    ("sdvgn_ef_remove_frame", C.c_int, [vp, C.c_int, vp, vp]),
    ("sdvgn_ef_update_frames", C.c_int, [vp, C.c_int, f64p, f64p, f64p, f32p]),
    ("sdvgn_ef_insert_points", C.c_int, [vp, C.c_int, i32p, f32p, f32p, f32p, f32p, f32p, f32p, u8p, u8p, vp]),
    ("sdvgn_ef_remove_points", C.c_int, [vp, C.c_int, i32p]),
    ("sdvgn_ef_insert_residuals", C.c_int, [vp, C.c_int, i32p, i32p, i32p, u8p, f64p]),
    ("sdvgn_ef_update_residuals", C.c_int, [vp, C.c_int, i32p, i32p, i32p, u8p, f64p]),
    ("sdvgn_ef_drop_residuals", C.c_int, [vp, C.c_int, i32p, i32p]),
    ("sdvgn_ef_make_idx", C.c_int, [vp]),
    ("sdvgn_ef_get_point_ids", C.c_int, [vp, vp]),
    ("sdvgn_ef_get_residual_table", C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp]),
    ("sdvgn_ef_get_look_ahead", C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
]


class EnergyFunctional:
    """Flattened EnergyFunctional window on the GPU; method names follow the reference
    (EnergyFunctional.h:51-72, FullSystemOptimize.cpp:99-159, 504-520)."""

    STREAM_OWN = 1      # SDVGN_STREAM_OWN: a HIP stream of the handle's own (windows optimised side by side, optimize_batch)

    def __init__(self, w, h, max_points, device=0, stream=None):
        from .api import check, load_library
        self.L = load_library()
        self._check = check
        self.w, self.h, self.max_points = w, h, max_points
        hnd = vp()
        check(self.L.sdvgn_ef_create(C.byref(hnd), device, w, h, max_points, stream))
        self.h_ = hnd

    def close(self):
        if getattr(self, "h_", None):
            self.L.sdvgn_ef_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calib(self, value_scaled, value_minus_value_zero):
        self._check(self.L.sdvgn_ef_set_calib(self.h_, np.ascontiguousarray(value_scaled, np.float64), np.ascontiguousarray(value_minus_value_zero, np.float64)))

    def load(self, W, raw_images=False):
        c = np.ascontiguousarray
        ck = self._check
        self.table_mode = False
        self.nF, self.nP, self.nR = W.nF, W.nP, W.nR
        ck(self.L.sdvgn_ef_set_calib(self.h_, c(W.value_scaled, np.float64), c(W.value_minus_value_zero, np.float64)))
        ck(self.L.sdvgn_ef_set_frames(self.h_, W.nF, c(W.evalPT, np.float64).reshape(-1), c(W.state, np.float64).reshape(-1),
                                      c(W.state_zero, np.float64).reshape(-1), c(W.frameID, np.int32), c(W.ab_exposure, np.float32),
                                      c(W.frameEnergyTH, np.float32)))
        for k in range(W.nF):
            if raw_images:
                ck(self.L.sdvgn_ef_set_frame_image_raw(self.h_, k, c(W.images[k], np.float32).reshape(-1)))
            else:
                ck(self.L.sdvgn_ef_set_frame_image(self.h_, k, c(W.pyr0[k], np.float32).reshape(-1)))
        ck(self.L.sdvgn_ef_set_points(self.h_, W.nP, c(W.host, np.int32), c(W.u, np.float32), c(W.v, np.float32), c(W.idepth, np.float32),
                                      c(W.idepth_zero, np.float32), c(W.color, np.float32).reshape(-1), c(W.weights, np.float32).reshape(-1),
                                      c(W.hasDepthPrior, np.uint8), c(W.isFromSensor, np.uint8)))
        ck(self.L.sdvgn_ef_set_residuals(self.h_, W.nR, c(W.r_point, np.int32), c(W.r_target, np.int32), c(W.r_state, np.int32),
                                         c(W.r_hasMatcher, np.uint8), c(W.r_matcher, np.float64).reshape(-1), c(W.r_isLinearized, np.uint8),
                                         c(W.r_isActive, np.uint8)))
        ck(self.L.sdvgn_ef_set_marg_prior(self.h_, c(W.HM, np.float64).reshape(-1), c(W.bM, np.float64)))
        if getattr(W, "nullspaces", None) is not None:
            ns = c(W.nullspaces, np.float64)
            ck(self.L.sdvgn_ef_set_nullspaces(self.h_, ns.shape[0], ns.reshape(-1)))
        self.setAdjointsF()
        self.setPrecalcValues()
        self.make_resident()
        return self

    def set_residual_jacobians(self, J24, res_toZero2=None):
        """EFResidual::takeDataF for Jacobians linearised by the caller ([nR][24] floats in sdvgn_ef_get_residual_J's layout)."""
        J24 = np.ascontiguousarray(J24, np.float32)
        r2z = None if res_toZero2 is None else np.ascontiguousarray(res_toZero2, np.float32)
        self._check(self.L.sdvgn_ef_set_residual_jacobians(self.h_, J24.shape[0], J24.reshape(-1), None if r2z is None else r2z.ctypes.data))

    def make_resident(self):
        """Upload what the setters left pending (window constants, frame states, calib) and drain the stream: optimize() then starts from
        HBM-resident inputs."""
        self._check(self.L.sdvgn_ef_make_resident(self.h_))

    def setPrecalcValues(self):
        self._check(self.L.sdvgn_ef_set_precalc(self.h_))

    def setAdjointsF(self):
        self._check(self.L.sdvgn_ef_set_adjoints(self.h_))

    def set_marg_prior(self, HM, bM):
        self._check(self.L.sdvgn_ef_set_marg_prior(self.h_, np.ascontiguousarray(HM, np.float64).reshape(-1), np.ascontiguousarray(bM, np.float64)))

    def compute_nullspaces(self):
        """FullSystem::getNullspaces from the loaded frames (7 vectors); returns them [7][4+6nF]."""
        self._check(self.L.sdvgn_ef_compute_nullspaces(self.h_))
        k = self.L.sdvgn_ef_get_nullspaces(self.h_, None, 0)
        out = np.zeros((k, 4 + 6 * self.nF))
        self.L.sdvgn_ef_get_nullspaces(self.h_, out.ctypes.data_as(vp), k)
        return out

    def set_nullspaces(self, ns):
        ns = np.ascontiguousarray(ns, np.float64)
        self._check(self.L.sdvgn_ef_set_nullspaces(self.h_, ns.shape[0], ns.reshape(-1)))

    def set_frame_states(self, state10):
        self._check(self.L.sdvgn_ef_set_frame_states(self.h_, np.ascontiguousarray(state10, np.float64).reshape(-1)))

    def set_host_range(self, h0, h1):
        self._check(self.L.sdvgn_ef_set_host_range(self.h_, h0, h1))

    def linearizeAll(self, want_energy=True):
        e = C.c_double(0)
        self._check(self.L.sdvgn_ef_linearize_all(self.h_, C.byref(e) if want_energy else None))
        return e.value

    def applyRes(self):
        self._check(self.L.sdvgn_ef_apply_res(self.h_))

    def solveSystemF(self, iteration, lam):
        x = np.zeros(self.dim)
        self._check(self.L.sdvgn_ef_solve_system(self.h_, iteration, lam, x.ctypes.data_as(vp)))
        return x

    def solve_status(self):
        """0: the last solve used the fast unpivoted LDL^T; 2: it fell back to the pivoted one (indefinite system)"""
        return self.L.sdvgn_ef_get_solve_status(self.h_)

    def set_arith(self, mode):
        """0: the reference's arithmetic in linearize (default, bit-exact); 1: tolerance mode (FMA, v_rcp / v_sqrt)"""
        self._check(self.L.sdvgn_ef_set_arith(self.h_, int(mode)))

    def accumulate(self):
        self._check(self.L.sdvgn_ef_accumulate(self.h_))

    def finish_solve(self, iteration, lam):
        x = np.zeros(self.dim)
        self._check(self.L.sdvgn_ef_finish_solve(self.h_, iteration, lam, x.ctypes.data_as(vp)))
        return x

    def accumulators_dev(self):
        p = vp()
        n = C.c_int(0)
        self._check(self.L.sdvgn_ef_accumulators_dev(self.h_, C.byref(p), C.byref(n)))
        return p.value, n.value

    def point_step(self, mode, fac=1.0):
        self._check(self.L.sdvgn_ef_point_step(self.h_, mode, fac))

    @property
    def dim(self):
        return self.L.sdvgn_ef_dim(self.h_)

    def system(self):
        n = self.dim
        out = dict(HA=np.zeros((n, n)), bA=np.zeros(n), Hsc=np.zeros((n, n)), bsc=np.zeros(n), HFinal=np.zeros((n, n)), bFinal=np.zeros(n))
        self._check(self.L.sdvgn_ef_get_system(self.h_, *[out[k].ctypes.data_as(vp) for k in ("HA", "bA", "Hsc", "bsc", "HFinal", "bFinal")]))
        return out

    def residual_J(self, which):
        out = np.zeros((self.nR, 24), np.float32)
        self._check(self.L.sdvgn_ef_get_residual_J(self.h_, which, out.reshape(-1)))
        return out

    def residual_state(self):
        ss = np.zeros(self.nR, np.int32)
        sn = np.zeros(self.nR, np.int32)
        en = np.zeros(self.nR, np.float32)
        eo = np.zeros(self.nR, np.float32)
        ac = np.zeros(self.nR, np.uint8)
        self._check(self.L.sdvgn_ef_get_residual_state(self.h_, ss.ctypes.data_as(vp), sn.ctypes.data_as(vp), en.ctypes.data_as(vp),
                                                       eo.ctypes.data_as(vp), ac.ctypes.data_as(vp)))
        return dict(state=ss, new_state=sn, new_energy=en, energy_with_outlier=eo, active=ac)

    def points(self):
        out = np.zeros((self.nP, 9), np.float32)
        self._check(self.L.sdvgn_ef_get_points(self.h_, out.reshape(-1)))
        return out

    def optimizeImmature(self, host, u, v, idepth_min, idepth_max, energyTH, color, weights, isFromSensor, minObs=1):
        """FullSystem::optimizeImmaturePoint for n points (FullSystemOptPoint.cpp:18-185): (result, idepth, res_state[n][nF])."""
        host = np.ascontiguousarray(host, np.int32)
        n = len(host)
        f = lambda a: np.ascontiguousarray(a, np.float32).reshape(-1)   # noqa: E731
        result = np.zeros(n, np.int32)
        idepth = np.zeros(n, np.float32)
        rs = np.zeros((n, self.nF), np.int32)
        self._check(self.L.sdvgn_ef_optimize_immature(self.h_, n, host, f(u), f(v), f(idepth_min), f(idepth_max), f(energyTH), f(color), f(weights),
                                                      np.ascontiguousarray(isFromSensor, np.uint8), minObs, result, idepth, rs.reshape(-1)))
        return result, idepth, rs

    def frame_image_dev(self, idx):
        """Device pointer (ctypes c_void_p) of key-frame idx's level-0 {I,dx,dy} image held by the window."""
        p = self.L.sdvgn_ef_frame_image_dev(self.h_, idx)
        if not p:
            raise RuntimeError("no device image for frame %d" % idx)
        return vp(p)

    def top_acc(self):
        out = np.zeros((self.nF * self.nF, 11, 11))
        n = C.c_int(0)
        self._check(self.L.sdvgn_ef_get_top_acc(self.h_, out.reshape(-1), C.byref(n)))
        return out, n.value

    def stream(self):
        return self.L.sdvgn_ef_stream(self.h_)

    def optimize(self, its=6, cap=128, want_trace=True, fixed_its=False, relinearize_on_reject=False, reuse_after_reject=False, time_linearize=False,
                 no_spec_solve=False):
        stride = 8 + self.dim   # ..., x[dim], frameEnergyTH of the newest frame after the trial linearizeAll
        flags = (1 if fixed_its else 0) | (2 if relinearize_on_reject else 0) | (4 if reuse_after_reject else 0) | (8 if time_linearize else 0) | \
            (16 if no_spec_solve else 0)
        if not want_trace:      # nothing to allocate or zero on the way in: rows of zeros, one per body, on the way out
            n = self._check(self.L.sdvgn_ef_optimize(self.h_, its, flags, None, stride, cap))
            return np.zeros((n, stride))
        trace = np.zeros((cap, stride))
        n = self._check(self.L.sdvgn_ef_optimize(self.h_, its, flags, trace.ctypes.data_as(vp), stride, cap))
        return trace[:n]

    def linearize_times_ms(self):
        """durations of the k_ef_linearize launches of the last optimize(time_linearize=True) call (HIP events), in launch order"""
        n = self.L.sdvgn_ef_get_linearize_times(self.h_, None, 0)
        out = np.zeros(max(n, 1), np.float32)
        self.L.sdvgn_ef_get_linearize_times(self.h_, out.ctypes.data_as(vp), n)
        return out[:n]

    def launch_linearize_only(self, reps=1):
        self._check(self.L.sdvgn_debug_launch_linearize(self.h_, reps))

    def optimize_finish(self):
        """Tail of FullSystem::optimize (FullSystemOptimize.cpp:460-470): (lastEnergy[0], relbs_max[nP], ngood_inc[nP], removed[nR])."""
        e = C.c_double(0)
        rb = np.zeros(self.nP, np.float32)
        ng = np.zeros(self.nP, np.int32)
        table = getattr(self, "table_mode", False)     # a window edited in place: `removed` is a slot table [nF][nP]
        rm = np.zeros(self.nF * self.nP if table else max(self.nR, 1), np.uint8)
        self._check(self.L.sdvgn_ef_optimize_finish(self.h_, C.byref(e), rb.ctypes.data_as(vp), ng.ctypes.data_as(vp), rm.ctypes.data_as(vp)))
        return e.value, rb, ng, (rm.reshape(self.nF, self.nP) if table else rm[:self.nR])

    def clear_error(self):
        """the handle's sticky error word as it was (0: none); cleared (sdvgn_ef_clear_error)"""
        return self._check(self.L.sdvgn_ef_clear_error(self.h_))

    def point_nogood(self):
        """per point: 1 if some solveSystemF of the last optimize() found it without an active residual (AccumulatedSCHessian.cpp:14-21 zeroes
        PointHessian::maxRelBaseline there)"""
        out = np.zeros(max(self.nP, 1), np.uint8)
        self._check(self.L.sdvgn_ef_get_point_nogood(self.h_, out.ctypes.data_as(vp)))
        return out[:self.nP]

    def frame_energy_th(self):
        th = np.zeros(self.nF, np.float32)
        self._check(self.L.sdvgn_ef_get_frame_energy_th(self.h_, th))
        return th

    # ---- marginalisation (the key-frame cycle around optimize) ----
    def fixLinearization(self, mask):
        m = np.ascontiguousarray(mask, np.uint8)
        self._check(self.L.sdvgn_ef_fix_linearization(self.h_, m.ctypes.data_as(vp)))

    def resetOOB(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self._check(self.L.sdvgn_ef_reset_oob(self.h_, None if m is None else m.ctypes.data_as(vp)))

    def marginalizePoints(self, marg, drop=None):
        m = np.ascontiguousarray(marg, np.uint8)
        d = None if drop is None else np.ascontiguousarray(drop, np.uint8)
        self._check(self.L.sdvgn_ef_marginalize_points(self.h_, m.ctypes.data_as(vp), None if d is None else d.ctypes.data_as(vp)))

    def marg_prior(self):
        n = self.dim
        HM, bM = np.zeros((n, n)), np.zeros(n)
        self._check(self.L.sdvgn_ef_get_marg_prior(self.h_, HM.ctypes.data_as(vp), bM.ctypes.data_as(vp)))
        return HM, bM

    def marginalizeFrame(self, idx):
        n = self.dim - 6
        HM, bM = np.zeros((n, n)), np.zeros(n)
        self._check(self.L.sdvgn_ef_marginalize_frame(self.h_, int(idx), HM.ctypes.data_as(vp), bM.ctypes.data_as(vp)))
        return HM, bM

    def res_toZero(self):
        out = np.zeros((self.nR, 2), np.float32)
        lin = np.zeros(self.nR, np.uint8)
        self._check(self.L.sdvgn_ef_get_res_toZero(self.h_, out.ctypes.data_as(vp), lin.ctypes.data_as(vp)))
        return out, lin

    def accepted_steps(self):
        return self._check(self.L.sdvgn_ef_get_accepted_steps(self.h_))

    def look_ahead(self):
        """(rejected cases solved ahead, bodies that started from one) of the last optimize call"""
        a, b = C.c_int(0), C.c_int(0)
        self._check(self.L.sdvgn_ef_get_look_ahead(self.h_, C.byref(a), C.byref(b)))
        return a.value, b.value

    def loop_counters(self):
        """(look-ahead solves launched, bodies served by one, accept tests merged into the next accumulate, accumulates queued ahead) of the last optimize call"""
        out = np.zeros(4, np.int32)
        self._check(self.L.sdvgn_debug_loop_counters(self.h_, out.ctypes.data_as(vp)))
        return tuple(int(v) for v in out)

    def iteration_times_us(self):
        n = self.L.sdvgn_ef_get_iteration_times(self.h_, None, 0)
        out = np.zeros(max(n, 1))
        self.L.sdvgn_ef_get_iteration_times(self.h_, out.ctypes.data_as(vp), n)
        return out[:n]

    def state(self):
        vs = np.zeros(4)
        st = np.zeros(10 * self.nF)
        idp = np.zeros(self.nP, np.float32)
        self._check(self.L.sdvgn_ef_get_state(self.h_, vs.ctypes.data_as(vp), st.ctypes.data_as(vp), idp.ctypes.data_as(vp)))
        return vs, st.reshape(self.nF, 10), idp


    # ---- the window edited in place (EnergyFunctional.h:51-58; csrc/backend_window.inc) ----------------------------------------------
    def insertFrame(self, evalPT7, state10, state_zero10, frameID, ab_exposure, frameEnergyTH, dI=None, image=None):
        c = np.ascontiguousarray
        a = None if dI is None else c(dI, np.float32).reshape(-1)
        b = None if image is None else c(image, np.float32).reshape(-1)
        self._keep = (getattr(self, "_keep", None) or []) + [(a, b)]      # (the uploads are asynchronous: every inserted frame's buffers live until the commit)
        idx = self._check(self.L.sdvgn_ef_insert_frame(self.h_, c(evalPT7, np.float64), c(state10, np.float64), c(state_zero10, np.float64), int(frameID),
                                                       float(ab_exposure), float(frameEnergyTH), None if a is None else a.ctypes.data,
                                                       None if b is None else b.ctypes.data))
        return idx

    def removeFrame(self, idx, HM=None, bM=None):
        h = None if HM is None else np.ascontiguousarray(HM, np.float64)
        b = None if bM is None else np.ascontiguousarray(bM, np.float64)
        self._check(self.L.sdvgn_ef_remove_frame(self.h_, int(idx), None if h is None else h.ctypes.data, None if b is None else b.ctypes.data))

    def updateFrames(self, evalPT, state, state_zero, ab_exposure):
        c = np.ascontiguousarray
        self._check(self.L.sdvgn_ef_update_frames(self.h_, len(ab_exposure), c(evalPT, np.float64).reshape(-1), c(state, np.float64).reshape(-1),
                                                  c(state_zero, np.float64).reshape(-1), c(ab_exposure, np.float32)))

    def insertPoints(self, host, u, v, idepth, idepth_zero, color, weights, hasDepthPrior, isFromSensor):
        c = np.ascontiguousarray
        n = len(host)
        ids = np.zeros(max(n, 1), np.int32)
        self._check(self.L.sdvgn_ef_insert_points(self.h_, n, c(host, np.int32), c(u, np.float32), c(v, np.float32), c(idepth, np.float32), c(idepth_zero, np.float32),
                                                  c(color, np.float32).reshape(-1), c(weights, np.float32).reshape(-1), c(hasDepthPrior, np.uint8),
                                                  c(isFromSensor, np.uint8), ids.ctypes.data_as(vp)))
        return ids[:n]

    def removePoints(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        self._check(self.L.sdvgn_ef_remove_points(self.h_, len(ids), ids))

    def insertResiduals(self, point_id, target, state=None, hasMatcher=None, matcher=None, update=False):
        c = np.ascontiguousarray
        n = len(point_id)
        state = np.zeros(n, np.int32) if state is None else state
        hasMatcher = np.ones(n, np.uint8) if hasMatcher is None else hasMatcher
        fn = self.L.sdvgn_ef_update_residuals if update else self.L.sdvgn_ef_insert_residuals
        self._check(fn(self.h_, n, c(point_id, np.int32), c(target, np.int32), c(state, np.int32), c(hasMatcher, np.uint8), c(matcher, np.float64).reshape(-1)))

    def dropResiduals(self, point_id, target):
        c = np.ascontiguousarray
        self._check(self.L.sdvgn_ef_drop_residuals(self.h_, len(point_id), c(point_id, np.int32), c(target, np.int32)))

    def makeIDX(self, nF=None):
        """EnergyFunctional::makeIDX: commit the edits; returns the point id at every dense index"""
        self._check(self.L.sdvgn_ef_make_idx(self.h_))
        self.nF = self.L.sdvgn_ef_dim(self.h_)
        self.nF = (self.nF - 4) // 6
        ids = np.empty(self.max_points, np.int32)
        self.nP = self._check(self.L.sdvgn_ef_get_point_ids(self.h_, ids.ctypes.data_as(vp)))
        self.nR = 0
        self.table_mode = True
        self._keep = None
        return ids[:self.nP].copy()

    def residual_table(self):
        """the per-slot planes, [nF][nP] each"""
        n = self.nF * self.nP
        ex, act = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        st, sn = np.zeros(n, np.int8), np.zeros(n, np.int8)
        en, enn, ew = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        self._check(self.L.sdvgn_ef_get_residual_table(self.h_, ex.ctypes.data_as(vp), st.ctypes.data_as(vp), sn.ctypes.data_as(vp), en.ctypes.data_as(vp),
                                                       enn.ctypes.data_as(vp), ew.ctypes.data_as(vp), act.ctypes.data_as(vp)))
        sh = (self.nF, self.nP)
        return dict(exists=ex.reshape(sh), state=st.reshape(sh), new_state=sn.reshape(sh), energy=en.reshape(sh), new_energy=enn.reshape(sh),
                    energy_with_outlier=ew.reshape(sh), active=act.reshape(sh))


def optimize_batch(windows, its=6, fixed_its=False):
    """sdvgn_ef_optimize_batch: FullSystem::optimize on independent windows side by side (handles created with stream=EnergyFunctional.STREAM_OWN
    overlap on the device).  Returns the loop bodies every window ran."""
    from .api import check
    B = len(windows)
    arr = (vp * B)(*[w.h_ for w in windows])
    out = (C.c_int * B)()
    check(windows[0].L.sdvgn_ef_optimize_batch(C.cast(arr, vp), B, int(its), 1 if fixed_its else 0, C.cast(out, vp)))
    return list(out)


def optimize_lockstep(windows, its=6, fixed_its=False, want_trace=True, cap=128):
    """sdvgn_ef_optimize_lockstep: the B loops as ONE launch sequence (csrc/backend_lockstep.inc).  Returns (bodies per window, traces) --
    traces[b] has the rows of EnergyFunctional.optimize's trace for window b (None without want_trace)."""
    from .api import check
    B = len(windows)
    arr = (vp * B)(*[w.h_ for w in windows])
    out = (C.c_int * B)()
    stride = 8 + max(w.dim for w in windows)
    trace = np.zeros((B, cap, stride)) if want_trace else None
    check(windows[0].L.sdvgn_ef_optimize_lockstep(C.cast(arr, vp), B, int(its), 1 if fixed_its else 0, C.cast(out, vp),
                                                   trace.ctypes.data_as(vp) if want_trace else None, stride, cap))
    its_out = list(out)
    if not want_trace:
        return its_out, None
    return its_out, [trace[b, :its_out[b], :8 + windows[b].dim] for b in range(B)]
