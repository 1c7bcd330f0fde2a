"""CPU oracle loader -- TEST INFRASTRUCTURE ONLY.  PARITY PINNED: every function of the oracle is checked against the reference's own
translation units compiled unmodified into oracle/_ref/libref.so (oracle/README.md; oracle/refpin.py, RefTracker below,
oracle/backend.py::RefEF; tests/test_ref_pin*.py).

ctypes front-end for ``oracle/liborc.so`` (built by ``make -C oracle`` from orc_tracker.cpp /
orc_backend.cpp, the plain-C++ restatement of the reference's CPU hot path).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``sdv-loam_amd`` / ``libsdvgn.so``) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so



class _Missing:
    """an entry point the wrapped library does not have"""

    def __init__(self, name):
        object.__setattr__(self, "_name", name)

    def __setattr__(self, k, v):
        pass

    def __call__(self, *a):
        raise NotImplementedError(self._name)


class _Prefixed:
    """`P.orc_xxx` -> the function `<prefix>xxx` of the wrapped library: the reference-built library (oracle/_ref/libref.so,
    oracle/ref_glue_*.cpp) exports the oracle's entry points under the prefix ref_ with the same signatures."""

    def __init__(self, L, prefix):
        object.__setattr__(self, "_L", L)
        object.__setattr__(self, "_p", prefix)

    def __getattr__(self, name):
        assert name.startswith("orc_")
        try:
            return getattr(self._L, self._p + name[4:])
        except AttributeError:
            return _Missing(self._p + name[4:])


def _bind_tracker(L):
    L.orc_tracker_create.restype = C.c_void_p
    L.orc_tracker_create.argtypes = [C.c_int, C.c_int, C.c_int]
    L.orc_tracker_destroy.argtypes = [C.c_void_p]
    L.orc_tracker_set_settings.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_tracker_make_K.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_tracker_get_K.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
    L.orc_tracker_set_ref.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, f32p, f32p, f32p]
    L.orc_tracker_set_ref_frame.argtypes = [C.c_void_p, C.c_float, C.c_double, C.c_double]
    L.orc_tracker_set_new_image.argtypes = [C.c_void_p, f32p, C.c_float]
    L.orc_tracker_set_new_pyr.argtypes = [C.c_void_p, C.c_int, f32p, C.c_float]
    L.orc_tracker_get_pyr.argtypes = [C.c_void_p, C.c_int, f32p]
    L.orc_make_images.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
    L.orc_calc_res.argtypes = [C.c_void_p, C.c_int, f64p, C.c_double, C.c_double, C.c_float, f64p]
    L.orc_get_warped.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_get_warped.restype = C.c_int
    L.orc_calc_gs.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, f64p, f64p]
    L.orc_track.argtypes = [C.c_void_p, f64p, f64p, C.c_int, f64p, f64p, f64p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_track.restype = C.c_int
    L.orc_trace_stride.restype = C.c_int
    L.orc_tracker_make_coarse_depth.argtypes = [C.c_void_p, C.c_int, np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"),
                                                np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS"), f32p, f32p]
    L.orc_tracker_get_ref.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_tracker_get_ref.restype = C.c_int
    i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
    L.orc_struct_trace_stride.restype = C.c_int
    L.orc_struct_pose.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, i32p, f64p, f64p, f64p, C.c_void_p, C.c_void_p]
    L.orc_struct_pose.restype = C.c_int
    L.orc_struct_res_Hb.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, i32p, f64p, f64p, f64p, f64p, f64p, C.c_void_p, C.c_void_p]
    L.orc_se3_exp.argtypes = [f64p, f64p]
    L.orc_se3_log.argtypes = [f64p, f64p]
    L.orc_se3_mul.argtypes = [f64p, f64p, f64p]
    L.orc_se3_inverse.argtypes = [f64p, f64p]
    L.orc_se3_matrix.argtypes = [f64p, f64p]
    L.orc_se3_adj.argtypes = [f64p, f64p]
    L.orc_ldlt_solve.argtypes = [C.c_int, f64p, f64p, f64p]
    L.orc_inv3f.argtypes = [f32p, f32p]
    L.orc_interp33.argtypes = [f32p, C.c_float, C.c_float, C.c_int, f32p]



def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "liborc.so")
    if not os.path.exists(so):
        build()
    L = C.CDLL(so)
    _bind_tracker(L)
    _bind_backend(L)
    _LIB = L
    return L


def _bind_backend(L):
    if not hasattr(L, "orc_ef_create"):
        return
    from . import backend as _b  # noqa: F401  (binds its own prototypes lazily)


# ------------------------------------------------------------------------------------------------
# maths helpers
# ------------------------------------------------------------------------------------------------
def se3_exp(a):
    out = np.zeros(7)
    lib().orc_se3_exp(np.ascontiguousarray(a, np.float64), out)
    return out


def se3_log(p):
    out = np.zeros(6)
    lib().orc_se3_log(np.ascontiguousarray(p, np.float64), out)
    return out


def se3_mul(a, b):
    out = np.zeros(7)
    lib().orc_se3_mul(np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64), out)
    return out


def se3_inverse(a):
    out = np.zeros(7)
    lib().orc_se3_inverse(np.ascontiguousarray(a, np.float64), out)
    return out


def se3_matrix(a):
    out = np.zeros(9)
    lib().orc_se3_matrix(np.ascontiguousarray(a, np.float64), out)
    return out.reshape(3, 3)


def se3_adj(a):
    out = np.zeros(36)
    lib().orc_se3_adj(np.ascontiguousarray(a, np.float64), out)
    return out.reshape(6, 6)


def ldlt_solve(A, b):
    A = np.ascontiguousarray(A, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    x = np.zeros_like(b)
    lib().orc_ldlt_solve(A.shape[0], A, b, x)
    return x


def make_images(color, w, h, levels):
    """a1: list of per-level AoS arrays (h_l, w_l, 3) = {I, dx, dy}."""
    color = np.ascontiguousarray(color, np.float32).reshape(-1)
    sizes = [(w >> l) * (h >> l) * 3 for l in range(levels)]
    out = np.zeros(sum(sizes), np.float32)
    lib().orc_make_images(color, w, h, levels, out)
    res, off = [], 0
    for l, s in enumerate(sizes):
        res.append(out[off:off + s].reshape(h >> l, w >> l, 3).copy())
        off += s
    return res


IDENTITY_POSE = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


class OracleTracker:
    """Mirror of the reference's CoarseTracker call surface (CoarseTracker.h:17-107) on the CPU oracle."""

    @classmethod
    def _library(cls):
        return lib()

    def __init__(self, w, h, levels):
        self.L = self._library()
        self.w, self.h, self.levels = w, h, levels
        self.h_ = self.L.orc_tracker_create(w, h, levels)

    def __del__(self):
        try:
            self.L.orc_tracker_destroy(self.h_)
        except Exception:
            pass

    def set_settings(self, huber=6.0, cutoff=20.0, aff_a=0.0, aff_b=0.0):
        self.L.orc_tracker_set_settings(self.h_, huber, cutoff, aff_a, aff_b)

    def makeK(self, fx, fy, cx, cy):
        self.L.orc_tracker_make_K(self.h_, fx, fy, cx, cy)

    def get_K(self, lvl):
        k4 = np.zeros(4, np.float32)
        ki = np.zeros(9, np.float32)
        self.L.orc_tracker_get_K(self.h_, lvl, k4, ki)
        return k4, ki.reshape(3, 3)

    def set_ref(self, lvl, u, v, idepth, color):
        u, v, idepth, color = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color))
        self.L.orc_tracker_set_ref(self.h_, lvl, len(u), u, v, idepth, color)

    def set_ref_frame(self, exposure=1.0, a=0.0, b=0.0):
        self.L.orc_tracker_set_ref_frame(self.h_, exposure, a, b)

    def makeCoarseDepth(self, u, v, new_idepth, weight):
        """makeCoarseDepthL0 / makeCoarseDepthForFirstFrame from their splat tuples; lastRef->dIp = the current new-frame pyramid."""
        u, v = (np.ascontiguousarray(x, np.int32) for x in (u, v))
        self.L.orc_tracker_make_coarse_depth(self.h_, len(u), u, v, np.ascontiguousarray(new_idepth, np.float32),
                                             np.ascontiguousarray(weight, np.float32))

    def get_ref(self, lvl):
        n = self.L.orc_tracker_get_ref(self.h_, lvl, None, None, None, None)
        out = [np.zeros(n, np.float32) for _ in range(4)]
        self.L.orc_tracker_get_ref(self.h_, lvl, *[a.ctypes.data_as(C.c_void_p) for a in out])
        return dict(u=out[0], v=out[1], idepth=out[2], color=out[3])

    def set_new_image(self, color, exposure=1.0):
        self.L.orc_tracker_set_new_image(self.h_, np.ascontiguousarray(color, np.float32).reshape(-1), exposure)

    def set_new_pyr(self, lvl, aos3, exposure=1.0):
        self.L.orc_tracker_set_new_pyr(self.h_, lvl, np.ascontiguousarray(aos3, np.float32).reshape(-1), exposure)

    def get_pyr(self, lvl):
        out = np.zeros((self.h >> lvl) * (self.w >> lvl) * 3, np.float32)
        self.L.orc_tracker_get_pyr(self.h_, lvl, out)
        return out.reshape(self.h >> lvl, self.w >> lvl, 3)

    def calcRes(self, lvl, pose7, a, b, cutoff):
        out = np.zeros(6)
        self.L.orc_calc_res(self.h_, lvl, np.ascontiguousarray(pose7, np.float64), a, b, cutoff, out)
        return out

    def warped(self):
        n = self.L.orc_get_warped(self.h_, None)
        out = np.zeros((8, n), np.float32)
        if n:
            self.L.orc_get_warped(self.h_, out.ctypes.data_as(C.c_void_p))
        return out

    def calcGS(self, lvl, a, b):
        H = np.zeros(64)
        bb = np.zeros(8)
        self.L.orc_calc_gs(self.h_, lvl, a, b, H, bb)
        return H.reshape(8, 8), bb

    def trackNewestCoarse(self, pose7, aff, coarsest, min_res=None, trace_cap=512):
        pose = np.array(pose7, np.float64)
        aff = np.array(aff, np.float64)
        mr = np.full(5, np.nan) if min_res is None else np.array(min_res, np.float64)
        last_res = np.zeros(5)
        flow = np.zeros(3)
        stride = self.L.orc_trace_stride()
        trace = np.zeros((trace_cap, stride))
        ntr = C.c_int(0)
        ok = self.L.orc_track(self.h_, pose, aff, coarsest, mr, last_res, flow,
                              trace.ctypes.data_as(C.c_void_p), trace_cap, C.byref(ntr))
        return bool(ok), pose, aff, last_res, flow, trace[:min(ntr.value, trace_cap)]

    # -- f1: structPoseEstimation (CoarseTracker.cpp:840-1007) -------------------------------------------
    @staticmethod
    def _struct_args(u, v, idepth, host_idx, host_poses7, obs):
        u, v, idepth = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth))
        host_idx = np.ascontiguousarray(host_idx, np.int32)
        host_poses7 = np.ascontiguousarray(np.array(host_poses7, np.float64).reshape(-1, 7))
        obs = np.ascontiguousarray(np.array(obs, np.float64).reshape(-1, 2))
        assert len(u) == len(v) == len(idepth) == len(host_idx) == len(obs)
        return u, v, idepth, host_idx, host_poses7, obs

    def structPoseEstimation(self, curToWorld7, u, v, idepth, host_idx, host_poses7, obs):
        u, v, idepth, host_idx, hp, obs = self._struct_args(u, v, idepth, host_idx, host_poses7, obs)
        pose = np.array(curToWorld7, np.float64)
        stride = self.L.orc_struct_trace_stride()
        trace = np.zeros((10, stride))
        fr = C.c_double(0)
        its = self.L.orc_struct_pose(self.h_, len(u), u, v, idepth, host_idx, hp.reshape(-1), obs.reshape(-1), pose,
                                     trace.ctypes.data_as(C.c_void_p), C.byref(fr))
        return pose, trace[:its], fr.value

    def structResHb(self, worldToCur7, u, v, idepth, host_idx, host_poses7, obs):
        u, v, idepth, host_idx, hp, obs = self._struct_args(u, v, idepth, host_idx, host_poses7, obs)
        H = np.zeros(36)
        b = np.zeros(6)
        e = C.c_double(0)
        n = C.c_int(0)
        self.L.orc_struct_res_Hb(self.h_, len(u), u, v, idepth, host_idx, hp.reshape(-1), obs.reshape(-1),
                                 np.ascontiguousarray(worldToCur7, np.float64), H, b, C.byref(e), C.byref(n))
        return H.reshape(6, 6), b, e.value, n.value


_REF_TRACKER_LIB = None


class RefTracker(OracleTracker):
    """The same call surface on the REFERENCE'S OWN CoarseTracker / FrameHessian::makeImages / Sophus, compiled unmodified into
    oracle/_ref/libref.so (oracle/Makefile target `ref`) and driven by oracle/ref_glue_tracker.cpp.  trackNewestCoarse returns an
    empty trace (the reference exposes none)."""

    @classmethod
    def _library(cls):
        global _REF_TRACKER_LIB
        if _REF_TRACKER_LIB is None:
            from . import refpin
            R = refpin.ref_lib()
            if R is None:
                raise RuntimeError("oracle/_ref/libref.so has not been built (needs /root/reference; `make -C oracle ref`)")
            P = _Prefixed(R, "ref_")
            _bind_tracker(P)
            R.ref_tracker_make_coarse_depth_pts.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, f32p, C.c_int]
            _REF_TRACKER_LIB = P
        return _REF_TRACKER_LIB

    def makeCoarseDepthPts(self, u, v, idepth, HdiF, first_frame=False):
        u, v, idepth, HdiF = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, HdiF))
        self.L._L.ref_tracker_make_coarse_depth_pts(self.h_, len(u), u, v, idepth, HdiF, 1 if first_frame else 0)


def ref_se3(name, *args):
    """se3_exp / log / mul / inverse / matrix / adj evaluated by the vendored Sophus inside libref.so"""
    P = RefTracker._library()
    shapes = dict(exp=7, log=6, mul=7, inverse=7, matrix=9, adj=36)
    out = np.zeros(shapes[name])
    getattr(P, "orc_se3_" + name)(*[np.ascontiguousarray(a, np.float64) for a in args], out)
    return out
