"""ctypes wrapper of the CPU oracle's Reprojector restatement (oracle/orc_reproject.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import lib

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_bound = set()


def _L(raw=None, prefix="orc_"):
    from . import _Prefixed
    L = _Prefixed(lib() if raw is None else raw, prefix)
    if prefix not in _bound:
        _bound.add(prefix)
        L.orc_rp_create.restype = C.c_void_p
        L.orc_rp_create.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_rp_destroy.argtypes = [C.c_void_p]
        L.orc_rp_set_calib.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_rp_set_frame.argtypes = [C.c_void_p, C.c_int, f64p, f32p, C.c_float, C.c_double, C.c_double]
        L.orc_rp_set_cur_pose.argtypes = [C.c_void_p, f64p, C.c_float, C.c_double, C.c_double]
        L.orc_rp_set_cur_level.argtypes = [C.c_void_p, C.c_int, f32p]
        L.orc_rp_project.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, i32p, f64p, i32p, f32p]
        L.orc_rp_find_match.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, i32p, i32p, i32p, f64p, i32p, i32p]
    return L


class OracleReprojector:
    """Per-candidate part of class Reprojector (src/FullSystem/Reprojector.h:17-112) on the CPU oracle."""

    _prefix = "orc_"

    @classmethod
    def _raw(cls):
        return None

    def __init__(self, w, h, levels):
        self.L = _L(self._raw(), self._prefix)
        self.w, self.h, self.levels = w, h, levels
        self.h_ = self.L.orc_rp_create(w, h, levels)

    def __del__(self):
        if getattr(self, "h_", None):
            self.L.orc_rp_destroy(self.h_)
            self.h_ = None

    def set_calib(self, fx, fy, cx, cy):
        self.L.orc_rp_set_calib(self.h_, fx, fy, cx, cy)

    def set_frame(self, idx, camToWorld7, dI_aos3, exposure=1.0, a=0.0, b=0.0):
        self.L.orc_rp_set_frame(self.h_, idx, np.ascontiguousarray(camToWorld7, np.float64),
                                np.ascontiguousarray(dI_aos3, np.float32).reshape(-1), exposure, a, b)

    def set_cur(self, camToWorld7, pyr_aos3, exposure=1.0, a=0.0, b=0.0):
        self.L.orc_rp_set_cur_pose(self.h_, np.ascontiguousarray(camToWorld7, np.float64), exposure, a, b)
        for l, img in enumerate(pyr_aos3):
            self.L.orc_rp_set_cur_level(self.h_, l, np.ascontiguousarray(img, np.float32).reshape(-1))

    def project(self, u, v, idepth, host_idx):
        u, v, idepth = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth))
        host_idx = np.ascontiguousarray(host_idx, np.int32)
        n = len(u)
        px = np.zeros((n, 2))
        cell = np.zeros(n, np.int32)
        q = np.zeros(n, np.float32)
        self.L.orc_rp_project(self.h_, n, u, v, idepth, host_idx, px.reshape(-1), cell, q)
        return px, cell, q

    def find_match(self, u, v, idepth, host_idx, ref_idx, ptype, px):
        u, v, idepth = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth))
        host_idx, ref_idx, ptype = (np.ascontiguousarray(x, np.int32) for x in (host_idx, ref_idx, ptype))
        n = len(u)
        px = np.array(px, np.float64).reshape(n, 2).copy()
        ok = np.zeros(n, np.int32)
        lvl = np.zeros(n, np.int32)
        self.L.orc_rp_find_match(self.h_, n, u, v, idepth, host_idx, ref_idx, ptype, px.reshape(-1), ok, lvl)
        return ok.astype(bool), px, lvl


class RefReprojector(OracleReprojector):
    """the same calls on the REFERENCE'S OWN Reprojector (Reprojector.cpp compiled unmodified into oracle/_ref/libref.so; oracle/ref_glue_misc.cpp)"""
    _prefix = "ref_"

    @classmethod
    def _raw(cls):
        from . import refpin
        R = refpin.ref_lib()
        if R is None:
            raise RuntimeError("oracle/_ref/libref.so has not been built")
        return R
