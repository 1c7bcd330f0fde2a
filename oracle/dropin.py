"""The drop-in, driven -- TEST INFRASTRUCTURE ONLY.

`oracle/_ref/libref_dropin.so` (oracle/Makefile target `dropin`) is the reference's own FullSystem / EnergyFunctional / CoarseTracker object code
with exactly two member functions replaced at link time by the GPU-backed definitions of oracle/dropin/*.cpp:
EnergyFunctional::solveSystemF (EnergyFunctional.cpp:650-759) and CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:662-838).  The glue
(oracle/ref_glue*.cpp) is the one libref.so carries, so DropinEF / DropinTracker expose RefEF's / RefTracker's interface: the same test body
runs the reference all-CPU and the reference with libsdvgn plugged in (tests/test_dropin_gpu.py).
"""
import ctypes as C
import os

from . import OracleTracker, RefTracker, _bind_tracker, _Prefixed, f32p
from .backend import RefEF

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def dropin_path():
    return os.path.join(_HERE, "_ref", "libref_dropin.so")


def dropin_lib():
    """the library, or None when it has not been built (no /root/reference here and no prebuilt file)"""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = dropin_path()
    if not os.path.exists(p):
        return None
    from sdv_loam_amd import api
    api.load_library()          # libsdvgn first (torch's HIP runtime before it, see api.load_library): the drop-in links against it
    L = C.CDLL(p)
    for name in ("sdvgn_dropin_ef_calls", "sdvgn_dropin_tracker_calls"):
        getattr(L, name).restype = C.c_ulonglong
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("sdvgn_dropin_ef_release", "sdvgn_dropin_tracker_release", "sdvgn_dropin_tracker_invalidate"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [C.c_void_p]
    L.ref_ef_energy_functional.restype = C.c_void_p
    L.ref_ef_energy_functional.argtypes = [C.c_void_p]
    L.ref_tracker_object.restype = C.c_void_p
    L.ref_tracker_object.argtypes = [C.c_void_p]
    _LIB = L
    return L


class DropinEF(RefEF):
    """RefEF on libref_dropin.so: every solveSystemF the reference's host code issues runs on the GPU."""

    @classmethod
    def _raw_lib(cls):
        L = dropin_lib()
        if L is None:
            raise RuntimeError("oracle/_ref/libref_dropin.so has not been built (needs /root/reference; `make -C oracle dropin`)")
        return L

    def gpu_solves(self):
        R = self.L._L
        return int(R.sdvgn_dropin_ef_calls(R.ref_ef_energy_functional(self.h_)))

    def __del__(self):
        try:
            R = self.L._L
            R.sdvgn_dropin_ef_release(R.ref_ef_energy_functional(self.h_))
        except Exception:
            pass
        super().__del__()


_OPT_LIB = None


def dropin_opt_lib():
    """oracle/_ref/libref_dropin_opt.so (oracle/Makefile target `dropin_opt`): as libref_dropin.so, plus FullSystem::optimize itself replaced by
    oracle/dropin/FullSystemOptimizeGPU.cpp -- the resident-window form; None when it has not been built"""
    global _OPT_LIB
    if _OPT_LIB is not None:
        return _OPT_LIB
    p = os.path.join(_HERE, "_ref", "libref_dropin_opt.so")
    if not os.path.exists(p):
        return None
    from sdv_loam_amd import api
    api.load_library()
    L = C.CDLL(p)
    L.sdvgn_dropin_opt_calls.restype = C.c_ulonglong
    L.sdvgn_dropin_opt_calls.argtypes = [C.c_void_p]
    L.sdvgn_dropin_opt_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.sdvgn_dropin_opt_release.argtypes = [C.c_void_p]
    L.ref_ef_full_system.restype = C.c_void_p
    L.ref_ef_full_system.argtypes = [C.c_void_p]
    _OPT_LIB = L
    return L


class DropinOptEF(RefEF):
    """RefEF on libref_dropin_opt.so: the reference's FullSystem::optimize IS libsdvgn's loop on a window resident on the GPU; everything
    around it (removeOutliers, flagPointsForRemoval, marginalizePointsF, marginalizeFrame, setCoarseTrackingRef ...) is the reference's own
    host code working on what the drop-in wrote back."""

    @classmethod
    def _raw_lib(cls):
        L = dropin_opt_lib()
        if L is None:
            raise RuntimeError("oracle/_ref/libref_dropin_opt.so has not been built (needs /root/reference; `make -C oracle dropin_opt`)")
        return L

    def gpu_calls(self):
        R = self.L._L
        return int(R.sdvgn_dropin_opt_calls(R.ref_ef_full_system(self.h_)))

    def gpu_stats(self):
        """dict: images / points / residuals sent since creation, and the three phases of the last call in microseconds"""
        R = self.L._L
        out = (C.c_double * 9)()
        R.sdvgn_dropin_opt_stats(R.ref_ef_full_system(self.h_), out)
        k = ("frames_uploaded", "points_inserted", "points_removed", "residuals_inserted", "residuals_dropped", "residuals_updated", "us_sync", "us_gpu", "us_writeback")
        return dict(zip(k, list(out)))

    def __del__(self):
        try:
            R = self.L._L
            R.sdvgn_dropin_opt_release(R.ref_ef_full_system(self.h_))
        except Exception:
            pass
        super().__del__()


_FRAME_LIB = None


def dropin_frame_lib():
    """oracle/_ref/libref_dropin_frame.so (oracle/Makefile target `dropin_frame`, form B+): as libref_dropin_opt.so, plus FullSystem::traceNewCoarse,
    FullSystem::activatePointsMT_Reductor, CoarseTracker::makeCoarseDepthL0, CoarseTracker::structPoseEstimation and Reprojector::reprojectMap replaced by
    oracle/dropin/FullSystemFrameGPU.cpp; None when it has not been built"""
    global _FRAME_LIB
    if _FRAME_LIB is not None:
        return _FRAME_LIB
    p = os.path.join(_HERE, "_ref", "libref_dropin_frame.so")
    if not os.path.exists(p):
        return None
    from sdv_loam_amd import api
    api.load_library()
    L = C.CDLL(p)
    L.sdvgn_dropin_opt_calls.restype = C.c_ulonglong
    L.sdvgn_dropin_opt_calls.argtypes = [C.c_void_p]
    L.sdvgn_dropin_opt_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.sdvgn_dropin_opt_release.argtypes = [C.c_void_p]
    L.sdvgn_dropin_frame_stats.argtypes = [C.POINTER(C.c_double)]
    L.sdvgn_dropin_frame_release.argtypes = []
    L.ref_ef_full_system.restype = C.c_void_p
    L.ref_ef_full_system.argtypes = [C.c_void_p]
    for name in ("sdvgn_dropin_tracker_calls",):
        getattr(L, name).restype = C.c_ulonglong
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("sdvgn_dropin_tracker_release", "sdvgn_dropin_tracker_invalidate"):
        getattr(L, name).restype = None
        getattr(L, name).argtypes = [C.c_void_p]
    _FRAME_LIB = L
    return L


def frame_stats(L=None):
    """counters of oracle/dropin/FullSystemFrameGPU.cpp since the last release: which of the reference's per-frame call sites reached the GPU, how often"""
    L = L or dropin_frame_lib()
    out = (C.c_double * 9)()
    L.sdvgn_dropin_frame_stats(out)
    k = ("trace_calls", "trace_points", "trace_registrations", "activate_calls", "activate_points", "template_calls", "struct_pose_calls", "reproject_calls",
         "reproject_candidates")
    return dict(zip(k, [int(x) for x in out]))


class DropinFrameEF(DropinOptEF):
    """RefEF on libref_dropin_frame.so (form B+): optimize on the resident window AND the per-frame rows -- traceNewCoarse, the activation's
    optimizeImmaturePoint batch, the next tracking template -- on the GPU, at the reference's own call sites."""

    @classmethod
    def _raw_lib(cls):
        L = dropin_frame_lib()
        if L is None:
            raise RuntimeError("oracle/_ref/libref_dropin_frame.so has not been built (needs /root/reference; `make -C oracle dropin_frame`)")
        return L

    def frame_stats(self):
        return frame_stats(self.L._L)


_DROPIN_TRACKER_LIB = None


class DropinTracker(RefTracker):
    """RefTracker on libref_dropin.so: CoarseTracker::trackNewestCoarse runs on the GPU, everything else is the reference's CPU code."""

    @classmethod
    def _library(cls):
        global _DROPIN_TRACKER_LIB
        if _DROPIN_TRACKER_LIB is None:
            R = dropin_lib()
            if R is None:
                raise RuntimeError("oracle/_ref/libref_dropin.so has not been built (needs /root/reference; `make -C oracle dropin`)")
            P = _Prefixed(R, "ref_")
            _bind_tracker(P)
            R.ref_tracker_make_coarse_depth_pts.argtypes = [C.c_void_p, C.c_int, f32p, f32p, f32p, f32p, C.c_int]
            _DROPIN_TRACKER_LIB = P
        return _DROPIN_TRACKER_LIB

    def gpu_tracks(self):
        R = self.L._L
        return int(R.sdvgn_dropin_tracker_calls(R.ref_tracker_object(self.h_)))

    def _invalidate(self):
        # the glue rewrites pc_* and the new frame's pyramid IN PLACE (same lastRef / FrameHessian objects); the reference's front end makes a new
        # FrameHessian per frame and rebuilds the template only in setCoarseTrackingRef, which is what the drop-in keys its uploads on
        R = self.L._L
        R.sdvgn_dropin_tracker_invalidate(R.ref_tracker_object(self.h_))

    def set_ref(self, lvl, u, v, idepth, color):
        super().set_ref(lvl, u, v, idepth, color)
        self._invalidate()

    def set_new_image(self, color, exposure=1.0):
        super().set_new_image(color, exposure)
        self._invalidate()

    def set_new_pyr(self, lvl, aos3, exposure=1.0):
        super().set_new_pyr(lvl, aos3, exposure)
        self._invalidate()

    def __del__(self):
        try:
            R = self.L._L
            R.sdvgn_dropin_tracker_release(R.ref_tracker_object(self.h_))
        except Exception:
            pass
        super().__del__()


class RefFullSystemTracking:
    """FullSystem::trackNewCoarse (FullSystem.cpp:283-517) of the reference on a small world built from flat arrays (oracle/ref_glue_fs.cpp);
    `dropin=True` runs it in libref_dropin.so, where the trackNewestCoarse it calls (:419) is the GPU-backed definition."""

    def __init__(self, w, h, levels, calib, dropin=False):
        import numpy as np
        from . import refpin
        self.np = np
        self.dropin = dropin
        L = (dropin_frame_lib() if dropin == "frame" else dropin_lib()) if dropin else refpin.ref_lib()
        if L is None:
            raise RuntimeError("oracle/_ref/libref%s.so has not been built" % (("_dropin_frame" if dropin == "frame" else "_dropin") if dropin else ""))
        self.L = L
        f32 = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        f64 = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
        i32 = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
        L.ref_fs_create.restype = C.c_void_p
        L.ref_fs_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.ref_fs_destroy.argtypes = [C.c_void_p]
        L.ref_fs_add_keyframe.argtypes = [C.c_void_p, f64, f32, C.c_float, C.c_double, C.c_double]
        L.ref_fs_add_points.argtypes = [C.c_void_p, C.c_int, i32, f32, f32, f32, i32]
        L.ref_fs_set_tracker_ref.argtypes = [C.c_void_p, C.c_int, C.c_int, f32, f32, f32, f32]
        L.ref_fs_set_new_frame.argtypes = [C.c_void_p, f32, C.c_float]
        L.ref_fs_track_new_coarse.argtypes = [C.c_void_p, f64, f64, f64, f64, f64]
        L.ref_fs_last_log.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.ref_fs_last_log.restype = C.c_int
        L.ref_fs_coarse_tracker.restype = C.c_void_p
        L.ref_fs_coarse_tracker.argtypes = [C.c_void_p]
        L.ref_fs_set_last_coarse_rmse.argtypes = [C.c_void_p, C.c_double]
        self.h_ = L.ref_fs_create(w, h, levels, calib["fx"], calib["fy"], calib["cx"], calib["cy"])

    def __del__(self):
        try:
            if self.dropin:
                self.L.sdvgn_dropin_tracker_release(self.L.ref_fs_coarse_tracker(self.h_))
            self.L.ref_fs_destroy(self.h_)
        except Exception:
            pass

    def add_keyframe(self, camToWorld7, dI_aos3, exposure=1.0, a=0.0, b=0.0):
        np = self.np
        self.L.ref_fs_add_keyframe(self.h_, np.ascontiguousarray(camToWorld7, np.float64), np.ascontiguousarray(dI_aos3, np.float32).reshape(-1), exposure, a, b)

    def add_points(self, host, u, v, idepth, ptype):
        np = self.np
        self.L.ref_fs_add_points(self.h_, len(u), np.ascontiguousarray(host, np.int32), np.ascontiguousarray(u, np.float32), np.ascontiguousarray(v, np.float32),
                                 np.ascontiguousarray(idepth, np.float32), np.ascontiguousarray(ptype, np.int32))

    def set_tracker_ref(self, lvl, u, v, idepth, color):
        np = self.np
        u, v, idepth, color = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color))
        self.L.ref_fs_set_tracker_ref(self.h_, lvl, len(u), u, v, idepth, color)

    def set_new_frame(self, image, exposure=1.0):
        self.L.ref_fs_set_new_frame(self.h_, self.np.ascontiguousarray(image, self.np.float32).reshape(-1), exposure)

    def set_last_coarse_rmse(self, v):
        self.L.ref_fs_set_last_coarse_rmse(self.h_, float(v))

    def trackNewCoarse(self):
        np = self.np
        out4, c2w, c2r, aff, rmse = np.zeros(4), np.zeros(7), np.zeros(7), np.zeros(2), np.zeros(5)
        self.L.ref_fs_track_new_coarse(self.h_, out4, c2w, c2r, aff, rmse)
        n = self.L.ref_fs_last_log(self.h_, None, 0)
        buf = C.create_string_buffer(n + 1)
        self.L.ref_fs_last_log(self.h_, buf, n + 1)
        return dict(ret=out4, camToWorld=c2w, camToTrackingRef=c2r, aff=aff, lastCoarseRMSE=rmse, log=buf.value.decode(errors="replace"))

    def gpu_tracks(self):
        return int(self.L.sdvgn_dropin_tracker_calls(self.L.ref_fs_coarse_tracker(self.h_))) if self.dropin else 0
