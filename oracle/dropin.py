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
