"""ctypes binding of libsdvgn.so (include/sdvgn.h) + thin Python mirrors of the reference classes.

The product is the C-ABI library; this module only exists so that tests/ and bench.py can drive it from
Python.  It fails loudly (ImportError / RuntimeError) when the HIP library is missing or reports an error --
there is no CPU fallback on the product path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdvgn.so")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
vp = C.c_void_p

# (name, restype, argtypes) -- one row per symbol declared in include/sdvgn.h
PROTOTYPES = [
    ("sdvgn_version", C.c_char_p, []),
    ("sdvgn_error_string", C.c_char_p, [C.c_int]),
    ("sdvgn_tracker_create", C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    ("sdvgn_tracker_destroy", None, [vp]),
    ("sdvgn_tracker_set_settings", C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("sdvgn_tracker_set_precision", C.c_int, [vp, C.c_int]),
    ("sdvgn_tracker_set_arith", C.c_int, [vp, C.c_int]),
    ("sdvgn_tracker_set_team", C.c_int, [vp, C.c_int]),
    ("sdvgn_tracker_get_team", C.c_int, [vp]),
    ("sdvgn_tracker_records_dev", vp, [vp, C.c_int]),
    ("sdvgn_tracker_get_team_fallbacks", C.c_int, [vp]),
    ("sdvgn_tracker_res_and_gs_multi", C.c_int, [vp, C.c_int, C.c_int, vp, vp, f64p, f64p, C.c_float, vp]),
    ("sdvgn_tracker_ref_dev", vp, [vp, C.c_int]),
    ("sdvgn_tracker_make_K", C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("sdvgn_tracker_get_K", C.c_int, [vp, C.c_int, f32p, f32p]),
    ("sdvgn_tracker_set_ref", C.c_int, [vp, C.c_int, C.c_int, f32p, f32p, f32p, f32p]),
    ("sdvgn_tracker_set_ref_frame", C.c_int, [vp, C.c_float, C.c_double, C.c_double]),
    ("sdvgn_tracker_make_coarse_depth", C.c_int, [vp, C.c_int, i32p, i32p, f32p, f32p, vp]),
    ("sdvgn_tracker_get_ref", C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
    ("sdvgn_tracker_set_new_image", C.c_int, [vp, f32p, C.c_float]),
    ("sdvgn_tracker_set_new_image_dev", C.c_int, [vp, vp, C.c_float]),
    ("sdvgn_tracker_set_new_pyr", C.c_int, [vp, C.c_int, f32p, C.c_float]),
    ("sdvgn_tracker_get_pyr", C.c_int, [vp, C.c_int, f32p]),
    ("sdvgn_tracker_calc_res", C.c_int, [vp, C.c_int, f64p, C.c_double, C.c_double, C.c_float, f64p]),
    ("sdvgn_tracker_calc_gs", C.c_int, [vp, C.c_int, f64p, C.c_double, C.c_double, C.c_float, f64p, f64p]),
    ("sdvgn_tracker_res_and_gs", C.c_int, [vp, C.c_int, f64p, C.c_double, C.c_double, C.c_float, f64p, f64p, f64p]),
    ("sdvgn_tracker_get_point_terms", C.c_int, [vp, C.c_int, f32p, i32p]),
    ("sdvgn_tracker_track", C.c_int, [vp, f64p, f64p, C.c_int, f64p, f64p, f64p]),
    ("sdvgn_tracker_track_batch", C.c_int, [vp, C.c_int, f64p, f64p, C.c_int, vp, f64p, f64p, i32p]),
    ("sdvgn_tracker_get_trace", C.c_int, [vp, vp, C.c_int]),
    ("sdvgn_tracker_res_and_gs_batch", C.c_int, [vp, C.c_int, C.c_int, f64p, f64p, C.c_float, vp]),
    ("sdvgn_tracker_stream", vp, [vp]),
    ("sdvgn_tracker_pyr_dev", vp, [vp, C.c_int]),
    ("sdvgn_tracker_struct_pose", C.c_int, [vp, C.c_int, f32p, f32p, f32p, i32p, C.c_int, f64p, f64p, f64p, vp, vp]),
    ("sdvgn_tracker_struct_res_hb", C.c_int, [vp, C.c_int, f32p, f32p, f32p, i32p, C.c_int, f64p, f64p, f64p, f64p, f64p, vp, vp]),
    ("sdvgn_struct_trace_stride", C.c_int, []),
    ("sdvgn_tracker_trace_set_points", C.c_int, [vp, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, i32p]),
    ("sdvgn_tracker_trace_points", C.c_int, [vp, C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, i32p, f32p, f32p]),
]

_LIB = None


def load_library(path=LIB_PATH):
    """dlopen libsdvgn.so and bind every prototype.  No GPU is needed to load (only to create handles)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (same SONAME as /opt/rocm's).  When this
    # library is dlopen'ed first, torch later binds to a second runtime copy and sees no GPUs (RCCL refuses to start).
    # Importing torch first makes libsdvgn resolve libamdhip64 to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(path):
        raise ImportError("libsdvgn.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C sdv-loam_amd/csrc` (%s)" % path)
    L = C.CDLL(path)
    for name, res, args in PROTOTYPES + _extra_prototypes():
        fn = getattr(L, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def _extra_prototypes():
    out = []
    try:
        from .backend_api import PROTOTYPES as P2
        out += list(P2)
    except ImportError:
        pass
    try:
        from .reproject_api import PROTOTYPES as P3
        out += list(P3)
    except ImportError:
        pass
    return out


def check(rc):
    if rc < 0:
        raise RuntimeError("libsdvgn error %d: %s" % (rc, load_library().sdvgn_error_string(rc).decode()))
    return rc


IDENTITY_POSE = np.array([0, 0, 0, 1, 0, 0, 0], np.float64)


class CoarseTracker:
    """Python mirror of the reference's CoarseTracker surface (src/FullSystem/CoarseTracker.h:17-107) on the GPU."""

    def __init__(self, w, h, levels, max_points=None, max_batch=64, device=0, stream=None):
        self.L = load_library()
        self.w, self.h, self.levels = w, h, levels
        self.max_points = max_points or w * h
        self.max_batch = max_batch
        hnd = vp()
        check(self.L.sdvgn_tracker_create(C.byref(hnd), device, w, h, levels, self.max_points, max_batch, stream))
        self.h_ = hnd
        self.n = [0] * levels

    def close(self):
        if getattr(self, "h_", None):
            self.L.sdvgn_tracker_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ------------------------------------------------------------------------------
    def set_settings(self, huber=6.0, cutoff=20.0, aff_a=0.0, aff_b=0.0):
        check(self.L.sdvgn_tracker_set_settings(self.h_, huber, cutoff, aff_a, aff_b))

    def set_arith(self, mode):
        check(self.L.sdvgn_tracker_set_arith(self.h_, mode))

    def set_team(self, team):
        """Workgroups per hypothesis of trackBatch: 0 automatic, -1 one 1024-lane workgroup (k_track), 1..32 fixed (k_track_team)."""
        check(self.L.sdvgn_tracker_set_team(self.h_, team))

    def last_team(self):
        return self.L.sdvgn_tracker_get_team(self.h_)

    def team_fallbacks(self):
        """batches trackBatch re-ran on k_track because the members of a team did not meet (a busy device)"""
        return self.L.sdvgn_tracker_get_team_fallbacks(self.h_)

    def records_dev(self, lvl):
        """device pointer of the 64-byte-record copy of level lvl (set_precision(4)); built on demand"""
        return self.L.sdvgn_tracker_records_dev(self.h_, lvl)

    def ref_dev(self, lvl):
        return self.L.sdvgn_tracker_ref_dev(self.h_, lvl)

    def resAndGSMulti(self, lvl, pc_ptrs, img_ptrs, poses7, affs, cutoff, out_dev_ptr=None):
        """B independent problems in one launch: problem b = (template pc_ptrs[b], level image img_ptrs[b], pose b)."""
        poses = np.ascontiguousarray(np.array(poses7, np.float64).reshape(-1, 7))
        affs = np.ascontiguousarray(np.array(affs, np.float64).reshape(-1, 2))
        B = poses.shape[0]
        pa = (C.c_void_p * B)(*pc_ptrs)
        ia = (C.c_void_p * B)(*img_ptrs)
        check(self.L.sdvgn_tracker_res_and_gs_multi(self.h_, lvl, B, pa, ia, poses.reshape(-1), affs.reshape(-1), cutoff, out_dev_ptr))

    def set_precision(self, mode):
        check(self.L.sdvgn_tracker_set_precision(self.h_, mode))

    def makeK(self, fx, fy, cx, cy):
        check(self.L.sdvgn_tracker_make_K(self.h_, fx, fy, cx, cy))

    def get_K(self, lvl):
        k4 = np.zeros(4, np.float32)
        ki = np.zeros(9, np.float32)
        check(self.L.sdvgn_tracker_get_K(self.h_, lvl, k4, ki))
        return k4, ki.reshape(3, 3)

    def set_ref(self, lvl, u, v, idepth, color):
        u, v, idepth, color = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth, color))
        check(self.L.sdvgn_tracker_set_ref(self.h_, lvl, len(u), u, v, idepth, color))
        self.n[lvl] = len(u)

    def set_ref_frame(self, exposure=1.0, a=0.0, b=0.0):
        check(self.L.sdvgn_tracker_set_ref_frame(self.h_, exposure, a, b))

    def makeCoarseDepth(self, u, v, new_idepth, weight, ref_pyr_dev=None):
        """makeCoarseDepthL0 / makeCoarseDepthForFirstFrame from their splat tuples (CoarseTracker.cpp:108-425)."""
        u, v = (np.ascontiguousarray(x, np.int32) for x in (u, v))
        arr = None
        if ref_pyr_dev is not None:
            arr = (vp * len(ref_pyr_dev))(*ref_pyr_dev)
        check(self.L.sdvgn_tracker_make_coarse_depth(self.h_, len(u), u, v, np.ascontiguousarray(new_idepth, np.float32),
                                                     np.ascontiguousarray(weight, np.float32), arr))
        for l in range(self.levels):
            self.n[l] = self.L.sdvgn_tracker_get_ref(self.h_, l, None, None, None, None)

    def get_ref(self, lvl):
        n = check(self.L.sdvgn_tracker_get_ref(self.h_, lvl, None, None, None, None))
        out = [np.zeros(n, np.float32) for _ in range(4)]
        check(self.L.sdvgn_tracker_get_ref(self.h_, lvl, *[a.ctypes.data_as(vp) for a in out]))
        return dict(u=out[0], v=out[1], idepth=out[2], color=out[3])

    def set_new_image(self, color, exposure=1.0):
        color = np.ascontiguousarray(color, np.float32).reshape(-1)
        assert color.size == self.w * self.h
        check(self.L.sdvgn_tracker_set_new_image(self.h_, color, exposure))

    def set_new_image_dev(self, dev_ptr, exposure=1.0):
        check(self.L.sdvgn_tracker_set_new_image_dev(self.h_, dev_ptr, exposure))

    def set_new_pyr(self, lvl, aos3, exposure=1.0):
        aos3 = np.ascontiguousarray(aos3, np.float32).reshape(-1)
        check(self.L.sdvgn_tracker_set_new_pyr(self.h_, lvl, aos3, exposure))

    def get_pyr(self, lvl):
        out = np.zeros((self.h >> lvl) * (self.w >> lvl) * 3, np.float32)
        check(self.L.sdvgn_tracker_get_pyr(self.h_, lvl, out))
        return out.reshape(self.h >> lvl, self.w >> lvl, 3)

    # -- the two reference functions + fused form --------------------------------------------------------
    def calcRes(self, lvl, pose7, a, b, cutoff):
        out = np.zeros(6)
        check(self.L.sdvgn_tracker_calc_res(self.h_, lvl, np.ascontiguousarray(pose7, np.float64), a, b, cutoff, out))
        return out

    def calcGS(self, lvl, pose7, a, b, cutoff):
        H = np.zeros(64)
        bb = np.zeros(8)
        check(self.L.sdvgn_tracker_calc_gs(self.h_, lvl, np.ascontiguousarray(pose7, np.float64), a, b, cutoff, H, bb))
        return H.reshape(8, 8), bb

    def resAndGS(self, lvl, pose7, a, b, cutoff):
        out = np.zeros(6)
        H = np.zeros(64)
        bb = np.zeros(8)
        check(self.L.sdvgn_tracker_res_and_gs(self.h_, lvl, np.ascontiguousarray(pose7, np.float64), a, b, cutoff, out, H, bb))
        return out, H.reshape(8, 8), bb

    def point_terms(self, lvl):
        n = self.n[lvl]
        terms = np.zeros((8, n), np.float32)
        status = np.zeros(n, np.int32)
        check(self.L.sdvgn_tracker_get_point_terms(self.h_, lvl, terms.reshape(-1), status))
        return terms, status

    def warped(self, lvl):
        """The reference's buf_warped_* planes (compacted, zero-padded to a multiple of 4) from the parity hook."""
        terms, status = self.point_terms(lvl)
        sel = terms[:, status == 1]
        pad = (-sel.shape[1]) % 4
        return np.concatenate([sel, np.zeros((8, pad), np.float32)], axis=1), status

    # -- LM driver ----------------------------------------------------------------------------------
    def trackNewestCoarse(self, pose7, aff, coarsest, min_res=None):
        pose = np.array(pose7, np.float64)
        aff = np.array(aff, np.float64)
        mr = np.full(5, np.nan) if min_res is None else np.array(min_res, np.float64)
        last_res = np.zeros(5)
        flow = np.zeros(3)
        ok = check(self.L.sdvgn_tracker_track(self.h_, pose, aff, coarsest, mr, last_res, flow))
        n = self.L.sdvgn_tracker_get_trace(self.h_, None, 0)
        trace = np.zeros((max(n, 1), 15))
        self.L.sdvgn_tracker_get_trace(self.h_, trace.ctypes.data_as(vp), n)
        return bool(ok), pose, aff, last_res, flow, trace[:n]

    def trackBatch(self, poses7, affs, coarsest, min_res=None):
        poses = np.array(poses7, np.float64).reshape(-1, 7).copy()
        B = poses.shape[0]
        affs = np.array(affs, np.float64).reshape(B, 2).copy()
        mr = None if min_res is None else np.ascontiguousarray(np.array(min_res, np.float64).reshape(B, 5))
        last_res = np.zeros((B, 5))
        flow = np.zeros((B, 3))
        ok = np.zeros(B, np.int32)
        check(self.L.sdvgn_tracker_track_batch(self.h_, B, poses.reshape(-1), affs.reshape(-1), coarsest,
                                               None if mr is None else mr.ctypes.data_as(vp),
                                               last_res.reshape(-1), flow.reshape(-1), ok))
        return ok.astype(bool), poses, affs, last_res, flow

    def resAndGSBatch(self, lvl, poses7, affs, cutoff, out_dev_ptr=None):
        poses = np.ascontiguousarray(np.array(poses7, np.float64).reshape(-1, 7))
        affs = np.ascontiguousarray(np.array(affs, np.float64).reshape(-1, 2))
        check(self.L.sdvgn_tracker_res_and_gs_batch(self.h_, lvl, poses.shape[0], poses.reshape(-1), affs.reshape(-1),
                                                    cutoff, out_dev_ptr))

    def stream(self):
        return self.L.sdvgn_tracker_stream(self.h_)

    def pyr_dev(self, lvl):
        """Device pointer (ctypes c_void_p) of level `lvl` of the current new-frame pyramid, for zero-copy hand-over."""
        p = self.L.sdvgn_tracker_pyr_dev(self.h_, lvl)
        if not p:
            raise RuntimeError("no new-frame pyramid on this tracker (level %d)" % lvl)
        return vp(p)

    # -- structPoseEstimation (CoarseTracker.cpp:840-1007) ----------------------------------------------
    @staticmethod
    def _struct_args(u, v, idepth, host_idx, host_poses7, obs):
        u, v, idepth = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth))
        host_idx = np.ascontiguousarray(host_idx, np.int32)
        hp = np.ascontiguousarray(np.array(host_poses7, np.float64).reshape(-1, 7))
        obs = np.ascontiguousarray(np.array(obs, np.float64).reshape(-1, 2))
        assert len(u) == len(v) == len(idepth) == len(host_idx) == len(obs)
        return u, v, idepth, host_idx, hp, obs

    def structPoseEstimation(self, curToWorld7, u, v, idepth, host_idx, host_poses7, obs):
        u, v, idepth, host_idx, hp, obs = self._struct_args(u, v, idepth, host_idx, host_poses7, obs)
        pose = np.array(curToWorld7, np.float64)
        trace = np.zeros((10, self.L.sdvgn_struct_trace_stride()))
        fr = C.c_double(0)
        its = check(self.L.sdvgn_tracker_struct_pose(self.h_, len(u), u, v, idepth, host_idx, hp.shape[0], hp.reshape(-1),
                                                     obs.reshape(-1), pose, trace.ctypes.data_as(vp), C.cast(C.byref(fr), vp)))
        return pose, trace[:its], fr.value

    def structResHb(self, worldToCur7, u, v, idepth, host_idx, host_poses7, obs):
        u, v, idepth, host_idx, hp, obs = self._struct_args(u, v, idepth, host_idx, host_poses7, obs)
        H = np.zeros(36)
        b = np.zeros(6)
        e = C.c_double(0)
        n = C.c_int(0)
        check(self.L.sdvgn_tracker_struct_res_hb(self.h_, len(u), u, v, idepth, host_idx, hp.shape[0], hp.reshape(-1), obs.reshape(-1),
                                                 np.ascontiguousarray(worldToCur7, np.float64), H, b, C.cast(C.byref(e), vp),
                                                 C.cast(C.byref(n), vp)))
        return H.reshape(6, 6), b, e.value, n.value

    # -- ImmaturePoint::traceOn for all immature points (ImmaturePoint.cpp:47-353; FullSystem::traceNewCoarse) ----------
    def traceSetPoints(self, u, v, energyTH, gradH, color, weights, host_idx):
        u, v, energyTH = (np.ascontiguousarray(x, np.float32) for x in (u, v, energyTH))
        gradH = np.ascontiguousarray(gradH, np.float32).reshape(-1)
        color = np.ascontiguousarray(color, np.float32).reshape(-1)
        weights = np.ascontiguousarray(weights, np.float32).reshape(-1)
        self._trace_n = len(u)
        check(self.L.sdvgn_tracker_trace_set_points(self.h_, len(u), u, v, energyTH, gradH, color, weights, np.ascontiguousarray(host_idx, np.int32)))

    def tracePoints(self, KRKi, Kt, aff, idepth_min, idepth_max, quality, status, lastTraceUV=None, interval=None):
        n = self._trace_n
        KRKi = np.ascontiguousarray(KRKi, np.float32).reshape(-1, 9)
        st = dict(idepth_min=np.array(idepth_min, np.float32), idepth_max=np.array(idepth_max, np.float32), quality=np.array(quality, np.float32),
                  status=np.array(status, np.int32), lastTraceUV=np.zeros((n, 2), np.float32) if lastTraceUV is None else np.array(lastTraceUV, np.float32),
                  interval=np.zeros(n, np.float32) if interval is None else np.array(interval, np.float32))
        check(self.L.sdvgn_tracker_trace_points(self.h_, KRKi.shape[0], KRKi.reshape(-1), np.ascontiguousarray(Kt, np.float32).reshape(-1),
                                                np.ascontiguousarray(aff, np.float32).reshape(-1), st["idepth_min"], st["idepth_max"], st["quality"],
                                                st["status"], st["lastTraceUV"].reshape(-1), st["interval"]))
        return st
