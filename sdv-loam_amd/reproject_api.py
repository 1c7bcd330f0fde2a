"""ctypes binding of the sdvgn_reproj_* entry points + the host part of class Reprojector (src/FullSystem/Reprojector.h:17-112).

The kernel evaluates reprojectPoint + findMatchDirect for every candidate; `select_matches` replays the reference's grid walk
(reprojectMap / reprojectCell, Reprojector.cpp:117-156,196-234) on those arrays.  Used by tests and bench.py only."""
import ctypes as C

import numpy as np

from .api import check, f32p, f64p, i32p, load_library, vp

PROTOTYPES = [
    ("sdvgn_reproj_create", C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    ("sdvgn_reproj_destroy", None, [vp]),
    ("sdvgn_reproj_stream", vp, [vp]),
    ("sdvgn_reproj_set_calib", C.c_int, [vp, C.c_float, C.c_float, C.c_float, C.c_float]),
    ("sdvgn_reproj_set_frame", C.c_int, [vp, C.c_int, f64p, vp, vp, C.c_float, C.c_double, C.c_double]),
    ("sdvgn_reproj_set_cur", C.c_int, [vp, f64p, C.c_float, C.c_double, C.c_double]),
    ("sdvgn_reproj_set_cur_level", C.c_int, [vp, C.c_int, vp, vp]),
    ("sdvgn_reproj_match", C.c_int, [vp, C.c_int, f32p, f32p, f32p, i32p, i32p, i32p, f64p, i32p, f32p, i32p, f64p, i32p]),
]

CELL_SIZE = 25   # Reprojector::initializeGrid, Reprojector.cpp:96


class Reprojector:
    def __init__(self, w, h, levels, max_frames=8, max_points=20000, device=0, stream=None):
        self.L = load_library()
        self.w, self.h, self.levels = w, h, levels
        hnd = vp()
        check(self.L.sdvgn_reproj_create(C.byref(hnd), device, w, h, levels, max_frames, max_points, stream))
        self.h_ = hnd

    def close(self):
        if getattr(self, "h_", None):
            self.L.sdvgn_reproj_destroy(self.h_)
            self.h_ = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_calib(self, fx, fy, cx, cy):
        check(self.L.sdvgn_reproj_set_calib(self.h_, fx, fy, cx, cy))

    def set_frame(self, idx, camToWorld7, dI_aos3=None, exposure=1.0, a=0.0, b=0.0, dev_ptr=None):
        img = None if dI_aos3 is None else np.ascontiguousarray(dI_aos3, np.float32).reshape(-1)
        check(self.L.sdvgn_reproj_set_frame(self.h_, idx, np.ascontiguousarray(camToWorld7, np.float64),
                                            None if img is None else img.ctypes.data_as(vp), dev_ptr, exposure, a, b))

    def set_cur(self, camToWorld7, pyr_aos3=None, exposure=1.0, a=0.0, b=0.0, dev_ptrs=None):
        check(self.L.sdvgn_reproj_set_cur(self.h_, np.ascontiguousarray(camToWorld7, np.float64), exposure, a, b))
        if dev_ptrs is not None:
            for l, p in enumerate(dev_ptrs):
                check(self.L.sdvgn_reproj_set_cur_level(self.h_, l, None, p))
        elif pyr_aos3 is not None:
            for l, img in enumerate(pyr_aos3):
                img = np.ascontiguousarray(img, np.float32).reshape(-1)
                check(self.L.sdvgn_reproj_set_cur_level(self.h_, l, img.ctypes.data_as(vp), None))

    def match(self, u, v, idepth, host_idx, ref_idx, ptype):
        u, v, idepth = (np.ascontiguousarray(x, np.float32) for x in (u, v, idepth))
        host_idx, ref_idx, ptype = (np.ascontiguousarray(x, np.int32) for x in (host_idx, ref_idx, ptype))
        n = len(u)
        px0 = np.zeros((n, 2))
        px = np.zeros((n, 2))
        cell = np.zeros(n, np.int32)
        q = np.zeros(n, np.float32)
        ok = np.zeros(n, np.int32)
        lvl = np.zeros(n, np.int32)
        check(self.L.sdvgn_reproj_match(self.h_, n, u, v, idepth, host_idx, ref_idx, ptype, px0.reshape(-1), cell, q, ok, px.reshape(-1), lvl))
        return dict(px0=px0, cell=cell, quality=q, success=ok.astype(bool), px=px, level=lvl)


def select_matches(cell, quality, success, px, order, cell_order, max_matches, active=None):
    """Host replay of reprojectMap / reprojectCell (Reprojector.cpp:117-156,196-234).
    order: candidate indices in the order reprojectPoint pushed them (key-frames by distance to the new frame, points in
    pointHessians order); cell_order: the shuffled cell permutation (:101-104); returns the list of (candidate, px) pairs = overlap_pts.
    Each cell's list is sorted with pointQualityComparator (ascending gradient norm, std::list::sort is stable) and walked until the
    first candidate whose findMatchDirect succeeded; the walk over cells stops when n_matches > max_matches."""
    n_cells = len(cell_order)
    cells = [[] for _ in range(n_cells)]
    for i in order:
        if cell[i] >= 0:
            cells[cell[i]].append(i)
    out = []
    n_matches = 0
    for k in cell_order:
        cand = sorted(cells[k], key=lambda i: quality[i])          # stable, like std::list::sort
        for i in cand:
            if active is not None and not active[i]:
                continue
            if success[i]:
                out.append((i, px[i].copy()))
                n_matches += 1
                break
        if n_matches > max_matches:
            break
    return out
