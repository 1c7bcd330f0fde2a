"""ctypes wrapper of the CPU oracle's ImmaturePoint::traceOn restatement (oracle/orc_trace.cpp).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from . import lib

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_SIG = [C.c_int, f32p, f32p, f32p, f32p, f32p, f32p, i32p, f32p, f32p, f32p, f32p, C.c_int, C.c_int, f32p, f32p, f32p, i32p, f32p, f32p]


def trace_on(P, dI_aos3, idepth_min, idepth_max, quality, status, lastTraceUV=None, interval=None, reference=False):
    """P: TraceProblem-like (u, v, energyTH, gradH, color, weights, host_idx, KRKi, Kt, aff, w, h).  Returns the updated state.
    reference=True: the REFERENCE'S OWN ImmaturePoint::traceOn (oracle/_ref/libref.so, oracle/ref_glue_misc.cpp)."""
    if reference:
        from . import refpin
        fn = refpin.ref_lib().ref_trace_on
    else:
        fn = lib().orc_trace_on
    fn.argtypes = _SIG
    fn.restype = None
    n = len(P.u)
    st = dict(idepth_min=np.array(idepth_min, np.float32), idepth_max=np.array(idepth_max, np.float32), quality=np.array(quality, np.float32),
              status=np.array(status, np.int32), lastTraceUV=np.zeros((n, 2), np.float32) if lastTraceUV is None else np.array(lastTraceUV, np.float32),
              interval=np.zeros(n, np.float32) if interval is None else np.array(interval, np.float32))
    c = lambda a: np.ascontiguousarray(a, np.float32).reshape(-1)   # noqa: E731
    fn(n, c(P.u), c(P.v), c(P.energyTH), c(P.gradH), c(P.color), c(P.weights), np.ascontiguousarray(P.host_idx, np.int32),
                   c(P.KRKi), c(P.Kt), c(P.aff), c(dI_aos3), P.w, P.h, st["idepth_min"], st["idepth_max"], st["quality"], st["status"],
                   st["lastTraceUV"].reshape(-1), st["interval"])
    return st
