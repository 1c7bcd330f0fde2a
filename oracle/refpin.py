"""Reference-pinned checks -- TEST INFRASTRUCTURE ONLY.

`oracle/_ref/libref.so` is code of the REFERENCE ITSELF: three of its headers (src/util/NumType.h, src/util/globalFuncs.h,
src/OptimizationBackend/MatrixAccumulators.h), compiled unmodified from /root/reference against a stand-in for Eigen's storage types
(oracle/ref_shim, oracle/ref_glue.cpp, `make -C oracle ref`).  This module drives that library and the oracle's restatements of the same
functions / classes (the orc_kat_* hooks of liborc.so) with the same inputs; tests/test_ref_pin.py asserts bit-identical outputs, and
tests/golden/ref_pin.npz holds the reference's outputs (tools/gen_ref_pin_golden.py) for machines without /root/reference.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")

_SIG = {
    "aff_from_to": [C.c_float, C.c_float, C.c_double, C.c_double, C.c_double, C.c_double, f64p],
    "interp33": [f32p, C.c_int, C.c_int, f32p, f32p, f32p],
    "interp31": [f32p, C.c_int, C.c_int, f32p, f32p, f32p],
    "acc9": [C.c_int, f32p, f32p, f32p, f64p],
    "acc_approx": [C.c_int, f32p, f32p, f64p],
    "acc11": [C.c_int, f32p, f32p],
}
# oracle hook(s) restating each reference function: every translation unit of the oracle has its own copy of the bilinear taps
ORC_HOOKS = {
    "aff_from_to": ["orc_kat_aff_from_to"],
    "interp33": ["orc_kat_interp33", "orc_kat_interp33_backend", "orc_kat_interp33_trace"],
    "interp31": ["orc_kat_interp31_trace", "orc_kat_interp31_reproject"],
    "acc9": ["orc_kat_acc9"],
    "acc_approx": ["orc_kat_acc_approx"],
    "acc11": ["orc_kat_acc11"],
}


def ref_path():
    return os.path.join(_HERE, "_ref", "libref.so")


_REF = None


def ref_lib():
    """the reference-built library, or None when it has not been built (no /root/reference on this machine and no prebuilt file)"""
    global _REF
    if _REF is not None:
        return _REF
    p = ref_path()
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    _REF = L
    for name, sig in _SIG.items():
        fn = getattr(L, "ref_" + name)
        fn.argtypes = sig
        fn.restype = None
    return L


def ref_settings(L):
    """{name: value} of the setting_* constants the reference's settings.cpp defines (compiled unmodified into libref.so)"""
    L.ref_setting_name.restype = C.c_char_p
    L.ref_setting.argtypes = [C.c_char_p, C.POINTER(C.c_double)]
    out = {}
    v = C.c_double()
    for i in range(L.ref_setting_count()):
        nm = L.ref_setting_name(i)
        assert L.ref_setting(nm, C.byref(v)) == 1
        out[nm.decode()] = v.value
    return out


def source_settings(paths):
    """`setting_xxx = <literal expression>` declarations found in source files -> [(file, name, value)]; the oracle and the product keep the
    reference's names for the constants they carry as literals (a suffix after the reference's name, e.g. _imm, marks a second copy)."""
    import re
    found = []
    pat = re.compile(r"\b(setting_[A-Za-z0-9_]+)\s*=\s*([-+0-9.eE*f ()]+?)\s*[,;]")
    for p in paths:
        for m in pat.finditer(open(p).read()):
            expr = re.sub(r"(?<=[0-9.])f\b", "", m.group(2))
            try:
                val = float(eval(expr, {"__builtins__": {}}))
            except Exception:  # noqa: BLE001
                continue
            found.append((os.path.basename(p), m.group(1), val))
    return found


def orc_fn(name):
    import oracle
    fn = getattr(oracle.lib(), name)
    key = [k for k, v in ORC_HOOKS.items() if name in v][0]
    fn.argtypes = _SIG[key]
    fn.restype = None
    return fn


def cases(seed=0):
    """the input sets of the pin: {case name: (kind, args...)}; sizes beyond 1000 entries exercise the accumulators' tier shifts"""
    rng = np.random.default_rng(seed)
    out = {}
    w, h = 97, 61
    img = rng.normal(100, 40, (h, w, 3)).astype(np.float32)
    n = 4000
    x = rng.uniform(0, w - 1.001, n).astype(np.float32)
    y = rng.uniform(0, h - 1.001, n).astype(np.float32)
    x[:8] = np.float32([0, 1, 2.5, w - 2, 3, 7.25, 0.999, 12])            # integer and near-integer positions
    y[:8] = np.float32([0, 1, 3.5, h - 2, 4.75, 9, 0.001, 12])
    out["interp"] = (img, w, x, y)
    out["aff"] = [(1.0, 1.0, 0.0, 0.0, 0.0, 0.0), (1.0, 1.0, 0.02, -3.0, -0.04, 5.5), (0.8, 1.3, 0.3, 10.0, -0.2, -7.0), (0.0, 1.5, 0.1, 1.0, 0.2, 2.0),
                  (2.0, 0.0, -0.1, 1.0, 0.05, -2.0)]
    for tag, n4 in (("small", 7), ("tiers", 1501)):                     # 1501 groups > 1000: S -> S1k shift; 4 x 45 lanes
        J = rng.normal(0, 30, (9, 4 * n4)).astype(np.float32)
        J[7] = -1.0
        wts = rng.uniform(0.05, 1.0, 4 * n4).astype(np.float32)
        out["acc9_" + tag] = (n4, J, wts)
    for tag, n in (("small", 50), ("tiers", 2600)):
        a = rng.normal(0, 3, (n, 35)).astype(np.float32)
        out["acc_approx_" + tag] = (n, a)
    for tag, n in (("small", 300), ("tiers", 16000)):                   # 16000: what calcLEnergyPt sums at the named window size
        v = (rng.uniform(0, 1, n) ** 4 * 50).astype(np.float32)
        out["acc11_" + tag] = (n, v)
    return out


def run(L, prefix, hooks=None):
    """all cases through one library (`L` = ref_lib(), prefix "ref_"; or the oracle with prefix None and the hook table) -> {name: array}"""
    cs = cases()
    res = {}

    def fn(kind, k=0):
        if prefix is not None:
            return getattr(L, prefix + kind)
        return orc_fn(ORC_HOOKS[kind][k])

    def variants(kind):
        return range(1) if prefix is not None else range(len(ORC_HOOKS[kind]))

    img, w, x, y = cs["interp"]
    flat = np.ascontiguousarray(img.reshape(-1))
    for k in variants("interp33"):
        o = np.zeros(3 * len(x), np.float32)
        fn("interp33", k)(flat, w, len(x), x, y, o)
        res["interp33" + ("" if prefix is not None else ":" + ORC_HOOKS["interp33"][k])] = o
    for k in variants("interp31"):
        o = np.zeros(len(x), np.float32)
        fn("interp31", k)(flat, w, len(x), x, y, o)
        res["interp31" + ("" if prefix is not None else ":" + ORC_HOOKS["interp31"][k])] = o
    ab = np.zeros((len(cs["aff"]), 2))
    for i, a in enumerate(cs["aff"]):
        o = np.zeros(2)
        fn("aff_from_to")(*a, o)
        ab[i] = o
    res["aff_from_to"] = ab
    for name, c in cs.items():
        if name.startswith("acc9_"):
            H = np.zeros(81, np.float32); num = np.zeros(1)
            fn("acc9")(c[0], np.ascontiguousarray(c[1].reshape(-1)), c[2], H, num)
            res[name] = np.concatenate([H, num.astype(np.float32)])
        elif name.startswith("acc_approx_"):
            H = np.zeros(169, np.float32); num = np.zeros(1)
            fn("acc_approx")(c[0], np.ascontiguousarray(c[1].reshape(-1)), H, num)
            res[name] = np.concatenate([H, num.astype(np.float32)])
        elif name.startswith("acc11_"):
            A = np.zeros(1, np.float32)
            fn("acc11")(c[0], c[1], A)
            res[name] = A
    return res
