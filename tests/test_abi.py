"""The C-ABI library loads without a GPU and exports every symbol include/sdvgn.h declares (CPU only)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if not fn.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(sdvgn_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol(sdvgn_lib):
    syms = declared_symbols()
    assert len(syms) >= 20
    raw = ctypes.CDLL(os.path.join(ROOT, "sdv-loam_amd", "libsdvgn.so"))
    missing = [s for s in syms if not hasattr(raw, s)]
    assert not missing, missing


def test_python_binding_covers_header(sdvgn_lib):
    from sdv_loam_amd import api
    bound = {p[0] for p in api.PROTOTYPES + api._extra_prototypes()}
    assert set(declared_symbols()) <= bound, sorted(set(declared_symbols()) - bound)


def test_version_and_error_strings(sdvgn_lib):
    assert b"gfx950" in sdvgn_lib.sdvgn_version()
    assert sdvgn_lib.sdvgn_error_string(0) == b"ok"
    assert b"argument" in sdvgn_lib.sdvgn_error_string(-10001)


def test_product_does_not_reference_oracle():
    """The product path must never route through oracle/ (only tests, smoke() and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "sdv-loam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".inc", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liborc" not in txt, f


def test_batch_entry_points_check_their_arguments_without_a_gpu(sdvgn_lib):
    """sdvgn_ef_optimize_batch / _lockstep refuse NULL tables, B out of range, NULL handles and a trace without a shape before anything touches
    a device (no compute call is made here)."""
    import ctypes as C
    L = sdvgn_lib
    E_ARG = -10001
    L.sdvgn_ef_optimize_lockstep.restype = C.c_int
    L.sdvgn_ef_optimize_lockstep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.sdvgn_ef_optimize_batch.restype = C.c_int
    L.sdvgn_ef_optimize_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    arr = (C.c_void_p * 2)(None, None)
    for B in (0, -1, 257):
        assert L.sdvgn_ef_optimize_lockstep(C.cast(arr, C.c_void_p), B, 6, 0, None, None, 0, 0) == E_ARG
        assert L.sdvgn_ef_optimize_batch(C.cast(arr, C.c_void_p), B, 6, 0, None) == E_ARG
    assert L.sdvgn_ef_optimize_lockstep(None, 2, 6, 0, None, None, 0, 0) == E_ARG
    assert L.sdvgn_ef_optimize_batch(None, 2, 6, 0, None) == E_ARG
    assert L.sdvgn_ef_optimize_lockstep(C.cast(arr, C.c_void_p), 2, 6, 0, None, None, 0, 0) == E_ARG      # NULL handles
    assert L.sdvgn_ef_optimize_batch(C.cast(arr, C.c_void_p), 2, 6, 0, None) == E_ARG
    assert L.sdvgn_ef_optimize_lockstep(C.cast(arr, C.c_void_p), 2, -1, 0, None, None, 0, 0) == E_ARG     # negative body count


def test_batch_entry_points_refuse_host_only_handles(sdvgn_lib):
    """Host-only handles (device -1: the CPU side of the multi-GPU logic) hold no device window: the batched optimize refuses them -- the
    lock-step form with SDVGN_E_ARG, the per-window form with the error of sdvgn_ef_optimize -- instead of falling back to anything on the CPU."""
    import ctypes as C
    L = sdvgn_lib
    hs = [C.c_void_p(), C.c_void_p()]
    for h in hs:
        assert L.sdvgn_ef_create(C.byref(h), -1, 64, 48, 16, None) == 0
    arr = (C.c_void_p * 2)(hs[0], hs[1])
    L.sdvgn_ef_optimize_lockstep.restype = C.c_int
    L.sdvgn_ef_optimize_lockstep.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.sdvgn_ef_optimize_batch.restype = C.c_int
    L.sdvgn_ef_optimize_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    assert L.sdvgn_ef_optimize_lockstep(C.cast(arr, C.c_void_p), 2, 6, 0, None, None, 0, 0) == -10001
    assert L.sdvgn_ef_optimize_batch(C.cast(arr, C.c_void_p), 2, 6, 0, None) < 0
    assert L.sdvgn_ef_optimize_batch(C.cast(arr, C.c_void_p), 1, 6, 0, None) < 0
    for h in hs:
        L.sdvgn_ef_destroy(h)


def test_c_host_loop_harness_builds_against_the_header_alone(sdvgn_lib, tmp_path):
    """tools/kf_host_loop.c (the bench's key-frame cycle from a plain-C host loop) is a consumer of include/sdvgn.h and nothing else: it compiles
    as C99 with -Wall -Werror against the header, links libsdvgn.so and exports kf_host_loop (no GPU needed to load it)."""
    import subprocess
    out = tmp_path / "libkfloop.so"
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tools", "kf_host_loop.c"), "-L", os.path.join(ROOT, "sdv-loam_amd"), "-lsdvgn",
                           "-Wl,-rpath," + os.path.join(ROOT, "sdv-loam_amd"), "-o", str(out)])
    lib = ctypes.CDLL(str(out))
    assert hasattr(lib, "kf_host_loop")
