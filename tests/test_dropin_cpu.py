"""oracle/_ref/libref_dropin.so without a GPU: it loads (against the in-tree libsdvgn.so), the two replaced member functions are the GPU-backed
definitions of oracle/dropin/*.cpp (they are what calls the C ABI), and nothing else of the reference was replaced."""
import os
import subprocess

import pytest


def _lib():
    from oracle import dropin
    return dropin.dropin_lib()


needs_dropin = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                                                                  "libref_dropin.so")), reason="oracle/_ref/libref_dropin.so not built here")

SSF = "_ZN8sdv_loam16EnergyFunctional12solveSystemFEidPNS_12CalibHessianE"
TNC = "_ZN8sdv_loam13CoarseTracker17trackNewestCoarseE"


@needs_dropin
def test_dropin_library_loads_and_binds_the_c_abi(sdvgn_lib):
    from oracle import dropin
    L = _lib()
    assert L is not None
    for name in ("ref_ef_create", "ref_ef_optimize_full", "ref_track", "sdvgn_dropin_ef_calls", "sdvgn_dropin_tracker_calls"):
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", dropin.dropin_path()], text=True)
    undefined = {ln.split()[-1] for ln in out.splitlines() if " U " in ln}
    for name in ("sdvgn_ef_solve_system", "sdvgn_ef_set_residual_jacobians", "sdvgn_ef_set_residuals", "sdvgn_tracker_track", "sdvgn_tracker_set_ref"):
        assert name in undefined, name                     # the drop-in reaches the product only through include/sdvgn.h's entry points
    defined = [ln.split()[-1] for ln in out.splitlines() if " T " in ln or " W " in ln]
    assert SSF in defined and any(s.startswith(TNC) for s in defined)
    needed = subprocess.check_output(["readelf", "-d", dropin.dropin_path()], text=True)
    assert "libsdvgn.so" in needed


@needs_dropin
def test_only_the_two_members_were_replaced(sdvgn_lib):
    """the all-CPU library and the drop-in export the same reference symbols; the replaced two call sdvgn_* in the drop-in only"""
    from oracle import dropin, refpin
    def syms(p):
        out = subprocess.check_output(["nm", "-D", "--defined-only", p], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith(("_ZN8sdv_loam", "_ZNK8sdv_loam"))}
    a, b = syms(refpin.ref_path()), syms(dropin.dropin_path())
    assert a == b
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", dropin.dropin_path()], text=True)
    def calls(sym_prefix):
        body, on = [], False
        for ln in dis.splitlines():
            if ln.endswith(">:"):
                on = ("<" + sym_prefix) in ln
            elif on and "call" in ln:
                body.append(ln)
        return body
    assert any("sdvgn_ef_solve_system" in c for c in calls(SSF))
    assert any("sdvgn_tracker_track" in c for c in calls(TNC))


OPT = "_ZN8sdv_loam10FullSystem8optimizeEi"
needs_dropin_opt = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                                                                      "libref_dropin_opt.so")), reason="oracle/_ref/libref_dropin_opt.so not built here")


@needs_dropin_opt
def test_form_b_replaces_exactly_full_system_optimize(sdvgn_lib):
    """oracle/_ref/libref_dropin_opt.so (INTEGRATION.md form B) without a GPU: same reference symbols as the all-CPU library, FullSystem::optimize is
    the definition of oracle/dropin/FullSystemOptimizeGPU.cpp -- it calls sdvgn_ef_optimize / _optimize_finish / _make_idx and the edit entry points,
    and reaches the product through include/sdvgn.h alone."""
    from oracle import dropin, refpin
    path = os.path.join(os.path.dirname(dropin.dropin_path()), "libref_dropin_opt.so")
    L = dropin.dropin_opt_lib()
    assert L is not None
    for name in ("sdvgn_dropin_opt_calls", "sdvgn_dropin_opt_stats", "sdvgn_dropin_opt_release", "ref_ef_keyframe_tail"):
        assert hasattr(L, name), name
    def syms(p):      # functions and members of namespace sdv_loam (std:: template instances over its types are the drop-in's own containers)
        out = subprocess.check_output(["nm", "-D", "--defined-only", p], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith(("_ZN8sdv_loam", "_ZNK8sdv_loam"))}
    assert syms(refpin.ref_path()) == syms(path)
    out = subprocess.check_output(["nm", "-D", path], text=True)
    undefined = {ln.split()[-1] for ln in out.splitlines() if " U " in ln}
    for name in ("sdvgn_ef_optimize", "sdvgn_ef_optimize_finish", "sdvgn_ef_make_idx", "sdvgn_ef_insert_frame", "sdvgn_ef_insert_points", "sdvgn_ef_insert_residuals",
                 "sdvgn_ef_remove_points", "sdvgn_ef_remove_frame", "sdvgn_ef_drop_residuals", "sdvgn_ef_update_residuals", "sdvgn_ef_get_residual_table"):
        assert name in undefined, name
    assert not [u for u in undefined if u.startswith("orc_")]          # nothing of the oracle port
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", path], text=True)
    body, on = [], False
    for ln in dis.splitlines():
        if ln.endswith(">:"):
            on = ("<" + OPT) in ln
        elif on and "call" in ln:
            body.append(ln)
    assert any("sdvgn_ef_optimize@" in c or "sdvgn_ef_optimize>" in c for c in body) and any("sdvgn_ef_optimize_finish" in c for c in body)


FRAME_MEMBERS = {                                   # reference member -> the product entry point its form B+ definition must call
    "_ZN8sdv_loam10FullSystem14traceNewCoarseEPNS_12FrameHessianE": "sdvgn_tracker_trace_points",
    "_ZN8sdv_loam10FullSystem25activatePointsMT_Reductor": "sdvgn_ef_optimize_immature",
    "_ZN8sdv_loam13CoarseTracker17makeCoarseDepthL0E": "sdvgn_tracker_make_coarse_depth",
    "_ZN8sdv_loam13CoarseTracker20structPoseEstimationE": "sdvgn_tracker_struct_pose",
    "_ZN8sdv_loam11Reprojector12reprojectMapE": "sdvgn_reproj_match",
}
needs_dropin_frame = pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref",
                                                                        "libref_dropin_frame.so")), reason="oracle/_ref/libref_dropin_frame.so not built here")


@needs_dropin_frame
def test_form_b_plus_replaces_the_five_frame_level_members(sdvgn_lib):
    """oracle/_ref/libref_dropin_frame.so (INTEGRATION.md form B+) without a GPU: the same sdv_loam symbols as the all-CPU library; each of the five
    frame-level members (SURVEY.md section 8 f-rows) is the definition of oracle/dropin/FullSystemFrameGPU.cpp and calls its sdvgn_* entry point;
    nothing of the oracle port is linked."""
    from oracle import dropin, refpin
    path = os.path.join(os.path.dirname(dropin.dropin_path()), "libref_dropin_frame.so")
    L = dropin.dropin_frame_lib()
    assert L is not None
    for name in ("sdvgn_dropin_frame_stats", "sdvgn_dropin_frame_release", "sdvgn_dropin_opt_stats", "ref_ef_trace_new_frame", "ref_ef_activate_points"):
        assert hasattr(L, name), name
    def syms(p):
        out = subprocess.check_output(["nm", "-D", "--defined-only", p], text=True)
        return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith(("_ZN8sdv_loam", "_ZNK8sdv_loam"))}
    assert syms(refpin.ref_path()) == syms(path)
    out = subprocess.check_output(["nm", "-D", path], text=True)
    undefined = {ln.split()[-1] for ln in out.splitlines() if " U " in ln}
    assert not [u for u in undefined if u.startswith("orc_")]
    strong = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    dis = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", path], text=True)
    calls, cur = {}, None
    for ln in dis.splitlines():
        if ln.endswith(">:"):
            cur = ln.split("<", 1)[1][:-2]
        elif cur is not None and "call" in ln:
            calls.setdefault(cur, []).append(ln)
    for member, entry in FRAME_MEMBERS.items():
        full = [s for s in strong if s.startswith(member)]
        assert len(full) == 1, (member, full)                     # defined once, strongly (the reference's own copy was weakened and lost)
        assert entry in undefined, entry
        assert any(entry in c for c in calls.get(full[0], [])), (member, entry)
