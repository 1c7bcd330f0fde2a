"""The oracle's restatements of Accumulator9, AccumulatorApprox, Accumulator11, getInterpolatedElement33 / 31 and
AffLight::fromToVecExposure against the REFERENCE's own code: oracle/_ref/libref.so is built from three reference headers compiled
unmodified (oracle/ref_glue.cpp, oracle/ref_shim); tests/golden/ref_pin.npz holds that library's outputs for machines without
/root/reference (tools/gen_ref_pin_golden.py).  Bit-exact: these are float sums in a prescribed order."""
import os

import numpy as np
import pytest

from oracle import refpin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pin.npz")


def _compare(ref, orc):
    for name, want in ref.items():
        if name.startswith("settings_") or name.startswith("scale_") or name == "pattern":
            continue
        got = [v for k, v in orc.items() if k == name or k.startswith(name + ":")]
        assert got, name
        for g in got:
            assert g.shape == want.shape and np.array_equal(g, want), name


def test_oracle_matches_reference_fixture():
    ref = dict(np.load(GOLD))
    _compare(ref, refpin.run(None, None))


def test_oracle_matches_reference_library():
    L = refpin.ref_lib()
    if L is None:
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference: make -C oracle ref)")
    ref = refpin.run(L, "ref_")
    _compare(ref, refpin.run(None, None))
    gold = dict(np.load(GOLD))                      # and the committed fixture IS what the reference code produces
    assert set(gold) - {"settings_names", "settings_values", "scale_names", "scale_values", "pattern"} == set(ref)
    for k in ref:
        assert np.array_equal(ref[k], gold[k]), k
    st = refpin.ref_settings(L)
    assert list(gold["settings_names"]) == sorted(st) and np.array_equal(gold["settings_values"], [st[k] for k in sorted(st)])


def test_literal_settings_match_reference_settings_cpp():
    """Every `setting_xxx = literal` the oracle and the product carry equals the value the reference's own settings.cpp defines (compiled
    unmodified into oracle/_ref; values from the committed fixture).  float settings compare as float32."""
    import glob
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    gold = np.load(GOLD)
    ref = dict(zip([str(n) for n in gold["settings_names"]], gold["settings_values"]))
    srcs = glob.glob(os.path.join(root, "oracle", "orc_*.cpp")) + glob.glob(os.path.join(root, "oracle", "*.hpp")) + \
        glob.glob(os.path.join(root, "sdv-loam_amd", "csrc", "*"))
    found = refpin.source_settings(srcs)
    checked = 0
    for fname, name, val in found:
        base = name
        while base not in ref and "_" in base:          # a suffixed second copy (setting_huberTH_imm) checks against its base name
            base = base.rsplit("_", 1)[0]
        if base not in ref:
            continue
        assert np.float32(val) == np.float32(ref[base]), (fname, name, val, ref[base])
        checked += 1
    assert checked >= 30, checked
    # the product's own names for some of them
    import re
    alias = {"kInitialRotPrior": "setting_initialRotPrior", "kInitialTransPrior": "setting_initialTransPrior",
             "kInitialCalibHessian": "setting_initialCalibHessian", "kIdepthFixPrior": "setting_idepthFixPrior",
             "huberTH": "setting_huberTH", "coarseCutoffTH": "setting_coarseCutoffTH"}
    text = open(os.path.join(root, "sdv-loam_amd", "csrc", "backend.hip")).read() + open(os.path.join(root, "sdv-loam_amd", "csrc", "tracker.hip")).read()
    for name, ref_name in alias.items():
        m = re.search(r"\b%s\s*=\s*([-+0-9.eE* ]+)f?\s*[,;]" % name, text)
        assert m, name
        assert np.float32(eval(m.group(1), {"__builtins__": {}})) == np.float32(ref[ref_name]), (name, m.group(1), ref[ref_name])


def test_tier_shift_matters():
    """the 16000-entry float sum is NOT the sequential float sum: the accumulator's tiers are part of the arithmetic being pinned"""
    n, v = refpin.cases()["acc11_tiers"]
    seq = np.float32(0)
    for x in v:
        seq = np.float32(seq + x)
    ref = dict(np.load(GOLD))["acc11_tiers"][0]
    assert ref != seq and abs(float(ref) - float(seq)) / float(seq) < 1e-4


def test_scale_macros_match_reference_header():
    """SCALE_* (HessianBlocks.h:33-40, read from the header text into the fixture) against the product's SDVGN_SCALE_* and the oracle's SCALE_*"""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    gold = np.load(GOLD)
    ref = dict(zip([str(n) for n in gold["scale_names"]], gold["scale_values"]))
    assert len(ref) == 8
    checked = 0
    for p in glob.glob(os.path.join(root, "sdv-loam_amd", "csrc", "*")) + glob.glob(os.path.join(root, "oracle", "orc_*")):
        text = open(p).read()
        for name, val in re.findall(r"\b(?:SDVGN_)?(SCALE_[A-Z_]+)\s*(?:=\s*)?([0-9.]+)f", text):
            if name in ref:
                assert float(val) == ref[name], (p, name, val)
                checked += 1
    assert checked >= 10, checked


def test_residual_pattern_matches_reference():
    """patternP = staticPattern[8] of the reference's settings.cpp (from the compiled object, via the fixture) against every copy of the 8
    offsets in the oracle and in the kernels"""
    import glob
    import re
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    want = np.load(GOLD)["pattern"]
    assert want.shape == (8, 2)
    copies = 0
    for p in glob.glob(os.path.join(root, "sdv-loam_amd", "csrc", "*")) + glob.glob(os.path.join(root, "oracle", "orc_*")):
        for m in re.finditer(r"(?:pat|patternP)\[8\]\[2\]\s*=\s*\{((?:\s*\{\s*-?\d+\s*,\s*-?\d+\s*\}\s*,?)+)\}", open(p).read()):
            got = np.array(re.findall(r"-?\d+", m.group(1)), np.int32).reshape(-1, 2)
            assert np.array_equal(got, want), p
            copies += 1
    assert copies >= 4, copies
