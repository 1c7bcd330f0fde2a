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
    assert set(gold) == set(ref)
    for k in ref:
        assert np.array_equal(ref[k], gold[k]), k


def test_tier_shift_matters():
    """the 16000-entry float sum is NOT the sequential float sum: the accumulator's tiers are part of the arithmetic being pinned"""
    n, v = refpin.cases()["acc11_tiers"]
    seq = np.float32(0)
    for x in v:
        seq = np.float32(seq + x)
    ref = dict(np.load(GOLD))["acc11_tiers"][0]
    assert ref != seq and abs(float(ref) - float(seq)) / float(seq) < 1e-4
