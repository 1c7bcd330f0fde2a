#!/usr/bin/env python3
"""tests/golden/ref_pin.npz = outputs of the REFERENCE's own code (oracle/_ref/libref.so: three reference headers compiled unmodified, see
oracle/ref_glue.cpp) for the input sets of oracle/refpin.py.  Run in the container that has /root/reference:
    make -C oracle ref && python tools/gen_ref_pin_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import refpin  # noqa: E402

L = refpin.ref_lib()
if L is None:
    sys.exit("oracle/_ref/libref.so is missing: make -C oracle ref (needs /root/reference)")
out = refpin.run(L, "ref_")
st = refpin.ref_settings(L)
out["settings_names"] = np.array(sorted(st))
out["settings_values"] = np.array([st[k] for k in sorted(st)])
import ctypes as C  # noqa: E402
pat = (C.c_int * 80)()
npat = L.ref_pattern(pat)
out["pattern"] = np.array(list(pat)[:2 * npat], np.int32).reshape(npat, 2)
# the SCALE_* macros of src/FullSystem/HessianBlocks.h:33-40 (a header that cannot be compiled here: read as text)
import re  # noqa: E402
hb = open("/root/reference/src/FullSystem/HessianBlocks.h").read()
sc = dict(re.findall(r"#define (SCALE_[A-Z_]+) ([0-9.]+)f", hb))
out["scale_names"] = np.array(sorted(sc))
out["scale_values"] = np.array([float(sc[k]) for k in sorted(sc)])
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_pin.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: v.shape for k, v in out.items()})
