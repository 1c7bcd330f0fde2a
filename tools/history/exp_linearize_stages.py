#!/usr/bin/env python3
"""Stage timeline of k_ef_linearize from wall_clock64() stamps written by lane 0 of every wave.

usage (GPU box):  SDVGN_DEBUG_FLAGS=32 python tools/exp_linearize_stages.py      # bit5 = stamps
Stamps: 0 wave start | 1 slot / point loads arrived | 2 pattern projection done | 7 the 16 tap loads issued | 3 centre projection +
Jacobian row done | 4 taps arrived | 5 taps consumed (stores issued next) | 6 all stores acknowledged, wave end.
wall_clock64() ticks are 10 ns (constant 100 MHz)."""
import os
import sys
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("SDVGN_DEBUG_FLAGS", "32")
import torch  # noqa: E402,F401
from sdv_loam_amd import backend_api, synthetic as syn  # noqa: E402

W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)
G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
for _ in range(5):
    G.linearizeAll(want_energy=False)
torch.cuda.synchronize()
buf = np.zeros(8 * 8 * 2 * 64 * 4 * 8, np.uint64)
n = G.L.sdvgn_debug_read_stamps(G.h_, buf.ctypes.data_as(C.c_void_p), buf.size)
assert n > 0, "diagnostics off (SDVGN_DEBUG_FLAGS bit5)"
st = buf[:n].reshape(-1, 8).astype(np.int64)
role = (np.arange(len(st)) % 4) % 2          # wave 2g + r of a workgroup has role r
keep = st[:, 0] > 0
st, role = st[keep], role[keep]
tick_us = 1.0 / 100.0                       # wall_clock64(): constant 100 MHz
names = ["start", "slot loads in", "pattern proj done", "16 tap loads issued", "geometry done", "taps in", "taps consumed", "stores acked"]
order = [0, 1, 2, 7, 3, 4, 5, 6]
d = np.diff(st[:, order], axis=1) * tick_us    # per-wave stage durations (clock offsets between XCDs cancel)
tot = (st[:, 6] - st[:, 0]) * tick_us
print("%d waves with stamps; per-wave start -> end: median %.2f us, p10 %.2f, p90 %.2f" % (len(st), np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
for r in (0, 1):
    print("role %d: per-wave stage durations (us)   median    p10    p90" % r)
    for k in range(7):
        x = d[role == r, k]
        print("  %-18s -> %-18s %6.2f %6.2f %6.2f" % (names[k], names[k + 1], np.median(x), np.percentile(x, 10), np.percentile(x, 90)))

# dispatch ramp: workgroups go round-robin to the 8 XCDs by linear id; stamps of one XCD share a clock
full = buf[:n].reshape(-1, 4, 8).astype(np.int64)       # [workgroup][wave][8]
nwg = full.shape[0]
print("\nper XCD: wave start / end relative to the XCD's first wave start (us)")
print("xcd   waves   start p50   start p90   start max     end p50     end max")
for x in range(8):
    wg = full[x::8].reshape(-1, 8)
    wg = wg[wg[:, 0] > 0]
    if not len(wg):
        continue
    b = wg[:, 0].min()
    s_rel, e_rel = (wg[:, 0] - b) * tick_us, (wg[:, 6] - b) * tick_us
    print("%3d %7d %11.2f %11.2f %11.2f %11.2f %11.2f" % (x, len(wg), np.median(s_rel), np.percentile(s_rel, 90), s_rel.max(), np.median(e_rel), e_rel.max()))

# where do the slowest waves spend their time?
slow = tot >= np.percentile(tot, 90)
fast = tot <= np.percentile(tot, 10)
print("\nmean stage durations (us): slowest 10 %% of the waves (total %.2f) | fastest 10 %% (total %.2f)" % (tot[slow].mean(), tot[fast].mean()))
for k in range(7):
    print("  %-20s -> %-20s %6.2f | %6.2f" % (names[k], names[k + 1], d[slow, k].mean(), d[fast, k].mean()))
