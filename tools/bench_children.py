#!/usr/bin/env python3
"""The processes bench.py (and the counter / trace tools) run UNDER rocprofv3: each does one thing and nothing else, so that a kernel trace or a counter pass holds
only that.  Moved out of bench.py in round 6 (they were hidden --*-child flags there).

    python tools/bench_children.py trace            the headline protocol: 8 fresh windows x optimize(6)      (roofline.in_loop_trace, tools/gap_report.py)
    python tools/bench_children.py lock B           B windows, three sdvgn_ef_optimize_lockstep calls           (batched_windows.roofline)
    python tools/bench_children.py pmc              k_ef_linearize 20 times back to back (SDVGN_PMC_LOOP=1: ten loop bodies instead)      (roofline.traffic)
    python tools/bench_children.py pmc-tracker N    the N-problem batched launch of the fused tracker kernel, 6 times
SDVGN_BENCH_ARITH selects the arithmetic mode of the linearise (0 exact, 1 tolerance)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _head_kw():
    from bench import HEAD_KW
    return HEAD_KW


def pmc_child():
    """Body of the profiled child process: the window is loaded and k_ef_linearize is launched 20 times (SDVGN_PMC_LOOP=1: ten bodies of
    the optimize loop instead, so that every kernel of the loop appears in the counters)."""
    import torch  # noqa: F401
    from bench import backend_setup
    W, G = backend_setup(0)
    G.set_arith(int(os.environ.get("SDVGN_BENCH_ARITH", "0")))
    if os.environ.get("SDVGN_PMC_LOOP"):
        # (--pmc runs the kernels of all streams one at a time: the side-stream look-ahead of the rejected case cannot overlap anything there
        # and would only add its bounded wait to every body -- the loop runs without it)
        G.optimize(10, fixed_its=True, want_trace=False, no_spec_solve=True)
    else:
        G.launch_linearize_only(20)
    torch.cuda.synchronize()


def pmc_child_tracker(batch):
    """Body of the profiled child process: the 64-problem batched launch of the fused tracker kernel, 6 times."""
    import torch  # noqa: F401
    import oracle
    from sdv_loam_amd import api
    from tools.bench_rows import tracker_problem, load_tracker, distinct_batch
    P = tracker_problem()
    G = load_tracker(api, P, 0, max(batch, 64))
    Gs, launch = distinct_batch(api, oracle, P, G, 0, batch, records=os.environ.get("SDVGN_BENCH_RECORDS") == "1")
    for _ in range(6):
        launch()
    torch.cuda.synchronize()


def trace_child():
    """Body of the kernel-trace child: the headline protocol (fresh perturbed windows, optimize(6) each) on 8 windows, nothing else."""
    import torch  # noqa: F401
    from sdv_loam_amd import backend_api, synthetic as syn
    Wh = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **_head_kw())
    rs = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh) for _ in range(8)]
    for r in rs:
        r.set_arith(int(os.environ.get("SDVGN_BENCH_ARITH", "0")))
    rs[0].optimize(6, fixed_its=True, want_trace=False)
    rs[0].load(Wh)
    for r in rs:
        r.optimize(6, fixed_its=True, want_trace=False)
    torch.cuda.synchronize()


def lock_child(B):
    """Body of the profiled child for the batched launch: B windows of the named size, one warm-up sdvgn_ef_optimize_lockstep call, then the two
    that count (fresh windows each time)."""
    import torch  # noqa: F401
    from sdv_loam_amd import backend_api, synthetic as syn
    Wh = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **_head_kw())
    hs = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh) for _ in range(B)]
    for _ in range(3):
        for h_ in hs:
            h_.load(Wh)
        torch.cuda.synchronize()
        backend_api.optimize_lockstep(hs, 6, fixed_its=True, want_trace=False)
    torch.cuda.synchronize()


def main(argv):
    if len(argv) < 2:
        sys.exit(__doc__)
    mode = argv[1]
    if mode == "trace":
        trace_child()
    elif mode == "lock":
        lock_child(int(argv[2]))
    elif mode == "pmc":
        pmc_child()
    elif mode == "pmc-tracker":
        pmc_child_tracker(int(argv[2]))
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main(sys.argv)
