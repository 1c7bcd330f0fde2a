"""Multi-GPU host logic on CPU (world_size 2, gloo): host-frame sharding, accumulator packing, one all-reduce of the
packed buffer, then the product's own host-side stitch + solve (host-only sdvgn_ef handle) -- compared with the
single-process oracle solve of the full window."""
import copy
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_hosts():
    from sdv_loam_amd.parallel import shard_hosts
    assert shard_hosts(8, 1) == [(0, 8)]
    assert shard_hosts(8, 2) == [(0, 4), (4, 8)]
    assert shard_hosts(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_hosts(7, 4) == [(0, 2), (2, 4), (4, 6), (6, 7)]
    for nF in range(1, 9):
        for w in range(1, 9):
            r = shard_hosts(nF, w)
            assert r[0][0] == 0 and r[-1][1] == nF and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    import torch
    import torch.distributed as dist
    from oracle.backend import OracleEF
    from sdv_loam_amd import api, parallel, synthetic as syn
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    W = syn.make_window(w=320, h=160, nF=4, pts_per_kf=120, seed=7, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))
    lo, hi = parallel.shard_hosts(W.nF, world)[rank]
    # this rank's shard: only residuals hosted in [lo, hi) exist
    Ws = copy.copy(W)
    keep = (W.host[W.r_point] >= lo) & (W.host[W.r_point] < hi)
    for name in ("r_point", "r_target", "r_matcher", "r_state", "r_hasMatcher", "r_isLinearized", "r_isActive"):
        setattr(Ws, name, getattr(W, name)[keep])
    Ws.nR = int(keep.sum())
    O = OracleEF(Ws.w, Ws.h).load(Ws)
    O.linearizeAll(); O.applyRes(); O.solveSystemF(0, 0.1)
    buf = parallel.pack_accumulators(W.nF, O.top_acc().astype(np.float64), *[a.astype(np.float64) for a in O.sc_acc()], O.resInA())
    t = torch.from_numpy(buf)
    dist.all_reduce(t)                                   # the one collective of a GN iteration
    # product host logic on the reduced buffer (host-only handle: no GPU needed)
    L = api.load_library()
    h = C.c_void_p()
    assert L.sdvgn_ef_create(C.byref(h), -1, W.w, W.h, W.nP, None) == 0
    c = np.ascontiguousarray
    assert L.sdvgn_ef_set_calib(h, c(W.value_scaled, np.float64), c(W.value_minus_value_zero, np.float64)) == 0
    assert L.sdvgn_ef_set_frames(h, W.nF, c(W.evalPT, np.float64).reshape(-1), c(W.state, np.float64).reshape(-1), c(W.state_zero, np.float64).reshape(-1),
                                 c(W.frameID, np.int32), c(W.ab_exposure, np.float32), c(W.frameEnergyTH, np.float32)) == 0
    assert L.sdvgn_ef_set_marg_prior(h, c(W.HM, np.float64).reshape(-1), c(W.bM, np.float64)) == 0
    assert L.sdvgn_ef_set_adjoints(h) == 0 and L.sdvgn_ef_set_precalc(h) == 0
    assert L.sdvgn_ef_accumulator_count(h) == parallel.acc_count(W.nF) == t.numel()
    x = np.zeros(4 + 6 * W.nF)
    assert L.sdvgn_ef_stitch_solve_host(h, t.numpy(), 0, 0.1, x.ctypes.data_as(C.c_void_p)) == 0
    # a device-only entry point must refuse to run on a host-only handle (no silent CPU fallback)
    assert L.sdvgn_ef_apply_res(h) < 0
    L.sdvgn_ef_destroy(h)
    if rank == 0:
        Of = OracleEF(W.w, W.h).load(W)
        Of.linearizeAll(); Of.applyRes(); Of.solveSystemF(0, 0.1)
        q.put((x, Of.system()["x"], int(t.numpy()[-1]), Of.resInA()))
    else:
        q.put((x, None, None, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_accumulate_allreduce_solve_gloo(orc, sdvgn_lib):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    xs = [r[0] for r in res]
    ref = [r for r in res if r[1] is not None][0]
    assert np.array_equal(xs[0], xs[1])                                   # every rank solves the same system
    assert np.linalg.norm(xs[0] - ref[1]) / np.linalg.norm(ref[1]) < 1e-6  # == single-process solve of the full window
    assert ref[2] == ref[3]


# ---- coarse tracker, hypothesis-parallel (SURVEY 8e tracker row): host logic with the CPU oracle as the evaluator -----------------------------
def _hyp_problem():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from common import load_problem, small_problem, start_pose
    P = small_problem(seed=3, n=500, w=256, h=192, levels=3, noise=1.0)
    O = load_problem(oracle.OracleTracker(P.w, P.h, P.levels), P)
    # 7 tries: the true start last but one, the others displaced more and more (like the rotation fan of trackNewCoarse)
    poses = np.stack([start_pose(oracle, P, 100 + i, sigma_t=0.02 * (6 - i) + 0.01, sigma_r=0.004 * (6 - i) + 0.001) for i in range(7)])

    def evaluate(ps, affs, coarsest):
        out = [O.trackNewestCoarse(p, tuple(a), coarsest) for p, a in zip(ps, affs)]
        return (np.array([o[0] for o in out]), np.stack([o[1] for o in out]), np.stack([o[2] for o in out]),
                np.stack([o[3] for o in out]), np.stack([o[4] for o in out]))
    return P, O, poses, evaluate


def _hyp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from sdv_loam_amd import parallel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    P, O, poses, evaluate = _hyp_problem()
    sel, table = parallel.track_hypotheses(evaluate, poses, (0.0, 0.0), P.levels - 1, rank=rank, world=world)
    q.put((rank, sel, table))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tracker_hypothesis_split_gloo(orc, sdvgn_lib):
    import torch.multiprocessing as mp
    from sdv_loam_amd import parallel
    assert parallel.hypothesis_slice(7, 0, 2) == [0, 2, 4, 6] and parallel.hypothesis_slice(7, 1, 2) == [1, 3, 5]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 40)
    procs = [ctx.Process(target=_hyp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, t0), (_, s1, t1) = res
    assert np.array_equal(t0, t1, equal_nan=True)                       # one collective: every rank holds the full table
    assert s0["good"] and s0["index"] == s1["index"] and np.array_equal(s0["pose"], s1["pose"])
    # single process, all tries: same table, same winner
    P, O, poses, evaluate = _hyp_problem()
    s, t = parallel.track_hypotheses(evaluate, poses, (0.0, 0.0), P.levels - 1)
    assert np.array_equal(t, t0, equal_nan=True) and s["index"] == s0["index"]
    # the reference's sequential loop (later tries receive achievedRes as minResForAbort, FullSystem.cpp:412-463) picks the same try
    achieved, good, win = np.full(5, np.nan), False, -1
    for i in range(len(poses)):
        ok, p, a, lr, fl, _ = O.trackNewestCoarse(poses[i], (0.0, 0.0), P.levels - 1, min_res=achieved)
        if ok and np.isfinite(np.float32(lr[0])) and not (lr[0] >= achieved[0]):
            good, win = True, i
        if good:
            for l in range(5):
                if not np.isfinite(np.float32(achieved[l])) or achieved[l] > lr[l]:
                    achieved[l] = lr[l]
    assert good and win == s0["index"]
    assert np.allclose(achieved[:P.levels], s0["achieved_res"][:P.levels], rtol=1e-12)


def test_select_hypothesis_aborted_try_with_smallest_level0(sdvgn_lib):
    """A try the reference would have cut on a coarse level (residual > 1.5 x the achieved one, CoarseTracker.cpp:808-810) returns with its
    finer-level residuals still NaN (:674): even if its complete run ends with the smallest level-0 residual it must neither win nor
    lower the level-0 bar for a later, legitimate try."""
    from sdv_loam_amd import parallel
    nan = np.nan
    T = np.zeros((3, parallel.HYP_COLS))
    T[:, 0] = 1
    T[0, 1:6] = [1.0, 2.0, 3.0, nan, nan]
    T[1, 1:6] = [0.5, 2.0, 5.0, nan, nan]          # level 2: 5.0 > 1.5 * 3.0 -> cut there; 0.5 and 2.0 never exist in the reference
    T[2, 1:6] = [0.8, 1.9, 2.9, nan, nan]
    T[:, 6:13] = np.arange(21).reshape(3, 7)
    s = parallel.select_hypothesis(T, coarsest=2)
    assert s["good"] and s["index"] == 2
    assert np.array_equal(s["achieved_res"][:3], [0.8, 1.9, 2.9])
    # the early-out must not fire on a residual the winning pose does not have
    s = parallel.select_hypothesis(T, last_coarse_rmse0=0.4, retrack_threshold=1.5, coarsest=2)     # bar 0.6: only the phantom 0.5 is below
    assert s["tries"] == 3 and s["index"] == 2
    # a try cut on level 1 still lowers the achieved residual of the level it finished (2)
    T2 = T.copy()
    T2[1, 1:6] = [0.5, 3.5, 2.5, nan, nan]         # level 2 fine (2.5 < 4.5), level 1: 3.5 > 3.0 -> cut after level 1
    s = parallel.select_hypothesis(T2, coarsest=2)
    assert s["index"] == 2 and np.array_equal(s["achieved_res"][:3], [0.8, 1.9, 2.5])
