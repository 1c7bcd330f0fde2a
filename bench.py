#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations/s of the SDV-LOAM hot path on MI355X (contract: see DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload tracker|tracker_batch] [--no-cpu]

One "step" = one Gauss-Newton iteration: for the tracker, one LM trial at pyramid level 0 =
fused calcRes + calcGSSSE over the 2000 reference points of BASELINE.json configs[1] (1241x376, KITTI-00 calib),
including the read-back of the 8x8 H, b and the Vec6 that the host LM logic needs before it can issue the next trial.
Inputs (pyramid, reference points) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
BYTES_PER_POINT = 64           # 16 B point record + 4 taps x 12 B {I,dx,dy}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="tracker")
    ap.add_argument("--batch", type=int, default=2048, help="problems per launch for the batched roofline run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    return ap.parse_args()


def dist_setup(n):
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
    return rank, local, world


def barrier_sync(world):
    import torch
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world):
    import torch
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def tracker_setup(device, max_batch):
    from sdv_loam_amd import api, synthetic as syn
    P = syn.make_tracker_problem(1241, 376, 4, 2000, seed=0, calib=syn.KITTI00,
                                 gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
    G = api.CoarseTracker(P.w, P.h, P.levels, max_points=4096, max_batch=max_batch, device=device)
    G.makeK(**P.calib)
    for l in range(P.levels):
        G.set_ref(l, **P.ref[l])
    G.set_ref_frame(1.0, 0.0, 0.0)
    G.set_new_image(P.image, 1.0)
    return P, G, syn


def cpu_baseline_tracker(P, syn, budget_s=10.0):
    """The oracle (CPU restatement, 1 thread -- reference default multiThreading=false) on the same LM-trial workload."""
    import oracle
    O = oracle.OracleTracker(P.w, P.h, P.levels)
    O.makeK(**P.calib)
    for l in range(P.levels):
        O.set_ref(l, **P.ref[l])
    O.set_ref_frame(1.0, 0.0, 0.0)
    O.set_new_image(P.image, 1.0)
    start = oracle.se3_mul(oracle.se3_exp(syn.perturbation(0)), P.gt_pose)
    for _ in range(20):
        O.calcRes(0, start, 0.02, 2.0, 20.0)
        O.calcGS(0, 0.02, 2.0)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 200000:
        for _ in range(100):
            O.calcRes(0, start, 0.02, 2.0, 20.0)
            O.calcGS(0, 0.02, 2.0)
        n += 100
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="GN iters/s", cores=1, kind="port",
                sample="%d LM trials (calcRes+calcGSSSE, level 0, 2000 pts, 1241x376) in %.1f s on 1 host thread; host has %d logical CPUs"
                       % (n, dt, os.cpu_count()))


def main():
    args = parse()
    import torch
    rank, local, world = dist_setup(args.gpus)
    import oracle  # only for start-pose maths of the synthetic workload and the cpu_baseline leg
    P, G, syn = tracker_setup(local, max(args.batch, 64))
    ext = torch.cuda.ExternalStream(G.stream(), device=torch.device("cuda", local))
    start = oracle.se3_mul(oracle.se3_exp(syn.perturbation(rank)), P.gt_pose)
    K, W = args.steps, args.warmup

    # ---- timed region: K sequential LM trials (kernel + finalize + 640-B read-back + sync each) --------------
    for _ in range(W):
        G.resAndGS(0, start, 0.02, 2.0, 20.0)
    barrier_sync(world)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(ext)
    for _ in range(K):
        G.resAndGS(0, start, 0.02, 2.0, 20.0)
    e1.record(ext)
    barrier_sync(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world)
    ms_per_step = 1e3 * dt / K
    value = world * K / dt

    # ---- roofline of the dominant kernel (k_res_gs): batched launch, HIP events on the tracker's stream ---------
    B = args.batch
    poses = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(1000 + i)), P.gt_pose) for i in range(B)])
    affs = np.tile([0.02, 2.0], (B, 1))
    for _ in range(3):
        G.resAndGSBatch(0, poses, affs, 20.0)
    torch.cuda.synchronize()
    reps = 20
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(ext)
        G.resAndGSBatch(0, poses, affs, 20.0)
        b.record(ext)
    torch.cuda.synchronize()
    ms = np.median([a.elapsed_time(b) for a, b in evs])
    alg_bytes = B * P.ref[0]["u"].size * BYTES_PER_POINT
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    roof = dict(bound="hbm", kernel="k_res_gs(+k_finalize)", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                frac=achieved / HBM_PEAK_GBS, traffic=None,
                note="batched launch: %d LM trials x 2000 pts x 64 B algorithmic = %.1f MB per launch, %.3f ms per launch "
                     "(events bracket k_res_gs + k_finalize + the params upload)" % (B, alg_bytes / 1e6, ms))
    batched_its = B / (ms * 1e-3)

    out = {
        "metric": "Gauss-Newton iters/sec (KITTI res, 2000 pts tracker LM trial)",
        "value": value, "unit": "GN iters/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic 1241x376 4-level pyramid, 1 ref + 1 target frame, 2000 points, "
                               "coarse tracker LM trial at level 0 (fused calcRes+calcGSSSE + read-back)",
                   "parallelism": "replicas" if world > 1 else "single"},
        "roofline": roof,
        "batched_gn_iters_per_s": batched_its,
    }
    if rank == 0 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_tracker(P, syn)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
