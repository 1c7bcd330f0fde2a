#!/usr/bin/env python3
"""bench.py -- Gauss-Newton iterations/s of the SDV-LOAM hot path on MI355X (contract: DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu] [--quick]

Headline workload (BASELINE.json metric "Gauss-Newton iters/sec (KITTI res, 8 KF x 2000 pts)" = configs[2]):
one step = one body of the FullSystem::optimize loop (FullSystemOptimize.cpp:395-458) on a synthetic KITTI-00
1241x376 window of 8 key-frames x 2000 points = 112 000 residuals:
    backupState -> solveSystemF (accumulate A/L/SC, stitch, (4+6*8)^2 LDLT, resubstitute) -> doStepFromBackup
    -> linearizeAll -> accept (applyRes) | reject (loadSateBackup + re-linearise)
Protocol: the K steps are ceil(K/6) FullSystem::optimize calls of 6 loop bodies (the reference's iteration count per key-frame), each
on its own freshly loaded, perturbed window (mixed accepted / rejected steps) -- the protocol the cpu_baseline leg runs on the host.
All windows (images, points, residual tables) are resident in HBM before the one timed region starts.
Extra fields report the coarse tracker (configs[1]) and the roofline of the dominant kernel.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tools.bench_rows import (HBM_PEAK_GBS, TRACKER_BYTES_PER_POINT, event_ms, event_avg_ms, tracker_problem, load_tracker, distinct_batch,  # noqa: E402
                              tracker_extras, cfg5_extras, reproject_extras, trace_extras, immature_extras, marginalize_extras)
LINEARIZE_BYTES_PER_RES = 584    # 76 point + 16 matcher + 384 gathers + 96 J out + 12 state
FUSED_APPLY_BYTES_PER_RES = 30   # what the loop's launches additionally write since round 6 (applyRes fused into the linearise: flags 1 + state 1 + energy 4 + JpJdF 24)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="LM trials per launch for the batched tracker roofline run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--quick", action="store_true", help="skip the tracker extras and the PMC traffic passes")
    ap.add_argument("--repeats", type=int, default=7, help="runs of the --steps/--warmup protocol; `value` is the median run")
    ap.add_argument("--worker", action="store_true", help="be the measuring process itself (a bare `python bench.py` starts one and passes its line and exit code on)")
    return ap.parse_args()


def dist_setup():
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SDVGN_BENCH_SHARE_GPU=1 (test rigs only): all ranks use cuda:0 and talk through gloo -- lets the N > 1 code path of this file
    # run on a one-GPU box (two ranks cannot form an RCCL clique on one device).  The driver's runs use nccl = RCCL.
    share = os.environ.get("SDVGN_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
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
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------------------
def backend_setup(device, **window_kw):
    from sdv_loam_amd import backend_api, synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **window_kw)
    G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP, device=device).load(W)
    return W, G


def cpu_baseline_backend(W, budget_s=12.0):
    """The oracle (CPU restatement, 1 thread: reference default multiThreading=false, settings.cpp:164) running the
    same optimize-loop bodies on the same window; additionally with 6 worker threads (the reference's NUM_THREADS = 6
    IndexThreadReduce width, its multiThreading=true configuration) as `threads6`."""
    from oracle.backend import OracleEF

    def run(threads, budget):
        O = OracleEF(W.w, W.h).load(W)
        O.set_threads(threads)
        O.optimize(2)                      # warm-up
        # time only the optimize calls (loading is setup)
        its, tt, per_body = 0, 0.0, []
        while tt < budget:
            O.load(W)
            O.set_threads(threads)
            t1 = time.perf_counter()
            tr = O.optimize(6, fixed_its=True)      # exactly 6 bodies per call, like the device's fresh-window protocol
            dt1 = time.perf_counter() - t1
            tt += dt1
            its += len(tr)
            per_body.append(1e3 * dt1 / max(len(tr), 1))
        return its, tt, per_body

    its, tt, pb = run(1, budget_s * 0.4)
    its6, tt6, _ = run(6, budget_s * 0.2)
    port = dict(value=its / tt, unit="GN iters/s", cores=1, kind="port",
                ms_per_body=dict(median=float(np.median(pb)), p10=float(np.percentile(pb, 10)), p90=float(np.percentile(pb, 90)), calls=len(pb)),
                sample="%d optimize-loop bodies (8 KF x 2000 pts, 112000 residuals; the headline's window and protocol: fresh window, "
                       "optimize(6) timed, load untimed) in %.1f s on 1 host thread; host has %d logical CPUs" % (its, tt, os.cpu_count()),
                threads6=dict(value=its6 / tt6, unit="GN iters/s", cores=6,
                              sample="%d loop bodies in %.1f s with 6 OpenMP workers on linearizeAll / accumulate / resubstitute "
                                     "(reference: multiThreading=true, NUM_THREADS=6)" % (its6, tt6)))
    ref = cpu_baseline_reference(W, budget_s * 0.4)
    if ref is None:
        return port
    # Two CPU lines exist: the reference's own translation units (kind "reference": compiled unmodified, but against the builder-written Eigen
    # STAND-IN of oracle/ref_shim -- eager evaluation, no packet paths -- i.e. a lower bound of what a real Eigen build reaches) and the oracle
    # port (kind "port": the same algorithm in plain C++).  The FASTER of the two is the cpu_baseline of the line (a slow
    # stand-in must not inflate a speed-up); the other one is nested under its kind, and the 6-thread port figure under threads6.
    ref["label"] = "reference translation units + Eigen stand-in (oracle/ref_shim), NOT a real Eigen build: a lower bound"
    best = dict(port if port["value"] >= ref["value"] else ref)
    best["port"] = port
    best["reference"] = ref
    best["chosen"] = "the faster of `port` and `reference` on this host"
    return best


def cpu_baseline_reference(W, budget_s):
    """The REFERENCE'S OWN FullSystem::optimize (oracle/_ref/libref.so: its translation units compiled unmodified with its own flags,
    -O3 / SSE2, against the Eigen stand-in of oracle/ref_shim -- see oracle/README.md) on the same windows, 1 thread (its default).
    Each timed call is optimize(6) with setting_minOptIterations = 6, i.e. exactly 6 loop bodies plus the function's fixed part (initial
    linearizeAll + applyRes; tail: setEvalPT, adjoints, precalc, linearizeAll(true)); the fixed part is measured with optimize(0) on
    fresh windows and subtracted, so that the figure is loop bodies per second like the headline.  None if the library is absent."""
    try:
        from oracle.backend import RefEF
        R = RefEF(W.w, W.h).load(W)
    except Exception:  # noqa: BLE001
        return None
    R.compute_nullspaces()
    R.optimize_full(1, min_its=1)                 # warm-up
    fixed = []
    for _ in range(2):
        R = RefEF(W.w, W.h).load(W); R.compute_nullspaces()
        R.optimize_full(0, min_its=0)
        fixed.append(R.last_seconds)
    F = float(np.median(fixed))
    tt, bodies, per_body = 0.0, 0, []
    while tt < budget_s:
        R = RefEF(W.w, W.h).load(W); R.compute_nullspaces()
        _, steps, _, _ = R.optimize_full(6, min_its=6)
        assert len(steps) == 6
        tt += R.last_seconds
        bodies += len(steps)
        per_body.append(1e3 * (R.last_seconds - F) / 6)
    calls = len(per_body)
    return dict(value=bodies / (tt - calls * F), unit="GN iters/s", cores=1, kind="reference",
                ms_per_body=dict(median=float(np.median(per_body)), p10=float(np.percentile(per_body, 10)), p90=float(np.percentile(per_body, 90)), calls=calls),
                fixed_part_ms=1e3 * F,
                sample="%d loop bodies of the reference's own FullSystem::optimize (8 KF x 2000 pts, 112000 residuals, the headline's window; "
                       "%d calls of optimize(6) on fresh windows, %.1f s, the call's fixed part of %.0f ms measured with optimize(0) and "
                       "subtracted) on 1 host thread; reference translation units compiled unmodified (-O3, SSE2) against the Eigen stand-in "
                       "of oracle/ref_shim; host has %d logical CPUs" % (bodies, calls, tt, 1e3 * F, os.cpu_count()))


# ------------------------------------------------------------------------------------------------------------
def measure_inloop_kernel(kernel="k_ef_linearize", timeout=240, arith=0, child=("trace",), keep=(8, 9), summary="inloop_trace_summary_arith%d.txt",
                          what="the headline protocol alone: 8 fresh windows x optimize(6); the untimed warm-up call and the window loads before it are not in the table"):
    """Duration of every launch of `kernel` inside the optimize loops of the headline protocol, from a rocprofv3 --kernel-trace of a child
    process that runs nothing but that protocol (HIP event pairs around single launches inside a loop read several us too long)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="sdvgn_trace_", dir="/tmp")
    try:
        subprocess.run([exe, "--kernel-trace", "-d", d, "-o", "tr", "--", sys.executable, os.path.join(ROOT, "tools", "bench_children.py")] + list(child),
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", SDVGN_BENCH_ARITH=str(arith)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=timeout, check=True)
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        con = sqlite3.connect(dbs[0])
        allk = con.execute("select name, start, duration from kernels order by start").fetchall()
        lin = [(st, dur) for (nm, st, dur) in allk if kernel in nm]
        if not lin:
            return None
        # the child runs one untimed warm-up call and then the 8 calls that count (trace_child): the statistics cover the launches of those 8
        # (the process's very first launches include code-object loading -- 0.1 ms once in a while -- and are warm-up in the bench proper too)
        timed = lin[-(len(lin) * keep[0] // keep[1]):] if len(lin) >= keep[1] else lin
        t_cut = timed[0][0]
        du = np.array([dur for (_, dur) in timed], np.float64) / 1e6
        try:   # the per-kernel summary of this trace as a file (gpurun_out/, copied to profiles/ by hand): what roofline.achieved is computed from
            agg = {}
            for nm, st, dur in allk:
                if st >= t_cut:
                    agg.setdefault(nm, []).append(dur)
            rows = sorted(((nm, len(v), float(np.mean(v)), float(np.sum(v)), float(np.min(v)), float(np.max(v))) for nm, v in agg.items()), key=lambda r: -r[3])
            tot = sum(r[3] for r in rows)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", summary % arith if "%d" in summary else summary), "w") as fsum:
                fsum.write("# rocprofv3 --kernel-trace -- python tools/bench_children.py %s   (SDVGN_BENCH_ARITH=%d; %s)\n" % (" ".join(child), arith, what))
                fsum.write("%-100s %8s %12s %10s %10s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
                for r in rows[:30]:
                    fsum.write("%-100s %8d %12.1f %10.3f %10.3f %10.3f %6.2f\n" % (r[0][:100], r[1], r[3] / 1e3, r[2] / 1e3, r[4] / 1e3, r[5] / 1e3, 100 * r[3] / tot))
        except Exception:  # noqa: BLE001
            pass
        return dict(mean_ms=float(du.mean()), median_ms=float(np.median(du)), p90_ms=float(np.percentile(du, 90)), launches=int(len(du)),
                    source="rocprofv3 --kernel-trace of a child process: " + what)
    except Exception as ex:  # noqa: BLE001
        return dict(error=repr(ex))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_traffic(kernel="k_ef_linearize", timeout=240, child=("pmc",), env_extra=None):
    """HBM-side bytes per launch of `kernel` from rocprofv3 PMC counters, two separate passes (FETCH_SIZE, WRITE_SIZE; the TCC
    block cannot hold both), corrected as MI355X_MICROARCH.md prescribes and as profiles/r01_counter_calibration.txt confirms
    for this project's access patterns: FETCH_SIZE x2 (128-B requests are tallied at 64 B), WRITE_SIZE x1, KiB -> bytes.
    Returns (bytes_per_launch, detail) or (None, reason)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="sdvgn_pmc_", dir="/tmp")
            env = dict(os.environ, TMPDIR="/tmp", **(env_extra or {}))
            subprocess.run([exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "tools", "bench_children.py")] + list(child),
                           cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            con = sqlite3.connect(dbs[0])
            row = con.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?",
                              (ctr, "%" + kernel + "%")).fetchone()
            vals[ctr] = (float(row[0]), int(row[1]))
            shutil.rmtree(d, ignore_errors=True)
        fetch = vals["FETCH_SIZE"][0] * 1024.0 * 2.0
        write = vals["WRITE_SIZE"][0] * 1024.0
        return fetch + write, "FETCH_SIZE %.0f KiB x2 + WRITE_SIZE %.0f KiB x1 per launch (avg of %d launches, separate --pmc passes)" % (
            vals["FETCH_SIZE"][0], vals["WRITE_SIZE"][0], vals["FETCH_SIZE"][1])
    except Exception as ex:  # noqa: BLE001
        return None, "PMC pass failed: %r" % (ex,)


HEAD_KW = dict(state_sigma=3e-3, idepth_sigma=0.02)   # headline window: perturbed so that optimize(6) mixes accepted and rejected steps


def run_protocol(runners, bodies, world, reload_with=None, warm=None, **opt_kw):
    """The fresh-window protocol: one FullSystem::optimize call (its initial linearizeAll + applyRes, then `nb` loop bodies) per resident
    window, all inside ONE timed region (barrier + synchronize on both sides, max over ranks).  Returns (seconds, traces).
    warm = (handle, bodies): the untimed warm-up bodies run on a window of their own, AFTER the timed windows were (re)loaded, i.e. right
    before the timed region -- a reload is tens of milliseconds of PCIe copies with idle compute units."""
    if reload_with is not None:
        for r in runners:
            (r.reload if hasattr(r, "reload") else r.load)(reload_with)
    if warm is not None:
        wr, wb = warm
        done = 0
        while done < wb:
            nb = min(6, wb - done)
            wr.optimize(nb, fixed_its=True, want_trace=False)
            done += nb
    barrier_sync(world)
    t0 = time.perf_counter()
    traces = []
    marks = []
    for i, nb in enumerate(bodies):
        traces.append(runners[i % len(runners)].optimize(nb, fixed_its=True, **opt_kw))
        marks.append(time.perf_counter() - t0)
    barrier_sync(world)
    dt = time.perf_counter() - t0
    if os.environ.get("SDVGN_BENCH_DEBUG"):
        sys.stderr.write("[bench] region %.1f us; bodies %s; calls returned at %s us\n" % (1e6 * dt, list(bodies)[:8], np.round(1e6 * np.array(marks[:8]), 1)))
    return max_over_ranks(dt, world), traces


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` from a bare shell: launch the N ranks ourselves (one process per GPU over RCCL), the way the driver
        # does with torch.distributed.run; rank 0 of the children prints the JSON line, which passes through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if "WORLD_SIZE" not in os.environ and not args.worker and os.environ.get("SDVGN_BENCH_NO_WRAPPER") != "1":
        # `python bench.py` from a bare shell, one GPU: the measurement runs in ONE worker process whose JSON line passes through.  The worker's
        # exit code is this process's exit code: a GPU memory fault aborts the process that owns the queue, and that is a failure of the run
        # (round 3 restarted the worker and accepted the line "whatever the exit code" -- which hid such an abort).  If the
        # worker got as far as printing its line before it died, the line is passed on WITH the exit code in it, and the run still fails.
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--worker"] + sys.argv[1:]
        p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if lines:
            line = lines[-1]
            if p.returncode:
                try:
                    d = json.loads(line)
                    d["bench_worker_exit_code"] = p.returncode
                    line = json.dumps(d)
                except Exception:  # noqa: BLE001
                    pass
            print(line, flush=True)
        if p.returncode or not lines:
            sys.stderr.write("bench.py: the worker ended with rc %d and %d JSON line(s)\n" % (p.returncode, len(lines)))
            sys.exit(p.returncode or 1)
        return
    import torch
    from sdv_loam_amd import backend_api, synthetic as syn
    rank, local, world = dist_setup()
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used" % (args.gpus, world), file=sys.stderr)
    K, Wm = args.steps, args.warmup
    W, G = backend_setup(local)                       # the unperturbed window: kernel timings, per-row extras, the soak
    Wh = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **HEAD_KW)
    ext = torch.cuda.ExternalStream(G.stream(), device=torch.device("cuda", local))
    n_calls = (K + 5) // 6
    bodies = [6] * (n_calls - 1) + [K - 6 * (n_calls - 1)]
    parallelism, scaling = "single GPU", "strong"
    runners = None
    rccl_ranks = collectives_per_body = None
    if world > 1:
        # configs[3]: every window sharded by host key-frame across the ranks, ONE all-reduce of the packed message per GN iteration (strong
        # scaling: total work fixed).  `--gpus N` was asked for: if the sharded path cannot start this run FAILS (non-zero exit code) --
        # it does not quietly measure N independent replicas instead.
        from sdv_loam_amd.parallel import ShardedEnergyFunctional, shard_hosts
        import torch.distributed as dist
        try:
            runners = [ShardedEnergyFunctional(Wh, rank, world, local) for _ in range(min(n_calls, 16))]
            c0 = runners[0].collective_count()
            tr0 = runners[0].optimize(2, want_trace=True, fixed_its=True)       # the first collectives happen here
            runners[0].reload(Wh)
        except Exception as ex:  # noqa: BLE001
            sys.stderr.write("bench.py: rank %d: the sharded path (configs[3]) could not start with --gpus %d: %r\n" % (rank, world, ex))
            sys.stderr.flush()
            os._exit(3)
        via = "RCCL, issued by the library" if getattr(runners[0], "direct_rccl", False) else ("torch.distributed/%s callback" % dist.get_backend())
        rccl_ranks = int(runners[0].ef.L.sdvgn_ef_rccl_ranks(runners[0].ef.h_)) if getattr(runners[0], "direct_rccl", False) else 0
        if getattr(runners[0], "one_collective", False):
            parallelism = "host-keyframe shards %s + ONE all-reduce per loop body (282 kB fp64: packed accumulators | 4 statistics | quantile candidates; trial applied and accumulated speculatively) via %s" % (
                shard_hosts(Wh.nF, world), via)
        else:
            parallelism = "host-keyframe shards %s + 2 all-reduces per iteration (154 kB packed accumulators; 128 kB statistics + threshold candidates) via %s" % (
                shard_hosts(Wh.nF, world), via)
    if runners is None:
        # 80 MB per window: every window of the timed region is resident in HBM before it starts (288 GB would hold thousands)
        runners = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP, device=local).load(Wh) for _ in range(min(n_calls, 1024))]
    # ---- warm-up: Wm bodies on a window of its own, right before the timed region (after the timed windows were reloaded, untimed) ----
    warm_runner = type(runners[0])(Wh, rank, world, local) if hasattr(runners[0], "reload") else \
        backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP, device=local).load(Wh)
    # ---- timed region: K loop bodies = ceil(K / 6) optimize calls on fresh windows --------------------------------------------------------
    # the timed windows were loaded when their handles were created and have never been optimised: fresh by construction, no reload
    # The driver's flags time K = 20 bodies = 1.4 ms: one half-millisecond stall of the box moves such a region by a third (r04: one of 27 workers
    # read 9 951 it/s against 13.7-14.3 k).  So the SAME protocol -- W warm-up bodies, then exactly K timed bodies between barriers -- is run
    # `repeats` times on freshly reloaded windows and `value` is the MEDIAN run; every run's value is reported (value_runs).
    coll0 = sum(r.collective_count() for r in runners) if world > 1 else 0
    repeats = max(1, args.repeats)
    dts = []
    for rep in range(repeats):
        dt_i, traces = run_protocol(runners, bodies, world, reload_with=None if rep == 0 else Wh, warm=(warm_runner, Wm), want_trace=False)
        dts.append(dt_i)
        if rep == 0 and world > 1:
            # counted by the library: all-reduces issued inside the timed region / loop bodies run in it (one per body + one per optimize call)
            collectives_per_body = (sum(r.collective_count() for r in runners) - coll0) / float(K)
    dt = float(np.median(dts))
    fac = world if scaling == "weak" else 1
    value = fac * K / dt
    value_runs = dict(n=repeats, median=value, min=fac * K / max(dts), max=fac * K / min(dts), values=[fac * K / d for d in dts],
                      note="the --steps / --warmup protocol repeated on reloaded windows; `value` is the median run")
    single = world == 1
    look_ahead = None
    if single:      # the rejected case solved ahead (side stream): how many bodies of the last timed run started from such a solution
        la = [r.look_ahead() for r in runners[:n_calls]]
        look_ahead = dict(active=bool(sum(a for a, _ in la) > 0), solves_launched=int(sum(a for a, _ in la)), bodies_served=int(sum(b for _, b in la)))
    accepted_fraction = None
    if single:   # every handle remembers the accepted steps of its last call
        R = len(runners)
        last = {j % R: bodies[j] for j in range(n_calls)}
        accepted_fraction = float(sum(runners[i].accepted_steps() for i in last)) / float(sum(last.values()))
    it_us = np.concatenate([(r.ef if hasattr(r, "ef") else r).iteration_times_us() for r in runners[:n_calls]])
    iter_stats = dict(median_us=float(np.median(it_us)), p10_us=float(np.percentile(it_us, 10)), p90_us=float(np.percentile(it_us, 90)),
                      n=int(len(it_us))) if len(it_us) else None

    # ---- the same protocol with every window handed over as HOST buffers inside the timed region (8 images + points + residuals over
    # PCIe, ~80 MB per window): what a caller that keeps nothing on the device pays.  Never the headline value.
    upload_inclusive = None
    if single and not args.quick:
        nu = min(4, len(runners))
        ext.synchronize()
        t0u = time.perf_counter()
        for r in runners[:nu]:
            r.load(Wh)
            r.optimize(6, fixed_its=True, want_trace=False)
        ext.synchronize()
        dtu = time.perf_counter() - t0u
        upload_inclusive = dict(value=6 * nu / dtu, unit="GN iters/s", ms_per_window=1e3 * dtu / nu,
                                note="load() of the whole window from host buffers + optimize(6) per window, both timed (PCIe-inclusive)")

    # ---- N > 1 only: the same protocol as N independent replicas (every rank optimises its own copies of the full window, no
    # collective) -- the weak-scaling view next to the strong-scaling headline that configs[3] asks for ----
    replicas_value = None
    if world > 1 and scaling == "strong":
        reps = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP, device=local).load(Wh) for _ in range(min(n_calls, 16))]
        reps[0].optimize(min(6, max(Wm, 1)), fixed_its=True, want_trace=False)
        dt_r, _ = run_protocol(reps, bodies, world, warm=(warm_runner, Wm), reload_with=Wh, want_trace=False)
        replicas_value = world * K / dt_r
        del reps

    value_relin = value_reuse = value_nospec = soak = other = lin_inloop = None
    tol = batched = None
    if single and not args.quick:
        # ---- B independent windows in one call (sdvgn_ef_optimize_batch): the lock-step launch sequence (default, csrc/backend_lockstep.inc), the
        # same calls one after the other (sdvgn_ef_optimize per window, same handles, same fresh windows), and -- at B = 8 -- the round-3 form
        # (one host thread + stream per window, SDVGN_BATCH_THREADS=1).  B x 80 MB of window data is live at once (B = 16: 1.3 GB, beyond the
        # 256 MB Infinity Cache) ----
        batched = {}

        def time_calls(hs, fn, reps=6):
            tt = 0.0
            for r_ in range(reps + 1):
                for h_ in hs:
                    h_.load(Wh)
                torch.cuda.synchronize()
                t0b = time.perf_counter()
                fn(hs)
                torch.cuda.synchronize()
                if r_:
                    tt += time.perf_counter() - t0b
            return tt / reps

        def sequential(hs):
            for h_ in hs:
                h_.optimize(6, fixed_its=True, want_trace=False)

        for Bw in (2, 4, 8, 16):
            try:
                hs = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP, device=local).load(Wh) for _ in range(Bw)]
                os.environ.pop("SDVGN_BATCH_THREADS", None)
                t_lock = time_calls(hs, lambda hs_: backend_api.optimize_batch(hs_, 6, fixed_its=True))
                t_seq = time_calls(hs, sequential)
                row = dict(value=6 * Bw / t_lock, unit="GN iters/s (aggregate)", ms_per_batch_call=1e3 * t_lock,
                           sequential_calls_value=6 * Bw / t_seq, speedup_vs_sequential_calls=t_seq / t_lock, speedup_vs_headline=6 * Bw / t_lock / value)
                del hs
                if Bw == 8:
                    hs = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP, device=local, stream=backend_api.EnergyFunctional.STREAM_OWN).load(Wh)
                          for _ in range(Bw)]
                    os.environ["SDVGN_BATCH_THREADS"] = "1"
                    t_thr = time_calls(hs, lambda hs_: backend_api.optimize_batch(hs_, 6, fixed_its=True))
                    os.environ.pop("SDVGN_BATCH_THREADS", None)
                    row["host_thread_per_window_value"] = 6 * Bw / t_thr
                    del hs
                batched["B%d" % Bw] = row
            except Exception as ex:  # noqa: BLE001
                os.environ.pop("SDVGN_BATCH_THREADS", None)
                batched["B%d" % Bw] = dict(error=repr(ex))
        batched["note"] = ("B fresh perturbed windows per call, optimize(6) each (7 linearizeAll + 6 loop bodies per window).  value: ONE launch sequence for all "
                           "B windows (sdvgn_ef_optimize_lockstep, the default of sdvgn_ef_optimize_batch): per window bit-identical to its own call "
                           "(tests/test_backend_gpu.py::test_lockstep_*).  sequential_calls_value: the same handles and windows, sdvgn_ef_optimize one after "
                           "the other, timed the same way; speedup_vs_headline divides by the headline value instead")
        if rank == 0 and world == 1:
            # the bandwidth kernel of the batched launch at B = 16 (1.3 GB of window data, beyond the Infinity Cache): duration from a kernel trace
            # of a child that runs nothing else, HBM-side bytes from the PMC counters of the same child
            Bt = 16
            ltr = measure_inloop_kernel(kernel="k_lock_linearize", child=("lock", str(Bt)), keep=(2, 3), summary="lockstep_trace_summary_B16.txt",
                                        what="%d windows of the named size, sdvgn_ef_optimize_lockstep(6) x 3 calls, the first one untimed" % Bt)
            alg_b = Bt * Wh.nR * LINEARIZE_BYTES_PER_RES
            if ltr and "mean_ms" in ltr:
                ach = alg_b / (ltr["mean_ms"] * 1e-3) / 1e9
                tb_, how_ = measure_traffic(kernel="k_lock_linearize", child=("lock", str(Bt)))
                batched["roofline"] = dict(bound="hbm", kernel="k_lock_linearize (B = %d windows in one launch)" % Bt, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                                           frac=ach / HBM_PEAK_GBS, traffic=tb_, traffic_note=how_, traffic_over_algorithmic=(tb_ / alg_b) if tb_ else None,
                                           in_loop_trace=ltr,
                                           note="%d windows x %d residuals x %d B algorithmic = %.0f MB per launch; mean duration of the launches inside the "
                                                "lock-step calls" % (Bt, Wh.nR, LINEARIZE_BYTES_PER_RES, alg_b / 1e6))
            else:
                batched["roofline"] = dict(error=repr(ltr))
    if single:
        # ---- the same protocol in tolerance-mode arithmetic of k_ef_linearize (sdvgn_ef_set_arith(1): FMA, v_rcp_f32 / v_sqrt_f32; increments
        # within the contract's 1e-4, tests/test_backend_gpu.py::test_arith_mode_tolerance) ----
        for r in runners:
            r.set_arith(1)
        warm_runner.set_arith(1)
        dt_t, _ = run_protocol(runners, bodies, world, warm=(warm_runner, Wm), reload_with=Wh, want_trace=False)
        tol = dict(value=K / dt_t, ms_per_step=1e3 * dt_t / K)
        for r in runners:
            r.set_arith(0)
        warm_runner.set_arith(0)
    if single:
        # ---- the same protocol with the reference's literal re-linearisation after every rejected step (A/B; results identical) ----
        dt_l, _ = run_protocol(runners, bodies, world, warm=(warm_runner, Wm), reload_with=Wh, want_trace=False, relinearize_on_reject=True)
        value_relin = K / dt_l
        # ---- opt-in flag bit2: bodies that follow a rejected step re-use the stitched system (bit-identical results; extra key only) ----
        dt_u, _ = run_protocol(runners, bodies, world, warm=(warm_runner, Wm), reload_with=Wh, want_trace=False, reuse_after_reject=True)
        value_reuse = K / dt_u
        # ---- A/B of round 4's default: without the rejected case solved ahead on the side stream (flags bit4; results bit-identical) ----
        dt_ns, _ = run_protocol(runners, bodies, world, warm=(warm_runner, Wm), reload_with=Wh, want_trace=False, no_spec_solve=True)
        value_nospec = K / dt_ns
        # ---- k_ef_linearize inside the loop: a HIP event pair around every launch of the same protocol (own pass: the event packets
        # cost ~2 us per body, so this pass is not the headline) ----
        nl = min(len(runners), 8)
        run_protocol(runners[:nl], [6] * nl, world, reload_with=Wh, want_trace=False, time_linearize=True)
        lt = np.concatenate([r.linearize_times_ms() for r in runners[:nl]])
        lin_inloop = dict(mean_ms=float(lt.mean()), median_ms=float(np.median(lt)), p90_ms=float(np.percentile(lt, 90)), launches=int(len(lt)))
        # ---- one window, 60 bodies in ONE optimize call (the round-1 headline, kept as an extra: after a few accepted steps the window
        # is converged and nearly every further step is rejected) ----
        G.load(W)
        G.optimize(Wm, want_trace=False, fixed_its=True)
        G.load(W)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr_s = G.optimize(60, want_trace=True, fixed_its=True, cap=64)
        soak = dict(value=60 / (time.perf_counter() - t1), bodies=60, accepted_fraction=float(np.mean(tr_s[:, 2])),
                    note="unperturbed window, one optimize call of 60 bodies")
        # ---- the protocol on two other windows: unperturbed (1 of 6 steps accepted) and perturbed far enough that all 6 are ----
        other = {}
        for name, kw in (("unperturbed", {}), ("all_steps_accepted", dict(state_sigma=1e-2, idepth_sigma=0.05))):
            Wx = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **kw)
            nx = min(len(runners), 8)
            dtx, trx = run_protocol(runners[:nx], [6] * nx, world, warm=(warm_runner, Wm), reload_with=Wx, want_trace=True)
            other[name] = dict(value=6 * nx / dtx, accepted_fraction=float(np.mean(np.concatenate([t[:, 2] for t in trx]))), window=kw)
        G.load(W)
    del runners

    # ---- the shape the reference really runs: setting_maxFrames = 7 key-frames, ~2000 active points in the WHOLE window
    # (setting_desiredPointDensity, settings.cpp:46-47,52-53), images cropped to 1200 x 360 (calib/KITTI/00.txt) -- same protocol, with its
    # own 1-thread CPU line ----
    ref_shape = None
    if single and not args.quick:
        cal7 = dict(fx=718.856, fy=718.856, cx=607.1928 - 20.5, cy=185.2157 - 8.0)      # KITTI-00 intrinsics, principal point moved by the crop
        W7 = syn.make_window(w=1200, h=360, nF=7, pts_per_kf=286, seed=0, calib=cal7, **HEAD_KW)
        r7 = [backend_api.EnergyFunctional(W7.w, W7.h, max_points=W7.nP, device=local).load(W7) for _ in range(8)]
        r7[0].optimize(6, fixed_its=True, want_trace=False)
        dt7, tr7 = run_protocol(r7, [6] * 8, world, warm=(warm_runner, Wm), reload_with=W7, want_trace=True)
        it7 = np.concatenate([r.iteration_times_us() for r in r7])
        ref_shape = dict(workload="7 key-frames x 286 points (2002 points, %d residuals), 1200x360" % W7.nR, value=48 / dt7, unit="GN iters/s",
                         accepted_fraction=float(np.mean(np.concatenate([t[:, 2] for t in tr7]))), median_body_us=float(np.median(it7)))
        del r7
        if not args.no_cpu:
            from oracle.backend import OracleEF
            O7 = OracleEF(W7.w, W7.h)
            tt, its = 0.0, 0
            while tt < 2.0:
                O7.load(W7)
                t1 = time.perf_counter()
                its += len(O7.optimize(6, fixed_its=True))
                tt += time.perf_counter() - t1
            ref_shape["cpu_oracle_1_thread"] = its / tt

    # ---- dominant kernel: k_ef_linearize, HIP events on the library's stream -----------------------------------
    G.launch_linearize_only(5)
    torch.cuda.synchronize()
    ms_lin_b2b = event_avg_ms(torch, ext, lambda: G.launch_linearize_only(1), 50)
    ms_lin_b2b_each = event_ms(torch, ext, lambda: G.launch_linearize_only(1), 50)     # the same launches with an event pair around each one
    ev_overhead = max(0.0, ms_lin_b2b_each - ms_lin_b2b)                                 # what an event pair adds to a single launch on this box
    alg = W.nR * LINEARIZE_BYTES_PER_RES
    if lin_inloop:
        lin_inloop["event_pair_overhead_ms"] = ev_overhead
        lin_inloop["mean_ms_minus_event_overhead"] = lin_inloop["mean_ms"] - ev_overhead
    lin_trace = measure_inloop_kernel() if (rank == 0 and world == 1 and not args.quick) else None
    if lin_trace and "mean_ms" in lin_trace:
        ms_lin, how = lin_trace["mean_ms"], "rocprofv3 kernel trace of the protocol run in a child process of this bench"
    else:
        ms_lin, how = ms_lin_b2b, "one HIP event pair around 50 back-to-back launches"
    achieved = alg / (ms_lin * 1e-3) / 1e9
    roof = dict(bound="hbm", kernel="k_ef_linearize", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                traffic=None, in_loop_trace=lin_trace, back_to_back_ms=ms_lin_b2b, in_loop_event_pairs=lin_inloop,
                note="%d residuals x %d B algorithmic = %.1f MB per launch; %.4f ms = mean duration of the launches INSIDE the optimize loops of "
                     "the headline protocol (%s); the same kernel back to back, one HIP event pair around 50 launches on the library "
                     "stream: %.4f ms; a HIP event pair around every single in-loop launch reads several us too long (in_loop_event_pairs: "
                     "the pair adds %.4f ms even back to back) and is reported for completeness only" % (
                         W.nR, LINEARIZE_BYTES_PER_RES, alg / 1e6, ms_lin, how, ms_lin_b2b, ev_overhead))
    if tol is not None:
        G.set_arith(1)
        G.launch_linearize_only(3)
        tol_b2b = event_avg_ms(torch, ext, lambda: G.launch_linearize_only(1), 50)
        G.set_arith(0)
        tr_t = measure_inloop_kernel(arith=1) if (rank == 0 and world == 1 and not args.quick) else None
        ms_t = tr_t["mean_ms"] if (tr_t and "mean_ms" in tr_t) else tol_b2b
        tol["roofline"] = dict(bound="hbm", kernel="k_ef_linearize (lin_fast)", achieved=alg / (ms_t * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                               frac=alg / (ms_t * 1e-3) / 1e9 / HBM_PEAK_GBS, in_loop_trace=tr_t, back_to_back_ms=tol_b2b)
        tol["note"] = "sdvgn_ef_set_arith(1) on every window: k_ef_linearize with fused multiply-adds and 1-ulp v_rcp_f32 / v_sqrt_f32 instead of IEEE division / sqrt sequences; same protocol, same windows"
    ms_acc = event_ms(torch, ext, lambda: G.accumulate(), 50)
    # what a plain streaming copy reaches on this box (1 GiB read + 1 GiB written), for reading `frac` against the achievable rate
    try:
        src = torch.empty(1 << 28, dtype=torch.float32, device=torch.device("cuda", local))
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        roof["hbm_copy_measured_GBps"] = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del src, dst
    except Exception:  # noqa: BLE001
        roof["hbm_copy_measured_GBps"] = None

    try:     # ... and what a hand-written float4 copy KERNEL reaches (the guide's practical peak: ~6.3 TB/s): the denominator `frac_of_practical_peak` uses
        import ctypes as C_
        L_ = G.L
        L_.sdvgn_debug_copy_rate.restype = C_.c_double
        L_.sdvgn_debug_copy_rate.argtypes = [C_.c_size_t, C_.c_int]
        roof["hbm_copy_kernel_float4_GBps"] = float(L_.sdvgn_debug_copy_rate(1 << 30, 8))
        if roof["hbm_copy_kernel_float4_GBps"] > 0:
            roof["frac_of_practical_peak"] = roof["achieved"] / roof["hbm_copy_kernel_float4_GBps"]
    except Exception:  # noqa: BLE001
        roof["hbm_copy_kernel_float4_GBps"] = None
    if batched and isinstance(batched.get("roofline"), dict) and batched["roofline"].get("traffic") and roof.get("hbm_copy_measured_GBps"):
        br = batched["roofline"]
        br["traffic_rate_GBps"] = br["traffic"] / (br["in_loop_trace"]["mean_ms"] * 1e-3) / 1e9
        br["hbm_copy_measured_GBps"] = roof["hbm_copy_measured_GBps"]
        br["traffic_rate_over_measured_copy_rate"] = br["traffic_rate_GBps"] / roof["hbm_copy_measured_GBps"]
        br["note"] += ("; the counters' bytes per launch / that duration = %.0f GB/s = %.2f of what a plain 1 GiB device-to-device copy reaches on this box "
                       "(hbm_copy_measured_GBps): at B = 16 the working set is 1.3 GB, every launch streams from HBM" % (
                           br["traffic_rate_GBps"], br["traffic_rate_over_measured_copy_rate"]))
    out = {
        "metric": "Gauss-Newton iters/sec (KITTI res, 8 KF x 2000 pts)",
        "value": value, "unit": "GN iters/s", "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K,
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[2]: KITTI-00 calib 1241x376, 8-keyframe window, 2000 points/KF, 112000 residuals; "
                               "one step = one FullSystem::optimize loop body (solveSystemF + step + linearizeAll + accept/reject); "
                               "protocol: ceil(K/6) optimize calls of 6 bodies, each on its own freshly loaded window resident in HBM "
                               "(window perturbed: state_sigma 3e-3, idepth_sigma 0.02), every call incl. its initial linearizeAll + applyRes",
                   "parallelism": parallelism, "windows": int(min(n_calls, 1024 if world == 1 else 16)), "optimize_calls": int(n_calls),
                   "rccl_ranks": rccl_ranks, "collectives_per_body": collectives_per_body,
                   "collectives_note": None if world == 1 else "rccl_ranks = ncclCommCount of the communicator the library issues its all-reduces on (0: "
                                       "torch.distributed callback path); collectives_per_body = all-reduces counted by the library inside the timed "
                                       "region / K (ONE per loop body + one per optimize call: (K + calls) / K expected)"},
        "roofline": roof,
        "value_runs": value_runs,
        "look_ahead_active": None if look_ahead is None else look_ahead["active"],
        "look_ahead": look_ahead,
        "bench_worker_exit_code": 0,
        "accepted_fraction": accepted_fraction,
        "iteration_us": iter_stats,
        "kernel_ms": {"k_ef_linearize_back_to_back": ms_lin_b2b, "accumulate(fused point+top+sc, reduce)": ms_acc},
        "tolerance_arith": tol,
        "batched_windows": batched,
        "value_with_literal_relinearize_on_reject": value_relin,
        "value_with_system_reuse_after_rejected_steps": value_reuse,
        "value_without_rejected_case_solved_ahead": value_nospec,
        "one_window_soak": soak,
        "other_windows_same_protocol": other,
        "reference_shape_7kf_2000pts": ref_shape,
        "replicas_value_weak_scaling": replicas_value,
        "value_window_upload_inclusive": upload_inclusive,
    }
    if rank == 0 and world == 1 and not args.quick:
        traffic, how = measure_traffic()
        roof["traffic"] = traffic
        roof["traffic_note"] = how
        roof["counters_note"] = ("the PMC child runs the kernel alone, back to back: under --pmc the profiler serialises the streams, so the look-ahead of the "
                                 "shipping loop (side-stream solve of the rejected case) cannot run beside the linearise there and stands down; timings in this "
                                 "line (value, in_loop_trace) are of the shipping loop with the look-ahead active (look_ahead.active), counters are of the kernel")
    # the per-row extras and the CPU baseline are N = 1 material: at N > 1 the other ranks would only wait for rank 0 at the final barrier
    if rank == 0 and world == 1 and not args.quick:
        import oracle
        out["tracker"] = tracker_extras(torch, local, args.batch, oracle, not args.no_cpu)
        out["cfg5_fp16_tolerance_study"] = cfg5_extras(torch, local, oracle, not args.no_cpu)
        tb, how = measure_traffic(kernel="k_res_gs", child=("pmc-tracker", str(args.batch)))
        alg_t = args.batch * 2000 * TRACKER_BYTES_PER_POINT
        bi = out["tracker"]["batched_independent_problems"]
        bi["hbm_traffic_bytes_per_launch"] = tb
        bi["traffic_over_algorithmic"] = (tb / alg_t) if tb else None
        if tb:
            ms_sp = bi["exact"]["ms_per_launch"]
            bi["hbm_traffic_GBps"] = tb / (ms_sp * 1e-3) / 1e9
            bi["hbm_traffic_frac_of_peak"] = bi["hbm_traffic_GBps"] / HBM_PEAK_GBS
        bi["traffic_note"] = how + "; exact arithmetic; sparse 24-byte gathers pull whole 128-byte lines, so the launch is bound by HBM TRAFFIC, not by its algorithmic bytes"
        tbr, howr = measure_traffic(kernel="k_res_gs", child=("pmc-tracker", str(args.batch)), env_extra={"SDVGN_BENCH_RECORDS": "1"})
        rl = bi.get("record_layout")
        if rl is not None:
            rl["hbm_traffic_bytes_per_launch"] = tbr
            rl["traffic_over_algorithmic"] = (tbr / alg_t) if tbr else None
            if tbr:
                rl["hbm_traffic_GBps"] = tbr / (rl["ms_per_launch"] * 1e-3) / 1e9
                rl["hbm_traffic_frac_of_peak"] = rl["hbm_traffic_GBps"] / HBM_PEAK_GBS
            rl["traffic_note"] = howr
    if rank == 0 and world == 1 and not args.quick:
        out["reprojector"] = reproject_extras(W, G, local, not args.no_cpu)
        out["trace_points"] = trace_extras(W, local, not args.no_cpu)
        out["optimize_immature"] = immature_extras(W, G, not args.no_cpu)
        out["marginalize"] = marginalize_extras(torch, W, G, not args.no_cpu)
    if rank == 0 and world == 1 and not args.quick:
        # ---- a KEY-FRAME, not a loop body: the window edited in place, and the reference's own host loop on the drop-in (tools/bench_legs.py) ----
        from tools import bench_legs, exp_keyframe_update
        try:
            W9 = exp_keyframe_update.world()
            kf = bench_legs.keyframe_update_leg(8, W9)
            out["value_keyframe_update_inclusive"] = kf
            dl = bench_legs.dropin_legs(W9, want_cpu=not args.no_cpu)
            out["dropin"] = dl
            out["dropin_frame"] = bench_legs.frame_legs(want_cpu=not args.no_cpu)
            for k_ in ("dropin_optimize", "dropin_optimize_4_host_threads", "dropin_solveSystemF", "cpu_reference"):
                if isinstance(dl.get(k_), dict):
                    out[k_ + "_its_per_s"] = dl[k_]["its_per_s"]
        except Exception as ex:  # noqa: BLE001
            out["value_keyframe_update_inclusive"] = dict(error=repr(ex))
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline_backend(Wh)
    if rank == 0:
        # ONE stdout line: the contract's keys only, < 4 kB (tools/bench_line.py; round 5's 21.8 kB line came back `parsed: null` from the driver).
        # Everything else -- per-row extras, A/B legs, CPU sub-figures -- goes to bench_extras.json; a numbers-only digest goes to stderr.
        from tools import bench_line
        roof["bytes_per_launch"] = alg
        # the in-loop launches also perform applyRes (their k_ef_stats_apply pass is gone): with the bytes that adds, for comparison with earlier rounds' 0.51
        roof["frac_incl_fused_apply_bytes"] = (W.nR * (LINEARIZE_BYTES_PER_RES + FUSED_APPLY_BYTES_PER_RES)) / (ms_lin * 1e-3) / 1e9 / HBM_PEAK_GBS
        roof["short_note"] = "%d residuals x %d B / mean duration of the in-loop launches (rocprofv3 kernel trace, child run)" % (W.nR, LINEARIZE_BYTES_PER_RES) \
            if (lin_trace and "mean_ms" in lin_trace) else "%d residuals x %d B / back-to-back launch duration (HIP events on the library stream)" % (W.nR, LINEARIZE_BYTES_PER_RES)
        bench_line.write_extras(out, ROOT)
        sys.stderr.write(bench_line.digest(out) + "\n")
        sys.stderr.flush()
        print(bench_line.compact_line(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
