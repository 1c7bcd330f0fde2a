"""The ONE stdout line of bench.py: the contract's keys and nothing else, bounded at 4 kB.

Round 5's line had grown to 21.8 kB (every per-row extra nested in it) and the driver's record came back `parsed: null`.  The line is now
built here from the full result dictionary: the contract keys, a reduced `roofline` and `cpu_baseline`, and `value_runs` -- everything else
goes to `bench_extras.json` (repo root; also gpurun_out/ when that directory exists) and, as a short digest, to stderr.
tests/test_bench_line.py runs this on a canned round-5 result and asserts the size bound and the JSON round trip.
"""
import json
import math
import os

MAX_LINE = 4096
MAX_NOTE = 120

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "parallelism", "rccl_ranks", "collectives_per_body")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _num(x, digits=6):
    """floats to `digits` significant figures; NaN / inf (not JSON) to None"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == 0.0:
            return 0.0
        return float("%.*g" % (digits, x))
    if isinstance(x, (int, str)):
        return x
    try:                                   # numpy scalars
        return _num(x.item(), digits)
    except Exception:  # noqa: BLE001
        return str(x)


def _clip(s, n=MAX_NOTE):
    if s is None:
        return None
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def compact(out):
    """the contract's dictionary from the full result `out` (see bench.py main())"""
    line = {k: _num(out.get(k)) for k in CONTRACT_KEYS}
    cfg = out.get("config") or {}
    line["config"] = {k: (_clip(cfg.get(k), 400) if isinstance(cfg.get(k), str) else _num(cfg.get(k))) for k in CONFIG_KEYS}
    roof = out.get("roofline") or {}
    r = {k: (_clip(roof.get(k)) if isinstance(roof.get(k), str) else _num(roof.get(k))) for k in ROOFLINE_KEYS}
    tr = roof.get("in_loop_trace") or {}
    if isinstance(tr, dict) and "mean_ms" in tr:
        r["in_loop_trace"] = {"mean_ms": _num(tr.get("mean_ms")), "launches": _num(tr.get("launches"))}
    for k in ("back_to_back_ms", "hbm_copy_kernel_float4_GBps", "frac_of_practical_peak", "bytes_per_launch", "frac_incl_fused_apply_bytes"):
        if roof.get(k) is not None:
            r[k] = _num(roof.get(k))
    r["note"] = _clip(roof.get("short_note") or "algorithmic bytes per launch / mean in-loop launch duration (kernel trace); details: bench_extras.json")
    line["roofline"] = r
    cpu = out.get("cpu_baseline")
    if isinstance(cpu, dict):
        c = {k: (_clip(cpu.get(k), 240) if isinstance(cpu.get(k), str) else _num(cpu.get(k))) for k in CPU_KEYS}
        for k in ("port", "reference", "threads6"):            # the other CPU figures, numbers only
            sub = cpu.get(k)
            if isinstance(sub, dict) and "value" in sub:
                c[k + "_value"] = _num(sub["value"])
                if "cores" in sub:
                    c[k + "_cores"] = _num(sub["cores"])
        line["cpu_baseline"] = c
    else:
        line["cpu_baseline"] = None
    vr = out.get("value_runs")
    if isinstance(vr, dict):
        line["value_runs"] = {k: _num(vr.get(k)) for k in ("n", "median", "min", "max")}
    for k in ("accepted_fraction", "bench_worker_exit_code"):
        if out.get(k) is not None:
            line[k] = _num(out.get(k))
    line["extras"] = out.get("extras_file", "bench_extras.json")
    return line


def compact_line(out):
    """the line itself; asserts the bound (a longer line is a bug of this file, not something to ride)"""
    d = compact(out)
    s = json.dumps(d, separators=(", ", ": "), allow_nan=False)
    if len(s) >= MAX_LINE:                  # cannot happen with the clips above; shrink the free-text fields further rather than fail the run
        d["config"]["workload"] = _clip(d["config"].get("workload"), 160)
        d["config"]["parallelism"] = _clip(d["config"].get("parallelism"), 120)
        if d.get("cpu_baseline"):
            d["cpu_baseline"]["sample"] = _clip(d["cpu_baseline"].get("sample"), 120)
        s = json.dumps(d, separators=(", ", ": "), allow_nan=False)
    assert len(s) < MAX_LINE, len(s)
    json.loads(s)
    return s


def _sanitize(x):
    if isinstance(x, dict):
        return {str(k): _sanitize(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sanitize(v) for v in x]
    if isinstance(x, float):
        return x if math.isfinite(x) else None
    if isinstance(x, (int, str, bool)) or x is None:
        return x
    try:
        return _sanitize(x.tolist())
    except Exception:  # noqa: BLE001
        return str(x)


def write_extras(out, root):
    """the full result dictionary, next to bench.py (and under gpurun_out/ when it exists, so that a gpurun call brings it home)"""
    paths = [os.path.join(root, "bench_extras.json")]
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        paths.append(os.path.join(root, "gpurun_out", "bench_extras.json"))
    txt = json.dumps(_sanitize(out), indent=1)
    written = []
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(txt)
            written.append(p)
        except OSError:
            pass
    return written


def digest(out):
    """a few lines for stderr: the extras a reader of the driver's log looks for first (numbers only)"""
    rows = []

    def get(*path):
        d = out
        for p in path:
            if not isinstance(d, dict) or p not in d:
                return None
            d = d[p]
        return d

    for label, path in (("keyframe_update it/s", ("value_keyframe_update_inclusive", "value")),
                        ("dropin_optimize it/s", ("dropin_optimize_its_per_s",)),
                        ("dropin_frame ms/frame", ("dropin_frame", "dropin_frame", "ms_per_frame")),
                        ("dropin_frame ms/key-frame", ("dropin_frame", "dropin_frame", "ms_per_keyframe")),
                        ("cpu frame ms/key-frame", ("dropin_frame", "cpu_reference", "ms_per_keyframe")),
                        ("cpu_reference it/s", ("cpu_reference_its_per_s",)),
                        ("lockstep B8 speedup", ("batched_windows", "B8", "speedup_vs_sequential_calls")),
                        ("lockstep B16 frac", ("batched_windows", "roofline", "frac")),
                        ("tolerance it/s", ("tolerance_arith", "value")),
                        ("tracker trials/s", ("tracker", "value"))):
        v = get(*path)
        if isinstance(v, (int, float)):
            rows.append("%s=%.4g" % (label, v))
    return "bench extras: " + "; ".join(rows)
