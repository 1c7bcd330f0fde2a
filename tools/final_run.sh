# round-6 validation run on the GPU box: tests, smoke, bench (driver flags and defaults), rocprof summary, in-loop trace summaries, PMC counters of
# k_ef_linearize.  Everything judged is copied from gpurun_out/ into profiles/ afterwards.
set -x
R=r06
O=gpurun_out/${R}
mkdir -p $O
export AMD_LOG_LEVEL=0
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -aE "passed|failed|^FAILED|^ERROR" | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err ) 2>&1 | grep real
echo "bench (driver flags) exit code $?"
cp bench_extras.json $O/bench_driver_flags_extras.json 2>/dev/null
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err
echo "bench (defaults) exit code $?"
cp bench_extras.json $O/bench_final_extras.json 2>/dev/null
cp gpurun_out/inloop_trace_summary_arith0.txt $O/inloop_trace_summary_exact.txt 2>/dev/null
cp gpurun_out/inloop_trace_summary_arith1.txt $O/inloop_trace_summary_tolerance.txt 2>/dev/null
cp gpurun_out/lockstep_trace_summary_B16.txt $O/lockstep_trace_summary_B16.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${R}h -- python /root/repo/bench.py --worker --steps 20 --warmup 5 --no-cpu --quick > /root/repo/$O/prof_bench.log 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob, json
for f in ("bench_driver_flags", "bench_final"):
    try:
        line = open("gpurun_out/r06/%s.json" % f).read().strip().splitlines()[-1]
        d = json.loads(line)
        x = json.load(open("gpurun_out/r06/%s_extras.json" % f))
        kf = x.get("value_keyframe_update_inclusive", {})
        fr = x.get("dropin_frame", {})
        print(f, "line bytes", len(line), "value", d["value"], "ms", d["ms_per_step"], "runs", d.get("value_runs", {}).get("min"), d.get("value_runs", {}).get("max"), "frac", d["roofline"]["frac"],
              "cpu", d.get("cpu_baseline", {}).get("value"), "look_ahead", x.get("look_ahead"), "exit", d.get("bench_worker_exit_code"), "| kf", kf.get("value"), kf.get("ms_per_keyframe"),
              "| dropin", x.get("dropin_optimize_its_per_s"), x.get("dropin_solveSystemF_its_per_s"), x.get("cpu_reference_its_per_s"),
              "| frame B+ ms/kf", fr.get("dropin_frame", {}).get("ms_per_keyframe"), "cpu", fr.get("cpu_reference", {}).get("ms_per_keyframe"))
    except Exception as ex:
        print(f, "unreadable:", repr(ex))
db = glob.glob("/tmp/prof/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[3] for r in rows)
out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --worker --steps 20 --warmup 5 --no-cpu --quick   (MI355X, round 6, last build; --worker: the measuring process itself)",
       "%-100s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%")]
for r in rows[:50]:
    out.append("%-100s %8d %12.1f %10.3f %10.3f %10.3f %6.2f" % (r[0][:100], r[1], r[3] / 1e3, r[2] / 1e3, r[4] / 1e3, r[5] / 1e3, 100 * r[3] / tot))
open("gpurun_out/r06/rocprof_summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out[:16]))
PY
SDVGN_PMC_SAFE=1 SDVGN_BENCH_ARITH=0 timeout 600 python tools/pmc_linearize.py > $O/linearize_counters_exact.txt 2>&1
tail -12 $O/linearize_counters_exact.txt

SDVGN_DEBUG_FLAGS=64 timeout 300 python tools/exp_tail_stamps.py > $O/tail_stamps.txt 2>&1
tail -3 $O/tail_stamps.txt | cut -c1-300
bash tools/pmc_loop_kernels.sh r06 > /dev/null 2>&1
# the three forms of the loop on this box (default | accept test as a launch of its own | applyRes workgroups in the statistics launch: rounds 3-5), interleaved
bash tools/ab_bench.sh 4 - SDVGN_DEBUG_FLAGS=512 SDVGN_FUSED_APPLY=0 2>&1 | sort -k2,2 -s > $O/ab_loop_forms.txt
tail -12 $O/ab_loop_forms.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/gp -o g -- python /root/repo/tools/bench_children.py trace > /dev/null 2>&1
cd /root/repo
python tools/gap_report.py /tmp/gp 50 > $O/loop_timeline.txt 2>&1
tail -3 $O/loop_timeline.txt
