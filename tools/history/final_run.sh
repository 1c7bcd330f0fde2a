# round-4 validation run on the GPU box: tests, smoke, bench (driver setting), rocprof summary, PMC counters of k_ef_linearize (both modes)
set -x
R=r04
timeout 900 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/${R}_bench_final.json 2> gpurun_out/${R}_bench_final.err
cp gpurun_out/inloop_trace_summary_arith0.txt gpurun_out/${R}_inloop_trace_summary_exact.txt 2>/dev/null
cp gpurun_out/inloop_trace_summary_arith1.txt gpurun_out/${R}_inloop_trace_summary_tolerance.txt 2>/dev/null
cp gpurun_out/lockstep_trace_summary_B16.txt gpurun_out/${R}_lockstep_trace_summary_B16.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o ${R}h -- python /root/repo/bench.py --worker --steps 20 --warmup 5 --no-cpu --quick > /root/repo/gpurun_out/${R}_prof_bench.log 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob, json
d=json.loads(open("gpurun_out/r04_bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["iteration_us"], d["accepted_fraction"])
db=glob.glob("/tmp/prof/**/*.db", recursive=True)[0]
con=sqlite3.connect(db)
rows=con.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot=sum(r[3] for r in rows)
out=["# rocprofv3 --kernel-trace --stats -- python bench.py --worker --steps 20 --warmup 5 --no-cpu --quick   (MI355X, last build; --worker: the measuring process itself, not the restarting wrapper)",
     "%-100s %8s %12s %10s %10s %10s %6s" % ("kernel","calls","total_us","avg_us","min_us","max_us","%")]
for r in rows[:45]:
    out.append("%-100s %8d %12.1f %10.3f %10.3f %10.3f %6.2f" % (r[0][:100], r[1], r[3]/1e3, r[2]/1e3, r[4]/1e3, r[5]/1e3, 100*r[3]/tot))
open("gpurun_out/r04_rocprof_summary.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[:14]))
PY
SDVGN_PMC_SAFE=1 SDVGN_BENCH_ARITH=0 timeout 600 python tools/pmc_linearize.py > gpurun_out/${R}_linearize_counters_exact.txt 2>&1
tail -20 gpurun_out/${R}_linearize_counters_exact.txt
