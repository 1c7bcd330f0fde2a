set -x
timeout 600 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED" | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_v13.json 2> gpurun_out/r02_bench_v13.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r02h -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu > /root/repo/gpurun_out/prof_bench_v8.log 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob, json
d=json.loads(open("gpurun_out/r02_bench_v13.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["iteration_us"], d["accepted_fraction"])
db=glob.glob("/tmp/prof/**/*.db", recursive=True)[0]
con=sqlite3.connect(db)
rows=con.execute("select name, count(*), avg(duration), sum(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot=sum(r[3] for r in rows)
out=["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu   (MI355X, round 2, last build)",
     "%-100s %8s %12s %10s %10s %10s %6s" % ("kernel","calls","total_us","avg_us","min_us","max_us","%")]
for r in rows[:45]:
    out.append("%-100s %8d %12.1f %10.3f %10.3f %10.3f %6.2f" % (r[0][:100], r[1], r[3]/1e3, r[2]/1e3, r[4]/1e3, r[5]/1e3, 100*r[3]/tot))
open("gpurun_out/r02_rocprof_v8_summary.txt","w").write("\n".join(out)+"\n")
print("\n".join(out[:14]))
PY
