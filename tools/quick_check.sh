#!/bin/bash
# one GPU call while iterating on a kernel: the GPU suite (or the files given), three bench workers, the in-loop kernel trace of the headline protocol
export AMD_LOG_LEVEL=0
mkdir -p gpurun_out/loop
timeout 1500 python -m pytest ${@:-tests} -x -q -m gpu -p no:cacheprovider 2>&1 | tail -4
bash tools/loop_bench.sh 3 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/qprof -o q -- python /root/repo/tools/bench_children.py trace > /dev/null 2>&1
cd /root/repo
python - <<'PY'
import sqlite3, glob
db = glob.glob("/tmp/qprof/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
t = [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
try:
    rows = con.execute("select name, count(*), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
except Exception as ex:
    rows = []
    print("kernels view missing:", ex)
for r in rows[:14]:
    print("%-70s %5d %9.3f %9.3f %9.3f" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3))
PY
