#!/usr/bin/env python3
"""Hardware counters of k_ef_linearize (rocprofv3 --pmc, one pass per counter group, kernel trace only) on the cfg3 window.

usage (GPU box):  python tools/pmc_linearize.py > profiles/rNN_linearize_counters.txt
The profiled child is `tools/bench_children.py pmc`: the window is loaded and k_ef_linearize is launched 20 times back to back, alone.
Values are per launch (mean over the launches), summed over all instances of the block as rocprofv3 reports them."""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
GROUPS = [
    ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES"],
    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVE_CYCLES"],
    ["SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VMEM"],
    ["SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR"],
    ["TA_TA_BUSY", "TA_TOTAL_WAVEFRONTS", "TA_FLAT_READ_WAVEFRONTS"],
    ["TA_ADDR_STALLED_BY_TC_CYCLES", "TA_DATA_STALLED_BY_TC_CYCLES", "TA_ADDR_STALLED_BY_TD_CYCLES"],
    ["TCP_TOTAL_CACHE_ACCESSES", "TCP_PENDING_STALL_CYCLES", "TCP_TOTAL_READ"],
    ["TCP_TCC_READ_REQ", "TCP_TCC_WRITE_REQ", "TCP_TCC_READ_REQ_LATENCY"],
    ["TCP_TCP_TA_ADDR_STALL_CYCLES", "TCP_TCP_TA_DATA_STALL_CYCLES", "TCP_READ_TAGCONFLICT_STALL_CYCLES"],
    ["TCP_TAGRAM0_REQ", "TCP_TAGRAM1_REQ", "TCP_GATE_EN1"],
    ["TD_TD_BUSY", "TD_TC_STALL", "TD_LOAD_WAVEFRONT"],
    ["TCC_HIT", "TCC_MISS", "TCC_REQ"],
]


def main():
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    kernel = sys.argv[1] if len(sys.argv) > 1 else "k_ef_linearize"
    global GROUPS
    if os.environ.get("SDVGN_PMC_SAFE"):     # the TA / TCP / TD groups never returned on this pool (profiles/r02_notes.txt): SQ + TCC only
        GROUPS = [g for g in GROUPS if not g[0].startswith(("TA_", "TCP_", "TD_"))] + [["FETCH_SIZE"], ["WRITE_SIZE"]]
    print("# rocprofv3 --pmc <group> --kernel-trace -- python tools/bench_children.py pmc   (MI355X; %s, mean per launch)" % kernel)
    for grp in GROUPS:
        d = tempfile.mkdtemp(prefix="sdvgn_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc"] + grp + ["--kernel-trace", "-d", d, "-o", "pmc", "--", sys.executable, os.path.join(ROOT, "tools", "bench_children.py"), "pmc"],
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
            con = sqlite3.connect(dbs[0])
            for c in grp:
                row = con.execute("select avg(value), count(*) from counters_collection where counter_name=? and kernel_name like ?", (c, "%" + kernel + "%")).fetchone()
                print("%-36s %16.1f   (%d launches)" % (c, row[0] if row[0] is not None else float("nan"), row[1]))
        except Exception as ex:  # noqa: BLE001
            print("# group %s failed: %r" % (grp, ex))
        finally:
            shutil.rmtree(d, ignore_errors=True)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
