#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace --stats) as plain text.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_x.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"{'calls':>6} {'avg_us':>12} {'total_us':>14} {'pct':>7}  kernel")
    for name, calls, total, avg, pct in rows[:15]:
        print(f"{calls:6d} {avg:12.2f} {total:14.2f} {pct:7.3f}  {name[:110]}")
    q = """select name, count(*), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count),
                  max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x)
           from kernels where name like '%hipfeat%' group by name"""
    print("\n# hipfeat kernels: per-dispatch resources")
    for r in c.execute(q):
        print(
            f"{r[0][:80]}\n   dispatches={r[1]} avg={r[2]/1e3:.2f}us min={r[3]/1e3:.2f}us max={r[4]/1e3:.2f}us "
            f"vgpr={r[5]} agpr={r[6]} sgpr={r[7]} lds={r[8]}B scratch={r[9]} grid={r[10]} wg={r[11]}"
        )


if __name__ == "__main__":
    main(sys.argv[1])
