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
    print("# (registers: this rocprofv3 reports HALF of a wave64 kernel's allocation on gfx950 -- it applies the 4-register granule to the")
    print("#  kernel descriptor's granulated count, the hardware allocates in granules of 8: 126 VGPRs in the build -> 128 allocated -> 64 here;")
    print("#  143 -> 72, 161 -> 84, 181 -> 92, checked against -Rpass-analysis=kernel-resource-usage on six instances.  `registers` below is")
    print("#  2 x (vgpr + agpr) of the trace = the allocation that decides occupancy: waves/SIMD = floor(512 / registers).)")
    for r in c.execute(q):
        regs = 2 * (int(r[5] or 0) + int(r[6] or 0))
        print(
            f"{r[0][:80]}\n   dispatches={r[1]} avg={r[2]/1e3:.2f}us min={r[3]/1e3:.2f}us max={r[4]/1e3:.2f}us "
            f"registers={regs} ({512 // regs if regs else '?'} waves/SIMD by registers; trace fields vgpr={r[5]} agpr={r[6]}) sgpr={r[7]} lds={r[8]}B scratch={r[9]} grid={r[10]} wg={r[11]}"
        )

if __name__ == "__main__":
    main(sys.argv[1])


def gaps(path):
    """Busy time vs span of the hipfeat dispatches (one-stream traces: how much of the wall time is gaps between kernels)."""
    c = sqlite3.connect(path)
    try:
        rows = c.execute("select start, end, name from kernels where name like '%hipfeat%' order by start").fetchall()
    except Exception as e:  # noqa: BLE001
        print("# no start/end columns:", e)
        return
    if len(rows) < 4:
        return
    rows = rows[len(rows) // 4:]  # skip warm-up
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gl = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    gl.sort()
    print(f"\n# timeline of the last {len(rows)} hipfeat dispatches: span {span/1e3:.1f} us, kernels busy {busy/1e3:.1f} us ({busy/span:.3f}), "
          f"gap median {gl[len(gl)//2]/1e3:.2f} us, p90 {gl[int(len(gl)*0.9)]/1e3:.2f} us, negative gaps (overlap) {sum(1 for g in gl if g < 0)}")


if __name__ == "__main__" and len(sys.argv) > 1:
    gaps(sys.argv[1])
