#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats``, ROCm 7.2 writes sqlite by default)
into the per-kernel text table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}  (durations in us)")
    print(f"{'calls':>6} {'total_us':>13} {'avg_us':>12} {'min_us':>12} {'max_us':>12} {'%':>6}  "
          f"{'grid':>7} {'wg':>5} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds':>7} {'scratch':>7}  kernel")
    rows = list(cur.execute(
        "select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, "
        "max(grid_x), max(workgroup_x), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
        "max(scratch_size) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1.0
    for r in rows:
        print(f"{r[1]:6d} {r[2]:13.1f} {r[3]:12.1f} {r[4]:12.1f} {r[5]:12.1f} {100 * r[2] / tot:6.2f}  "
              f"{r[6]:7d} {r[7]:5d} {r[8]:5d} {r[9]:5d} {r[10]:5d} {r[11]:7d} {r[12]:7d}  {r[0]}")
    # the network kernel is launched with two grid shapes per render (coarse 64, fine 192 samples/ray): split them
    print("\n# per (kernel, grid) -- separates the coarse and fine passes of net_kernel")
    for r in cur.execute(
            "select name, grid_x, count(*), avg(duration)/1e3, min(duration)/1e3 from kernels "
            "where name like '%net_kernel%' or name like '%composite%' group by name, grid_x, lds_size order by name"):
        print(f"  grid {r[1]:8d}  calls {r[2]:4d}  avg {r[3]:12.1f} us  min {r[4]:12.1f} us  {r[0][:90]}")


if __name__ == "__main__":
    main(sys.argv[1])
