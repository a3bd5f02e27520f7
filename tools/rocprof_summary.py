#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats``, ROCm 7.2 writes sqlite by default)
into the per-kernel text table committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r02_kernel_stats.txt
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
    # The network kernel runs twice per render (persistent grid): coarse pass (64 samples/ray), then fine pass (192
    # samples/ray).  On the split-bender path the two are different instantiations (the fine one has no bender layers)
    # and show up as separate rows above; on the fused path they are the same kernel and alternate, so split by parity.
    names = [n for n, in cur.execute("select distinct name from kernels where name like '%net_kernel%'").fetchall()]
    print("\n# net_kernel by pass")
    if len(names) == 1:
        d = [r[0] / 1e3 for r in cur.execute("select duration from kernels where name = ? order by start", (names[0],))]
        parts = [("coarse (even launches)", d[0::2], names[0]), ("fine   (odd launches) ", d[1::2], names[0])]
    else:
        parts = []
        for name in names:
            d = [r[0] / 1e3 for r in cur.execute("select duration from kernels where name = ? order by start", (name,))]
            parts.append(("", d, name))
        parts.sort(key=lambda p: sum(p[1]) / max(len(p[1]), 1))
        parts = [(("coarse" if i == 0 else "fine  ") + " (own instantiation) ", d, n) for i, (_, d, n) in enumerate(parts)]
    for label, part, name in parts:
        if part:
            print(f"  {label}: calls {len(part):4d}  avg {sum(part) / len(part):12.1f} us  min {min(part):12.1f} us  "
                  f"max {max(part):12.1f} us   {name[:90]}")


if __name__ == "__main__":
    main(sys.argv[1])
