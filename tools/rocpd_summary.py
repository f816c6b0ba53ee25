"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel count / total / average.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, sgpr_count from kernels").fetchall() \
    if 'kernels' in [r[0] for r in c.execute("select name from sqlite_master")] else []
agg = collections.OrderedDict()
for name, s, e, grid, wg, lds, scr, vg, sg in rows:
    a = agg.setdefault(name, [0, 0, 10**18, 0, grid, wg, lds, scr, vg, sg])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
print("%-100s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct"))
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-100s %6d %12.3f %12.3f %12.3f %12.3f %6.1f   grid=%s wg=%s lds=%s scratch=%s vgpr=%s sgpr=%s"
          % (name[:100], a[0], a[1] / 1e6, a[1] / a[0] / 1e6, a[2] / 1e6, a[3] / 1e6, 100.0 * a[1] / tot, *a[4:]))
