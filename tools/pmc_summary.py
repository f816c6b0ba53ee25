"""Per-kernel averages of the counters in a rocprofv3 --pmc result database.
    python tools/pmc_summary.py gpurun_out/pmc_x/pmc_results.db [substring-filter]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else "traceKernel"
rows = c.execute("select kernel_name, dispatch_id, counter_name, value, duration from counters_collection").fetchall()
per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
dur = {}
for k, d, n, v, du in rows:
    if flt in k:
        per[(k, d)][n] += v
        dur[(k, d)] = du
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for (k, d), cs in per.items():
    for n, v in cs.items():
        agg[k][n].append(v)
    agg[k]["duration_ns(profiled)"].append(dur[(k, d)])
for k, cs in agg.items():
    print(k)
    for n, vs in sorted(cs.items()):
        print("    %-34s dispatches=%d  avg=%.6g  min=%.6g  max=%.6g" % (n, len(vs), sum(vs) / len(vs), min(vs), max(vs)))
