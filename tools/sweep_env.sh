#!/bin/bash
# Bench the same build under several values of one environment knob.  usage: bash tools/sweep_env.sh NAME v1 v2 ... -- [bench args]
name=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
    export $name=$v
    timeout -k 10 200 python3 bench.py --cpu-seconds 0 "$@" > gpurun_out/sweep_${name}_$v.json 2> gpurun_out/sweep_${name}_$v.err || { echo "$name=$v failed"; tail -3 gpurun_out/sweep_${name}_$v.err; exit 1; }
    python3 - "$name" "$v" gpurun_out/sweep_${name}_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%s=%s: %.1f Mrays/s, step %.3f ms, kernel %.3f ms, child records %.2f, tri %.2f, sph %.2f" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], r["kernel_ms"],
      r["per_ray_fetches"]["child_records_12B"], r["per_ray_fetches"]["triangle_plane_rows_16B"], r["per_ray_fetches"]["sphere_records_16B"]), flush=True)
PY
done
