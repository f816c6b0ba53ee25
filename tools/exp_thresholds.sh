#!/bin/bash
# The wide kernel's wave-level thresholds on BASELINE configs[3] (env only, no rebuild).
cd "$GRAFT_REPO_ROOT" || exit 1
run() {
  env "$@" timeout -k 10 120 python bench.py --cpu-seconds 0 --no-sweep --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*', 'Mrays/s', d['value'], 'kernel_ms', r['kernel_ms'])" || exit 1
}
run X=0
run YART_WIDE_NODE_MIN=36
run YART_WIDE_NODE_MIN=44
run YART_WIDE_NODE_MIN=48
run YART_WIDE_REFILL_MIN=12
run YART_WIDE_REFILL_MIN=20
run YART_WIDE_SPHERE_MIN=4
run YART_WIDE_SPHERE_MIN=12
run YART_WIDE_SPHERE_MIN=16
run YART_WIDE_CHUNK=128
run X=0
