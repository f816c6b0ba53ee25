#!/bin/bash
# A/B of library builds on one box: bench.py per build and soup scale; prints kernel ms.   bash tools/exp_libs.sh "<libs>" "<scales>"
cd "$GRAFT_REPO_ROOT" || exit 1
for s in $2; do
  for lib in $1; do
    if [ "$lib" = "default" ]; then unset YART_AMD_LIB; else export YART_AMD_LIB=$PWD/yart_amd/libyart_amd_$lib.so; fi
    timeout -k 10 120 python bench.py --cpu-seconds 0 --no-sweep --steps 5 --s $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$lib', 's=$s', 'Mrays/s', d['value'], 'kernel_ms', r['kernel_ms'], 'other', r['other_kernels_ms'], r['per_ray_fetches'])" || exit 1
  done
done
