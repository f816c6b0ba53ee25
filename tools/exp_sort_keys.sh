#!/bin/bash
# Sort-key shapes on the three soups of configs[3]: Morton grid bits per axis x position of the octant bits.
cd "$GRAFT_REPO_ROOT" || exit 1
for s in 0.002 0.01 0.005; do
  for gb in 7 6; do
    for os in 0 6 12; do
      YART_SORT_GRID_BITS=$gb YART_SORT_OCT_SHIFT=$os timeout -k 10 120 python bench.py --cpu-seconds 0 --no-sweep --steps 5 --s $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('s=$s grid_bits=$gb oct_shift=$os', 'Mrays/s', d['value'], 'kernel_ms', r['kernel_ms'], 'sort', r['other_kernels_ms']['sort'])" || exit 1
    done
  done
done
