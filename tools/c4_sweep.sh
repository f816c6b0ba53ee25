#!/bin/bash
# C4 sweep of SURVEY.md 8d: soup edge scale s in {0.002, 0.005, 0.01} and batch sizes 2^16..2^24.
#   usage (via gpurun): bash tools/c4_sweep.sh > gpurun_out/c4_sweep.jsonl
for s in 0.002 0.005 0.01; do
    timeout -k 10 300 python bench.py --cpu-seconds 0 --s $s || exit 1
done
for r in 65536 262144 1048576 4194304; do
    timeout -k 10 300 python bench.py --cpu-seconds 0 --rays $r --steps 10 || exit 1
done
