#!/bin/bash
# One counter pass (instruction and wait counters) over the bench.  usage: bash tools/pmc_sq.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES \
    -d gpurun_out/pmc_${tag}_sq -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > gpurun_out/pmc_${tag}_sq.json 2> gpurun_out/pmc_${tag}_sq.err || { echo "pass failed"; tail -5 gpurun_out/pmc_${tag}_sq.err; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_sq/pmc_results.db "traceWide<false, false, true>" > gpurun_out/${tag}_sq.txt || exit 1
rm -rf gpurun_out/pmc_${tag}_sq/
cat gpurun_out/${tag}_sq.txt
