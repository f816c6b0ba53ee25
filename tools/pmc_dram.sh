#!/bin/bash
# One counter pass (HBM-side request counters) over the bench.  usage: bash tools/pmc_dram.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum \
    -d gpurun_out/pmc_${tag}_dram -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > gpurun_out/pmc_${tag}_dram.json 2> gpurun_out/pmc_${tag}_dram.err || { echo "pass failed"; tail -5 gpurun_out/pmc_${tag}_dram.err; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_dram/pmc_results.db "traceWide<false, false, true>" > gpurun_out/${tag}_dram.txt || exit 1
rm -rf gpurun_out/pmc_${tag}_dram/
cat gpurun_out/${tag}_dram.txt
