#!/bin/bash
# Three short counter passes over the bench (wide kernel diagnosis).  usage: bash tools/pmc_quick.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
: > gpurun_out/${tag}_pmc.txt
for pass in "fetch:FETCH_SIZE TCC_HIT_sum" "miss:WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
            "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
            "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 240 rocprofv3 --pmc $ctrs -d gpurun_out/pmc_${tag}_$name -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" \
        > gpurun_out/pmc_${tag}_$name.json 2> gpurun_out/pmc_${tag}_$name.err || { echo "pass $name failed"; tail -5 gpurun_out/pmc_${tag}_$name.err; exit 1; }
    echo "==== pass $name: rocprofv3 --pmc $ctrs -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 $*" >> gpurun_out/${tag}_pmc.txt
    python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_$name/pmc_results.db trace >> gpurun_out/${tag}_pmc.txt || exit 1
done
rm -rf gpurun_out/pmc_${tag}_*/
echo done
