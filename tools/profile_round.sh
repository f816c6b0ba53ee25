#!/bin/bash
# Round profile on the GPU box: kernel-trace stats of the default bench command, then the HBM
# traffic counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit
# one pass; never combined with trace modes other than --kernel-trace).
#   usage (via gpurun): bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>_*.txt/.json
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_${tag}_*
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python3 bench.py "$@" \
    > gpurun_out/${tag}_bench_under_rocprof.json 2> gpurun_out/${tag}_bench_under_rocprof.err || exit 1
python3 tools/rocpd_summary.py gpurun_out/prof_$tag/bench_results.db > gpurun_out/${tag}_kernel_stats.txt || exit 1
: > gpurun_out/${tag}_pmc.txt
for pass in "fetch:FETCH_SIZE SQ_WAVES" "write:WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
            "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" \
            "sizes:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_READ_SECTORS_sum" \
            "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum" \
            "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
            "tcp:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    timeout -s KILL 240 rocprofv3 --pmc $ctrs -d gpurun_out/pmc_${tag}_$name -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" \
        > gpurun_out/pmc_${tag}_$name.json 2> gpurun_out/pmc_${tag}_$name.err || { echo "pass $name failed"; exit 1; }
    echo "==== pass $name: rocprofv3 --pmc $ctrs -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0" >> gpurun_out/${tag}_pmc.txt
    python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_$name/pmc_results.db trace >> gpurun_out/${tag}_pmc.txt || exit 1
done
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_${tag}_*/
echo done
