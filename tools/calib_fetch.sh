#!/bin/bash
# FETCH_SIZE / TCC_EA0_RDREQ calibration passes (tools/calib_fetch.hip).  usage (via gpurun): bash tools/calib_fetch.sh <tag>
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/${tag}_calib_fetch.txt
: > $out
./tools/calib_fetch 1024 3 >> $out || exit 1
for pass in "fetch:FETCH_SIZE" "rdreq:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_MISS_sum" "sizes:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_READ_SECTORS_sum" "dram:TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rm -rf gpurun_out/calib_$name
    timeout -s KILL 120 rocprofv3 --pmc $ctrs -d gpurun_out/calib_$name -o pmc -- ./tools/calib_fetch 1024 3 > gpurun_out/calib_$name.log 2>&1 || { echo "pass $name failed"; tail -5 gpurun_out/calib_$name.log; exit 1; }
    echo "==== rocprofv3 --pmc $ctrs -- ./tools/calib_fetch 1024 3" >> $out
    python3 tools/pmc_summary.py gpurun_out/calib_$name/pmc_results.db calib >> $out || exit 1
    rm -rf gpurun_out/calib_$name
done
echo calib done
