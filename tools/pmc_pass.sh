#!/bin/bash
# One rocprofv3 counter pass over a short bench run.  usage: tools/pmc_pass.sh <name> "<counters>" [bench args]
# (each pass is its own process; never combined with trace modes other than --kernel-trace)
name=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -s KILL 240 rocprofv3 --pmc $ctrs -d gpurun_out/pmc_$name -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 "$@" > gpurun_out/pmc_$name.json 2> gpurun_out/pmc_$name.err
echo "pass $name exit=$?"
