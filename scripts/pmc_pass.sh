#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counter> [<counter> ...]   — one rocprofv3 PMC pass over a short bench run (GPU box)
# PMC_PROG="scripts/ngp_frame_timing.py --frames 2" selects another program of this repo (default: bench.py, 3 steps)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
cd /tmp
timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/${PMC_PROG:-bench.py --steps 3 --warmup 1 --no-cpu-baseline} > $R/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag rc=$?"
