#!/bin/bash
# usage (GPU box): scripts/power_trace.sh [bench args]   — samples rocm-smi power / clocks while `bench.py --steps 900` runs (the evidence behind DESIGN.md 4.2's "power-bound")
cd ${GRAFT_REPO_ROOT:-/root/repo}
python3 bench.py --steps 900 --warmup 3 --pmc off --no-cpu-baseline --no-extras "$@" > /tmp/pt_bench.json 2>/dev/null &
BP=$!
sleep 6                                   # model build + warm-up
for i in $(seq 1 10); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|mclk clock level" | tr '\n' ' '
  echo
  sleep 0.4
done
wait $BP
python3 -c "
import json;d=json.loads(open('/tmp/pt_bench.json').read().strip().splitlines()[-1]);print('bench', d['config']['arith'], 'ms/step %.2f' % d['ms_per_step'], d['kernel_ms'])"
