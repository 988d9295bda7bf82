#!/bin/bash
# usage: scripts/ab_bench.sh <variant> [<variant> ...]   (GPU box) — kernel_ms of `bench.py --steps 10` per variant library
# ("default" = lib/libtvr.so, otherwise lib/variants/libtvr_<variant>.so); two interleaved rounds, same box (cdna guide rule 24)
R=${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset TVR_LIB_PATH; else export TVR_LIB_PATH=$R/jittor-myc-nerfs_amd/lib/variants/libtvr_$v.so; fi
  python3 $R/bench.py --steps 10 --warmup 3 --pmc off --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
ra=d.get('roofline_all',{}); cm=(ra.get('march') or {}).get('clock_GHz') or 0.0; cs=(d.get('roofline') or {}).get('clock_GHz') or 0.0
print('%-14s round $round  ms/step %.2f  march %.2f  shade %.2f  composite %.3f   clock GHz march %.3f shade %.3f' % ('$v', d['ms_per_step'], k['march'], k['shade'], k['composite'], cm, cs))"
done; done
