#!/bin/bash
# usage (GPU box): scripts/emulate8_ab.sh — rank 0's share of the 8-way split of the bench frame, two interleaved rounds:
# plain calls / ShardedFramePipeline without and with the hipGraph capture, without and with the (device-side) exchange on the side stream / shard tile sizes
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 bench.py --steps 40 --warmup 3 --no-cpu-baseline --pmc off "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']
print('%-62s ms/step %.3f  rays %d  march %.3f shade %.3f comp %.3f' % ('$*', d['ms_per_step'], d['config']['rays_per_rank'], k['march'], k['shade'], k['composite']))"; }
for round in 1 2; do
run
run --emulate-world 8 --no-pipeline
run --emulate-world 8 --no-overlap-exchange
run --emulate-world 8
run --emulate-world 8 --graph --no-overlap-exchange
run --emulate-world 8 --graph
run --emulate-world 8 --tile 1024
run --emulate-world 8 --tile 512
run --emulate-world 8 --tile 256
done
