#!/bin/bash
# GPU box: the 8-way share through the RCCL branch of the pipeline (a ONE-member nccl group: same code and streams as an 8-GPU run, no wire) — one share in flight, two shares in
# flight, and two shares in flight with more hardware queues than HIP's default four (the pipeline then has five streams: caller, two render streams, the exchange's side stream, RCCL's own)
for rnd in 1 2; do
for cfg in "one:--one-share-in-flight:" "two::" "two+8q::GPU_MAX_HW_QUEUES=8"; do
  name=${cfg%%:*}; rest=${cfg#*:}; flag=${rest%%:*}; envs=${rest#*:}
  env $envs python3 bench.py --emulate-world 8 --one-rank-exchange --steps 40 --warmup 5 --pmc off --no-cpu-baseline --no-extras $flag 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $rnd %-8s ms/step %.3f  rays/rank %s  two=%s backend=%s' % ('$name', d['ms_per_step'], d['config'].get('rays_per_rank'), d.get('split_step',{}).get('two_shares_in_flight'), d.get('exchange_backend')))"
done; done
