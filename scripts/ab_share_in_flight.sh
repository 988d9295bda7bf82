for rnd in 1 2; do
for n in 8 4 2; do
for flag in "" "--two-shares-in-flight"; do
python3 bench.py --emulate-world $n --steps 40 --warmup 5 --pmc off --no-cpu-baseline --no-extras $flag 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n round $rnd %-24s ms/step %.3f  two=%s' % ('$flag' or 'one share (default)', d['ms_per_step'], d.get('split_step',{}).get('two_shares_in_flight')))"
done; done; done
