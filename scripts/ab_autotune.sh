for i in 1 2; do
python3 bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-extras --autotune 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['config'].get('pieces_autotune'); print('autotune      ms/step %.3f  chose %s (%.3f in pieces / %.3f one launch set)' % (d['ms_per_step'], a['chosen'], a['ms_per_frame_in_pieces'], a['ms_per_frame_one_launch_set']))"
python3 bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pieces        ms/step %.3f' % d['ms_per_step'])"
python3 bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-extras --pieces 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one launch set ms/step %.3f' % d['ms_per_step'])"
done
