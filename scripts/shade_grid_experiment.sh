cd ${GRAFT_REPO_ROOT:-/root/repo}
export TVR_LIB_PATH=$PWD/jittor-myc-nerfs_amd/lib/variants/libtvr_expgrid.so
for g in 256 192 128 64; do
TVR_EXP_GRID_SHADE=$g python3 bench.py --steps 10 --warmup 3 --pmc off --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; r=d['roofline_all']['shade']
print('shade grid $g: %.2f ms  (x CUs = %.0f CU-ms)  clock %.3f GHz' % (k['shade'], k['shade']*$g, r['clock_GHz']))"
done
