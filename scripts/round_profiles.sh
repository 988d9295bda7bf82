#!/bin/bash
# GPU box: the round's evidence under gpurun_out/<tag>_* — the default bench line, the same command under rocprofv3 --kernel-trace --stats, and the two secondary kernels
# the round-4 review asked a fraction for (ngp_render_kernel, bg_mlp_kernel).  usage: scripts/round_profiles.sh r06
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r06}
O=$R/gpurun_out
cd $R
echo "== bench (default command)"; python3 bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || { tail -5 $O/${tag}_bench.err; exit 1; }
tail -c 600 $O/${tag}_bench.json; echo
cd /tmp
echo "== bench under rocprofv3 --kernel-trace --stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_bench -- python3 $R/bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-extras > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_prof_bench.err
echo "== round 6: the frame as ONE launch set per call (--pieces 0: the kernels the roofline is quoted on), bench line + rocprofv3 kernel trace"
python3 $R/bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-extras --pieces 0 > $O/${tag}_bench_pieces0.json 2> $O/${tag}_bench_pieces0.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_bench_pieces0 -- python3 $R/bench.py --steps 20 --warmup 3 --pmc off --no-cpu-baseline --no-extras --pieces 0 > $O/${tag}_bench_pieces0_under_rocprof.json 2> $O/${tag}_prof_bench_pieces0.err
echo "== NGP alt path under rocprofv3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_ngp -- python3 $R/bench.py --model NGPNetworks --steps 10 --warmup 2 --pmc off --no-cpu-baseline > $O/${tag}_ngp_bench_under_rocprof.json 2> $O/${tag}_prof_ngp.err
echo "== NGP alt path, bench line with live PMC"
cd $R; python3 bench.py --model NGPNetworks --steps 10 --warmup 2 --no-cpu-baseline > $O/${tag}_ngp_bench.json 2> $O/${tag}_ngp_bench.err || true
cd /tmp
echo "== NerfPlusPlus background-network kernel under rocprofv3"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_prof_npp -- python3 $R/scripts/npp_roofline.py > $O/${tag}_npp_roofline.json 2> $O/${tag}_prof_npp.err
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${tag}_pmc_npp_$n -- python3 $R/scripts/npp_roofline.py > /dev/null 2> $O/${tag}_pmc_npp_$n.err || echo "pmc pass $n failed"
done
cd $R
find $O/${tag}_prof_* -name "*kernel_stats.csv" | while read f; do echo "--- $f"; head -6 "$f" | cut -c1-200; done
cat $O/${tag}_npp_roofline.json | tail -1 | cut -c1-600
