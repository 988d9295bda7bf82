cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_nppframe
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nppframe -- python3 $GRAFT_REPO_ROOT/scripts/npp_frame_timing.py > $GRAFT_REPO_ROOT/gpurun_out/r6v_npp_frame_prof.txt 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_nppframe -name "*kernel_stats.csv" | head -n 1)
head -40 $f | cut -c1-200 > $GRAFT_REPO_ROOT/gpurun_out/r6v_npp_kernel_stats_head.txt
cat $GRAFT_REPO_ROOT/gpurun_out/r6v_npp_kernel_stats_head.txt | cut -c1-170
