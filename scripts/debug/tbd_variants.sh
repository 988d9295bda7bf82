mkdir -p /root/repo/gpurun_out/r3t
cd /tmp && export TMPDIR=/tmp
for v in default tbd1 tbd2 tbd4 tbd8 tbd15; do
  if [ "$v" = default ]; then unset TVR_LIB_PATH; else export TVR_LIB_PATH=/root/repo/jittor-myc-nerfs_amd/lib/variants/libtvr_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3t/tbd_$v -- python3 /root/repo/scripts/train_step_timing.py > /root/repo/gpurun_out/r3t/tbd_$v.log 2>&1
  echo "$v: $(grep march_backward /root/repo/gpurun_out/r3t/tbd_$v/*/*_kernel_stats.csv | cut -d, -f2-4 | tail -1)  $(tail -1 /root/repo/gpurun_out/r3t/tbd_$v.log | cut -c1-20)"
done
