#!/bin/bash
# usage (GPU box): scripts/power_vs_data.sh — board power and clocks (rocm-smi) while the SAME kernels render the bench frame with the real appearance-network weights and with
# W1 = W2 = 0 (scripts/hwprobe/weights_power.py): if the board sits at its limit in both and only the time differs, the limit — not the instruction stream — sets the pace.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in real zeroW1W2 real zeroW1W2; do
  WP_VARIANTS=$v WP_LOOP_S=9 python3 scripts/hwprobe/weights_power.py > /tmp/pvd_$v.txt 2>/dev/null &
  BP=$!
  sleep 6                                  # import + scene build + warm-up
  for i in $(seq 1 8); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level" | sed 's/^GPU\[[0-9]*\][ \t]*: //' | tr '\n' ' '
    echo
    sleep 0.5
  done | sed "s/^/$v  /"
  wait $BP
  cat /tmp/pvd_$v.txt | grep loop
done
