# GPU box: the training-step table of DESIGN 7 (profiles/r03_train_step_timing.txt) — both models, eager chain / static step / hipGraph
mkdir -p /root/repo/gpurun_out/r3t
cd /root/repo
for model in TensorVMSplit REFTensoRF; do
  for mode in "0 0" "1 0" "1 1"; do
    set -- $mode
    TVR_MODEL=$model TVR_STATIC=$1 TVR_GRAPH=$2 python3 scripts/train_step_timing.py 2>&1 | tail -1
  done
done
