#!/bin/bash
# GPU box: a training step of a scene with TensorBase's default encoding frequencies (view_pe = fea_pe = 6: 390 MLP inputs) under rocprofv3 --kernel-trace --stats.
# Round 5: the eager chain of autograd Functions (Linears on tvr_linear_dx / tvr_gemm_tn); round 6: the FUSED step (tvr_train_forward / _backward, 16 / 48 components) —
# the trace must hold no library GEMM (`Cijk_*`).  usage: scripts/pe6_train_trace.sh <out dir>
set -e
export TMPDIR=/tmp TVR_PE=6
R=${GRAFT_REPO_ROOT:-/root/repo}
O=${1:-$R/gpurun_out/r06_pe6}
mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/scripts/train_step_timing.py > $O/train_step.txt 2> $O/prof.err
tail -3 $O/train_step.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
echo "library GEMM kernels in the trace: $(grep -c Cijk $f || true)"
head -12 $f | cut -c1-150
