#!/bin/bash
# GPU box (round 5): does the phase rule of csrc/tvr_shade.hip protect anything?
# lib/variants/libtvr_pfree.so (scripts/build_variant.sh pfree -DTVR_PHASE_FREE=1) is the shipped two-frequency render kernel with 12 global loads per lane
# issued BETWEEN layer 1's and layer 2's MFMAs (the next tile's first tap set).  Run ONCE each: the bitwise-reproducibility test, the full-size chunk /
# permutation / oracle test and the config1 golden test through that library; then the 8 bench frames from both libraries, two repeats each, as digests.
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/jittor-myc-nerfs_amd/lib/variants/libtvr_pfree.so
cd $R
echo "== tests through the phase-free library"
TVR_LIB_PATH=$V python3 -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "run_to_run_determinism or full_size_properties or config1_against_golden or tiny_dense" 2>&1 | tail -5 || exit 1
echo "== frame digests (default library, then the phase-free one)"
python3 scripts/frame_digest.py || exit 1
TVR_LIB_PATH=$V python3 scripts/frame_digest.py || exit 1
echo "== interleaved timing"
scripts/ab_bench.sh default pfree
