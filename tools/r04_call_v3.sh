#!/bin/bash
# winobf3 (balanced form) through the ablation build's switch: parity, then same-box A/B against winobf2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/v3
AB=$GRAFT_REPO_ROOT/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so
RVC_AMD_LIB=$AB RVC_WBF_V3=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "winograd_bf16x3_matches" 2>&1 | tail -15 | tee gpurun_out/v3/tests.txt
