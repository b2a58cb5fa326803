#!/bin/bash
# 64-row blocks of winobf2 (ablation build) against winobf.hip at the 64-channel stage; stamps of the 3-tap form
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k3
AB=$GRAFT_REPO_ROOT/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so
echo "== winobf.hip (64 x 128 blocks)" | tee gpurun_out/k3/c64.txt
RVC_AMD_LIB=$AB BENCH_C=64 BENCH_K=7,11 timeout 600 python tools/bench_convbf.py 2>&1 | grep "C=" | tee -a gpurun_out/k3/c64.txt
echo "== winobf2.hip, 64-row blocks (RVC_WBF_V2_64=1)" | tee -a gpurun_out/k3/c64.txt
RVC_AMD_LIB=$AB RVC_WBF_V2_64=1 BENCH_C=64 BENCH_K=7,11 timeout 600 python tools/bench_convbf.py 2>&1 | grep "C=" | tee -a gpurun_out/k3/c64.txt
RVC_AMD_LIB=$AB RVC_WBF_V2_64=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "winograd_bf16x3_matches" 2>&1 | tail -3 | tee -a gpurun_out/k3/c64.txt
K=3 timeout 300 python tools/stamp_winobf2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k3/stamps_k3.txt
