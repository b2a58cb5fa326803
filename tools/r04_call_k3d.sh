#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k3
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "winograd_bf16x3_matches" 2>&1 | tail -3
for C in 256 128; do
  BENCH_C=$C BENCH_K=3,7,11 timeout 600 python tools/bench_convbf.py 2>&1 | grep "C=" | tee -a gpurun_out/k3/shapes_f.txt
done
K=3 timeout 300 python tools/stamp_winobf2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k3/stamps_k3_f.txt
K=11 timeout 300 python tools/stamp_winobf2.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/k3/stamps_k11_f.txt
