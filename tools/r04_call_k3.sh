#!/bin/bash
# three-tap F(4,3) form on winobf2.hip: parity, then timing against the fp32 Winograd kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k3
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "winograd_bf16x3_matches" -s 2>&1 | tail -45 > gpurun_out/k3/tests.txt
tail -5 gpurun_out/k3/tests.txt
for C in 256 128; do
  BENCH_C=$C BENCH_K=3,7 timeout 600 python tools/bench_convbf.py 2>&1 | tee -a gpurun_out/k3/shapes.txt
done
