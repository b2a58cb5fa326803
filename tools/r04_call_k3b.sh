#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/k3
timeout 1200 python -m pytest tests/test_fullsize_gpu.py tests/test_decoder_gpu.py -m gpu -x -q -k "stage_by_stage or decoder" 2>&1 | tail -8 | tee gpurun_out/k3/tests_dec.txt
timeout 600 python bench.py --steps 8 --warmup 2 2>&1 | tail -1 | tee gpurun_out/k3/bench_cfg2.json
