#!/bin/bash
mkdir -p gpurun_out/r04
D=1 timeout 200 python tools/stamp_winobf2.py 2>&1 | tail -22 | tee gpurun_out/r04/winobf2_stamps_b.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "bf16x3_matches or decoder_matches_oracle" 2>&1 | tail -3
BENCH_C=128 timeout 600 python tools/bench_convbf.py 2>&1 | grep "C=" | cut -c1-26,86-150
BENCH_C=256 timeout 600 python tools/bench_convbf.py 2>&1 | grep "C=" | cut -c1-26,86-150
