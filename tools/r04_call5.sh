#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "bf16x3_matches or decoder_matches_oracle or cfg4 or two_streams" > gpurun_out/r04/tests_winobf2_b.txt 2>&1
tail -3 gpurun_out/r04/tests_winobf2_b.txt
timeout 600 python tools/bench_convbf.py > gpurun_out/r04/convbf_v2b.txt 2>&1
grep "C=" gpurun_out/r04/convbf_v2b.txt | cut -c1-26,86-150
DBGS="0 64 1 2 4 16 61" bash tools/ablate_winobf2.sh > gpurun_out/r04/winobf2_ablation_b.txt 2>&1
cat gpurun_out/r04/winobf2_ablation_b.txt | cut -c1-26,86-150
