#!/bin/bash
# round 4, GPU call 2: cohabitation with round 3's own aggressor + bisection; first run of winobf2 (tests, per-shape times)
mkdir -p gpurun_out/r04
timeout 900 tools/micro/mfma_cohab 200 > gpurun_out/r04/cohab_micro2.txt 2>&1
for fix in 0 64 128 192 256; do
  echo "== RVC_WINO_FIX=$fix" >> gpurun_out/r04/cohab_bisect2.txt
  RVC_WINO_FIX=$fix timeout 300 tools/micro/mfma_cohab_ablate 200 W3 >> gpurun_out/r04/cohab_bisect2.txt 2>&1
done
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s -k "bf16x3_matches or decoder_matches_oracle or cfg4 or two_streams" > gpurun_out/r04/tests_winobf2.txt 2>&1
tail -5 gpurun_out/r04/tests_winobf2.txt
timeout 600 python tools/bench_convbf.py > gpurun_out/r04/convbf_v2.txt 2>&1
RVC_AMD_LIB=$PWD/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so RVC_WBF_V2=0 timeout 600 python tools/bench_convbf.py > gpurun_out/r04/convbf_v1.txt 2>&1
cat gpurun_out/r04/convbf_v2.txt | cut -c1-150
