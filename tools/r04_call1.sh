#!/bin/bash
# round 4, GPU call 1: cohabitation micro-reproducer + bisection of wino.hip, the new full-length parity tests, a baseline bench
mkdir -p gpurun_out/r04
timeout 600 tools/micro/mfma_cohab 200 > gpurun_out/r04/cohab_micro.txt 2>&1
for fix in 0 64 128 192 256; do
  echo "== RVC_WINO_FIX=$fix" >> gpurun_out/r04/cohab_bisect.txt
  RVC_WINO_FIX=$fix timeout 200 tools/micro/mfma_cohab_ablate 200 W3 >> gpurun_out/r04/cohab_bisect.txt 2>&1
done
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -s -k "T3198 or plain_peaked or inflight2" > gpurun_out/r04/tests_fullsize_new.txt 2>&1
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_cfg2_start.json 2> gpurun_out/r04/bench_cfg2_start.err
timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04/bench_cfg4_start.json 2> gpurun_out/r04/bench_cfg4_start.err
tail -3 gpurun_out/r04/tests_fullsize_new.txt
