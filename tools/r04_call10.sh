#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 tools/micro/mfma_cohab 100 W > gpurun_out/r04/cohab_micro4_W.txt 2>&1
timeout 600 tools/micro/mfma_cohab 100 X3 > gpurun_out/r04/cohab_micro4_X.txt 2>&1
echo "== RVC_WINO_FIX=2048" > gpurun_out/r04/cohab_bisect4.txt
RVC_WINO_FIX=2048 timeout 300 tools/micro/mfma_cohab_ablate 100 W3 2>&1 | grep -v "bare\|packed" >> gpurun_out/r04/cohab_bisect4.txt
grep -h "^W\|^X\|next to\|==" gpurun_out/r04/cohab_micro4_W.txt gpurun_out/r04/cohab_micro4_X.txt gpurun_out/r04/cohab_bisect4.txt | cut -c1-210
