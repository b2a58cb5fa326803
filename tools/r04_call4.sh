#!/bin/bash
mkdir -p gpurun_out/r04
rm -f gpurun_out/r04/cohab_bisect3.txt
for fix in 512 1024 768; do
  echo "== RVC_WINO_FIX=$fix" >> gpurun_out/r04/cohab_bisect3.txt
  RVC_WINO_FIX=$fix timeout 300 tools/micro/mfma_cohab_ablate 100 W3 2>&1 | grep -v "bare\|packed" >> gpurun_out/r04/cohab_bisect3.txt
done
cut -c1-200 gpurun_out/r04/cohab_bisect3.txt
