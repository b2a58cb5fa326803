#!/bin/bash
# K10b "where does the time go": the ablation library with parts of the kernel left out (wrong results on purpose)
export RVC_AMD_LIB=codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so
for d in 0 1 2 4 8 16 3 14 30 31; do
  echo "== RVC_C2B_DEBUG=$d"
  RVC_C2B_DEBUG=$d timeout 120 python tools/bench_conv2dbf.py 2>&1 | grep -v amdgpu.ids | grep -- '->' | awk '{print $1,$2,$3,$4,$5,$6, "K10b", $14, "us"}' | tr '\n' ';'; echo | cat
done
