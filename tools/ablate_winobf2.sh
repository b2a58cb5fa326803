#!/bin/bash
# winobf2.hip with parts compiled out (RVC_W2_DBG bits: 1 input transform, 2 matrix instructions, 4 tap loads, 8 raw-row
# staging, 16 epilogue, 32 window-fragment reads); C = 128, K = 11, 383 760 columns.  Ablation build of the library only.
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
for dbg in ${DBGS:-0 1 2 4 8 16 32 3 37 45 61 63}; do
  echo "== RVC_W2_DBG=$dbg"
  RVC_W2_DBG=$dbg BENCH_C=128 BENCH_K=11 timeout 120 python3 tools/bench_convbf.py 2>&1 | grep "d=1"
done
