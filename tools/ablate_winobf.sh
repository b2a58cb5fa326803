#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# winobf.hip with parts compiled out (RVC_WBF_DBG bits: 1 input transform, 2 matrix instructions, 4 tap DMA, 8 per-step
# barrier, 16 raw-row staging); C = 128, K = 11, 383 760 columns
cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 3 4 8 16 32; do
  echo "== RVC_WBF_DBG=$dbg"
  RVC_WBF_DBG=$dbg BENCH_C=128 BENCH_K=11 timeout 120 python3 tools/bench_convbf.py 2>&1 | grep "d=1"
done
