#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# ablation of the Winograd conv (k = 11, C = 128): which part of the loop costs what
export BENCH_C=${BENCH_C:-128} BENCH_K=11 RVC_WINO_R4=0   # the ablation builds exist for the F(4,3) form
for d in ${DBGS:-0 16 32 64 15 31 47 63 127}; do echo "DBG=$d"; RVC_WINO_DBG=$d python tools/bench_conv.py 2>&1 | grep "C="; done
