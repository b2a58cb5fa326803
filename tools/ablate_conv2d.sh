#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
export NO_TORCH=1 SHAPES=${SHAPES:-2,3,5}
for d in 0 1 2 3 7 15; do echo "DEBUG=$d"; RVC_C2_DEBUG=$d python tools/bench_conv2d.py 2>&1 | grep -- "->" | sed 's/|.*//'; done
for sp in 1 2 4 8; do echo "SPLIT=$sp"; RVC_C2_SPLIT=$sp python tools/bench_conv2d.py 2>&1 | grep -- "->" | sed 's/|.*//'; done
