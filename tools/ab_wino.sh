#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# A/B of the vocoder conv selection: RVC_WINO=0 (direct + fused only) vs 1 (fast form wherever supported)
mkdir -p gpurun_out/ab
for w in ${MODES:-0 1}; do
  RVC_WINO=$w python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/ab/wino$w.json 2> gpurun_out/ab/wino$w.err
  echo "WINO=$w rc=$?"; tail -3 gpurun_out/ab/wino$w.err
done
