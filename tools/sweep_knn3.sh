#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
cd $GRAFT_REPO_ROOT
for b in 1024 4096 16384 60000; do
  echo -n "2M blocks=$b: "; N=2000000 QS=1599 RVC_KNN_SCREEN_BLOCKS=$b python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/(.*identical/ identical/'
done
for s in 32 36; do
  echo -n "100k sample=$s blocks=2800: "; QS=1599 RVC_KNN_SCREEN_BLOCKS=2800 RVC_KNN_SAMPLE_TILES=$s python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/(.*identical/ identical/'
done
