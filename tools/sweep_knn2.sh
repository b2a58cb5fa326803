#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# screened kNN at the cfg-2 shape (1599 x 100000 x 768): block count of the main pass x sampled tiles
cd $GRAFT_REPO_ROOT
for b in 252 504 1024 2800; do for s in 36 63 73; do
  echo -n "blocks=$b sample_tiles=$s: "; QS=1599 RVC_KNN_SCREEN_BLOCKS=$b RVC_KNN_SAMPLE_TILES=$s python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/(.*identical/ identical/'
done; done
