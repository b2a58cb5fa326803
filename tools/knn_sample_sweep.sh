#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# bench.py's kNN leg (the real utterance's queries) against the number of sampled tiles and main-pass blocks
for s in ${SAMPLES:-32 40}; do for b in ${BLOCKS:-1024 1400 2800}; do
  echo -n "sample_tiles=$s blocks=$b: "; RVC_KNN_SAMPLE_TILES=$s RVC_KNN_SCREEN_BLOCKS=$b python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
l = json.loads([x for x in sys.stdin if x.startswith('{')][-1]); k = l['roofline_knn']; print(k['avg_search_ms'], k['frac'])"
done; done
