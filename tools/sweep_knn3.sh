#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1024 4096 16384 60000; do
  echo -n "2M blocks=$b: "; N=2000000 QS=1599 RVC_KNN_SCREEN_BLOCKS=$b python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/(.*identical/ identical/'
done
for s in 32 36; do
  echo -n "100k sample=$s blocks=2800: "; QS=1599 RVC_KNN_SCREEN_BLOCKS=2800 RVC_KNN_SAMPLE_TILES=$s python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/(.*identical/ identical/'
done
