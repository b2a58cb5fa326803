cd $GRAFT_REPO_ROOT
for g in 1 0; do
  echo -n "glds=$g 100k: "; QS=1599 RVC_KNN_GLDS=$g python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/GB.*//'
  echo -n "glds=$g 2M:   "; N=2000000 QS=1599 RVC_KNN_GLDS=$g python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/GB.*//'
done
