cd $GRAFT_REPO_ROOT
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
for g in 1 0; do
  echo -n "glds=$g 100k: "; QS=1599 RVC_KNN_GLDS=$g python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/GB.*//'
  echo -n "glds=$g 2M:   "; N=2000000 QS=1599 RVC_KNN_GLDS=$g python tools/bench_knn.py 2>/dev/null | sed 's/.*screened *//; s/GB.*//'
done
