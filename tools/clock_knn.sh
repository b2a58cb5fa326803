#!/bin/bash
# the RVC_* switches below exist only in the ablation build of the library (-DRVC_ABLATE, __graft_entry__.build_ablate())
export RVC_AMD_LIB=${RVC_AMD_LIB:-$(cd "$(dirname "$0")/.." && pwd)/codename-rvc-fork-3_amd/rvc_amd/_lib/librvc_amd_ablate.so}
# shader clock during the kNN main pass under each ablation: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel duration
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export N=${N:-2000000}
for d in 0 32 34 48 50 54 40; do
  export RVC_KNN_DBG=$d
  rm -rf /tmp/pg; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pg -o g -- python3 $R/tools/ablate_knn.py > /dev/null 2>&1
  python3 - <<PY
import csv, collections
v = collections.defaultdict(list); dur = []
for r in csv.DictReader(open("/tmp/pg/g_counter_collection.csv")):
    if "knn_screen_kernel<true" in r["Kernel_Name"]:
        v[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
g = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]); t = sum(dur) / len(dur)
m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / max(1, len(v["SQ_VALU_MFMA_BUSY_CYCLES"]))
print(f"RVC_KNN_DBG=$d: main pass {t/1e3:8.1f} us, GRBM_GUI_ACTIVE {g:.4e} -> {g/8/t:.3f} GHz; matrix pipe busy {m/(g/8*1024):.3f} of the SIMD cycles")
PY
done
