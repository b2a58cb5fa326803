#!/bin/bash
# SQ counters of the kNN screening GEMM (main pass): where do the wave cycles go, how busy are the LDS and the matrix pipe
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export N=${N:-2000000}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o c -- python3 $R/tools/ablate_knn.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pd -o d -- python3 $R/tools/ablate_knn.py > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d /tmp/pe -o e -- python3 $R/tools/ablate_knn.py > /dev/null 2>&1
python3 - <<'PY'
import csv, collections, os
for path in ("/tmp/pc/c_counter_collection.csv", "/tmp/pd/d_counter_collection.csv", "/tmp/pe/e_counter_collection.csv"):
    if not os.path.exists(path): print("missing", path); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "knn_screen" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-44:] + " grid " + r["Grid_Size"]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, d in agg.items():
        print(f"{k}  avg {sum(dur[k])/len(dur[k])/1e3:8.1f} us")
        for name, v in sorted(d.items()):
            print(f"    {name:28s} {sum(v)/len(v):.4e}")
PY
