#!/bin/bash
# SQ counters of winobf_conv_kernel (C = 128, K = 11): where do the wave cycles go, how busy are the LDS and the matrix pipe
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export BENCH_C=${BENCH_C:-128} BENCH_K=${BENCH_K:-11}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o c -- python3 $R/tools/bench_convbf.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pd -o d -- python3 $R/tools/bench_convbf.py > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
for path in ("/tmp/pc/c_counter_collection.csv", "/tmp/pd/d_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "winobf_conv_kernel" in r["Kernel_Name"] or "wino_conv_kernel" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0][-40:]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, d in agg.items():
        print(f"{k}  avg {sum(dur[k])/len(dur[k])/1e3:8.1f} us")
        for name, v in sorted(d.items()):
            print(f"    {name:28s} {sum(v)/len(v):.4e}")
PY
