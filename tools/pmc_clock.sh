R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export BENCH_C=${BENCH_C:-128}
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pg -o g -- python3 $R/tools/bench_conv.py > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pg/g_counter_collection.csv")):
    if "conv_mfma_kernel" in r["Kernel_Name"] or "resblock_layer" in r["Kernel_Name"] or "wino_conv" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-44:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    t = sum(dur[k]) / len(dur[k]) / 2          # each dispatch appears once per counter
    g = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"]); m = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
    print(f"{k} avg {sum(dur[k])/len(dur[k])/1e3:8.1f} us  GUI_ACTIVE {g:.3e} -> {g/(sum(dur[k])/len(dur[k])):.3f} GHz   MFMA busy/SIMD {m/1024:.3e} = {m/1024/g:.3f} of active cycles")
PY
