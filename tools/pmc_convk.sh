R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export BENCH_C=${BENCH_C:-128}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pc -o c -- python3 $R/tools/bench_conv.py > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open("/tmp/pc/c_counter_collection.csv")):
    if "conv_mfma_kernel" in r["Kernel_Name"] or "wino_conv_kernel" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-44:]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in agg.items():
    wc = sum(d["SQ_WAVE_CYCLES"]) / len(d["SQ_WAVE_CYCLES"])
    line = f"{k}  avg {sum(dur[k])/len(dur[k])/1e3:8.1f} us |"
    for name in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"):
        m = sum(d[name]) / len(d[name]); line += f" {name[3:]} {m / wc:5.3f}"
    mf = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
    line += f" | mfma busy cycles/launch {mf:.3e}"
    print(line)
PY
