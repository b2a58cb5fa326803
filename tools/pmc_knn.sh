R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export N=${N:-100000} QS=1599 RVC_KNN_TILE=${RVC_KNN_TILE:-264}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pq -o q -- python3 $R/tools/bench_knn.py > /dev/null 2>&1
python3 - <<'PY'
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("/tmp/pq/q_counter_collection.csv")):
    if "knn_screen_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    wc = sum(d["SQ_WAVE_CYCLES"]) / len(d["SQ_WAVE_CYCLES"])
    for name, v in d.items():
        m = sum(v) / len(v)
        print(f"   {name:28s} {m:14.0f}  ({m / wc:6.3f} of wave cycles)")
PY
