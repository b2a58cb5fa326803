#!/bin/bash
# inter-kernel gaps inside one screened kNN search (1599 x 100 k): kernel-trace timeline of tools/bench_knn.py
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kg
export QS=1599
rocprofv3 --kernel-trace --output-format csv -d /tmp/kg -o t -- python3 $R/tools/bench_knn.py > /tmp/kg.log 2>&1
grep -v -e rocprofv3 -e "^[WEI]2026" /tmp/kg.log | tail -15; find /tmp/kg -name "*.csv" | head
F=$(find /tmp/kg -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    for k in ("knn_to_half", "knn_screen_kernel<false", "knn_screen_kernel<true", "knn_select", "knn_finalize", "knn_partial", "knn_direct"):
        if k in n:
            return k
    return n.split("(")[0][-30:]
seq = [(short(r["Kernel_Name"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, (n, s, e) in enumerate(seq) if "knn_to_half" in n]
for i in idx[-4:]:
    chunk = seq[i:i + 6]
    t0 = chunk[0][1]
    print(" | ".join(f"{n} +{(s - t0)/1e3:.0f}..{(e - t0)/1e3:.0f}us" for n, s, e in chunk))
PY
