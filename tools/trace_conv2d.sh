#!/bin/bash
# kernel durations (rocprofv3 --kernel-trace) of tools/bench_conv2d.py: the event timings there include host launch gaps
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c2 -o t -- python3 $R/tools/bench_conv2d.py > /tmp/c2.log 2>&1
python3 - <<'PY'
import csv, collections
d = collections.OrderedDict()
for r in csv.DictReader(open("/tmp/c2/t_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if any(k in n for k in ("conv2d", "miopen", "Cijk", "Im2d", "igemm", "bias_relu", "transpose", "Sp3")):
        d.setdefault((n[:58], r.get("Grid_Size", r.get("Grid_Size_X", "?"))), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print(f"{k[0]:58s} grid {k[1]:>9s} n {len(v):4d} avg {sum(v)/len(v)/1e3:8.1f} us")
PY
