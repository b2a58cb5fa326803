#!/bin/bash
# fraction of the timed region during which at least one kernel is running (union of kernel intervals), and how many overlap
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/gb
rocprofv3 --kernel-trace --output-format csv -d /tmp/gb -o t -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-rooflines --inflight ${INFLIGHT:-2} > /tmp/gb.log 2>&1
tail -1 /tmp/gb.log | cut -c1-200
python3 - <<'PY'
import csv
ev = []
rows = list(csv.DictReader(open("/tmp/gb/t_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last 60 % of the trace (steady state)
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
lo = t0 + (t1 - t0) * 0.5
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if int(r["Start_Timestamp"]) >= lo]
iv.sort()
busy = 0; cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = iv[-1][1] - iv[0][0] if iv else 1
tot = sum(e - s for s, e in iv)
print(f"steady-state window {span/1e6:.1f} ms: some kernel running {busy/span:.3f} of the time; sum of kernel durations / window = {tot/span:.2f}; {len(iv)} kernels, {len(iv)/(span/1e6):.0f} per ms")
gaps = []
PY
