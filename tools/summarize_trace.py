#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel-trace CSV by (kernel, grid) -> calls, avg us, total ms."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 4 and sys.argv[4] == "lastgap":
    # keep only what ran after the last idle gap > 300 ms (the script sleeps before its final pass)
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 300e6: cut = i
    rows = rows[cut:]
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    if "rvc::" not in name and len(sys.argv) <= 3: continue
    key = (name.split("(")[0][:60] if len(sys.argv) > 3 else name.split("(")[0][-60:], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
tot = 0
for k, v in agg.items():
    lastgap = len(sys.argv) > 4 and sys.argv[4] == "lastgap"
    v2 = v if lastgap else (v[len(v) // 3:] if len(v) >= 3 else v)   # drop the first (warm-up) third
    runs = 1 if lastgap else 3
    print(f"{k[0]:62s} grid {k[1]:>8s},{k[2]:>4s} calls {len(v):4d} avg {sum(v2)/len(v2):9.1f} us  total/run {sum(v2)/len(v2)*len(v)/runs/1e3:8.2f} ms")
    tot += sum(v2) / len(v2) * len(v) / runs / 1e3
print("total per run (ms):", round(tot, 2))
