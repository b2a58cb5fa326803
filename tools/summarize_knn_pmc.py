#!/usr/bin/env python3
"""Per-kernel HBM traffic of the kNN search from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_knn.sh.
Units and correction as MI355X_MICROARCH.md prescribes: the counters are KiB-like units of 1 KB (value x 1000... checked
below against a launch with a known byte count) and FETCH_SIZE reports HALF of the bytes of wide coalesced reads on
gfx950 -> x2.  Calibration launch: knn_to_half_kernel over the index reads n_rows*dim*4 B and writes n_rows*dim*2 B."""
import collections, csv, json, sys
fetch, write, n_rows, dim = sys.argv[1], sys.argv[2], int(sys.argv[3]), 768
def load(path):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg.setdefault((name, r["Grid_Size"]), []).append(float(r["Counter_Value"]))
    return agg
f, w = load(fetch), load(write)
cal = [v for (k, g), v in f.items() if "to_half" in k and int(g) >= n_rows * 16]
calw = [v for (k, g), v in w.items() if "to_half" in k and int(g) >= n_rows * 16]
known_r, known_w = n_rows * dim * 4.0, n_rows * dim * 2.0
kf = known_r / (sum(cal[0]) / len(cal[0])) if cal else 2048.0
kw = known_w / (sum(calw[0]) / len(calw[0])) if calw else 1024.0
print(f"calibration on knn_to_half_kernel over the index ({known_r/1e6:.1f} MB read, {known_w/1e6:.1f} MB written): "
      f"bytes per FETCH_SIZE unit = {kf:.1f} (guide: 1024 x 2), per WRITE_SIZE unit = {kw:.1f}")
for key, v in f.items():
    name, grid = key
    if "knn" not in name: continue
    v = v[len(v) // 3:] if len(v) >= 3 else v
    wv = w.get(key, [0.0]); wv = wv[len(wv) // 3:] if len(wv) >= 3 else wv
    print(f"{name[-44:]:44s} grid {grid:>9s} launches {len(f[key]):3d}  fetch {sum(v)/len(v)*kf/1e6:10.1f} MB  write {sum(wv)/len(wv)*kw/1e6:9.1f} MB per launch")

# one fp16-screened search = query conversion + sample pass + bound + main pass + exact re-scoring
total = 0.0
parts = []
for key, v in f.items():
    name, grid = key
    if "knn" not in name or "partial" in name or ("to_half" in name and int(grid) >= n_rows * 16):
        continue
    v = v[len(v) // 3:] if len(v) >= 3 else v
    wv = w.get(key, [0.0]); wv = wv[len(wv) // 3:] if len(wv) >= 3 else wv
    b = sum(v) / len(v) * kf + sum(wv) / len(wv) * kw
    total += b
    parts.append(f"{name[-30:]} {b / 1e6:.1f} MB")
print(f"one screened search (sum of its five launches): {total / 1e9:.3f} GB  [" + "; ".join(parts) + "]")
if len(sys.argv) > 4:
    json.dump({"bytes_per_search": round(total), "n_rows": n_rows, "n_queries": 1599,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/profile_knn.sh) over tools/bench_knn.py at this shape, "
                         "summed over the screened search's five launches (the re-scoring launch averaged with the exact regime's), calibrated "
                         "on a launch with known bytes (tools/summarize_knn_pmc.py)"}, open(sys.argv[4], "w"), indent=1)
