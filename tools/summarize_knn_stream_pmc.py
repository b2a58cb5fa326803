#!/usr/bin/env python3
"""HBM traffic of one STREAMING-regime kNN search (32 queries: knn_direct_kernel + knn_finalize_kernel) from the rocprofv3 --pmc
FETCH_SIZE / WRITE_SIZE passes over tools/pmc_knn_stream.py.  Calibrated on knn_to_half_kernel over the index (known bytes), which
also absorbs the guide's gfx950 correction (FETCH_SIZE reports half the bytes of wide coalesced reads).
usage: summarize_knn_stream_pmc.py FETCH.csv WRITE.csv N [out.json]"""
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
total, parts = 0.0, []
for key, v in f.items():
    name, grid = key
    if "knn" not in name or ("to_half" in name and int(grid) >= n_rows * 16): continue
    v = v[len(v) // 3:] if len(v) >= 3 else v
    wv = w.get(key, [0.0]); wv = wv[len(wv) // 3:] if len(wv) >= 3 else wv
    fb, wb = sum(v) / len(v) * kf, sum(wv) / len(wv) * kw
    print(f"{name[-44:]:44s} grid {grid:>9s} launches {len(f[key]):3d}  fetch {fb/1e6:10.1f} MB  write {wb/1e6:9.1f} MB per launch")
    total += fb + wb
    parts.append(f"{name[-26:]} {(fb + wb) / 1e6:.1f} MB")
print(f"one streaming search (32 queries): {total / 1e6:.1f} MB against {known_r / 1e6:.1f} MB of fp32 index rows = {total / known_r:.2f} x  [" + "; ".join(parts) + "]")
if len(sys.argv) > 4:
    json.dump({"bytes_per_search": round(total), "n_rows": n_rows, "n_queries": 32,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/pmc_knn_stream.py, summed over the search's "
                         "launches, calibrated on a launch with known bytes (tools/summarize_knn_stream_pmc.py)"}, open(sys.argv[4], "w"), indent=1)
