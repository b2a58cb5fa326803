#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_rbf.py -> HBM bytes per launch of resblock_bf_kernel<K, C>, calibrated on the tensor copy
in front of each channel count (16 bytes per lane both ways, like the kernel's own row accesses are dword / 16-byte mixes: the
read figure is therefore given with both the copy's factor and the guide's 4-byte-per-lane factor of 1).  usage: F.csv W.csv [out.json]"""
import csv, json, sys, collections, statistics as st
F, W = (list(csv.DictReader(open(p))) for p in sys.argv[1:3])
def by_kernel(rows):
    d = collections.OrderedDict()
    for r in rows:
        d.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]) * 1024.0)
    return d
f, w = by_kernel(F), by_kernel(W)
out = {}
for name in f:
    if "resblock_bf_kernel" not in name: continue
    K, C = (int(v) for v in name.split("<")[1].split(">")[0].split(",")[:2])
    L = {32: 1535040, 64: 767520, 128: 383760}[C]
    tensor = C * L * 4
    fr, wr = st.mean(f[name]), st.mean(w.get(name, [0.0]))
    print(f"{name:44s} launches {len(f[name]):3d}: FETCH_SIZE raw {fr/1e6:7.1f} MB (x2 for 16-byte reads: {2*fr/1e6:7.1f}), WRITE_SIZE {wr/1e6:7.1f} MB; "
          f"algorithmic {tensor/1e6:.1f} read + {tensor/1e6:.1f} written; the two launches it replaces move {5*tensor/1e6:.0f} MB")
    out[f"K{K}_C{C}"] = {"fetch_raw": round(fr), "write": round(wr), "tensor_bytes": tensor}
copies = [k for k in f if "copy" in k.lower() or "elementwise" in k.lower()]
for k in copies[:2]:
    print(f"calibration {k[:60]}: FETCH_SIZE raw {st.mean(f[k])/1e6:.1f} MB, WRITE_SIZE {st.mean(w.get(k, [0]))/1e6:.1f} MB (a 196.5 MB tensor copy)")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
