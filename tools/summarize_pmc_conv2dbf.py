#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_conv2dbf.py -> HBM bytes per launch of conv2dbf_kernel per U-Net level.  usage: F.csv W.csv"""
import csv, sys, statistics as st
def load(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"]) * 1024.0) for r in rows]
f, w = load(sys.argv[1]), load(sys.argv[2])
T = 3008
print("HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in KiB -> bytes).  FETCH_SIZE is given raw and with the guide's gfx950 correction for 4-byte-per-lane")
print("reads (x1.77; the kernel's input staging reads 4 bytes per lane, its skip path 16 -- x2 --, so the truth lies between x1.77 and x2).")
def groups(rows, want):
    out, cur = [], []
    for name, v in rows:
        if want in name:
            cur.append(v)
            if len(cur) == 3: out.append(cur); cur = []
    return out
fc, wc = groups(f, "conv2dbf_kernel"), groups(w, "conv2dbf_kernel")
ff, wf = groups(f, "conv2d_finish"), groups(w, "conv2d_finish")
cp_f = [v for n, v in f if "copyBuffer" in n]; cp_w = [v for n, v in w if "copyBuffer" in n]
if cp_f: print(f"calibration: a 24.6 MB tensor copy (16 bytes per lane): FETCH_SIZE raw {st.mean(cp_f)/1e6:.1f} MB (x2: {2*st.mean(cp_f)/1e6:.1f}), WRITE_SIZE {st.mean(cp_w)/1e6:.1f} MB")
fi = 0
for lvl, (a, b) in enumerate(zip(fc, wc)):
    c = 16 << lvl
    mb = c * (T >> lvl) * (128 >> lvl) * 4 / 1e6
    taps = max(1, c // 32) * (c // 16) * 9 * 3072 / 1e6
    line = (f"level {lvl} ({c:3d} -> {c:3d}, {T >> lvl:4d} x {128 >> lvl:3d}): FETCH_SIZE raw {st.mean(a)/1e6:6.1f} MB (x1.77: {1.77*st.mean(a)/1e6:6.1f}), WRITE_SIZE {st.mean(b)/1e6:5.1f} MB; "
            f"algorithmic: map {mb:.1f} + skip path {mb:.1f} read, {mb:.1f} written, {taps:.2f} MB of tap fragments")
    if lvl >= 3 and fi < len(ff):
        line += f"; + finish pass FETCH_SIZE raw {st.mean(ff[fi])/1e6:.1f} MB, WRITE_SIZE {st.mean(wf[fi])/1e6:.1f} MB (partial sums written by the conv, read back here)"
        fi += 1
    print(line)
