#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of tools/pmc_hubert_front.py -> HBM bytes per launch of the feature extractor's kernels, next to their
algorithmic bytes (planes: 6 B per value; the last layer writes fp32).  usage: F.csv W.csv"""
import csv, sys, collections, statistics as st
F, W = (list(csv.DictReader(open(p))) for p in sys.argv[1:3])
def seq(rows):
    out = []
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        out.append((r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"]) * 1024.0))
    return out
f, w = seq(F), seq(W)
frames = [(512000 - 10) // 5 + 1]
for k in (3, 3, 3, 3, 2, 2):
    frames.append((frames[-1] - k) // 2 + 1)
names = [n for n, _ in f]
def avg(rows, name, nth, per):        # the nth launch of `name` in each of the three repetitions
    v = [b for n, b in rows if n == name]
    return st.mean(v[nth::per]) if v else float("nan")
for k in (n for n in dict.fromkeys(names) if "copy" in n.lower() or "elementwise" in n.lower()):
    print(f"calibration {k[:50]}: FETCH_SIZE raw {st.mean([b for n, b in f if n == k])/1e6:.1f} MB, WRITE_SIZE {st.mean([b for n, b in w if n == k])/1e6:.1f} MB (a 196.5 MB tensor copy)")
    break
for name in dict.fromkeys(names):
    if "hubert_conv0" in name:
        print(f"{name:40s} FETCH_SIZE raw {avg(f, name, 0, 1)/1e6:8.2f} MB (x2: {2*avg(f, name, 0, 1)/1e6:8.2f}), WRITE_SIZE {avg(w, name, 0, 1)/1e6:8.2f} MB; "
              f"algorithmic: 2.05 MB of samples read" + (f", {frames[0] * 512 * 6 / 1e6:.1f} MB of planes written" if "apply" in name else ""))
lin = [n for n in dict.fromkeys(names) if "linbf_kernel" in n]
order = [(n, b) for n, b in f if "linbf_kernel" in n]
worder = [(n, b) for n, b in w if "linbf_kernel" in n]
for i in range(6):
    fr = st.mean([order[i + 6 * r][1] for r in range(3)]); wr = st.mean([worder[i + 6 * r][1] for r in range(3)])
    k = (3, 3, 3, 3, 2, 2)[i]
    rd = frames[i] * 512 * 6 + 512 * 512 * k * 6; wt = frames[i + 1] * 512 * (6 if i < 5 else 4)
    print(f"layer {i + 1} ({order[i][0]}, {frames[i]} -> {frames[i + 1]} frames): FETCH_SIZE raw {fr/1e6:7.2f} MB (x2 for 16-byte reads: {2*fr/1e6:7.2f}), WRITE_SIZE {wr/1e6:7.2f} MB; "
          f"algorithmic {rd/1e6:.1f} read (planes once + weights) + {wt/1e6:.1f} written")
