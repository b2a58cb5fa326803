#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_conv.py -> calibrated HBM bytes per launch of the roofline kernel."""
import csv, sys, statistics as st
fdir, wdir = sys.argv[1], sys.argv[2]
def load(path):
    return list(csv.DictReader(open(path)))
F, W = load(fdir), load(wdir)
def sel(rows, pred):
    return [float(r["Counter_Value"]) for r in rows if pred(r)]
KiB = 1024.0
out = []
cal = {}
for C, L in ((128, 383760),):
    tensor = C * L * 4
    grid = str(((L + 127) // 128) * (C // 128) * 256)
    k1 = lambda r: "conv_mfma_kernel<1," in r["Kernel_Name"] and r["Grid_Size"] == grid
    f1, w1 = st.median(sel(F, k1)) * KiB, st.median(sel(W, k1)) * KiB
    cal[C] = f1 / (2 * tensor)
    out.append(f"calibration K=1 C={C} L={L}: known reads {2*tensor/1e6:.1f} MB -> FETCH_SIZE {f1/1e6:.1f} MB (factor {cal[C]:.3f}); "
               f"known writes {tensor/1e6:.1f} MB -> WRITE_SIZE {w1/1e6:.1f} MB (factor {w1/tensor:.3f})")
k11 = lambda r: "conv_mfma_kernel<11, 2, 2, 2, 2, 4, false>" in r["Kernel_Name"]
f = [ (float(r["Counter_Value"]) * KiB, r["Grid_Size"]) for r in F if k11(r)]
w = [ float(r["Counter_Value"]) * KiB for r in W if k11(r)]
# FETCH_SIZE is calibrated on the same kernel family and tensor size (196 MB, past the 256 MB Infinity Cache once x, res
# and y are counted) with known byte counts, as MI355X_MICROARCH.md prescribes
reads = sum(v / cal[128] for v, g in f) / len(f)
writes = sum(w) / len(w)
out.append(f"roofline kernel conv_mfma_kernel<11,2,2,2,2,4,false>: {len(f)} launches profiled; mean FETCH_SIZE raw {sum(v for v,_ in f)/len(f)/1e6:.1f} MB, "
           f"calibrated reads {reads/1e6:.1f} MB, writes {writes/1e6:.1f} MB per launch")
out.append(f"TRAFFIC_BYTES_PER_LAUNCH {reads + writes:.0f}")
print("\n".join(out))
