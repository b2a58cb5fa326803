#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_conv.py -> calibrated HBM bytes per launch of the roofline kernel
(wino_conv_kernel<11,2,2,8,0>, the 6-launch stage-1 mix)."""
import csv, sys, statistics as st
F, W = (list(csv.DictReader(open(p))) for p in sys.argv[1:3])
KiB = 1024.0
C, L = 128, 383760
tensor = C * L * 4
def vals(rows, pred):
    return [float(r["Counter_Value"]) * KiB for r in rows if pred(r)]
out = []
# 4-byte-per-lane loads: the direct kernel at K = 1 reads x and res once each (known 2 tensors)
grid = str(((L + 127) // 128) * (C // 128) * 256)
k1 = lambda r: "conv_mfma_kernel<1," in r["Kernel_Name"] and r["Grid_Size"] == grid
f1, w1 = st.median(vals(F, k1)), st.median(vals(W, k1))
cal4 = f1 / (2 * tensor)
out.append(f"calibration, 4 B/lane loads (direct conv, K=1, x + res): known reads {2*tensor/1e6:.1f} MB -> FETCH_SIZE {f1/1e6:.1f} MB "
           f"(factor {cal4:.3f}); known writes {tensor/1e6:.1f} MB -> WRITE_SIZE {w1/1e6:.1f} MB (factor {w1/tensor:.3f})")
# 16-byte-per-lane loads: torch's copy kernel
cp = lambda r: "elementwise" in r["Kernel_Name"] and "copy" in r["Kernel_Name"].lower()
fc = vals(F, cp)
cal16 = 0.5
if fc:
    m = st.median(fc); cal16 = m / tensor
    out.append(f"calibration, 16 B/lane loads (tensor copy): known reads {tensor/1e6:.1f} MB -> FETCH_SIZE {m/1e6:.1f} MB (factor {cal16:.3f})")
wk = lambda r: "wino_conv_kernel<11, 2, 2, 8, 0>" in r["Kernel_Name"]
f, w = vals(F, wk), vals(W, wk)
n = len(f) // 6
pos_f = [st.mean(f[i::6][-n:]) for i in range(6)]
pos_w = [st.mean(w[i::6][-n:]) for i in range(6)]
names = ["conv1 d=1", "conv2 (+res)", "conv1 d=3", "conv2 (+res)", "conv1 d=5", "conv2 (+res +sum)"]
tot = 0.0
for i in range(6):
    # conv1 launches: x through 4 B/lane buffer loads (+ the tap slab, 16 B/lane LDS-DMA, L2-resident after the first blocks);
    # conv2 launches add the residual (and the running sum) through 16 B/lane loads: raw excess over the d = 1 conv1 launch
    x_raw = pos_f[0] if i % 2 else pos_f[i]
    extra_raw = pos_f[i] - x_raw if i % 2 else 0.0
    reads = x_raw / cal4 + extra_raw / cal16
    tot += reads + pos_w[i]
    out.append(f"  launch {i} {names[i]:18s}: FETCH_SIZE raw {pos_f[i]/1e6:7.1f} MB -> reads {reads/1e6:7.1f} MB; WRITE_SIZE {pos_w[i]/1e6:7.1f} MB")
out.append(f"roofline kernel wino_conv_kernel<11,2,2,8,0>: {len(f)} launches profiled ({n} runs of the 6-launch mix)")
out.append(f"TRAFFIC_BYTES_PER_LAUNCH {tot / 6:.0f}")
print("\n".join(out))
