#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_conv.py -> calibrated HBM bytes per launch of the roofline kernel
(winobf2_conv_kernel<11,128,0>: the 12 launches per vocoder forward, stages 0-1 x (conv1 d, conv2) x d = 1, 3, 5).
usage: summarize_pmc.py FETCH.csv WRITE.csv [out.json]  -- the JSON is what bench.py's `roofline.traffic` reads."""
import csv, json, sys, statistics as st
NL = 12
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
cal16 = 0.5   # the guide's factor for 16 B/lane streaming reads (measured 0.500 on a tensor copy in round 1)
out.append(f"calibration, 4 B/lane loads (direct conv, K=1, x + res): known reads {2*tensor/1e6:.1f} MB -> FETCH_SIZE {f1/1e6:.1f} MB "
           f"(factor {cal4:.3f}); known writes {tensor/1e6:.1f} MB -> WRITE_SIZE {w1/1e6:.1f} MB (factor {w1/tensor:.3f}); "
           f"16 B/lane loads: factor {cal16}")
wk = lambda r: "winobf2_conv_kernel<11" in r["Kernel_Name"]
f, w = vals(F, wk), vals(W, wk)
n = len(f) // NL
pos_f = [st.mean(f[i::NL][-n:]) for i in range(NL)]
pos_w = [st.mean(w[i::NL][-n:]) for i in range(NL)]
names = ["conv1 d=1", "conv2 (+res)", "conv1 d=3", "conv2 (+res)", "conv1 d=5", "conv2 (+res +sum)"]
tot = 0.0
for i in range(NL):
    stage, j = divmod(i, 6)
    # conv1 launches: x rows, 256 of a block's <= 336 samples per row by 16 B/lane loads and the rest by 4 B/lane loads (+ the tap
    # fragments, 16 B/lane, L2-resident after the first blocks): blended factor; conv2 launches add the residual (and the
    # running sum) through 16 B/lane loads: raw excess over the stage's d = 1 conv1
    cal_x = 1.0 / (0.76 / cal16 + 0.24 / cal4)
    x_raw = pos_f[stage * 6] if j % 2 else pos_f[i]
    extra_raw = pos_f[i] - x_raw if j % 2 else 0.0
    reads = x_raw / cal_x + extra_raw / cal16
    tot += reads + pos_w[i]
    out.append(f"  stage {stage} launch {j} {names[j]:18s}: FETCH_SIZE raw {pos_f[i]/1e6:7.1f} MB -> reads {reads/1e6:7.1f} MB; WRITE_SIZE {pos_w[i]/1e6:7.1f} MB")
names_seen = sorted({r["Kernel_Name"].split("(")[0] for r in F if wk(r)})
out.append(f"roofline kernel {names_seen}: {len(f)} launches profiled ({n} runs of the {NL}-launch mix)")
out.append(f"TRAFFIC_BYTES_PER_LAUNCH {tot / NL:.0f}")
print("\n".join(out))
if len(sys.argv) > 3:
    json.dump({"bytes_per_launch": round(tot / NL), "kernel": names_seen, "launches_profiled": len(f),
               "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_conv.py (bench.roofline_mix: the 12 "
                         "launches of this symbol in one vocoder forward), average per launch; FETCH_SIZE calibrated per access width on launches "
                         "with known byte counts (tools/summarize_pmc.py, profiles/r04_pmc_winobf2.txt)"}, open(sys.argv[3], "w"), indent=1)
