#!/usr/bin/env python3
"""Kernel trace of bench.py with two utterances in flight (tools/trace_inflight2.sh) -> how the wall time divides: some whole-CU kernel
(the matrix-core kernels that request the CU's whole LDS, plus the vocoder's fp32 convs) running / only other kernels running /
nothing running; and the same per stream.  usage: analyze_overlap.py t_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
WHOLE = ("winobf2_conv_kernel", "winobf_conv_kernel", "resblock_bf_kernel", "linbf_kernel", "posconv_kernel", "attention_bf_kernel",
         "knn_screen_kernel", "conv_mfma_kernel", "gemmbf_kernel", "wino_conv_kernel", "convbf1_kernel")
# the timed region: from the first nsf_carry after warm-up (3 x 2 utterances) to the last conv_post of the first pass
posts = [r for r in rows if "conv_post_kernel" in r["Kernel_Name"]]
t_lo, t_hi = int(posts[6]["End_Timestamp"]), int(posts[6 + 10 - 1]["End_Timestamp"])
ev = []
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if b <= t_lo or a >= t_hi: continue
    a, b = max(a, t_lo), min(b, t_hi)
    w = any(k in r["Kernel_Name"] for k in WHOLE)
    ev.append((a, 1, w)); ev.append((b, -1, w))
ev.sort()
n_w = n_o = 0; last = t_lo; acc = collections.Counter()
for t, d, w in ev:
    state = "whole-CU kernel running" if n_w else ("only other kernels" if n_o else "nothing running")
    acc[state] += t - last; last = t
    if w: n_w += d
    else: n_o += d
acc["nothing running"] += t_hi - last
tot = t_hi - t_lo
print(f"window {tot/1e6:.2f} ms = 10 utterances, {tot/1e7:.2f} ms per utterance")
for k, v in acc.items(): print(f"  {k:28s} {v/1e6:8.2f} ms  {100*v/tot:5.1f} %")
both = 0
# time with whole-CU kernels of BOTH streams in flight at once (they then share the chip block by block)
ev2 = []
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if b <= t_lo or a >= t_hi or not any(k in r["Kernel_Name"] for k in WHOLE): continue
    ev2.append((max(a, t_lo), 1)); ev2.append((min(b, t_hi), -1))
ev2.sort(); n = 0; last = t_lo; hist = collections.Counter()
for t, d in ev2:
    hist[min(n, 3)] += t - last; last = t; n += d
print("  whole-CU kernels in flight at once:", {k: f"{100*v/tot:.1f} %" for k, v in sorted(hist.items())})
