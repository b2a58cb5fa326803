#!/usr/bin/env python3
"""Time the 3- (C >= 128) / 7- / 11-tap ResBlock conv shapes of the 48 kHz vocoder at the cfg-2 lengths in three forms: direct fp32 MFMA
(conv.hip), fp32 Winograd (wino.hip), bf16x3 Winograd (winobf.hip).  HIP events, median of 5 batches of 12 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
only = int(os.environ.get("BENCH_C", "0"))
for C, L in ((256, 38376), (128, 383760), (64, 767520)):
    if only and C != only: continue
    x = torch.randn(1, C, L, device=dev); res = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
    y = torch.empty_like(x)
    for K in tuple(int(k) for k in os.environ.get("BENCH_K", "3,7,11").split(",")):
        if K == 3 and C % 128: continue          # the 3-tap bf16x3 form exists for 128-row blocks only
        wt = torch.randn(C, C, K) * 0.03
        w = _native.conv1d_pack_weight(wt, dev); u = _native.conv1d_wino_pack_weight(wt, dev); ub = _native.conv1d_winobf_pack_weight(wt, dev)
        for dil in (1, 3, 5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, batches=5, reps=12):
                for _ in range(3): fn()
                out = []
                for _ in range(batches):
                    e0.record()
                    for _ in range(reps): fn()
                    e1.record(); torch.cuda.synchronize()
                    out.append(e0.elapsed_time(e1) / reps)
                return sorted(out)[len(out) // 2]

            msd = timed(lambda: _native.conv1d_forward(x, w, bias, C, K, dil, 0.1, res=res, out=y)) if os.environ.get("BENCH_DIRECT") else float("nan")
            msw = timed(lambda: _native.conv1d_wino_forward(x, u, bias, C, K, dil, 0.1, res=res, out=y))
            msb = timed(lambda: _native.conv1d_winobf_forward(x, ub, bias, C, K, dil, 0.1, res=res, out=y))
            gf = 2.0 * C * C * K * L / 1e9
            G = (K + 3) // 4 if K != 3 else 1
            exe = 2.0 * C * C * (6 if K == 3 else 7) * G * (L / 4) * 6 / 1e9      # bf16 matrix flops executed
            print(f"C={C:3d} K={K:2d} d={dil} L={L:7d}: direct {msd*1e3:7.1f} us | fp32 winograd {msw*1e3:7.1f} us | bf16x3 winograd {msb*1e3:7.1f} us "
                  f"x{msw/msb:.2f} ({gf/msb:6.1f} TF/s algorithmic, {exe/msb:6.1f} TF/s on the bf16 pipe = {exe/msb/2500*100:.0f} % of 2.5 PF)", flush=True)
