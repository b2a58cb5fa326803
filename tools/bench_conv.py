#!/usr/bin/env python3
"""Time the 12 ResBlock conv shapes of the 48 kHz vocoder at the cfg-2 lengths (HIP events, 10 reps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
tot = 0.0
print("env", {k: v for k, v in os.environ.items() if k.startswith("RVC_")})
only = int(os.environ.get("BENCH_C", "0"))
for C, L in ((256, 38376), (128, 383760), (64, 767520), (32, 1535040)):
    if only and C != only: continue
    x = torch.randn(1, C, L, device=dev); res = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
    for K in tuple(int(k) for k in os.environ.get("BENCH_K", "3,7,11").split(",")):
        w = _native.conv1d_pack_weight(torch.randn(C, C, K) * 0.03, dev)
        for dil in (1, 5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, batches=5, reps=12):   # median batch: single batches of a 1 ms kernel scatter by +-10 % with the clocks
                for _ in range(3): fn()
                out = []
                for _ in range(batches):
                    e0.record()
                    for _ in range(reps): fn()
                    e1.record(); torch.cuda.synchronize()
                    out.append(e0.elapsed_time(e1) / reps)
                return sorted(out)[len(out) // 2]

            ms = timed(lambda: _native.conv1d_forward(x, w, bias, C, K, dil, 0.1, res=res))
            gf = 2.0 * C * C * K * L / 1e9
            u = _native.conv1d_wino_pack_weight(torch.randn(C, C, K) * 0.03, dev)
            msw = timed(lambda: _native.conv1d_wino_forward(x, u, bias, C, K, dil, 0.1, res=res))
            print(f"C={C:3d} K={K:2d} d={dil} L={L:8d}: direct {ms*1e3:8.1f} us {gf/ms:7.1f} TF/s | winograd {msw*1e3:8.1f} us {gf/msw:7.1f} TF/s-equivalent  x{ms/msw:.2f}")
            tot += ms * 3  # 3 dilations ~ (1, 3, 5) and the k-tap second conv (d=1) -> 6 convs per (C,K); d=1 and d=5 sampled
print(f"estimated resblock total per utterance: {tot:.1f} ms")
