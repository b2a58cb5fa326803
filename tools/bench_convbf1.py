#!/usr/bin/env python3
"""K3d (convbf1.hip: one conv with bf16-valued taps, direct form, one-term taps) against K3y (winobf2.hip: bf16x3 Winograd on fragments of the
same bf16-valued taps) at the cfg-4 vocoder's 256- and 128-channel shapes: the (dilated conv, conv + residual) pair of one ResBlock
dilation as two launches of either kernel.  HIP events, median of 5 batches of 6 pairs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
for C, L in ((256, 38376), (128, 383760)):
    x = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
    t1 = torch.empty_like(x); y = torch.empty_like(x)
    for K in tuple(int(k) for k in os.environ.get("BENCH_K", "3,7,11").split(",")):
        w1 = (torch.randn(C, C, K) * 0.03).bfloat16().float(); w2 = (torch.randn(C, C, K) * 0.03).bfloat16().float()
        d1, d2 = _native.conv1d_bf16w_pack_weight(w1, dev), _native.conv1d_bf16w_pack_weight(w2, dev)
        u1, u2 = _native.conv1d_winobf_pack_weight(w1, dev), _native.conv1d_winobf_pack_weight(w2, dev)
        for dil in (1, 3, 5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, batches=5, reps=6):
                for _ in range(2): fn()
                out = []
                for _ in range(batches):
                    e0.record()
                    for _ in range(reps): fn()
                    e1.record(); torch.cuda.synchronize()
                    out.append(e0.elapsed_time(e1) / reps)
                return sorted(out)[len(out) // 2]

            def wino():
                _native.conv1d_winobf_forward(x, u1, bias, C, K, dil, 0.1, out=t1)
                _native.conv1d_winobf_forward(t1, u2, bias, C, K, 1, 0.1, res=x, out=y)

            def direct():
                _native.conv1d_bf16w_forward(x, d1, bias, K, dil, 0.1, out=t1)
                _native.conv1d_bf16w_forward(t1, d2, bias, K, 1, 0.1, res=x, out=y)
            msw, msd = timed(wino), timed(direct)
            exe = 2 * 2.0 * C * C * K * L * 3 / 1e9          # bf16 matrix flops executed (direct form, three products)
            print(f"C={C:3d} K={K:2d} d={dil} L={L:7d}: bf16 taps, the pair as two launches: bf16x3 Winograd (K3y) {msw*1e3:7.1f} us | direct one-term (K3d) {msd*1e3:7.1f} us "
                  f"x{msw/msd:.2f} ({exe/msd:6.1f} TF/s on the bf16 pipe = {exe/msd/2500*100:.0f} % of 2.5 PF)", flush=True)
