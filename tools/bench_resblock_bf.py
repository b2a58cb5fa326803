#!/usr/bin/env python3
"""Time one ResBlock (dilated conv -> conv + residual) pair of the 48 kHz vocoder's 32- / 64-channel stages at the cfg-2 lengths:
the fused bf16x3 pair (resblock_bf.hip, K3f) against the two launches it replaces (fp32 Winograd / bf16x3 Winograd per conv, or the
fused fp32 layer at C = 32, K = 3).  HIP events, median of 5 batches of 6 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
only = int(os.environ.get("BENCH_C", "0"))
ONE = int(os.environ.get("BENCH_ONE", "0"))
for C, L in ((32, 1535040), (64, 767520), (128, 383760)):
    if only and C != only: continue
    x = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
    t1 = torch.empty_like(x); y = torch.empty_like(x)
    for K in tuple(int(k) for k in os.environ.get("BENCH_K", "3,7,11").split(",")):
        if C == 128 and K == 11: continue
        w1 = torch.randn(C, C, K) * 0.03; w2 = torch.randn(C, C, K) * 0.03
        if ONE: w1, w2 = w1.bfloat16().float(), w2.bfloat16().float()     # BENCH_ONE=1: bf16-valued taps (cfg 4); the two launches run on fragments of them
        up = _native.resblock_bf16x3_pack_weight(w1, w2, dev)
        up1 = _native.resblock_bf16x3_pack_weight(w1, w2, dev, bf16_taps=True) if ONE else None
        bf = (C >= 64 and K != 3) or C >= 128
        pk = _native.conv1d_winobf_pack_weight if bf else _native.conv1d_wino_pack_weight
        fw = _native.conv1d_winobf_forward if bf else _native.conv1d_wino_forward
        u1, u2 = pk(w1, dev), pk(w2, dev)
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

            def two():
                fw(x, u1, bias, C, K, dil, 0.1, out=t1)
                fw(t1, u2, bias, C, K, 1, 0.1, res=x, out=y)
            ms2 = timed(two)
            msf = timed(lambda: _native.resblock_bf16x3_forward(x, up, bias, bias, K, dil, 0.1, out=y))
            exe = 2 * 2.0 * C * C * K * L * 6 / 1e9          # bf16 matrix flops executed (direct form, six products)
            byt = 2 * C * L * 4 / 1e6
            if ONE:
                ms1 = timed(lambda: _native.resblock_bf16x3_forward(x, up1, bias, bias, K, dil, 0.1, out=y, bf16_taps=True))
                print(f"C={C:3d} K={K:2d} d={dil} L={L:7d}: bf16 taps: two launches {ms2*1e3:7.1f} us | three-term pair {msf*1e3:7.1f} us | ONE-TERM pair {ms1*1e3:7.1f} us "
                      f"x{ms2/ms1:.2f} / x{msf/ms1:.2f} ({exe/2/ms1:6.1f} TF/s on the bf16 pipe = {exe/2/ms1/2500*100:.0f} % of 2.5 PF)", flush=True)
                continue
            print(f"C={C:3d} K={K:2d} d={dil} L={L:7d}: two launches ({'bf16x3' if bf else 'fp32'} winograd) {ms2*1e3:7.1f} us | fused bf16x3 pair {msf*1e3:7.1f} us "
                  f"x{ms2/msf:.2f} ({exe/msf:6.1f} TF/s on the bf16 pipe = {exe/msf/2500*100:.0f} % of 2.5 PF; {byt/msf:6.0f} GB/s of x + y)", flush=True)
