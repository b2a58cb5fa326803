#!/usr/bin/env python3
"""Fresh-buffer stress of the fused ResBlock pair (resblock_bf.hip, K3f) -- the regime of profiles/r05_rbf_notes.txt item 5 (a form of the
stager prologue returned the SECOND tile of a few blocks wrong on the first launches into fresh output buffers; the order that
shipped was never seen to fail, the cause was never isolated).  This process loads the library and, as its first GPU work, launches
every benchmarked shape `N` times, each time into ANOTHER output buffer that no kernel has written before (the first half untouched
hipMalloc memory, the second half NaN-poisoned so that an element the kernel leaves out shows); every output must equal the first
BIT FOR BIT, contain no NaN, and the first must agree with two launches of the fp32 direct-form kernel (conv.hip).
Form "direct": K3d (convbf1.hip), the single conv with one-term taps that shares K3f's stager design, at C = 128 / 256.
usage: stress_rbf.py C [one|three|direct] [N]   (exit code 1 on any difference)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native

dev = "cuda:0"
C = int(sys.argv[1]); form = sys.argv[2] if len(sys.argv) > 2 else "three"; one = form == "one"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 50
L = {32: 1535040, 64: 767520, 128: 383760, 256: 38376}[C]
taps = [k for k in (3, 7, 11) if not (C == 128 and k == 11)]
if form == "three":   # the shapes the fp32-weight decoder takes (resblock_bf_preferred)
    taps = [k for k in taps if C == 32 or (C == 64 and k != 11) or (C == 128 and k == 3)]
bad = 0
if form == "direct":
    g = torch.Generator().manual_seed(C)
    x = torch.randn(1, C, L, generator=g).to(dev); res = torch.randn(1, C, L, generator=g).to(dev)
    for K in ((11,) if C == 128 else (3, 7, 11)):
        w = (torch.randn(C, C, K, generator=g) / (C * K) ** 0.5).bfloat16().float()
        b = torch.randn(C, generator=g).to(dev)
        u = _native.conv1d_bf16w_pack_weight(w, dev)
        for dil in (1, 3, 5):
            torch.cuda.empty_cache()
            outs = [torch.empty(1, C, L, device=dev) for _ in range(N)]
            for o in outs[N // 2:]: o.fill_(float("nan"))
            torch.cuda.synchronize()
            for o in outs: _native.conv1d_bf16w_forward(x, u, b, K, dil, 0.1, res=res, out=o)
            torch.cuda.synchronize()
            n_diff = sum(int((o != outs[0]).any().item()) for o in outs[1:])
            n_nan = int(torch.isnan(outs[0]).any().item())
            ref = _native.conv1d_forward(x, _native.conv1d_pack_weight(w, dev), b, C, K, dil, 0.1, res=res)
            err = (outs[0] - ref).abs().max().item()
            ok = n_diff == 0 and n_nan == 0 and err <= 1e-4
            bad += not ok
            print(f"C={C} K={K} d={dil} direct one-term conv: {N} fresh buffers, {n_diff} differ from the first, NaN {n_nan}, "
                  f"max |first - fp32 direct conv| {err:.2e} {'ok' if ok else 'FAILED'}", flush=True)
            del outs
    sys.exit(1 if bad else 0)
g = torch.Generator().manual_seed(C)
x = torch.randn(1, C, L, generator=g).to(dev)
for K in taps:
    w1 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5; w2 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
    if one: w1, w2 = w1.bfloat16().float(), w2.bfloat16().float()
    b1 = torch.randn(C, generator=g).to(dev); b2 = torch.randn(C, generator=g).to(dev)
    u = _native.resblock_bf16x3_pack_weight(w1, w2, dev, bf16_taps=one)
    for dil in (1, 3, 5):
        torch.cuda.empty_cache()                          # the buffers below come from hipMalloc, not from torch's free list
        outs = [torch.empty(1, C, L, device=dev) for _ in range(N)]
        for o in outs[N // 2:]: o.fill_(float("nan"))
        torch.cuda.synchronize()
        for o in outs: _native.resblock_bf16x3_forward(x, u, b1, b2, K, dil, 0.1, out=o, bf16_taps=one)
        torch.cuda.synchronize()
        n_diff = sum(int((o != outs[0]).any().item()) for o in outs[1:])      # (NaN != NaN: a NaN anywhere counts)
        n_nan = int(torch.isnan(outs[0]).any().item())
        t = _native.conv1d_forward(x, _native.conv1d_pack_weight(w1, dev), b1, C, K, dil, 0.1)
        ref = _native.conv1d_forward(t, _native.conv1d_pack_weight(w2, dev), b2, C, K, 1, 0.1, res=x)
        err = (outs[0] - ref).abs().max().item()
        ok = n_diff == 0 and n_nan == 0 and err <= 1e-4
        bad += not ok
        print(f"C={C} K={K} d={dil} {'one-term' if one else 'three-term'} taps: {N} fresh buffers, {n_diff} differ from the first, NaN {n_nan}, "
              f"max |first - fp32 direct pair| {err:.2e} {'ok' if ok else 'FAILED'}", flush=True)
        del outs
sys.exit(1 if bad else 0)
