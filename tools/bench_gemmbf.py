#!/usr/bin/env python3
"""HuBERT's GEMM-shaped layers at the cfg-2 sizes: torch fp32 (hipBLASLt / MIOpen) vs gemmbf.hip (exact bf16x3 splits on the
bf16 matrix cores).  HIP events, median of 5 batches of 20 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch, torch.nn.functional as F
from rvc_amd import _native
dev = "cuda:0"
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timed(fn, batches=5, reps=20):
    for _ in range(3): fn()
    out = []
    for _ in range(batches):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]

T = 1599
for name, k, m, act in (("qkv", 768, 2304, "none"), ("out_proj", 768, 768, "none"), ("ff1+gelu", 768, 3072, "gelu"), ("ff2", 3072, 768, "none")):
    x = torch.randn(T, k, device=dev); w = torch.randn(m, k, device=dev) * k ** -0.5; b = torch.randn(m, device=dev)
    a = _native.gemm_bf16x3_pack_weight(w, dev)
    lib = timed(lambda: F.gelu(F.linear(x, w, b)) if act == "gelu" else F.linear(x, w, b))
    own = timed(lambda: _native.linear_bf16x3(x, a, b, m, act=act))
    gf = 2.0 * T * k * m / 1e9
    print(f"linear {name:9s} {T} x {k:4d} -> {m:4d}: torch fp32 {lib*1e3:7.1f} us ({gf/lib:6.1f} TF/s) | bf16x3 {own*1e3:7.1f} us ({gf/own:6.1f} TF/s fp32-equivalent, "
          f"{6*gf/own:7.1f} TF/s on the bf16 pipe) x{lib/own:.2f}", flush=True)
L = 102399
for i, (k, s) in enumerate(((3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2))):
    x = torch.randn(1, 512, L, device=dev); w = torch.randn(512, 512, k, device=dev) * (512 * k) ** -0.5
    a = _native.gemm_bf16x3_pack_weight(w, dev)
    lib = timed(lambda: F.gelu(F.conv1d(x, w, None, stride=s)))
    own = timed(lambda: _native.conv1d_bf16x3(x, a, None, 512, k, stride=s, act="gelu"))
    Lo = (L - k) // s + 1
    gf = 2.0 * 512 * 512 * k * Lo / 1e9
    print(f"conv layer {i + 1} 512->512 k {k} s {s} L {L:6d}: torch fp32 conv + gelu {lib*1e3:7.1f} us ({gf/lib:6.1f} TF/s) | bf16x3 {own*1e3:7.1f} us ({gf/own:6.1f} TF/s fp32-equivalent) x{lib/own:.2f}", flush=True)
    L = Lo
