#!/usr/bin/env python3
"""Time HuBERT's attention (1599 frames, 12 heads x 64) through rvc_attention_qkv_f32: HIP events, median of 5 batches of 12."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
T, H, D = int(os.environ.get("T", 1599)), 12, 64
qkv = torch.randn(1, T, 3 * H * D, device=dev) * 1.5
f = lambda: _native.attention_qkv(qkv, H, D ** -0.5)
for _ in range(5): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
out = []
for _ in range(5):
    e0.record()
    for _ in range(12): f()
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 12 * 1e3)
v = qkv.double().view(1, T, 3, H, D).permute(2, 0, 3, 1, 4)
ref = (torch.softmax(v[0] @ v[1].transpose(-1, -2) * D ** -0.5, -1) @ v[2]).transpose(1, 2).reshape(1, T, -1)
err = (f().double() - ref).abs().max().item()
print(f"attention T={T} H={H} D={D}: {sorted(out)[2]:.1f} us per call (pack + kernel + combine); max abs error vs float64 {err:.2e}")
