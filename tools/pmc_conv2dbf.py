#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over K10b (conv2dbf.hip): the C -> C conv of every U-Net level at 3008 frames with bias, ReLU and
the skip path, three launches each, behind a tensor copy of the level-0 map as the calibration (dword loads, like the kernel's staging)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native as N
dev = "cuda:0"
T = 3008
for c, h, w in ((16, T, 128), (32, T // 2, 64), (64, T // 4, 32), (128, T // 8, 16), (256, T // 16, 8), (512, T // 32, 4)):
    x = torch.randn(1, c, h, w, device=dev); res = torch.randn(1, c, h, w, device=dev); y = torch.empty_like(x); b = torch.zeros(c, device=dev)
    u = N.conv2d_bf16x3_pack_weight(torch.randn(c, c, 3, 3) / (c * 9) ** 0.5, dev)
    if c == 16:
        for _ in range(3):
            y.copy_(x)
    for _ in range(3):
        N.conv2d_bf16x3_forward(x, u, b, c, relu=True, res=res, out=y)
torch.cuda.synchronize()
