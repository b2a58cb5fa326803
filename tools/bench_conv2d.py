#!/usr/bin/env python3
"""RMVPE U-Net conv shapes at the cfg-2 length (3232 frames): native conv2d (K10) vs torch / MIOpen, median batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch, torch.nn.functional as F
from rvc_amd import _native
dev = "cuda:0"
T = int(os.environ.get("T", 3232))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, batches=5, reps=10):
    for _ in range(3): fn()
    out = []
    for _ in range(batches):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[len(out) // 2]


shapes = [(16, 16, T, 128), (32, 32, T // 2, 64), (64, 64, T // 4, 32), (128, 128, T // 8, 16), (256, 256, T // 16, 8),
          (512, 512, T // 32, 4), (256, 512, T // 32, 4), (512, 256, T // 16, 8), (256, 128, T // 8, 16), (128, 64, T // 4, 32),
          (64, 32, T // 2, 64), (32, 16, T, 128)]
only = os.environ.get("SHAPES")
if only:
    shapes = [shapes[int(i)] for i in only.split(",")]
skip_torch = bool(os.environ.get("NO_TORCH"))
for ci, co, h, w in shapes:
    x = torch.randn(1, ci, h, w, device=dev); res = torch.randn(1, co, h, w, device=dev)
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05; b = torch.randn(co, device=dev)
    wp = _native.conv2d_pack_weight(wt, dev)
    y = torch.empty(1, co, h, w, device=dev)
    tn = timed(lambda: _native.conv2d_forward(x, wp, b, co, 3, relu=True, res=res, out=y))
    tt = tn if skip_torch else timed(lambda: _native.bias_relu_add_(F.conv2d(x, wt, None, 1, 1), b, res))
    gf = 2.0 * ci * co * 9 * h * w / 1e9
    print(f"{ci:3d}->{co:3d} {h:5d}x{w:3d}: native {tn*1e3:7.1f} us {gf/tn:6.1f} TF/s | torch conv + bias_relu_add {tt*1e3:7.1f} us  x{tt/tn:.2f}")
