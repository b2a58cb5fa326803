#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over HuBERT's feature extractor on time-major frames (hubert_front.hip K13, linbf.hip K12
with a row stride): a calibration copy with known bytes, then layer 0 and layers 1-6 of a 30 s clip (512 000 samples), three times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native as N
dev = "cuda:0"
x = torch.randn(1, 32, 1535040, device=dev); y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)                          # known: 196.5 MB read, 196.5 MB written (16 B per lane)
wav = torch.randn(512000, device=dev) * 0.3
w0 = torch.randn(512, 1, 10, device=dev) * 0.4
gam = torch.randn(512, device=dev)
ws = [N.gemm_bf16x3_pack_weight((torch.randn(512, 512, k) * (512 * k) ** -0.5).permute(0, 2, 1).reshape(512, -1).contiguous(), dev) for k in (3, 3, 3, 3, 2, 2)]
for _ in range(3):
    xs, n = N.hubert_conv0_frames_bf16x3(wav, w0, gam, gam, 1e-5, stride=5)
    for i, k in enumerate((3, 3, 3, 3, 2, 2)):
        xs, n = N.conv1d_frames_bf16x3(xs, n, ws[i], None, 512, k, 2, "gelu_planes" if i < 5 else "gelu_f32")
torch.cuda.synchronize()
