#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over K3d (convbf1.hip: one conv with one-term taps, direct form): every (C, K) the
bf16-storage decoder runs on it at the cfg-2 lengths, dilations 1 / 3 / 5, with the residual of a ResBlock's second conv."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
for C, L, Ks in ((256, 38376, (3, 7, 11)), (128, 383760, (11,))):
    x = torch.randn(1, C, L, device=dev); y = torch.empty_like(x); res = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
    for _ in range(3):
        y.copy_(x)                      # known: one tensor read, one written (16 B per lane)
    for K in Ks:
        u = _native.conv1d_bf16w_pack_weight(torch.randn(C, C, K) * 0.03, dev)
        for d in (1, 3, 5):
            for _ in range(3):
                _native.conv1d_bf16w_forward(x, u, bias, K, d, 0.1, res=res, out=y)
torch.cuda.synchronize()
