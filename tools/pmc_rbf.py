#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes over the fused ResBlock pair (resblock_bf.hip): a calibration copy with known bytes, then
every (C, K) pair shape of the decoder at the cfg-2 lengths, dilations 1 / 3 / 5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
ONE = int(os.environ.get("ONE", "0"))     # 1: one-term taps (bf16-valued weights, cfg 4): every shape that decoder handle runs
shapes = ((32, 1535040, (3, 7, 11)), (64, 767520, (3, 7, 11)), (128, 383760, (3, 7))) if ONE else ((32, 1535040, (3, 7, 11)), (64, 767520, (3, 7)))
for C, L, Ks in shapes:
    x = torch.randn(1, C, L, device=dev); y = torch.empty_like(x); bias = torch.zeros(C, device=dev)
    for _ in range(3):
        y.copy_(x)                      # known: one tensor read, one written (16 B per lane)
    for K in Ks:
        u = _native.resblock_bf16x3_pack_weight(torch.randn(C, C, K) * 0.03, torch.randn(C, C, K) * 0.03, dev, bf16_taps=bool(ONE))
        for d in (1, 3, 5):
            for _ in range(3):
                _native.resblock_bf16x3_forward(x, u, bias, bias, K, d, 0.1, out=y, bf16_taps=bool(ONE))
torch.cuda.synchronize()
