#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: calibration launches with known byte counts, then the launch mix that
bench.py's `roofline` object times (bench.roofline_mix: the 12 launches of winobf2_conv_kernel<11,128,0> of one vocoder forward)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
import bench
dev = "cuda:0"
T, rates = 3198, [12, 10, 2, 2]
for C, L in ((128, T * 120),):
    x = torch.randn(1, C, L, device=dev); res = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev)
    y = torch.empty_like(x)
    w1 = _native.conv1d_pack_weight(torch.randn(C, C, 1) * 0.03, dev)
    for _ in range(3):
        y.copy_(x)                                                            # known: 1 tensor read, 1 written (16 B/lane)
        _native.conv1d_forward_into(x, w1, bias, C, 1, 1, 0.1, res=res, out=y)    # known: x + res read (4 B/lane loads), y written
    print("calibration tensor bytes", C, L, x.numel() * 4)
run, flops, launches, alg, executed = bench.roofline_mix(torch, _native, dev, T, rates, "f32")
for _ in range(3):
    run()
torch.cuda.synchronize()
print("mix: flops", flops, "executed", executed, "launches", launches, "algorithmic bytes", alg)
