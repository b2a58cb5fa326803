#!/usr/bin/env python3
"""kNN kernel at several query counts: time, index GB/s (one pass of the index per query tile), fp32 TFLOP/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
N = int(os.environ.get("N", 100000))
index = torch.randn(N, 768, device=dev) * 0.35
norms = _native.knn_index_norms(index)
for Q in (1, 8, 32, 64, 128, 256, 1599):
    q = index[torch.randint(0, N, (Q,), device=dev)] + 0.03 * torch.randn(Q, 768, device=dev)
    for _ in range(3): _native.knn_search(index, norms, q)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): _native.knn_search(index, norms, q)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    passes = -(-Q // 128)
    print(f"Q={Q:5d}: {t*1e6:8.1f} us  index stream {passes*N*3072/t/1e9:8.1f} GB/s  ({N*3072/t/1e9:7.1f} GB/s single pass)  {2.0*Q*N*768/t/1e12:6.2f} TF/s")
