#!/usr/bin/env python3
"""Workload for the rocprofv3 passes over the STREAMING kNN regime (<= 64 queries: knn_direct_kernel makes one pass over the fp32
index per 32-query tile): N (env) rows x 768, 32 queries, 12 searches.  knn_index_build's knn_to_half_kernel over the index (reads
N x 3072 B, writes N x 1536 B) is the calibration launch tools/summarize_knn_stream_pmc.py scales the counters by."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
N = int(os.environ.get("N", 100000))
g = torch.Generator(device=dev).manual_seed(0)
centres = torch.randn(512, 768, device=dev, generator=g) * 0.35
index = torch.empty(N, 768, device=dev)
for s in range(0, N, 1 << 18):
    e = min(N, s + (1 << 18))
    index[s:e] = centres[torch.randint(0, 512, (e - s,), device=dev, generator=g)] + 0.05 * torch.randn(e - s, 768, device=dev, generator=g)
aux = _native.knn_index_build(index)
q = torch.randn(32, 768, device=dev, generator=g)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2): _native.knn_search(index, aux, q)
e0.record()
for _ in range(10): _native.knn_search(index, aux, q)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
print(f"N={N} Q=32 (streaming regime): {t*1e3:.3f} ms per search = {N*3072/t/1e9:.0f} GB/s of fp32 index rows = {N*3072/t/8e12:.3f} of 8 TB/s")
