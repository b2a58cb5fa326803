#!/usr/bin/env python3
"""kNN search time by regime at the BASELINE shapes: N (env, default 100000) rows x 768, several query counts.
mode 1 = exact fp32 regimes (GEMM with top-8 lists / streaming), mode 2 = fp16-screened regime.  Clustered synthetic
index (rvc_amd.lib.synthetic.synth_index recipe drawn on the device)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
N = int(os.environ.get("N", 100000))
QS = [int(v) for v in os.environ.get("QS", "32,64,256,599,1599").split(",")]
g = torch.Generator(device=dev).manual_seed(0)
centres = torch.randn(512, 768, device=dev, generator=g) * 0.35
index = torch.empty(N, 768, device=dev)
for s in range(0, N, 1 << 18):
    e = min(N, s + (1 << 18))
    index[s:e] = centres[torch.randint(0, 512, (e - s,), device=dev, generator=g)] + 0.05 * torch.randn(e - s, 768, device=dev, generator=g)
aux = _native.knn_index_build(index)
for Q in QS:
    q = torch.randn(Q, 768, device=dev, generator=g)          # HuBERT-like: unit-variance features, far from the centres
    q[: Q // 2] = index[torch.randint(0, N, (Q // 2,), device=dev, generator=g)] + 0.03 * torch.randn(Q // 2, 768, device=dev, generator=g)
    res = {}
    for mode in (1, 2):
        _native.knn_set_mode(mode)
        for _ in range(2): out = _native.knn_search(index, aux, q)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10 if N <= 200000 else 3
        e0.record()
        for _ in range(reps): out = _native.knn_search(index, aux, q)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / reps * 1e-3, out)
    _native.knn_set_mode(0)
    same = torch.equal(res[1][1][1], res[2][1][1]) and torch.equal(res[1][1][0], res[2][1][0])
    t1, t2 = res[1][0], res[2][0]
    fl = 2.0 * Q * N * 768
    print(f"N={N} Q={Q:5d}: exact {t1*1e3:8.3f} ms ({fl/t1/1e12:6.1f} TF)   screened {t2*1e3:8.3f} ms ({fl/t2/1e12:7.1f} TF fp16-equivalent, "
          f"{-(-Q//256)*N*1536/t2/1e9:7.1f} GB/s of fp16 index per 256-query pass)   identical results: {same}")
