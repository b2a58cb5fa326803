#!/usr/bin/env python3
"""Time of one fp16-screened search (1599 queries x N rows x 768) under the RVC_KNN_DBG ablation set in the environment;
prints the main-pass-dominated search time.  Results are wrong under a non-zero RVC_KNN_DBG."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
os.environ.setdefault("RVC_AMD_LIB", os.path.join(ROOT, "codename-rvc-fork-3_amd", "rvc_amd", "_lib", "librvc_amd_ablate.so"))   # RVC_KNN_DBG exists only there
import torch
from rvc_amd import _native
dev = "cuda:0"
N = int(os.environ.get("N", 100000)); Q = 1599
g = torch.Generator(device=dev).manual_seed(0)
centres = torch.randn(512, 768, device=dev, generator=g) * 0.35
index = torch.empty(N, 768, device=dev)
for s in range(0, N, 1 << 18):
    e = min(N, s + (1 << 18))
    index[s:e] = centres[torch.randint(0, 512, (e - s,), device=dev, generator=g)] + 0.05 * torch.randn(e - s, 768, device=dev, generator=g)
aux = _native.knn_index_build(index)
q = torch.randn(Q, 768, device=dev, generator=g)
q[: Q // 2] = index[torch.randint(0, N, (Q // 2,), device=dev, generator=g)] + 0.03 * torch.randn(Q // 2, 768, device=dev, generator=g)
_native.knn_set_mode(2)
for _ in range(2): out = _native.knn_search(index, aux, q)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10 if N <= 200000 else 4
ts = []
for _ in range(3):
    e0.record()
    for _ in range(reps): out = _native.knn_search(index, aux, q)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
t = sorted(ts)[1]
print(f"N={N} dbg={os.environ.get('RVC_KNN_DBG', '0')} glds={os.environ.get('RVC_KNN_GLDS', '0')}: {t:.3f} ms per search ({2.0 * Q * N * 768 / t / 1e9:.0f} TF fp16-equivalent)")
