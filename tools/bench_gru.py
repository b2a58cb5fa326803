#!/usr/bin/env python3
"""RMVPE BiGRU recurrence (rvc_bigru_forward) at the cfg-2 length: time per call and per step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
T = int(os.environ.get("T", 3232))
g = torch.Generator().manual_seed(0)
gi = (torch.randn(1, T, 2, 768, generator=g) * 0.5).to(dev)
whhT = (torch.randn(2, 256, 768, generator=g) * 0.05).to(dev)
bhh = (torch.randn(2, 768, generator=g) * 0.1).to(dev)
for _ in range(3): out = _native.bigru_forward(gi, whhT, bhh)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); out = _native.bigru_forward(gi, whhT, bhh); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
t = sorted(ts)[2]
print(f"T={T}: {t:.3f} ms per call, {t / T * 1e3:.2f} us per step; redone sequences {_native.bigru_redone(1)}; checksum {out.double().sum().item():.6f}")
