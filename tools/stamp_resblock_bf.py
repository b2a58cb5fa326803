#!/usr/bin/env python3
"""Per-phase cycle breakdown of resblock_bf_kernel (ablation build, RVC_RBF_DBG=128: compute wave 0 and stager wave 4 stamp s_memtime
eight times per tile).  C, K, D, L from the environment (default 32, 7, 1, 1535040)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
os.environ.setdefault("RVC_AMD_LIB", os.path.join(ROOT, "codename-rvc-fork-3_amd", "rvc_amd", "_lib", "librvc_amd_ablate.so"))
os.environ["RVC_RBF_DBG"] = "128"
import numpy as np, torch
from rvc_amd import _native
dev = "cuda:0"
C, K, D = int(os.environ.get("C", 32)), int(os.environ.get("K", 7)), int(os.environ.get("D", 1))
L = int(os.environ.get("L", 1535040 if C == 32 else 767520))
x = torch.randn(1, C, L, device=dev); bias = torch.zeros(C, device=dev); y = torch.empty_like(x)
u = _native.resblock_bf16x3_pack_weight(torch.randn(C, C, K) * 0.03, torch.randn(C, C, K) * 0.03, dev)
stamps = torch.zeros(1, C, L, device=dev)          # the kernel writes [block][2][64] uint64 into it
for _ in range(3):
    _native.resblock_bf16x3_forward(x, u, bias, bias, K, D, 0.1, acc=stamps, out=y)
stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_native.resblock_bf16x3_forward(x, u, bias, bias, K, D, 0.1, acc=stamps, out=y)
e1.record()
torch.cuda.synchronize()
raw = stamps.view(-1).view(torch.int64).cpu().numpy()
nb = 256
s = raw[: nb * 128].reshape(nb, 2, 64).astype(np.float64)
cw, st = s[:, 0].reshape(nb, 8, 8), s[:, 1].reshape(nb, 8, 8)     # [block][tile][stamp]
ok = cw[:, 7, 7] > 0                                               # blocks that ran at least eight tiles
cw, st = cw[ok], st[ok]
m = lambda a: float(np.mean(a))
print(f"C = {C}, K = {K}, d = {D}, L = {L}: {e0.elapsed_time(e1) * 1e3:.1f} us for the stamped launch; {ok.sum()} blocks, tiles 1..6 of each, mean cycles (s_memtime ticks)")
t = slice(1, 7)
per_tile = m(cw[:, 2:8, 0] - cw[:, 1:7, 0])
print(f"  one tile, barrier A to barrier A                {per_tile:8.0f}")
print(f"  compute wave 0: waits at barrier A              {m(cw[:, t, 1] - cw[:, t, 0]):8.0f}")
print(f"    io swap (residual in, previous outputs out)   {m(cw[:, t, 2] - cw[:, t, 1]):8.0f}")
print(f"    waits at barrier D                            {m(cw[:, t, 3] - cw[:, t, 2]):8.0f}")
print(f"    conv1 (matrix loop)                           {m(cw[:, t, 4] - cw[:, t, 3]):8.0f}")
print(f"    conv1 epilogue (bias, leaky, split, LDS)      {m(cw[:, t, 5] - cw[:, t, 4]):8.0f}")
print(f"    waits at barrier B                            {m(cw[:, t, 6] - cw[:, t, 5]):8.0f}")
print(f"    conv2 (matrix loop)                           {m(cw[:, t, 7] - cw[:, t, 6]):8.0f}")
print(f"    conv2 epilogue (bias, residual)               {m(cw[:, 2:8, 0] - cw[:, 1:7, 7]):8.0f}")
print(f"  stager wave 4: arrives at barrier A             {m(st[:, t, 0] - cw[:, t, 0]):8.0f} after compute wave 0")
print(f"    barrier D, stores the previous tile, requests {m(st[:, t, 2] - st[:, t, 1]):8.0f}")
print(f"    waits at barrier B                            {m(st[:, t, 3] - st[:, t, 2]):8.0f}")
print(f"    waits for every outstanding load / store      {m(st[:, t, 4] - st[:, t, 3]):8.0f}")
print(f"    x tile split and written                      {m(st[:, t, 5] - st[:, t, 4]):8.0f}")
