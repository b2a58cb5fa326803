#!/usr/bin/env python3
"""Where does the fused pair differ from two direct-form launches?  C K D L from the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from rvc_amd import _native
dev = "cuda:0"
C, K, D, L = (int(os.environ.get(k, v)) for k, v in (("C", 64), ("K", 7), ("D", 1), ("L", 767520)))
g = torch.Generator().manual_seed(C * 1000 + K * 10 + D)
x = torch.randn(1, C, L, generator=g).to(dev)
w1 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5; w2 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
u = _native.resblock_bf16x3_pack_weight(w1, w2, dev)
pw1, pw2 = _native.conv1d_pack_weight(w1, dev), _native.conv1d_pack_weight(w2, dev)
t32 = _native.conv1d_forward(x, pw1, None, C, K, D, 0.1)
ref = _native.conv1d_forward(t32, pw2, None, C, K, 1, 0.1, res=x)
N1 = 256 if C == 32 else 128
BN = (N1 - (K - 1)) // 4 * 4
keep = []
for rep in range(4):
    if os.environ.get("REUSE") and rep > 0:
        got = _native.resblock_bf16x3_forward(x, u, None, None, K, D, 0.1, out=got)
    else:
        got = _native.resblock_bf16x3_forward(x, u, None, None, K, D, 0.1)
    if os.environ.get("KEEP"): keep.append(got)
    torch.cuda.synchronize()
    diff = (got - ref).abs()[0]
    bad = (diff > 1e-4).nonzero().cpu().numpy()
    print(f"rep {rep}: max diff {diff.max().item():.3e}, {len(bad)} elements above 1e-4")
    if len(bad):
        ch, t = bad[:, 0], bad[:, 1]
        tile, col = t // BN, t % BN
        print("  channels:", np.unique(ch)[:40], "...")
        print("  tiles (first 20):", np.unique(tile)[:20], " n tiles:", len(np.unique(tile)), "of", -(-L // BN))
        print("  columns in tile: min", col.min(), "max", col.max(), " hist by 32:", np.bincount(col // 32))
        n_t = -(-L // BN); per = -(-n_t // 8)
        ut = np.unique(tile)
        print("  tile position inside its XCD range mod 32 (= block slot):", np.bincount((ut % per) % 32, minlength=32))
        print("  iteration of its block:", np.bincount((ut % per) // 32))
        t0 = int(ut[0]) * BN
        blk = diff[:, t0:t0 + BN].cpu().numpy()
        np.set_printoptions(linewidth=250, precision=1)
        print("  first bad tile", ut[0], ": per-column max diff:", blk.max(0))
        print("  per-channel max diff:", blk.max(1))
        print("  previous tile max", diff[:, t0 - BN:t0].max().item(), " next tile max", diff[:, t0 + BN:t0 + 2 * BN].max().item())
        v = diff[ch[0], t[0] - 2: t[0] + 3].cpu().numpy()
        print("  sample diffs around first bad element:", v, "ref", ref[0, ch[0], t[0]].item())
