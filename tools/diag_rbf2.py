#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from rvc_amd import _native
dev = "cuda:0"
C, K, D, L, B = (int(os.environ.get(k, v)) for k, v in (("C", 32), ("K", 7), ("D", 1), ("L", 5003), ("B", 1)))
g = torch.Generator().manual_seed(1)
x = torch.randn(B, C, L, generator=g).to(dev); acc = torch.randn(B, C, L, generator=g).to(dev)
w1 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5; w2 = torch.randn(C, C, K, generator=g) / (C * K) ** 0.5
u = _native.resblock_bf16x3_pack_weight(w1, w2, dev)
pw1, pw2 = _native.conv1d_pack_weight(w1, dev), _native.conv1d_pack_weight(w2, dev)
t32 = _native.conv1d_forward(x, pw1, None, C, K, D, 0.1)
ref = _native.conv1d_forward(t32, pw2, None, C, K, 1, 0.1, res=x)
np.set_printoptions(linewidth=220)
for name, kw, want in (("plain", {}, ref), ("acc", dict(acc=acc, out_scale=1 / 3), (ref + acc) / 3)):
    got = _native.resblock_bf16x3_forward(x, u, None, None, K, D, 0.1, **kw)
    torch.cuda.synchronize()
    diff = (got - want).abs()
    bad = (diff > 1e-4).nonzero().cpu().numpy()
    print(f"{name}: max diff {diff.max().item():.3e}, {len(bad)} bad of {diff.numel()}")
    if len(bad):
        b, ch, t = bad[:, 0], bad[:, 1], bad[:, 2]
        print("  batch:", np.unique(b), " channels:", np.unique(ch)[:16], " t mod 4:", np.bincount(t % 4, minlength=4), " t range", t.min(), t.max())
        print("  first bad:", bad[:6].tolist(), " got", got[b[0], ch[0], t[0]].item(), "want", want[b[0], ch[0], t[0]].item(),
              " got/want elsewhere:", got[0, 0, :3].tolist(), want[0, 0, :3].tolist())
