"""K10b soak: 300 random (channels, row length, height, batch) draws per seed against F.conv2d in float64 (bias + ReLU + skip path);
usage: python tools/soak_conv2dbf.py [seed].  Round 6: seeds 1-3 clean, worst absolute error 4.3e-6 on unit-variance data."""
import os, sys, ctypes
sys.path.insert(0, "codename-rvc-fork-3_amd")
import numpy as np, torch, torch.nn.functional as F
from rvc_amd import _native as N
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
worst = 0.0
for it in range(300):
    c_out = int(rng.choice([1, 3, 7, 16, 24, 32, 48, 64, 128, 256, 384, 512]))
    c_in = int(rng.choice([16, 32, 48, 64, 96, 128, 256, 512]))
    w = int(rng.choice([4, 8, 16, 32, 64, 128]))
    batch = int(rng.integers(1, 4))
    h = int(rng.integers(1, max(2, min(1200, 400_000 // (w * max(c_in, c_out) * batch)))))
    if not N.conv2d_bf16x3_packable(c_in, c_out) or not N.conv2d_bf16x3_supported(c_in, c_out, h, w):
        continue
    g = torch.Generator().manual_seed(5000 + it)
    x = torch.randn(batch, c_in, h, w, generator=g)
    wt = torch.randn(c_out, c_in, 3, 3, generator=g) / (c_in * 9) ** 0.5
    b = torch.randn(c_out, generator=g); res = torch.randn(batch, c_out, h, w, generator=g)
    ref = F.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1)) + res.double()
    u = N.conv2d_bf16x3_pack_weight(wt, dev)
    got = N.conv2d_bf16x3_forward(x.to(dev), u, b.to(dev), c_out, relu=True, res=res.to(dev)).cpu()
    err = (got.double() - ref).abs().max().item()
    worst = max(worst, err)
    if err > 1.5e-5 or not torch.isfinite(got).all():
        print("BAD", it, c_in, c_out, h, w, batch, err); sys.exit(1)
print("300 draws ok, worst abs err", worst)
