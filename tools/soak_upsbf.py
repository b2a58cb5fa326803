"""K3u soak: random lengths / batches over every stage shape of the 48 / 40 / 32 k vocoders (and their noise convs) against
F.conv_transpose1d + F.conv1d in float64; usage: python tools/soak_upsbf.py [seed]."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "codename-rvc-fork-3_amd"))
import numpy as np, torch, torch.nn.functional as F
from rvc_amd import _native as N
dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 3)
shapes = [(512, 256, 12, 24, 0, 1), (256, 128, 10, 20, 8, 4), (128, 64, 2, 4, 4, 2), (64, 32, 2, 4, 1, 1), (512, 256, 10, 16, 0, 1),
          (256, 128, 10, 16, 8, 4), (256, 128, 8, 16, 8, 4), (128, 64, 2, 4, 0, 1), (256, 128, 10, 20, 0, 1), (512, 256, 12, 24, 0, 1)]
worst = 0.0
for it in range(120):
    c_in, c_out, rate, ksize, nc_k, nc_stride = shapes[int(rng.integers(0, len(shapes)))]
    batch = int(rng.integers(1, 4))
    length = int(rng.integers(1, max(2, min(6000, 3_000_000 // (c_out * rate * batch)))))
    g = torch.Generator().manual_seed(9000 + it)
    pad = (ksize - rate) // 2
    x = torch.randn(batch, c_in, length, generator=g)
    w = torch.randn(c_in, c_out, ksize, generator=g) / (2 * c_in) ** 0.5
    b = torch.randn(c_out, generator=g)
    l_out = (length - 1) * rate - 2 * pad + ksize
    ref = F.conv_transpose1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), stride=rate, padding=pad)
    har, nw, nc_pad = None, None, 0
    if nc_k:
        nc_pad = 0 if nc_stride == 1 else (nc_k - nc_stride) // 2
        har = torch.randn(batch, l_out * nc_stride, generator=g)
        nw = torch.randn(c_out, 1, nc_k, generator=g) * 0.3
        ref = ref + F.conv1d(har.double()[:, None], nw.double(), None, stride=nc_stride, padding=nc_pad)[:, :, :l_out]
    packed = N.upsample_bf16x3_pack_weight(w, nw, b, rate, nc_stride, dev)
    got = N.upsample_bf16x3_forward(x.to(dev), har.to(dev) if har is not None else None, packed, c_out, rate, ksize, pad, nc_stride, nc_pad, 0.1).cpu()
    err = (got.double() - ref).abs().max().item()
    worst = max(worst, err)
    if got.shape != ref.shape or err > 2e-5 * max(1.0, ref.abs().max().item()) or not torch.isfinite(got).all():
        print("BAD", it, c_in, c_out, rate, ksize, nc_k, nc_stride, length, batch, err); sys.exit(1)
print("120 draws ok, worst abs err", worst)
