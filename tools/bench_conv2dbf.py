"""K10b (conv2dbf.hip) against K10 (conv2d.hip) on the shapes of RMVPE's U-Net for a 30 s clip (3008 frames): us per conv.
usage: python tools/bench_conv2dbf.py [frames]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "codename-rvc-fork-3_amd"))
import torch
from rvc_amd import _native as N

T = int(sys.argv[1]) if len(sys.argv) > 1 else 3008
dev = torch.device("cuda:0")
shapes = [(16, 16, T, 128), (32, 16, T, 128), (32, 32, T // 2, 64), (64, 32, T // 2, 64), (64, 64, T // 4, 32), (128, 64, T // 4, 32),
          (128, 128, T // 8, 16), (256, 128, T // 8, 16), (256, 256, T // 16, 8), (512, 256, T // 16, 8), (256, 512, T // 32, 4),
          (512, 512, T // 32, 4), (16, 3, T, 128)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for c_in, c_out, h, w in shapes:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, c_in, h, w, generator=g).to(dev)
    wt = torch.randn(c_out, c_in, 3, 3, generator=g) / (c_in * 9) ** 0.5
    b = torch.randn(c_out, generator=g).to(dev)
    res = torch.randn(1, c_out, h, w, generator=g).to(dev)
    wp = N.conv2d_pack_weight(wt, dev)
    u = N.conv2d_bf16x3_pack_weight(wt, dev)
    y0 = torch.empty_like(res)
    y1 = torch.empty_like(res)
    t_old = timeit(lambda: N.conv2d_forward(x, wp, b, c_out, 3, relu=True, res=res, out=y0))
    t_new = timeit(lambda: N.conv2d_bf16x3_forward(x, u, b, c_out, relu=True, res=res, out=y1))
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double().to(dev), b.double(), padding=1)) + res.double()
    e0, e1 = (y0.double() - ref).abs().max().item(), (y1.double() - ref).abs().max().item()
    gf = 2 * c_in * c_out * 9 * h * w / 1e9
    print(f"{c_in:4d} -> {c_out:4d}  {h:5d} x {w:3d}  {gf:5.2f} GFLOP   K10 {t_old:7.1f} us ({e0:.1e})   K10b {t_new:7.1f} us ({e1:.1e})   {gf / t_new * 1e3:6.1f} TF fp32-equivalent")
