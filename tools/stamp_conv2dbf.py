"""K10b per-phase timeline (ablation build, RVC_C2B_DEBUG=64): cycle stamps of compute wave 0 and stager wave 4 of a few workgroups.
usage: RVC_AMD_LIB=.../librvc_amd_ablate.so RVC_C2B_DEBUG=64 python tools/stamp_conv2dbf.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "codename-rvc-fork-3_amd"))
import torch
from rvc_amd import _native as N

dev = torch.device("cuda:0")
for c_in, c_out, h, w in [(16, 16, 3008, 128), (32, 32, 1504, 64), (64, 64, 752, 32)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, c_in, h, w, generator=g).to(dev)
    wt = torch.randn(c_out, c_in, 3, 3, generator=g) / (c_in * 9) ** 0.5
    b = torch.randn(c_out, generator=g).to(dev)
    res = torch.randn(1, c_out, h, w, generator=g).to(dev)
    u = N.conv2d_bf16x3_pack_weight(wt, dev)
    y = torch.empty_like(res)
    st = torch.zeros(256 * 2 * 64, dtype=torch.int64, device=dev)
    for _ in range(3):
        rc = N._lib.rvc_conv2d_bf16x3_forward(x.data_ptr(), u.data_ptr(), b.data_ptr(), res.data_ptr(), y.data_ptr(), 1, c_in, c_out, h, w, 1,
                                              st.data_ptr(), st.numel() * 8, N._stream())
        assert rc == 0
    torch.cuda.synchronize()
    s = st.cpu().view(256, 2, 64)
    print(f"== {c_in} -> {c_out}, {h} x {w}")
    for blk in (0, 97, 200):
        for role, name, per in ((0, "compute", 4), (1, "stager ", 5)):
            v = s[blk, role]
            n = int((v != 0).sum())
            t0 = int(v[0])
            rel = [(int(v[i]) - t0) for i in range(n)]
            deltas = [rel[i] - rel[i - 1] for i in range(1, n)]
            print(f"block {blk:3d} {name}: total {rel[-1]:7d} cycles; deltas (start | " + ("mfma B io nextA" if role == 0 else "write issue store+request B nextA") + ")")
            print("    ", deltas[:1], [deltas[1 + i:1 + i + per] for i in range(0, len(deltas) - 1, per)])
