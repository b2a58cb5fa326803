import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
def t(C, L, K, d, res=True):
    x = torch.randn(1, C, L, device=dev); r = torch.randn(1, C, L, device=dev) if res else None; b = torch.zeros(C, device=dev); y = torch.empty_like(x)
    u = _native.conv1d_winobf_pack_weight(torch.randn(C, C, K) * 0.03, dev)
    f = lambda: _native.conv1d_winobf_forward(x, u, b, C, K, d, 0.1, res=r, out=y)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C} L={L} K={K} d={d} res={res}: {e0.elapsed_time(e1)/10*1e3:.1f} us", flush=True)
for C, L in ((256, 11980), (128, 119800), (256, 38376), (128, 383760), (128, 119808), (128, 120000), (256, 12000)):
    for d in (1, 3):
        t(C, L, 11, d)
t(128, 119800, 11, 1, res=False)
