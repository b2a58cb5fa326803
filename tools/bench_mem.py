import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev="cuda:0"
def t(fn, n=10):
    for _ in range(2): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
for C,L in ((32,1535040),(64,767520),(128,383760)):
    x=torch.randn(1,C,L,device=dev); res=torch.randn(1,C,L,device=dev); bias=torch.zeros(C,device=dev); y=torch.empty_like(x)
    print(f"C={C}: torch add {t(lambda: torch.add(x,res,out=y)):.0f} us; copy {t(lambda: y.copy_(x)):.0f} us", end="; ")
    for K in (1,3):
        w=_native.conv1d_pack_weight(torch.randn(C,C,K)*0.03,dev)
        print(f"K={K} res {t(lambda: _native.conv1d_forward(x,w,bias,C,K,1,0.1,res=res)):.0f} us, nores {t(lambda: _native.conv1d_forward(x,w,bias,C,K,1,0.1)):.0f} us", end="; ")
    print()
