#!/usr/bin/env python3
"""K14 (posconv.hip) at HuBERT-base's shape: time per call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native as N
dev="cuda:0"
x=torch.randn(1499,768,device=dev); w=torch.randn(768,48,128)*0.01; b=torch.randn(768,device=dev)
a=N.posconv_bf16x3_pack_weight(w,16,dev)
for _ in range(3): N.posconv_gelu_bf16x3(x,a,b,16,128,64)
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
ts=[]
for _ in range(5):
    e0.record()
    for _ in range(20): N.posconv_gelu_bf16x3(x,a,b,16,128,64)
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/20*1e3)
print("posconv 1499 x 768: %.1f us" % sorted(ts)[2])
