#!/usr/bin/env python3
"""Thread A: a gemmbf conv in a loop.  Thread B: one ResBlock conv form in a loop, compared bit for bit with its own
one-at-a-time reference."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
dev = "cuda:0"
Cg = 512
a = _native.gemm_bf16x3_pack_weight(torch.randn(Cg, Cg, 3) * 0.03, dev); gb = torch.randn(Cg, device=dev)
xg = torch.randn(1, Cg, 51000, device=dev)
FORM = os.environ.get("FORM", "wino"); C = int(os.environ.get("C", 128)); K = int(os.environ.get("K", 11)); D = int(os.environ.get("D", 1)); L = int(os.environ.get("L", 60000))
wt = torch.randn(C, C, K) * 0.03
x = torch.randn(1, C, L, device=dev); res = torch.randn(1, C, L, device=dev); bias = torch.randn(C, device=dev)
if FORM == "wino":
    u = _native.conv1d_wino_pack_weight(wt, dev); f = lambda: _native.conv1d_wino_forward(x, u, bias, C, K, D, 0.1, res=res)
elif FORM == "winobf":
    u = _native.conv1d_winobf_pack_weight(wt, dev); f = lambda: _native.conv1d_winobf_forward(x, u, bias, C, K, D, 0.1, res=res)
elif FORM == "conv2d":
    wv = _native.conv2d_pack_weight(torch.randn(64, 64, 3, 3) * 0.05, dev); xv = torch.randn(1, 64, 752, 32, device=dev); bv = torch.randn(64, device=dev)
    f = lambda: _native.conv2d_forward(xv, wv, bv, 64, 3, relu=True)
else:
    u = _native.conv1d_pack_weight(wt, dev); f = lambda: _native.conv1d_forward(x, u, bias, C, K, D, 0.1, res=res)
ref = f().clone(); torch.cuda.synchronize()
stop = False; bad = 0
CO = os.environ.get("CO", "gemmbf")
if CO == "conv2d":
    w2 = _native.conv2d_pack_weight(torch.randn(32, 32, 3, 3) * 0.05, dev); x2 = torch.randn(1, 32, 1504, 64, device=dev); b2 = torch.randn(32, device=dev)
if CO == "wino3":
    u3 = _native.conv1d_wino_pack_weight(torch.randn(64, 64, 3) * 0.05, dev); x3 = torch.randn(1, 64, 200000, device=dev); b3 = torch.randn(64, device=dev)
def co():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        while not stop:
            if CO == "gemmbf": _native.conv1d_bf16x3(xg, a, gb, Cg, 3, 2, 0, "gelu")
            elif CO == "conv2d": _native.conv2d_forward(x2, w2, b2, 32, 3, relu=True)
            elif CO == "wino3": _native.conv1d_wino_forward(x3, u3, b3, 64, 3, 1, 0.1)
            st.synchronize()
def victim():
    global stop, bad
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for rep in range(300):
            out = f(); st.synchronize()
            d = (out - ref).abs()
            if d.max().item() > 0:
                bad += 1
                if bad <= 2: print(f"rep {rep}: max abs diff {d.max().item():.3e}, {int((d > 0).sum())} differ, first {int(torch.argmax((d.flatten() > 0).float()))}", flush=True)
    stop = True
th = [threading.Thread(target=co), threading.Thread(target=victim)]
for t in th: t.start()
for t in th: t.join()
print(f"{FORM} C={C} K={K} d={D}: runs that differ: {bad} of 300")
