#!/usr/bin/env python3
"""Thread A: gemmbf mode 1 (HuBERT conv shape) repeatedly; thread B: decoder forwards.  Both must stay bit-identical to their
one-at-a-time references."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.weights import fold_weight_norm
dev = "cuda:0"
cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
dec = _native.Decoder("HiFi-GAN", 48000, folded)
T = 600
z = torch.randn(1, 192, T, device=dev); f0 = torch.full((1, T), 220.0, device=dev); g = torch.randn(1, 256, device=dev)
nz = torch.zeros(1, T * 480, 1, device=dev); rnd = torch.zeros(1, 1, device=dev)
ref_dec = dec.forward(z, f0, g, src_randn=nz, src_rand=rnd).clone()
C = 512
w = torch.randn(C, C, 3) * 0.03
a = _native.gemm_bf16x3_pack_weight(w, dev); bias = torch.randn(C, device=dev)
x = torch.randn(1, C, int(os.environ.get("L", 3000)), device=dev)
ref_conv = _native.conv1d_bf16x3(x, a, bias, C, 3, 2, 0, "gelu").clone()
torch.cuda.synchronize()
bad = {"conv": 0, "dec": 0}
stop = False
def conv_worker():
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        while not stop:
            out = _native.conv1d_bf16x3(x, a, bias, C, 3, 2, 0, "gelu"); st.synchronize()
            if (out - ref_conv).abs().max().item() > 0: bad["conv"] += 1
def dec_worker():
    global stop
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        for rep in range(40):
            out = dec.forward(z, f0, g, src_randn=nz, src_rand=rnd); st.synchronize()
            if (out - ref_dec).abs().max().item() > 0: bad["dec"] += 1
    stop = True
th = [threading.Thread(target=conv_worker), threading.Thread(target=dec_worker)]
for t in th: t.start()
for t in th: t.join()
print("runs that differ:", bad)
