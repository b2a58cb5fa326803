#!/usr/bin/env python3
"""Decoder-only run for rocprofv3 (BASELINE cfg-2 shape: T = 3198 frames, NSF 48k).  usage: profile_decoder.py [vocoder [f32|bf16]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.weights import fold_weight_norm
voc = sys.argv[1] if len(sys.argv) > 1 else "HiFi-GAN"
cpt = S.make_synth_checkpoint(48000, voc, seed=0)
folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
storage = sys.argv[2] if len(sys.argv) > 2 else "f32"       # "bf16": BASELINE cfg 4's weight storage (K3f with one-term taps)
dec = _native.Decoder(voc, 48000, folded, **({"weight_storage": "bf16"} if storage == "bf16" else {}))
dev = "cuda:0"
T = 3198
dim = 9 if voc.startswith("MRF") else 1
z = torch.randn(1, 192, T, device=dev); f0 = torch.full((1, T), 220.0, device=dev); g = torch.randn(1, 256, device=dev)
nz = torch.randn(1, T * 480, dim, device=dev); rnd = torch.rand(1, dim, device=dev)
adain = None
if voc == "RefineGAN":
    n, length, ch = 0, T, 512
    for r in (12, 10, 2, 2):
        length, ch = length * r, ch // 2
        n += 6 * ch * length
    adain = torch.randn(n, device=dev)
for _ in range(int(os.environ.get("REPS", "3"))):
    dec.forward(z, f0, g, src_randn=nz, src_rand=rnd, adain_randn=adain)
torch.cuda.synchronize()
