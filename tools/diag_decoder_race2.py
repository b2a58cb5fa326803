#!/usr/bin/env python3
"""Two host threads run decoder forwards (same handle, different lengths) on their own streams: each output must equal its
one-at-a-time reference bit for bit."""
import os, sys, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.weights import fold_weight_norm
VOC = os.environ.get("VOC", "HiFi-GAN")
cpt = S.make_synth_checkpoint(48000, VOC, seed=0)
folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
dec = _native.Decoder(VOC, 48000, folded)
DIM = 9 if VOC.startswith("MRF") else 1
dev = "cuda:0"
def inputs(T, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn(1, 192, T, device=dev, generator=g), torch.full((1, T), 220.0, device=dev), torch.randn(1, 256, device=dev, generator=g),
            torch.zeros(1, T * 480, DIM, device=dev), torch.zeros(1, DIM, device=dev))
Ts = [int(v) for v in os.environ.get("TS", "500,800").split(",")]
ins = [inputs(T, i) for i, T in enumerate(Ts)]
refs = []
for z, f0, g, nz, rnd in ins:
    refs.append(dec.forward(z, f0, g, src_randn=nz, src_rand=rnd).clone())
torch.cuda.synchronize()
bad = [0] * len(Ts)
def worker(i, reps):
    st = torch.cuda.Stream(device=dev)
    z, f0, g, nz, rnd = ins[i]
    with torch.cuda.stream(st):
        for rep in range(reps):
            out = dec.forward(z, f0, g, src_randn=nz, src_rand=rnd)
            st.synchronize()
            d = (out - refs[i]).abs()
            if d.max().item() > 0:
                bad[i] += 1
                if bad[i] <= 3:
                    print(f"thread {i} (T={Ts[i]}) rep {rep}: max abs diff {d.max().item():.3e}, {int((d > 0).sum())} samples differ, first at {int(torch.argmax((d.flatten() > 0).float()))}", flush=True)
th = [threading.Thread(target=worker, args=(i, int(os.environ.get("REPS", 30)))) for i in range(len(Ts))]
for t in th: t.start()
for t in th: t.join()
print("runs that differ per thread:", bad)
