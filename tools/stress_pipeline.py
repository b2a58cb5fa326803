#!/usr/bin/env python3
"""Sporadic-corruption hunt on SHORT clips (one full-suite run of round 6 failed the 2 s / 40 kHz sweep case at 1.1e-2 where reruns give
2.5e-6; the RMVPE path is bit-reproducible across processes, tools/diag_rmvpe_repro.py, so an f0 tie is not the obvious explanation).
Part A: the vocoder alone (bit-stable kernels) on fixed inputs, many iterations with allocator churn (fresh buffers every few
iterations): every output must equal the first BIT FOR BIT.  Part B: the whole pipeline on the failing case, rms difference to the
first run (the library GEMMs wobble at 1e-6).  usage: stress_pipeline.py [iterations A] [iterations B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from rvc_amd import _native
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.weights import fold_weight_norm
dev = "cuda:0"
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 400
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for sr, voc, T in ((40000, "HiFi-GAN", 408), (48000, "HiFi-GAN", 330), (48000, "MRF HiFi-GAN", 300), (32000, "HiFi-GAN", 250)):
    cpt = S.make_synth_checkpoint(sr, voc, seed=0)
    folded = {k[4:]: v for k, v in fold_weight_norm(cpt["weight"]).items() if k.startswith("dec.")}
    rates, ksizes = cpt["config"][12], cpt["config"][14]
    dec = _native.Decoder(voc, sr, folded, upsample_rates=rates, upsample_kernel_sizes=ksizes)
    upp = int(np.prod(rates)); dim = 9 if voc.startswith("MRF") else 1
    g = torch.Generator(device=dev).manual_seed(T)
    z = torch.randn(1, 192, T, device=dev, generator=g); f0 = torch.full((1, T), 220.0, device=dev); gv = torch.randn(1, 256, device=dev, generator=g)
    nz = torch.randn(1, T * upp, dim, device=dev, generator=g); rnd = torch.rand(1, dim, device=dev, generator=g)
    ref = dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd).clone()
    n_diff = 0
    junk = []
    for it in range(NA):
        if it % 7 == 0:
            junk = [torch.empty(int(np.random.randint(1, 50)) * 1_000_003, device=dev) for _ in range(3)]   # move the allocator's blocks around
        if it % 13 == 0:
            torch.cuda.empty_cache()
        out = dec.forward(z, f0, gv, src_randn=nz, src_rand=rnd)
        if not torch.equal(out, ref):
            n_diff += 1
            d = (out - ref).abs()
            print(f"  {voc} {sr} T={T} iteration {it}: DIFFERS, max {d.max().item():.3e}, {int((d > 0).sum())} elements, first at {int((d > 0).nonzero()[0, -1])}", flush=True)
    bad += n_diff
    print(f"A: vocoder {voc} {sr} Hz T={T}: {NA} forwards with allocator churn, {n_diff} differ from the first", flush=True)
from rvc_amd.infer.infer import VoiceConverter
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(40000, "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(S.synth_index(3000, seed=2))
audio = S.synth_audio(33333, seed=33333 % 89)
run = lambda: vc.vc.pipeline(vc.hubert_model, vc.net_g, 108, audio.copy(), 0, "rmvpe", "", 0.3, True, 3, 1, "v2", 0.49, 128, False, 1, None, noise_seed=33333)
first = run()
worst = 0.0
for it in range(NB):
    if it % 5 == 0: torch.cuda.empty_cache()
    out = run()
    e = float(np.sqrt(np.mean((out.astype(np.float64) - first) ** 2)))
    worst = max(worst, e)
    if e > 1e-4:
        bad += 1
        print(f"  pipeline iteration {it}: rms difference to the first run {e:.3e}", flush=True)
print(f"B: pipeline (2 s, 40 kHz, the failing sweep case): {NB} runs, worst rms difference to the first {worst:.3e}")
sys.exit(1 if bad else 0)
