#!/usr/bin/env python3
"""ResBlock branches of a vocoder stage on side streams against every launch on one stream: same box, alternating, bit-identity of
the waveform checked.  Four decoders: NSF 48 k at T = 3198 (BASELINE cfg 2), NSF 40 k at 1000, MRF 48 k at 3198, NSF 48 k at 301."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import torch
from rvc_amd import _native
from rvc_amd.lib import synthetic as S

dev = "cuda:0"
for voc, sr, T in (("HiFi-GAN", 48000, 3198), ("HiFi-GAN", 40000, 1000), ("MRF HiFi-GAN", 48000, 3198), ("HiFi-GAN", 48000, 301)):
    from rvc_amd.infer.infer import VoiceConverter
    vc = VoiceConverter(device=dev)
    vc.load_checkpoint_dict(S.make_synth_checkpoint(sr, voc, seed=0))
    dec = vc.net_g.dec
    upp = dec.upp
    dim = 9 if voc.startswith("MRF") else 1
    torch.manual_seed(1)
    z = torch.randn(1, 192, T, device=dev); f0 = torch.full((1, T), 220.0, device=dev); g = torch.randn(1, 256, device=dev)
    nz = torch.randn(1, T * upp, dim, device=dev); rnd = torch.rand(1, dim, device=dev)

    def run(n_side, reps):
        dec.set_branch_parallel(n_side)
        out = dec.forward(z, f0, g, src_randn=nz, src_rand=rnd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = dec.forward(z, f0, g, src_randn=nz, src_rand=rnd)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / reps * 1e3
        lat = []
        for _ in range(reps):      # one call from an idle device: what a caller that is not ahead of the GPU sees
            t0 = time.perf_counter()
            dec.forward(z, f0, g, src_randn=nz, src_rand=rnd)
            th = time.perf_counter()
            torch.cuda.synchronize()
            lat.append(((time.perf_counter() - t0) * 1e3, (th - t0) * 1e3))
        return out, t, min(l[0] for l in lat), min(l[1] for l in lat)

    ts, outs, lat, host = {}, {}, {}, {}
    for rnd_i in range(3):
        for n_side in (0, 1, 2):
            o, t, l, h = run(n_side, 10)
            ts.setdefault(n_side, []).append(t)
            lat[n_side] = min(lat.get(n_side, 1e9), l)
            host[n_side] = min(host.get(n_side, 1e9), h)
            outs[n_side] = o
    same = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    print(f"{voc} {sr} T={T}: " + " | ".join(f"{n} side stream(s): {min(ts[n]):.3f} ms back to back, {lat[n]:.3f} ms alone (host returns after {host[n]:.3f})"
                                              for n in (0, 1, 2)) + f" | waveforms bit-identical: {same}")
    assert same
