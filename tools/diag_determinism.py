#!/usr/bin/env python3
"""Which parts of the path are bit-reproducible run to run on the GPU?  Each network / kernel is evaluated several times on
the same input; prints the largest difference between runs (0 = bit-stable)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
from rvc_amd import _native  # noqa: E402
from rvc_amd.infer.infer import VoiceConverter  # noqa: E402
from rvc_amd.lib import synthetic as S  # noqa: E402

DEV = "cuda:0"
vc = VoiceConverter(device=DEV)
cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
vc.load_checkpoint_dict(cpt)
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
audio = torch.from_numpy(S.synth_audio(512000, seed=0)).float().to(DEV)


def spread(name, fn, reps=4):
    outs = [fn().float().clone() for _ in range(reps)]
    torch.cuda.synchronize()
    d = max(float((o - outs[0]).abs().max()) for o in outs[1:])
    print(f"{name:28s} max |run_i - run_0| = {d:.3e}  (|x| max {float(outs[0].abs().max()):.3e})")


rm = vc.vc.model_rmvpe
mel, n = _native.logmel_rmvpe(audio.unsqueeze(0))
spread("logmel (HIP)", lambda: _native.logmel_rmvpe(audio.unsqueeze(0))[0])
spread("RMVPE U-Net (MIOpen+K8)", lambda: rm.unet_features(mel))
gi = rm.unet_features(mel)
spread("BiGRU + fc (HIP + hipBLASLt)", lambda: rm.gru_head(gi, n))
spread("HuBERT", lambda: vc.hubert_model(audio.view(1, -1))["last_hidden_state"])
feats = vc.hubert_model(audio.view(1, -1))["last_hidden_state"]
T = 3198
phone = feats.repeat_interleave(2, dim=1)[:, :T].contiguous()
pitch = torch.randint(1, 255, (1, T), device=DEV)
pitchf = torch.rand(1, T, device=DEV) * 200 + 100
sid = torch.zeros(1, dtype=torch.long, device=DEV)
nz = vc.net_g._draw(None, 1, T)
lengths = torch.tensor([T], device=DEV)


def synth(part):
    o, mask, (z, z_p, m_p, logs_p) = vc.net_g.infer(phone, lengths, pitch, pitchf, sid, noise=nz, phone_lengths_host=[T])
    return {"m_p": m_p, "z": z, "o": o}[part]


spread("TextEncoder m_p", lambda: synth("m_p"))
spread("flow z", lambda: synth("z"))
spread("whole synthesizer o", lambda: synth("o"))
z = torch.randn(1, 192, T, device=DEV)
f0 = torch.full((1, T), 220.0, device=DEV)
g = torch.randn(1, 256, device=DEV)
spread("vocoder (HIP)", lambda: vc.net_g.dec.forward(z, f0, g, src_randn=nz["src_randn"].contiguous()))
idx = torch.from_numpy(S.synth_index(100_000, seed=0)).to(DEV)
norms = _native.knn_index_norms(idx)
q = feats[0].contiguous()
spread("kNN d2 (HIP)", lambda: _native.knn_search(idx, norms, q)[0])
