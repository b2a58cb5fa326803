#!/usr/bin/env python3
"""Per-stage HIP-event timing of one BASELINE cfg-2 utterance (steady state, after warm-up)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
import torch.nn.functional as F
from scipy import signal
from rvc_amd import _native
from rvc_amd.infer.infer import VoiceConverter
from rvc_amd.infer import pipeline as P
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.encoders import text_encoder
from rvc_amd.lib.algorithm.residuals import flow_reverse

dev = "cuda:0"
vc = VoiceConverter(device=dev)
vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0))
vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0))
vc.vc.set_index(S.synth_index(100_000, seed=0))
audio = S.synth_audio(480_000, seed=0)

class Timer:
    def __init__(self): self.t = {}
    def run(self, name, fn):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize(); self.t.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return out

tm = Timer()
pl, net, hub = vc.vc, vc.net_g, vc.hubert_model
for it in range(3):
    a = tm.run("host filtfilt+pad", lambda: np.pad(signal.filtfilt(P.bh, P.ah, audio), (16000, 16000), mode="reflect"))
    ad = tm.run("H2D audio", lambda: torch.from_numpy(a).float().to(dev))
    mel, n = tm.run("logmel (HIP)", lambda: _native.logmel_rmvpe(ad.unsqueeze(0)))
    w = pl.model_rmvpe.w
    def unet():
        return pl.model_rmvpe.mel2hidden(mel, n)
    hidden = tm.run("rmvpe net (unet+gru)", unet)
    gi = torch.randn(1, 3232, 2, 768, device=dev)
    tm.run("  of which bigru multi-CU", lambda: _native.bigru_forward(gi, w["gru.whhT"], w["gru.bhh"], True))
    tm.run("  (bigru single-CU variant)", lambda: _native.bigru_forward(gi, w["gru.whhT"], w["gru.bhh"], False))
    f0 = tm.run("rmvpe decode + D2H", lambda: pl.model_rmvpe.decode(hidden[0]).cpu().numpy())
    def quant():
        f0_mel = 1127 * np.log(1 + f0 / 700)
        f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - pl.f0_mel_min) * 254 / (pl.f0_mel_max - pl.f0_mel_min) + 1
        f0_mel[f0_mel <= 1] = 1; f0_mel[f0_mel > 255] = 255
        c = np.rint(f0_mel).astype(int)
        return torch.tensor(c[:3200], device=dev).unsqueeze(0).long(), torch.tensor(f0[:3200], device=dev).unsqueeze(0).float()
    pitch, pitchf = tm.run("f0 quantise + H2D", quant)
    feats = tm.run("hubert", lambda: hub(ad.view(1, -1))["last_hidden_state"])
    feats2 = tm.run("knn search+blend (HIP)", lambda: pl._retrieve_speaker_embeddings(feats, pl._preset_index, None, 0.75))
    ph = F.interpolate(feats2.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    T = 3198
    pitch, pitchf = pitch[:, :T], pitchf[:, :T]
    sid = torch.tensor([0], device=dev)
    lens = torch.tensor([T], device=dev)
    g = F.embedding(sid, net.w["emb_g.weight"]).unsqueeze(-1)
    m_p, logs_p, x_mask = tm.run("text encoder", lambda: text_encoder(net.w, ph, pitch, lens))
    nz = tm.run("noise (device RNG)", lambda: net._draw(None, 1, T))
    z_p = (m_p + torch.exp(logs_p) * nz["z"] * 0.66666) * x_mask
    z = tm.run("flow", lambda: flow_reverse(net.w, z_p, x_mask, g))
    o = tm.run("decoder (HIP)", lambda: net.dec.forward((z * x_mask).contiguous(), pitchf.contiguous(), g[:, :, 0].contiguous(), src_randn=nz["src_randn"]))
    out = tm.run("crop+normalise+D2H", lambda: o[0, 0, 48000:-48000].cpu().numpy())
    tm.run("whole pipeline()", lambda: vc.convert_array(audio, index_rate=0.75))
print(f"{'stage':28s} {'ms (last run)':>12s}")
tot = 0
for k, v in tm.t.items():
    print(f"{k:28s} {v[-1]:12.2f}")
    if k != "whole pipeline()" and not k.startswith("  "): tot += v[-1]
print(f"{'sum of stages':28s} {tot:12.2f}")
