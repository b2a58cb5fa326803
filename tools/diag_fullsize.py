#!/usr/bin/env python3
"""Where does the full-length (cfg 2) product-vs-oracle waveform difference come from?  Prints the error per second of
output, the f0 difference between the GPU RMVPE and the oracle's, and the error again with the oracle's f0 contour
injected into the product (isolates the sensitivity of the NSF phase integral to f0)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
from oracle import rvc_oracle as O  # noqa: E402
from rvc_amd.infer.infer import VoiceConverter  # noqa: E402
from rvc_amd.lib import synthetic as S  # noqa: E402

DEV = "cuda:0"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, dtype=np.float64) ** 2)))


cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
hub_sd, rm_sd = S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0)
vc = VoiceConverter(device=DEV)
vc.load_checkpoint_dict(cpt)
vc.load_hubert_state_dict(hub_sd)
vc.vc.load_rmvpe_state_dict(rm_sd)
big = S.synth_index(rows, seed=0) if rows else None
rate = 0.75 if rows else 0.0
if rows:
    vc.vc.set_index(big)
audio = S.synth_audio(int(16000 * secs), seed=0)
taps = {}
torch.manual_seed(1234)
want = O.pipeline(hub_sd, rm_sd, cpt, audio.copy(), sid=0, pitch=0, big_npy=big, index_rate=rate, protect=0.5, taps=taps,
                  knn_dtype=np.float64)


def product():
    return vc.vc.pipeline(vc.hubert_model, vc.net_g, 0, audio.copy(), 0, "rmvpe", "", rate, True, 3, 1, "v2", 0.5, 128, False, 1,
                          None, noise_seed=1234)


got = product()
print(f"plain product: rms err {rms(got - want):.3e} (oracle rms {rms(want):.3f})")
per = [rms(got[i:i + 48000] - want[i:i + 48000]) for i in range(0, len(want), 48000)]
print("error per second of output:", " ".join(f"{e:.1e}" for e in per))

a = np.pad(O.highpass(audio), (16000, 16000), mode="reflect")
f0_gpu = vc.vc.model_rmvpe.infer_from_audio_device(torch.from_numpy(a).float().to(DEV), thred=0.03).cpu().numpy()
f0_ref = taps["f0"]
n = min(len(f0_gpu), len(f0_ref))
voiced = (f0_ref[:n] > 0) & (f0_gpu[:n] > 0)
rel = np.abs(f0_gpu[:n][voiced] - f0_ref[:n][voiced]) / f0_ref[:n][voiced]
print(f"f0: {voiced.sum()} voiced frames, relative difference mean {rel.mean():.2e} max {rel.max():.2e}; "
      f"voicing decisions differ on {(f0_ref[:n] > 0).sum() - voiced.sum()} frames")
drift = np.cumsum(f0_gpu[:n] - f0_ref[:n]) / 100.0
print(f"integrated f0 difference (cycles): final {drift[-1]:+.4f}, max |.| {np.abs(drift).max():.4f}")

# inject the oracle's contour
f0_t = torch.from_numpy(np.ascontiguousarray(taps["f0"])).to(DEV)
full = O.rmvpe_infer_from_audio(a, rm_sd)
f0_full = torch.from_numpy(full).to(DEV)
vc.vc.model_rmvpe.back_half_device = lambda gi, n_frames, thred=0.03, taps=None: f0_full
got2 = product()
print(f"product with the oracle's f0 injected: rms err {rms(got2 - want):.3e}")
per = [rms(got2[i:i + 48000] - want[i:i + 48000]) for i in range(0, len(want), 48000)]
print("error per second of output:", " ".join(f"{e:.1e}" for e in per))
