#!/usr/bin/env python3
"""RMVPE salience accuracy and time on the GPU vs the oracle's CPU fp32 evaluation, at the cfg-2 shape.
Run once per MIOpen setting (environment variables must be set before the process starts):
    python tools/diag_rmvpe.py                       # default solver choice
    MIOPEN_DEBUG_CONV_WINOGRAD=0 python tools/diag_rmvpe.py
Prints max |salience difference|, how many frames change their arg-max bin, and the U-Net time."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
from oracle import rvc_oracle as O  # noqa: E402
from rvc_amd import _native  # noqa: E402
from rvc_amd.lib import synthetic as S  # noqa: E402
from rvc_amd.lib.predictors.RMVPE import RMVPE0Predictor  # noqa: E402

DEV = "cuda:0"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rm_sd = S.make_rmvpe_state_dict(0)
audio = S.synth_audio(int(16000 * secs), seed=0)
a = np.pad(O.highpass(audio), (16000, 16000), mode="reflect").astype(np.float32)
cache = f"/tmp/rmvpe_hidden_{int(secs)}.npy"
if os.path.exists(cache):
    ref = np.load(cache)
else:
    with torch.no_grad():
        mel = O.logmel_rmvpe(torch.from_numpy(a).unsqueeze(0))
        ref = O.rmvpe_mel2hidden(mel, rm_sd)[0].numpy()
    np.save(cache, ref)
pred = RMVPE0Predictor(device=DEV, state_dict=rm_sd)
x = torch.from_numpy(a).to(DEV).unsqueeze(0)
mel, n = _native.logmel_rmvpe(x)
hid = pred.mel2hidden(mel, n)[0]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    gi = pred.unet_features(mel)
torch.cuda.synchronize()
t_unet = (time.perf_counter() - t0) / 5
hid = hid.cpu().numpy()
d = np.abs(hid - ref)
flips = int((hid.argmax(1) != ref.argmax(1)).sum())
# how close are the two best bins of the reference salience (what a perturbation has to overcome)?
srt = np.sort(ref, axis=1)
gap = srt[:, -1] - srt[:, -2]
print(f"env WINOGRAD={os.environ.get('MIOPEN_DEBUG_CONV_WINOGRAD', '-')} FIND_MODE={os.environ.get('MIOPEN_FIND_MODE', '-')}: "
      f"salience max abs diff {d.max():.3e} mean {d.mean():.3e}; arg-max flips {flips} of {ref.shape[0]} frames; "
      f"U-Net {t_unet * 1e3:.2f} ms; reference top-2 gap: min {gap.min():.2e} 1st percentile {np.percentile(gap, 1):.2e} median {np.median(gap):.2e}")
