#!/usr/bin/env python3
"""Device filtfilt (K6) on a cfg-2 utterance: time per call, error against scipy.signal.filtfilt."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "codename-rvc-fork-3_amd")]
import numpy as np, torch
from scipy import signal
from rvc_amd import _native
from rvc_amd.lib import synthetic as S
b, a = signal.butter(N=5, Wn=48, btype="high", fs=16000)
x = S.synth_audio(480_000, seed=0)
xd = torch.from_numpy(x).to("cuda:0")
for _ in range(3): y = _native.filtfilt_order5(xd, b, a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): y = _native.filtfilt_order5(xd, b, a)
e1.record(); torch.cuda.synchronize()
ref = signal.filtfilt(b, a, x)
print(f"filtfilt 480000 samples: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per call, max |err| vs scipy {np.abs(y.cpu().numpy() - ref).max():.2e}")
