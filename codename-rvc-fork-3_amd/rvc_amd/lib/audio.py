"""File-level input front end (rvc/lib/utils.py:21-85: ``load_audio`` / ``load_audio_infer``): read a WAV, fold to mono,
resample to the requested rate.  ``soundfile`` and ``soxr`` are third-party and absent from the image, so

* decoding covers RIFF/WAVE only -- PCM 8/16/24/32 bit, IEEE float 32/64, plain and WAVE_FORMAT_EXTENSIBLE headers --
  parsed here with ``struct`` (the reference reads anything libsndfile reads);
* resampling is this build's own polyphase Kaiser-windowed sinc (``resample_filter``) applied on the device by
  librvc_amd's ``rvc_resample_poly_f64`` -- the reference asks librosa for ``res_type="soxr_vhq"``.  **Parity unpinned**
  against soxr; the filter is designed to VHQ-class figures (pass band to 0.91 of the lower Nyquist, > 140 dB rejection).
"""
from __future__ import annotations

import math
import os
import struct
from functools import lru_cache

import numpy as np
import torch


def read_wav(path: str):
    """-> (float64 array [n] or [n, channels] in [-1, 1), sample rate)"""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] not in (b"RIFF", b"RF64") or data[8:12] != b"WAVE":
        raise RuntimeError(f"{path}: not a RIFF/WAVE file (only WAV input is supported without soundfile)")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        tag, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8: pos + 8 + size]
        if tag == b"fmt ":
            fmt = body
        elif tag == b"data":
            payload = data[pos + 8: pos + 8 + size] if size != 0xFFFFFFFF else data[pos + 8:]
            break
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or len(fmt) < 16:
        raise RuntimeError(f"{path}: missing fmt or data chunk")
    code, channels, sr, _, _, bits = struct.unpack_from("<HHIIHH", fmt, 0)
    if code == 0xFFFE and len(fmt) >= 26:          # WAVE_FORMAT_EXTENSIBLE: the real code opens the sub-format GUID
        code = struct.unpack_from("<H", fmt, 24)[0]
    if channels < 1:
        raise RuntimeError(f"{path}: no channels")
    if code == 1 and bits == 16:
        a = np.frombuffer(payload, dtype="<i2").astype(np.float64) / 32768.0
    elif code == 1 and bits == 8:
        a = (np.frombuffer(payload, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif code == 1 and bits == 24:
        raw = np.frombuffer(payload[: len(payload) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = raw[:, 0] | (raw[:, 1] << 8) | (raw[:, 2] << 16)
        a = np.where(v >= 1 << 23, v - (1 << 24), v).astype(np.float64) / float(1 << 23)
    elif code == 1 and bits == 32:
        a = np.frombuffer(payload, dtype="<i4").astype(np.float64) / float(1 << 31)
    elif code == 3 and bits == 32:
        a = np.frombuffer(payload, dtype="<f4").astype(np.float64)
    elif code == 3 and bits == 64:
        a = np.frombuffer(payload, dtype="<f8").astype(np.float64)
    else:
        raise RuntimeError(f"{path}: unsupported WAV encoding (format code {code}, {bits} bits)")
    a = a[: a.size // channels * channels]
    return (a.reshape(-1, channels) if channels > 1 else a), int(sr)


@lru_cache(maxsize=16)
def resample_filter(up: int, down: int, zeros: int = 48, rejection_db: float = 150.0, passband: float = 0.91) -> np.ndarray:
    """FIR for rational resampling by up/down at the up-sampled rate: Kaiser-windowed sinc, cut-off midway between
    ``passband`` x the lower Nyquist and that Nyquist, pass-band gain ``up``.  float64, odd length."""
    from scipy import signal
    m = max(up, down)
    beta = 0.1102 * (rejection_db - 8.7)
    width = (1.0 - passband) / m                               # transition band, in units of the up-sampled Nyquist
    numtaps = int(math.ceil((rejection_db - 7.95) / (2.285 * math.pi * width))) | 1
    numtaps = max(numtaps, 2 * zeros * m + 1) if zeros else numtaps
    cutoff = (1.0 + passband) / 2.0 / m
    return np.ascontiguousarray(signal.firwin(numtaps, cutoff, window=("kaiser", beta)) * up, dtype=np.float64)


_filters_dev = {}


def resample(audio, orig_sr: int, target_sr: int, device="cuda:0"):
    """1-D float64 NumPy (or device tensor) at orig_sr -> the same kind at target_sr; ceil(n * ratio) samples, like
    librosa.resample(fix=True)."""
    from rvc_amd import _native
    if orig_sr == target_sr:
        return audio
    g = math.gcd(int(orig_sr), int(target_sr))
    up, down = int(target_sr) // g, int(orig_sr) // g
    as_numpy = not torch.is_tensor(audio)
    x = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float64)).to(device) if as_numpy else audio.to(dtype=torch.float64)
    key = (up, down, str(x.device))
    if key not in _filters_dev:
        _filters_dev[key] = torch.from_numpy(resample_filter(up, down)).to(x.device)
    with torch.cuda.device(x.device):   # the native call launches on the CURRENT stream: make it the stream of x's device
        y = _native.resample_poly(x, up, down, _filters_dev[key])
    return y.cpu().numpy() if as_numpy else y


def _load(file: str, sample_rate: int, device):
    audio, sr = read_wav(file)
    if audio.ndim > 1:
        audio = audio.mean(axis=1)                # librosa.to_mono
    if sr != sample_rate:
        audio = resample(audio, sr, sample_rate, device=device)
    return np.asarray(audio, dtype=np.float64).flatten()


def load_audio(file, sample_rate, device="cuda:0"):
    """rvc/lib/utils.py:21-50"""
    try:
        file = file.strip(" ").strip('"').strip("\n").strip('"').strip(" ")
        return _load(file, sample_rate, device)
    except Exception as error:
        raise RuntimeError(f"An error occurred loading the audio: {error}")


def load_audio_infer(file, sample_rate, device="cuda:0", **kwargs):
    """rvc/lib/utils.py:53-85 (formant shifting -- stftpitchshift, third party -- is outside the hot path)."""
    if kwargs.get("formant_shifting", False):
        raise RuntimeError("An error occurred loading the audio: formant shifting (stftpitchshift) is not part of this build")
    try:
        file = file.strip(" ").strip('"').strip("\n").strip('"').strip(" ")
        if not os.path.isfile(file):
            raise FileNotFoundError(f"File not found: {file}")
        return _load(file, sample_rate, device)
    except Exception as error:
        raise RuntimeError(f"An error occurred loading the audio: {error}")
