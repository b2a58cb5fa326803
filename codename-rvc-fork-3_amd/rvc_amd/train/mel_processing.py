"""Training-side spectrogram / log-mel features with the reference's function surface
(rvc/train/mel_processing.py:53-146), computed by librvc_amd's K4b transform (include/rvc_amd.h: rvc_mel_*): the
windowed DFT as one fp32 matrix-core GEMM, magnitude and the sparse mel projection as epilogue kernels -- the same kernel
chain that serves RMVPE's front end at inference.

``librosa.filters.mel`` (third party, absent here) is restated from librosa 0.11's published algorithm (Slaney scale,
Slaney area normalisation -- the defaults the reference calls it with, mel_processing.py:113-115).
"""
from __future__ import annotations

import numpy as np
import torch

from rvc_amd import _native


def dynamic_range_compression_torch(x, C=1, clip_val=1e-5):
    return torch.log(torch.clamp(x, min=clip_val) * C)


def dynamic_range_decompression_torch(x, C=1):
    return torch.exp(x) / C


def spectral_normalize_torch(magnitudes):
    return dynamic_range_compression_torch(magnitudes)


def spectral_de_normalize_torch(magnitudes):
    return dynamic_range_decompression_torch(magnitudes)


def _hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def librosa_mel_fn(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr=..., n_fft=..., n_mels=..., fmin=..., fmax=...) with its defaults htk=False, norm="slaney",
    dtype float32: triangular filters on the Slaney mel scale, each scaled to unit area."""
    fmax = float(sr) / 2 if fmax is None else fmax
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


_transforms = {}


def _transform(n_fft, hop_size, win_size, mel_key, device):
    """mel_key: None (spectrogram only) or the (num_mels, sample_rate, fmin, fmax) of _basis -- scalars, so that looking a
    cached handle up in the training loop does not hash half a megabyte of filter bank per call."""
    key = (n_fft, hop_size, win_size, mel_key, str(device))
    if key not in _transforms:
        with torch.cuda.device(device):
            m = _basis(n_fft, *mel_key) if mel_key is not None else np.zeros((1, n_fft // 2 + 1), dtype=np.float32)
            _transforms[key] = _native.MelTransform(n_fft, hop_size, win_size, int((n_fft - hop_size) / 2), m, mag_eps=1e-6,
                                                    log_floor=1e-5)
    return _transforms[key]


def _require_device(y):
    if not y.is_cuda:
        raise RuntimeError("rvc_amd.train.mel_processing runs on a HIP device only (the CPU restatement lives in oracle/)")
    return y.float().contiguous()


def spectrogram_torch(y, n_fft, hop_size, win_size, center=False):
    """mel_processing.py:53-97: y [B, n] -> sqrt(re^2 + im^2 + 1e-6) [B, n_fft/2+1, n/hop]."""
    if center:
        raise NotImplementedError("the reference only calls this with center=False")
    y = _require_device(y)
    return _transform(n_fft, hop_size, win_size, None, y.device).forward(y, want_mel=False, want_spec=True)[1]


_mel_basis = {}


def _basis(n_fft, num_mels, sample_rate, fmin, fmax):
    key = (n_fft, num_mels, sample_rate, fmin, fmax)
    if key not in _mel_basis:
        _mel_basis[key] = librosa_mel_fn(sr=sample_rate, n_fft=n_fft, n_mels=num_mels, fmin=fmin, fmax=fmax)
    return _mel_basis[key]


def spec_to_mel_torch(spec, n_fft, num_mels, sample_rate, fmin, fmax):
    """mel_processing.py:100-123 (the projection of an existing magnitude spectrogram stays a plain matmul)."""
    mel = torch.from_numpy(_basis(n_fft, num_mels, sample_rate, fmin, fmax)).to(dtype=spec.dtype, device=spec.device)
    return spectral_normalize_torch(torch.matmul(mel, spec))


def mel_spectrogram_torch(y, n_fft, num_mels, sample_rate, hop_size, win_size, fmin, fmax, center=False):
    """mel_processing.py:126-146, fused: frames -> DFT GEMM -> |.| -> sparse mel -> log, one pass over the audio."""
    if center:
        raise NotImplementedError("the reference only calls this with center=False")
    y = _require_device(y)
    return _transform(n_fft, hop_size, win_size, (num_mels, sample_rate, fmin, fmax), y.device).forward(y)[0]
