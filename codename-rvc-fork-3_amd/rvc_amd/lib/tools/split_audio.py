"""Silence splitting around the conversion (rvc/lib/tools/split_audio.py:5-79), host NumPy: it runs once per file on
the 16 kHz input, outside the per-utterance hot path.

``librosa.effects.split`` is not available here; ``_nonsilent_intervals`` restates librosa 0.11's published algorithm
(centre-padded RMS frames -> dB relative to the loudest frame -> runs of frames above -top_db -> sample indices)."""
from __future__ import annotations

import numpy as np


def _frame_rms(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    y = np.pad(y, (frame_length // 2, frame_length // 2), mode="constant")
    n_frames = 1 + (y.shape[0] - frame_length) // hop_length
    idx = np.arange(frame_length)[:, None] + hop_length * np.arange(n_frames)[None, :]
    return np.sqrt(np.mean(np.abs(y[idx]) ** 2, axis=0))


def _nonsilent_intervals(y: np.ndarray, top_db: float, frame_length: int, hop_length: int) -> np.ndarray:
    """librosa.effects.split(y, top_db=..., frame_length=..., hop_length=...) with its defaults ref=np.max, amin=1e-5."""
    mag = _frame_rms(y, frame_length, hop_length)
    amin = 1e-5 ** 2
    db = 10.0 * np.log10(np.maximum(amin, mag ** 2)) - 10.0 * np.log10(np.maximum(amin, np.max(mag) ** 2))
    loud = db > -top_db
    edges = [np.flatnonzero(np.diff(loud.astype(int))) + 1]
    if loud[0]:
        edges.insert(0, np.array([0]))
    if loud[-1]:
        edges.append(np.array([len(loud)]))
    samples = np.minimum(np.concatenate(edges) * hop_length, y.shape[-1])
    return samples.reshape(-1, 2)


def process_audio(audio, sr=16000, silence_thresh=-60, min_silence_len=250):
    """split_audio.py:5-26: (list of non-silent segments, their [start, end) sample intervals)."""
    frame_length = int(min_silence_len / 1000 * sr)
    hop_length = frame_length // 2
    intervals = _nonsilent_intervals(audio, -silence_thresh, frame_length, hop_length)
    return [audio[start:end] for start, end in intervals], intervals


def merge_audio(audio_segments_org, audio_segments_new, intervals, sr_orig, sr_new):
    """split_audio.py:29-79: converted segments back on the original time line (at sr_new), gaps as zeros, plus the
    reference's per-segment length compensation (the converted segment is a few frames shorter than its source)."""
    dtype = audio_segments_new[0].dtype
    ratio = sr_new / sr_orig
    pieces = []
    for i, (start, end) in enumerate(intervals):
        start_new, end_new = int(start * ratio), int(end * ratio)
        diff = len(audio_segments_new[i]) / sr_new - len(audio_segments_org[i]) / sr_orig
        pad = np.zeros(int(abs(diff) * sr_new), dtype=dtype)
        if i == 0 and start_new > 0:
            pieces.append(np.zeros(start_new, dtype=dtype))
        if diff > 0:
            pieces.append(pad)
        pieces.append(audio_segments_new[i])
        if diff < 0:
            pieces.append(pad)
        if i < len(intervals) - 1:
            gap = int(intervals[i + 1][0] * ratio) - end_new
            if gap > 0:
                pieces.append(np.zeros(gap, dtype=dtype))
    return np.concatenate(pieces) if pieces else np.array([], dtype=dtype)
