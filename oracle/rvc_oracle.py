"""ORACLE -- CPU restatement of the reference's RVC inference hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product (``codename-rvc-fork-3_amd/``) never imports anything from ``oracle/``.

What it is: a plain, functional torch-CPU / NumPy restatement of the algorithm in
``/root/reference`` (codename0og/codename-rvc-fork-3 @ 2025-07-04), one function per
reference function, each citing the file:line it follows.  It consumes the same
exported state dicts the reference consumes (legacy ``weight_g/weight_v`` keys) and
takes every random draw either from the torch CPU generator *in the reference's order
and shapes* (so ``torch.manual_seed(s)`` + oracle == ``torch.manual_seed(s)`` + reference)
or from explicit noise tensors.

Pinning: the reference has no tests and no golden vectors.  The oracle is
pinned against outputs of the reference itself, imported in the build container by
``tests/golden/make_golden.py`` and committed under ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every one of them.  Two third-party pieces have
no source under /root/reference: ``transformers.HubertModel`` (pinned 4.44.2; the
container has 5.15.0, whose HubertModel generated the HuBERT fixture) and ``faiss``
(pinned faiss-cpu 1.7.3, absent here -> the kNN oracle is exact brute-force squared-L2,
``parity unpinned`` against a real IVF index).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from scipy import signal

Tensor = torch.Tensor

# ----------------------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------------------


def fold_weight_norm(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """weight = g * v / ||v|| through ``torch._weight_norm``, the op the reference's parametrization runs on every forward
    (the reference never removes weight-norm at inference: SURVEY §3.3).

    Accepts both the legacy export naming (``.weight_g/.weight_v``,
    extract_model.py:99-105) and ``.parametrizations.weight.original0/1``.
    The norm runs over every dim except the one where ``g`` is not 1.
    """
    out: Dict[str, Tensor] = {}
    done = set()
    for key, val in sd.items():
        if key.endswith(".weight_v") or key.endswith(".parametrizations.weight.original1"):
            if key.endswith(".weight_v"):
                base = key[: -len(".weight_v")]
                gk = base + ".weight_g"
            else:
                base = key[: -len(".parametrizations.weight.original1")]
                gk = base + ".parametrizations.weight.original0"
            g = sd[gk].float()
            v = val.float()
            keep = [d for d in range(v.dim()) if g.shape[d] != 1]
            # torch.nn.utils.parametrizations._WeightNorm.forward -> torch._weight_norm(v, g, dim): the op the reference runs
            out[base + ".weight"] = torch._weight_norm(v, g, keep[0] if keep else 0)
            done.add(key)
            done.add(gk)
    for key, val in sd.items():
        if key in done:
            continue
        out[key] = val.float() if val.is_floating_point() else val
    return out


class TorchNoise:
    """Draws from the torch CPU global generator, like the reference's randn_like / rand."""

    def randn(self, *shape):
        return torch.randn(*shape)

    def rand(self, *shape):
        return torch.rand(*shape)


class ListNoise:
    """Replays explicit tensors in draw order (explicit-noise mode)."""

    def __init__(self, tensors):
        self.t = list(tensors)
        self.i = 0

    def _next(self, shape):
        x = self.t[self.i]
        self.i += 1
        assert tuple(x.shape) == tuple(shape), (tuple(x.shape), tuple(shape))
        return x

    def randn(self, *shape):
        return self._next(shape)

    def rand(self, *shape):
        return self._next(shape)


# ----------------------------------------------------------------------------------------------
# pipeline front: high-pass, segmentation, padding   (rvc/infer/pipeline.py)
# ----------------------------------------------------------------------------------------------

# pipeline.py:23-28
BH, AH = signal.butter(N=5, Wn=48, btype="high", fs=16000)

X_PAD, X_QUERY, X_CENTER, X_MAX = 1, 6, 38, 41  # rvc/configs/config.py:116-118 (fp32)
WINDOW = 160  # pipeline.py:136


def highpass(audio: np.ndarray) -> np.ndarray:
    """pipeline.py:562"""
    return signal.filtfilt(BH, AH, audio)


def split_points(audio: np.ndarray, x_query=X_QUERY, x_center=X_CENTER, x_max=X_MAX):
    """pipeline.py:563-577: quietest sample (160-tap box sum) within +-t_query of every t_center."""
    t_query, t_center, t_max = 16000 * x_query, 16000 * x_center, 16000 * x_max
    audio_pad = np.pad(audio, (WINDOW // 2, WINDOW // 2), mode="reflect")
    opt_ts = []
    if audio_pad.shape[0] > t_max:
        audio_sum = np.zeros_like(audio)
        for i in range(WINDOW):
            audio_sum += audio_pad[i: i - WINDOW]
        for t in range(t_center, audio.shape[0], t_center):
            seg = np.abs(audio_sum[t - t_query: t + t_query])
            opt_ts.append(t - t_query + np.where(seg == seg.min())[0][0])
    return opt_ts


def segment_plan(n_audio: int, opt_ts, x_pad=X_PAD):
    """pipeline.py:578-680 integer bookkeeping.  Returns per segment
    (start, stop) in padded-audio samples and (p0, p1) pitch-frame slice bounds (None = open)."""
    t_pad = 16000 * x_pad
    t_pad2 = 2 * t_pad
    n_pad = n_audio + 2 * t_pad
    segs = []
    s = 0
    t = None
    for t in opt_ts:
        t = t // WINDOW * WINDOW
        segs.append((s, t + t_pad2 + WINDOW, s // WINDOW, (t + t_pad2) // WINDOW))
        s = t
    if t is None:
        segs.append((0, n_pad, 0, None))
    else:
        segs.append((t, n_pad, t // WINDOW, None))
    return segs


def frame_counts(n_seg: int):
    """HuBERT frames F=(n-400)//320+1 (7 strided convs), synth frames T=min(n//160, 2F) (pipeline.py:467)."""
    f = (n_seg - 400) // 320 + 1
    return f, min(n_seg // WINDOW, 2 * f)


# ----------------------------------------------------------------------------------------------
# F0: RMVPE log-mel, network, decode, coarse quantisation
# ----------------------------------------------------------------------------------------------


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(sr=16000, n_fft=1024, n_mels=128, fmin=30.0, fmax=8000.0) -> np.ndarray:
    """librosa.filters.mel(htk=True, norm='slaney') as called at RMVPE.py:371-378
    (librosa 0.11 algorithm; the reference vendors an identical clone at
    rvc/lib/predictors/torchfcpe/mel_fn_librosa.py:8)."""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz_htk(np.linspace(hz_to_mel_htk(fmin), hz_to_mel_htk(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


_MEL_BASIS = None


def logmel_rmvpe(audio: Tensor) -> Tensor:
    """RMVPE.py:388-417 at the :438 parameters: audio [B,n] f32 -> log-mel [B,128,1+n//160]."""
    global _MEL_BASIS
    if _MEL_BASIS is None:
        _MEL_BASIS = torch.from_numpy(mel_filterbank()).float()
    win = torch.hann_window(1024)
    fft = torch.stft(audio, n_fft=1024, hop_length=160, win_length=1024, window=win, center=True,
                     return_complex=True)
    mag = torch.sqrt(fft.real.pow(2) + fft.imag.pow(2))
    return torch.log(torch.clamp(torch.matmul(_MEL_BASIS, mag), min=1e-5))


def _bn_eval(x, sd, p, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _conv_block_res(x, sd, p):
    """RMVPE.py:13-57"""
    y = F.conv2d(x, sd[p + ".conv.0.weight"], None, 1, 1)
    y = F.relu(_bn_eval(y, sd, p + ".conv.1"))
    y = F.conv2d(y, sd[p + ".conv.3.weight"], None, 1, 1)
    y = F.relu(_bn_eval(y, sd, p + ".conv.4"))
    if p + ".shortcut.weight" in sd:
        return y + F.conv2d(x, sd[p + ".shortcut.weight"], sd[p + ".shortcut.bias"])
    return y + x


def rmvpe_e2e(mel: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """E2E(4,1,(2,2)).forward, RMVPE.py:335-339 (DeepUnet :246-286, BiGRU :515-536). mel [B,128,T32]."""
    x = mel.transpose(-1, -2).unsqueeze(1)
    x = _bn_eval(x, sd, "unet.encoder.bn")
    skips = []
    for i in range(5):
        for m in range(4):
            x = _conv_block_res(x, sd, f"unet.encoder.layers.{i}.conv.{m}")
        skips.append(x)
        x = F.avg_pool2d(x, (2, 2))
    for i in range(4):
        for m in range(4):
            x = _conv_block_res(x, sd, f"unet.intermediate.layers.{i}.conv.{m}")
    for i in range(5):
        p = f"unet.decoder.layers.{i}"
        x = F.conv_transpose2d(x, sd[p + ".conv1.0.weight"], None, stride=(2, 2), padding=(1, 1),
                               output_padding=(1, 1))
        x = F.relu(_bn_eval(x, sd, p + ".conv1.1"))
        x = torch.cat((x, skips[-1 - i]), dim=1)
        for m in range(4):
            x = _conv_block_res(x, sd, f"{p}.conv2.{m}")
    x = F.conv2d(x, sd["cnn.weight"], sd["cnn.bias"], 1, 1)
    x = x.transpose(1, 2).flatten(-2)  # [B,T,384]
    # nn.GRU(384, 256, bidirectional, batch_first) evaluated functionally (constructing an nn.GRU here
    # would draw its init values from the global generator and shift the reference's noise stream)
    flat = [sd["fc.0.gru." + n + sfx] for sfx in ("", "_reverse")
            for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    h0 = torch.zeros(2, x.shape[0], 256, dtype=x.dtype)
    x = torch._VF.gru(x, h0, flat, True, 1, 0.0, False, True, True)[0]
    return torch.sigmoid(F.linear(x, sd["fc.1.weight"], sd["fc.1.bias"]))


def rmvpe_mel2hidden(mel: Tensor, sd) -> Tensor:
    """RMVPE.py:444-457"""
    n_frames = mel.shape[-1]
    mel = F.pad(mel, (0, 32 * ((n_frames - 1) // 32 + 1) - n_frames), mode="reflect")
    with torch.no_grad():
        return rmvpe_e2e(mel, sd)[:, :n_frames]


CENTS_MAPPING = np.pad(20 * np.arange(360) + 1997.3794084376191, (4, 4))  # RMVPE.py:441-442


def rmvpe_decode(hidden: np.ndarray, thred=0.03) -> np.ndarray:
    """RMVPE.py:459-512 (per-frame python loop replaced by an index gather with the same
    [T,9] operands, so the NumPy reductions see identical arrays)."""
    center = np.argmax(hidden, axis=1)
    sal = np.pad(hidden, ((0, 0), (4, 4)))
    center = center + 4
    idx = (center - 4)[:, None] + np.arange(9)[None, :]
    todo_sal = np.take_along_axis(sal, idx, axis=1)
    todo_map = CENTS_MAPPING[idx]
    product_sum = np.sum(todo_sal * todo_map, 1)
    weight_sum = np.sum(todo_sal, 1)
    cents = product_sum / weight_sum
    cents[np.max(sal, axis=1) <= thred] = 0
    f0 = 10 * (2 ** (cents / 1200))
    f0[f0 == 10] = 0
    return f0


def rmvpe_infer_from_audio(audio: np.ndarray, sd, thred=0.03, taps=None) -> np.ndarray:
    """RMVPE.py:472-485"""
    a = torch.from_numpy(audio).float().unsqueeze(0)
    hidden = rmvpe_mel2hidden(logmel_rmvpe(a), sd)
    if taps is not None:
        taps["salience"] = hidden.squeeze(0).numpy()
    return rmvpe_decode(hidden.squeeze(0).numpy(), thred)


F0_MEL_MIN = 1127 * np.log(1 + 50 / 700)  # pipeline.py:144-147
F0_MEL_MAX = 1127 * np.log(1 + 1100 / 700)


# pipeline.py:149-204: G1 .. C6, two decimals
REF_FREQS = [49.00, 51.91, 55.00, 58.27, 61.74, 65.41, 69.30, 73.42, 77.78, 82.41, 87.31, 92.50, 98.00, 103.83, 110.00,
             116.54, 123.47, 130.81, 138.59, 146.83, 155.56, 164.81, 174.61, 185.00, 196.00, 207.65, 220.00, 233.08,
             246.94, 261.63, 277.18, 293.66, 311.13, 329.63, 349.23, 369.99, 392.00, 415.30, 440.00, 466.16, 493.88,
             523.25, 554.37, 587.33, 622.25, 659.25, 698.46, 739.99, 783.99, 830.61, 880.00, 932.33, 987.77, 1046.50]


def autotune_f0(f0: np.ndarray, strength: float) -> np.ndarray:
    """Autotune.autotune_f0, pipeline.py:103-114: every frame -- unvoiced zeros included -- is pulled towards the
    nearest listed note (first one wins a tie, as Python's min does)."""
    out = np.zeros_like(f0)
    for i, freq in enumerate(f0):
        closest = min(REF_FREQS, key=lambda x: abs(x - freq))
        out[i] = freq + (closest - freq) * strength
    return out


def librosa_rms(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """librosa.feature.rms (librosa 0.11.0 as pinned by requirements.txt; absent here -> restated from its published
    algorithm, parity unpinned): centre-pad with zeros, frame, sqrt(mean(|x|^2)).  Returns [1, n_frames]."""
    y = np.pad(y, (frame_length // 2, frame_length // 2), mode="constant")
    n_frames = 1 + (y.shape[0] - frame_length) // hop_length
    frames = np.lib.stride_tricks.as_strided(y, shape=(frame_length, n_frames),
                                             strides=(y.strides[0], y.strides[0] * hop_length), writeable=False)
    return np.sqrt(np.mean(np.abs(frames) ** 2, axis=-2, keepdims=True))


def change_rms(source_audio: np.ndarray, source_rate: int, target_audio: np.ndarray, target_rate: int, rate: float):
    """AudioProcessor.change_rms, pipeline.py:38-85 (the caller passes 16000 for BOTH rates, :682-685)."""
    rms1 = librosa_rms(source_audio, source_rate // 2 * 2, source_rate // 2)
    rms2 = librosa_rms(target_audio, target_rate // 2 * 2, target_rate // 2)
    rms1 = F.interpolate(torch.from_numpy(rms1).float().unsqueeze(0), size=target_audio.shape[0], mode="linear").squeeze()
    rms2 = F.interpolate(torch.from_numpy(rms2).float().unsqueeze(0), size=target_audio.shape[0], mode="linear").squeeze()
    rms2 = torch.maximum(rms2, torch.zeros_like(rms2) + 1e-6)
    return target_audio * (torch.pow(rms1, 1 - rate) * torch.pow(rms2, rate - 1)).numpy()


def f0_to_coarse(f0: np.ndarray, pitch: float = 0, f0_autotune: bool = False, f0_autotune_strength: float = 1):
    """pipeline.py:385-388, 401-410.  Returns (coarse int array 1..255, shifted f0)."""
    if f0_autotune is True:
        f0 = autotune_f0(f0, f0_autotune_strength)
    f0 = f0 * pow(2, pitch / 12)
    f0bak = f0.copy()
    f0_mel = 1127 * np.log(1 + f0 / 700)
    f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - F0_MEL_MIN) * 254 / (F0_MEL_MAX - F0_MEL_MIN) + 1
    f0_mel[f0_mel <= 1] = 1
    f0_mel[f0_mel > 255] = 255
    return np.rint(f0_mel).astype(int), f0bak


# ----------------------------------------------------------------------------------------------
# HuBERT-base (transformers.HubertModel; call site pipeline.py:450)
# ----------------------------------------------------------------------------------------------


def hubert_forward(sd: Dict[str, Tensor], wav: Tensor, n_layers=12, n_heads=12, consume_layerdrop_rng=True) -> Tensor:
    """transformers HubertModel.forward in eval mode with the default HubertConfig()
    (feat_extract_norm='group', conv_bias=False, do_stable_layer_norm=False, gelu):
    wav [B,n] -> last_hidden_state [B,F,768].

    transformers' HubertEncoder draws ``torch.rand([])`` once per layer for LayerDrop even in
    eval mode (the draw is ignored but advances the CPU generator), so the reference's
    Synthesizer noise (pipeline.py:486 -> synthesizers.py:245) starts 12 draws into the stream.
    ``consume_layerdrop_rng`` reproduces that so equal seeds give equal noise."""
    w = fold_weight_norm(sd)
    x = wav[:, None, :]
    strides = (5, 2, 2, 2, 2, 2, 2)
    for i, s in enumerate(strides):
        x = F.conv1d(x, w[f"feature_extractor.conv_layers.{i}.conv.weight"], None, stride=s)
        if i == 0:
            x = F.group_norm(x, x.shape[1], w["feature_extractor.conv_layers.0.layer_norm.weight"],
                             w["feature_extractor.conv_layers.0.layer_norm.bias"], 1e-5)
        x = F.gelu(x)
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), w["feature_projection.layer_norm.weight"],
                     w["feature_projection.layer_norm.bias"], 1e-5)
    x = F.linear(x, w["feature_projection.projection.weight"], w["feature_projection.projection.bias"])
    pos = F.conv1d(x.transpose(1, 2), w["encoder.pos_conv_embed.conv.weight"], w["encoder.pos_conv_embed.conv.bias"],
                   padding=64, groups=16)
    pos = F.gelu(pos[:, :, :-1]).transpose(1, 2)
    x = x + pos
    x = F.layer_norm(x, (768,), w["encoder.layer_norm.weight"], w["encoder.layer_norm.bias"], 1e-5)
    B, T, D = x.shape
    hd = D // n_heads
    for i in range(n_layers):
        L = f"encoder.layers.{i}"
        if consume_layerdrop_rng:
            torch.rand([])
        q = F.linear(x, w[L + ".attention.q_proj.weight"], w[L + ".attention.q_proj.bias"]) * hd ** -0.5
        k = F.linear(x, w[L + ".attention.k_proj.weight"], w[L + ".attention.k_proj.bias"])
        v = F.linear(x, w[L + ".attention.v_proj.weight"], w[L + ".attention.v_proj.bias"])
        q = q.view(B, T, n_heads, hd).transpose(1, 2)
        k = k.view(B, T, n_heads, hd).transpose(1, 2)
        v = v.view(B, T, n_heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, T, D)
        a = F.linear(a, w[L + ".attention.out_proj.weight"], w[L + ".attention.out_proj.bias"])
        x = F.layer_norm(x + a, (D,), w[L + ".layer_norm.weight"], w[L + ".layer_norm.bias"], 1e-5)
        h = F.gelu(F.linear(x, w[L + ".feed_forward.intermediate_dense.weight"],
                            w[L + ".feed_forward.intermediate_dense.bias"]))
        h = F.linear(h, w[L + ".feed_forward.output_dense.weight"], w[L + ".feed_forward.output_dense.bias"])
        x = F.layer_norm(x + h, (D,), w[L + ".final_layer_norm.weight"], w[L + ".final_layer_norm.bias"], 1e-5)
    return x


# ----------------------------------------------------------------------------------------------
# kNN retrieval (faiss IndexFlat-style exact squared-L2; pipeline.py:497-507)
# ----------------------------------------------------------------------------------------------


def knn_search(big_npy: np.ndarray, q: np.ndarray, k: int = 8, dtype=np.float64, chunk: int = 65536):
    """Exact brute-force ``index.search(q, k)``: squared L2, ascending, ties -> lower id first.
    dtype=float64 is the ground truth; float32 mimics faiss' ||q||^2 - 2 q.x + ||x||^2 BLAS path."""
    q = np.ascontiguousarray(q, dtype=dtype)
    nq = q.shape[0]
    qn = (q * q).sum(1)
    best_d = np.full((nq, k), np.inf, dtype=dtype)
    best_i = np.full((nq, k), -1, dtype=np.int64)
    for s in range(0, big_npy.shape[0], chunk):
        x = np.ascontiguousarray(big_npy[s: s + chunk], dtype=dtype)
        d = qn[:, None] - 2.0 * (q @ x.T) + (x * x).sum(1)[None, :]
        np.maximum(d, 0, out=d)
        if d.shape[1] > 4 * k:
            # The chunk's k best per query by selection instead of a full sort (a 2 M-row index is 31 chunks of 1599 x 65536
            # distances: sorting them all was 230 s of the GPU suite).  Same result as the stable sort below, ties included:
            # a row whose k-th smallest value is shared with entries left outside the selection (equal distances at the
            # cut) is re-done with the stable sort, so "ties -> lower id first" holds exactly.
            part = np.argpartition(d, k - 1, axis=1)[:, :k]
            part.sort(axis=1)                                            # ascending ids: the stable sort below then breaks ties by id
            pd = np.take_along_axis(d, part, axis=1)
            kth = pd.max(axis=1)
            ambiguous = np.nonzero((d <= kth[:, None]).sum(axis=1) > k)[0]
            for r in ambiguous:
                part[r] = np.sort(np.argsort(d[r], kind="stable")[:k])
            pd = np.take_along_axis(d, part, axis=1)
            d, ids = pd, part.astype(np.int64) + s
        else:
            ids = np.broadcast_to(np.arange(s, s + x.shape[0], dtype=np.int64), d.shape)
        cat_d = np.concatenate([best_d, d], axis=1)
        cat_i = np.concatenate([best_i, ids], axis=1)
        order = np.argsort(cat_d, axis=1, kind="stable")[:, :k]
        best_d = np.take_along_axis(cat_d, order, axis=1)
        best_i = np.take_along_axis(cat_i, order, axis=1)
    return best_d.astype(np.float32), best_i


def ivf_search(centroids: np.ndarray, list_ids, big_npy: np.ndarray, q: np.ndarray, k: int = 8, nprobe: int = 1):
    """faiss IndexIVFFlat.search as the reference configures it (extract_index.py:62-64: "IVF{n},Flat", nprobe = 1;
    call site pipeline.py:499): the nprobe nearest coarse centroids of each query, then exact squared L2 over the members
    of those inverted lists only; fewer than k members -> id -1, distance +inf.  float64; ties -> lower id.
    faiss itself is absent here: restated from its published algorithm, parity unpinned."""
    q64, c64 = q.astype(np.float64), centroids.astype(np.float64)
    dc = (q64 ** 2).sum(1)[:, None] - 2 * q64 @ c64.T + (c64 ** 2).sum(1)[None, :]
    probe = np.argsort(dc, axis=1, kind="stable")[:, :nprobe]
    d2 = np.full((q.shape[0], k), np.inf)
    ids = np.full((q.shape[0], k), -1, dtype=np.int64)
    for i in range(q.shape[0]):
        members = np.concatenate([np.asarray(list_ids[j], dtype=np.int64) for j in probe[i]])
        if members.size == 0:
            continue
        d = ((q64[i][None, :] - big_npy[members].astype(np.float64)) ** 2).sum(1)
        order = np.lexsort((members, d))[:k]
        d2[i, :order.size], ids[i, :order.size] = d[order], members[order]
    return d2, ids


def knn_blend(feats: np.ndarray, score: np.ndarray, ix: np.ndarray, big_npy: np.ndarray, index_rate: float):
    """pipeline.py:500-506 on host arrays: w = (1/d^2)^2 normalised; sum_k w*x[ix]; blend."""
    weight = np.square(1 / score)
    weight /= weight.sum(axis=1, keepdims=True)
    npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
    return npy * index_rate + (1 - index_rate) * feats


def retrieve_speaker_embeddings(feats: Tensor, big_npy: np.ndarray, index_rate: float, dtype=np.float64):
    """pipeline.py:497-507 with the exact-search stand-in for faiss."""
    npy = feats[0].numpy()
    score, ix = knn_search(big_npy, npy, 8, dtype)
    weight = np.square(1 / score)
    weight /= weight.sum(axis=1, keepdims=True)
    npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
    return torch.from_numpy(npy).unsqueeze(0) * index_rate + (1 - index_rate) * feats, score, ix


# ----------------------------------------------------------------------------------------------
# Synthesizer: TextEncoder, flow, decoders   (rvc/lib/algorithm/**)
# ----------------------------------------------------------------------------------------------


def _channel_layer_norm(x, gamma, beta, eps=1e-5):
    """normalization.py:19-26"""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


def _rel_attention(x, w, p, n_heads=2, window=10, mask=None):
    """attentions.py:79-180 (MultiHeadAttention with window-10 relative keys/values).
    The zero-padded [2T-1] relative tables of :143-153 are applied in banded form:
    only |j-i| <= window carries a non-zero embedding."""
    q = F.conv1d(x, w[p + ".conv_q.weight"], w[p + ".conv_q.bias"])
    k = F.conv1d(x, w[p + ".conv_k.weight"], w[p + ".conv_k.bias"])
    v = F.conv1d(x, w[p + ".conv_v.weight"], w[p + ".conv_v.bias"])
    b, d, t = q.shape
    kc = d // n_heads
    q = q.view(b, n_heads, kc, t).transpose(2, 3)
    k = k.view(b, n_heads, kc, t).transpose(2, 3)
    v = v.view(b, n_heads, kc, t).transpose(2, 3)
    qs = q / math.sqrt(kc)
    scores = torch.matmul(qs, k.transpose(-2, -1))
    ek, ev = w[p + ".emb_rel_k"][0], w[p + ".emb_rel_v"][0]  # [21, kc]
    rel_logits = torch.matmul(qs, ek.t())  # [b,h,t,21]; entry r <-> offset r - window
    ar = torch.arange(t)
    for r in range(2 * window + 1):
        off = r - window
        i0, i1 = max(0, -off), min(t, t - off)
        if i0 >= i1:
            continue
        ii = ar[i0:i1]
        scores[:, :, ii, ii + off] += rel_logits[:, :, i0:i1, r]
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    band = torch.zeros(b, n_heads, t, 2 * window + 1, dtype=x.dtype)
    for r in range(2 * window + 1):
        off = r - window
        i0, i1 = max(0, -off), min(t, t - off)
        if i0 >= i1:
            continue
        ii = ar[i0:i1]
        band[:, :, i0:i1, r] = p_attn[:, :, ii, ii + off]
    out = out + torch.matmul(band, ev)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return F.conv1d(out, w[p + ".conv_o.weight"], w[p + ".conv_o.bias"])


def text_encoder(w, phone: Tensor, pitch: Optional[Tensor], lengths: Tensor, n_layers=6, hidden=192, out_ch=192):
    """encoders.py:128-144 + Encoder :72-85 + FFN attentions.py:221-243"""
    x = F.linear(phone, w["enc_p.emb_phone.weight"], w["enc_p.emb_phone.bias"])
    if pitch is not None:
        x = x + F.embedding(pitch, w["enc_p.emb_pitch.weight"])
    x = x * math.sqrt(hidden)
    x = F.leaky_relu(x, 0.1)
    x = x.transpose(1, -1)
    t = x.size(2)
    x_mask = (torch.arange(t)[None, :] < lengths[:, None]).unsqueeze(1).to(x.dtype)
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        y = _rel_attention(x, w, f"enc_p.encoder.attn_layers.{i}", mask=attn_mask)
        x = _channel_layer_norm(x + y, w[f"enc_p.encoder.norm_layers_1.{i}.gamma"], w[f"enc_p.encoder.norm_layers_1.{i}.beta"])
        f = f"enc_p.encoder.ffn_layers.{i}"
        y = F.conv1d(F.pad(x * x_mask, (1, 1)), w[f + ".conv_1.weight"], w[f + ".conv_1.bias"])
        y = torch.relu(y)
        y = F.conv1d(F.pad(y * x_mask, (1, 1)), w[f + ".conv_2.weight"], w[f + ".conv_2.bias"]) * x_mask
        x = _channel_layer_norm(x + y, w[f"enc_p.encoder.norm_layers_2.{i}.gamma"], w[f"enc_p.encoder.norm_layers_2.{i}.beta"])
    x = x * x_mask
    stats = F.conv1d(x, w["enc_p.proj.weight"], w["enc_p.proj.bias"]) * x_mask
    m, logs = torch.split(stats, out_ch, dim=1)
    return m, logs, x_mask


def _wavenet(x, x_mask, g, w, p, hidden=192, n_layers=3, k=5):
    """modules.py:78-109 (dilation_rate 1) with commons.py:142-157 gate."""
    output = torch.zeros_like(x)
    gc = F.conv1d(g, w[p + ".cond_layer.weight"], w[p + ".cond_layer.bias"])
    for i in range(n_layers):
        x_in = F.conv1d(x, w[f"{p}.in_layers.{i}.weight"], w[f"{p}.in_layers.{i}.bias"], padding=(k - 1) // 2)
        in_act = x_in + gc[:, i * 2 * hidden: (i + 1) * 2 * hidden, :]
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        rs = F.conv1d(acts, w[f"{p}.res_skip_layers.{i}.weight"], w[f"{p}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def flow_reverse(w, z_p: Tensor, x_mask: Tensor, g: Tensor, half=96):
    """residuals.py:157-170 (reverse), :239-264 (mean-only coupling), :100-106 (Flip)."""
    x = z_p
    for n in (6, 4, 2, 0):
        x = torch.flip(x, [1])
        p = f"flow.flows.{n}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, w[p + ".pre.weight"], w[p + ".pre.bias"]) * x_mask
        h = _wavenet(h, x_mask, g, w, p + ".enc")
        m = F.conv1d(h, w[p + ".post.weight"], w[p + ".post.bias"]) * x_mask
        x1 = (x1 - m) * x_mask  # logs == 0 -> exp(-logs) == 1
        x = torch.cat([x0, x1], 1)
    return x


def stride_f0s(rates):
    return [int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else 1 for i in range(len(rates))]


def noise_conv_geometry(stride):
    """hifigan_nsf.py:142-144"""
    kernel = 1 if stride == 1 else stride * 2 - stride % 2
    padding = 0 if stride == 1 else (kernel - stride) // 2
    return kernel, padding


def ups_padding(u, k):
    """hifigan_nsf.py:113-117,127"""
    return ((k - u) // 2 if u % 2 == 0 else u // 2 + u % 2), u % 2


def source_nsf(f0: Tensor, upp: int, sr: int, lin_w: Tensor, lin_b: Tensor, noise) -> Tensor:
    """hifigan.py:156-228 + hifigan_nsf.py:48-52. f0 [B,T] -> har_source [B,1,T*upp]."""
    f0 = f0.unsqueeze(-1)
    b, t, _ = f0.shape
    grid = torch.arange(1, upp + 1, dtype=f0.dtype)
    phase = (f0 / sr) * grid  # [B,T,upp]
    rem = torch.fmod(phase[:, :-1, -1:] + 0.5, 1.0) - 0.5
    cum = rem.cumsum(dim=1).fmod(1.0).to(f0.dtype)
    phase = phase + F.pad(cum, (0, 0, 1, 0), mode="constant")
    phase = phase.reshape(b, -1, 1)
    phase = phase * torch.arange(1, 2, dtype=f0.dtype).reshape(1, 1, -1)
    rnd = noise.rand(1, 1, 1)
    rnd[..., 0] = 0
    phase = phase + rnd
    sine = torch.sin(2 * np.pi * phase) * 0.1
    uv = (f0 > 0).float()
    uv = F.interpolate(uv.transpose(2, 1), scale_factor=float(upp), mode="nearest").transpose(2, 1)
    amp = uv * 0.003 + (1 - uv) * (0.1 / 3)
    sine = sine * uv + amp * noise.randn(*sine.shape)
    return torch.tanh(F.linear(sine, lin_w, lin_b)).transpose(1, 2)


def _sine_mrf(f0_up: Tensor, sr: int, dim: int, noise):
    """hifigan_mrf.py:129-175 / refinegan.py:220-260. f0_up [B,L,1] -> [B,L,dim]."""
    f0_buf = torch.zeros(f0_up.shape[0], f0_up.shape[1], dim)
    f0_buf[:, :, 0] = f0_up[:, :, 0]
    for idx in range(dim - 1):
        f0_buf[:, :, idx + 1] = f0_buf[:, :, 0] * (idx + 2)
    rad = (f0_buf / sr) % 1
    rand_ini = noise.rand(f0_buf.shape[0], f0_buf.shape[2])
    rand_ini[:, 0] = 0
    rad[:, 0, :] = rad[:, 0, :] + rand_ini
    tmp = torch.cumsum(rad, 1) % 1
    over = (tmp[:, 1:, :] - tmp[:, :-1, :]) < 0
    shift = torch.zeros_like(rad)
    shift[:, 1:, :] = over * -1.0
    sines = torch.sin(torch.cumsum(rad + shift, dim=1) * 2 * np.pi) * 0.1
    uv = (f0_up > 0).to(f0_up.dtype)
    amp = uv * 0.003 + (1 - uv) * 0.1 / 3
    return sines * uv + amp * noise.randn(*sines.shape)


def _resblock(x, w, names, k, dil=(1, 3, 5), slope=0.1):
    """residuals.py:75-86 / hifigan_mrf.py:45-50,76-79 / refinegan.py:72-80"""
    for (c1, c2), d in zip(names, dil):
        y = F.leaky_relu(x, slope)
        y = F.conv1d(y, w[c1 + ".weight"], w[c1 + ".bias"], padding=(k * d - d) // 2, dilation=d)
        y = F.leaky_relu(y, slope)
        y = F.conv1d(y, w[c2 + ".weight"], w[c2 + ".bias"], padding=(k - 1) // 2)
        x = y + x
    return x


def decoder_nsf(w, x: Tensor, f0: Tensor, g: Tensor, rates, ksizes, sr: int, noise, taps=None) -> Tensor:
    """HiFiGANNSFGenerator.forward, hifigan_nsf.py:173-207."""
    upp = int(np.prod(rates))
    har = source_nsf(f0, upp, sr, w["dec.m_source.l_linear.weight"], w["dec.m_source.l_linear.bias"], noise)
    if taps is not None:
        taps["har_source"] = har
    x = F.conv1d(x, w["dec.conv_pre.weight"], w["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, w["dec.cond.weight"], w["dec.cond.bias"])
    strides = stride_f0s(rates)
    for i, (u, k) in enumerate(zip(rates, ksizes)):
        x = F.leaky_relu(x, 0.1)
        pad, opad = ups_padding(u, k)
        x = F.conv_transpose1d(x, w[f"dec.ups.{i}.weight"], w[f"dec.ups.{i}.bias"], stride=u, padding=pad,
                               output_padding=opad)
        nk, npad = noise_conv_geometry(strides[i])
        x = x + F.conv1d(har, w[f"dec.noise_convs.{i}.weight"], w[f"dec.noise_convs.{i}.bias"], stride=strides[i],
                         padding=npad)
        if taps is not None:
            taps[f"ups{i}"] = x
        xs = None
        for m, kk in enumerate((3, 7, 11)):
            r = f"dec.resblocks.{i * 3 + m}"
            names = [(f"{r}.convs1.{j}", f"{r}.convs2.{j}") for j in range(3)]
            y = _resblock(x, w, names, kk)
            xs = y if xs is None else xs + y
        x = xs / 3
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x)
    return torch.tanh(F.conv1d(x, w["dec.conv_post.weight"], None, padding=3))


def decoder_mrf(w, x: Tensor, f0: Tensor, g: Tensor, rates, ksizes, sr: int, noise, taps=None) -> Tensor:
    """HiFiGANMRFGenerator.forward, hifigan_mrf.py:339-366."""
    upp = int(np.prod(rates))
    f0_up = F.interpolate(f0[:, None, :], scale_factor=float(upp), mode="nearest").transpose(-1, -2)
    sine = _sine_mrf(f0_up, sr, 9, noise)
    har = torch.tanh(F.linear(sine, w["dec.m_source.l_linear.weight"], w["dec.m_source.l_linear.bias"])).transpose(-1, -2)
    if taps is not None:
        taps["har_source"] = har
    x = F.conv1d(x, w["dec.conv_pre.weight"], w["dec.conv_pre.bias"], padding=3)
    x = x + F.conv1d(g, w["dec.cond.weight"], w["dec.cond.bias"])
    strides = stride_f0s(rates)
    for i, (u, k) in enumerate(zip(rates, ksizes)):
        x = F.leaky_relu(x, 0.1)
        pad, opad = ups_padding(u, k)
        x = F.conv_transpose1d(x, w[f"dec.upsamples.{i}.weight"], w[f"dec.upsamples.{i}.bias"], stride=u, padding=pad,
                               output_padding=opad)
        nk, npad = noise_conv_geometry(strides[i])
        x = x + F.conv1d(har, w[f"dec.noise_convs.{i}.weight"], w[f"dec.noise_convs.{i}.bias"], stride=strides[i],
                         padding=npad)
        xs = None
        for m, kk in enumerate((3, 7, 11)):
            r = f"dec.mrfs.{i}.{m}"
            names = [(f"{r}.layers.{j}.conv1", f"{r}.layers.{j}.conv2") for j in range(3)]
            y = _resblock(x, w, names, kk)
            xs = y if xs is None else xs + y
        x = xs / 3
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x)
    return torch.tanh(F.conv1d(x, w["dec.conv_post.weight"], w["dec.conv_post.bias"], padding=3))


def decoder_refine(w, mel: Tensor, f0: Tensor, g: Tensor, rates, sr: int, noise, taps=None) -> Tensor:
    """RefineGANGenerator.forward, refinegan.py:368-405."""
    upp = int(np.prod(rates))
    slope = 0.2
    T = mel.shape[-1]
    f0_up = F.interpolate(f0.unsqueeze(1), size=T * upp, mode="linear")
    sine = _sine_mrf(f0_up.transpose(1, 2), sr, 1, noise)
    har = torch.tanh(F.linear(sine, w["dec.m_source.merge.0.weight"])).transpose(1, 2)
    if taps is not None:
        taps["har_source"] = har
    x = F.conv1d(har, w["dec.pre_conv.weight"], w["dec.pre_conv.bias"], padding=3)
    x = F.interpolate(x, size=T, mode="linear")
    mel = F.conv1d(mel, w["dec.mel_conv.weight"], w["dec.mel_conv.bias"], padding=3)
    mel = mel + F.conv1d(g, w["dec.cond.weight"], w["dec.cond.bias"])
    x = torch.cat([mel, x], dim=1)
    strides = stride_f0s(rates)
    for i, u in enumerate(rates):
        x = F.leaky_relu(x, slope)
        x = F.interpolate(x, scale_factor=float(u), mode="linear")
        nk, npad = noise_conv_geometry(strides[i])
        d = F.conv1d(har, w[f"dec.downsample_blocks.{i}.weight"], w[f"dec.downsample_blocks.{i}.bias"],
                     stride=strides[i], padding=npad)
        x = torch.cat([x, d], dim=1)
        p = f"dec.upsample_conv_blocks.{i}"
        x = F.conv1d(x, w[p + ".input_conv.weight"], w[p + ".input_conv.bias"], padding=3)
        ys = []
        for m, kk in enumerate((3, 7, 11)):
            b = f"{p}.blocks.{m}"
            y = F.leaky_relu(x + noise.randn(*x.shape) * w[b + ".0.weight"][None, :, None], slope)
            names = [(f"{b}.1.convs1.{j}", f"{b}.1.convs2.{j}") for j in range(3)]
            y = _resblock(y, w, names, kk, slope=slope)
            y = F.leaky_relu(y + noise.randn(*y.shape) * w[b + ".2.weight"][None, :, None], slope)
            ys.append(y)
        x = torch.stack(ys, dim=0).mean(dim=0)
        if taps is not None:
            taps[f"stage{i}"] = x
    x = F.leaky_relu(x, slope)
    return torch.tanh(F.conv1d(x, w["dec.conv_post.weight"], None, padding=3))


def synthesizer_infer(cpt: dict, phone: Tensor, phone_lengths: Tensor, pitch: Tensor, nsff0: Tensor, sid: Tensor,
                      noise=None, w=None, taps=None, rate=None):
    """Synthesizer.infer, synthesizers.py:223-260; `rate` (a float) drops the head of z_p / x_mask / nsff0 before the flow
    and the vocoder (:247-251).  Returns (o, x_mask, (z, z_p, m_p, logs_p))."""
    noise = noise or TorchNoise()
    if w is None:
        w = fold_weight_norm(cpt["weight"])
    cfg = cpt["config"]
    rates, ksizes, sr = cfg[12], cfg[14], cfg[17]
    vocoder = cpt.get("vocoder", "HiFi-GAN")
    with torch.no_grad():
        g = F.embedding(sid, w["emb_g.weight"]).unsqueeze(-1)
        m_p, logs_p, x_mask = text_encoder(w, phone, pitch, phone_lengths)
        z_p = (m_p + torch.exp(logs_p) * noise.randn(*m_p.shape) * 0.66666) * x_mask
        if rate is not None:   # synthesizers.py:247-251
            head = int(z_p.shape[2] * (1.0 - float(rate)))
            z_p, x_mask, nsff0 = z_p[:, :, head:], x_mask[:, :, head:], nsff0[:, head:]
        z = flow_reverse(w, z_p, x_mask, g)
        if vocoder == "MRF HiFi-GAN":
            o = decoder_mrf(w, z * x_mask, nsff0, g, rates, ksizes, sr, noise, taps)
        elif vocoder == "RefineGAN":
            o = decoder_refine(w, z * x_mask, nsff0, g, rates, sr, noise, taps)
        else:
            o = decoder_nsf(w, z * x_mask, nsff0, g, rates, ksizes, sr, noise, taps)
    return o, x_mask, (z, z_p, m_p, logs_p)


# ----------------------------------------------------------------------------------------------
# voice_conversion / pipeline   (rvc/infer/pipeline.py:412-495, 509-694)
# ----------------------------------------------------------------------------------------------


def voice_conversion(hubert_sd, cpt, w, sid: Tensor, audio0: np.ndarray, pitch: Tensor, pitchf: Tensor,
                     big_npy: Optional[np.ndarray], index_rate: float, protect: float, noise=None,
                     knn_dtype=np.float64, taps=None) -> np.ndarray:
    with torch.no_grad():
        feats = torch.from_numpy(audio0).float().view(1, -1)
        feats = hubert_forward(hubert_sd, feats)
        if cpt.get("version", "v1") == "v1":  # pipeline.py:451-453
            feats = F.linear(feats[0], hubert_sd["final_proj.weight"], hubert_sd["final_proj.bias"]).unsqueeze(0)
        feats0 = feats.clone()
        if big_npy is not None and index_rate > 0:
            feats, score, ix = retrieve_speaker_embeddings(feats, big_npy, index_rate, knn_dtype)
            if taps is not None:
                taps["knn_ids"], taps["knn_d2"] = ix, score
        feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
        p_len = min(audio0.shape[0] // WINDOW, feats.shape[1])
        feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
        pitch, pitchf = pitch[:, :p_len], pitchf[:, :p_len]
        if protect < 0.5:
            pitchff = pitchf.clone()
            pitchff[pitchf > 0] = 1
            pitchff[pitchf < 1] = protect
            feats = feats * pitchff.unsqueeze(-1) + feats0 * (1 - pitchff.unsqueeze(-1))
        if taps is not None:
            taps["feats"] = feats
        o = synthesizer_infer(cpt, feats.float(), torch.tensor([p_len]).long(), pitch, pitchf.float(), sid,
                              noise=noise, w=w, taps=taps)[0]
    return o[0, 0].float().numpy()


def pipeline(hubert_sd, rmvpe_sd, cpt, audio: np.ndarray, *, sid=0, pitch=0, big_npy=None, index_rate=0.0,
             protect=0.5, noise=None, knn_dtype=np.float64, taps=None, volume_envelope=1, f0_autotune=False,
             f0_autotune_strength=1, x_query=X_QUERY, x_center=X_CENTER, x_max=X_MAX, f0_override=None,
             dec_bf16=False) -> np.ndarray:
    """Pipeline.pipeline, pipeline.py:509-694, rmvpe branch.  x_query / x_center / x_max: the memory-tier constants of
    rvc/configs/config.py:116-121 that Pipeline.__init__ reads from its config object (pipeline.py:124-127).
    f0_override (test infrastructure, not a reference argument): the raw RMVPE contour to use instead of evaluating the
    network -- the tie-aware full-length tests hand over the product's contour when the two differ only on frames whose
    salience arg-max is a certified near-tie (taps["salience"] holds this side's salience for that certificate)."""
    tgt_sr = cpt["config"][-1]
    w = fold_weight_norm(cpt["weight"])
    if dec_bf16:   # BASELINE cfg 4: fp32 math on the vocoder weights a bf16 copy holds
        w = {k: (v.float().bfloat16().float() if k.startswith("dec.") else v) for k, v in w.items()}
    t_pad, t_pad_tgt = 16000 * X_PAD, tgt_sr * X_PAD
    audio = highpass(audio)
    opt_ts = split_points(audio, x_query, x_center, x_max)
    audio_pad = np.pad(audio, (t_pad, t_pad), mode="reflect")
    p_len = audio_pad.shape[0] // WINDOW
    sid_t = torch.tensor(sid).unsqueeze(0).long()
    f0 = rmvpe_infer_from_audio(audio_pad, rmvpe_sd, taps=taps) if f0_override is None else np.array(f0_override, dtype=np.float64)
    if taps is not None:
        taps["f0_raw"] = f0.copy()
    coarse, f0bak = f0_to_coarse(f0, pitch, f0_autotune, f0_autotune_strength)
    coarse, f0bak = coarse[:p_len], f0bak[:p_len]
    pitch_t = torch.tensor(coarse).unsqueeze(0).long()
    pitchf_t = torch.tensor(f0bak).unsqueeze(0).float()
    if taps is not None:
        taps["opt_ts"], taps["coarse"], taps["f0"] = list(opt_ts), coarse, f0bak
    out = []
    for (s0, s1, p0, p1) in segment_plan(audio.shape[0], opt_ts):
        seg = voice_conversion(hubert_sd, cpt, w, sid_t, audio_pad[s0:s1], pitch_t[:, p0:p1], pitchf_t[:, p0:p1],
                               big_npy, index_rate, protect, noise=noise, knn_dtype=knn_dtype, taps=taps)
        out.append(seg[t_pad_tgt:-t_pad_tgt])
    audio_opt = np.concatenate(out)
    if volume_envelope != 1:  # pipeline.py:682-685
        audio_opt = change_rms(audio, 16000, audio_opt, 16000, volume_envelope)
    audio_max = np.abs(audio_opt).max() / 0.99
    if audio_max > 1:
        audio_opt /= audio_max
    return audio_opt


# ----------------------------------------------------------------------------------------------
# training-side features: rvc/train/mel_processing.py, rvc/train/extract/extract.py
# ----------------------------------------------------------------------------------------------


def slaney_mel_basis(sr: int, n_fft: int, n_mels: int, fmin: float = 0.0, fmax=None) -> np.ndarray:
    """librosa.filters.mel with its defaults (htk=False, norm="slaney"), what mel_processing.py:113-115 asks for.  Third
    party and absent here: restated from librosa 0.11's published algorithm; pinned by the fixture basis_* of
    tests/golden/train_features.npz, which comes from the reference's own vendored clone of that function."""
    fmax = sr / 2 if fmax is None else fmax
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= min_log_hz / f_sp, min_log_hz * np.exp(logstep * (m - min_log_hz / f_sp)), f_sp * m)

    freqs = np.fft.rfftfreq(n_fft, 1.0 / sr)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff, ramps = np.diff(mel_f), np.subtract.outer(mel_f, freqs)
    w = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float32)
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w


def spectrogram(y: Tensor, n_fft: int, hop: int, win: int) -> Tensor:
    """spectrogram_torch, mel_processing.py:53-97 (center=False): reflect pad (n_fft - hop) / 2, periodic hann, |X| with
    the 1e-6 inside the square root."""
    pad = int((n_fft - hop) / 2)
    y = F.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft=n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win, dtype=y.dtype), center=False,
                      normalized=False, onesided=True, return_complex=True)
    return torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-6)


def mel_spectrogram(y: Tensor, n_fft: int, n_mels: int, sr: int, hop: int, win: int, fmin: float = 0.0, fmax=None) -> Tensor:
    """mel_spectrogram_torch, mel_processing.py:126-146: log(clamp(M |X|, 1e-5))."""
    basis = torch.from_numpy(slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax))
    return torch.log(torch.clamp(torch.matmul(basis, spectrogram(y, n_fft, hop, win)), min=1e-5))


def extract_coarse_f0(f0: np.ndarray) -> np.ndarray:
    """FeatureInput.coarse_f0, extract.py:76-87."""
    f0_mel = 1127.0 * np.log(1.0 + f0 / 700.0)
    f0_mel = np.clip((f0_mel - F0_MEL_MIN) * 254 / (F0_MEL_MAX - F0_MEL_MIN) + 1, 1, 255)
    return np.rint(f0_mel).astype(int)
