"""Deterministic synthetic checkpoints, inputs and feature indices.

There are no pretrained weights on the GPU box (no network), and the reference
itself cannot travel there, so every test / bench run builds the same weights
from a seed.  The tensors are produced in the *exported* formats the reference
consumes, so the product loaders are exercised exactly as with real files:

* synthesizer ``.pth`` dict  -- ``rvc/train/process/extract_model.py:56-107``
  (``weight`` with legacy ``.weight_g/.weight_v`` keys, 18-item ``config`` list,
  ``f0``, ``version``, ``sr``, ``vocoder``); consumed by
  ``rvc/infer/infer.py:464-485``.
* HuBERT-base state dict in ``transformers`` naming.
* RMVPE ``E2E(4, 1, (2, 2))`` state dict -- ``rvc/lib/predictors/RMVPE.py:289-339``.

Every tensor is drawn from its own ``numpy.random.default_rng`` stream keyed by
(seed, crc32(name)), so the values do not depend on generation order.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

# model hyper-parameters per sample rate: rvc/configs/48000.json, 40000.json, 32000.json
MODEL_CONFIGS = {
    48000: dict(filter_length=2048, upsample_rates=[12, 10, 2, 2], upsample_kernel_sizes=[24, 20, 4, 4]),
    40000: dict(filter_length=2048, upsample_rates=[10, 10, 2, 2], upsample_kernel_sizes=[16, 16, 4, 4]),
    32000: dict(filter_length=1024, upsample_rates=[10, 8, 2, 2], upsample_kernel_sizes=[20, 16, 4, 4]),
}


def config_list(sr: int, *, upsample_initial_channel: int = 512, spk_embed_dim: int = 109):
    """The 18-item positional ``config`` of an exported model (extract_model.py:61-80)."""
    c = MODEL_CONFIGS[sr]
    return [
        c["filter_length"] // 2 + 1,  # spec_channels
        32,  # segment_size
        192,  # inter_channels
        192,  # hidden_channels
        768,  # filter_channels
        2,  # n_heads
        6,  # n_layers
        3,  # kernel_size
        0,  # p_dropout
        "1",  # resblock
        [3, 7, 11],
        [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
        list(c["upsample_rates"]),
        upsample_initial_channel,
        list(c["upsample_kernel_sizes"]),
        spk_embed_dim,
        256,  # gin_channels
        sr,
    ]


class _Gen:
    def __init__(self, seed: int):
        self.seed = int(seed)
        self.out = OrderedDict()

    def _rng(self, name):
        return np.random.default_rng([self.seed, zlib.crc32(name.encode())])

    def normal(self, name, shape, std=1.0, mean=0.0):
        a = self._rng(name).standard_normal(tuple(shape), dtype=np.float32) * np.float32(std) + np.float32(mean)
        self.out[name] = torch.from_numpy(a.astype(np.float32))
        return self.out[name]

    def positive(self, name, shape, base=1.0, spread=0.1):
        a = np.float32(base) + np.float32(spread) * np.abs(self._rng(name).standard_normal(tuple(shape), dtype=np.float32))
        self.out[name] = torch.from_numpy(a.astype(np.float32))
        return self.out[name]

    # --- helpers for the layer families -------------------------------------------------
    def conv(self, prefix, cout, cin, k, *, bias=True, std=None, gain=0.5773503):
        fan_in = cin * k
        self.normal(prefix + ".weight", (cout, cin, k), std if std is not None else gain / math.sqrt(fan_in))
        if bias:
            self.normal(prefix + ".bias", (cout,), 0.5 / math.sqrt(fan_in))

    def wn_conv(self, prefix, shape, *, bias_len=None, std=0.01, norm_dim=0):
        """Weight-normed conv in the legacy export naming (.weight_g / .weight_v)."""
        v = self.normal(prefix + ".weight_v", shape, std)
        dims = [d for d in range(v.dim()) if d != norm_dim]
        nrm = v.norm(2, dim=dims, keepdim=True)
        jitter = 1.0 + 0.1 * self._rng(prefix + ".weight_g").standard_normal(tuple(nrm.shape), dtype=np.float32)
        self.out[prefix + ".weight_g"] = (nrm * torch.from_numpy(jitter.astype(np.float32))).float()
        if bias_len:
            fan_in = int(np.prod(shape[1:]))
            self.normal(prefix + ".bias", (bias_len,), 0.5 / math.sqrt(fan_in))

    def layer_norm(self, prefix, n, wname="weight", bname="bias"):
        self.normal(f"{prefix}.{wname}", (n,), 0.1, 1.0)
        self.normal(f"{prefix}.{bname}", (n,), 0.1)

    def linear(self, prefix, cout, cin, *, bias=True, std=None):
        self.normal(prefix + ".weight", (cout, cin), std if std is not None else 0.5773503 / math.sqrt(cin))
        if bias:
            self.normal(prefix + ".bias", (cout,), 0.5 / math.sqrt(cin))


# residual-branch gain of the synthetic vocoder weights: large enough that every dilated conv moves the
# waveform (the reference's init std 0.01 would leave the output dominated by the last noise conv)
RES_GAIN = 0.8


def _stride_f0s(rates):
    return [int(np.prod(rates[i + 1:])) if i + 1 < len(rates) else 1 for i in range(len(rates))]


def _noise_conv_geometry(stride):
    # hifigan_nsf.py:142-144
    kernel = 1 if stride == 1 else stride * 2 - stride % 2
    padding = 0 if stride == 1 else (kernel - stride) // 2
    return kernel, padding


def _text_encoder(g: _Gen, hidden=192, filt=768, inter=192, layers=6, k=3, emb=768, heads=2, window=10, smooth_pitch=False):
    # encoders.py:88-144, attentions.py:6-77
    g.linear("enc_p.emb_phone", hidden, emb)
    if smooth_pitch:
        # a trained pitch embedding varies smoothly with the coarse-pitch index (neighbouring bins are neighbouring
        # pitches); i.i.d. rows make ONE frame whose mel-scaled f0 rounds the other way at a .5 boundary swap in an
        # unrelated vector.  Rows = a random band-limited function of the index (12 harmonics over 256 bins), unit RMS.
        rng = g._rng("enc_p.emb_pitch.weight/smooth")
        kk = np.arange(1, 13)[:, None, None]
        idx = np.arange(256)[None, :, None]
        amp = rng.standard_normal((12, 1, hidden)) / np.sqrt(12.0 / 2.0)
        ph = rng.uniform(0, 2 * np.pi, (12, 1, hidden))
        tab = (amp * np.cos(2 * np.pi * kk * idx / 256.0 + ph)).sum(0)
        g.out["enc_p.emb_pitch.weight"] = torch.from_numpy(tab.astype(np.float32))
    else:
        g.normal("enc_p.emb_pitch.weight", (256, hidden), 1.0)
    kc = hidden // heads
    for i in range(layers):
        a = f"enc_p.encoder.attn_layers.{i}"
        g.normal(a + ".emb_rel_k", (1, 2 * window + 1, kc), kc ** -0.5)
        g.normal(a + ".emb_rel_v", (1, 2 * window + 1, kc), kc ** -0.5)
        for nm in ("conv_q", "conv_k", "conv_v", "conv_o"):
            g.conv(f"{a}.{nm}", hidden, hidden, 1, std=math.sqrt(2.0 / (2 * hidden)))
        g.layer_norm(f"enc_p.encoder.norm_layers_1.{i}", hidden, "gamma", "beta")
        g.conv(f"enc_p.encoder.ffn_layers.{i}.conv_1", filt, hidden, k)
        g.conv(f"enc_p.encoder.ffn_layers.{i}.conv_2", hidden, filt, k)
        g.layer_norm(f"enc_p.encoder.norm_layers_2.{i}", hidden, "gamma", "beta")
    g.conv("enc_p.proj", inter * 2, hidden, 1)


def _flow(g: _Gen, inter=192, hidden=192, gin=256, n_layers=3, k=5):
    # residuals.py:109-267, modules.py:5-76 ; flows at even indices, Flip at odd
    half = inter // 2
    for n in (0, 2, 4, 6):
        p = f"flow.flows.{n}"
        g.conv(p + ".pre", hidden, half, 1)
        g.wn_conv(p + ".enc.cond_layer", (2 * hidden * n_layers, gin, 1), bias_len=2 * hidden * n_layers,
                  std=0.5773503 / math.sqrt(gin))
        for j in range(n_layers):
            g.wn_conv(f"{p}.enc.in_layers.{j}", (2 * hidden, hidden, k), bias_len=2 * hidden,
                      std=0.5773503 / math.sqrt(hidden * k))
            rs = hidden if j == n_layers - 1 else 2 * hidden
            g.wn_conv(f"{p}.enc.res_skip_layers.{j}", (rs, hidden, 1), bias_len=rs,
                      std=0.5773503 / math.sqrt(hidden))
        # the reference zero-inits `post` (residuals.py:236-237); trained models do not keep it zero
        g.conv(p + ".post", half, hidden, 1)


def _dec_nsf(g: _Gen, rates, ksizes, init_ch, inter=192, gin=256, rk=(3, 7, 11), nd=3):
    # hifigan_nsf.py:75-171
    g.normal("dec.m_source.l_linear.weight", (1, 1), 0.2, 1.5)
    g.normal("dec.m_source.l_linear.bias", (1,), 0.05)
    g.conv("dec.conv_pre", init_ch, inter, 7)
    strides = _stride_f0s(rates)
    for i, (u, k) in enumerate(zip(rates, ksizes)):
        cin, cout = init_ch // 2 ** i, init_ch // 2 ** (i + 1)
        g.wn_conv(f"dec.ups.{i}", (cin, cout, k), bias_len=cout, std=1.0 / math.sqrt(cin * k / u))
        nk, _ = _noise_conv_geometry(strides[i])
        g.conv(f"dec.noise_convs.{i}", cout, 1, nk)
        for m, kk in enumerate(rk):
            for j in range(nd):
                g.wn_conv(f"dec.resblocks.{i * len(rk) + m}.convs1.{j}", (cout, cout, kk), bias_len=cout, std=RES_GAIN / math.sqrt(cout * kk))
                g.wn_conv(f"dec.resblocks.{i * len(rk) + m}.convs2.{j}", (cout, cout, kk), bias_len=cout, std=RES_GAIN / math.sqrt(cout * kk))
    g.conv("dec.conv_post", 1, init_ch // 2 ** len(rates), 7, bias=False)
    g.conv("dec.cond", init_ch, gin, 1)


def _dec_mrf(g: _Gen, rates, ksizes, init_ch, inter=192, gin=256, rk=(3, 7, 11), nd=3, harmonics=8):
    # hifigan_mrf.py:245-337
    g.linear("dec.m_source.l_linear", 1, harmonics + 1, std=0.4)
    g.wn_conv("dec.conv_pre", (init_ch, inter, 7), bias_len=init_ch, std=0.5773503 / math.sqrt(inter * 7))
    strides = _stride_f0s(rates)
    for i, (u, k) in enumerate(zip(rates, ksizes)):
        cin, cout = init_ch // 2 ** i, init_ch // 2 ** (i + 1)
        g.wn_conv(f"dec.upsamples.{i}", (cin, cout, k), bias_len=cout, std=1.0 / math.sqrt(cin * k / u))
        nk, _ = _noise_conv_geometry(strides[i])
        g.conv(f"dec.noise_convs.{i}", cout, 1, nk)
        for m, kk in enumerate(rk):
            for j in range(nd):
                g.wn_conv(f"dec.mrfs.{i}.{m}.layers.{j}.conv1", (cout, cout, kk), bias_len=cout, std=RES_GAIN / math.sqrt(cout * kk))
                g.wn_conv(f"dec.mrfs.{i}.{m}.layers.{j}.conv2", (cout, cout, kk), bias_len=cout, std=RES_GAIN / math.sqrt(cout * kk))
    cl = init_ch // 2 ** len(rates)
    g.wn_conv("dec.conv_post", (1, cl, 7), bias_len=1, std=0.5773503 / math.sqrt(cl * 7))
    g.conv("dec.cond", init_ch, gin, 1)


def _dec_refine(g: _Gen, rates, init_ch, inter=192, gin=256, rk=(3, 7, 11), nd=3):
    # refinegan.py:285-365
    g.normal("dec.m_source.merge.0.weight", (1, 1), 0.2, 1.5)
    g.wn_conv("dec.pre_conv", (init_ch // 2, 1, 7), bias_len=init_ch // 2, std=0.5773503 / math.sqrt(7))
    strides = _stride_f0s(rates)
    ch = init_ch
    for i, _u in enumerate(rates):
        nk, _ = _noise_conv_geometry(strides[i])
        co = init_ch // 2 ** (i + 2)
        g.wn_conv(f"dec.downsample_blocks.{i}", (co, 1, nk), bias_len=co, std=0.5773503 / math.sqrt(nk))
    g.wn_conv("dec.mel_conv", (init_ch // 2, inter, 7), bias_len=init_ch // 2, std=0.5773503 / math.sqrt(inter * 7))
    g.conv("dec.cond", init_ch // 2, gin, 1)
    for i, _u in enumerate(rates):
        new = ch // 2
        p = f"dec.upsample_conv_blocks.{i}"
        g.conv(p + ".input_conv", new, ch + ch // 4, 7, std=1.0 / math.sqrt((ch + ch // 4) * 7))
        for m, kk in enumerate(rk):
            g.normal(f"{p}.blocks.{m}.0.weight", (new,), 0.05, 0.3)  # AdaIN scale (ref init = 1)
            g.normal(f"{p}.blocks.{m}.2.weight", (new,), 0.05, 0.3)
            for j in range(nd):
                g.wn_conv(f"{p}.blocks.{m}.1.convs1.{j}", (new, new, kk), bias_len=new, std=RES_GAIN / math.sqrt(new * kk))
                g.wn_conv(f"{p}.blocks.{m}.1.convs2.{j}", (new, new, kk), bias_len=new, std=RES_GAIN / math.sqrt(new * kk))
        ch = new
    g.wn_conv("dec.conv_post", (1, ch, 7), std=0.5773503 / math.sqrt(ch * 7))


def make_synth_checkpoint(sr: int = 48000, vocoder: str = "HiFi-GAN", seed: int = 0, *,
                          upsample_initial_channel: int = 512, spk_embed_dim: int = 109,
                          half: bool = False, version: str = "v2", smooth_pitch: bool = False) -> dict:
    """An in-memory equivalent of an exported ``.pth`` (extract_model.py:56-107).

    ``half=True`` stores the weights as fp16 exactly like real exports do
    (``extract_model.py:58``); the default keeps fp32 so parity tests isolate arithmetic.
    """
    cfg = config_list(sr, upsample_initial_channel=upsample_initial_channel, spk_embed_dim=spk_embed_dim)
    rates, ksizes = cfg[12], cfg[14]
    g = _Gen(seed)
    # v1 models take the 256-dim final_proj features (infer.py:472)
    _text_encoder(g, emb=768 if version == "v2" else 256, smooth_pitch=smooth_pitch)
    if vocoder == "MRF HiFi-GAN":
        _dec_mrf(g, rates, ksizes, upsample_initial_channel)
    elif vocoder == "RefineGAN":
        _dec_refine(g, rates, upsample_initial_channel)
    else:
        _dec_nsf(g, rates, ksizes, upsample_initial_channel)
    _flow(g)
    g.normal("emb_g.weight", (spk_embed_dim, 256), 1.0)
    weight = OrderedDict((k, (v.half() if half else v)) for k, v in g.out.items())
    return {
        "weight": weight,
        "config": cfg,
        "f0": 1,
        "version": version,
        "sr": sr,
        "vocoder": vocoder,
        "epoch": 0,
        "step": 0,
        "model_name": f"synthetic-{vocoder}-{sr}-seed{seed}",
    }


def make_hubert_state_dict(seed: int = 1, *, layers: int = 12, hidden: int = 768, ffn: int = 3072,
                           conv_dim: int = 512) -> "OrderedDict[str, torch.Tensor]":
    """HuBERT-base / ContentVec weights in transformers naming."""
    g = _Gen(seed)
    kernels = (10, 3, 3, 3, 3, 2, 2)
    cin = 1
    for i, k in enumerate(kernels):
        g.normal(f"feature_extractor.conv_layers.{i}.conv.weight", (conv_dim, cin, k), math.sqrt(2.0 / (cin * k)))
        cin = conv_dim
    g.layer_norm("feature_extractor.conv_layers.0.layer_norm", conv_dim)
    g.layer_norm("feature_projection.layer_norm", conv_dim)
    g.linear("feature_projection.projection", hidden, conv_dim, std=0.02)
    p = "encoder.pos_conv_embed.conv"
    v = g.normal(p + ".parametrizations.weight.original1", (hidden, hidden // 16, 128),
                 2.0 * math.sqrt(1.0 / (128 * hidden)))
    nrm = v.norm(2, dim=(0, 1), keepdim=True)  # weight_norm(dim=2): g has shape [1,1,128]
    jit = 1.0 + 0.1 * g._rng(p + ".g").standard_normal((1, 1, 128), dtype=np.float32)
    g.out[p + ".parametrizations.weight.original0"] = (nrm * torch.from_numpy(jit)).float()
    g.normal(p + ".bias", (hidden,), 0.02)
    g.layer_norm("encoder.layer_norm", hidden)
    for i in range(layers):
        L = f"encoder.layers.{i}"
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            g.linear(f"{L}.attention.{nm}", hidden, hidden, std=0.02)
        g.layer_norm(f"{L}.layer_norm", hidden)
        g.linear(f"{L}.feed_forward.intermediate_dense", ffn, hidden, std=0.02)
        g.linear(f"{L}.feed_forward.output_dense", hidden, ffn, std=0.02)
        g.layer_norm(f"{L}.final_layer_norm", hidden)
    g.normal("masked_spec_embed", (hidden,), 1.0)
    g.linear("final_proj", 256, hidden, std=0.02)  # the fork's v1-only head (rvc/lib/utils.py:31-34)
    return g.out


# ---- "peaked" RMVPE tail ---------------------------------------------------------------------------
# A random-weight RMVPE has a flat, noise-like salience: the arg-max over its 360 bins sits within fp32 noise of
# the runner-up on about one frame per 30 s, and because RMVPE.py:487-512 averages +-4 bins around the arg-max, two
# correct fp32 evaluations of the network can then disagree on f0 by an octave on that frame.  A trained RMVPE is
# unimodal: a bump of a few bins around the pitch, ~0 elsewhere (its training targets are Gaussian-blurred one-hots),
# so an arg-max flip can only move to the neighbouring bin of the SAME bump and the local average barely moves.
# `make_rmvpe_state_dict(seed, peaked=True)` keeps the seeded U-Net and output conv and replaces the recurrent layer +
# linear head by a construction with exactly that property:
#   s      = u . x_t                         one scalar projection of the 384-d output-conv features of frame t
#   psi    = PSI_C + RHO (s - mu) / sigma    bump position in bins (mu, sigma: statistics of s on the SURVEY 8(d) input)
#   n_j    = tanh(A (psi - c_j))             the GRU's candidate state: 512 units tiling [PSI_C - HALF, PSI_C + HALF]
#                                            (forward units on even, backward units on odd positions)
#   h      = (1 - z) n + z h_prev            update gate ~0.08 (bias -2.5) with small seeded random recurrent / gate weights,
#                                            so the recurrence is live but the code of frame t dominates
#   logit  = L0 + V h                        V: least-squares readout with  V n(psi) = (L1 - L0) exp(-(i - psi)^2 / 2 SB^2)
# i.e. sigmoid(logit) is a Gaussian-shaped bump of height ~0.8 and sigma 1.25 bins (25 cents, RMVPE's label blur) at psi,
# and ~6e-6 elsewhere.  The bump POSITION is a continuous function of the U-Net's output, so every conv of the network
# still decides the result -- it follows the network's (random) features, not the input's true pitch.
PEAKED = dict(PSI_C=180.0, RHO=8.0, HALF=24.0, A=1.5, SB=1.25, L0=-12.0, L1=3.0, Z_BIAS=-2.5)
# (mean, std) of s over the frames of synth_audio(30 s, seed 0) after the pipeline's high-pass + 1 s reflect padding,
# evaluated with the REFERENCE's own E2E module (tests/golden/make_golden_peaked.py prints them)
PEAKED_STATS = {0: (82.851207, 85.552342)}


def _peaked_tail(g: "_Gen", seed: int, stats=None):
    P = PEAKED
    if stats is None:
        if seed not in PEAKED_STATS:
            raise ValueError(f"peaked RMVPE: no feature statistics recorded for seed {seed}; pass stats=(mean, std)")
        stats = PEAKED_STATS[seed]
    mu, sigma = float(stats[0]), float(stats[1])
    hid, inp = 256, 384
    u = g._rng("peaked.u").standard_normal(inp)
    u /= np.linalg.norm(u)
    dc = 2.0 * P["HALF"] / (2 * hid)
    centres = {}
    for d, sfx in enumerate(("", "_reverse")):
        c = P["PSI_C"] - P["HALF"] + (2.0 * np.arange(hid) + d + 0.5) * dc
        centres[sfx] = c
        w_ih = np.zeros((3 * hid, inp))
        w_ih[:2 * hid] = g._rng(f"peaked.w_ih{sfx}").standard_normal((2 * hid, inp)) * (0.3 / (sigma * math.sqrt(inp)))
        w_ih[2 * hid:] = (P["A"] * P["RHO"] / sigma) * u[None, :]
        b_ih = np.zeros(3 * hid)
        b_ih[hid:2 * hid] = P["Z_BIAS"]
        b_ih[2 * hid:] = P["A"] * (P["PSI_C"] - P["RHO"] * mu / sigma - c)
        w_hh = g._rng(f"peaked.w_hh{sfx}").standard_normal((3 * hid, hid)) * 0.02
        w_hh[2 * hid:] *= 0.25
        g.out[f"fc.0.gru.weight_ih_l0{sfx}"] = torch.from_numpy(w_ih.astype(np.float32))
        g.out[f"fc.0.gru.weight_hh_l0{sfx}"] = torch.from_numpy(w_hh.astype(np.float32))
        g.out[f"fc.0.gru.bias_ih_l0{sfx}"] = torch.from_numpy(b_ih.astype(np.float32))
        g.out[f"fc.0.gru.bias_hh_l0{sfx}"] = torch.zeros(3 * hid)
    # readout: ridge least squares over a grid of bump positions (beyond the tiled range the target stays at the range end)
    lo, hi = P["PSI_C"] - P["HALF"], P["PSI_C"] + P["HALF"]
    psi = np.arange(lo - 8.0, hi + 8.0, 0.04)
    code = np.concatenate([np.tanh(P["A"] * (psi[:, None] - centres[sfx][None, :])) for sfx in ("", "_reverse")], axis=1)
    X = np.concatenate([code, np.ones((len(psi), 1))], axis=1)
    bins = np.arange(360)[None, :]
    target = (P["L1"] - P["L0"]) * np.exp(-0.5 * ((bins - np.clip(psi, lo + 2.0, hi - 2.0)[:, None]) / P["SB"]) ** 2)
    lam = 1e-4 * len(psi)
    A_ = X.T @ X + lam * np.eye(X.shape[1])
    A_[-1, -1] -= lam
    W = np.linalg.solve(A_, X.T @ target)          # [513, 360]
    g.out["fc.1.weight"] = torch.from_numpy(np.ascontiguousarray(W[:-1].T).astype(np.float32))
    g.out["fc.1.bias"] = torch.from_numpy((P["L0"] + W[-1]).astype(np.float32))
    return u


def make_rmvpe_state_dict(seed: int = 0, *, peaked: bool = False, peaked_stats=None) -> "OrderedDict[str, torch.Tensor]":
    """``E2E(4, 1, (2, 2))`` weights (RMVPE.py:289-339, 420-433), BatchNorm in eval form.

    ``peaked=True``: same U-Net / output conv, but a recurrent layer + head that turn the features into a unimodal,
    trained-like salience (see PEAKED above)."""
    g = _Gen(seed)

    def bn(prefix, c):
        g.normal(prefix + ".weight", (c,), 0.1, 1.0)
        g.normal(prefix + ".bias", (c,), 0.1)
        g.normal(prefix + ".running_mean", (c,), 0.1)
        g.positive(prefix + ".running_var", (c,), 1.0, 0.2)
        g.out[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def conv2d(prefix, cout, cin, kh, kw, bias=False, gain=1.0):
        fan_in = cin * kh * kw
        g.normal(prefix + ".weight", (cout, cin, kh, kw), gain / math.sqrt(fan_in))
        if bias:
            g.normal(prefix + ".bias", (cout,), 0.5 / math.sqrt(fan_in))

    def conv_block_res(prefix, cin, cout):
        conv2d(prefix + ".conv.0", cout, cin, 3, 3)
        bn(prefix + ".conv.1", cout)
        conv2d(prefix + ".conv.3", cout, cout, 3, 3)
        bn(prefix + ".conv.4", cout)
        if cin != cout:
            conv2d(prefix + ".shortcut", cout, cin, 1, 1, bias=True)

    n_blocks = 4
    bn("unet.encoder.bn", 1)
    cin, cout = 1, 16
    for i in range(5):
        for m in range(n_blocks):
            conv_block_res(f"unet.encoder.layers.{i}.conv.{m}", cin if m == 0 else cout, cout)
        cin, cout = cout, cout * 2
    # after the loop: encoder.out_channel = 512, intermediate takes 256 -> 512
    ic, oc = 256, 512
    for i in range(4):
        for m in range(n_blocks):
            conv_block_res(f"unet.intermediate.layers.{i}.conv.{m}", ic if (m == 0 and i == 0) else oc, oc)
    ch = 512
    for i in range(5):
        out = ch // 2
        # ConvTranspose2d weight is [cin, cout, 3, 3]
        g.normal(f"unet.decoder.layers.{i}.conv1.0.weight", (ch, out, 3, 3), 1.0 / math.sqrt(ch * 9 / 4))
        bn(f"unet.decoder.layers.{i}.conv1.1", out)
        for m in range(n_blocks):
            conv_block_res(f"unet.decoder.layers.{i}.conv2.{m}", out * 2 if m == 0 else out, out)
        ch = out
    conv2d("cnn", 3, 16, 3, 3, bias=True)
    hid, inp = 256, 384
    for sfx in ("", "_reverse"):
        s = 1.0 / math.sqrt(hid) * 0.5773503 * 1.7
        g.normal(f"fc.0.gru.weight_ih_l0{sfx}", (3 * hid, inp), s)
        g.normal(f"fc.0.gru.weight_hh_l0{sfx}", (3 * hid, hid), s)
        g.normal(f"fc.0.gru.bias_ih_l0{sfx}", (3 * hid,), s)
        g.normal(f"fc.0.gru.bias_hh_l0{sfx}", (3 * hid,), s)
    g.linear("fc.1", 360, 512, std=0.08)
    if peaked:
        _peaked_tail(g, seed, peaked_stats)
    return g.out


# ---- synthetic inputs -----------------------------------------------------------

def synth_audio(n_samples: int, seed: int = 0, sr: int = 16000) -> np.ndarray:
    """16 kHz mono float64 voiced FM tone + noise: 0.3 sin(2pi int(180+40 sin(2pi 0.5 t))) + 0.01 N(0,1)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    f = 180.0 + 40.0 * np.sin(2 * np.pi * 0.5 * t)
    phase = 2 * np.pi * np.cumsum(f) / sr
    # a few quiet gaps so the >41 s segmentation has real minima to find
    env = 1.0 - 0.9 * (np.sin(2 * np.pi * t / 7.3) > 0.995)
    return 0.3 * env * np.sin(phase) + 0.01 * rng.standard_normal(n_samples)


def synth_index(n_rows: int, dim: int = 768, seed: int = 0, *, n_centres: int = 512,
                jitter: float = 0.05, centres: np.ndarray | None = None) -> np.ndarray:
    """Clustered ``N x 768`` float32 feature index (HuBERT-like rows + jitter).

    i.i.d. Gaussian rows concentrate all distances (every neighbour is a near tie);
    real indices are clustered, so rows are centre + N(0, jitter^2).
    """
    rng = np.random.default_rng([seed, 7])
    if centres is None:
        centres = rng.standard_normal((n_centres, dim), dtype=np.float32) * np.float32(0.35)
    centres = np.ascontiguousarray(centres, dtype=np.float32)
    out = np.empty((n_rows, dim), dtype=np.float32)
    step = 1 << 16
    for s in range(0, n_rows, step):
        e = min(n_rows, s + step)
        which = rng.integers(0, centres.shape[0], size=e - s)
        out[s:e] = centres[which] + rng.standard_normal((e - s, dim), dtype=np.float32) * np.float32(jitter)
    return out
