"""TextEncoder on PyTorch-ROCm (rvc/lib/algorithm/encoders.py:88-144, attentions.py:6-243,
normalization.py:19-26), functional over the folded state dict.

Differences from the reference that do not change the arithmetic being computed:
* the window-10 relative-position terms are applied in banded form (21 diagonals) instead of through
  zero-padded [T, 2T-1] tables and pad/reshape tricks (attentions.py:143-180), which at T = 3198 saves
  two 164 MB intermediates per layer;
* the attention mask is skipped when every frame is valid (single utterances: lengths == T).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

WINDOW = 10
FRAMES_PATH = True     # the unmasked GPU encoder keeps its activations time-major (tests compare it with the channel-major graph)


def channel_layer_norm(x, gamma, beta, eps=1e-5):
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


def prepare_attention_weights(w: Dict[str, torch.Tensor], n_layers: int) -> None:
    """Fused q/k/v projection and friends, built once when the state dict reaches the device (Synthesizer._to_device):
    forwards on several host threads share `w` and must find it complete."""
    for i in range(n_layers):
        p = f"enc_p.encoder.attn_layers.{i}"
        if p + ".conv_q.weight" not in w:
            continue
        w[p + ".qkv.weight"] = torch.cat([w[p + f".conv_{n}.weight"][:, :, 0] for n in "qkv"], 0).contiguous()
        w[p + ".qkv.bias"] = torch.cat([w[p + f".conv_{n}.bias"] for n in "qkv"], 0).contiguous()
        w[p + ".o.weight"] = w[p + ".conv_o.weight"][:, :, 0].contiguous()
        w[p + ".ek"] = w[p + ".emb_rel_k"][0].contiguous()
        w[p + ".ev"] = w[p + ".emb_rel_v"][0].contiguous()
        # the FFN's "same"-padded convs as GEMMs over time-major frames: the window [x[t-1], x[t], x[t+1]] of a frame is k * C
        # contiguous values of the zero-padded [T + k - 1, C] tensor, W2d[o][j * C + c] = conv.weight[o][c][j]
        f = f"enc_p.encoder.ffn_layers.{i}"
        for n in ("conv_1", "conv_2"):
            cw = w[f + f".{n}.weight"]
            w[f + f".{n}.w2d"] = cw.permute(0, 2, 1).reshape(cw.shape[0], -1).contiguous()
    if "enc_p.proj.weight" in w:
        w["enc_p.proj.w2d"] = w["enc_p.proj.weight"][:, :, 0].contiguous()


def _rel_attention_hip(x, w: Dict[str, torch.Tensor], p: str, n_heads: int):
    """Unmasked layer on the GPU: one fused q/k/v GEMM, librvc_amd K7 (scores, relative terms, softmax, PV in one
    launch + a split combine), output projection.  Time-major in between, so no [T, T] tensor is ever materialised."""
    from rvc_amd import _native
    d = x.shape[1]
    qkv = F.linear(x.transpose(1, 2), w[p + ".qkv.weight"], w[p + ".qkv.bias"]).contiguous()
    a = _native.attention_qkv(qkv, n_heads, 1.0 / math.sqrt(d // n_heads), w[p + ".ek"], w[p + ".ev"])
    return F.linear(a, w[p + ".o.weight"], w[p + ".conv_o.bias"]).transpose(1, 2)


def rel_attention(x, w: Dict[str, torch.Tensor], p: str, n_heads: int, attn_mask: Optional[torch.Tensor]):
    if x.is_cuda and attn_mask is None and w[p + ".emb_rel_k"].shape[0] == 1 and (x.shape[1] // n_heads) in (64, 96):
        return _rel_attention_hip(x, w, p, n_heads)
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
    ek, ev = w[p + ".emb_rel_k"][0], w[p + ".emb_rel_v"][0]
    rel_logits = torch.matmul(qs, ek.t())  # [b,h,t,21]
    for r in range(2 * WINDOW + 1):
        off = r - WINDOW
        i0, i1 = max(0, -off), min(t, t - off)
        if i0 < i1:
            torch.diagonal(scores, off, -2, -1).add_(rel_logits[:, :, i0:i1, r])
    if attn_mask is not None:
        scores = scores.masked_fill(attn_mask == 0, -1e4)
    p_attn = F.softmax(scores, dim=-1)
    out = torch.matmul(p_attn, v)
    band = torch.zeros(b, n_heads, t, 2 * WINDOW + 1, dtype=x.dtype, device=x.device)
    for r in range(2 * WINDOW + 1):
        off = r - WINDOW
        i0, i1 = max(0, -off), min(t, t - off)
        if i0 < i1:
            band[:, :, i0:i1, r] = torch.diagonal(p_attn, off, -2, -1)
    out = out + torch.matmul(band, ev)
    out = out.transpose(2, 3).contiguous().view(b, d, t)
    return F.conv1d(out, w[p + ".conv_o.weight"], w[p + ".conv_o.bias"])


def _same_conv_frames(x, w2d, bias, k: int):
    """nn.Conv1d(C, O, k, padding = (k - 1) // 2) over TIME-MAJOR frames x [B, T, C] as one GEMM: frame t's window is the k * C
    contiguous values starting at row t of the zero-padded tensor (an overlapping-row view; the GEMM reads a compact copy of it)."""
    b, t, c = x.shape
    pad = (k - 1) // 2
    xp = F.pad(x, (0, 0, pad, k - 1 - pad))
    win = xp.as_strided((b, t, k * c), (xp.stride(0), c, 1))
    return F.linear(win, w2d, bias)


def _text_encoder_frames(w: Dict[str, torch.Tensor], x, n_layers: int, n_heads: int, kernel_size: int, out_channels: int):
    """The unmasked encoder (every frame valid) with the activations kept time-major, [B, T, C]: the attention kernel and the
    projections are time-major already, LayerNorm over channels is then the plain last-dim one, and the FFN's k-tap convs are GEMMs
    over overlapping rows -- no transposes, no im2col (rvc/lib/algorithm/attentions.py: Encoder.forward, FFN.forward).  Per layer
    8 library launches + the attention instead of ~20."""
    from rvc_amd import _native
    d = x.shape[-1]
    scale = 1.0 / math.sqrt(d // n_heads)
    for i in range(n_layers):
        p = f"enc_p.encoder.attn_layers.{i}"
        qkv = F.linear(x, w[p + ".qkv.weight"], w[p + ".qkv.bias"])
        a = _native.attention_qkv(qkv, n_heads, scale, w[p + ".ek"], w[p + ".ev"])
        y = F.linear(a, w[p + ".o.weight"], w[p + ".conv_o.bias"])
        x = F.layer_norm(x + y, (d,), w[f"enc_p.encoder.norm_layers_1.{i}.gamma"], w[f"enc_p.encoder.norm_layers_1.{i}.beta"], 1e-5)
        f = f"enc_p.encoder.ffn_layers.{i}"
        y = torch.relu_(_same_conv_frames(x, w[f + ".conv_1.w2d"], w[f + ".conv_1.bias"], kernel_size))
        y = _same_conv_frames(y, w[f + ".conv_2.w2d"], w[f + ".conv_2.bias"], kernel_size)
        x = F.layer_norm(x + y, (d,), w[f"enc_p.encoder.norm_layers_2.{i}.gamma"], w[f"enc_p.encoder.norm_layers_2.{i}.beta"], 1e-5)
    stats = F.linear(x, w["enc_p.proj.w2d"], w["enc_p.proj.bias"]).transpose(1, 2)    # [B, 2 * out, T]: the flow is channel-major
    return torch.split(stats, out_channels, dim=1)


def text_encoder(w: Dict[str, torch.Tensor], phone, pitch, lengths, *, hidden=192, out_channels=192, n_heads=2,
                 n_layers=6, kernel_size=3, lengths_host=None):
    """``lengths_host``: the same lengths as Python ints when the caller has them (saves a device->host read)."""
    x = F.linear(phone, w["enc_p.emb_phone.weight"], w["enc_p.emb_phone.bias"])
    if pitch is not None:
        x = x + F.embedding(pitch, w["enc_p.emb_pitch.weight"])
    x = F.leaky_relu(x * math.sqrt(hidden), 0.1)
    t = x.size(1)
    x_mask = (torch.arange(t, device=x.device)[None, :] < lengths[:, None]).unsqueeze(1).to(x.dtype)
    full = all(int(n) == t for n in lengths_host) if lengths_host is not None else bool((lengths == t).all())
    if (full and x.is_cuda and FRAMES_PATH and "enc_p.proj.w2d" in w and (hidden // n_heads) in (64, 96)
            and w["enc_p.encoder.attn_layers.0.emb_rel_k"].shape[0] == 1):
        m, logs = _text_encoder_frames(w, x, n_layers, n_heads, kernel_size, out_channels)
        return m, logs, x_mask
    x = x.transpose(1, -1)
    attn_mask = None if full else x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    pad = (kernel_size - 1) // 2
    if full:   # every frame valid: the mask is all ones and multiplying by it is the identity (saves ~30 passes)
        for i in range(n_layers):
            y = rel_attention(x, w, f"enc_p.encoder.attn_layers.{i}", n_heads, None)
            x = channel_layer_norm(x + y, w[f"enc_p.encoder.norm_layers_1.{i}.gamma"], w[f"enc_p.encoder.norm_layers_1.{i}.beta"])
            f = f"enc_p.encoder.ffn_layers.{i}"
            y = torch.relu_(F.conv1d(x, w[f + ".conv_1.weight"], w[f + ".conv_1.bias"], padding=pad))
            y = F.conv1d(y, w[f + ".conv_2.weight"], w[f + ".conv_2.bias"], padding=pad)
            x = channel_layer_norm(x + y, w[f"enc_p.encoder.norm_layers_2.{i}.gamma"], w[f"enc_p.encoder.norm_layers_2.{i}.beta"])
        stats = F.conv1d(x, w["enc_p.proj.weight"], w["enc_p.proj.bias"])
        m, logs = torch.split(stats, out_channels, dim=1)
        return m, logs, x_mask
    x = x * x_mask
    for i in range(n_layers):
        y = rel_attention(x, w, f"enc_p.encoder.attn_layers.{i}", n_heads, attn_mask)
        x = channel_layer_norm(x + y, w[f"enc_p.encoder.norm_layers_1.{i}.gamma"], w[f"enc_p.encoder.norm_layers_1.{i}.beta"])
        f = f"enc_p.encoder.ffn_layers.{i}"
        y = torch.relu(F.conv1d(x * x_mask, w[f + ".conv_1.weight"], w[f + ".conv_1.bias"], padding=pad))
        y = F.conv1d(y * x_mask, w[f + ".conv_2.weight"], w[f + ".conv_2.bias"], padding=pad) * x_mask
        x = channel_layer_norm(x + y, w[f"enc_p.encoder.norm_layers_2.{i}.gamma"], w[f"enc_p.encoder.norm_layers_2.{i}.beta"])
    x = x * x_mask
    stats = F.conv1d(x, w["enc_p.proj.weight"], w["enc_p.proj.bias"]) * x_mask
    m, logs = torch.split(stats, out_channels, dim=1)
    return m, logs, x_mask
