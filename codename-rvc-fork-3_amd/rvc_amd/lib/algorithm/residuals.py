"""The normalising flow of the Synthesizer, reverse direction only (rvc/lib/algorithm/residuals.py:157-170,
239-264; WaveNet rvc/lib/algorithm/modules.py:78-109; gate commons.py:142-157).  34 GFLOP per 30 s clip.

Single full-length utterances on the GPU take a leaner formulation (no masks, conditioning folded into biases, the gate
as one librvc_amd kernel); batches with padding keep the reference's formulation below."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def wavenet(x, x_mask, g_cond, w: Dict[str, torch.Tensor], p: str, hidden=192, n_layers=3, k=5):
    output = torch.zeros_like(x)
    for i in range(n_layers):
        x_in = F.conv1d(x, w[f"{p}.in_layers.{i}.weight"], w[f"{p}.in_layers.{i}.bias"], padding=(k - 1) // 2)
        in_act = x_in + g_cond[:, i * 2 * hidden:(i + 1) * 2 * hidden, :]
        acts = torch.tanh(in_act[:, :hidden]) * torch.sigmoid(in_act[:, hidden:])
        rs = F.conv1d(acts, w[f"{p}.res_skip_layers.{i}.weight"], w[f"{p}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * x_mask
            output = output + rs[:, hidden:]
        else:
            output = output + rs
    return output * x_mask


def prepare_flow_weights(w: Dict[str, torch.Tensor], n_layers: int = 3) -> None:
    """The WaveNet in_layer biases of each coupling layer as one vector (the conditioning is added to it per call); built
    once at load, like encoders.prepare_attention_weights."""
    for key in [k for k in w if k.endswith(".enc.cond_layer.weight")]:
        q = key[:-len(".cond_layer.weight")]
        w[q + ".in_bias"] = torch.cat([w[f"{q}.in_layers.{i}.bias"] for i in range(n_layers)])


def _flow_reverse_full(w: Dict[str, torch.Tensor], z_p, g, half, hidden, n_flows, n_layers=3, k=5):
    """B == 1, every frame valid: no mask multiplies, the conditioning folded into the in_layer biases, the gate as
    one librvc_amd kernel (tanh * sigmoid of the two halves) instead of add + tanh + sigmoid + mul.  The convs stay
    with hipBLASLt / MIOpen: at 3198 columns they are too small for the vocoder's conv kernel (measured: 37 us per
    1x1 conv on 39 tiles against 7-25 us), ~130 launches instead of ~300."""
    from rvc_amd import _native
    x = z_p
    for n in range(2 * (n_flows - 1), -1, -2):
        x = torch.flip(x, [1])
        p = f"flow.flows.{n}"
        q = p + ".enc"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, w[p + ".pre.weight"], w[p + ".pre.bias"])
        bias_all = F.conv1d(g, w[q + ".cond_layer.weight"], w[q + ".cond_layer.bias"]).view(-1) + w[q + ".in_bias"]
        skip = None
        for i in range(n_layers):
            x_in = F.conv1d(h, w[f"{q}.in_layers.{i}.weight"], bias_all[i * 2 * hidden:(i + 1) * 2 * hidden],
                            padding=(k - 1) // 2)
            acts = _native.gate_tanh_sigmoid(x_in.contiguous())
            rs = F.conv1d(acts, w[f"{q}.res_skip_layers.{i}.weight"], w[f"{q}.res_skip_layers.{i}.bias"])
            if i < n_layers - 1:
                h = h + rs[:, :hidden]
                skip = rs[:, hidden:] if skip is None else skip + rs[:, hidden:]
            else:
                skip = rs if skip is None else skip + rs
        m = F.conv1d(skip, w[p + ".post.weight"], w[p + ".post.bias"])
        x = torch.cat([x0, x1 - m], 1)
    return x


def flow_reverse(w: Dict[str, torch.Tensor], z_p, x_mask, g, *, half=96, hidden=192, n_flows=4, full=False):
    """``full``: the caller knows every frame is valid (x_mask all ones), which the in-HBM path needs."""
    if full and z_p.is_cuda and z_p.shape[0] == 1:
        return _flow_reverse_full(w, z_p, g, half, hidden, n_flows)
    x = z_p
    for n in range(2 * (n_flows - 1), -1, -2):
        x = torch.flip(x, [1])  # Flip (residuals.py:100-106)
        p = f"flow.flows.{n}"
        x0, x1 = x[:, :half], x[:, half:]
        h = F.conv1d(x0, w[p + ".pre.weight"], w[p + ".pre.bias"]) * x_mask
        g_cond = F.conv1d(g, w[p + ".enc.cond_layer.weight"], w[p + ".enc.cond_layer.bias"])
        h = wavenet(h, x_mask, g_cond, w, p + ".enc", hidden)
        m = F.conv1d(h, w[p + ".post.weight"], w[p + ".post.bias"]) * x_mask
        x = torch.cat([x0, (x1 - m) * x_mask], 1)  # mean-only coupling: logs == 0
    return x
