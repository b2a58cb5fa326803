"""The normalising flow of the Synthesizer, reverse direction only (rvc/lib/algorithm/residuals.py:157-170,
239-264; WaveNet rvc/lib/algorithm/modules.py:78-109; gate commons.py:142-157).  34 GFLOP per 30 s clip:
stays on PyTorch-ROCm."""
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


def flow_reverse(w: Dict[str, torch.Tensor], z_p, x_mask, g, *, half=96, hidden=192, n_flows=4):
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
