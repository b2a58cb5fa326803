"""Checkpoint plumbing for the exported model format.

The reference keeps weight-norm parametrised at inference and recomputes ``w = g * v / ||v||`` on
every forward (``remove_weight_norm`` is never called on the inference path).  Here the
fold happens once at load, in fp32, through ``torch._weight_norm`` itself.
"""
from __future__ import annotations

from typing import Dict

import torch


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """fp16/fp32 state dict with legacy ``.weight_g/.weight_v`` (rvc/train/process/extract_model.py:99-105)
    or ``.parametrizations.weight.original0/1`` keys -> fp32 dict with plain ``.weight`` tensors.

    The norm is taken over every dim where ``g`` has extent 1 (dim=0 for the synthesizer's convs,
    dim=2 for HuBERT's positional conv)."""
    out: Dict[str, torch.Tensor] = {}
    consumed = set()
    for key, v in sd.items():
        for v_sfx, g_sfx in ((".weight_v", ".weight_g"),
                             (".parametrizations.weight.original1", ".parametrizations.weight.original0")):
            if key.endswith(v_sfx):
                base = key[: -len(v_sfx)]
                g = sd[base + g_sfx].float()
                vf = v.float()
                keep = [d for d in range(vf.dim()) if g.shape[d] != 1]
                # the very op the reference's parametrization executes per forward (torch._weight_norm), run once
                out[base + ".weight"] = torch._weight_norm(vf, g, keep[0] if keep else 0)
                consumed.update((key, base + g_sfx))
    for key, v in sd.items():
        if key not in consumed:
            out[key] = v.float() if v.is_floating_point() else v
    return out
