"""Inference configuration constants (rvc/configs/config.py:20-41, 108-121).

Unlike the reference's singleton this class has no side effects (the reference rewrites its JSON files when
it falls back to CPU, config.py:58-67,113) and never falls back to CPU: the device is a HIP device.
"""
from __future__ import annotations

import torch


class Config:
    def __init__(self, device: str | None = None):
        if device is None:
            device = "cuda:0"
        self.device = device
        self.is_half = False          # "Inference is only in FP32" (reference README.md:22)
        self.gpu_name = torch.cuda.get_device_name(int(device.split(":")[-1])) if torch.cuda.is_available() else None
        self.x_pad, self.x_query, self.x_center, self.x_max = self.device_config()

    def device_config(self):
        # config.py:116-118, fp32 branch; the reference's <= 4 GB low-memory variant (1, 5, 30, 32) is moot
        # on a 288 GB part
        return 1, 6, 38, 41
