"""Multi-GPU batch conversion: one process per GPU, utterances sharded by striding, the feature index
replicated with ONE broadcast at load time (RCCL over xGMI on a GPU node, gloo in CPU tests).

The reference converts batches with a sequential loop on one device (rvc/infer/infer.py:396-406); its only
multi-GPU precedent is the file striding of feature extraction, ``files[i::len(devices)]``
(rvc/train/extract/extract.py:145,198), which is the partitioning used here.  Utterances share no state, so
the steady state has no collective at all; a final reduce of (samples, seconds) produces the report.
"""
from __future__ import annotations

import os
import zlib
from typing import Callable, List, Sequence

import numpy as np
import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str | None = None):
    """Idempotent init from the torchrun environment; returns (rank, world, local_rank)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm; RVC_DIST_BACKEND=gloo lets several ranks share one GPU (control-flow tests)
            backend = os.environ.get("RVC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """utterance i -> rank i mod world"""
    return list(range(rank, n_items, world))


def broadcast_index(big_npy, device, src: int = 0) -> torch.Tensor:
    """Replicate the N x 768 fp32 feature index from rank ``src``: one broadcast of its shape, one of its bytes.

    Returns the device tensor on every rank.  Ranks other than ``src`` pass ``big_npy=None``."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return torch.as_tensor(big_npy, dtype=torch.float32).to(device).contiguous()
    shape = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        t = torch.as_tensor(big_npy, dtype=torch.float32).to(device).contiguous()
        shape[0], shape[1] = t.shape
    dist.broadcast(shape, src)
    if rank != src:
        t = torch.empty((int(shape[0]), int(shape[1])), dtype=torch.float32, device=device)
    dist.broadcast(t, src)
    return t


def tensor_checksum(t: torch.Tensor) -> int:
    """crc32 of the raw bytes (verification that every rank holds the root's index)."""
    return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes())


def checksums_agree(t: torch.Tensor) -> bool:
    c = torch.tensor([tensor_checksum(t)], dtype=torch.int64, device=t.device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    lo, hi = c.clone(), c.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(lo.item() == hi.item())


def convert_sharded(utterances: Sequence, convert: Callable, rank: int, world: int):
    """Run ``convert(i, utterance)`` for this rank's share; returns {global index: result}."""
    return {i: convert(i, utterances[i]) for i in shard_indices(len(utterances), rank, world)}


def reduce_report(samples: int, seconds: float, device):
    """(total samples over ranks, max seconds over ranks)"""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return samples, seconds
    s = torch.tensor([float(samples)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s.item()), float(t.item())
