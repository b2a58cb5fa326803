"""Multi-GPU batch conversion: one process per GPU, utterances sharded by striding, the feature index
replicated with ONE RCCL broadcast over xGMI at load time.

The reference converts batches with a sequential loop on one device (rvc/infer/infer.py:396-406); its only
multi-GPU precedent is feature extraction, which starts one worker per device itself and strides the file list,
``files[i::len(devices)]`` (rvc/train/extract/extract.py:141-152, 198).  Both are mirrored here: ``spawn_ranks``
starts the per-GPU processes, ``shard_indices`` is the striding.  Utterances share no state, so the steady state
has no collective at all; a final reduce of (samples, seconds) produces the report.

Transport: on GPUs the index travels through librvc_amd's own RCCL communicator (``rvc_index_broadcast``,
include/rvc_amd.h) -- torch.distributed only carries the 128-byte communicator id and the few report scalars.
Tensors that live on the host (the world_size-2 gloo tests on CPU) take torch.distributed's gloo broadcast instead;
that branch is chosen by where the caller's tensor lives, never as a fallback for a failed RCCL call.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
import socket
import subprocess
import sys
import time
from typing import Callable, List, Sequence

import numpy as np
import torch
import torch.distributed as dist

_MASK64 = (1 << 64) - 1


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


RENDEZVOUS_EXIT = 75   # EX_TEMPFAIL: rank 0 could not bind the rendezvous port (spawn_ranks starts the job again on a new one)


def init_process_group(backend: str | None = None):
    """Idempotent init from the torchrun-style environment; returns (rank, world, local_rank).

    On GPUs the group is "cpu:gloo,cuda:nccl" ("nccl" is RCCL on ROCm): device tensors (barrier, report reduce) go over
    RCCL, the 128-byte communicator id over gloo.  RVC_DIST_BACKEND=gloo lets several ranks share one GPU or none
    (control-flow tests)."""
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("RVC_DIST_BACKEND") or ("cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo")
        if "nccl" in backend:
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        except Exception as e:   # noqa: BLE001  (DistNetworkError / RuntimeError, depending on the torch build)
            msg = str(e).lower()
            if rank == 0 and ("address already in use" in msg or "eaddrinuse" in msg):
                print(f"[rvc_amd.distributed] rendezvous port {os.environ.get('MASTER_PORT')} is taken: {e}", file=sys.stderr)
                sys.exit(RENDEZVOUS_EXIT)
            raise
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """utterance i -> rank i mod world (extract.py:145: ``files[i::len(devices)]``)"""
    return list(range(rank, n_items, world))


def _has_gloo() -> bool:
    """Can the default group move host tensors?"""
    try:
        cfg = str(dist.get_backend_config())
    except Exception:
        cfg = str(dist.get_backend())
    return "gloo" in cfg.lower()


# ---- the one collective: index replication -----------------------------------------------------------------------
_comm = None          # librvc_amd RCCL communicator of this process (one per job)
_last_broadcast = {}  # facts of the last broadcast_index call, for the bench report


@contextlib.contextmanager
def _c_stdout_to_stderr():
    """RCCL prints a version banner on C stdout when its first communicator comes up (RCCL 2.26: always).  A report line
    on stdout -- bench.py's one JSON line -- must not share the stream with it, and being C-buffered the banner would even
    land AFTER a later Python print.  While the communicator is created, file descriptor 1 points at stderr."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def native_comm():
    """The job-wide RCCL communicator behind the C ABI, created on first use: rank 0 draws the id, torch.distributed
    hands its 128 bytes round, every rank joins on its own GPU."""
    global _comm
    if _comm is not None:
        return _comm
    from rvc_amd import _native
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ident = torch.zeros(_native.COMM_ID_BYTES, dtype=torch.uint8)
    if rank == 0:
        ident = torch.frombuffer(bytearray(_native.comm_unique_id()), dtype=torch.uint8).clone()
    if world > 1:
        if _has_gloo():
            dist.broadcast(ident, 0)
        else:   # an nccl-only group set up by somebody else: the id rides a device tensor
            ident_dev = ident.cuda()
            dist.broadcast(ident_dev, 0)
            ident = ident_dev.cpu()
    with _c_stdout_to_stderr():
        _comm = _native.Comm(bytes(ident.numpy().tobytes()), world, rank)
    return _comm


def destroy_native_comm():
    global _comm
    if _comm is not None:
        _comm.destroy()
        _comm = None


def broadcast_index(big_npy, device, src: int = 0, force_rccl: bool = False) -> torch.Tensor:
    """Replicate the N x dim fp32 feature index from rank ``src``: its shape, then ONE broadcast of its bytes.

    Returns the resident tensor on every rank; ranks other than ``src`` pass ``big_npy=None``.  ``force_rccl`` runs the
    RCCL broadcast even in a single-rank job (bench.py does, so that a 1-GPU run still proves the library path)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    on_gpu = torch.device(device).type == "cuda"
    _last_broadcast.clear()
    if world == 1 and not (force_rccl and on_gpu):
        return torch.as_tensor(big_npy, dtype=torch.float32).to(device).contiguous()
    shape = torch.zeros(2, dtype=torch.int64)
    if rank == src:
        t = torch.as_tensor(big_npy, dtype=torch.float32).to(device).contiguous()
        shape[0], shape[1] = t.shape
    if world > 1:
        if _has_gloo():
            dist.broadcast(shape, src)
        else:
            shape_dev = shape.to(device)
            dist.broadcast(shape_dev, src)
            shape = shape_dev.cpu()
    if rank != src:
        t = torch.empty((int(shape[0]), int(shape[1])), dtype=torch.float32, device=device)
    gloo_only = os.environ.get("RVC_DIST_BACKEND") == "gloo" and world > 1
    if on_gpu and gloo_only:
        # several ranks sharing one GPU (or a job told to stay off RCCL): RCCL refuses two ranks on one device, so the bytes
        # ride the gloo group through host memory.  Test / control-flow transport, never the measured one.
        t0 = time.perf_counter()
        host = t.cpu() if rank == src else torch.empty(t.shape, dtype=torch.float32)
        dist.broadcast(host, src)
        if rank != src:
            t.copy_(host)
        torch.cuda.synchronize()
        _last_broadcast.update(n_ranks=world, rank=rank, seconds=time.perf_counter() - t0, bytes=t.numel() * 4,
                               transport="gloo (staged through host memory)")
    elif on_gpu:
        comm = native_comm()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        comm.broadcast_(t, src)
        torch.cuda.synchronize()
        seconds = time.perf_counter() - t0
        _last_broadcast.update(comm.info(), seconds=seconds, bytes=t.numel() * 4, transport="rccl (rvc_index_broadcast)")
    else:
        t0 = time.perf_counter()
        dist.broadcast(t, src)
        _last_broadcast.update(n_ranks=world, rank=rank, seconds=time.perf_counter() - t0, bytes=t.numel() * 4,
                               transport="gloo (host tensors)")
    return t


def last_broadcast_info() -> dict:
    return dict(_last_broadcast)


def tensor_checksum(t: torch.Tensor):
    """(sum of the 32-bit words, sum of (i+1)*word_i) mod 2^64 of the tensor's bytes.  HBM tensors are reduced on the
    device (rvc_checksum64: the index never crosses PCIe again); host tensors (CPU tests) with the same formula in NumPy."""
    t = t.detach().contiguous()
    if t.is_cuda:
        from rvc_amd import _native
        s = _native.checksum64(t).cpu().numpy().view(np.uint64)
        return int(s[0]), int(s[1])
    raw = t.numpy().tobytes()
    raw += b"\0" * (-len(raw) % 4)
    w = np.frombuffer(raw, dtype="<u4").astype(np.uint64)
    with np.errstate(over="ignore"):
        s1 = int(w.sum(dtype=np.uint64))
        s2 = int((w * np.arange(1, w.size + 1, dtype=np.uint64)).sum(dtype=np.uint64))
    return s1 & _MASK64, s2 & _MASK64


def checksums_agree(t: torch.Tensor) -> bool:
    """True when every rank holds the same bytes (MIN == MAX of both checksum words over the ranks)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    s1, s2 = tensor_checksum(t)
    # int64 views of the two words, split in 32-bit halves so MIN/MAX compare exactly whatever the sign
    c = torch.tensor([s1 >> 32, s1 & 0xFFFFFFFF, s2 >> 32, s2 & 0xFFFFFFFF], dtype=torch.int64, device=t.device)
    lo, hi = c.clone(), c.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


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


def gather_seconds(seconds: float, device):
    """Every rank's own wall time of the timed region, in rank order (the bench line's per-rank min / max)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(seconds)]
    mine = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(o.item()) for o in out]


# ---- starting the ranks (extract.py:141-152 starts its per-device workers itself; so does this) ------------------
def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def count_gpus_sysfs() -> int | None:
    """GPUs of this node as the kernel driver lists them (KFD topology: nodes with SIMDs are GPUs, the others CPUs),
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when those hold plain index lists.  Reads sysfs only -- the
    launcher process never opens the GPU runtime.  None when the topology cannot be read (containers without /sys/class/kfd):
    the ranks then validate the device count themselves and exit 2."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split(None, 1) for line in f if " " in line)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [t for t in v.split(",") if t.strip() != ""]
            if all(t.strip().isdigit() for t in ids):
                n = min(n, len(ids))
    return n


def spawn_ranks(argv: Sequence[str], n_ranks: int, env_extra: dict | None = None, timeout: float | None = None,
                port_retries: int = 3) -> int:
    """Start ``n_ranks`` copies of ``argv`` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, one per
    GPU, and wait for them; returns the first non-zero exit code (0 if all succeeded).  Rank 0 inherits stdout, so its
    report line is the parent's.  When a rank fails -- or this process is interrupted (KeyboardInterrupt, SIGTERM) -- the
    other ranks are terminated by PID instead of being left at a barrier.

    The rendezvous port is found by binding port 0 and closing the socket again, so another process can take it before
    rank 0 listens; a job whose ranks all fail within the first seconds with the rendezvous exit code
    (RENDEZVOUS_EXIT = 75, raised by init_process_group when the store cannot be created) is started again on a new port,
    ``port_retries`` times.

    HSA_ENABLE_IPC_MODE_LEGACY=0 is set for the children unless the caller's environment already has a value: RCCL's
    intra-node transport shares device buffers across the rank processes through IPC handles, and on hosts whose driver
    only offers dmabuf IPC the legacy mode fails inside ncclCommInitRank with `hipIpcGetMemHandle: invalid argument`.

    The caller must not have touched the GPU: children are plain new processes (never an exec of a process that has
    initialised HIP)."""
    import signal
    rc = 0
    for attempt in range(port_retries + 1):
        rc = _spawn_once(argv, n_ranks, env_extra, timeout, signal)
        if rc != RENDEZVOUS_EXIT or attempt == port_retries:
            break
        print(f"[spawn_ranks] rendezvous failed (port taken?); attempt {attempt + 2} on a new port", file=sys.stderr)
    return rc



def _spawn_once(argv, n_ranks, env_extra, timeout, signal) -> int:
    port = free_port()
    procs = []

    def stop(which):
        for r in which:
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in which:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()

    def on_term(signum, frame):
        raise KeyboardInterrupt

    old_term = None
    try:
        old_term = signal.signal(signal.SIGTERM, on_term)
    except ValueError:          # not the main thread: SIGTERM keeps its handler, KeyboardInterrupt is still covered
        pass
    rc = 0
    try:
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            env.update(env_extra or {})
            procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
        deadline = None if timeout is None else time.time() + timeout
        alive = set(range(n_ranks))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is not None:
                    alive.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"[spawn_ranks] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr)
            if rc != 0 or (deadline is not None and time.time() > deadline):
                if rc == 0:
                    rc = 124
                    print(f"[spawn_ranks] timeout after {timeout} s; stopping all ranks", file=sys.stderr)
                stop(alive)
                break
            time.sleep(0.05)
    finally:
        stop(range(len(procs)))   # interrupted parent (or any exception above): no rank is left waiting at a barrier
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    return rc
