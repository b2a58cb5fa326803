"""ctypes binding of librvc_amd.so (include/rvc_amd.h) for torch tensors.

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is
missing this module raises at import, and every wrapper raises ``NativeError`` with the
library's own message when a call fails.  Tensors are passed as raw device pointers
(``tensor.data_ptr()``) and every launch goes on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes
import threading
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "librvc_amd.so")
# The ablation build (same sources with -DRVC_ABLATE: the RVC_* tuning switches and the kernels that leave work out) is only
# ever loaded when RVC_AMD_LIB names it -- tools/ablate_*.sh and the F(4,3) test's child process do.
if os.environ.get("RVC_AMD_LIB"):
    LIB_PATH = os.path.abspath(os.environ["RVC_AMD_LIB"])
    print(f"[rvc_amd] RVC_AMD_LIB: loading {LIB_PATH} instead of the product library", flush=True)


class NativeError(RuntimeError):
    pass


if not os.path.isfile(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(hipcc --offload-arch=gfx950).  rvc_amd has no fallback path without its HIP library.")

_lib = ctypes.CDLL(LIB_PATH)


class DecoderConfig(ctypes.Structure):
    _fields_ = [
        ("kind", c_int), ("sample_rate", c_int), ("in_channels", c_int), ("upsample_initial_channel", c_int),
        ("gin_channels", c_int), ("n_ups", c_int), ("upsample_rates", c_int * 8), ("upsample_kernel_sizes", c_int * 8),
        ("n_res_kernels", c_int), ("res_kernel_sizes", c_int * 4), ("res_dilations", c_int * 4),
        ("n_res_dilations", c_int), ("weight_storage", c_int),
    ]


class DecoderNoise(ctypes.Structure):
    _fields_ = [("src_rand_dev", c_void_p), ("src_randn_dev", c_void_p), ("adain_randn_dev", c_void_p)]


# every symbol include/rvc_amd.h declares: (restype, argtypes)
SYMBOLS = {
    "rvc_abi_version": (c_int, []),
    "rvc_last_error": (c_char_p, []),
    "rvc_knn_index_aux_bytes": (c_int, [c_int64, c_int, POINTER(c_size_t)]),
    "rvc_knn_index_build": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "rvc_knn_set_mode": (c_int, [c_int]),
    "rvc_knn_rank_candidates": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p,
                                        c_void_p, c_void_p]),
    "rvc_knn_workspace_bytes": (c_int, [c_int64, c_int64, c_int, c_int, POINTER(c_size_t)]),
    "rvc_knn_search": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                               c_void_p, c_size_t, c_void_p]),
    "rvc_knn_blend": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p,
                              c_void_p]),
    "rvc_logmel_workspace_bytes": (c_int, [c_int, c_int64, POINTER(c_size_t)]),
    "rvc_logmel_rmvpe": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]),
    "rvc_mel_create": (c_int, [c_int, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int, POINTER(c_void_p)]),
    "rvc_mel_destroy": (c_int, [c_void_p]),
    "rvc_mel_frames": (c_int, [c_void_p, c_int64, POINTER(c_int64)]),
    "rvc_mel_workspace_bytes": (c_int, [c_void_p, c_int, c_int64, POINTER(c_size_t)]),
    "rvc_mel_forward": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rvc_resample_poly_f64": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "rvc_filtfilt_workspace_bytes": (c_int, [c_int64, POINTER(c_size_t)]),
    "rvc_filtfilt_order5": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "rvc_bigru_workspace_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "rvc_bigru_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_size_t,
                                  c_void_p]),
    "rvc_bigru_status": (c_int, [c_void_p, c_int, POINTER(c_int), c_void_p]),
    "rvc_bigru_set_spin_limit": (c_int, [ctypes.c_uint]),
    "rvc_attention_workspace_bytes": (c_int, [c_int, c_int64, c_int, c_int, POINTER(c_size_t)]),
    "rvc_attention_qkv_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_float,
                                      c_void_p, c_size_t, c_void_p]),
    "rvc_bias_relu_add_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]),
    "rvc_gate_tanh_sigmoid_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p]),
    "rvc_rownorm_gelu_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, ctypes.c_float, c_void_p]),
    "rvc_set_concurrency_hint": (c_int, [c_int]),
    "rvc_decoder_create": (c_int, [POINTER(DecoderConfig), POINTER(c_void_p)]),
    "rvc_decoder_set_tensor": (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    "rvc_decoder_finalize": (c_int, [c_void_p]),
    "rvc_decoder_destroy": (c_int, [c_void_p]),
    "rvc_decoder_upp": (c_int, [c_void_p]),
    "rvc_decoder_workspace_bytes": (c_int, [c_void_p, c_int, c_int64, POINTER(c_size_t)]),
    "rvc_decoder_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(DecoderNoise), c_int, c_int64,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "rvc_decoder_set_tap": (c_int, [c_void_p, c_int, c_void_p]),
    "rvc_decoder_set_concurrency_hint": (c_int, [c_void_p, c_int]),
    "rvc_decoder_set_branch_parallel": (c_int, [c_void_p, c_int]),
    "rvc_posconv_bf16x3_weight_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_posconv_bf16x3_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_posconv_gelu_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "rvc_hubert_conv0_workspace_bytes": (c_int, [c_int, POINTER(c_size_t)]),
    "rvc_hubert_conv0_frames_bf16x3": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p,
                                               c_size_t, c_void_p, c_int64, c_void_p]),
    "rvc_conv1d_frames_bf16x3": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                         c_int, c_int, c_void_p]),
    "rvc_conv1d_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_conv1d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int64, c_int, c_int, c_float, c_float, c_void_p]),
    "rvc_conv1d_wino_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_conv1d_wino_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                        c_int64, c_int, c_int, c_float, c_float, c_void_p]),
    "rvc_conv1d_winobf_weight_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_conv1d_winobf_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_conv1d_winobf_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                          c_int64, c_int, c_int, c_float, c_float, c_void_p]),
    "rvc_resblock_bf16x3_weight_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "rvc_resblock_bf16x3_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rvc_resblock_bf16x3_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int,
                                            c_int, c_float, c_float, c_void_p]),
    "rvc_resblock_bf16w_weight_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "rvc_resblock_bf16w_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rvc_resblock_bf16w_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int,
                                           c_int, c_float, c_float, c_void_p]),
    "rvc_resblock_bf16x3_set_enabled": (c_int, [c_int]),
    "rvc_upsample_bf16x3_weight_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_upsample_bf16x3_pack_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_upsample_bf16x3_forward": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int,
                                            c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rvc_conv1d_bf16w_weight_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "rvc_conv1d_bf16w_pack_weight": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "rvc_conv1d_bf16w_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int,
                                         c_int, c_float, c_float, c_void_p]),
    "rvc_gemm_bf16x3_weight_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "rvc_gemm_bf16x3_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_linear_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "rvc_conv1d_bf16x3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "rvc_split_rows_bf16x3": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "rvc_linear_bf16x3_presplit": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                           c_void_p]),
    "rvc_bias_residual_layernorm_bf16x3": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                                   c_int64, c_int64, c_int, c_void_p]),
    "rvc_conv2d_packed_floats": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_conv2d_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_conv2d_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_conv2d_bf16x3_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rvc_conv2d_bf16x3_weight_bytes": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_conv2d_bf16x3_pack_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "rvc_conv2d_bf16x3_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "rvc_conv2d_bf16x3_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_size_t, c_void_p]),
    "rvc_conv2d_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "rvc_comm_unique_id": (c_int, [c_void_p]),
    "rvc_comm_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "rvc_comm_destroy": (c_int, [c_void_p]),
    "rvc_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), c_char_p, c_size_t]),
    "rvc_index_broadcast": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "rvc_checksum64": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
}

for _name, (_res, _args) in SYMBOLS.items():
    _fn = getattr(_lib, _name)  # AttributeError here = the .so is stale
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = 4
if _lib.rvc_abi_version() != ABI_VERSION:
    raise ImportError(f"librvc_amd.so ABI {_lib.rvc_abi_version()} != {ABI_VERSION}: rebuild it")


def _check(rc: int, what: str):
    if rc != 0:
        raise NativeError(f"{what}: {_lib.rvc_last_error().decode(errors='replace')}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise NativeError(f"{name} must live in HBM (got a {t.device} tensor); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise NativeError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


class _Workspace:
    """Grow-only scratch buffer per (tag, device, stream) so steady-state calls allocate nothing.  Keyed by the current
    stream as well: utterances that are in flight on different streams must not share scratch memory."""

    def __init__(self):
        self._buf = {}
        self._lock = threading.Lock()

    def get(self, tag: str, nbytes: int, device) -> torch.Tensor:
        key = (tag, str(device), torch.cuda.current_stream(device).cuda_stream)
        with self._lock:
            buf = self._buf.get(key)
            if buf is None or buf.numel() < nbytes:
                buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
                self._buf[key] = buf
        return buf


_ws = _Workspace()


# ---- K1 ------------------------------------------------------------------------------------------
def knn_index_build(index: torch.Tensor) -> torch.Tensor:
    """The per-index aux blob of rvc_knn_index_build (||x||^2, fp16 copy, screening statistics) as an opaque uint8 tensor."""
    index = _dev_f32(index, "index")
    need = c_size_t()
    _check(_lib.rvc_knn_index_aux_bytes(index.shape[0], index.shape[1], ctypes.byref(need)), "rvc_knn_index_aux_bytes")
    aux = torch.empty(need.value, dtype=torch.uint8, device=index.device)
    _check(_lib.rvc_knn_index_build(index.data_ptr(), index.shape[0], index.shape[1], aux.data_ptr(), aux.numel(), _stream()),
           "rvc_knn_index_build")
    return aux


knn_index_norms = knn_index_build   # round-1 name: the blob starts with the row norms


def knn_set_mode(mode: int) -> None:
    """0 auto, 1 exact fp32 regimes only, 2 fp16-screened whenever the shape allows (test hook; results do not change)."""
    _check(_lib.rvc_knn_set_mode(int(mode)), "rvc_knn_set_mode")


def knn_search(index: torch.Tensor, aux: torch.Tensor, queries: torch.Tensor, k: int = 8):
    index, queries = _dev_f32(index, "index"), _dev_f32(queries, "queries")
    if not aux.is_cuda or aux.dtype != torch.uint8:
        raise NativeError("aux must be the uint8 HBM blob returned by knn_index_build")
    nq = queries.shape[0]
    d2 = torch.empty((nq, k), dtype=torch.float32, device=index.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=index.device)
    if nq == 0:
        return d2, ids
    need = c_size_t()
    _check(_lib.rvc_knn_workspace_bytes(index.shape[0], nq, index.shape[1], k, ctypes.byref(need)), "rvc_knn_workspace_bytes")
    ws = _ws.get("knn", need.value, index.device)
    _check(_lib.rvc_knn_search(index.data_ptr(), aux.data_ptr(), index.shape[0], index.shape[1], queries.data_ptr(),
                               nq, k, d2.data_ptr(), ids.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
           "rvc_knn_search")
    return d2, ids


def knn_rank_candidates(index: torch.Tensor, aux: torch.Tensor, queries: torch.Tensor, cand: torch.Tensor, k: int = 8):
    """Exact top-k among the candidate rows cand [Q, cap] (int32, negative = empty) of each query."""
    index, queries = _dev_f32(index, "index"), _dev_f32(queries, "queries")
    if not cand.is_cuda or cand.dtype != torch.int32 or cand.dim() != 2 or cand.shape[0] != queries.shape[0]:
        raise NativeError("cand must be an int32 HBM tensor [n_queries, cap]")
    cand = cand.contiguous()
    nq = queries.shape[0]
    d2 = torch.empty((nq, k), dtype=torch.float32, device=index.device)
    ids = torch.empty((nq, k), dtype=torch.int64, device=index.device)
    _check(_lib.rvc_knn_rank_candidates(index.data_ptr(), aux.data_ptr(), index.shape[0], index.shape[1], queries.data_ptr(), nq,
                                        cand.data_ptr(), cand.shape[1], k, d2.data_ptr(), ids.data_ptr(), _stream()),
           "rvc_knn_rank_candidates")
    return d2, ids


def knn_blend(index: torch.Tensor, feats: torch.Tensor, d2: torch.Tensor, ids: torch.Tensor, index_rate: float):
    index, feats, d2 = _dev_f32(index, "index"), _dev_f32(feats, "feats"), _dev_f32(d2, "d2")
    ids = ids.contiguous()
    out = torch.empty_like(feats)
    _check(_lib.rvc_knn_blend(index.data_ptr(), index.shape[1], feats.data_ptr(), d2.data_ptr(), ids.data_ptr(),
                              feats.shape[0], d2.shape[1], float(index_rate), out.data_ptr(), _stream()),
           "rvc_knn_blend")
    return out


def knn_roofline_report(n_rows: int, n_queries: int, dim: int, seconds: float, peak_hbm_gbs: float,
                        peak_f32_tflops: float, peak_f16_tflops: float, traffic=None, traffic_source=None) -> dict:
    """SURVEY §8d's kNN roofline for one rvc_knn_search of (n_queries x n_rows) that took `seconds` (all launches of the
    search: query conversion, sample pass, bound, main pass, exact re-scoring).  §8d prices a pass over the index at
    n_rows * dim * 4 B (the fp32 rows the reference's algorithm reads) times ceil(Q / Qt) passes; the screened regime
    (Qt = 256) streams an fp16 copy, so the bytes it really moves per pass are half of that.  Since round 6 `frac` is the fraction of
    whatever BOUNDS the regime (see below); `hbm_frac_physical` and `frac_8d` carry the two HBM readings in every regime."""
    screened = n_queries > 64 and n_rows >= 16384 and dim % 256 == 0
    stream = n_queries <= 64
    q_tile = 256 if screened else (32 if stream else 128)
    passes = -(-n_queries // q_tile)
    bytes_8d = passes * n_rows * dim * 4.0
    bytes_moved = passes * n_rows * dim * (2.0 if screened else 4.0) + (8.0 * n_queries * dim * 4 if screened else 0.0)
    flops = 2.0 * n_queries * n_rows * dim
    mfma_peak = peak_f16_tflops if screened else peak_f32_tflops
    hbm_gbs = bytes_moved / seconds / 1e9
    mfma_tf = flops / seconds / 1e12
    # What bounds each regime decides which fraction is `frac`: the streaming regime (<= 64 queries, ONE pass over the fp32 rows) is
    # HBM-bound -- bytes / time / 8 TB/s; the screened regime (a 256 x 256 GEMM tile per block on the fp16 matrix cores, the index
    # mostly served from L2 / MALL) is bound by the matrix pipe's operand ingest -- its `frac` is the fp16 MFMA fraction, and the
    # HBM figures (physical bytes, and SURVEY 8d's formula on fp32 rows the kernel never reads) are footnotes.
    bound = "hbm" if stream else "mfma"
    out = {"kernel": ("knn_screen_kernel<false> (sample) + knn_select_kernel + knn_screen_kernel<true> (main, fp16 MFMA) + "
                      "knn_finalize_kernel (exact fp32 re-scoring)" if screened else
                      ("knn_direct_kernel" if stream else "knn_partial_kernel") + " + knn_finalize_kernel"),
           "regime": "fp16-screened, exact re-scoring" if screened else ("fp32 streaming" if stream else "fp32 GEMM"),
           "shape": f"{n_queries} queries x {n_rows} rows x {dim}", "bound": bound, "query_tile": q_tile, "passes": passes,
           "bytes_per_pass_8d": n_rows * dim * 4}
    if bound == "hbm":
        out.update({"achieved": round(hbm_gbs, 1), "peak": peak_hbm_gbs, "unit": "GB/s", "frac": round(hbm_gbs / peak_hbm_gbs, 4),
                    "frac_is": "physical: bytes the pass reads (n_rows x dim x 4) / search time / HBM peak"})
    else:
        out.update({"achieved": round(mfma_tf, 2), "peak": mfma_peak, "unit": "TFLOP/s", "frac": round(mfma_tf / mfma_peak, 4),
                    "frac_is": "2 x queries x rows x dim / search time / the dense " + ("fp16" if screened else "fp32") + " MFMA peak "
                               "(round 5 and earlier printed an HBM fraction here: now hbm_frac_physical / frac_8d)"})
    out.update({"bytes_moved_per_search": bytes_moved,
                "hbm_gbs_physical": round(hbm_gbs, 1), "hbm_frac_physical": round(hbm_gbs / peak_hbm_gbs, 4),
                "achieved_8d": round(bytes_8d / seconds / 1e9, 1), "frac_8d": round(bytes_8d / seconds / 1e9 / peak_hbm_gbs, 4),
                "frac_8d_is": "SURVEY 8d's formula (passes x n_rows x dim x 4 B / t / 8 TB/s): fp32 bytes the screened kernel never reads -- a footnote",
                "traffic": traffic, "traffic_source": traffic_source,   # measured HBM bytes per search (the caller's: profiles/pmc_knn_*.json)
                "mfma_tflops": round(mfma_tf, 2), "mfma_frac": round(mfma_tf / mfma_peak, 4),
                "mfma_peak_used": "fp16 dense 2500 TF" if screened else "fp32 157.3 TF",
                "avg_search_ms": round(seconds * 1e3, 4)})
    return out


# ---- K4 ------------------------------------------------------------------------------------------
def logmel_rmvpe(audio: torch.Tensor, pad_to: int = 32) -> tuple[torch.Tensor, int]:
    """audio [B, n] -> (log-mel [B, 128, T_padded], T) with the frame axis reflect-padded to a multiple of pad_to."""
    audio = _dev_f32(audio, "audio")
    b, n = audio.shape
    t = n // 160 + 1
    t_pad = pad_to * ((t - 1) // pad_to + 1) if pad_to > 1 else t
    mel = torch.empty((b, 128, t_pad), dtype=torch.float32, device=audio.device)
    need = c_size_t()
    _check(_lib.rvc_logmel_workspace_bytes(b, n, ctypes.byref(need)), "rvc_logmel_workspace_bytes")
    ws = _ws.get("logmel", need.value, audio.device)
    _check(_lib.rvc_logmel_rmvpe(audio.data_ptr(), b, n, mel.data_ptr(), t_pad, ws.data_ptr(), ws.numel(), _stream()),
           "rvc_logmel_rmvpe")
    return mel, t


class MelTransform:
    """Handle on rvc_mel_*: STFT magnitude / log-mel for any (n_fft, hop, window, pad, mel matrix) on the device."""

    def __init__(self, n_fft: int, hop: int, win_length: int, pad: int, mel_matrix, mag_eps: float = 0.0, log_floor: float = 1e-5):
        import numpy as np
        m = np.ascontiguousarray(mel_matrix, dtype=np.float32)
        if m.ndim != 2 or m.shape[1] != n_fft // 2 + 1:
            raise NativeError(f"mel matrix must be [n_mels, {n_fft // 2 + 1}], got {m.shape}")
        self.n_fft, self.hop, self.n_mels, self.bins = n_fft, hop, m.shape[0], n_fft // 2 + 1
        self._h = c_void_p()
        _check(_lib.rvc_mel_create(n_fft, hop, win_length, pad, float(mag_eps), float(log_floor), m.ctypes.data, m.shape[0],
                                   ctypes.byref(self._h)), "rvc_mel_create")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rvc_mel_destroy(self._h)
            self._h = None

    def n_frames(self, n_samples: int) -> int:
        t = c_int64()
        _check(_lib.rvc_mel_frames(self._h, n_samples, ctypes.byref(t)), "rvc_mel_frames")
        return t.value

    def forward(self, audio: torch.Tensor, want_mel: bool = True, want_spec: bool = False):
        """audio [B, n] float32 on the device -> (log-mel [B, n_mels, T] or None, magnitude [B, n_fft/2+1, T] or None)."""
        audio = _dev_f32(audio, "audio")
        b, n = audio.shape
        t = self.n_frames(n)
        mel = torch.empty((b, self.n_mels, t), dtype=torch.float32, device=audio.device) if want_mel else None
        spec = torch.empty((b, self.bins, t), dtype=torch.float32, device=audio.device) if want_spec else None
        need = c_size_t()
        _check(_lib.rvc_mel_workspace_bytes(self._h, b, n, ctypes.byref(need)), "rvc_mel_workspace_bytes")
        ws = _ws.get("mel", need.value, audio.device)
        _check(_lib.rvc_mel_forward(self._h, audio.data_ptr(), b, n, mel.data_ptr() if mel is not None else None,
                                    spec.data_ptr() if spec is not None else None, ws.data_ptr(), ws.numel(), _stream()),
               "rvc_mel_forward")
        return mel, spec


def resample_poly(x: torch.Tensor, up: int, down: int, h: torch.Tensor) -> torch.Tensor:
    """x [n] float64 on the device, h the FIR (float64, device) -> ceil(n * up / down) samples (resample_poly convention)."""
    if not (x.is_cuda and x.dtype == torch.float64 and x.dim() == 1 and h.is_cuda and h.dtype == torch.float64):
        raise NativeError("resample_poly wants 1-D float64 HBM tensors")
    x, h = x.contiguous(), h.contiguous()
    n_out = -(-x.numel() * up // down)
    y = torch.empty(n_out, dtype=torch.float64, device=x.device)
    _check(_lib.rvc_resample_poly_f64(x.data_ptr(), x.numel(), int(up), int(down), h.data_ptr(), h.numel(), y.data_ptr(), n_out,
                                      _stream()), "rvc_resample_poly_f64")
    return y


# ---- K6 ------------------------------------------------------------------------------------------
_ff_coef_cache = {}


def _filtfilt_coef(b, a):
    import numpy as np
    from scipy import signal
    key = (tuple(b), tuple(a))
    if key not in _ff_coef_cache:
        b = np.asarray(b, dtype=np.float64)
        a = np.asarray(a, dtype=np.float64)
        assert b.shape == (6,) and a.shape == (6,) and a[0] == 1.0
        _ff_coef_cache[key] = np.ascontiguousarray(np.concatenate([b, a, signal.lfilter_zi(b, a)]))
    return _ff_coef_cache[key]


def filtfilt_order5(x: torch.Tensor, b, a) -> torch.Tensor:
    """scipy.signal.filtfilt(b, a, x) for a 5th-order filter; x: 1-D float64 on the device."""
    if not x.is_cuda or x.dtype != torch.float64 or x.dim() != 1:
        raise NativeError("filtfilt_order5 wants a 1-D float64 HBM tensor")
    x = x.contiguous()
    coef = _filtfilt_coef(b, a)
    y = torch.empty_like(x)
    need = c_size_t()
    _check(_lib.rvc_filtfilt_workspace_bytes(x.numel(), ctypes.byref(need)), "rvc_filtfilt_workspace_bytes")
    ws = _ws.get("filtfilt", need.value, x.device)
    _check(_lib.rvc_filtfilt_order5(x.data_ptr(), x.numel(), coef.ctypes.data, y.data_ptr(), ws.data_ptr(),
                                    ws.numel(), _stream()), "rvc_filtfilt_order5")
    return y


# ---- K5 ------------------------------------------------------------------------------------------
def bigru_forward(gi: torch.Tensor, whh_t: torch.Tensor, bhh: torch.Tensor, multi_cu: bool = True) -> torch.Tensor:
    """gi [B,T,2,768] (input projections) -> [B,T,512]; see include/rvc_amd.h."""
    gi, whh_t, bhh = _dev_f32(gi, "gi"), _dev_f32(whh_t, "whh_t"), _dev_f32(bhh, "bhh")
    b, t = gi.shape[0], gi.shape[1]
    out = torch.empty((b, t, 512), dtype=torch.float32, device=gi.device)
    ws_ptr, ws_bytes = None, 0
    if multi_cu:
        need = c_size_t()
        _check(_lib.rvc_bigru_workspace_bytes(b, ctypes.byref(need)), "rvc_bigru_workspace_bytes")
        ws = _ws.get("bigru", need.value, gi.device)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    _check(_lib.rvc_bigru_forward(gi.data_ptr(), whh_t.data_ptr(), bhh.data_ptr(), out.data_ptr(), b, t, 256, ws_ptr,
                                  ws_bytes, _stream()), "rvc_bigru_forward")
    return out


def bigru_redone(batch: int = 1, device=None) -> int:
    """How many (batch item, direction) sequences of the last multi-workgroup forward on the current stream's workspace
    had to be recomputed by the single-workgroup kernel (synchronises the stream)."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    need = c_size_t()
    _check(_lib.rvc_bigru_workspace_bytes(batch, ctypes.byref(need)), "rvc_bigru_workspace_bytes")
    ws = _ws.get("bigru", need.value, device)
    n = c_int()
    _check(_lib.rvc_bigru_status(ws.data_ptr(), batch, ctypes.byref(n), _stream()), "rvc_bigru_status")
    return n.value


def bigru_set_spin_limit(polls: int) -> None:
    _check(_lib.rvc_bigru_set_spin_limit(int(polls)), "rvc_bigru_set_spin_limit")


# ---- HuBERT attention -------------------------------------------------------------------------------
def attention_qkv(qkv: torch.Tensor, n_heads: int, scale: float, emb_rel_k: torch.Tensor = None,
                  emb_rel_v: torch.Tensor = None) -> torch.Tensor:
    """qkv [B, T, 3 * n_heads * d] (fused projection output, d = 64 or 96) -> softmax(q k^T * scale [+ rel]) v [+ rel]
    as [B, T, n_heads * d].  emb_rel_k / emb_rel_v [21, d]: the TextEncoder's relative-position embeddings."""
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.is_contiguous() and qkv.dim() == 3
    b, t, c = qkv.shape
    hd = c // (3 * n_heads)
    rel = emb_rel_k is not None
    if rel:
        for e in (emb_rel_k, emb_rel_v):
            assert e.is_cuda and e.dtype == torch.float32 and e.is_contiguous() and tuple(e.shape) == (21, hd)
    out = torch.empty(b, t, n_heads * hd, dtype=torch.float32, device=qkv.device)
    need = c_size_t()
    _check(_lib.rvc_attention_workspace_bytes(b, t, n_heads, hd, ctypes.byref(need)), "rvc_attention_workspace_bytes")
    ws = _ws.get("attention", need.value, qkv.device)
    _check(_lib.rvc_attention_qkv_f32(qkv.data_ptr(), emb_rel_k.data_ptr() if rel else None,
                                      emb_rel_v.data_ptr() if rel else None, out.data_ptr(), b, t, n_heads, hd,
                                      float(scale), ws.data_ptr(), ws.numel(), _stream()), "rvc_attention_qkv_f32")
    return out


# ---- RMVPE U-Net conv epilogue ------------------------------------------------------------------------
def bias_relu_add_(x: torch.Tensor, bias: torch.Tensor = None, res: torch.Tensor = None, relu: bool = True) -> torch.Tensor:
    """In place: x = relu(x + bias[c]) + res for x [B, C, ...] (contiguous, trailing extent a multiple of 4)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() >= 3
    b, c = x.shape[0], x.shape[1]
    inner = x.numel() // (b * c)
    if res is not None:
        assert res.is_cuda and res.dtype == torch.float32 and res.is_contiguous() and res.shape == x.shape
    _check(_lib.rvc_bias_relu_add_f32(x.data_ptr(), bias.data_ptr() if bias is not None else None,
                                      res.data_ptr() if res is not None else None, x.data_ptr(), b, c, inner, int(relu),
                                      _stream()), "rvc_bias_relu_add_f32")
    return x


def rownorm_gelu_(x: torch.Tensor, gamma: torch.Tensor = None, beta: torch.Tensor = None, eps: float = 1e-5) -> torch.Tensor:
    """In place: x = gelu(group_norm(x, num_groups = channels)) for x [B, C, L] (fp32, HBM, contiguous)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    b, c, length = x.shape
    _check(_lib.rvc_rownorm_gelu_f32(x.data_ptr(), gamma.data_ptr() if gamma is not None else None,
                                     beta.data_ptr() if beta is not None else None, x.data_ptr(), b, c, length, float(eps), _stream()),
           "rvc_rownorm_gelu_f32")
    return x


def gate_tanh_sigmoid(x: torch.Tensor) -> torch.Tensor:
    """x [B, 2H, T] -> tanh(x[:, :H]) * sigmoid(x[:, H:]) [B, H, T]"""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3 and x.shape[1] % 2 == 0
    b, c2, t = x.shape
    out = torch.empty(b, c2 // 2, t, dtype=torch.float32, device=x.device)
    _check(_lib.rvc_gate_tanh_sigmoid_f32(x.data_ptr(), out.data_ptr(), b, c2 // 2, t, _stream()), "rvc_gate_tanh_sigmoid_f32")
    return out


def set_concurrency_hint(utterances_in_flight: int) -> None:
    _check(_lib.rvc_set_concurrency_hint(int(utterances_in_flight)), "rvc_set_concurrency_hint")


# ---- multi-GPU: RCCL index broadcast + device checksum ----------------------------------------------
COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(_lib.rvc_comm_unique_id(buf), "rvc_comm_unique_id")
    return buf.raw


class Comm:
    """RCCL communicator of the job (one rank per GPU), created collectively from rank 0's 128-byte id."""

    def __init__(self, unique_id: bytes, n_ranks: int, rank: int):
        assert len(unique_id) == COMM_ID_BYTES
        self._h = c_void_p()
        _check(_lib.rvc_comm_create(ctypes.create_string_buffer(unique_id, COMM_ID_BYTES), int(n_ranks), int(rank),
                                    ctypes.byref(self._h)), "rvc_comm_create")

    def info(self) -> dict:
        n, r, v = c_int(), c_int(), c_int()
        path = ctypes.create_string_buffer(512)
        _check(_lib.rvc_comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(v), path, 512), "rvc_comm_info")
        return {"n_ranks": n.value, "rank": r.value, "rccl_version": v.value, "library": path.value.decode()}

    def broadcast_(self, t: torch.Tensor, root: int = 0) -> torch.Tensor:
        """In-place broadcast of a contiguous HBM tensor's bytes from ``root`` on torch's current stream."""
        if not t.is_cuda or not t.is_contiguous():
            raise NativeError("Comm.broadcast_ wants a contiguous HBM tensor")
        _check(_lib.rvc_index_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), int(root), _stream()),
               "rvc_index_broadcast")
        return t

    def destroy(self):
        h, self._h = self._h, None
        if h:
            _check(_lib.rvc_comm_destroy(h), "rvc_comm_destroy")

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.rvc_comm_destroy(self._h)
            self._h = None


def checksum64(t: torch.Tensor) -> torch.Tensor:
    """[2] int64 device tensor: (sum of 32-bit words, sum of (i+1)*word_i) mod 2^64 of the tensor's bytes, computed in HBM."""
    if not t.is_cuda or not t.is_contiguous():
        raise NativeError("checksum64 wants a contiguous HBM tensor")
    out = torch.empty(2, dtype=torch.int64, device=t.device)
    _check(_lib.rvc_checksum64(t.data_ptr(), t.numel() * t.element_size(), out.data_ptr(), _stream()), "rvc_checksum64")
    return out


# ---- conv1d (unit-test entry) ----------------------------------------------------------------------
def conv1d_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    w = w.detach().float().cpu().contiguous()
    packed = torch.empty(w.numel(), dtype=torch.float32, device=device)
    _check(_lib.rvc_conv1d_pack_weight(w.data_ptr(), w.shape[0], w.shape[1], w.shape[2], packed.data_ptr(), _stream()),
           "rvc_conv1d_pack_weight")
    return packed


def conv1d_forward(x, w_packed, bias, c_out, k, dilation=1, slope_in=1.0, res=None, acc=None, out_scale=1.0):
    return conv1d_forward_into(x, w_packed, bias, c_out, k, dilation, slope_in, res=res, acc=acc, out_scale=out_scale)


def conv1d_forward_into(x, w_packed, bias, c_out, k, dilation=1, slope_in=1.0, res=None, acc=None, out_scale=1.0, out=None):
    x = _dev_f32(x, "x")
    b, c_in, length = x.shape
    y = out if out is not None else torch.empty((b, c_out, length), dtype=torch.float32, device=x.device)
    _check(_lib.rvc_conv1d_forward(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                   res.data_ptr() if res is not None else None,
                                   acc.data_ptr() if acc is not None else None, y.data_ptr(), b, c_in, c_out, length, k,
                                   dilation, float(slope_in), float(out_scale), _stream()), "rvc_conv1d_forward")
    return y


def conv1d_wino_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    w = w.detach().float().cpu().contiguous()
    c_out, c_in, k = w.shape
    u = torch.empty(3 * ((k + 2) // 3) * c_in * c_out, dtype=torch.float32, device=device)
    _check(_lib.rvc_conv1d_wino_pack_weight(w.data_ptr(), c_out, c_in, k, u.data_ptr(), _stream()), "rvc_conv1d_wino_pack_weight")
    return u


def conv1d_wino_forward(x, u_packed, bias, c_out, k, dilation=1, slope_in=1.0, res=None, acc=None, out_scale=1.0, out=None):
    x = _dev_f32(x, "x")
    b, c_in, length = x.shape
    y = out if out is not None else torch.empty((b, c_out, length), dtype=torch.float32, device=x.device)
    _check(_lib.rvc_conv1d_wino_forward(x.data_ptr(), u_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        res.data_ptr() if res is not None else None,
                                        acc.data_ptr() if acc is not None else None, y.data_ptr(), b, c_in, c_out, length, k,
                                        dilation, float(slope_in), float(out_scale), _stream()), "rvc_conv1d_wino_forward")
    return y


def conv1d_winobf_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    w = w.detach().float().cpu().contiguous()
    c_out, c_in, k = w.shape
    n = c_size_t()
    _check(_lib.rvc_conv1d_winobf_weight_bytes(c_out, c_in, k, ctypes.byref(n)), "rvc_conv1d_winobf_weight_bytes")
    u = torch.empty(n.value // 2, dtype=torch.int16, device=device)
    _check(_lib.rvc_conv1d_winobf_pack_weight(w.data_ptr(), c_out, c_in, k, u.data_ptr(), _stream()), "rvc_conv1d_winobf_pack_weight")
    return u


def conv1d_winobf_forward(x, u_packed, bias, c_out, k, dilation=1, slope_in=1.0, res=None, acc=None, out_scale=1.0, out=None):
    x = _dev_f32(x, "x")
    b, c_in, length = x.shape
    y = out if out is not None else torch.empty((b, c_out, length), dtype=torch.float32, device=x.device)
    _check(_lib.rvc_conv1d_winobf_forward(x.data_ptr(), u_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          res.data_ptr() if res is not None else None,
                                          acc.data_ptr() if acc is not None else None, y.data_ptr(), b, c_in, c_out, length, k,
                                          dilation, float(slope_in), float(out_scale), _stream()), "rvc_conv1d_winobf_forward")
    return y


# ---- K3f: fused ResBlock pair (dilated conv -> conv + residual) on the bf16 matrix cores -------------------------
def resblock_bf16x3_pack_weight(w1: torch.Tensor, w2: torch.Tensor, device, bf16_taps: bool = False) -> torch.Tensor:
    """The two nn.Conv1d weights [c, c, k] of one ResBlock dilation -> fragment slab on the device.  bf16_taps: the taps rounded to
    bf16 (BASELINE cfg 4's weight storage), one fragment per group instead of three -- for resblock_bf16x3_forward(bf16_taps=True)."""
    w1 = w1.detach().float().cpu().contiguous()
    w2 = w2.detach().float().cpu().contiguous()
    c, c_in, k = w1.shape
    if c != c_in or w2.shape != w1.shape:
        raise NativeError(f"resblock pair: square convs of one shape expected, got {tuple(w1.shape)} and {tuple(w2.shape)}")
    stem = "rvc_resblock_bf16w" if bf16_taps else "rvc_resblock_bf16x3"
    n = c_size_t()
    _check(getattr(_lib, stem + "_weight_bytes")(c, k, ctypes.byref(n)), stem + "_weight_bytes")
    u = torch.empty(n.value // 2, dtype=torch.int16, device=device)
    _check(getattr(_lib, stem + "_pack_weight")(w1.data_ptr(), w2.data_ptr(), c, k, u.data_ptr(), _stream()), stem + "_pack_weight")
    return u


def resblock_bf16x3_forward(x, u_packed, b1, b2, k, dilation=1, slope=0.1, acc=None, out_scale=1.0, out=None, bf16_taps: bool = False):
    """y = out_scale * (conv2(leaky(conv1_d(leaky(x)) + b1)) + b2 + x [+ acc]) for x [B, C, L] in HBM (residuals.py:75-86).
    bf16_taps: u_packed holds one-term fragments of bf16-valued taps (three products per multiply-add; hifigan_mrf.py:13-83 at cfg 4)."""
    x = _dev_f32(x, "x")
    b, c, length = x.shape
    y = out if out is not None else torch.empty_like(x)
    if y.data_ptr() == x.data_ptr():
        raise NativeError("resblock pair: x and y must not alias")
    name = "rvc_resblock_bf16w_forward" if bf16_taps else "rvc_resblock_bf16x3_forward"
    _check(getattr(_lib, name)(x.data_ptr(), u_packed.data_ptr(), b1.data_ptr() if b1 is not None else None,
                               b2.data_ptr() if b2 is not None else None, acc.data_ptr() if acc is not None else None,
                               y.data_ptr(), b, c, length, k, dilation, float(slope), float(out_scale), _stream()), name)
    return y


# ---- K3d: one square conv with bf16-valued taps, direct form (C = 128 / 256) --------------------------------------
def conv1d_bf16w_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    """nn.Conv1d weight [c, c, k] -> one-term direct-form fragment slab on the device (the taps ROUNDED to bf16)."""
    w = w.detach().float().cpu().contiguous()
    c, c_in, k = w.shape
    if c != c_in:
        raise NativeError(f"conv1d_bf16w: a square conv expected, got {tuple(w.shape)}")
    n = c_size_t()
    _check(_lib.rvc_conv1d_bf16w_weight_bytes(c, k, ctypes.byref(n)), "rvc_conv1d_bf16w_weight_bytes")
    u = torch.empty(n.value // 2, dtype=torch.int16, device=device)
    _check(_lib.rvc_conv1d_bf16w_pack_weight(w.data_ptr(), c, k, u.data_ptr(), _stream()), "rvc_conv1d_bf16w_pack_weight")
    return u


def conv1d_bf16w_forward(x, u_packed, bias, k, dilation=1, slope_in=1.0, res=None, acc=None, out_scale=1.0, out=None):
    """y = out_scale * (conv_d(leaky(x, slope_in)) + bias [+ res] [+ acc]) for x [B, C, L] in HBM, C = 128 / 256 (K3d)."""
    x = _dev_f32(x, "x")
    b, c, length = x.shape
    y = out if out is not None else torch.empty_like(x)
    if y.data_ptr() == x.data_ptr():
        raise NativeError("conv1d_bf16w: x and y must not alias")
    _check(_lib.rvc_conv1d_bf16w_forward(x.data_ptr(), u_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                         res.data_ptr() if res is not None else None, acc.data_ptr() if acc is not None else None,
                                         y.data_ptr(), b, c, length, k, dilation, float(slope_in), float(out_scale), _stream()),
           "rvc_conv1d_bf16w_forward")
    return y


# ---- K3u: upsampling step (polyphase ConvTranspose1d + folded noise conv) on the bf16 matrix cores -----------------
def upsample_bf16x3_pack_weight(up_w: torch.Tensor, noise_w, bias, rate: int, nc_stride: int, device):
    """ConvTranspose1d weight [c_in, c_out, ksize] (+ the noise conv's weight [c_out, 1, nc_k] or None, + the summed bias [c_out] or None)
    -> (fragment slab, noise weights [c_out, nc_k] or None, bias or None) on the device, the triple `upsample_bf16x3_forward` takes."""
    up_w = up_w.detach().float().cpu().contiguous()
    c_in, c_out, ksize = up_w.shape
    nc_k, nw = 0, None
    if noise_w is not None:
        nw = noise_w.detach().float().cpu().reshape(c_out, -1).contiguous()
        nc_k = nw.shape[1]
    b = bias.detach().float().cpu().contiguous() if bias is not None else None
    n = c_size_t()
    _check(_lib.rvc_upsample_bf16x3_weight_bytes(c_in, c_out, rate, ksize, nc_k, nc_stride, ctypes.byref(n)), "rvc_upsample_bf16x3_weight_bytes")
    u = torch.empty(n.value // 2, dtype=torch.int16, device=device)
    _check(_lib.rvc_upsample_bf16x3_pack_weight(up_w.data_ptr(), nw.data_ptr() if nw is not None else None, b.data_ptr() if b is not None else None,
                                                c_in, c_out, rate, ksize, nc_k, nc_stride, u.data_ptr(), _stream()), "rvc_upsample_bf16x3_pack_weight")
    return u, (nw.to(device) if nw is not None else None), (b.to(device) if b is not None else None)


def upsample_bf16x3_forward(x, har, packed, c_out, rate, ksize, pad, nc_stride=1, nc_pad=0, slope=0.1):
    """y [B, c_out, (L - 1) rate - 2 pad + ksize] = bias + conv_transpose1d(leaky(x)) + noise_conv(har) for x [B, c_in, L], har [B, Lh]."""
    u, nw, bias = packed
    x = _dev_f32(x, "x")
    b, c_in, length = x.shape
    l_out = (length - 1) * rate - 2 * pad + ksize
    y = torch.empty((b, c_out, l_out), dtype=torch.float32, device=x.device)
    if har is not None:
        har = _dev_f32(har, "har")
    _check(_lib.rvc_upsample_bf16x3_forward(x.data_ptr(), har.data_ptr() if har is not None else None, har.shape[-1] if har is not None else 0,
                                            u.data_ptr(), bias.data_ptr() if bias is not None else None,
                                            y.data_ptr(), b, c_in, c_out, length, rate, ksize, pad, nw.shape[1] if nw is not None else 0, nc_stride,
                                            nc_pad, float(slope), _stream()), "rvc_upsample_bf16x3_forward")
    return y


def resblock_bf16x3_set_enabled(enabled: bool) -> None:
    """Process-wide: decoder handles finalized while this is off keep their narrow ResBlock stages on the unfused kernels."""
    _check(_lib.rvc_resblock_bf16x3_set_enabled(1 if enabled else 0), "rvc_resblock_bf16x3_set_enabled")


# ---- K11: fp32 GEMM / strided conv1d as exact bf16x3 splits on the bf16 matrix cores ----------------------------
def gemm_bf16x3_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    """nn.Linear weight [out, in] or nn.Conv1d weight [c_out, c_in, taps] -> fragment slab on the device."""
    w = w.detach().float().cpu().contiguous()
    taps = w.shape[2] if w.dim() == 3 else 1
    m, k_total = w.shape[0], w.shape[1] * taps
    n = c_size_t()
    _check(_lib.rvc_gemm_bf16x3_weight_bytes(m, k_total, ctypes.byref(n)), "rvc_gemm_bf16x3_weight_bytes")
    a = torch.empty(n.value // 2, dtype=torch.int16, device=device)
    _check(_lib.rvc_gemm_bf16x3_pack_weight(w.data_ptr(), m, k_total, taps, a.data_ptr(), _stream()), "rvc_gemm_bf16x3_pack_weight")
    return a


def linear_bf16x3(x: torch.Tensor, a_packed: torch.Tensor, bias, out_features: int, act: str = "none", res=None) -> torch.Tensor:
    """y = act(x W^T + b) + res for x [..., in] (fp32, HBM, contiguous)."""
    x = _dev_f32(x, "x")
    in_features = x.shape[-1]
    n_rows = x.numel() // in_features
    y = torch.empty(x.shape[:-1] + (out_features,), dtype=torch.float32, device=x.device)
    if res is not None:
        res = _dev_f32(res, "res")
        assert res.shape == y.shape
    _check(_lib.rvc_linear_bf16x3(x.data_ptr(), a_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                  res.data_ptr() if res is not None else None, y.data_ptr(), n_rows, in_features, out_features,
                                  {"none": 0, "gelu": 1}[act], _stream()), "rvc_linear_bf16x3")
    return y


# ---- K12: HuBERT's transformer GEMMs with both operands pre-split (csrc/linbf.hip) ---------------------------------------
def rows_padded(n_rows: int, tile: int = 128) -> int:
    return -(-n_rows // tile) * tile


def planes_empty(n_rows: int, features: int, device) -> torch.Tensor:
    """Uninitialised [3][rows_padded][features] bf16 planes (the padding rows are never read into a stored result)."""
    return torch.empty((3, rows_padded(n_rows), features), dtype=torch.bfloat16, device=device)


def split_rows_bf16x3(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """fp32 [n_rows, k] -> three bf16 planes whose sum is x exactly."""
    x = _dev_f32(x, "x")
    n_rows, k = x.shape
    xs = out if out is not None else planes_empty(n_rows, k, x.device)
    _check(_lib.rvc_split_rows_bf16x3(x.data_ptr(), xs.data_ptr(), n_rows, xs.shape[1], k, _stream()), "rvc_split_rows_bf16x3")
    return xs


def linear_bf16x3_presplit(xs: torch.Tensor, a_packed: torch.Tensor, bias, n_rows: int, out_features: int, mode: str = "f32",
                           k_parts: int = 1, out: torch.Tensor | None = None) -> torch.Tensor:
    """mode "f32": y [n_rows, out] = x W^T + b; "gelu_planes": planes of gelu(x W^T + b); "parts": [k_parts, n_rows, out] partial sums."""
    _, n_pad, in_features = xs.shape
    m = {"f32": 0, "gelu_planes": 1, "parts": 2, "gelu_f32": 3}[mode]
    if m in (0, 3):
        y = out if out is not None else torch.empty((n_rows, out_features), dtype=torch.float32, device=xs.device)
    elif m == 1:
        y = out if out is not None else planes_empty(n_rows, out_features, xs.device)
    else:
        y = out if out is not None else torch.empty((k_parts, n_rows, out_features), dtype=torch.float32, device=xs.device)
    _check(_lib.rvc_linear_bf16x3_presplit(xs.data_ptr(), a_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           y.data_ptr() if m != 1 else None, y.data_ptr() if m == 1 else None, n_rows, n_pad,
                                           in_features, out_features, m, k_parts, _stream()), "rvc_linear_bf16x3_presplit")
    return y


def posconv_bf16x3_pack_weight(w: torch.Tensor, groups: int, device) -> torch.Tensor:
    """conv.weight [D, D / groups, taps] (weight norm folded) -> K14's fragment slab in HBM."""
    w = w.detach().float().cpu().contiguous()
    d, cg, taps = w.shape
    need = c_size_t()
    _check(_lib.rvc_posconv_bf16x3_weight_bytes(d, groups, taps, ctypes.byref(need)), "rvc_posconv_bf16x3_weight_bytes")
    a = torch.empty(need.value // 2, dtype=torch.int16, device=device)
    with torch.cuda.device(a.device):
        _check(_lib.rvc_posconv_bf16x3_pack_weight(w.data_ptr(), d, groups, taps, a.data_ptr(), _stream()), "rvc_posconv_bf16x3_pack_weight")
    return a


def posconv_gelu_bf16x3(x: torch.Tensor, a_packed: torch.Tensor, bias, groups: int, taps: int, padding: int) -> torch.Tensor:
    """gelu(grouped Conv1d over the frames of x [T, D], output frames 0 .. T - 1) -> [T, D] fp32."""
    x = _dev_f32(x, "x")
    t, d = x.shape
    y = torch.empty_like(x)
    _check(_lib.rvc_posconv_gelu_bf16x3(x.data_ptr(), a_packed.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                        t, d, groups, taps, padding, _stream()), "rvc_posconv_gelu_bf16x3")
    return y


def hubert_conv0_frames_bf16x3(wav: torch.Tensor, w: torch.Tensor, gamma, beta, eps: float, stride: int = 5) -> tuple[torch.Tensor, int]:
    """HuBERT's first feature-extractor layer (Conv1d(1, C, 10, stride 5) -> GroupNorm(C, C) -> GELU) of ONE clip wav [n] ->
    (time-major planes [3, frames_padded, C], frames)."""
    wav = _dev_f32(wav, "wav")
    w2 = _dev_f32(w.reshape(w.shape[0], -1), "w")
    c, taps = w2.shape
    n = wav.numel()
    frames = (n - taps) // stride + 1
    ys = planes_empty(frames, c, wav.device)
    need = c_size_t()
    _check(_lib.rvc_hubert_conv0_workspace_bytes(c, ctypes.byref(need)), "rvc_hubert_conv0_workspace_bytes")
    ws = _ws.get("hubert_conv0", need.value, wav.device)
    _check(_lib.rvc_hubert_conv0_frames_bf16x3(wav.data_ptr(), n, w2.data_ptr(), c, taps, stride,
                                               gamma.data_ptr() if gamma is not None else None,
                                               beta.data_ptr() if beta is not None else None, float(eps), ws.data_ptr(), ws.numel(),
                                               ys.data_ptr(), ys.shape[1], _stream()), "rvc_hubert_conv0_frames_bf16x3")
    return ys, frames


def conv1d_frames_bf16x3(xs: torch.Tensor, n_frames_in: int, a_packed: torch.Tensor, bias, out_channels: int, taps: int, stride: int,
                         mode: str = "gelu_planes") -> tuple[torch.Tensor, int]:
    """nn.Conv1d(C, out, taps, stride, no padding) (+ GELU) over time-major planes xs [3, frames_in_padded, C] ->
    (planes [3, frames_out_padded, out] or fp32 [frames_out, out], frames_out).  a_packed: gemm_bf16x3_pack_weight of
    conv.weight.permute(0, 2, 1).reshape(out, taps * C)."""
    _, n_pad_in, c = xs.shape
    m = {"f32": 0, "gelu_planes": 1, "gelu_f32": 3}[mode]
    frames = (n_frames_in - taps) // stride + 1
    if frames <= 0:
        raise NativeError("conv1d_frames_bf16x3: the input is shorter than one window")
    y = planes_empty(frames, out_channels, xs.device) if m == 1 else torch.empty((frames, out_channels), dtype=torch.float32, device=xs.device)
    _check(_lib.rvc_conv1d_frames_bf16x3(xs.data_ptr(), n_frames_in, n_pad_in, c, taps, stride, a_packed.data_ptr(),
                                         bias.data_ptr() if bias is not None else None, y.data_ptr() if m != 1 else None,
                                         y.data_ptr() if m == 1 else None, rows_padded(frames), out_channels, m, _stream()),
           "rvc_conv1d_frames_bf16x3")
    return y, frames


def bias_residual_layernorm_bf16x3(parts: torch.Tensor, bias, res, gamma, beta, eps: float, want_planes: bool = True,
                                   y: torch.Tensor | None = None, ys: torch.Tensor | None = None):
    """LayerNorm(sum(parts) + bias + res) -> (fp32 [n_rows, m], planes or None)."""
    n_parts, n_rows, m = parts.shape
    y = y if y is not None else torch.empty((n_rows, m), dtype=torch.float32, device=parts.device)
    if want_planes and ys is None:
        ys = planes_empty(n_rows, m, parts.device)
    _check(_lib.rvc_bias_residual_layernorm_bf16x3(parts.data_ptr(), n_parts, bias.data_ptr() if bias is not None else None,
                                                   res.data_ptr() if res is not None else None,
                                                   gamma.data_ptr() if gamma is not None else None,
                                                   beta.data_ptr() if beta is not None else None, float(eps), y.data_ptr(),
                                                   ys.data_ptr() if ys is not None else None, n_rows, ys.shape[1] if ys is not None else n_rows,
                                                   m, _stream()), "rvc_bias_residual_layernorm_bf16x3")
    return y, ys


def conv1d_bf16x3(x: torch.Tensor, a_packed: torch.Tensor, bias, c_out: int, k: int, stride: int = 1, padding: int = 0,
                  act: str = "none") -> torch.Tensor:
    """nn.Conv1d(c_in, c_out, k, stride, padding) + activation for x [batch, c_in, L] (fp32, HBM, contiguous)."""
    x = _dev_f32(x, "x")
    b, c_in, l_in = x.shape
    l_out = (l_in + 2 * padding - k) // stride + 1
    y = torch.empty((b, c_out, l_out), dtype=torch.float32, device=x.device)
    _check(_lib.rvc_conv1d_bf16x3(x.data_ptr(), a_packed.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(), b,
                                  c_in, c_out, l_in, k, stride, padding, {"none": 0, "gelu": 1}[act], _stream()), "rvc_conv1d_bf16x3")
    return y


# ---- K10: conv2d 3x3 / 1x1 of the RMVPE U-Net -------------------------------------------------------------
def conv2d_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    """torch conv2d weight [C_out, C_in, kh, kw] (3x3 or 1x1) -> packed taps in HBM."""
    w = w.detach().float().cpu().contiguous()
    c_out, c_in, kh, kw = w.shape
    n = c_size_t()
    _check(_lib.rvc_conv2d_packed_floats(c_out, c_in, kh, kw, ctypes.byref(n)), "rvc_conv2d_packed_floats")
    out = torch.empty(n.value, dtype=torch.float32, device=device)
    _check(_lib.rvc_conv2d_pack_weight(w.data_ptr(), c_out, c_in, kh, kw, out.data_ptr(), _stream()), "rvc_conv2d_pack_weight")
    return out


def conv2d_supported(c_in: int, width: int) -> bool:
    return c_in % 8 == 0 and 4 <= width <= 128 and width & (width - 1) == 0


def conv2d_forward(x, w_packed, bias, c_out, ksize=3, relu=False, res=None, out=None):
    """y = act(conv2d(x, w, padding=ksize // 2) + bias) + res  (x [B, C_in, H, W] float32 on the device)."""
    x = _dev_f32(x, "x")
    b, c_in, h, wd = x.shape
    y = out if out is not None else torch.empty((b, c_out, h, wd), dtype=torch.float32, device=x.device)
    need = c_size_t()
    _check(_lib.rvc_conv2d_workspace_bytes(b, c_in, c_out, h, wd, ksize, ksize, ctypes.byref(need)), "rvc_conv2d_workspace_bytes")
    ws = _ws.get("conv2d", need.value, x.device) if need.value else None
    _check(_lib.rvc_conv2d_forward(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                   res.data_ptr() if res is not None else None, y.data_ptr(), b, c_in, c_out, h, wd, ksize, ksize,
                                   1 if relu else 0, ws.data_ptr() if ws is not None else None, need.value, _stream()),
           "rvc_conv2d_forward")
    return y


# ---- K10b: the same 3x3 conv as exact bf16x3 products on the bf16 matrix cores ------------------------------
def conv2d_bf16x3_supported(c_in: int, c_out: int, height: int, width: int) -> bool:
    return bool(_lib.rvc_conv2d_bf16x3_supported(c_in, c_out, height, width))


def conv2d_bf16x3_packable(c_in: int, c_out: int) -> bool:
    m_pad = (c_out + 31) // 32 * 32
    return c_in % 16 == 0 and (m_pad in (32, 64) or m_pad % 128 == 0)


def conv2d_bf16x3_pack_weight(w: torch.Tensor, device) -> torch.Tensor:
    """torch conv2d weight [C_out, C_in, 3, 3] -> bf16x3 tap fragments in HBM (int16 words)."""
    w = w.detach().float().cpu().contiguous()
    c_out, c_in, kh, kw = w.shape
    n = c_size_t()
    _check(_lib.rvc_conv2d_bf16x3_weight_bytes(c_out, c_in, kh, kw, ctypes.byref(n)), "rvc_conv2d_bf16x3_weight_bytes")
    out = torch.empty(n.value // 2, dtype=torch.int16, device=device)
    _check(_lib.rvc_conv2d_bf16x3_pack_weight(w.data_ptr(), c_out, c_in, kh, kw, out.data_ptr(), _stream()), "rvc_conv2d_bf16x3_pack_weight")
    return out


def conv2d_bf16x3_forward(x, u, bias, c_out, relu=False, res=None, out=None):
    """y = act(conv2d(x, w, padding=1) + bias) + res  (x [B, C_in, H, W] float32 on the device; u from conv2d_bf16x3_pack_weight)."""
    x = _dev_f32(x, "x")
    b, c_in, h, wd = x.shape
    y = out if out is not None else torch.empty((b, c_out, h, wd), dtype=torch.float32, device=x.device)
    need = c_size_t()
    _check(_lib.rvc_conv2d_bf16x3_workspace_bytes(b, c_in, c_out, h, wd, ctypes.byref(need)), "rvc_conv2d_bf16x3_workspace_bytes")
    ws = _ws.get("conv2d", need.value, x.device) if need.value else None
    _check(_lib.rvc_conv2d_bf16x3_forward(x.data_ptr(), u.data_ptr(), bias.data_ptr() if bias is not None else None,
                                          res.data_ptr() if res is not None else None, y.data_ptr(), b, c_in, c_out, h, wd,
                                          1 if relu else 0, ws.data_ptr() if ws is not None else None, need.value, _stream()),
           "rvc_conv2d_bf16x3_forward")
    return y


# ---- K2/K3 -----------------------------------------------------------------------------------------
DEC_KINDS = {"HiFi-GAN": 0, "MRF HiFi-GAN": 1, "RefineGAN": 2}


class Decoder:
    """Handle on the library's vocoder: weights are repacked into HBM once, forward() launches the chain."""

    def __init__(self, vocoder: str, sr: int, folded_weights: dict, *, in_channels=192, upsample_initial_channel=512,
                 gin_channels=256, upsample_rates=(12, 10, 2, 2), upsample_kernel_sizes=(24, 20, 4, 4),
                 res_kernel_sizes=(3, 7, 11), res_dilations=(1, 3, 5), weight_storage: str = "f32"):
        if not torch.cuda.is_available():
            raise NativeError("rvc_amd.Decoder needs a HIP device (no CPU fallback)")
        cfg = DecoderConfig()
        cfg.kind = DEC_KINDS[vocoder]
        cfg.sample_rate = sr
        cfg.in_channels = in_channels
        cfg.upsample_initial_channel = upsample_initial_channel
        cfg.gin_channels = gin_channels
        cfg.n_ups = len(upsample_rates)
        for i, (u, k) in enumerate(zip(upsample_rates, upsample_kernel_sizes)):
            cfg.upsample_rates[i] = int(u)
            cfg.upsample_kernel_sizes[i] = int(k)
        cfg.n_res_kernels = len(res_kernel_sizes)
        for i, k in enumerate(res_kernel_sizes):
            cfg.res_kernel_sizes[i] = int(k)
        cfg.weight_storage = {"f32": 0, "bf16": 1}[weight_storage]
        cfg.n_res_dilations = len(res_dilations)
        for i, d in enumerate(res_dilations):
            cfg.res_dilations[i] = int(d)
        self._h = c_void_p()
        self.vocoder = vocoder
        _check(_lib.rvc_decoder_create(ctypes.byref(cfg), ctypes.byref(self._h)), "rvc_decoder_create")
        for name, t in folded_weights.items():
            t = t.detach().float().cpu().contiguous()
            shape = (c_int64 * t.dim())(*t.shape)
            _check(_lib.rvc_decoder_set_tensor(self._h, name.encode(), t.data_ptr(), shape, t.dim()),
                   f"rvc_decoder_set_tensor({name})")
        _check(_lib.rvc_decoder_finalize(self._h), "rvc_decoder_finalize")
        self.upp = _lib.rvc_decoder_upp(self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _lib is not None:  # _lib is None during interpreter shutdown
            _lib.rvc_decoder_destroy(h)
            self._h = None

    def set_concurrency_hint(self, utterances_in_flight: int):
        _check(_lib.rvc_decoder_set_concurrency_hint(self._h, int(utterances_in_flight)), "rvc_decoder_set_concurrency_hint")

    def set_branch_parallel(self, side_streams: int):
        """ResBlock branches of a short stage on this many side streams of the handle (-1: one per branch after the first;
        0, the default: every launch on the caller's stream)."""
        _check(_lib.rvc_decoder_set_branch_parallel(self._h, int(side_streams)), "rvc_decoder_set_branch_parallel")

    def set_tap(self, stage: int, tap: torch.Tensor | None):
        _check(_lib.rvc_decoder_set_tap(self._h, stage, tap.data_ptr() if tap is not None else None),
               "rvc_decoder_set_tap")

    def forward(self, z: torch.Tensor, f0: torch.Tensor, g: torch.Tensor, *, src_randn: torch.Tensor,
                src_rand: torch.Tensor | None = None, adain_randn: torch.Tensor | None = None) -> torch.Tensor:
        z, f0, g = _dev_f32(z, "z"), _dev_f32(f0, "f0"), _dev_f32(g, "g")
        b, _, t = z.shape
        noise = DecoderNoise()
        src_randn = _dev_f32(src_randn, "src_randn")
        noise.src_randn_dev = src_randn.data_ptr()
        if src_rand is not None:
            src_rand = _dev_f32(src_rand, "src_rand")
            noise.src_rand_dev = src_rand.data_ptr()
        if adain_randn is not None:
            adain_randn = _dev_f32(adain_randn, "adain_randn")
            noise.adain_randn_dev = adain_randn.data_ptr()
        out = torch.empty((b, 1, t * self.upp), dtype=torch.float32, device=z.device)
        need = c_size_t()
        _check(_lib.rvc_decoder_workspace_bytes(self._h, b, t, ctypes.byref(need)), "rvc_decoder_workspace_bytes")
        ws = _ws.get("decoder", need.value, z.device)
        _check(_lib.rvc_decoder_forward(self._h, z.data_ptr(), f0.data_ptr(), g.data_ptr(), ctypes.byref(noise), b, t,
                                        out.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "rvc_decoder_forward")
        return out
