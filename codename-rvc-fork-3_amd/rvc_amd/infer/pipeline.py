"""``Pipeline`` -- the per-utterance voice-conversion pipeline with the reference's method surface
(rvc/infer/pipeline.py:117-694), rebuilt so that everything heavy stays on the MI355X:

* HuBERT, TextEncoder, flow, RMVPE network: PyTorch-ROCm (fp32)
* feature retrieval (``_retrieve_speaker_embeddings``), log-mel, vocoder: librvc_amd HIP kernels
* float64 host math the reference pins with integers (filtfilt, split points, f0 quantisation) is kept
  verbatim in NumPy/SciPy so those integers are bit-exact.

Signatures and defaults follow the reference; additions are keyword-only and optional:
``noise_seed`` (parity mode: reproduce the reference's CPU RNG stream) and ``set_index`` /
``.npy`` index files (faiss is not a dependency here).
"""
from __future__ import annotations

import os
import threading

import numpy as np
import torch
import torch.nn.functional as F
from scipy import signal

from rvc_amd import _native
from rvc_amd.lib.predictors.RMVPE import RMVPE0Predictor

FILTER_ORDER = 5
CUTOFF_FREQUENCY = 48  # Hz
SAMPLE_RATE = 16000  # Hz
bh, ah = signal.butter(N=FILTER_ORDER, Wn=CUTOFF_FREQUENCY, btype="high", fs=SAMPLE_RATE)  # pipeline.py:23-28


# pipeline.py:149-204: the note table of Autotune (G1 .. C6, two decimals)
REF_FREQS = [49.00, 51.91, 55.00, 58.27, 61.74, 65.41, 69.30, 73.42, 77.78, 82.41, 87.31, 92.50, 98.00, 103.83, 110.00,
             116.54, 123.47, 130.81, 138.59, 146.83, 155.56, 164.81, 174.61, 185.00, 196.00, 207.65, 220.00, 233.08,
             246.94, 261.63, 277.18, 293.66, 311.13, 329.63, 349.23, 369.99, 392.00, 415.30, 440.00, 466.16, 493.88,
             523.25, 554.37, 587.33, 622.25, 659.25, 698.46, 739.99, 783.99, 830.61, 880.00, 932.33, 987.77, 1046.50]


class Autotune:
    """pipeline.py:88-114."""

    def __init__(self, ref_freqs):
        self.ref_freqs = ref_freqs
        self.note_dict = self.ref_freqs

    def autotune_f0(self, f0, f0_autotune_strength):
        """Every frame (unvoiced zeros included, as in the reference) moves towards its nearest note; argmin keeps the
        first of two equidistant notes, like the reference's min(..., key=...)."""
        notes = np.asarray(self.note_dict, dtype=np.float64)
        f0 = np.asarray(f0, dtype=np.float64)
        closest = notes[np.argmin(np.abs(notes[None, :] - f0[:, None]), axis=1)]
        return f0 + (closest - f0) * f0_autotune_strength


class AudioProcessor:
    """pipeline.py:33-85, on the device."""

    @staticmethod
    def _rms(y: torch.Tensor, frame_length: int, hop_length: int) -> torch.Tensor:
        # librosa.feature.rms (0.11): zero centre-padding, frames of frame_length every hop_length, sqrt(mean(x^2))
        y = F.pad(y, (frame_length // 2, frame_length // 2))
        return y.unfold(0, frame_length, hop_length).pow(2).mean(-1).sqrt()

    @staticmethod
    def change_rms(source_audio, source_rate: int, target_audio, target_rate: int, rate: float):
        """Tensors in (source float64 as the pipeline holds it, target float32) -> float32 tensor; NumPy in -> NumPy out."""
        as_numpy = not torch.is_tensor(target_audio)
        if as_numpy:
            target_audio = torch.from_numpy(np.ascontiguousarray(target_audio))
        if not torch.is_tensor(source_audio):
            source_audio = torch.from_numpy(np.ascontiguousarray(source_audio)).to(target_audio.device)
        n = target_audio.shape[0]
        rms1 = AudioProcessor._rms(source_audio, source_rate // 2 * 2, source_rate // 2).float()
        rms2 = AudioProcessor._rms(target_audio, target_rate // 2 * 2, target_rate // 2).float()
        rms1 = F.interpolate(rms1.view(1, 1, -1), size=n, mode="linear").view(-1)
        rms2 = F.interpolate(rms2.view(1, 1, -1), size=n, mode="linear").view(-1)
        rms2 = torch.maximum(rms2, torch.zeros_like(rms2) + 1e-6)
        out = target_audio * (torch.pow(rms1, 1 - rate) * torch.pow(rms2, rate - 1))
        return out.numpy() if as_numpy else out


def _reflect_pad(x: torch.Tensor, pad: int) -> torch.Tensor:
    """np.pad(x, (pad, pad), mode="reflect") for a 1-D device tensor, including pads longer than the signal (NumPy keeps
    reflecting; torch's F.pad refuses): the even periodic extension with period 2 (n - 1)."""
    n = x.shape[0]
    if pad < n:
        return F.pad(x.view(1, 1, -1), (pad, pad), mode="reflect").view(-1)
    period = 2 * (n - 1)
    m = torch.remainder(torch.arange(-pad, n + pad, device=x.device), period)
    return x[torch.where(m < n, m, period - m)]


class _PinnedIO:
    """Page-locked staging buffers for the host boundary of ``Pipeline.pipeline`` (NumPy in, NumPy out, pipeline.py:509-528),
    one pair per HIP stream (utterances in flight on different streams must not share them), grow-only.  The upload is an
    asynchronous copy on the utterance's own stream; the download is one copy + one stream synchronise."""

    def __init__(self):
        self._buf = {}
        self._lock = threading.Lock()

    def _get(self, kind, n, dtype):
        key = (kind, torch.cuda.current_stream().cuda_stream)
        with self._lock:
            b = self._buf.get(key)
            if b is None or b.numel() < n or b.dtype != dtype:
                b = torch.empty(max(n, 1 << 16), dtype=dtype, pin_memory=True)
                self._buf[key] = b
        return b

    def upload(self, audio: np.ndarray, device) -> torch.Tensor:
        stage = self._get("in", audio.shape[0], torch.float64)[:audio.shape[0]]
        # the previous upload from this buffer ran on this same stream and has long completed (its utterance was downloaded)
        stage.numpy()[:] = audio
        return stage.to(device, non_blocking=True)

    def download(self, out: torch.Tensor) -> np.ndarray:
        stage = self._get("out", out.shape[0], out.dtype)[:out.shape[0]]
        stage.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return stage.numpy().copy()


_pinned = _PinnedIO()


class FeatureIndex:
    """Device-resident replacement of the faiss index + ``big_npy`` pair (pipeline.py:553-556).

    ``search_mode``:
      "exact"  (default) squared-L2 over every reconstructed vector -- what ``big_npy`` admits and what the parity tests pin;
      "ivf"    the reference's own approximate search: an ``IVF{n},Flat`` index probed with ``nprobe`` lists
               (extract_index.py:62-64): nearest centroids first, then only the members of those lists.  Needs the
               inverted-file structure of a ``.index`` file (``ivf=``).
    Truthy as a plain object, as the caller's ``if index:`` requires (pipeline.py:456-458)."""

    def __init__(self, big_npy, device, ivf=None, search_mode: str = "exact"):
        if isinstance(big_npy, np.ndarray):
            big_npy = torch.from_numpy(np.ascontiguousarray(big_npy, dtype=np.float32))
        self.vectors = big_npy.to(device=device, dtype=torch.float32).contiguous()
        self.aux = _native.knn_index_build(self.vectors)   # ||x||^2, fp16 copy, screening statistics
        self.ntotal = int(self.vectors.shape[0])
        self.search_mode = search_mode
        self.ivf = None
        if ivf is not None:
            cent = torch.from_numpy(np.array(ivf.centroids, dtype=np.float32)).to(self.vectors.device)
            if int(ivf.nprobe) > 8:
                # the centroid search returns 8 neighbours; silently probing fewer lists than the file asks for would be a
                # different (worse) search than the reference's.  extract_index.py:62-64 writes nprobe = 1.
                raise ValueError(f"IVF index asks for nprobe = {ivf.nprobe}; this build probes at most 8 lists (the reference writes 1)")
            self.ivf = {"centroids": cent, "aux": _native.knn_index_build(cent), "nprobe": max(1, int(ivf.nprobe)),
                        "lists": torch.from_numpy(np.array(ivf.padded_lists())).to(self.vectors.device)}
        elif search_mode == "ivf":
            raise ValueError("search_mode='ivf' needs the inverted lists of a faiss .index file")
        # built on the loading thread's stream; utterances on other streams (convert_batch) read it afterwards
        torch.cuda.current_stream(self.vectors.device).synchronize()

    def __bool__(self):
        return True

    def search_device(self, queries: torch.Tensor, k: int = 8):
        if self.search_mode == "ivf" and self.ivf is not None:
            _, near = _native.knn_search(self.ivf["centroids"], self.ivf["aux"], queries, k)       # nearest centroids
            probe = near[:, :self.ivf["nprobe"]]
            # fewer lists than nprobe: the missing probes come back as -1 and contribute NO candidates (gathering list 0 for
            # them would enter its rows twice and double their weight in the blend)
            cand = self.ivf["lists"][probe.clamp_min(0)].masked_fill((probe < 0).unsqueeze(-1), -1)
            cand = cand.reshape(queries.shape[0], -1)                                              # members of those lists
            return _native.knn_rank_candidates(self.vectors, self.aux, queries, cand, k)
        return _native.knn_search(self.vectors, self.aux, queries, k)

    def search(self, npy: np.ndarray, k: int = 8):
        """faiss-style host API: (D2 [Q,k] float32 ascending, I [Q,k] int64)."""
        d2, ids = self.search_device(torch.from_numpy(np.ascontiguousarray(npy, dtype=np.float32)).to(self.vectors.device), k)
        return d2.cpu().numpy(), ids.cpu().numpy()

    def reconstruct_n(self, start: int, n: int):
        return self.vectors[start:start + n].cpu().numpy()


def _load_index_file(path: str):
    """-> (big_npy, inverted-file structure or None).  ``.npy`` holds big_npy directly; anything else is parsed as the
    reference's faiss ``IVF{n},Flat`` file (pipeline.py:555-556) by rvc_amd.lib.faiss_index -- faiss itself is not needed.
    A file that cannot be parsed raises; the caller reports it the way pipeline.py:557-559 does."""
    if path.endswith(".npy"):
        return np.load(path), None
    from rvc_amd.lib.faiss_index import read_index
    ivf = read_index(path)
    return ivf.reconstruct_n(0, ivf.ntotal), ivf


class Pipeline:
    def __init__(self, tgt_sr, config):
        self.x_pad, self.x_query, self.x_center, self.x_max = config.x_pad, config.x_query, config.x_center, config.x_max
        self.sample_rate = 16000
        self.window = 160
        self.t_pad = self.sample_rate * self.x_pad
        self.t_pad_tgt = tgt_sr * self.x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sample_rate * self.x_query
        self.t_center = self.sample_rate * self.x_center
        self.t_max = self.sample_rate * self.x_max
        self.time_step = self.window / self.sample_rate * 1000
        self.f0_min, self.f0_max = 50, 1100
        self.f0_mel_min = 1127 * np.log(1 + self.f0_min / 700)
        self.f0_mel_max = 1127 * np.log(1 + self.f0_max / 700)
        self.device = config.device
        rmvpe_path = os.path.join("rvc", "models", "predictors", "rmvpe.pt")  # pipeline.py:207-210
        self.model_rmvpe = RMVPE0Predictor(rmvpe_path if os.path.isfile(rmvpe_path) else None, device=self.device)
        # shared by the host threads of VoiceConverter.convert_batch: filled under the lock, read-only afterwards
        self._lock = threading.Lock()
        self._index_cache = {}
        self._preset_index = None
        self._f0_streams = {}      # side stream per caller stream (several utterances may be in flight)
        self._coarse_thr = torch.from_numpy(self._coarse_thresholds()).to(self.device)
        self.index_search = "exact"   # "ivf": search faiss .index files the reference's way (nprobe lists); see FeatureIndex
        self.debug_taps = None     # tests: a dict here receives "f0_raw" and "salience" of the last call (device tensors)
        self.ref_freqs = REF_FREQS
        self.autotune = Autotune(self.ref_freqs)
        self.note_dict = self.autotune.note_dict

    # ---- additions -------------------------------------------------------------------------------------
    def load_rmvpe_state_dict(self, sd):
        self.model_rmvpe.load_state_dict(sd)

    def set_index(self, big_npy):
        """Install a feature index directly (N x 768 float32) instead of reading a faiss file per call."""
        self._preset_index = FeatureIndex(big_npy, self.device) if big_npy is not None else None

    def _get_index(self, file_index, index_rate):
        if index_rate <= 0:
            return None
        if file_index != "" and os.path.exists(file_index):
            key = (file_index, os.path.getmtime(file_index))
            with self._lock:   # one thread loads and uploads; the others wait and share the resident copy
                index = self._index_cache.get(key)
                if index is None:
                    try:
                        big_npy, ivf = _load_index_file(file_index)
                        index = FeatureIndex(big_npy, self.device, ivf=ivf,
                                             search_mode=self.index_search if ivf is not None else "exact")
                    except Exception as error:  # pipeline.py:557-559: warn and continue without retrieval
                        print(f"An error occurred reading the FAISS index: {error}")
                        return None
                    self._index_cache = {key: index}
            return index
        return self._preset_index

    # ---- F0 ------------------------------------------------------------------------------------------------
    def get_f0(self, input_audio_path, x, p_len, pitch, f0_method, filter_radius, hop_length, f0_autotune,
               f0_autotune_strength, inp_f0=None):
        """pipeline.py:322-410, rmvpe branch (the other estimators are out of scope)."""
        if f0_method != "rmvpe":
            raise NotImplementedError(f"f0_method={f0_method!r}: only 'rmvpe' is implemented")
        if torch.is_tensor(x):
            f0 = self.model_rmvpe.infer_from_audio_device(x, thred=0.03).cpu().numpy()
        else:
            f0 = self.model_rmvpe.infer_from_audio(x, thred=0.03)
        return self._postprocess_f0(f0, pitch, f0_autotune, inp_f0, f0_autotune_strength)

    def _postprocess_f0(self, f0, pitch, f0_autotune=False, inp_f0=None, f0_autotune_strength=1):
        """pipeline.py:385-410 on a float64 host contour: autotune, key shift, optional f0-file override, coarse bins."""
        if f0_autotune is True:
            f0 = Autotune.autotune_f0(self, f0, f0_autotune_strength)
        f0 *= pow(2, pitch / 12)
        tf0 = self.sample_rate // self.window
        if inp_f0 is not None:  # pipeline.py:390-400
            delta_t = np.round((inp_f0[:, 0].max() - inp_f0[:, 0].min()) * tf0 + 1).astype("int16")
            replace_f0 = np.interp(list(range(delta_t)), inp_f0[:, 0] * 100, inp_f0[:, 1])
            shape = f0[self.x_pad * tf0: self.x_pad * tf0 + len(replace_f0)].shape[0]
            f0[self.x_pad * tf0: self.x_pad * tf0 + len(replace_f0)] = replace_f0[:shape]
        f0bak = f0.copy()
        f0_mel = 1127 * np.log(1 + f0 / 700)
        f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - self.f0_mel_min) * 254 / (self.f0_mel_max - self.f0_mel_min) + 1
        f0_mel[f0_mel <= 1] = 1
        f0_mel[f0_mel > 255] = 255
        return np.rint(f0_mel).astype(int), f0bak

    def _coarse_thresholds(self) -> np.ndarray:
        """thr[b-2] = the smallest float64 f0 that `_postprocess_f0` maps to coarse bin >= b (b = 2..255).

        The quantiser (pipeline.py:402-408) is a monotone step function of f0, so its 254 step positions pin it
        completely: they are located once on the host by bisection over the float64 bit patterns with the very
        NumPy expression the reference evaluates, and the device then only compares (no device log involved)."""
        def coarse(f0):
            f0_mel = 1127 * np.log(1 + f0 / 700)
            f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - self.f0_mel_min) * 254 / (self.f0_mel_max - self.f0_mel_min) + 1
            f0_mel[f0_mel <= 1] = 1
            f0_mel[f0_mel > 255] = 255
            return np.rint(f0_mel).astype(np.int64)

        want = np.arange(2, 256, dtype=np.int64)
        lo = np.zeros(254, dtype=np.int64)                                  # bits of 0.0   -> bin 1
        hi = np.full(254, np.float64(1e6).view(np.int64), dtype=np.int64)   # bits of 1e6   -> bin 255
        while np.any(hi - lo > 1):
            mid = lo + (hi - lo) // 2
            ge = coarse(mid.view(np.float64).copy()) >= want
            hi, lo = np.where(ge, mid, hi), np.where(ge, lo, mid)
        return hi.view(np.float64)

    def _postprocess_f0_device(self, f0, pitch):
        """`_postprocess_f0` without leaving HBM (no f0 file, no autotune): same key shift (one float64 multiply) and
        the same coarse integers, read off the threshold table instead of re-evaluating log on the device."""
        f0 = f0 * pow(2, pitch / 12)
        coarse = torch.searchsorted(self._coarse_thr, f0, right=True) + 1
        return coarse, f0

    # ---- per-segment conversion -------------------------------------------------------------------------
    def _extract_features(self, model, audio0, index, big_npy, index_rate, version):
        """pipeline.py:445-465: HuBERT features, retrieval blend, x2 nearest upsampling (pitch-independent half)."""
        if torch.is_tensor(audio0):
            feats = audio0.to(self.device).float()
        else:
            feats = torch.from_numpy(np.ascontiguousarray(audio0)).float().to(self.device)
        feats = feats.mean(-1) if feats.dim() == 2 else feats
        assert feats.dim() == 1, feats.dim()
        n_audio = feats.shape[0]
        feats = model(feats.view(1, -1))["last_hidden_state"]
        feats = model.final_proj(feats[0]).unsqueeze(0) if version == "v1" else feats
        feats0 = feats
        if index:
            feats = self._retrieve_speaker_embeddings(feats, index, big_npy, index_rate)
        feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
        return feats, feats0, n_audio

    def _synthesize(self, net_g, sid, feats, feats0, n_audio, pitch, pitchf, protect, noise):
        """pipeline.py:466-490: length bookkeeping, protect blend, net_g.infer."""
        p_len = min(n_audio // self.window, feats.shape[1])
        pitch, pitchf = pitch[:, :p_len], pitchf[:, :p_len]
        if protect < 0.5:  # pipeline.py:474-481
            feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
            # pitchff[pitchf > 0] = 1; pitchff[pitchf < 1] = protect -- as selects (boolean-mask stores sync the host)
            pitchff = torch.where(pitchf < 1, torch.full_like(pitchf, protect),
                                  torch.where(pitchf > 0, torch.ones_like(pitchf), pitchf))
            feats = feats * pitchff.unsqueeze(-1) + feats0 * (1 - pitchff.unsqueeze(-1))
            feats = feats.to(feats0.dtype)
        p_len_t = torch.full((1,), p_len, device=self.device, dtype=torch.long)
        return net_g.infer(feats.float(), p_len_t, pitch, pitchf.float(), sid, noise=noise,
                           phone_lengths_host=[p_len])[0][0, 0]

    def voice_conversion(self, model, net_g, sid, audio0, pitch, pitchf, index, big_npy, index_rate, version, protect,
                         noise=None, as_tensor=False):
        """pipeline.py:412-495.  ``audio0``: NumPy or device tensor (1-D, 16 kHz)."""
        with torch.no_grad():
            if pitch is None or pitchf is None:
                raise NotImplementedError("models without pitch guidance are not supported (SURVEY §2 item 3b)")
            feats, feats0, n_audio = self._extract_features(model, audio0, index, big_npy, index_rate, version)
            audio1 = self._synthesize(net_g, sid, feats, feats0, n_audio, pitch, pitchf, protect, noise)
            if as_tensor:
                return audio1
            return audio1.data.cpu().float().numpy()

    def _retrieve_speaker_embeddings(self, feats, index, big_npy, index_rate):
        """pipeline.py:497-507 without the device->host->faiss->device round trip: exact L2 top-8 and the
        (1/d^2)^2 blend both run in HBM (librvc_amd rvc_knn_search / rvc_knn_blend)."""
        if not isinstance(index, FeatureIndex):  # a foreign index object with the faiss API: wrap its vectors once
            index = FeatureIndex(big_npy, self.device)
        q = feats[0].contiguous()
        if isinstance(getattr(self, "debug_taps", None), dict):   # tests / bench.py's kNN leg: the queries of this utterance
            self.debug_taps["knn_queries"] = q
        d2, ids = index.search_device(q, 8)
        return _native.knn_blend(index.vectors, q, d2, ids, float(index_rate)).unsqueeze(0)

    # ---- whole utterance ------------------------------------------------------------------------------------
    def pipeline(self, model, net_g, sid, audio, pitch, f0_method, file_index, index_rate, pitch_guidance,
                 filter_radius, volume_envelope, version, protect, hop_length, f0_autotune, f0_autotune_strength,
                 f0_file, *, noise_seed=None):
        """pipeline.py:509-694.  ``audio``: 1-D float NumPy @16 kHz -> float32 NumPy @tgt_sr (a device tensor in
        gives a device tensor out: the HBM-resident entry used by bench.py).

        noise_seed: None -> noise is drawn on the device; int -> parity mode: seed torch's CPU generator and
        draw every random tensor in the reference's order (including the 12 LayerDrop draws transformers'
        HuBERT makes per forward), so the result matches the reference CPU path run under the same seed."""
        if not pitch_guidance:
            raise NotImplementedError("models without pitch guidance are not supported (SURVEY §2 item 3b)")
        index = self._get_index(file_index, index_rate)
        big_npy = index.vectors if index is not None else None
        # high-pass, split points and padding stay float64 but run in HBM (the reference does them in NumPy/SciPy)
        as_tensor = torch.is_tensor(audio)
        n_in = int(audio.shape[0])
        long_input = n_in + self.window > self.t_max
        if long_input:
            # Long inputs are cut at arg-min positions of a 160-tap box sum of the filtered signal (below); this direct-form
            # high-pass amplifies 1-ulp differences to ~2e-8, enough to move such an arg-min, so the split integers are only
            # reproducible with SciPy's exact SEQUENTIAL recurrence -- 2 x n dependent float64 steps of three operations each, which
            # one GPU lane walks at ~75-96 ns a step (45-140 ms per 45 s clip; measured, csrc/filtfilt.hip) against ~9 ms for
            # SciPy's C loop: that one filter runs on the host.  The reference's boundary hands over a HOST array (pipeline.py:509),
            # so the filter runs on it BEFORE the upload: no device -> host copy, no stream synchronisation.  Only the HBM-resident
            # entry (a device tensor in: bench.py's `value`) pays a copy each way.
            host = audio.detach().to(dtype=torch.float64).cpu().numpy() if as_tensor else np.ascontiguousarray(audio, dtype=np.float64)
            filtered = np.ascontiguousarray(signal.filtfilt(bh, ah, host))
            with torch.cuda.device(self.device):
                audio = _pinned.upload(filtered, self.device)
        else:
            if as_tensor:
                audio = audio.to(device=self.device, dtype=torch.float64)
            else:
                with torch.cuda.device(self.device):
                    audio = _pinned.upload(np.ascontiguousarray(audio, dtype=np.float64), self.device)
            audio = _native.filtfilt_order5(audio, bh, ah)                  # pipeline.py:562, in HBM
        n_audio = audio.shape[0]
        opt_ts = []
        if long_input:                                                       # pipeline.py:563-577
            pad = F.pad(audio.view(1, 1, -1), (self.window // 2, self.window // 2), mode="reflect").view(-1)
            audio_sum = torch.zeros_like(audio)
            for i in range(self.window):                                     # same 160 sequential float64 adds
                audio_sum += pad[i: i + n_audio]
            # every split window at once, ONE device -> host read for all the integers: row r = |audio_sum| on
            # [t_r - t_query, t_r + t_query) (a window that runs past the end is shorter in the reference: masked with +inf here),
            # split = the FIRST position of the row's minimum (np.where(...)[0][0], pipeline.py:571-576)
            ts = torch.arange(self.t_center, n_audio, self.t_center, device=self.device)
            if ts.numel():
                pos = ts[:, None] - self.t_query + torch.arange(2 * self.t_query, device=self.device)[None, :]
                inside = pos < n_audio
                seg = torch.where(inside, audio_sum[pos.clamp(max=n_audio - 1)].abs(), torch.full((), float("inf"), dtype=audio_sum.dtype, device=self.device))
                is_min = seg == seg.min(dim=1, keepdim=True).values
                first = torch.where(is_min, torch.arange(2 * self.t_query, device=self.device)[None, :], 2 * self.t_query).min(dim=1).values
                opt_ts = (ts - self.t_query + first).tolist()
        s = 0
        audio_opt = []
        t = None
        audio_pad = _reflect_pad(audio, self.t_pad)  # pipeline.py:581
        p_len = audio_pad.shape[0] // self.window
        inp_f0 = None
        if hasattr(f0_file, "name"):  # pipeline.py:584-593
            try:
                with open(f0_file.name, "r") as f:
                    lines = f.read().strip("\n").split("\n")
                inp_f0 = np.array([[float(i) for i in line.split(",")] for line in lines], dtype="float32")
            except Exception as error:
                print(f"An error occurred reading the F0 file: {error}")
        sid = torch.full((1,), int(sid), device=self.device, dtype=torch.long)  # a fill, not a blocking H2D copy
        audio_dev = audio_pad.float()  # segments are views of it

        noise = None
        if noise_seed is not None:
            torch.manual_seed(int(noise_seed))
            noise = "cpu"

        # segment plan (pipeline.py:614-680): (audio slice, pitch slice) per segment
        plan = []
        for t in opt_ts:
            t = t // self.window * self.window
            plan.append((slice(s, t + self.t_pad2 + self.window), slice(s // self.window, (t + self.t_pad2) // self.window)))
            s = t
        plan.append((slice(t, None), slice(t // self.window if t is not None else None, None)))

        with torch.no_grad():
            # F0 (RMVPE) is independent of HuBERT + retrieval until the synthesizer.  Its U-Net is throughput-bound like
            # HuBERT, its BiGRU is latency-bound (3232 sequential steps on 8 CUs, ~5.5 ms): the U-Net goes FIRST on the
            # main stream, then the recurrence + decode + pitch quantisation run on a side stream underneath HuBERT
            # and the retrieval, so f0 is ready before the synthesizer needs it and no CU idles waiting for the GRU.
            main = torch.cuda.current_stream()
            with self._lock:
                side = self._f0_streams.get(main.cuda_stream)
                if side is None:
                    side = self._f0_streams[main.cuda_stream] = torch.cuda.Stream(device=self.device)
            if f0_method != "rmvpe":
                raise NotImplementedError(f"f0_method={f0_method!r}: only 'rmvpe' is implemented")
            gi, n_f0 = self.model_rmvpe.front_half_device(audio_dev)
            side.wait_stream(main)
            gi.record_stream(side)
            with torch.cuda.stream(side):
                f0_dev = self.model_rmvpe.back_half_device(gi, n_f0, thred=0.03, taps=self.debug_taps)
                if self.debug_taps is not None:
                    self.debug_taps["f0_raw"] = f0_dev
                if inp_f0 is None and f0_autotune is not True:
                    # the contour never leaves HBM and the host never waits for it: everything below is enqueued
                    # while the GPU is still busy with HuBERT
                    pitch, pitchf = self._postprocess_f0_device(f0_dev, pitch)
                    pitch = pitch[:p_len].unsqueeze(0)
                    pitchf = pitchf[:p_len].unsqueeze(0).float()
            feats_list = [self._extract_features(model, audio_dev[a], index, big_npy, index_rate, version) for a, _ in plan]
            if inp_f0 is None and f0_autotune is not True:
                main.wait_stream(side)
                pitch.record_stream(main)
                pitchf.record_stream(main)
            else:  # f0-file override: the reference's host code, on the host
                with torch.cuda.stream(side):
                    f0_host = f0_dev.cpu().numpy()  # waits for the side stream only
                pitch, pitchf = self._postprocess_f0(f0_host, pitch, f0_autotune, inp_f0, f0_autotune_strength)
                pitch, pitchf = pitch[:p_len], pitchf[:p_len]
                pitch = torch.tensor(pitch, device=self.device).unsqueeze(0).long()
                pitchf = torch.tensor(pitchf, device=self.device).unsqueeze(0).float()
            for (feats, feats0, n_audio), (_, ps) in zip(feats_list, plan):
                if noise_seed is not None:
                    for _ in range(12):  # transformers' HuBERT LayerDrop draws, made per segment before the synthesizer's
                        torch.rand([])
                seg = self._synthesize(net_g, sid, feats, feats0, n_audio, pitch[:, ps], pitchf[:, ps], protect, noise)
                audio_opt.append(seg[self.t_pad_tgt: -self.t_pad_tgt])
        out = torch.cat(audio_opt) if len(audio_opt) > 1 else audio_opt[0]
        if volume_envelope != 1:  # pipeline.py:682-685 (both rates are passed as 16000 there, too)
            out = AudioProcessor.change_rms(audio, self.sample_rate, out, self.sample_rate, volume_envelope)
        audio_max = out.abs().max() / 0.99  # pipeline.py:686-688
        out = torch.where(audio_max > 1, out / audio_max, out)
        if as_tensor:
            return out
        with torch.cuda.device(self.device):
            return _pinned.download(out.contiguous())
