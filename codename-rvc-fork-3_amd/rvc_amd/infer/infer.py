"""``VoiceConverter`` -- orchestration around the pipeline with the reference's surface
(rvc/infer/infer.py:41-493).  File decoding/resampling, noise reduction and the pedalboard effects are out of
scope: ``convert_audio`` reads WAV files of any rate / channel count / PCM or float encoding
(rvc_amd.lib.audio: own RIFF parser, device-side polyphase resampler to 16 kHz in place of soxr) and writes 16-bit WAV, or
use ``convert_array`` with a NumPy signal.
"""
from __future__ import annotations

import os
import time
import traceback
import wave

import numpy as np
import torch

from rvc_amd.configs.config import Config
from rvc_amd.infer.pipeline import Pipeline as VC
from rvc_amd.lib.audio import load_audio_infer
from rvc_amd.lib.algorithm.synthesizers import Synthesizer
from rvc_amd.lib.hubert import HubertModelWithFinalProj
from rvc_amd.lib.tools.split_audio import merge_audio, process_audio


def _write_wav(path, audio, sr):
    pcm = np.clip(audio, -1.0, 1.0)
    pcm = (pcm * 32767.0).astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(int(sr))
        f.writeframes(pcm.tobytes())


class VoiceConverter:
    def __init__(self, device: str | None = None):
        self.config = Config(device)
        self.hubert_model = None
        self.last_embedder_model = None
        self.tgt_sr = None
        self.net_g = None
        self.vc = None
        self.cpt = None
        self.version = None
        self.n_spk = None
        self.use_f0 = None
        self.loaded_model = None
        self.dec_weight_dtype = "f32"   # "bf16": the vocoder's conv weights are stored as bf16 in HBM (BASELINE cfg 4)
        self.branch_streams = 0         # Decoder.set_branch_parallel for convert_batch: 0 = every vocoder launch on the utterance's stream, -1 = one side stream per ResBlock branch

    # ---- embedder (infer.py:64-74; file layout rvc/lib/utils.py:96-146) ----
    def load_hubert(self, embedder_model: str, embedder_model_custom: str = None):
        root = os.path.join(os.getcwd(), "rvc", "models", "embedders")
        path = embedder_model_custom if embedder_model == "custom" and embedder_model_custom else \
            os.path.join(root, {"chinese-hubert-base": "chinese_hubert_base", "japanese-hubert-base": "japanese_hubert_base",
                                "korean-hubert-base": "korean_hubert_base"}.get(embedder_model, embedder_model))
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        self.load_hubert_state_dict(sd)

    def load_hubert_state_dict(self, sd):
        self.hubert_model = HubertModelWithFinalProj(sd, device=self.config.device).float().eval()

    # ---- model (infer.py:416-493) ----
    def get_vc(self, weight_root, sid):
        if not self.loaded_model or self.loaded_model != weight_root:
            self.load_model(weight_root)
            if self.cpt is not None:
                self.setup_network()
                self.setup_vc_instance()
            self.loaded_model = weight_root

    def load_model(self, weight_root):
        self.cpt = torch.load(weight_root, map_location="cpu", weights_only=True) if os.path.isfile(weight_root) else None

    def load_checkpoint_dict(self, cpt: dict):
        """Same as get_vc() for an in-memory checkpoint dict (extract_model.py:56-107 format)."""
        self.cpt = dict(cpt)
        self.cpt["config"] = list(cpt["config"])
        self.setup_network()
        self.setup_vc_instance()
        self.loaded_model = cpt.get("model_name", "<dict>")

    def setup_network(self):
        if self.cpt is not None:
            self.tgt_sr = self.cpt["config"][-1]
            self.cpt["config"][-3] = self.cpt["weight"]["emb_g.weight"].shape[0]
            self.use_f0 = self.cpt.get("f0", 1)
            self.version = self.cpt.get("version", "v1")
            self.text_enc_hidden_dim = 768 if self.version == "v2" else 256
            self.vocoder = self.cpt.get("vocoder", "HiFi-GAN")
            self.net_g = Synthesizer(*self.cpt["config"], use_f0=self.use_f0,
                                     text_enc_hidden_dim=self.text_enc_hidden_dim, vocoder=self.vocoder)
            del self.net_g.enc_q
            self.net_g.dec_weight_dtype = self.dec_weight_dtype
            self.net_g.load_state_dict(self.cpt["weight"], strict=False)
            self.net_g = self.net_g.to(self.config.device).float()
            self.net_g.eval()

    def setup_vc_instance(self):
        if self.cpt is not None:
            self.vc = VC(self.tgt_sr, self.config)
            self.n_spk = self.cpt["config"][-3]

    # ---- conversion ----
    def convert_array(self, audio: np.ndarray, *, index_path: str = "", pitch: int = 0, f0_file=None,
                      f0_method: str = "rmvpe", index_rate: float = 0.75, volume_envelope: float = 1,
                      protect: float = 0.5, hop_length: int = 128, f0_autotune: bool = False,
                      f0_autotune_strength: float = 1, filter_radius: float = 3.0, sid: int = 0, noise_seed=None,
                      split_audio: bool = False):
        """The array-level core of convert_audio (infer.py:262-311): 16 kHz float array in, float32 @tgt_sr out."""
        if torch.is_tensor(audio):  # HBM-resident entry: same peak limiting on the device
            audio = audio.to(device=self.config.device, dtype=torch.float64)
            audio_max = audio.abs().max() / 0.95
            audio = torch.where(audio_max > 1, audio / audio_max, audio)
        else:
            audio = np.asarray(audio, dtype=np.float64).copy()
            audio_max = np.abs(audio).max() / 0.95  # infer.py:262-265
            if audio_max > 1:
                audio /= audio_max
        file_index = index_path.strip().strip('"').strip("\n").strip('"').strip().replace("trained", "added")

        def run(chunk):
            return self.vc.pipeline(model=self.hubert_model, net_g=self.net_g, sid=sid, audio=chunk, pitch=pitch,
                                    f0_method=f0_method, file_index=file_index, index_rate=index_rate,
                                    pitch_guidance=self.use_f0, filter_radius=filter_radius,
                                    volume_envelope=volume_envelope, version=self.version, protect=protect,
                                    hop_length=hop_length, f0_autotune=f0_autotune,
                                    f0_autotune_strength=f0_autotune_strength, f0_file=f0_file, noise_seed=noise_seed)

        if not split_audio:
            return run(audio)
        # infer.py:283-318: cut at silences (host, on the 16 kHz input), convert the chunks, put them back on the time line
        host = audio.cpu().numpy() if torch.is_tensor(audio) else audio
        chunks, intervals = process_audio(host, 16000)
        print(f"Audio split into {len(chunks)} chunks for processing.")
        converted = []
        for c in chunks:
            out = run(torch.from_numpy(np.ascontiguousarray(c)).to(audio.device) if torch.is_tensor(audio) else c)
            converted.append(out.cpu().numpy() if torch.is_tensor(out) else out)
            print(f"Converted audio chunk {len(converted)}")
        merged = merge_audio(chunks, converted, intervals, 16000, self.tgt_sr)
        return torch.from_numpy(merged).to(audio.device) if torch.is_tensor(audio) else merged

    def convert_batch(self, audios, *, inflight: int = 3, **kwargs):
        """Convert a list of 16 kHz utterances with ``inflight`` of them on the GPU at a time, each on its own HIP stream
        (one host thread per stream; the HIP / PyTorch calls release the GIL).  Utterances share no state,
        and one utterance alone leaves the chip partly idle -- the 8-CU BiGRU, the 1.17-round first vocoder stage, ~1500
        small launches -- so interleaved utterances finish sooner than in sequence: 30.4 / 26.0 / 25.1-25.7 ms per 30 s
        utterance at 1 / 2 / 3 in flight (round 6, profiles/r06_inflight_sweep.txt; each one in flight keeps ~2.5 GB of workspaces).  Results keep the input
        order; device tensors in give device tensors out (valid once this returns)."""
        import threading
        audios = list(audios)
        results = [None] * len(audios)
        errors = []
        dev = self.config.device
        n_workers = max(1, min(int(inflight), len(audios)))
        if kwargs.get("noise_seed") is not None:
            n_workers = 1   # parity mode replays torch's GLOBAL CPU generator stream: one utterance at a time
        if not hasattr(self, "_batch_streams"):
            self._batch_streams = []
            self._batch_slots = {}
        while len(self._batch_streams) < n_workers:
            self._batch_streams.append(torch.cuda.Stream(device=dev))
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))   # inputs produced on the caller's stream
        lock, cursor = threading.Lock(), [0]

        def work(tid):
            # Host arrays (the reference's boundary, pipeline.py:509-528) cross PCIe through two page-locked slots per
            # stream, both ways asynchronously on the utterance's own stream: the worker enqueues utterance k + 1 while
            # utterance k is still running and only waits for a slot's event when it needs that slot again -- no
            # per-utterance stream synchronise, so the stream never drains between utterances.
            # (the page-locked buffers belong to the stream, not to the call: allocating them costs milliseconds)
            slots = self._batch_slots.setdefault(tid, [dict(inp=None, out=None, up=None, done=None, idx=None, n=0) for _ in range(2)])
            for slot in slots:
                slot["idx"] = None   # a previous call that raised mid-batch may have left a result pending: it is not this call's

            def finish(slot):
                if slot["idx"] is not None:
                    slot["done"].synchronize()
                    results[slot["idx"]] = slot["out"][:slot["n"]].numpy().copy()
                    slot["idx"] = None

            try:
                stream = self._batch_streams[tid]
                stream.wait_event(ready)
                k = 0
                with torch.cuda.stream(stream):
                    while True:
                        with lock:                 # shared queue: a stream takes the next utterance when it is free
                            i = cursor[0]
                            cursor[0] += 1
                        if i >= len(audios):
                            break
                        a = audios[i]
                        # (a host array longer than x_max is filtered on the host BEFORE its upload -- Pipeline.pipeline's long-input
                        # branch -- so it takes the plain host path: no device -> host copy of the input, one synchronise at its end)
                        long_host = not torch.is_tensor(a) and len(a) + self.vc.window > self.vc.t_max
                        if torch.is_tensor(a) or long_host or kwargs.get("split_audio") or kwargs.get("noise_seed") is not None:
                            results[i] = self.convert_array(a, **kwargs)
                            continue
                        slot = slots[k & 1]
                        k += 1
                        finish(slot)               # its previous utterance (two back) has long completed
                        a = np.ascontiguousarray(a, dtype=np.float64)
                        if slot["inp"] is None or slot["inp"].numel() < a.shape[0]:
                            slot["inp"] = torch.empty(max(a.shape[0], 1 << 16), dtype=torch.float64, pin_memory=True)
                            slot["up"] = torch.cuda.Event()
                        else:
                            slot["up"].synchronize()   # the upload that last read this buffer
                        slot["inp"][:a.shape[0]].numpy()[:] = a
                        a_dev = slot["inp"][:a.shape[0]].to(dev, non_blocking=True)
                        slot["up"].record(stream)
                        out = self.convert_array(a_dev, **kwargs).contiguous()
                        if slot["out"] is None or slot["out"].numel() < out.shape[0] or slot["out"].dtype != out.dtype:
                            slot["out"] = torch.empty(max(out.shape[0], 1 << 16), dtype=out.dtype, pin_memory=True)
                            slot["done"] = torch.cuda.Event()
                        slot["out"][:out.shape[0]].copy_(out, non_blocking=True)
                        slot["done"].record(stream)
                        slot["idx"], slot["n"] = i, out.shape[0]
                for slot in slots:
                    finish(slot)
            except Exception as error:  # surfaced after the join
                errors.append(error)

        self.net_g.dec.set_concurrency_hint(n_workers)   # per decoder handle: other converters in the process are unaffected
        # ResBlock branches of the vocoder's short stages on side streams: opt-in (self.branch_streams = -1).  It shortens ONE
        # forward at a time (-0.3 ms of 32.9 per 30 s utterance) and is level or worse with two in flight; off by default also because
        # side-by-side launches make a kernel's traced duration contain the time it shares the chip (profiles/r05_branch_streams.txt)
        self.net_g.dec.set_branch_parallel(self.branch_streams or 0)
        threads = [threading.Thread(target=work, args=(t,)) for t in range(n_workers)]
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            self.net_g.dec.set_concurrency_hint(0)
        for stream in self._batch_streams[:n_workers]:
            torch.cuda.current_stream(dev).wait_stream(stream)
        if errors:
            raise errors[0]
        return results

    def convert_audio(self, audio_input_path: str, audio_output_path: str, model_path: str, index_path: str,
                      pitch: int = 0, f0_file: str = None, f0_method: str = "rmvpe", index_rate: float = 0.75,
                      volume_envelope: float = 1, protect: float = 0.5, hop_length: int = 128,
                      split_audio: bool = False, f0_autotune: bool = False, f0_autotune_strength: float = 1,
                      filter_radius: float = 3.0, embedder_model: str = "contentvec",
                      embedder_model_custom: str = None, clean_audio: bool = False, clean_strength: float = 0.5,
                      export_format: str = "WAV", post_process: bool = False, resample_sr: int = 0, sid: int = 0,
                      **kwargs):
        """infer.py:193-348: same arguments, defaults and error convention (never raises; prints and returns None)."""
        if not model_path:
            print("No model path provided. Aborting conversion.")
            return
        self.get_vc(model_path, sid)
        try:
            start_time = time.time()
            print(f"Converting audio '{audio_input_path}'...")
            if clean_audio or post_process or export_format != "WAV":
                raise NotImplementedError("clean_audio / post_process / non-WAV export are outside the hot path "
                                          "(SURVEY §2 items 2, 9)")
            audio = load_audio_infer(audio_input_path, 16000, device=self.config.device)   # infer.py:257-260
            if not self.hubert_model or embedder_model != self.last_embedder_model:
                self.load_hubert(embedder_model, embedder_model_custom)
                self.last_embedder_model = embedder_model
            if self.tgt_sr != resample_sr >= 16000:  # infer.py:280-281
                self.tgt_sr = resample_sr
            audio_opt = self.convert_array(audio, index_path=index_path, pitch=pitch, f0_file=f0_file,
                                           f0_method=f0_method, index_rate=index_rate,
                                           volume_envelope=volume_envelope, protect=protect, hop_length=hop_length,
                                           f0_autotune=f0_autotune, f0_autotune_strength=f0_autotune_strength,
                                           filter_radius=filter_radius, sid=sid, split_audio=split_audio)
            _write_wav(audio_output_path, audio_opt, self.tgt_sr)
            print(f"Conversion completed at '{audio_output_path}' in {time.time() - start_time:.2f} seconds.")
        except Exception as error:
            print(f"An error occurred during audio conversion: {error}")
            print(traceback.format_exc())

    def convert_audio_batch(self, audio_input_paths: str, audio_output_path: str, **kwargs):
        """infer.py:350-414: every WAV of a folder, skipping existing outputs.  The reference loops sequentially; here the
        first file loads the models, the rest are converted ``inflight`` (default 3) at a time, each on its own HIP
        stream (see convert_batch)."""
        import threading
        inflight = max(1, int(kwargs.pop("inflight", 3)))
        try:
            start_time = time.time()
            print(f"Converting audio batch '{audio_input_paths}'...")
            audio_files = [f for f in sorted(os.listdir(audio_input_paths)) if f.lower().endswith("wav")]
            print(f"Detected {len(audio_files)} audio files for inference.")
            jobs = []
            for a in audio_files:
                new_input = os.path.join(audio_input_paths, a)
                new_output = os.path.join(audio_output_path, os.path.splitext(a)[0] + "_output.wav")
                if not os.path.exists(new_output):
                    jobs.append((new_input, new_output))
            if jobs:   # the first conversion loads the checkpoint, the embedder and the index cache
                self.convert_audio(audio_input_path=jobs[0][0], audio_output_path=jobs[0][1], **kwargs)
            rest = jobs[1:]
            n_workers = min(inflight, len(rest))
            dev = self.config.device

            def work(tid):
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    for src, dst in rest[tid::n_workers]:
                        self.convert_audio(audio_input_path=src, audio_output_path=dst, **kwargs)   # never raises

            threads = [threading.Thread(target=work, args=(t,)) for t in range(n_workers)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            print(f"Batch conversion completed in {time.time() - start_time:.2f} seconds.")
        except Exception as error:
            print(f"An error occurred during audio batch conversion: {error}")
            print(traceback.format_exc())
