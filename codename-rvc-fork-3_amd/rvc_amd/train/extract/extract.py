"""Feature extraction for training data on the MI355X with the reference's surface (rvc/train/extract/extract.py:29-212):
``FeatureInput`` (RMVPE f0 + coarse bins per utterance) and ``process_file_embedding`` (HuBERT features per utterance),
fanned out over the GPUs of the node exactly like the reference does -- one worker per device, files strided
``files[i::len(devices)]`` (extract.py:141-152, 196-208).  The networks and kernels are the inference path's own
(rvc_amd.lib.predictors.RMVPE, rvc_amd.lib.hubert, librvc_amd K4/K5/K7): SURVEY §8f rank 4.

Out of scope, as for inference: crepe / fcpe estimators, audio decoding other than WAV (rvc_amd.lib.audio).
"""
from __future__ import annotations

import concurrent.futures
import os
import time

import numpy as np
import torch

from rvc_amd.lib.audio import load_audio


class FeatureInput:
    def __init__(self, sample_rate=16000, hop_size=160, device="cuda:0"):
        self.fs = sample_rate
        self.hop = hop_size
        self.f0_bin = 256
        self.f0_max = 1100.0
        self.f0_min = 50.0
        self.f0_mel_min = 1127 * np.log(1 + self.f0_min / 700)
        self.f0_mel_max = 1127 * np.log(1 + self.f0_max / 700)
        self.device = device
        self.model_rmvpe = None

    def compute_f0(self, audio_array, method, hop_length):
        if method != "rmvpe":
            raise NotImplementedError(f"f0 method {method!r}: only 'rmvpe' is built (SURVEY §2 item 10)")
        return self.model_rmvpe.infer_from_audio(audio_array, thred=0.03)

    def coarse_f0(self, f0):
        """extract.py:76-87"""
        f0_mel = 1127.0 * np.log(1.0 + f0 / 700.0)
        f0_mel = np.clip((f0_mel - self.f0_mel_min) * (self.f0_bin - 2) / (self.f0_mel_max - self.f0_mel_min) + 1, 1,
                         self.f0_bin - 1)
        return np.rint(f0_mel).astype(int)

    def process_file(self, file_info, f0_method, hop_length):
        inp_path, opt_path_coarse, opt_path_full, _ = file_info
        if os.path.exists(opt_path_coarse) and os.path.exists(opt_path_full):
            return
        try:
            np_arr = load_audio(inp_path, self.fs, device=self.device)   # resampler tensors on THIS worker's GPU
            feature_pit = self.compute_f0(np_arr, f0_method, hop_length)
            np.save(opt_path_full, feature_pit, allow_pickle=False)
            np.save(opt_path_coarse, self.coarse_f0(feature_pit), allow_pickle=False)
        except Exception as error:
            print(f"An error occurred extracting file {inp_path} on {self.device}: {error}")

    def process_files(self, files, f0_method, hop_length, device, threads=1, rmvpe_state_dict=None):
        """One device's share.  The reference runs `threads` host threads per device (extract.py:124-131); the GPU path is
        a stream of kernels per file, so files are simply walked in order on this device's stream."""
        from rvc_amd.lib.predictors.RMVPE import RMVPE0Predictor
        self.device = device
        if f0_method == "rmvpe":
            path = os.path.join("rvc", "models", "predictors", "rmvpe.pt")
            self.model_rmvpe = RMVPE0Predictor(path if rmvpe_state_dict is None else None, device=device, state_dict=rmvpe_state_dict)
        with torch.cuda.device(device):
            for f in files:
                self.process_file(f, f0_method, hop_length)


def _stride(files, devices):
    """extract.py:145, 198: device i takes files[i::len(devices)]"""
    return [files[i::len(devices)] for i in range(len(devices))]


def run_pitch_extraction(files, devices, f0_method, hop_length, threads=1, rmvpe_state_dict=None):
    print(f"Starting pitch extraction on {', '.join(devices)} using {f0_method}...")
    start_time = time.time()
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(devices)) as executor:   # one host thread per GPU
        tasks = [executor.submit(FeatureInput().process_files, share, f0_method, hop_length, dev, max(1, threads // len(devices)),
                                 rmvpe_state_dict) for share, dev in zip(_stride(files, devices), devices)]
        for t in tasks:
            t.result()
    print(f"Pitch extraction completed in {time.time() - start_time:.2f} seconds.")


def process_file_embedding(files, embedder_model, embedder_model_custom, device_num, device, n_threads=1, hubert_state_dict=None):
    """extract.py:155-181: HuBERT last_hidden_state of every file -> .npy; NaN outputs are skipped with a message."""
    from rvc_amd.lib.hubert import HubertModelWithFinalProj
    if hubert_state_dict is None:
        root = os.path.join(os.getcwd(), "rvc", "models", "embedders")
        path = embedder_model_custom if embedder_model == "custom" and embedder_model_custom else os.path.join(root, embedder_model)
        hubert_state_dict = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    model = HubertModelWithFinalProj(hubert_state_dict, device=device).float().eval()
    with torch.cuda.device(device):
        for wav_file_path, _, _, out_file_path in files:
            if os.path.exists(out_file_path):
                continue
            feats = torch.from_numpy(load_audio(wav_file_path, 16000, device=device)).to(device).float().view(1, -1)
            with torch.no_grad():
                result = model(feats)["last_hidden_state"]
            feats_out = result.squeeze(0).float().cpu().numpy()
            if not np.isnan(feats_out).any():
                np.save(out_file_path, feats_out, allow_pickle=False)
            else:
                print(f"{wav_file_path} produced NaN values; skipping.")


def run_embedding_extraction(files, devices, embedder_model, embedder_model_custom, threads=1, hubert_state_dict=None):
    print(f"Starting embedding extraction on {', '.join(devices)}...")
    start_time = time.time()
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(devices)) as executor:
        tasks = [executor.submit(process_file_embedding, share, embedder_model, embedder_model_custom, i, dev,
                                 max(1, threads // len(devices)), hubert_state_dict)
                 for i, (share, dev) in enumerate(zip(_stride(files, devices), devices))]
        for t in tasks:
            t.result()
    print(f"Embedding extraction completed in {time.time() - start_time:.2f} seconds.")
