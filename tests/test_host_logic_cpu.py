"""CPU tests of the product's host-side logic (no HIP calls): weight-norm folding, f0 post-processing, RMVPE decode,
the synthetic checkpoint formats, and the banded relative attention of the TextEncoder / the flow against the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import rvc_oracle as O
from rvc_amd.lib import synthetic as S
from rvc_amd.lib.algorithm.weights import fold_weight_norm


def test_fold_weight_norm_matches_oracle_and_torch():
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    a, b = fold_weight_norm(cpt["weight"]), O.fold_weight_norm(cpt["weight"])
    assert a.keys() == b.keys()
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not any(k.endswith(("weight_g", "weight_v")) for k in a)
    # the same formula torch's parametrization applies on every forward
    v, g = cpt["weight"]["dec.ups.0.weight_v"], cpt["weight"]["dec.ups.0.weight_g"]
    assert torch.equal(a["dec.ups.0.weight"], torch._weight_norm(v, g, 0))
    # fp16 exports are folded in fp32
    half = S.make_synth_checkpoint(40000, "HiFi-GAN", seed=1, half=True)
    f = fold_weight_norm(half["weight"])
    assert all(t.dtype == torch.float32 for t in f.values() if t.is_floating_point())
    # HuBERT's positional conv is normalised over dim 2 (g: [1,1,128])
    hs = S.make_hubert_state_dict(1)
    fw = fold_weight_norm(hs)["encoder.pos_conv_embed.conv.weight"]
    v, g = hs["encoder.pos_conv_embed.conv.parametrizations.weight.original1"], hs["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    assert torch.allclose(fw, torch._weight_norm(v, g, 2))


def test_checkpoint_format_matches_extract_model():
    for voc in ("HiFi-GAN", "MRF HiFi-GAN", "RefineGAN"):
        cpt = S.make_synth_checkpoint(48000, voc, seed=0)
        assert len(cpt["config"]) == 18 and cpt["config"][-1] == 48000 and cpt["f0"] == 1 and cpt["version"] == "v2"
        assert not any(k.startswith("enc_q") for k in cpt["weight"])
        assert cpt["weight"]["emb_g.weight"].shape == (109, 256)


class _Cfg:
    x_pad, x_query, x_center, x_max, device = 1, 6, 38, 41, "cpu"


def _pipeline_cpu():
    from rvc_amd.infer.pipeline import Pipeline
    p = Pipeline.__new__(Pipeline)  # constants only; no device objects
    p.x_pad, p.sample_rate, p.window = 1, 16000, 160
    p.f0_mel_min = 1127 * np.log(1 + 50 / 700)
    p.f0_mel_max = 1127 * np.log(1 + 1100 / 700)
    return p


def test_f0_postprocess_bit_exact_vs_reference_golden():
    g = load_golden("f0_coarse")
    p = _pipeline_cpu()
    for shift in (0, 5, -7):
        coarse, f0bak = p._postprocess_f0(g["f0"].copy(), shift)
        assert np.array_equal(coarse.astype(np.int64), g[f"coarse_{shift}"])
        assert np.array_equal(f0bak, g[f"f0bak_{shift}"])


def test_f0_postprocess_device_table_equals_host_formula():
    """The HBM-resident quantiser (threshold table + searchsorted) against the reference's NumPy expression: on the
    golden contours, on random contours, and on both sides of every one of the 254 step positions."""
    import torch
    g = load_golden("f0_coarse")
    p = _pipeline_cpu()
    thr = p._coarse_thresholds()
    p.device, p._coarse_thr = "cpu", torch.from_numpy(thr)   # Pipeline.__init__ builds the table once, at construction
    assert thr.shape == (254,) and np.all(np.diff(thr) > 0)
    rng = np.random.default_rng(0)
    edge = np.concatenate([thr, np.nextafter(thr, -np.inf), np.nextafter(thr, np.inf)])
    cases = [g["f0"], np.concatenate([[0.0, 1e-300, 49.9, 1100.0, 5000.0], rng.uniform(0, 1500, 20000), edge])]
    for f0 in cases:
        for shift in (0, 5, -7):
            want_c, want_f = p._postprocess_f0(f0.copy(), shift)
            got_c, got_f = p._postprocess_f0_device(torch.from_numpy(f0.copy()), shift)
            assert np.array_equal(got_c.numpy(), want_c.astype(np.int64))
            assert np.array_equal(got_f.numpy(), want_f)


def test_autotune_and_change_rms_match_reference_golden():
    """The optional branches of SURVEY §8 a19 in the product's host mirror (pipeline.py:38-114, 385-386, 682-685)."""
    import torch
    from rvc_amd.infer.pipeline import AudioProcessor, Autotune, REF_FREQS
    g = load_golden("autotune")
    assert np.array_equal(np.array(REF_FREQS), g["ref_freqs"])
    p = _pipeline_cpu()
    p.note_dict = Autotune(REF_FREQS).note_dict
    for strength in (1.0, 0.4):
        assert np.array_equal(Autotune.autotune_f0(p, g["f0"].copy(), strength), g[f"tuned_{strength}"])
    coarse, f0bak = p._postprocess_f0(g["f0"].copy(), 3, True, None, 0.4)
    want = p._postprocess_f0(g["tuned_0.4"].copy(), 3)
    assert np.array_equal(coarse, want[0]) and np.array_equal(f0bak, want[1])
    c = load_golden("change_rms")
    for rate in (0.25, 0.0):
        out = AudioProcessor.change_rms(torch.from_numpy(c["source"]), 16000, torch.from_numpy(c["target"]), 16000, rate)
        ref = c[f"out_{rate}"]
        assert out.dtype == torch.float32
        # the frame means are summed in torch's order, not NumPy's pairwise order: a few float32 ulps on the envelope
        assert np.abs(out.numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max())
        assert np.array_equal(AudioProcessor.change_rms(c["source"], 16000, c["target"], 16000, rate), out.numpy())


def test_split_and_merge_match_reference_golden():
    """rvc/lib/tools/split_audio.py:5-79.  merge_audio is pinned bit-exactly by the reference's own output; the silence
    detector restates librosa.effects.split (absent here, `parity unpinned`) and is checked against hand-derived edges."""
    from rvc_amd.lib.tools.split_audio import merge_audio, process_audio
    g = load_golden("split_merge")
    chunks, intervals = process_audio(g["signal"], 16000)
    assert np.array_equal(np.asarray(intervals), g["intervals"])
    # frames of 4000 every 2000, centred: the first frame holding signal (>= sample 5000) is centred at 4000, the last
    # one before the pause (< 15000) at 16000 -> [4000, 18000); then [22000, 46000)
    assert np.asarray(intervals).tolist() == [[4000, 18000], [22000, 46000]]
    conv = [np.full(int(n), 0.01 * (i + 1), dtype=np.float32) for i, n in enumerate(g["conv_lengths"])]
    merged = merge_audio(chunks, conv, intervals, 16000, 48000)
    assert merged.dtype == np.float32 and np.array_equal(merged, g["merged"])
    # levels are relative to the loudest frame (librosa's ref=np.max): an all-zero input is one single interval
    assert np.asarray(process_audio(np.zeros(16000), 16000)[1]).tolist() == [[0, 16000]]


def test_reflect_pad_longer_than_signal_equals_numpy():
    import torch
    from rvc_amd.infer.pipeline import _reflect_pad
    for n, pad in ((11000, 16000), (5, 16), (50, 20), (2, 7)):
        x = np.arange(n, dtype=np.float64) ** 1.5
        assert np.array_equal(_reflect_pad(torch.from_numpy(x), pad).numpy(), np.pad(x, (pad, pad), mode="reflect"))


def test_f0_file_override():
    p = _pipeline_cpu()
    f0 = np.full(400, 100.0)
    inp = np.array([[0.0, 200.0], [1.0, 300.0]], dtype="float32")  # seconds, Hz (pipeline.py:390-400)
    coarse, f0bak = p._postprocess_f0(f0.copy(), 0, False, inp)
    assert f0bak[99] == 100.0 and f0bak[100] == 200.0 and abs(f0bak[150] - 250.0) < 1e-9 and f0bak[201] == 100.0


def test_rmvpe_decode_matches_reference_golden():
    from rvc_amd.lib.predictors.RMVPE import RMVPE0Predictor
    g = load_golden("rmvpe")
    f0 = RMVPE0Predictor(device="cpu").decode(torch.from_numpy(g["hidden"])).numpy()
    assert np.allclose(f0, g["f0"], rtol=1e-14, atol=0)
    assert np.array_equal(O.f0_to_coarse(f0.copy())[0], O.f0_to_coarse(g["f0"].copy())[0])


def test_text_encoder_and_flow_match_reference_golden(ref_inputs):
    """The product's TextEncoder (banded relative attention) and flow, run on CPU tensors, against the reference."""
    from rvc_amd.lib.algorithm.encoders import text_encoder
    from rvc_amd.lib.algorithm.residuals import flow_reverse
    g = load_golden("synth_nsf48")
    feats, f0c, f0f = ref_inputs
    T = int(g["T"])
    w = fold_weight_norm(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)["weight"])
    phone = torch.from_numpy(np.repeat(feats, 2, axis=0)[:T]).unsqueeze(0)
    pitch = torch.from_numpy(f0c[:T].astype(np.int64)).unsqueeze(0)
    with torch.no_grad():
        m_p, logs_p, x_mask = text_encoder(w, phone, pitch, torch.tensor([T]))
        assert np.abs(m_p[0].numpy() - g["m_p"]).max() <= 1e-4 and np.abs(logs_p[0].numpy() - g["logs_p"]).max() <= 1e-4
        gvec = w["emb_g.weight"][int(g["sid"])].view(1, 256, 1)
        z = flow_reverse(w, torch.from_numpy(g["z_p"]).unsqueeze(0), x_mask, gvec)
    assert np.abs(z[0].numpy() - g["z"]).max() <= 1e-4
    # padded batch item: masked frames do not leak into valid ones
    with torch.no_grad():
        phone2 = torch.cat([phone, torch.randn(1, 16, 768)], 1)
        pitch2 = torch.cat([pitch, torch.ones(1, 16, dtype=torch.long)], 1)
        m2, _, mask2 = text_encoder(w, phone2, pitch2, torch.tensor([T]))
    assert mask2[0, 0, T:].sum() == 0 and np.abs(m2[0, :, :T].numpy() - g["m_p"]).max() <= 1e-4


def test_hubert_cpu_matches_transformers_golden():
    from rvc_amd.lib.hubert import HubertModelWithFinalProj
    g = load_golden("hubert")
    m = HubertModelWithFinalProj(S.make_hubert_state_dict(1), device="cpu")
    feats = m(torch.from_numpy(g["wav"]))["last_hidden_state"].numpy()
    assert np.abs(feats - g["feats"]).max() <= 2e-4
    assert m.final_proj(torch.from_numpy(feats[0])).shape == (49, 256)


def test_reference_surface_signatures():
    """Argument names and defaults of the drop-in classes equal the reference's (infer.py:193-219, pipeline.py:509-528,
    synthesizers.py:223-231)."""
    import inspect
    from rvc_amd.infer.infer import VoiceConverter
    from rvc_amd.infer.pipeline import Pipeline
    from rvc_amd.lib.algorithm.synthesizers import Synthesizer
    sig = inspect.signature(VoiceConverter.convert_audio)
    want = dict(pitch=0, f0_file=None, f0_method="rmvpe", index_rate=0.75, volume_envelope=1, protect=0.5, hop_length=128,
                split_audio=False, f0_autotune=False, f0_autotune_strength=1, filter_radius=3.0, embedder_model="contentvec",
                embedder_model_custom=None, clean_audio=False, clean_strength=0.5, export_format="WAV", post_process=False,
                resample_sr=0, sid=0)
    names = list(sig.parameters)
    assert names[:5] == ["self", "audio_input_path", "audio_output_path", "model_path", "index_path"]
    for k, v in want.items():
        assert sig.parameters[k].default == v, k
    pnames = list(inspect.signature(Pipeline.pipeline).parameters)
    assert pnames[:18] == ["self", "model", "net_g", "sid", "audio", "pitch", "f0_method", "file_index", "index_rate",
                           "pitch_guidance", "filter_radius", "volume_envelope", "version", "protect", "hop_length",
                           "f0_autotune", "f0_autotune_strength", "f0_file"]
    inames = list(inspect.signature(Synthesizer.infer).parameters)
    assert inames[:7] == ["self", "phone", "phone_lengths", "pitch", "nsff0", "sid", "rate"]
    vnames = list(inspect.signature(Pipeline.voice_conversion).parameters)
    assert vnames[:12] == ["self", "model", "net_g", "sid", "audio0", "pitch", "pitchf", "index", "big_npy", "index_rate",
                           "version", "protect"]


def test_convert_audio_never_raises(capsys, tmp_path):
    """infer.py:346-348: errors are printed, not raised; missing model path aborts quietly."""
    from rvc_amd.infer.infer import VoiceConverter
    vc = VoiceConverter.__new__(VoiceConverter)
    vc.loaded_model, vc.cpt = None, None
    assert vc.convert_audio("in.wav", "out.wav", "", "") is None
    assert "No model path provided" in capsys.readouterr().out
