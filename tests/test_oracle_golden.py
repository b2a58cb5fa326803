"""Pin the oracle (oracle/rvc_oracle.py) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by tests/golden/make_golden.py, which imports
/root/reference in the build container.  These tests run on CPU and never touch /root/reference.
Tolerances are written next to each check; integer quantities are bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rms
from oracle import rvc_oracle as O
from rvc_amd.lib import synthetic as S

SYNTH_CASES = [("nsf48", 48000, "HiFi-GAN"), ("nsf40", 40000, "HiFi-GAN"), ("nsf32", 32000, "HiFi-GAN"),
               ("mrf48", 48000, "MRF HiFi-GAN"), ("refine48", 48000, "RefineGAN")]


def test_butter_and_filtfilt():
    g = load_golden("filtfilt")
    assert np.array_equal(O.BH, g["bh"]) and np.array_equal(O.AH, g["ah"])
    assert np.allclose(O.BH[:2], [0.96996, -4.84980], atol=1e-5)  # SURVEY §8c measured values
    assert np.array_equal(O.highpass(g["x"]), g["y"])  # same scipy call -> bit-exact


def test_segmentation_integers():
    g = load_golden("segmentation")
    for secs in (10, 30, 41, 45, 100):
        audio = O.highpass(S.synth_audio(16000 * secs, seed=secs))
        plan = O.segment_plan(audio.shape[0], O.split_points(audio))
        got = []
        for (s0, s1, p0, p1) in plan:
            n = s1 - s0
            n_pitch = (audio.shape[0] + 32000) // 160
            plen = (n_pitch if p1 is None else p1) - p0
            f, t = O.frame_counts(n)
            got.append((n, plen, f, t))
        assert np.array_equal(np.array(got, dtype=np.int64), g[f"segs_{secs}"]), secs
        out_len = sum(t * 480 - 2 * 48000 for (_, _, _, t) in got)
        assert out_len == int(g[f"outlen_{secs}"])
    # the two BASELINE shapes
    assert tuple(g["segs_10"][0]) == (192000, 1200, 599, 1198)
    assert tuple(g["segs_30"][0]) == (512000, 3200, 1599, 3198)


def test_f0_coarse_bit_exact():
    g = load_golden("f0_coarse")
    for shift in (0, 5, -7):
        coarse, f0bak = O.f0_to_coarse(g["f0"].copy(), shift)
        assert np.array_equal(coarse.astype(np.int64), g[f"coarse_{shift}"])
        assert np.array_equal(f0bak, g[f"f0bak_{shift}"])
        assert coarse.min() >= 1 and coarse.max() <= 255


def test_autotune_bit_exact_and_coarse_ints():
    """Autotune.autotune_f0 (pipeline.py:103-114), fixture from tests/golden/make_golden_a19.py."""
    g = load_golden("autotune")
    assert np.array_equal(np.array(O.REF_FREQS), g["ref_freqs"])
    for strength in (1.0, 0.4):
        assert np.array_equal(O.autotune_f0(g["f0"].copy(), strength), g[f"tuned_{strength}"])
    coarse, f0bak = O.f0_to_coarse(g["f0"].copy(), 3, True, 0.4)
    want_c, want_f = O.f0_to_coarse(g["tuned_0.4"].copy(), 3)
    assert np.array_equal(coarse, want_c) and np.array_equal(f0bak, want_f)


def test_change_rms_matches_reference():
    """AudioProcessor.change_rms (pipeline.py:38-85) run by the reference over the oracle's librosa.feature.rms
    restatement: same float32 ops in the same order -> bit-exact."""
    g = load_golden("change_rms")
    for rate in (0.25, 0.0):
        out = O.change_rms(g["source"], 16000, g["target"], 16000, rate)
        assert out.dtype == np.float32 and np.array_equal(out, g[f"out_{rate}"])
    # librosa.feature.rms by hand on a tiny case: zero centre-padding, hop 2, frame 4
    y = np.arange(1.0, 9.0)
    want = [np.sqrt(np.mean(np.array(f) ** 2)) for f in ([0, 0, 1, 2], [1, 2, 3, 4], [3, 4, 5, 6], [5, 6, 7, 8], [7, 8, 0, 0])]
    assert np.allclose(O.librosa_rms(y, 4, 2)[0], want, rtol=0, atol=1e-15)


def test_mel_filterbank_and_logmel():
    g = load_golden("logmel")
    fb = O.mel_filterbank()
    assert fb.shape == (128, 513) and int((fb != 0).sum()) == int((g["mel_basis"] != 0).sum())
    assert np.abs(fb - g["mel_basis"]).max() <= 1e-8
    mel = O.logmel_rmvpe(torch.from_numpy(g["audio"])).numpy()
    assert mel.shape == g["mel"].shape == (1, 128, 101)
    assert np.abs(mel - g["mel"]).max() <= 1e-5


def test_rmvpe_network_and_decode():
    g = load_golden("rmvpe")
    sd = S.make_rmvpe_state_dict(0)
    f0_from_golden_hidden = O.rmvpe_decode(g["hidden"].copy())
    assert np.array_equal(f0_from_golden_hidden, g["f0"])  # decode on identical salience: bit-exact
    a = torch.from_numpy(g["audio"]).float().unsqueeze(0)
    hidden = O.rmvpe_mel2hidden(O.logmel_rmvpe(a), sd).squeeze(0).numpy()
    assert hidden.shape == g["hidden"].shape
    assert np.abs(hidden - g["hidden"]).max() <= 2e-5


def test_hubert_matches_transformers():
    g = load_golden("hubert")
    sd = S.make_hubert_state_dict(1)
    with torch.no_grad():
        feats = O.hubert_forward(sd, torch.from_numpy(g["wav"])).numpy()
    assert feats.shape == g["feats"].shape == (1, 49, 768)
    assert np.abs(feats - g["feats"]).max() <= 2e-4 * max(1.0, np.abs(g["feats"]).max())


def test_knn_ids_and_blend():
    g = load_golden("knn")
    big = S.synth_index(int(g["index_rows"]), seed=int(g["index_seed"]))
    d2, ids = O.knn_search(big, g["q"], 8, np.float64)
    assert np.array_equal(ids, g["ids"])
    assert np.allclose(d2, g["d2"], rtol=1e-6, atol=0)
    blended = O.knn_blend(g["q"], d2, ids, big, 0.75)
    assert np.abs(blended - g["blended"][0]).max() <= 1e-6
    # the faiss-like float32 path must find the same neighbours on this clustered index
    d2f, idsf = O.knn_search(big, g["q"], 8, np.float32)
    assert (idsf == ids).mean() > 0.99


@pytest.mark.parametrize("tag,sr,voc", SYNTH_CASES)
def test_synthesizer_infer(tag, sr, voc, ref_inputs):
    g = load_golden("synth_" + tag)
    feats, f0c, f0f = ref_inputs
    T = int(g["T"])
    phone = torch.from_numpy(np.repeat(feats, 2, axis=0)[:T]).unsqueeze(0)
    pitch = torch.from_numpy(f0c[:T].astype(np.int64)).unsqueeze(0)
    pitchf = torch.from_numpy(f0f[:T]).float().unsqueeze(0)
    cpt = S.make_synth_checkpoint(sr, voc, seed=0)
    torch.manual_seed(int(g["seed"]))
    o, _, (z, z_p, m_p, logs_p) = O.synthesizer_infer(cpt, phone, torch.tensor([T]), pitch, pitchf,
                                                      torch.tensor([int(g["sid"])]))
    assert np.abs(m_p[0].numpy() - g["m_p"]).max() <= 1e-4
    assert np.abs(logs_p[0].numpy() - g["logs_p"]).max() <= 1e-4
    assert np.abs(z_p[0].numpy() - g["z_p"]).max() <= 1e-4
    assert np.abs(z[0].numpy() - g["z"]).max() <= 1e-4
    out = o[0, 0].numpy()
    assert out.shape == g["o"].shape
    # north_star gate is 1e-3 waveform RMS; the restatement sits orders of magnitude under it
    assert rms(out - g["o"]) <= 2e-5, rms(out - g["o"])
    assert rms(g["o"]) > 0.02  # the comparison is not vacuous


@pytest.mark.parametrize("tag,sr", [("nsf48", 48000), ("nsf40", 40000)])
def test_synthesizer_infer_rate(tag, sr, ref_inputs):
    """`rate` (synthesizers.py:247-251) against the reference's own output (make_golden.py --only-rate)."""
    g = load_golden("synth_rate")
    feats, f0c, f0f = ref_inputs
    T = int(g["T"])
    phone = torch.from_numpy(np.repeat(feats, 2, axis=0)[:T]).unsqueeze(0)
    pitch = torch.from_numpy(f0c[:T].astype(np.int64)).unsqueeze(0)
    pitchf = torch.from_numpy(f0f[:T]).float().unsqueeze(0)
    torch.manual_seed(int(g["seed"]))
    o, _, (z, _, _, _) = O.synthesizer_infer(S.make_synth_checkpoint(sr, "HiFi-GAN", seed=0), phone, torch.tensor([T]), pitch, pitchf,
                                             torch.tensor([int(g["sid"])]), rate=float(g["rate_" + tag]))
    assert z.shape[2] == g["z_" + tag].shape[1] < T
    assert np.abs(z[0].numpy() - g["z_" + tag]).max() <= 1e-4
    out = o[0, 0].numpy()
    assert out.shape == g["o_" + tag].shape
    assert rms(out - g["o_" + tag]) <= 2e-5, rms(out - g["o_" + tag])


@pytest.mark.parametrize("tag,sr", [("nsf48", 48000), ("nsf40", 40000)])
def test_whole_pipeline(tag, sr):
    g = load_golden("pipeline_" + tag)
    cpt = S.make_synth_checkpoint(sr, "HiFi-GAN", seed=0)
    big = S.synth_index(4096, seed=0) if float(g["index_rate"]) > 0 else None
    torch.manual_seed(int(g["seed"]))
    out = O.pipeline(S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0), cpt, g["audio"].copy(),
                     sid=int(g["sid"]), pitch=int(g["pitch"]), big_npy=big, index_rate=float(g["index_rate"]),
                     protect=float(g["protect"]))
    assert out.shape == g["out"].shape
    assert rms(out - g["out"]) <= 1e-4, rms(out - g["out"])


def test_whole_pipeline_v1_model():
    """v1 checkpoints: 256-dim features through HuBERT's final_proj (pipeline.py:451-453); fixture from
    tests/golden/make_golden_v1.py (the reference's Pipeline.pipeline, version="v1")."""
    g = load_golden("pipeline_v1")
    cpt = S.make_synth_checkpoint(40000, "HiFi-GAN", seed=2, version="v1")
    torch.manual_seed(int(g["seed"]))
    out = O.pipeline(S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0), cpt, g["audio"].copy(), sid=int(g["sid"]),
                     big_npy=g["index"], index_rate=float(g["index_rate"]))
    assert out.shape == g["out"].shape
    assert rms(out - g["out"]) <= 2e-5


def test_whole_pipeline_multi_segment():
    """Three segments (pipeline.py:563-577 split points, :614-681 per-segment conversion + crop + concat) at the smaller
    memory tier x_query/x_center/x_max = 1/3/4: fixture from make_golden.py::multiseg (the reference's own Pipeline)."""
    g = load_golden("pipeline_multiseg")
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0)
    taps = {}
    torch.manual_seed(int(g["seed"]))
    out = O.pipeline(S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0), cpt, g["audio"].copy(), sid=int(g["sid"]),
                     big_npy=S.synth_index(4096, seed=0), index_rate=float(g["index_rate"]), protect=float(g["protect"]),
                     x_query=int(g["x_query"]), x_center=int(g["x_center"]), x_max=int(g["x_max"]), taps=taps)
    assert len(taps["opt_ts"]) == 2 and g["segs"].shape == (3, 2)
    assert out.shape == g["out"].shape                       # segment lengths, crops: integer bookkeeping
    assert rms(out - g["out"]) <= 2e-5, rms(out - g["out"])


def test_rmvpe_peaked_matches_reference():
    """The trained-like (unimodal salience) synthetic RMVPE, synthetic.make_rmvpe_state_dict(peaked=True): oracle vs the
    reference's own RMVPE0Predictor on a 3 s clip (fixture: make_golden_peaked.py), plus the property the unconditional
    full-length GPU tests rest on -- one bump per frame, recorded there for the full-length inputs."""
    g = load_golden("rmvpe_peaked")
    assert (float(g["proj_mean"]), float(g["proj_std"])) == pytest.approx(S.PEAKED_STATS[0], rel=1e-5)
    for k in ("30s_seed0", "45s_seed45"):
        assert float(g[k + "_peak_min"]) >= 0.4 and float(g[k + "_off_bump_max"]) <= 1e-4
    sd = S.make_rmvpe_state_dict(0, peaked=True)
    assert np.array_equal(O.rmvpe_decode(g["salience"].copy()), g["f0"])
    taps = {}
    f0 = O.rmvpe_infer_from_audio(g["audio_pad"], sd, taps=taps)
    assert taps["salience"].shape == g["salience"].shape
    assert np.abs(taps["salience"] - g["salience"]).max() <= 2e-5
    assert np.all(f0 > 0) and np.abs(f0 / g["f0"] - 1).max() <= 2e-5


def test_whole_pipeline_peaked_rmvpe_smooth_pitch_embedding():
    g = load_golden("pipeline_peaked")
    cpt = S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0, smooth_pitch=True)
    torch.manual_seed(int(g["seed"]))
    out = O.pipeline(S.make_hubert_state_dict(1), S.make_rmvpe_state_dict(0, peaked=True), cpt, g["audio"].copy(),
                     sid=int(g["sid"]), big_npy=S.synth_index(4096, seed=0), index_rate=float(g["index_rate"]),
                     protect=float(g["protect"]))
    assert out.shape == g["out"].shape
    assert rms(out - g["out"]) <= 2e-5, rms(out - g["out"])
