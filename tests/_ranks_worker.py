"""Rank process of tests/test_multirank_gpu.py (started by rvc_amd.infer.distributed.spawn_ranks): joins the group, takes the
index from rank 0's broadcast, converts its i mod world share of the utterances under per-utterance noise seeds and writes
the waveforms + the index checksum to OUT_DIR/rank<r>.npz.  Mirrors bench.py's rank body without the timing."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]


def main():
    import torch
    from rvc_amd.infer import distributed as D
    from rvc_amd.infer.infer import VoiceConverter
    from rvc_amd.lib import synthetic as S
    out_dir, n_utt, secs = os.environ["OUT_DIR"], int(os.environ["N_UTT"]), float(os.environ["UTT_SECONDS"])
    rank, world, local = D.init_process_group()
    assert torch.cuda.is_available()
    dev = f"cuda:{local % torch.cuda.device_count()}"
    torch.cuda.set_device(dev)
    vc = VoiceConverter(device=dev)
    # trained-like RMVPE / pitch embedding: the random-head RMVPE flips an arg-max on ~1 frame in 10^4 under the library GEMMs'
    # run-to-run noise, which would show as 1-2e-5 between the 2-rank and the 1-rank job
    vc.load_checkpoint_dict(S.make_synth_checkpoint(48000, "HiFi-GAN", seed=0, smooth_pitch=True))
    vc.load_hubert_state_dict(S.make_hubert_state_dict(1))
    vc.vc.load_rmvpe_state_dict(S.make_rmvpe_state_dict(0, peaked=True))
    big = S.synth_index(20_000, seed=0) if rank == 0 else None
    index = D.broadcast_index(big, dev)
    agree = D.checksums_agree(index)
    vc.vc.set_index(index)
    utts = [S.synth_audio(int(secs * 16000), seed=100 + i) for i in range(n_utt)]

    def convert(i, audio):
        return vc.vc.pipeline(vc.hubert_model, vc.net_g, 0, audio.copy(), 0, "rmvpe", "", 0.75, True, 3, 1, "v2", 0.5, 128,
                              False, 1, None, noise_seed=1000 + i)
    res = D.convert_sharded(utts, convert, rank, world)
    total, t_max = D.reduce_report(sum(len(v) for v in res.values()), 1.0 + rank, dev if world == 1 or os.environ.get("RVC_DIST_BACKEND") != "gloo" else "cpu")
    s1, s2 = D.tensor_checksum(index)
    np.savez(os.path.join(out_dir, f"rank{rank}_of{world}.npz"), checksum=np.array([s1, s2], dtype=np.uint64),
             agree=np.array(agree), total=np.int64(total), t_max=np.float64(t_max), transport=np.array(D.last_broadcast_info().get("transport", "none")),
             **{f"utt{i}": v for i, v in res.items()})
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
