"""The N>1 path on CPU: world_size-2 gloo run of the sharding + index broadcast + report reduction."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from rvc_amd.infer import distributed as D
    from rvc_amd.lib import synthetic as S
    r, w, _ = D.init_process_group("gloo")
    assert (r, w) == (rank, world)
    big = S.synth_index(512, seed=0) if rank == 0 else None
    idx = D.broadcast_index(big, "cpu")
    ok = D.checksums_agree(idx)
    utts = [np.full(10 + i, i, dtype=np.float32) for i in range(7)]
    res = D.convert_sharded(utts, lambda i, u: float(u.sum() + idx[i, 0]), rank, world)
    total, tmax = D.reduce_report(sum(len(utts[i]) for i in res), 1.0 + rank, "cpu")
    q.put((rank, D.tensor_checksum(idx), ok, sorted(res.items()), total, tmax))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_sharding_and_broadcast():
    sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]
    from rvc_amd.infer import distributed as D
    from rvc_amd.lib import synthetic as S
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    big = S.synth_index(512, seed=0)
    want_crc = D.tensor_checksum(torch.from_numpy(big))
    (r0, c0, ok0, res0, tot0, t0), (r1, c1, ok1, res1, tot1, t1) = outs
    assert c0 == c1 == want_crc and ok0 and ok1            # every rank holds the root's bytes
    assert [i for i, _ in res0] == [0, 2, 4, 6] and [i for i, _ in res1] == [1, 3, 5]   # i mod world
    merged = dict(res0 + res1)
    # N-rank result == 1-rank result per utterance
    for i in range(7):
        assert merged[i] == float((10 + i) * i + big[i, 0])
    assert tot0 == tot1 == sum(10 + i for i in range(7)) and t0 == t1 == 2.0   # SUM of samples, MAX of seconds


def test_shard_indices_cover_everything_once():
    sys.path[:0] = [REPO, os.path.join(REPO, "codename-rvc-fork-3_amd")]
    from rvc_amd.infer.distributed import shard_indices
    for world in (1, 2, 4, 8):
        for n in (0, 1, 7, 512):
            allidx = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert allidx == list(range(n))
    assert len(shard_indices(512, 3, 8)) == 64  # BASELINE cfg 3: 512 utterances -> 64 per rank
