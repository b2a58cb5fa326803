"""The native reader of the reference's faiss ``IVF{n},Flat`` index files (rvc/train/process/extract_index.py:59-71,
read back at rvc/infer/pipeline.py:553-556).  faiss is absent from the image, so the byte layout is restated from faiss'
published serialisation code and these tests are round trips against the writer in the same module (parity unpinned)."""
import os
import struct

import numpy as np
import pytest

from rvc_amd.lib import faiss_index as FI
from rvc_amd.lib import synthetic as S


def _small_index(n=3000, nlist=None, seed=3):
    big = S.synth_index(n, seed=seed)
    nlist = nlist or min(int(16 * np.sqrt(n)), n // 39)          # extract_index.py:59
    return big, FI.build_ivf_flat(big, nlist, seed=1)


def test_round_trip_reconstructs_big_npy_in_id_order(tmp_path):
    big, ivf = _small_index()
    assert ivf.nlist == 3000 // 39 and ivf.ntotal == 3000 and ivf.nprobe == 1
    path = os.path.join(tmp_path, "added_IVF76_Flat_nprobe_1.index")
    FI.write_index(ivf, path)
    back = FI.read_index(path)
    assert (back.d, back.nlist, back.ntotal, back.nprobe, back.metric) == (768, ivf.nlist, 3000, 1, 1)
    assert np.array_equal(back.centroids, ivf.centroids)
    for a, b in zip(back.list_ids, ivf.list_ids):
        assert np.array_equal(a, b)
    assert np.array_equal(back.reconstruct_n(0, back.ntotal), big)          # what pipeline.py:556 calls big_npy
    assert np.array_equal(back.reconstruct_n(100, 50), big[100:150])
    table = back.padded_lists()
    assert table.dtype == np.int32 and table.shape[0] == ivf.nlist and (np.sort(table[table >= 0]) == np.arange(3000)).all()


def test_file_layout_matches_the_published_format(tmp_path):
    """Byte-level check of the header fields a real faiss file carries (fourcc codes, field widths, list layout)."""
    big, ivf = _small_index(n=400, nlist=10)
    path = os.path.join(tmp_path, "t.index")
    FI.write_index(ivf, path)
    raw = open(path, "rb").read()
    assert raw[:4] == b"IwFl"
    d, ntotal, dummy1, dummy2, trained, metric = struct.unpack_from("<iqqqBi", raw, 4)
    assert (d, ntotal, dummy1, dummy2, trained, metric) == (768, 400, 1 << 20, 1 << 20, 1, 1)
    off = 4 + 33
    nlist, nprobe = struct.unpack_from("<QQ", raw, off)
    assert (nlist, nprobe) == (10, 1) and raw[off + 16: off + 20] == b"IxF2"
    off += 20 + 33
    assert struct.unpack_from("<Q", raw, off)[0] == 10 * 768           # centroid floats
    off += 8 + 10 * 768 * 4
    assert raw[off] == 0 and struct.unpack_from("<Q", raw, off + 1)[0] == 0      # no direct map
    off += 9
    assert raw[off: off + 4] == b"ilar" and struct.unpack_from("<QQ", raw, off + 4) == (10, 3072)
    assert raw[off + 20: off + 24] == b"full"
    sizes = np.frombuffer(raw, dtype="<u8", count=10, offset=off + 32)
    assert int(sizes.sum()) == 400
    assert len(raw) == off + 32 + 80 + 400 * (3072 + 8)


def test_sparse_list_layout_and_empty_lists(tmp_path):
    big = S.synth_index(64, seed=5)
    cent = S.synth_index(9, seed=6)
    ids = [np.array([], dtype=np.int64)] * 9
    ids[2], ids[7] = np.arange(0, 40), np.arange(40, 64)                   # 2 of 9 lists non-empty -> "sprs"
    ivf = FI.IVFFlatIndex(768, cent, ids, [big[i] for i in ids])
    path = os.path.join(tmp_path, "s.index")
    FI.write_index(ivf, path)
    assert b"sprs" in open(path, "rb").read()[:40000]
    back = FI.read_index(path)
    assert [a.shape[0] for a in back.list_ids] == [0, 0, 40, 0, 0, 0, 0, 24, 0]
    assert np.array_equal(back.reconstruct_n(0, 64), big)


def test_unreadable_files_say_so(tmp_path):
    big, ivf = _small_index(n=400, nlist=10)
    good = os.path.join(tmp_path, "g.index")
    FI.write_index(ivf, good)
    raw = open(good, "rb").read()
    cases = {"truncated": raw[: len(raw) // 2], "flat": b"IxF2" + raw[4:], "garbage": os.urandom(4096),
             "empty": b"", "pq": b"IwPQ" + raw[4:]}
    for name, blob in cases.items():
        p = os.path.join(tmp_path, name + ".index")
        open(p, "wb").write(blob)
        with pytest.raises(FI.FaissFormatError) as e:
            FI.read_index(p)
        assert name + ".index" in str(e.value)
    # ids that do not cover 0..ntotal-1 (an index built with add_with_ids): reconstruct_n must refuse, not return zeros
    ivf.list_ids[0] = ivf.list_ids[0] + 10_000
    with pytest.raises(FI.FaissFormatError):
        ivf.reconstruct_n(0, ivf.ntotal)


def test_oracle_ivf_search_semantics():
    """nprobe = 1 returns neighbours from the nearest centroid's list only (approximate), nprobe = nlist is exact."""
    from oracle import rvc_oracle as O
    big, ivf = _small_index(n=2000, nlist=20)
    rng = np.random.default_rng(0)
    q = big[rng.integers(0, 2000, 30)] + 0.02 * rng.standard_normal((30, 768)).astype(np.float32)
    d1, i1 = O.ivf_search(ivf.centroids, ivf.list_ids, big, q, 8, nprobe=1)
    dall, iall = O.ivf_search(ivf.centroids, ivf.list_ids, big, q, 8, nprobe=20)
    dex, iex = O.knn_search(big, q, 8, np.float64)
    assert np.array_equal(iall, iex) and np.allclose(dall, dex, rtol=1e-6, atol=1e-6)   # knn_search returns float32
    a = np.concatenate([np.full(len(m), j) for j, m in enumerate(ivf.list_ids)])[np.argsort(np.concatenate(ivf.list_ids))]
    for row in range(30):
        got = i1[row][i1[row] >= 0]
        assert len(set(a[got])) == 1                      # one list only
    assert (d1 >= dall - 1e-9).all()
