"""Native reader (and a minimal writer) of the faiss index files the reference produces and consumes -- no faiss import.

The reference builds its retrieval index with ``faiss.index_factory(768, f"IVF{n_ivf},Flat")``, ``nprobe = 1``, trains it,
adds ``big_npy`` in batches of 8192 and saves it with ``faiss.write_index`` (rvc/train/process/extract_index.py:59-71); at
inference it does ``index = faiss.read_index(path); big_npy = index.reconstruct_n(0, index.ntotal)`` and
``index.search(npy, k=8)`` (rvc/infer/pipeline.py:499, 553-556).

faiss (pinned faiss-cpu 1.7.3) is a third-party dependency whose source is not under /root/reference and which is absent
from this image, so the byte layout below is restated from faiss' published serialisation code
(faiss/impl/index_write.cpp / index_read.cpp, v1.7.x) and is **parity unpinned**: it round-trips with the writer in this
file, but no file written by faiss itself was available to check it against.

Layout of an ``IndexIVFFlat`` file (little-endian, no padding):

    "IwFl"                                   fourcc
    index header   d:int32  ntotal:int64  dummy:int64  dummy:int64  is_trained:uint8  metric_type:int32 (1 = L2)
    nlist:uint64  nprobe:uint64
    quantizer      "IxF2"  index header (d, ntotal = nlist, ...)  n_units:uint64 (= nlist*d)  float32[nlist*d]
    direct map     type:uint8  n:uint64  int64[n]                (type 0 = none, n = 0)
    inverted lists "ilar"  nlist:uint64  code_size:uint64 (= 4 d)
                   "full"  n:uint64 (= nlist)  uint64[nlist] list sizes
                 | "sprs"  n:uint64  (list_no, size) pairs for the non-empty lists
                   for every non-empty list, in list order:  codes uint8[size*code_size]  ids int64[size]

``reconstruct_n(0, ntotal)`` scatters every stored vector to row ``id`` (faiss IndexIVF::reconstruct_n), which for an index
filled by sequential ``add`` calls is ``big_npy`` in the order it was added.
"""
from __future__ import annotations

import struct
from typing import List

import numpy as np


class FaissFormatError(ValueError):
    pass


class _Cursor:
    def __init__(self, buf: memoryview, path: str):
        self.buf, self.pos, self.path = buf, 0, path

    def take(self, n: int) -> memoryview:
        if n < 0 or self.pos + n > len(self.buf):
            raise FaissFormatError(f"{self.path}: truncated at byte {self.pos} (wanted {n} more bytes of {len(self.buf)})")
        out = self.buf[self.pos:self.pos + n]
        self.pos += n
        return out

    def scalar(self, fmt: str):
        return struct.unpack("<" + fmt, self.take(struct.calcsize("<" + fmt)))[0]

    def fourcc(self) -> str:
        return bytes(self.take(4)).decode("latin-1")

    def array(self, dtype, count: int) -> np.ndarray:
        dt = np.dtype(dtype).newbyteorder("<")
        return np.frombuffer(self.take(count * dt.itemsize), dtype=dt, count=count)

    def vector(self, dtype) -> np.ndarray:
        """faiss WRITEVECTOR: element count as uint64, then the elements."""
        return self.array(dtype, self.scalar("Q"))


def _index_header(c: _Cursor):
    d = c.scalar("i")
    ntotal = c.scalar("q")
    c.scalar("q"), c.scalar("q")          # two historical dummies (1 << 20)
    is_trained = c.scalar("B")
    metric = c.scalar("i")
    if metric > 1:
        c.scalar("f")                     # metric_arg
    if d <= 0 or ntotal < 0:
        raise FaissFormatError(f"{c.path}: implausible header (d = {d}, ntotal = {ntotal})")
    return d, ntotal, bool(is_trained), metric


class IVFFlatIndex:
    """What the path needs of a faiss ``IndexIVFFlat``: the coarse centroids, the inverted lists, the vectors by id."""

    def __init__(self, d: int, centroids: np.ndarray, list_ids: List[np.ndarray], list_vecs: List[np.ndarray], nprobe: int = 1,
                 metric: int = 1):
        self.d, self.nprobe, self.metric = int(d), int(nprobe), int(metric)
        self.centroids = np.ascontiguousarray(centroids, dtype=np.float32).reshape(-1, self.d)
        self.nlist = self.centroids.shape[0]
        if len(list_ids) != self.nlist or len(list_vecs) != self.nlist:
            raise FaissFormatError("one id array and one vector array per inverted list, please")
        self.list_ids = [np.ascontiguousarray(a, dtype=np.int64) for a in list_ids]
        self.list_vecs = [np.ascontiguousarray(v, dtype=np.float32).reshape(-1, self.d) for v in list_vecs]
        self.ntotal = int(sum(a.shape[0] for a in self.list_ids))

    def reconstruct_n(self, i0: int = 0, n: int | None = None) -> np.ndarray:
        """faiss IndexIVF::reconstruct_n: row (id - i0) <- the vector stored under ``id`` (pipeline.py:556)."""
        n = self.ntotal - i0 if n is None else n
        if not (0 <= i0 and i0 + n <= self.ntotal):
            raise IndexError(f"reconstruct_n({i0}, {n}) outside [0, {self.ntotal})")
        out = np.zeros((n, self.d), dtype=np.float32)
        seen = np.zeros(n, dtype=bool)
        for ids, vecs in zip(self.list_ids, self.list_vecs):
            keep = (ids >= i0) & (ids < i0 + n)
            out[ids[keep] - i0] = vecs[keep]
            seen[ids[keep] - i0] = True
        if not seen.all():
            raise FaissFormatError(f"{int((~seen).sum())} of the ids {i0}..{i0 + n - 1} are not stored in any inverted list "
                                   "(the reference only writes indices filled by sequential add calls)")
        return out

    def padded_lists(self) -> np.ndarray:
        """[nlist, max list length] int32 table of member ids, -1 padded (what the device-side nprobe search gathers)."""
        width = max(1, max(a.shape[0] for a in self.list_ids))
        table = np.full((self.nlist, width), -1, dtype=np.int32)
        for i, a in enumerate(self.list_ids):
            table[i, :a.shape[0]] = a
        return table


def read_index(path: str) -> IVFFlatIndex:
    """``faiss.read_index`` for the one index type the reference writes (``IVF{n},Flat``)."""
    with open(path, "rb") as f:
        data = f.read()
    c = _Cursor(memoryview(data), path)
    kind = c.fourcc()
    if kind != "IwFl":
        known = {"IxF2": "a flat L2 index", "IxFI": "a flat inner-product index", "IwPQ": "an IVF-PQ index", "IvFl": "a legacy (pre-1.5) IVF-Flat index",
                 "IxPT": "an index with a pre-transform", "IHNf": "an HNSW index"}
        raise FaissFormatError(f"{path}: expected a faiss IndexIVFFlat ('IwFl', what extract_index.py:62 builds), found "
                               f"{kind!r}{' = ' + known[kind] if kind in known else ''}")
    d, ntotal, _, metric = _index_header(c)
    nlist, nprobe = c.scalar("Q"), c.scalar("Q")
    qkind = c.fourcc()
    if qkind not in ("IxF2", "IxFI", "IxFl"):
        raise FaissFormatError(f"{path}: coarse quantizer {qkind!r} is not a flat index")
    qd, qn, _, _ = _index_header(c)
    centroids = c.vector(np.float32)
    if qd != d or qn != nlist or centroids.size != nlist * d:
        raise FaissFormatError(f"{path}: quantizer holds {centroids.size} floats for {qn} x {qd} centroids (nlist {nlist}, d {d})")
    dm_type = c.scalar("B")
    c.vector(np.int64)                                    # direct-map array (empty unless maintained)
    if dm_type == 2:                                      # hashtable direct map: (id, location) pairs
        c.array(np.int64, 2 * c.scalar("Q"))
    ilk = c.fourcc()
    if ilk != "ilar":
        raise FaissFormatError(f"{path}: inverted lists of kind {ilk!r} (only in-file array lists 'ilar' are supported)")
    il_nlist, code_size = c.scalar("Q"), c.scalar("Q")
    if il_nlist != nlist or code_size != 4 * d:
        raise FaissFormatError(f"{path}: inverted lists say nlist {il_nlist}, code size {code_size}; header says {nlist}, {4 * d}")
    layout = c.fourcc()
    raw = c.vector(np.uint64)
    sizes = np.zeros(nlist, dtype=np.int64)
    if layout == "full":
        if raw.size != nlist:
            raise FaissFormatError(f"{path}: {raw.size} list sizes for {nlist} lists")
        sizes[:] = raw
    elif layout == "sprs":
        pairs = raw.reshape(-1, 2)
        sizes[pairs[:, 0].astype(np.int64)] = pairs[:, 1]
    else:
        raise FaissFormatError(f"{path}: unknown inverted-list layout {layout!r}")
    if int(sizes.sum()) != ntotal:
        raise FaissFormatError(f"{path}: the inverted lists hold {int(sizes.sum())} vectors, the header says {ntotal}")
    list_ids, list_vecs = [], []
    for n in sizes:
        n = int(n)
        list_vecs.append(c.array(np.float32, n * d).reshape(n, d))
        list_ids.append(c.array(np.int64, n))
    return IVFFlatIndex(d, centroids.reshape(nlist, d), list_ids, list_vecs, nprobe=nprobe, metric=metric)


def write_index(index: IVFFlatIndex, path: str) -> None:
    """The inverse of ``read_index`` (same layout ``faiss.write_index`` produces for an IndexIVFFlat)."""
    d, nlist = index.d, index.nlist

    def header(ntotal):
        return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, index.metric)

    out = [b"IwFl", header(index.ntotal), struct.pack("<QQ", nlist, index.nprobe),
           b"IxF2" if index.metric == 1 else b"IxFI", header(nlist), struct.pack("<Q", nlist * d),
           index.centroids.astype("<f4").tobytes(), struct.pack("<BQ", 0, 0),
           b"ilar", struct.pack("<QQ", nlist, 4 * d)]
    sizes = np.array([a.shape[0] for a in index.list_ids], dtype="<u8")
    if int((sizes > 0).sum()) > nlist // 2:
        out += [b"full", struct.pack("<Q", nlist), sizes.tobytes()]
    else:
        nz = np.nonzero(sizes)[0]
        pairs = np.stack([nz.astype("<u8"), sizes[nz]], 1).astype("<u8")
        out += [b"sprs", struct.pack("<Q", pairs.size), pairs.tobytes()]
    for ids, vecs in zip(index.list_ids, index.list_vecs):
        if ids.shape[0]:
            out += [vecs.astype("<f4").tobytes(), ids.astype("<i8").tobytes()]
    with open(path, "wb") as f:
        for piece in out:
            f.write(piece)


def build_ivf_flat(big_npy: np.ndarray, nlist: int, seed: int = 0, iterations: int = 4) -> IVFFlatIndex:
    """A small stand-in for ``index.train`` + ``index.add`` (extract_index.py:65-69) so that ``.index`` files can be
    produced without faiss: a few Lloyd iterations from a random subset, then every row into the list of its nearest
    centroid under ids 0..N-1 in add order.  (faiss' own k-means differs in initialisation and iteration count; any
    centroid set gives a valid index.)"""
    x = np.ascontiguousarray(big_npy, dtype=np.float32)
    n, d = x.shape
    nlist = int(max(1, min(nlist, n)))
    rng = np.random.default_rng(seed)
    cent = x[rng.choice(n, nlist, replace=False)].copy()

    def assign(c):
        best = np.empty(n, dtype=np.int64)
        cn = (c.astype(np.float64) ** 2).sum(1)
        for s in range(0, n, 16384):
            blk = x[s:s + 16384].astype(np.float64)
            best[s:s + 16384] = np.argmin(cn[None, :] - 2.0 * blk @ c.astype(np.float64).T, axis=1)
        return best

    for _ in range(iterations):
        a = assign(cent)
        for j in range(nlist):
            m = a == j
            if m.any():
                cent[j] = x[m].mean(0)
    a = assign(cent)
    order = np.argsort(a, kind="stable")
    bounds = np.searchsorted(a[order], np.arange(nlist + 1))
    ids = [order[bounds[j]:bounds[j + 1]].astype(np.int64) for j in range(nlist)]
    return IVFFlatIndex(d, cent, ids, [x[i] for i in ids], nprobe=1)
