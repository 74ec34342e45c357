"""Reader for NextDenovo's varint overlap files and 2-bit read DBs (host side, numpy).

Formats (bit-exact):
* `.ovl` step-1 / sorted: stream of records of 8 big-endian base-128 varints
  (reference lib/ovl.c:109-203): |d qname|, flags, qs, qe-qs, |d tname|, ts, |len diff|, match;
  decoded to [qname, rev, qs, qe, tname, ts, te, match] exactly as decode_ovl returns them.
* `.2bit` + `.idx`: 2 magic bytes, then per read u32 id, u32 len, ceil(len/16) u32 words
  (lib/bseq.c:93-139); idx lines `id \\t byte offset of first word \\t len` (util/seq_dump.c:39).
"""
from __future__ import annotations

import os

import numpy as np


def decode_varints(raw: np.ndarray) -> np.ndarray:
    """All varints of a byte buffer -> uint64 values (vectorised)."""
    b = raw.astype(np.uint8)
    last = b < 128
    n_val = int(last.sum())
    if n_val == 0:
        return np.zeros(0, dtype=np.uint64)
    end = np.nonzero(last)[0]
    if end[-1] != b.size - 1:  # trailing partial value: ignore
        b = b[: end[-1] + 1]
        last = last[: end[-1] + 1]
    gid = np.cumsum(last) - last  # value index of every byte
    dist = end[gid] - np.arange(b.size)  # bytes until the terminating byte
    vals = np.zeros(n_val, dtype=np.uint64)
    np.add.at(vals, gid, (b & 127).astype(np.uint64) << (7 * dist).astype(np.uint64))
    return vals


def _native():
    import ctypes as C
    from . import overlap
    lib = overlap.load()
    lib.ndgpu_ovl_decode.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.ndgpu_ovl_decode.restype = C.c_int64
    lib.ndgpu_2bit_index.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.ndgpu_2bit_index.restype = C.c_int64
    return lib


def decode_ovl(path: str) -> np.ndarray:
    """Whole 8-field .ovl file -> uint32 array [n, 8] in decode_ovl order (lib/ovl.c:189-200): one native pass
    (ndgpu_ovl_decode) over the memory-mapped file."""
    if os.path.getsize(path) == 0:
        return np.zeros((0, 8), dtype=np.uint32)
    raw = np.memmap(path, dtype=np.uint8, mode="r")
    n_rec = int(np.count_nonzero(np.asarray(raw) < 128)) // 8
    out = np.empty((n_rec, 8), dtype=np.uint32)
    prev = np.zeros(2, dtype=np.uint32)
    buf = np.ascontiguousarray(raw)
    got = _native().ndgpu_ovl_decode(buf.ctypes.data, buf.size, prev.ctypes.data, out.ctypes.data, n_rec, None)
    return out[:got]


def decode_bytes(blob: bytes) -> np.ndarray:
    """decode_ovl() of an in-memory .ovl image."""
    buf = np.frombuffer(blob, dtype=np.uint8)
    if buf.size == 0:
        return np.zeros((0, 8), dtype=np.uint32)
    n_rec = int(np.count_nonzero(buf < 128)) // 8
    out = np.empty((n_rec, 8), dtype=np.uint32)
    prev = np.zeros(2, dtype=np.uint32)
    buf = np.ascontiguousarray(buf)
    got = _native().ndgpu_ovl_decode(buf.ctypes.data, buf.size, prev.ctypes.data, out.ctypes.data, n_rec, None)
    return out[:got]


def decode_ovl_numpy(path: str) -> np.ndarray:
    """The same decoder in numpy (kept as the cross-check of the native one in tests)."""
    raw = np.fromfile(path, dtype=np.uint8)
    v = decode_varints(raw)
    n = v.size // 8
    v = v[: n * 8].reshape(n, 8).astype(np.int64)
    flags = v[:, 1]
    dq = np.where(flags & 2, -v[:, 0], v[:, 0])
    dt = np.where(flags & 4, -v[:, 4], v[:, 4])
    out = np.empty((n, 8), dtype=np.int64)
    out[:, 0] = np.cumsum(dq)
    out[:, 1] = flags & 1
    out[:, 2] = v[:, 2]
    out[:, 3] = v[:, 3] + v[:, 2]
    out[:, 4] = np.cumsum(dt)
    out[:, 5] = v[:, 5]
    out[:, 6] = np.where(flags & 8, v[:, 5] + v[:, 3] + v[:, 6], v[:, 5] + v[:, 3] - v[:, 6])
    out[:, 7] = v[:, 7]
    return out.astype(np.uint32)


def twobit_name(idx_path: str) -> str:
    """`/dir/.input.seed.001.idx` -> `/dir/input.seed.001.2bit` (lib/ovlseq.c:24-37)."""
    d, f = os.path.split(idx_path)
    return os.path.join(d, f[1:-4] + ".2bit")


def load_read_db(idx_fofn: str):
    """Every `.2bit` named by the idx list -> (words uint32, word_off uint64[n], len uint32[n])
    indexed by read id, ready for api.ReadDB (ids are dense, util/seq_dump.c:83-84)."""
    chunks, base = [], 0
    ids, offs, lens = [], [], []
    with open(idx_fofn) as f:
        files = [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
    for idx in files:
        data = np.fromfile(twobit_name(idx), dtype=np.uint8)
        w = np.frombuffer(data[2:2 + ((data.size - 2) // 4) * 4].tobytes(), dtype=np.uint32)
        tab = np.loadtxt(idx, dtype=np.int64, ndmin=2) if os.path.getsize(idx) else np.zeros((0, 3), dtype=np.int64)
        ids.append(tab[:, 0])
        offs.append((tab[:, 1] - 2) // 4 + base)
        lens.append(tab[:, 2])
        chunks.append(w)
        base += w.size
    ids = np.concatenate(ids) if ids else np.zeros(0, dtype=np.int64)
    n = int(ids.max()) + 1 if ids.size else 0
    word_off = np.zeros(n, dtype=np.uint64)
    rlen = np.zeros(n, dtype=np.uint32)
    word_off[ids] = np.concatenate(offs).astype(np.uint64)
    rlen[ids] = np.concatenate(lens).astype(np.uint32)
    words = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint32)
    return words, word_off, rlen


def read_2bit(path: str):
    """One `.2bit` file -> (ids uint32[n], lens uint32[n], words uint32, word_off uint64[n]) in file
    order (lib/bseq.c:257-299 kbit_read: u32 id, u32 len, ceil(len/16) words per read)."""
    data = np.fromfile(path, dtype=np.uint8)
    w = np.frombuffer(data[2:2 + ((data.size - 2) // 4) * 4].tobytes(), dtype=np.uint32)
    lib = _native()
    n = int(lib.ndgpu_2bit_index(w.ctypes.data, w.size, None, None, None, 0))
    if n < 0:
        raise ValueError("%s: truncated or corrupt .2bit file (a record runs past the end of the file)" % path)
    ids, lens, offs = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint64)
    lib.ndgpu_2bit_index(w.ctypes.data, w.size, ids.ctypes.data, lens.ctypes.data, offs.ctypes.data, n)
    return ids, lens, w, offs


def unpack_codes(words: np.ndarray, word_off: np.ndarray, lens: np.ndarray):
    """2-bit words (16 bases per u32, first base in the top bits) -> (codes uint8, off uint64[n])."""
    lens = np.asarray(lens, dtype=np.int64)
    off = np.zeros(lens.size, dtype=np.uint64)
    if lens.size:
        off[1:] = np.cumsum(lens)[:-1]
    out = np.empty(int(lens.sum()), dtype=np.uint8)
    sh = (30 - 2 * np.arange(16)).astype(np.uint32)
    for i in range(lens.size):
        n = int(lens[i])
        cnt = (n + 15) >> 4
        ws = words[int(word_off[i]): int(word_off[i]) + cnt]
        out[int(off[i]): int(off[i]) + n] = ((ws[:, None] >> sh[None, :]) & 3).astype(np.uint8).reshape(-1)[:n]
    return out, off


def write_2bit(path: str, ids, lens, words, word_off):
    """Writer of the `.2bit` container (lib/bseq.c:93-139): magic {0,254}, then u32 id, u32 len, words per read."""
    ids = np.asarray(ids, dtype=np.uint32)
    lens = np.asarray(lens, dtype=np.uint32)
    word_off = np.asarray(word_off, dtype=np.int64)
    cnt = (lens.astype(np.int64) + 15) >> 4
    start = np.zeros(ids.size + 1, dtype=np.int64)
    np.cumsum(cnt + 2, out=start[1:])
    out = np.empty(int(start[-1]), dtype=np.uint32)
    out[start[:-1]] = ids
    out[start[:-1] + 1] = lens
    # sequence words of all reads in one gather
    total = int(cnt.sum())
    if total:
        r = np.repeat(np.arange(ids.size, dtype=np.int64), cnt)
        within = np.arange(total, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt)
        out[start[:-1][r] + 2 + within] = np.asarray(words, dtype=np.uint32)[word_off[r] + within]
    with open(path, "wb") as f:
        f.write(bytes([0, 254]))
        f.write(out.tobytes())
