"""ctypes binding of libndgpu_nextcorrect.so -- the host-side mirror of the reference's
Python<->C boundary (lib/nextcorrect.py:19-24,56-90): same struct, same argument order,
same return convention (len, identity, sequence)."""
from __future__ import annotations

import ctypes as C
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per device context (read when HIP initialises)

from . import build as _build

_LIB = None
_BIG = C.c_char * (1 << 30)   # (one array type for every record's view; never instantiated over memory it does not own)


class ConsensusTrimed(C.Structure):
    # lib/nextcorrect.py:19-24
    _fields_ = [("len", C.c_uint), ("identity", C.c_float), ("seq", C.c_void_p)]


PILES_DONE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32), C.c_int)   # ndgpu_piles_done_fn


class Stats(C.Structure):
    _fields_ = [("tasks", C.c_uint64), ("wide_tasks", C.c_uint64), ("cells", C.c_uint64), ("d_steps", C.c_uint64),
                ("trace_bits", C.c_uint64), ("columns", C.c_uint64), ("pool_bases", C.c_uint64), ("seq_bases", C.c_uint64),
                ("max_band", C.c_uint32), ("forward_launches", C.c_uint32), ("forward_ms", C.c_double),
                ("traceback_ms", C.c_double), ("tags_ms", C.c_double), ("links_ms", C.c_double),
                ("score_ms", C.c_double), ("extract_ms", C.c_double), ("piles", C.c_uint64), ("tags", C.c_uint64),
                ("cells_msa", C.c_uint64), ("path_items", C.c_uint64), ("links", C.c_uint64),
                ("score_launches", C.c_uint64), ("backtrack_ms", C.c_double), ("score_segments", C.c_uint64),
                ("score_repairs", C.c_uint64), ("score_slow_piles", C.c_uint64), ("trace_words", C.c_uint64), ("lq_rounds", C.c_uint64), ("lq_declined", C.c_uint64),
                ("lq_ms", C.c_double), ("allocs", C.c_uint64), ("alloc_ms", C.c_double), ("level_allocs", C.c_uint64), ("level_ms", C.c_double),
                ("traceback_launches", C.c_uint64), ("lq_launches", C.c_uint64), ("lq_columns", C.c_uint64), ("lq_aln_columns", C.c_uint64),
                ("lq_bases", C.c_uint64), ("lq_out", C.c_uint64), ("lq_jobs", C.c_uint64), ("lq_repairs", C.c_uint64),
                ("tb_tasks", C.c_uint64), ("tb_walkers", C.c_uint64), ("tb_fallbacks", C.c_uint64)]


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """Load the native library.  Raises (never falls back to a CPU path) when it is absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise RuntimeError("libndgpu_nextcorrect.so is not built; run `python -m nextdenovo_amd.build`")
        _build.build()
    _LIB = _bind(C.CDLL(path))
    return _LIB


def _bind(lib):
    """Argument / result types of the entry points declared in include/ndgpu_nextcorrect.h."""
    lib.nextCorrect.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_uint, C.c_uint,
                                C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_uint, C.c_uint, C.c_int]
    lib.nextCorrect.restype = C.POINTER(ConsensusTrimed)
    lib.free_consensus_trimed.argtypes = [C.POINTER(ConsensusTrimed)]
    lib.ndgpu_correct_batch.argtypes = [C.c_int, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_uint)),
                                        C.POINTER(C.POINTER(C.c_uint)), C.POINTER(C.c_uint), C.POINTER(C.c_uint),
                                        C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_uint,
                                        C.c_uint, C.c_int, C.c_int, C.POINTER(C.POINTER(ConsensusTrimed))]
    lib.ndgpu_correct_batch.restype = C.c_int
    lib.ndgpu_db_create.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ndgpu_db_create.restype = C.c_void_p
    lib.ndgpu_db_destroy.argtypes = [C.c_void_p]
    lib.ndgpu_correct_piles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint,
                                        C.c_uint, C.c_float, C.c_uint, C.c_uint, C.c_int, C.c_int,
                                        C.POINTER(C.POINTER(ConsensusTrimed))]
    lib.ndgpu_correct_piles.restype = C.c_int
    lib.ndgpu_correct_piles_stream.argtypes = lib.ndgpu_correct_piles.argtypes + [PILES_DONE_FN, C.c_void_p]
    lib.ndgpu_correct_piles_stream.restype = C.c_int
    lib.ndgpu_write_records.argtypes = [C.POINTER(C.POINTER(ConsensusTrimed)), C.POINTER(C.c_uint32), C.c_int, C.c_void_p, C.c_uint32, C.c_double,
                                        C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p]
    lib.ndgpu_write_records.restype = C.c_int
    lib.ndgpu_get_stats.argtypes = [C.POINTER(Stats)]
    lib.ndgpu_reset_stats.argtypes = []
    lib.ndgpu_device_count.restype = C.c_int
    return lib


def _take(lib, r):
    ln = r.contents.len
    ide = r.contents.identity
    seq = C.string_at(r.contents.seq, ln) if ln > 4 else b""
    lib.free_consensus_trimed(r)
    return ln, ide, seq


def correct(seqs, aln_start, aln_end, max_aln_length, min_len_aln=500, max_cov_aln=130, min_cov_base=4,
            max_lq_length=10000, min_error_corrected_ratio=0.8, split=0, fast=0, read_type=1):
    """Mirror of lib/nextcorrect.py:72-90 `correct()`; returns (len, identity, sequence bytes)."""
    lib = load()
    n = len(seqs)
    c_seqs = (C.c_char_p * n)()
    c_seqs[:] = seqs
    st = (C.c_uint * n)(*aln_start)
    en = (C.c_uint * n)(*aln_end)
    r = lib.nextCorrect(c_seqs, st, en, n, max_aln_length, min_len_aln, max_cov_aln, min_cov_base, max_lq_length,
                        min_error_corrected_ratio, split, fast, read_type)
    return _take(lib, r)


def correct_batch(piles, min_len_aln=500, max_cov_aln=130, min_cov_base=4, min_error_corrected_ratio=0.8, split=0,
                  fast=0, read_type=1, host_threads=0):
    """piles: list of (seqs, aln_start, aln_end, max_aln_length, max_lq_length).
    Returns a list of (len, identity, sequence bytes) in pile order."""
    lib = load()
    n = len(piles)
    if n == 0:
        return []
    keep = []
    a_seqs = (C.POINTER(C.c_char_p) * n)()
    a_st = (C.POINTER(C.c_uint) * n)()
    a_en = (C.POINTER(C.c_uint) * n)()
    cnt = (C.c_uint * n)()
    mml = (C.c_uint * n)()
    mlq = (C.c_uint * n)()
    for i, (seqs, st, en, max_aln, max_lq) in enumerate(piles):
        k = len(seqs)
        cs = (C.c_char_p * k)()
        cs[:] = seqs
        s = (C.c_uint * k)(*st)
        e = (C.c_uint * k)(*en)
        keep.append((cs, s, e))
        a_seqs[i] = C.cast(cs, C.POINTER(C.c_char_p))
        a_st[i] = C.cast(s, C.POINTER(C.c_uint))
        a_en[i] = C.cast(e, C.POINTER(C.c_uint))
        cnt[i] = k
        mml[i] = max_aln
        mlq[i] = max_lq
    out = (C.POINTER(ConsensusTrimed) * n)()
    lib.ndgpu_correct_batch(n, a_seqs, a_st, a_en, cnt, mml, mlq, min_len_aln, max_cov_aln, min_cov_base,
                            min_error_corrected_ratio, split, fast, read_type, host_threads, out)
    return [_take(lib, out[i]) for i in range(n)]


class ReadDB:
    """Device-resident 2-bit read database (additive replacement for ovlseq.so's
    init_ovls/getseq, lib/nextcorrect.py:62-69).  words/word_off/lens as produced by
    synth.pack_db() or read from a reference .2bit file."""

    def __init__(self, words, word_off, lens):
        import numpy as np
        self._lib = load()
        self._words = np.ascontiguousarray(words, dtype=np.uint32)
        self._off = np.ascontiguousarray(word_off, dtype=np.uint64)
        self._len = np.ascontiguousarray(lens, dtype=np.uint32)
        self.n_reads = int(self._len.size)
        self._h = self._lib.ndgpu_db_create(self.n_reads, self._words.ctypes.data, self._off.ctypes.data,
                                            self._len.ctypes.data)
        if not self._h:
            raise MemoryError("ndgpu_db_create: the read DB does not fit the device memory")

    def close(self):
        if self._h:
            self._lib.ndgpu_db_destroy(self._h)
            self._h = None

    def correct_piles(self, recs, pile_off, min_len_aln=500, max_cov_aln=130, min_cov_base=4, max_lq_length=10000,
                      min_error_corrected_ratio=0.8, split=0, fast=0, read_type=1, host_threads=0, lengths_only=False, fasta=None, lib_wall=None):
        """fasta = (OUT, IDX, seed names, min_len_seed, min_error_corrected_ratio): the accepted records are written as
        lib/nextcorrect.py:236-260 writes them and [(len, identity)] comes back instead of the sequences."""
        import numpy as np
        recs = np.ascontiguousarray(recs, dtype=np.uint32)
        pile_off = np.ascontiguousarray(pile_off, dtype=np.uint64)
        n = int(pile_off.size) - 1
        out = (C.POINTER(ConsensusTrimed) * n)()
        import time
        t0 = time.perf_counter()
        if fasta is not None:
            # the records are handed over sub-batch by sub-batch, while the later ones are still on the device: the parent of
            # lib/nextcorrect.py:232-260 prints each seed as its worker returns it, in no particular order
            OUT, IDX, names, min_len_seed, min_ratio = fasta
            res = [None] * n
            state = {"pos": OUT.tell(), "err": None, "t": 0.0}
            # real files: the library formats and writes a sub-batch's records itself (ndgpu_write_records: one writev per 340
            # records, the bases straight from where it holds them); anything else that has a write(): the loop below, record by record
            try:
                fd_out, fd_idx = OUT.fileno(), (IDX.fileno() if IDX is not None else -1)
                OUT.flush()
                if IDX is not None:
                    IDX.flush()
            except (AttributeError, OSError, ValueError):
                fd_out = None
            if fd_out is not None:
                names32 = np.ascontiguousarray(names, dtype=np.uint32)
                lens, ides = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.float32)
                pos = C.c_uint64(state["pos"])
                wr = self._lib.ndgpu_write_records

                def done(_user, ids, cnt):
                    tw = time.perf_counter()
                    if wr(out, ids, cnt, names32.ctypes.data, int(min_len_seed), float(min_ratio), fd_out, fd_idx, C.byref(pos),
                          lens.ctypes.data, ides.ctypes.data) != 0:
                        state["err"] = OSError("writing the corrected records failed")
                    state["t"] += time.perf_counter() - tw
            else:
                def done(_user, ids, cnt):
                    tw = time.perf_counter()
                    try:
                        self._write_records(out, [ids[k] for k in range(cnt)], res, state, OUT, IDX, names, min_len_seed, min_ratio)
                    except BaseException as e:  # noqa: BLE001  (an exception must not unwind through the C caller)
                        state["err"] = e
                    state["t"] += time.perf_counter() - tw

            cb = PILES_DONE_FN(done)
            rc = self._lib.ndgpu_correct_piles_stream(self._h, n, recs.ctypes.data, pile_off.ctypes.data, min_len_aln, max_cov_aln,
                                                      min_cov_base, max_lq_length, min_error_corrected_ratio, split, fast, read_type,
                                                      host_threads, out, cb, None)
            if state["err"] is not None or rc != 0:
                # records that were never handed over (the call failed, or a hand-over did): they are the caller's to free
                for i in range(n):
                    if out[i]:
                        self._lib.free_consensus_trimed(out[i])
                        out[i] = None
            if state["err"] is not None:
                raise state["err"]
        else:
            rc = self._lib.ndgpu_correct_piles(self._h, n, recs.ctypes.data, pile_off.ctypes.data, min_len_aln, max_cov_aln,
                                               min_cov_base, max_lq_length, min_error_corrected_ratio, split, fast, read_type,
                                               host_threads, out)
        if lib_wall is not None:   # (the library call, hand-over included when it streams; [1]: the time inside the hand-over)
            lib_wall[0] = time.perf_counter() - t0
            if len(lib_wall) > 1:
                lib_wall[1] = state["t"] if fasta is not None else 0.0
        if rc == -2:
            raise ValueError("ndgpu_correct_piles: an overlap record names a read or a window that is not in this read DB "
                             "(sorted.ovl and the .idx / .2bit files do not belong together?)")
        if rc != 0:
            raise RuntimeError("ndgpu_correct_piles failed (%d)" % rc)
        if fasta is not None:
            if fd_out is not None:
                OUT.seek(0, 2)   # (the descriptor moved under the file objects)
                if IDX is not None:
                    IDX.seek(0, 2)
                res = list(zip(lens.tolist(), ides.tolist()))
            return res
        if lengths_only:
            res = []
            for i in range(n):
                res.append((out[i].contents.len, out[i].contents.identity))
                self._lib.free_consensus_trimed(out[i])
            return res
        return [_take(self._lib, out[i]) for i in range(n)]

    def _write_records(self, out, ids, res, state, OUT, IDX, names, min_len_seed, min_ratio):
        """The output loop of lib/nextcorrect.py:236-260 (no -s) straight from the library's records: header, the bases as the
        library holds them (one copy, into the file), the .idx line.  OUT / IDX are binary files; res[i] = (len, identity)."""
        pos = state["pos"]
        free = self._lib.free_consensus_trimed
        for i in ids:
            c = out[i].contents
            ln, ide = c.len, c.identity
            name = int(names[i])
            if ln >= min_len_seed and ln > 4 and ide >= min_ratio:
                head = b">%d %d %f\n" % (name, ln, ide)
                OUT.write(head)
                OUT.write(memoryview(_BIG.from_address(c.seq))[:ln])   # (a view of the library's bytes: no copy before the file's)
                OUT.write(b"\n")
                pos += len(head) + ln + 1
                if IDX is not None:
                    IDX.write(b"%d\t%d\t%d\n" % (name, pos - ln - 1, ln))
            elif ln != 3 and IDX is not None:
                IDX.write(b"%d\t0\t0\n" % name)
            res[i] = (ln, ide)
            free(out[i])
            out[i] = None   # (handed over: nobody frees it again)
        state["pos"] = pos


class ExtJob(C.Structure):
    _fields_ = [("q", C.c_char_p), ("q_len", C.c_int32), ("t", C.c_char_p), ("t_len", C.c_int32), ("max_d", C.c_int32),
                ("band_size", C.c_int32), ("d_factor", C.c_float), ("kind", C.c_int32)]


class ExtResult(C.Structure):
    _fields_ = [("done", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("pos", C.c_uint32 * 6)]


EXT_KINDS = {"ide": 0, "alnpos": 1, "extend_fwd": 2, "extend_rev": 3}


def ext_batch(jobs):
    """jobs: [(kind, q bytes, t bytes, max_d, band_size, d_factor)] -> [(done, a, b, pos[6])] through ndgpu_ext_batch
    (the ide / alnpos / extend_fwd / extend_rev family of lib/align.h:51-58, one device launch)."""
    lib = load()
    n = len(jobs)
    arr = (ExtJob * max(1, n))()
    for i, (kind, q, t, max_d, band, f) in enumerate(jobs):
        arr[i] = ExtJob(q, len(q), t, len(t), max_d, band, f, EXT_KINDS[kind])
    res = (ExtResult * max(1, n))()
    lib.ndgpu_ext_batch.argtypes = [C.POINTER(ExtJob), C.c_int, C.POINTER(ExtResult)]
    lib.ndgpu_ext_batch.restype = C.c_int
    if lib.ndgpu_ext_batch(arr, n, res) != 0:
        raise RuntimeError("ndgpu_ext_batch failed: no usable HIP device?")
    return [(r.done, r.a, r.b, list(r.pos)) for r in res[:n]]


def stats() -> dict:
    lib = load()
    s = Stats()
    lib.ndgpu_get_stats(C.byref(s))
    return {f[0]: getattr(s, f[0]) for f in Stats._fields_}


def release_device_memory() -> int:
    """Hand the consensus contexts' device buffers back (ndgpu_release_memory); returns the bytes released."""
    lib = load()
    lib.ndgpu_release_memory.restype = C.c_uint64
    return int(lib.ndgpu_release_memory())


def reserve_device_memory(n_bytes: int):
    """Device bytes every later consensus call leaves free (ndgpu_reserve_device_memory)."""
    lib = load()
    lib.ndgpu_reserve_device_memory.argtypes = [C.c_uint64]
    lib.ndgpu_reserve_device_memory.restype = None
    lib.ndgpu_reserve_device_memory(int(max(0, n_bytes)))


def reset_stats():
    load().ndgpu_reset_stats()


def device_count() -> int:
    return int(load().ndgpu_device_count())
