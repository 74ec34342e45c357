"""ctypes binding of the MI355X overlap engine (include/ndgpu_overlap.h, libndgpu_overlap.so).

The host-side mirror of what `minimap2-nd --step 1` does around its C core: build the index of the
target reads, map the query reads, encode the overlaps (see nextdenovo_amd/minimap2_nd.py for the CLI).
There is no CPU path: every call needs a HIP device.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # one hardware queue per device context (read when HIP initialises)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libndgpu_overlap.so")


class Opt(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("k", "w", "hpc", "no_diag", "no_dual", "min_cnt", "min_chain_score", "bw", "max_gap",
                                         "max_chain_skip", "max_chain_iter", "minlen", "seed", "dvt", "maxhan1", "maxhan2")] \
        + [("mid_occ_frac", C.c_float), ("mid_occ", C.c_int32), ("mode", C.c_int32), ("d_factor", C.c_float), ("step", C.c_int32),
           ("minide", C.c_float), ("minmatch", C.c_int32), ("max_occ", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("sketch_ms", "index_sort_ms", "seed_ms", "sort_ms", "exact_sort_ms", "chain_ms",
                                          "hits_ms")] \
        + [(n, C.c_uint64) for n in ("bases_sketched", "minimizers", "anchors", "tie_reads", "chain_cells", "chains",
                                     "overlaps", "map_calls", "batches", "ext_problems", "ext_launches")] \
        + [("ext_ms", C.c_double), ("rechained", C.c_uint64)]


REC = np.dtype([(n, np.uint32) for n in ("rev", "qname", "qs", "qe", "tname", "ts", "te", "match")])
REC10 = np.dtype([(n, np.uint32) for n in ("rev", "qname", "qs", "qe", "qlen", "tname", "ts", "te", "tlen", "identity")])  # --step 2

_lib = None


class AlnOpt(C.Structure):
    """ndgpu_ovl_aln_opt: the scoring side of mm_mapopt_t that -c uses (minimap2/options.c:36-43)."""
    _fields_ = [(n, C.c_int32) for n in ("a", "b", "q", "e", "q2", "e2", "sc_ambi", "zdrop", "zdrop_inv", "end_bonus", "min_dp_max", "min_ksw_len")] \
        + [("max_sw_mat", C.c_int64), ("host_threads", C.c_int32)]


class CigarStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("chains", "first_pass", "second_pass", "inversion_tests", "inversions", "cells", "overlaps", "inversions_aligned",
                                         "splits", "chains_ns", "ksw_ns", "ksw_ll_ns", "total_ns")]


def aln_opt(**kw) -> AlnOpt:
    o = AlnOpt()
    load().ndgpu_ovl_aln_opt_default(C.byref(o), 40)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build
        build.build()
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def _bind(lib):
    """Argument / result types of the entry points declared in include/ndgpu_overlap.h."""
    P = C.c_void_p
    lib.ndgpu_ovl_opt_preset.argtypes = [C.c_char_p, C.POINTER(Opt)]
    lib.ndgpu_ovl_index_create.argtypes = [C.POINTER(Opt), C.c_uint32, P, C.c_uint64, P, P, P]
    lib.ndgpu_ovl_index_create.restype = P
    lib.ndgpu_ovl_index_destroy.argtypes = [P]
    lib.ndgpu_ovl_index_mid_occ.argtypes = [P, C.c_float]
    lib.ndgpu_ovl_index_mid_occ.restype = C.c_int32
    lib.ndgpu_ovl_index_stat.argtypes = [P, P]
    lib.ndgpu_ovl_map.argtypes = [P, C.POINTER(Opt), C.c_int32, C.c_uint32, P, C.c_uint64, P, P, P, C.POINTER(P)]
    lib.ndgpu_ovl_map.restype = C.c_int64
    lib.ndgpu_ovl_encode.argtypes = [P, C.c_int64, P, P]
    lib.ndgpu_ovl_encode.restype = C.c_int64
    lib.ndgpu_ovl_free.argtypes = [P]
    lib.ndgpu_ovl_sketch.argtypes = [C.POINTER(Opt), C.c_uint32, P, C.c_uint64, P, P, C.c_int, C.POINTER(P), C.POINTER(P), P]
    lib.ndgpu_ovl_sketch.restype = C.c_int64
    lib.ndgpu_ovl_index_dump.argtypes = [P, P, P, P]
    lib.ndgpu_ovl_debug_anchors.argtypes = [P, C.c_uint32, C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(P)]
    lib.ndgpu_ovl_debug_anchors.restype = C.c_int64
    lib.ndgpu_pack_2bit.argtypes = [C.c_uint32, P, C.c_uint64, P, P, P, P]
    lib.ndgpu_pack_2bit.restype = C.c_int64
    lib.ndgpu_ovl_get_stats.argtypes = [P, C.POINTER(Stats)]
    lib.ndgpu_ovl_reset_stats.argtypes = [P]
    lib.ndgpu_ovl_map2.argtypes = lib.ndgpu_ovl_map.argtypes
    lib.ndgpu_ovl_map2.restype = C.c_int64
    lib.ndgpu_ovl_map_regs.argtypes = [P, C.POINTER(Opt), C.c_int32, C.c_uint32, P, C.c_uint64, P, P, P, P, P, C.c_int, C.POINTER(P), C.POINTER(P), P]
    lib.ndgpu_ovl_map_regs.restype = C.c_int64
    lib.ndgpu_ovl_map2_realign.argtypes = [P, P, P, C.POINTER(Opt), C.c_int32, C.c_int32, C.c_uint32, P, C.c_uint64, P, P, P, C.c_uint32, P, C.c_uint64,
                                           P, P, P, C.POINTER(P)]
    lib.ndgpu_ovl_map2_realign.restype = C.c_int64
    lib.ndgpu_ovl_aln_opt_default.argtypes = [C.POINTER(AlnOpt), C.c_int32]
    lib.ndgpu_ovl_map_chains.argtypes = [P, C.POINTER(Opt), C.c_int32, C.c_uint32, P, C.c_uint64, P, P, P, C.POINTER(P), C.POINTER(P), C.POINTER(P),
                                         C.POINTER(P), C.POINTER(P)]
    lib.ndgpu_ovl_map_chains.restype = C.c_int64
    lib.ndgpu_ovl_map_cigar.argtypes = [P, C.POINTER(Opt), C.POINTER(AlnOpt), C.c_int32, C.c_uint32, P, C.c_uint64, P, P, P, P, P, P, P, C.POINTER(P),
                                        C.POINTER(CigarStats)]
    lib.ndgpu_ovl_map_cigar.restype = C.c_int64
    lib.ndgpu_s2_new.restype = P
    lib.ndgpu_s2_free.argtypes = [P]
    lib.ndgpu_s2_filter_encode.argtypes = [P, P, C.c_int64, C.c_int32, C.c_int32, P, C.POINTER(P), P]
    lib.ndgpu_s2_filter_encode.restype = C.c_int64
    lib.ndgpu_s2_bl.argtypes = [P, C.POINTER(P)]
    lib.ndgpu_s2_bl.restype = C.c_int64
    return lib


def preset(name: str, **kw) -> Opt:
    o = Opt()
    if load().ndgpu_ovl_opt_preset(name.encode() if name else None, C.byref(o)) != 0:
        raise ValueError("unsupported preset %r" % name)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _take(lib, p, n, dtype):
    """The library's malloc'd block as an array.  Small blocks are copied and freed; a large one IS the array's memory (freed when the
    last array over it goes): the copy of 18 MB of sorted records was 3 ms of every step."""
    nbytes = int(n) * np.dtype(dtype).itemsize
    if nbytes < (1 << 20):
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).copy() if n else np.zeros(0, dtype=np.uint8)
        lib.ndgpu_ovl_free(p)
        return a.view(dtype)
    addr = p.value if isinstance(p, C.c_void_p) else C.cast(p, C.c_void_p).value
    block = (C.c_uint8 * nbytes).from_address(addr)
    weakref.finalize(block, lib.ndgpu_ovl_free, C.c_void_p(addr))
    return np.frombuffer(block, dtype=dtype)


class ReadSet:
    """Reads as stored in a .2bit file (see ovl.read_2bit)."""

    def __init__(self, ids, lens, words, word_off):
        self.ids = np.ascontiguousarray(ids, dtype=np.uint32)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint32)
        self.words = np.ascontiguousarray(words, dtype=np.uint32)
        self.word_off = np.ascontiguousarray(word_off, dtype=np.uint64)

    @classmethod
    def from_2bit(cls, path):
        from . import ovl
        return cls(*ovl.read_2bit(path))

    def __len__(self):
        return int(self.ids.size)

    def subset(self, lo, hi):
        return ReadSet(self.ids[lo:hi], self.lens[lo:hi], self.words, self.word_off[lo:hi])


def sketch(opt: Opt, rs: ReadSet, rid_is_index=False):
    """K1 only -> (x, y, off)."""
    lib = load()
    x, y = C.c_void_p(), C.c_void_p()
    off = np.zeros(len(rs) + 1, dtype=np.uint64)
    n = lib.ndgpu_ovl_sketch(C.byref(opt), len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off), _ptr(rs.lens),
                             1 if rid_is_index else 0, C.byref(x), C.byref(y), _ptr(off))
    if n < 0:
        raise RuntimeError("ndgpu_ovl_sketch failed (%d): no usable HIP device?" % n)
    return _take(lib, x, n, np.uint64), _take(lib, y, n, np.uint64), off


def _fail(lib, what):
    """MemoryError when the library says the call ran out of device memory (the caller may release memory and retry),
    RuntimeError otherwise."""
    lib.ndgpu_ovl_last_error.restype = C.c_int
    if int(lib.ndgpu_ovl_last_error()) == 1:
        return MemoryError(what + ": out of device memory")
    return RuntimeError(what)


class Index:
    def __init__(self, opt: Opt, rs: ReadSet):
        self.lib = load()
        self.opt = opt
        self.h = self.lib.ndgpu_ovl_index_create(C.byref(opt), len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off),
                                                 _ptr(rs.lens), _ptr(rs.ids))
        if not self.h:
            raise _fail(self.lib, "ndgpu_ovl_index_create failed (the overlap engine needs a HIP device: no CPU path)")

    def close(self):
        if self.h:
            self.lib.ndgpu_ovl_index_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def stat(self):
        n = np.zeros(3, dtype=np.uint64)
        self.lib.ndgpu_ovl_index_stat(self.h, _ptr(n))
        return dict(minimizers=int(n[0]), keys=int(n[1]), reads=int(n[2]))

    def mid_occ(self, frac=None) -> int:
        f = self.opt.mid_occ_frac if frac is None else frac
        m = int(self.lib.ndgpu_ovl_index_mid_occ(self.h, np.float32(f)))
        if m < 0:  # (a threshold of -1 would silently filter every minimizer: no overlaps at all)
            raise _fail(self.lib, "ndgpu_ovl_index_mid_occ failed (%d)" % m)
        return m

    def dump(self):
        s = self.stat()
        key = np.zeros(s["keys"], dtype=np.uint64)
        start = np.zeros(s["keys"] + 1, dtype=np.uint64)
        pos = np.zeros(s["minimizers"], dtype=np.uint64)
        self.lib.ndgpu_ovl_index_dump(self.h, _ptr(key), _ptr(start), _ptr(pos))
        return key, start, pos

    def map(self, rs: ReadSet, mid_occ: int, opt: Opt | None = None) -> np.ndarray:
        opt = opt or self.opt
        recs = C.c_void_p()
        n = self.lib.ndgpu_ovl_map(self.h, C.byref(opt), mid_occ, len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off),
                                   _ptr(rs.lens), _ptr(rs.ids), C.byref(recs))
        if n < 0:
            raise _fail(self.lib, "ndgpu_ovl_map failed (%d)" % n)
        return _take(self.lib, recs, n, REC)

    def map2(self, rs: ReadSet, mid_occ: int, opt: Opt | None = None) -> np.ndarray:
        """--step 2 (--mode 0): the 10-field records that passed the mapper's own filters, in output order; the dovetail /
        contained filter and the encoder follow on the host (Step2Filter)."""
        opt = opt or self.opt
        recs = C.c_void_p()
        n = self.lib.ndgpu_ovl_map2(self.h, C.byref(opt), mid_occ, len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off),
                                    _ptr(rs.lens), _ptr(rs.ids), C.byref(recs))
        if n < 0:
            raise _fail(self.lib, "ndgpu_ovl_map2 failed (%d)" % n)
        return _take(self.lib, recs, n, REC10)

    def map_regs(self, rs: ReadSet, mid_occ: int, want_off=None, want=None, nameless=False, opt: Opt | None = None):
        """The hits of every read of `rs`, nothing judged (ndgpu_ovl_map_regs) -> (records, hits per read)."""
        opt = opt or self.opt
        recs, cnt = C.c_void_p(), C.c_void_p()
        wo = None if want_off is None else np.ascontiguousarray(want_off, dtype=np.uint64)
        wa = None if want is None else np.ascontiguousarray(want, dtype=np.uint32)
        n = self.lib.ndgpu_ovl_map_regs(self.h, C.byref(opt), mid_occ, len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off), _ptr(rs.lens),
                                        _ptr(rs.ids), None if wo is None else _ptr(wo), None if wa is None else _ptr(wa), 1 if nameless else 0,
                                        C.byref(recs), C.byref(cnt), None)
        if n < 0:
            raise _fail(self.lib, "ndgpu_ovl_map_regs failed (%d)" % n)
        return _take(self.lib, recs, n, REC), _take(self.lib, cnt, len(rs), np.uint32)

    def map2_realign(self, target: ReadSet, rs: ReadSet, mid_occ: int, q_mini: "Index", t_mini: "Index", cn: int = 20, opt: Opt | None = None) -> np.ndarray:
        """--step 2 as nextDenovo runs it (--mode 2: marked candidates mapped again with the short k-mer sketch,
        ndgpu_ovl_map2_realign): the 10-field records that passed the mapper's own filters, in output order."""
        opt = opt or self.opt
        recs = C.c_void_p()
        n = self.lib.ndgpu_ovl_map2_realign(self.h, q_mini.h, t_mini.h, C.byref(opt), mid_occ, cn, len(target), _ptr(target.words), target.words.size,
                                            _ptr(target.word_off), _ptr(target.lens), _ptr(target.ids), len(rs), _ptr(rs.words), rs.words.size,
                                            _ptr(rs.word_off), _ptr(rs.lens), _ptr(rs.ids), C.byref(recs))
        if n < 0:
            raise _fail(self.lib, "ndgpu_ovl_map2_realign failed (%d)" % n)
        return _take(self.lib, recs, n, REC10)

    def map_chains(self, rs: ReadSet, mid_occ: int, opt: Opt | None = None):
        """The chains of every read as -c's base-level alignment takes them (ndgpu_ovl_map_chains) ->
        (chains, chains per read, anchor x, anchor y, anchor offsets per read)."""
        opt = opt or self.opt
        ch, cnt, ax, ay, off = (C.c_void_p() for _ in range(5))
        n = self.lib.ndgpu_ovl_map_chains(self.h, C.byref(opt), mid_occ, len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off), _ptr(rs.lens),
                                          _ptr(rs.ids), C.byref(ch), C.byref(cnt), C.byref(ax), C.byref(ay), C.byref(off))
        if n < 0:
            raise _fail(self.lib, "ndgpu_ovl_map_chains failed (%d)" % n)
        a_off = _take(self.lib, off, len(rs) + 1, np.uint64)
        na = int(a_off[-1])
        return _take(self.lib, ch, n, REC), _take(self.lib, cnt, len(rs), np.uint32), _take(self.lib, ax, na, np.uint64), _take(self.lib, ay, na, np.uint64), a_off

    def map_cigar(self, target: ReadSet, rs: ReadSet, mid_occ: int, aopt: AlnOpt | None = None, opt: Opt | None = None, want_stats: bool = False):
        """`--step 1 -c` (ndgpu_ovl_map_cigar): the step-1 records after base-level alignment through the chains; `target` = the
        reads this index was built from."""
        opt = opt or self.opt
        aopt = aopt or aln_opt()
        recs, st = C.c_void_p(), CigarStats()
        n = self.lib.ndgpu_ovl_map_cigar(self.h, C.byref(opt), C.byref(aopt), mid_occ, len(rs), _ptr(rs.words), rs.words.size, _ptr(rs.word_off),
                                         _ptr(rs.lens), _ptr(rs.ids), _ptr(target.words), _ptr(target.word_off), _ptr(target.lens), _ptr(target.ids),
                                         C.byref(recs), C.byref(st))
        if n < 0:
            raise _fail(self.lib, "ndgpu_ovl_map_cigar failed (%d)" % n)
        out = _take(self.lib, recs, n, REC)
        return (out, {k: int(getattr(st, k)) for k, _ in CigarStats._fields_}) if want_stats else out

    def debug_anchors(self, q: int):
        ax, ay, f, p = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = self.lib.ndgpu_ovl_debug_anchors(self.h, q, C.byref(ax), C.byref(ay), C.byref(f), C.byref(p))
        if n < 0:
            raise IndexError(q)
        return (_take(self.lib, ax, n, np.uint64), _take(self.lib, ay, n, np.uint64), _take(self.lib, f, n, np.int32),
                _take(self.lib, p, n, np.int32))

    def stats(self) -> dict:
        st = Stats()
        self.lib.ndgpu_ovl_get_stats(self.h, C.byref(st))
        return {n: getattr(st, n) for n, _ in Stats._fields_}

    def reset_stats(self):
        self.lib.ndgpu_ovl_reset_stats(self.h)


def pool_bytes():
    """(in use, cached, peak in use) device bytes of the overlap library's block pool."""
    lib = load()
    out = (C.c_uint64 * 3)()
    lib.ndgpu_ovl_pool_bytes.argtypes = [C.c_void_p]
    lib.ndgpu_ovl_pool_bytes.restype = None
    lib.ndgpu_ovl_pool_bytes(out)
    return int(out[0]), int(out[1]), int(out[2])


def pool_calls(reset: bool = False):
    """(hipMalloc / hipFree calls the overlap library's block pool made, seconds they took) since the last reset."""
    lib = load()
    out = (C.c_uint64 * 2)()
    lib.ndgpu_ovl_pool_calls.argtypes = [C.c_void_p, C.c_int]
    lib.ndgpu_ovl_pool_calls.restype = None
    lib.ndgpu_ovl_pool_calls(out, 1 if reset else 0)
    return int(out[0]), out[1] * 1e-9


def words_resident(words: np.ndarray) -> None:
    """Upload a read set's 2-bit words once (ndgpu_ovl_words_resident): later calls that name words inside this array use the device
    copy.  The array must stay alive and unchanged until words_release(words)."""
    lib = load()
    lib.ndgpu_ovl_words_resident.argtypes = [C.c_void_p, C.c_uint64]
    lib.ndgpu_ovl_words_resident.restype = C.c_int
    if lib.ndgpu_ovl_words_resident(_ptr(words), words.size) != 0:
        raise _fail(lib, "ndgpu_ovl_words_resident failed")


def words_release(words: np.ndarray) -> None:
    lib = load()
    lib.ndgpu_ovl_words_release.argtypes = [C.c_void_p]
    lib.ndgpu_ovl_words_release.restype = None
    lib.ndgpu_ovl_words_release(_ptr(words))


def trim() -> int:
    """Release the device blocks the overlap library keeps cached between calls (ndgpu_ovl_trim)."""
    lib = load()
    lib.ndgpu_ovl_trim.restype = C.c_uint64
    return int(lib.ndgpu_ovl_trim())


def assemble_piles(srt: np.ndarray, n_ids: int, min_len_seed: int, min_len_aln: int = 500, max_cov_aln: int = 130, min_cov_seed: int = 10,
                   skip=()):
    """lib/nextcorrect.py:92-143 on sorted.ovl records (ndgpu_assemble_piles, host logic of libndgpu_overlap.so).
    Returns (recs uint32[n,8] in nextcorrect's field order, pile_off uint64[p+1], seeds uint32[p])."""
    lib = load()
    if not hasattr(lib.ndgpu_assemble_piles, "_bound"):
        P = C.c_void_p
        lib.ndgpu_assemble_piles.argtypes = [P, C.c_int64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P, C.c_int64,
                                             C.POINTER(P), C.POINTER(P), C.POINTER(P), C.POINTER(C.c_int64)]
        lib.ndgpu_assemble_piles.restype = C.c_int64
        lib.ndgpu_assemble_piles._bound = True
    srt = np.ascontiguousarray(srt, dtype=REC)
    sk = np.ascontiguousarray(np.fromiter(skip, dtype=np.uint32, count=len(skip)) if len(skip) else np.zeros(0, dtype=np.uint32))
    r8, off, seeds = C.c_void_p(), C.c_void_p(), C.c_void_p()
    npiles = C.c_int64(0)
    n = lib.ndgpu_assemble_piles(_ptr(srt), srt.size, int(n_ids), int(min_len_seed), int(min_len_aln), int(max_cov_aln), int(min_cov_seed),
                                 _ptr(sk), sk.size, C.byref(r8), C.byref(off), C.byref(seeds), C.byref(npiles))
    recs = _take(lib, r8, n * 8, np.uint32).reshape(-1, 8)
    return recs, _take(lib, off, npiles.value + 1, np.uint64), _take(lib, seeds, npiles.value, np.uint32)


def pack_2bit(ascii_buf: np.ndarray, ascii_off: np.ndarray, lens: np.ndarray):
    """seq2bit (lib/bseq.c:114-139) of a batch of reads on the device -> (words uint32, word_off uint64[n])."""
    lib = load()
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    ascii_off = np.ascontiguousarray(ascii_off, dtype=np.uint64)
    ascii_buf = np.ascontiguousarray(ascii_buf, dtype=np.uint8)
    nw = (lens.astype(np.uint64) + np.uint64(15)) // np.uint64(16)
    word_off = np.zeros(lens.size, dtype=np.uint64)
    if lens.size:
        word_off[1:] = np.cumsum(nw)[:-1]
    words = np.zeros(int(nw.sum()) + 1, dtype=np.uint32)
    n = lib.ndgpu_pack_2bit(lens.size, _ptr(ascii_buf), ascii_buf.size, _ptr(ascii_off), _ptr(lens), _ptr(word_off), _ptr(words))
    if n < 0:
        raise RuntimeError("ndgpu_pack_2bit failed (%d): no usable HIP device?" % n)
    return words[:n], word_off


class Step2Filter:
    """filter_ovl / encode_ovl_i / out_bl of one `minimap2-nd --step 2` run (lib/ovl.c:449-563, 205-253, 339-362): the state
    lives from the first record to the `.bl` table."""

    def __init__(self):
        self.lib = load()
        self.h = self.lib.ndgpu_s2_new()
        if not self.h:
            raise MemoryError("ndgpu_s2_new")
        self.prev = np.zeros(2, dtype=np.uint32)

    def feed(self, recs: np.ndarray, maxhan1: int, maxhan2: int, want_verdicts: bool = False):
        """Records of one map2 call -> bytes of the kept ones (and their verdicts)."""
        recs = np.ascontiguousarray(recs, dtype=REC10)
        out = C.c_void_p()
        kept = np.zeros(max(1, recs.size), dtype=np.uint8)
        n = self.lib.ndgpu_s2_filter_encode(self.h, _ptr(recs), recs.size, maxhan1, maxhan2, _ptr(self.prev), C.byref(out), _ptr(kept))
        if n < 0:
            raise RuntimeError("ndgpu_s2_filter_encode failed (%d)" % n)
        b = _take(self.lib, out, n, np.uint8).tobytes()
        return (b, kept[:recs.size].astype(bool)) if want_verdicts else b

    def bl(self) -> str:
        text = C.c_void_p()
        n = self.lib.ndgpu_s2_bl(self.h, C.byref(text))
        if n < 0:
            raise RuntimeError("ndgpu_s2_bl failed")
        return _take(self.lib, text, n, np.uint8).tobytes().decode()

    def close(self):
        if self.h:
            self.lib.ndgpu_s2_free(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def encode(recs: np.ndarray, prev: np.ndarray) -> bytes:
    """encode_ovl over a record array; prev = uint32[2] running state, updated in place."""
    lib = load()
    recs = np.ascontiguousarray(recs)
    out = np.zeros(40 * max(1, recs.size), dtype=np.uint8)
    n = lib.ndgpu_ovl_encode(_ptr(recs), recs.size, _ptr(prev), _ptr(out))
    return out[:n].tobytes()


class SortStats(C.Structure):
    _fields_ = [("gpu_ms", C.c_double)] + [(n, C.c_uint64) for n in ("raw_records", "candidates", "seeds", "kept", "ranges")]


def from_decoded(a: np.ndarray) -> np.ndarray:
    """[n,8] array in decode_ovl order (qname, rev, qs, qe, tname, ts, te, match) -> record array."""
    r = np.zeros(a.shape[0], dtype=REC)
    for i, n in enumerate(("qname", "rev", "qs", "qe", "tname", "ts", "te", "match")):
        r[n] = a[:, i]
    return r


def sort_overlaps(files, seed_len: np.ndarray, min_seed_len: int, max_bin_cov: int = 40, max_flank_len: int = 300, hq: bool = False):
    """The `ovl_sort` step on the device (ndgpu_ovl_sort).  files = list of record arrays (step-1 overlaps, one per
    input file, fofn order).  Returns (sorted records, [(seed id, 'c'|'k'), ...], stats dict)."""
    import time
    t_in = time.perf_counter()
    lib = load()
    if not hasattr(lib.ndgpu_ovl_sort, "_bound"):
        for fn in (lib.ndgpu_ovl_sort, lib.ndgpu_ovl_sort_hq):
            fn.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int32, C.c_void_p, C.c_uint32, C.c_int32,
                           C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                           C.POINTER(C.c_int64), C.POINTER(SortStats)]
            fn.restype = C.c_int64
        lib.ndgpu_ovl_sort._bound = True
    files = [np.ascontiguousarray(f, dtype=REC) for f in files]
    nf = len(files)
    ptrs = (C.c_void_p * max(1, nf))(*[f.ctypes.data for f in files])
    cnts = (C.c_int64 * max(1, nf))(*[f.size for f in files])
    seed_len = np.ascontiguousarray(seed_len, dtype=np.uint32)
    out, bid, bkind = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nbl = C.c_int64(0)
    st = SortStats()
    t_call = time.perf_counter()
    n = (lib.ndgpu_ovl_sort_hq if hq else lib.ndgpu_ovl_sort)(ptrs, cnts, nf, _ptr(seed_len), seed_len.size, int(min_seed_len), int(max_bin_cov), int(max_flank_len),
                           C.byref(out), C.byref(bid), C.byref(bkind), C.byref(nbl), C.byref(st))
    if n == -2:   # a device operation failed: MemoryError if it was memory (like every other entry point), RuntimeError otherwise
        raise _fail(lib, "ndgpu_ovl_sort failed (about 150 bytes of device memory per candidate overlap are needed; use more seed files: "
                         "seed_cutfiles)")
    if n < 0:
        raise RuntimeError({-1: "ndgpu_ovl_sort: no usable HIP device", -2: "ndgpu_ovl_sort: out of device memory (about 150 bytes per candidate "
                            "overlap are needed; use more seed files: seed_cutfiles)", -3: "ndgpu_ovl_sort: more than 2^31 candidate overlaps in "
                            "one call (use more seed files: seed_cutfiles)"}.get(int(n), "ndgpu_ovl_sort failed (%d)" % n))
    t_back = time.perf_counter()
    recs = _take(lib, out, n, REC)
    ids = _take(lib, bid, nbl.value, np.uint32)
    kinds = _take(lib, bkind, nbl.value, np.uint8)
    bl = list(zip(ids.tolist(), [chr(k) for k in kinds.tolist()]))
    if os.environ.get("NDGPU_PROF"):
        t = time.perf_counter()
        print("[sort_overlaps] before the call %.2f ms, ndgpu_ovl_sort %.2f ms, after %.2f ms" % ((t_call - t_in) * 1e3, (t_back - t_call) * 1e3, (t - t_back) * 1e3),
              file=sys.stderr)
    return recs, bl, {n_: getattr(st, n_) for n_, _ in SortStats._fields_}

