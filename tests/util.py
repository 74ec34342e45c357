"""Shared helpers for the tests: golden fixtures + ctypes views."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ASC = np.frombuffer(b"ACGT", dtype=np.uint8)


class OAln(C.Structure):
    _fields_ = [("status", C.c_int), ("aln_len", C.c_int), ("t_used", C.c_int), ("q_used", C.c_int),
                ("d_final", C.c_int), ("k_final", C.c_int), ("cells", C.c_long), ("d_steps", C.c_long),
                ("max_band", C.c_int)]


class Aln(C.Structure):
    _fields_ = [("shift", C.c_uint), ("aln_len", C.c_uint), ("aln_t_s", C.c_uint), ("aln_t_e", C.c_uint),
                ("aln_t_len", C.c_uint), ("aln_q_len", C.c_uint), ("q_aln_str", C.c_char_p),
                ("t_aln_str", C.c_char_p)]


import sys
sys.path.insert(0, os.path.dirname(HERE))
from nextdenovo_amd.api import ConsensusTrimed as CT  # noqa: E402  (one ctypes type for every binding)


def oracle_align(lib, q: bytes, t: bytes, hq=0):
    cap = len(q) + len(t) + 2
    o = OAln()
    ts, qs = C.create_string_buffer(cap), C.create_string_buffer(cap)
    ops = (C.c_uint8 * cap)()
    lib.nd_oracle_align(q, len(q), t, len(t), hq, C.byref(o), ts, qs, ops)
    return o, ts.raw[:o.aln_len], qs.raw[:o.aln_len], np.frombuffer(ops, dtype=np.uint8)[:o.aln_len].copy()


def strings_to_ops(ts: bytes, qs: bytes) -> np.ndarray:
    t = np.frombuffer(ts, dtype=np.uint8)
    q = np.frombuffer(qs, dtype=np.uint8)
    ops = np.zeros(t.size, dtype=np.uint8)
    ops[t == ord("-")] = 1
    ops[q == ord("-")] = 2
    return ops


def gpu_align(lib, q: bytes, t: bytes, hq=0):
    """Through the shipped C ABI (include/ndgpu_nextcorrect.h: align / align_hq)."""
    cap = len(q) + len(t) + 2
    a = Aln()
    tb, qb = C.create_string_buffer(cap), C.create_string_buffer(cap)
    a.t_aln_str = C.cast(tb, C.c_char_p)
    a.q_aln_str = C.cast(qb, C.c_char_p)
    a.aln_t_s = 0
    a.aln_len = 0
    f = lib.align_hq if hq else lib.align
    f.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Aln), C.c_void_p, C.c_void_p]
    f.restype = None
    f(q, len(q), t, len(t), C.byref(a), None, None)
    n = a.aln_len
    return n, a.aln_t_len, a.aln_q_len, tb.raw[:n], qb.raw[:n]


def load_pairs():
    d = np.load(os.path.join(GOLD, "align_pairs.npz"))
    out = []
    for i in range(d["hq"].size):
        q = ASC[d["q"][d["q_off"][i]:d["q_off"][i + 1]]].tobytes()
        t = ASC[d["t"][d["t_off"][i]:d["t_off"][i + 1]]].tobytes()
        ops = d["ops"][d["ops_off"][i]:d["ops_off"][i + 1]]
        out.append(dict(q=q, t=t, hq=int(d["hq"][i]), aln_len=int(d["aln_len"][i]), t_used=int(d["t_used"][i]),
                        q_used=int(d["q_used"][i]), ops=ops))
    return out


def unpack2(b: np.ndarray, n: int) -> np.ndarray:
    c = np.stack([b & 3, (b >> 2) & 3, (b >> 4) & 3, (b >> 6) & 3], axis=1).reshape(-1)
    return c[:n].astype(np.uint8)


def load_piles():
    d = np.load(os.path.join(GOLD, "piles.npz"))
    piles = []
    for p in range(d["pile_off"].size - 1):
        a, b = int(d["pile_off"][p]), int(d["pile_off"][p + 1])
        seqs = []
        for r in range(a, b):
            packed = d["codes"][d["codes_off"][r]:d["codes_off"][r + 1]]
            seqs.append(ASC[unpack2(packed, int(d["lens"][r]))].tobytes())
        piles.append(dict(seqs=seqs, aln_start=[int(x) for x in d["aln_start"][a:b]],
                          aln_end=[int(x) for x in d["aln_end"][a:b]], max_aln=int(d["max_aln"][p]),
                          max_lq=int(d["max_lq"][p]), read_type=int(d["read_type"][p]), fast=int(d["fast"][p]),
                          split=int(d["split"][p]), exp_len=int(d["exp_len"][p]), exp_ide=float(d["exp_ide"][p]),
                          exp_seq=d["exp_seq"][d["exp_seq_off"][p]:d["exp_seq_off"][p + 1]].tobytes()))
    return piles


def load_poa():
    d = np.load(os.path.join(GOLD, "poa.npz"))
    cases, k = [], 0
    for i, n in enumerate(d["count"]):
        seqs = []
        for _ in range(int(n)):
            seqs.append(ASC[d["seq"][d["seq_off"][k]:d["seq_off"][k + 1]]].tobytes())
            k += 1
        cases.append((seqs, d["res"][d["res_off"][i]:d["res_off"][i + 1]].tobytes()))
    return cases


def call_correct(fn, free, p, **over):
    """fn has the nextCorrect signature (lib/nextcorrect.h:161-162)."""
    seqs = p["seqs"]
    n = len(seqs)
    cs = (C.c_char_p * n)()
    cs[:] = seqs
    st = (C.c_uint * n)(*p["aln_start"])
    en = (C.c_uint * n)(*p["aln_end"])
    r = fn(cs, st, en, n, p["max_aln"], over.get("min_len_aln", 500), 130, 4, p["max_lq"], 0.8,
           over.get("split", p["split"]), over.get("fast", p["fast"]), p["read_type"])
    ln, ide = r.contents.len, r.contents.identity
    seq = C.string_at(r.contents.seq, ln) if ln > 4 else b""
    free(r)
    return ln, ide, seq


def bind_correct(lib, name="nextCorrect", free="free_consensus_trimed"):
    fn = getattr(lib, name)
    fn.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_uint,
                   C.c_uint, C.c_uint, C.c_uint, C.c_float, C.c_uint, C.c_uint, C.c_int]
    fn.restype = C.POINTER(CT)
    fr = getattr(lib, free)
    fr.argtypes = [C.POINTER(CT)]
    return fn, fr
