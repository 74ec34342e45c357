"""The CPU oracle (oracle/*.c) against the reference: committed golden vectors always,
and the compiled reference itself (oracle/_ref) wherever it is present."""
import ctypes as C
import os

import numpy as np
import pytest

import refpipe
import util


def test_oracle_align_matches_golden(oracle_lib):
    pairs = util.load_pairs()
    assert len(pairs) >= 60
    kinds = set()
    for i, p in enumerate(pairs):
        o, ts, qs, ops = util.oracle_align(oracle_lib, p["q"], p["t"], p["hq"])
        assert o.aln_len == p["aln_len"], i
        kinds.add(min(p["aln_len"], 3))
        if p["aln_len"] > 2:
            assert o.status == 1
            assert np.array_equal(ops, p["ops"]), i
            assert o.t_used == p["t_used"] and o.q_used == p["q_used"], i
            assert np.array_equal(util.strings_to_ops(ts, qs), p["ops"]), i
            # the two rows spell the inputs back
            assert ts.replace(b"-", b"") == p["t"][:o.t_used] and qs.replace(b"-", b"") == p["q"][:o.q_used]
        elif p["aln_len"] == 2:
            assert o.status == 2
        else:
            assert o.status == 0
    assert kinds == {0, 2, 3}, "golden set must cover failed / gap-abort / aligned"


@pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so"), reason="compiled reference not present")
def test_oracle_align_fuzz_vs_reference(oracle_lib):
    """Differential fuzz of the restatement against the reference's exported align/align_hq."""
    import sys
    sys.path.insert(0, util.GOLD)
    import make_golden as mg
    from nextdenovo_amd import synth
    lib = refpipe.ref_cns()
    lib.malloc_vd.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_void_p), C.c_uint64]
    lib.clean_V.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.destory_vd.argtypes = [C.POINTER(C.c_int), C.c_void_p]
    for f in (lib.align, lib.align_hq):
        f.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(mg.Aln), C.POINTER(C.c_int), C.c_void_p]
        f.restype = None
    rng = np.random.default_rng(99)
    n_ok = 0
    for it in range(300):
        L = int(rng.integers(1, 3000))
        base = rng.integers(0, 4 if it % 7 else 2, L, dtype=np.uint8)
        prof = ("ont", "clr", "hifi")[it % 3]
        q = synth.mutate(base, np.random.default_rng(2 * it), prof)[0]
        t = synth.mutate(base, np.random.default_rng(2 * it + 1), prof)[0]
        if it % 11 == 0:
            t = t[: max(0, t.size - int(rng.integers(0, 200)))]
        if it % 13 == 0:
            q = np.concatenate([q[: q.size // 2], rng.integers(0, 4, int(rng.integers(1, 400)), dtype=np.uint8),
                                q[q.size // 2:]])
        hq = int(it % 5 == 0)
        qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
        n, tu, qu, ts, qs = mg.ref_align(lib, qa, ta, hq)
        o, ots, oqs, ops = util.oracle_align(oracle_lib, qa, ta, hq)
        assert o.aln_len == n, it
        if n > 2:
            assert ots == ts and oqs == qs, it
            assert (o.t_used, o.q_used) == (tu, qu), it
            n_ok += 1
    assert n_ok > 150


def test_oracle_shift(oracle_lib):
    # get_align_shift(aln, 8): first/last run of 8 exact columns (lib/nextcorrect.c:102-154)
    ops = np.asarray([1, 0, 0, 2, 0] + [0] * 9 + [1, 2] + [0] * 8 + [2, 0, 0], dtype=np.uint8)
    ts, te, sh = C.c_uint(100), C.c_uint(100 + int((ops != 1).sum()) - 1), C.c_int(0)
    n = oracle_lib.nd_oracle_shift(ops.ctypes.data_as(C.POINTER(C.c_uint8)), ops.size, 8, C.byref(ts), C.byref(te),
                                   C.byref(sh))
    assert sh.value == 4 and n == 20
    assert ts.value == 100 + 3  # columns 0..3 hold three target bases
    assert te.value == 100 + int((ops != 1).sum()) - 1 - 3
    bad = np.asarray([0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 2], dtype=np.uint8)
    ts2, te2 = C.c_uint(0), C.c_uint(9)
    assert oracle_lib.nd_oracle_shift(bad.ctypes.data_as(C.POINTER(C.c_uint8)), bad.size, 8, C.byref(ts2),
                                      C.byref(te2), C.byref(sh)) == 0
