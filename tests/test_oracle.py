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


@pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so"), reason="compiled reference not present")
def test_prefix_and_extension_variants_vs_reference(oracle_lib):
    """oracle/ond_ext_oracle.c against the reference's exported `ide`, `alnpos`, `extend_fwd`, `extend_rev`
    (lib/align.h:51-58) with the call patterns of the HiFi mode-3 overlap path (minimap2/map.c:385-482, 941-956):
    extensions of unaligned read ends (d_factor 0.1, band 500) and trimmed alignments of homopolymer-compressed blocks."""
    from nextdenovo_amd import synth
    lib = refpipe.ref_cns()
    P = C.c_void_p
    lib.malloc_vd.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(P), C.c_uint64]
    lib.clean_V.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.destory_vd.argtypes = [C.POINTER(C.c_int), P]
    lib.ide.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.alnpos.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), P, C.c_int, C.c_int, C.POINTER(C.c_uint * 6)]
    for f in (lib.extend_fwd, lib.extend_rev):
        f.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), P, C.c_int, C.c_int, C.c_float,
                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for f in (lib.ide, lib.alnpos, lib.extend_fwd, lib.extend_rev):
        f.restype = None
    o = oracle_lib
    o.nd_oracle_ide.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    o.nd_oracle_alnpos.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint * 6)]
    o.nd_oracle_extend.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]
    mem_d = 6000                                    # opt->ide_ml (minimap2/options.c:60)
    V = C.POINTER(C.c_int)()
    D = P()
    lib.malloc_vd(C.byref(V), C.byref(D), mem_d)
    assert D.value
    rng = np.random.default_rng(123)
    seen = {"ide": 0, "alnpos": 0, "ext_peak": 0, "ext_stop": 0, "ext_end": 0, "unaligned": 0}
    try:
        for it in range(400):
            L = int(rng.integers(10, 4000))
            base = rng.integers(0, 4 if it % 9 else 2, L, dtype=np.uint8)
            prof = ("hifi", "hifi", "ont", "clr")[it % 4]
            q = synth.mutate(base, np.random.default_rng(3 * it), prof)[0]
            t = synth.mutate(base, np.random.default_rng(3 * it + 1), prof)[0]
            if it % 5 == 0:                          # the sequences diverge after a common prefix: the peak ends the extension
                cut = int(rng.integers(5, max(6, L // 2)))
                t = np.concatenate([t[:cut], rng.integers(0, 4, int(rng.integers(50, 1500)), dtype=np.uint8)])
            if it % 7 == 0:
                q = q[: max(1, q.size - int(rng.integers(0, 300)))]
            qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
            ql, tl = len(qa), len(ta)
            minlen = min(ql, tl)
            # nd_extend_ends' budgets (map.c:404-406): max_d = minlen/4 capped at ide_ml (minlen itself when <= 20), band 500
            max_d = min(mem_d, minlen // 4 if minlen > 20 else minlen)
            for rev, fn in ((0, lib.extend_fwd), (1, lib.extend_rev)):
                lib.clean_V(V, mem_d)
                bx, by, ox, oy = C.c_int(-7), C.c_int(-7), C.c_int(-9), C.c_int(-9)
                fn(qa, ql, ta, tl, V, D, max_d, 500, 0.1, C.byref(bx), C.byref(by))
                o.nd_oracle_extend(qa, ql, ta, tl, max_d, 500, 0.1, rev, C.byref(ox), C.byref(oy))
                assert (bx.value, by.value) == (ox.value, oy.value), (it, rev)
                if bx.value >= ql or by.value >= tl:
                    seen["ext_end"] += 1
                elif bx.value or by.value:
                    seen["ext_peak"] += 1
                else:
                    seen["ext_stop"] += 1
            # the trimmed re-alignment of map.c:941-956: max_d = alnlen/5 capped, band = max_d > 1500 ? 500 : max_d/3
            alnlen = max(ql, tl)
            md = min(mem_d, alnlen // 5)
            band = 500 if md > 1500 else md // 3
            lib.clean_V(V, mem_d)
            ra, oa = (C.c_uint * 6)(*([4242] * 6)), (C.c_uint * 6)(*([4242] * 6))
            lib.alnpos(qa, ql, ta, tl, V, D, md, band, C.byref(ra))
            o.nd_oracle_alnpos(qa, ql, ta, tl, md, band, C.byref(oa))
            assert list(ra) == list(oa), it
            seen["alnpos" if ra[0] != 4242 else "unaligned"] += 1
            lib.clean_V(V, mem_d)
            m1, b1, m2, b2 = C.c_int(-1), C.c_int(-1), C.c_int(-1), C.c_int(-1)
            lib.ide(qa, ql, ta, tl, V, D, md, band, C.byref(m1), C.byref(b1))
            o.nd_oracle_ide(qa, ql, ta, tl, md, band, C.byref(m2), C.byref(b2))
            assert (m1.value, b1.value) == (m2.value, b2.value), it
            seen["ide"] += m1.value >= 0
    finally:
        lib.destory_vd(V, D)
    assert seen["ide"] > 150 and seen["alnpos"] > 150 and seen["ext_end"] > 100 and seen["ext_peak"] > 50 and seen["unaligned"] > 5, seen


@pytest.mark.skipif(not refpipe.have_ref("ovlseq.so"), reason="compiled reference not present")
@pytest.mark.parametrize("seed,n_reads,n_ovl,han1,han2", [(1, 40, 1500, 500, 50), (2, 300, 6000, 5000, 500), (3, 9, 400, 200, 20), (4, 1200, 9000, 800, 100)])
def test_step2_filter_and_bl_vs_reference(oracle_lib, tmp_path, seed, n_reads, n_ovl, han1, han2):
    """oracle/step2_oracle.c against the reference's exported `filter_ovl` and `out_bl` (lib/ovl.c:449-563, 339-362): every
    verdict of a random stream of dovetail / contained / internal overlaps, and the `.bl` table at the end byte for byte
    (the order of its lines is the reference hash table's bucket order)."""
    ref = C.CDLL(os.path.join(refpipe.REFDIR, "ovlseq.so"))
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    class OvlI(C.Structure):       # overlap_i, lib/ovl.h:27-32
        _fields_ = [("rev", C.c_uint8), ("qname", C.c_uint32), ("qs", C.c_uint32), ("qe", C.c_uint32), ("qlen", C.c_uint32),
                    ("tname", C.c_uint32), ("ts", C.c_uint32), ("te", C.c_uint32), ("tlen", C.c_uint32), ("identity", C.c_uint32)]

    class S2(C.Structure):
        _fields_ = [(n, C.c_uint32) for n in ("rev", "qname", "qs", "qe", "qlen", "tname", "ts", "te", "tlen", "identity")]

    ref.filter_ovl.argtypes = [C.POINTER(OvlI), C.c_void_p, C.c_int32, C.c_int32]
    ref.filter_ovl.restype = C.c_int
    ref.out_bl.argtypes = [C.c_void_p, C.c_void_p]
    o = oracle_lib
    o.nd_s2_new.restype = C.c_void_p
    o.nd_s2_free.argtypes = [C.c_void_p]
    o.nd_s2_filter.argtypes = [C.c_void_p, C.POINTER(S2), C.c_int32, C.c_int32]
    o.nd_s2_out_bl.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    o.nd_s2_out_bl.restype = C.c_int64
    rng = np.random.default_rng(seed)
    names = rng.permutation(50000)[:n_reads] + 1            # scattered ids: the table's probing and growth are exercised
    lens = rng.integers(max(3 * han1 // 2, 300), 8 * han1, n_reads)
    table = (C.c_uint8 * 64)()                              # a zeroed khash_t is an empty table (kh_init = calloc)
    st = o.nd_s2_new()
    kept = 0
    try:
        for it in range(n_ovl):
            a, b = rng.choice(n_reads, 2, replace=False)
            ql, tl = int(lens[a]), int(lens[b])
            kind = it % 6

            def piece(L, where):
                if where == "lo":
                    s = int(rng.integers(0, han1 + han1 // 2)); e = int(min(L, s + rng.integers(han1 // 2, L)))
                elif where == "hi":
                    e = L - int(rng.integers(0, han1 + han1 // 2)); s = int(max(0, e - rng.integers(han1 // 2, L)))
                elif where == "all":
                    s = int(rng.integers(0, han2 * 2)); e = L - int(rng.integers(0, han2 * 2))
                else:
                    s = int(rng.integers(0, L // 2)); e = int(min(L, s + rng.integers(han2 + 21, L // 2 + han2 + 22)))
                if e - s < 21:
                    s, e = 0, min(L, 40)
                return s, e
            rev = int(rng.integers(0, 2))
            if kind == 0:
                (qs, qe), (ts, te) = (piece(ql, "lo"), piece(tl, "lo")) if rev else (piece(ql, "hi"), piece(tl, "lo"))
            elif kind == 1:
                (qs, qe), (ts, te) = (piece(ql, "hi"), piece(tl, "hi")) if rev else (piece(ql, "lo"), piece(tl, "hi"))
            elif kind == 2:
                (qs, qe), (ts, te) = piece(ql, "all"), piece(tl, "mid")
            elif kind == 3:
                (qs, qe), (ts, te) = piece(ql, "mid"), piece(tl, "all")
            else:
                (qs, qe), (ts, te) = piece(ql, "mid"), piece(tl, "mid")
            ide = int(rng.integers(500, 10001))
            r1 = OvlI(rev, int(names[a]), qs, qe, ql, int(names[b]), ts, te, tl, ide)
            r2 = S2(rev, int(names[a]), qs, qe, ql, int(names[b]), ts, te, tl, ide)
            v1 = ref.filter_ovl(C.byref(r1), table, han1, han2)
            v2 = o.nd_s2_filter(st, C.byref(r2), han1, han2)
            assert v1 == v2, it
            kept += v1
        path = str(tmp_path / "ref.bl").encode()
        fp = libc.fopen(path, b"w")
        ref.out_bl(table, fp)
        libc.fclose(fp)
        buf = C.create_string_buffer(64 * n_reads + 40 * n_ovl + 1024)
        n = o.nd_s2_out_bl(st, buf, len(buf))
        want = open(path, "rb").read()
        assert n > 0 and buf.raw[:n] == want
        assert 0 < kept < n_ovl and 0 < want.count(b"\n") <= n_reads
        assert b"\t2\n" in want                  # reads dropped as contained (two containing overlaps) are listed without their state
    finally:
        o.nd_s2_free(st)
