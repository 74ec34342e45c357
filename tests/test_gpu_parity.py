"""Parity tests proper: the HIP path (through the shipped C ABI) against the oracle, the
committed reference vectors, and -- where oracle/_ref travelled -- the compiled reference."""
import ctypes as C
import os

import numpy as np
import pytest

import refpipe
import util

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def lib(native_lib):
    assert native_lib.ndgpu_device_count() >= 1, "no HIP device: the HIP path has no fallback"
    return native_lib


def test_native_library_is_in_tree(lib):
    from nextdenovo_amd import api
    assert os.path.dirname(api.lib_path()).endswith("nextdenovo_amd")


def test_align_golden_vectors(lib):
    """Reference align()/align_hq() outputs, bit exact, incl. failures, the > 250-gap marker
    and a live band beyond the LDS fast path."""
    from nextdenovo_amd import api
    api.reset_stats()
    for i, p in enumerate(util.load_pairs()):
        n, tu, qu, ts, qs = util.gpu_align(lib, p["q"], p["t"], p["hq"])
        assert n == p["aln_len"], i
        if n > 2:
            assert np.array_equal(util.strings_to_ops(ts, qs), p["ops"]), i
            assert (tu, qu) == (p["t_used"], p["q_used"]), i
            assert ts.replace(b"-", b"") == p["t"] and qs.replace(b"-", b"") == p["q"], i
    st = api.stats()
    assert st["wide_tasks"] >= 1, "golden set must exercise the wide-band kernel"
    assert st["max_band"] > 253


def test_align_fuzz_vs_oracle(lib, oracle_lib):
    from nextdenovo_amd import synth
    rng = np.random.default_rng(2024)
    ok = 0
    for it in range(250):
        L = int(rng.integers(1, 6000))
        base = rng.integers(0, 4 if it % 9 else 2, L, dtype=np.uint8)
        prof = ("ont", "clr", "hifi")[it % 3]
        q = synth.mutate(base, np.random.default_rng(3 * it), prof)[0]
        t = synth.mutate(base, np.random.default_rng(3 * it + 1), prof)[0]
        if it % 10 == 0:
            q = q[int(rng.integers(0, 40)):]
        if it % 17 == 0:
            t = np.concatenate([t[: t.size // 3], rng.integers(0, 4, int(rng.integers(1, 300)), dtype=np.uint8),
                                t[t.size // 3:]])
        hq = int(it % 4 == 0)
        qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
        o, ots, oqs, _ = util.oracle_align(oracle_lib, qa, ta, hq)
        n, tu, qu, ts, qs = util.gpu_align(lib, qa, ta, hq)
        assert n == o.aln_len, it
        if o.status == 1:
            assert ts == ots and qs == oqs and (tu, qu) == (o.t_used, o.q_used), it
            ok += 1
    assert ok > 120


def test_empty_and_tiny(lib):
    for q, t in ((b"", b""), (b"A", b""), (b"", b"ACGT"), (b"A", b"A"), (b"ACGTACGTAC", b"ACGTACGTAC"), (b"AC", b"GT")):
        n, _, _, ts, qs = util.gpu_align(lib, q, t)
        if q == t and len(q) >= 3:
            assert n == len(q) and ts == qs == q
        else:
            assert n == 0 or (ts.replace(b"-", b"") == t and qs.replace(b"-", b"") == q)


def test_golden_piles_nextcorrect(lib):
    fn, fr = util.bind_correct(lib)
    for i, p in enumerate(util.load_piles()):
        ln, ide, seq = util.call_correct(fn, fr, p)
        assert ln == p["exp_len"], i
        if ln > 4:
            assert seq == p["exp_seq"], i
            assert np.float32(ide) == np.float32(p["exp_ide"]), i


def test_batch_equals_single(lib):
    from nextdenovo_amd import api
    piles = [p for p in util.load_piles() if p["fast"] == 0 and p["split"] == 0 and p["read_type"] == 1]
    res = api.correct_batch([(p["seqs"], p["aln_start"], p["aln_end"], p["max_aln"], p["max_lq"]) for p in piles],
                            read_type=1, host_threads=4)
    for p, r in zip(piles, res):
        assert r[0] == p["exp_len"] and r[2] == p["exp_seq"]


def _synth_set(gsize, mu, sigma, seed, depth=30):
    from nextdenovo_amd import synth
    g = synth.make_genome(gsize, seed=seed, n_repeats=0)
    rs = synth.simulate_reads(g, depth, "ont", seed=seed + 1, mu=mu, sigma=sigma)
    return rs, synth.build_piles(rs, seed_cutoff=1000)


def test_db_path_equals_ascii_path_and_host_oracle(lib, host_harness):
    """Resident-DB batched entry == per-pile ASCII entry == host engine with the oracle
    aligner (CPU), on piles with both strands."""
    from nextdenovo_amd import api, synth
    rs, piles = _synth_set(40000, 8.3, 0.4, 31)
    piles = piles[:12]
    words, off, lens = synth.pack_db(rs)
    db = api.ReadDB(words, off, lens)
    recs, poff = synth.flatten_piles(piles)
    got = db.correct_piles(recs, poff, host_threads=4)
    db.close()
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    assert any(int(p["recs"][:, 1].max()) == 1 for p in piles)
    for p, g in zip(piles, got):
        seqs, st, en, mal = synth.pile_sequences(rs, p)
        mlq = min(en[0] // 2, 10000)
        a = api.correct(seqs, st, en, mal, max_lq_length=mlq)
        c = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=1,
                                           fast=0, split=0))
        assert a[0] == g[0] == c[0]
        assert a[2] == g[2] == c[2]
        assert np.float32(a[1]) == np.float32(g[1]) == np.float32(c[1])


def test_hifi_db_path_equals_ascii_path_and_host_oracle(lib, host_harness):
    """read_type=3 (align_hq + k-mer phasing consensus) through the resident-DB entry."""
    from nextdenovo_amd import api, synth
    g = synth.make_genome(30000, seed=71, n_repeats=0)
    rs = synth.simulate_reads(g, 30, "hifi", seed=72, mu=8.3, sigma=0.3)
    piles = synth.build_piles(rs, seed_cutoff=1000)[:8]
    words, off, lens = synth.pack_db(rs)
    db = api.ReadDB(words, off, lens)
    recs, poff = synth.flatten_piles(piles)
    got = db.correct_piles(recs, poff, read_type=3, max_lq_length=1000, host_threads=4)
    db.close()
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    for p, g_ in zip(piles, got):
        seqs, st, en, mal = synth.pile_sequences(rs, p)
        mlq = min(en[0] // 2, 1000)
        a = api.correct(seqs, st, en, mal, max_lq_length=mlq, read_type=3)
        c = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=3,
                                           fast=0, split=0))
        assert a[0] == g_[0] == c[0] and a[0] > 1000
        assert a[2] == g_[2] == c[2]


def test_deep_piles_and_link_capacity_retry(lib, host_harness):
    """Depth-170 piles (more reads than K9 keeps register-resident, up to the 1.5 x max_cov_aln admission limit) against
    the host engine + oracle, and the same piles through K9's overflow path (small LDS link lists -> full capacity)."""
    import subprocess
    import sys
    from nextdenovo_amd import api, synth
    rs, piles = _synth_set(12000, 8.0, 0.3, 81, depth=170)
    piles = sorted(piles, key=lambda p: -len(p["recs"]))[:3]
    assert len(piles[0]["recs"]) > 140
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    want = []
    for p in piles:
        seqs, st, en, mal = synth.pile_sequences(rs, p)
        mlq = min(en[0] // 2, 10000)
        a = api.correct(seqs, st, en, mal, max_lq_length=mlq)
        c = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=1,
                                           fast=0, split=0))
        assert a[0] == c[0] and a[2] == c[2] and a[0] > 1000
        want.append((a[0], a[2]))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_gpu_parity import _synth_set\nfrom nextdenovo_amd import api, synth\n"
            "rs, piles = _synth_set(12000, 8.0, 0.3, 81, depth=170)\n"
            "piles = sorted(piles, key=lambda p: -len(p['recs']))[:3]\n"
            "for p in piles:\n"
            "    seqs, st, en, mal = synth.pile_sequences(rs, p)\n"
            "    a = api.correct(seqs, st, en, mal, max_lq_length=min(en[0] // 2, 10000))\n"
            "    print(a[0], a[2].decode())\n") % (HERE, os.path.dirname(HERE))
    env = dict(os.environ, NDGPU_K9_FORCE_RETRY="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [ln.split() for ln in r.stdout.strip().splitlines()]
    assert [(int(a), b.encode()) for a, b in got] == want


def test_full_size_properties(lib, host_harness):
    """BASELINE config-2 sized reads (lognormal mu 9.55: 10-60 kb overlaps).  Size-independent
    properties: every alignment's two rows spell its inputs back (round trip), and a
    full-size pile equals the host engine + oracle result."""
    from nextdenovo_amd import api, synth
    rs, piles = _synth_set(150000, 9.55, 0.6, 51, depth=20)
    big = max(piles, key=lambda p: int(p["recs"][0][3]))
    seqs, st, en, mal = synth.pile_sequences(rs, big)
    assert en[0] + 1 > 20000
    for i in range(1, min(6, len(seqs))):
        t = seqs[0][st[i]:en[i] + 1]
        n, tu, qu, ts, qs = util.gpu_align(lib, seqs[i], t)
        assert n > 0
        assert ts.replace(b"-", b"") == t and qs.replace(b"-", b"") == seqs[i]
        cols = np.frombuffer(ts, dtype=np.uint8) != np.frombuffer(qs, dtype=np.uint8)
        assert np.all((np.frombuffer(ts, dtype=np.uint8)[cols] == ord("-")) |
                      (np.frombuffer(qs, dtype=np.uint8)[cols] == ord("-")))  # O(ND): no mismatch columns
    mlq = min(en[0] // 2, 10000)
    a = api.correct(seqs, st, en, mal, max_lq_length=mlq)
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    c = util.call_correct(fn, fr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=mlq, read_type=1,
                                       fast=0, split=0))
    assert a[0] == c[0] and a[2] == c[2] and a[0] > 15000


@pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so", "minimap2-nd", "seq_dump", "ovl_sort", "ovlseq.so"),
                    reason="compiled reference chain did not travel")
def test_live_reference_chain_on_gpu(lib, tmp_path):
    from nextdenovo_amd import api, synth
    rfn, rfr = util.bind_correct(refpipe.ref_cns())
    g = synth.make_genome(60000, seed=61, n_repeats=0)
    rs = synth.simulate_reads(g, 30, "ont", seed=62, mu=8.5, sigma=0.4, min_len=1000)
    fa = str(tmp_path / "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    idxs, so = refpipe.run_overlap_chain(str(tmp_path), fa, seed_cutoff=3000)
    piles = [(seqs, st, en, mal, min(en[0] // 2, 10000)) for _, seqs, st, en, mal, _ in
             refpipe.read_piles(idxs, so, min_len_seed=1500)]
    assert len(piles) > 40
    got = api.correct_batch(piles, read_type=1)
    for p, g_ in zip(piles, got):
        a = util.call_correct(rfn, rfr, dict(seqs=p[0], aln_start=p[1], aln_end=p[2], max_aln=p[3], max_lq=p[4],
                                             read_type=1, fast=0, split=0))
        assert a[0] == g_[0] and (a[0] <= 4 or (a[2] == g_[2] and np.float32(a[1]) == np.float32(g_[1])))


def _fork_worker(args):
    from nextdenovo_amd import api
    return api.correct(*args)[0]


def test_fork_pool_like_nextcorrect_py(lib):
    """lib/nextcorrect.py loads the library, THEN forks its worker pool (:56,:232).  HIP must
    initialise lazily inside each child."""
    import subprocess
    import sys
    code = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from multiprocessing import get_context
import util
from nextdenovo_amd import api
api.load()                      # parent: dlopen only, no HIP call
import test_gpu_parity as T
p = util.load_piles()[0]
args = (p["seqs"], p["aln_start"], p["aln_end"], p["max_aln"], 500, 130, 4, p["max_lq"])
with get_context("fork").Pool(2) as pool:
    r = pool.map(T._fork_worker, [args, args, args])
assert r == [p["exp_len"]] * 3, r
print("FORK_OK")
''' % (os.path.dirname(util.HERE), util.HERE)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "FORK_OK" in out.stdout, out.stdout + out.stderr


def test_prefix_and_extension_family_vs_oracle(lib, oracle_lib):
    """`ide`, `alnpos`, `extend_fwd`, `extend_rev` on the device (ndgpu_ext_batch, one lane per problem, and the single-call
    exports with the reference's signatures) against oracle/ond_ext_oracle.c, which tests/test_oracle.py pins to the compiled
    reference: budgets and bands of the HiFi mode-3 overlap path (minimap2/map.c:404-406, 941-943)."""
    from nextdenovo_amd import api, synth
    o = oracle_lib
    o.nd_oracle_ide.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    o.nd_oracle_alnpos.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint * 6)]
    o.nd_oracle_extend.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.POINTER(C.c_int),
                                   C.POINTER(C.c_int)]
    rng = np.random.default_rng(321)
    jobs, want = [], []
    for it in range(260):
        L = int(rng.integers(10, 5000))
        base = rng.integers(0, 4 if it % 9 else 2, L, dtype=np.uint8)
        prof = ("hifi", "hifi", "ont", "clr")[it % 4]
        q = synth.mutate(base, np.random.default_rng(5 * it), prof)[0]
        t = synth.mutate(base, np.random.default_rng(5 * it + 1), prof)[0]
        if it % 5 == 0:
            cut = int(rng.integers(5, max(6, L // 2)))
            t = np.concatenate([t[:cut], rng.integers(0, 4, int(rng.integers(50, 1500)), dtype=np.uint8)])
        if it % 7 == 0:
            q = q[: max(1, q.size - int(rng.integers(0, 300)))]
        qa, ta = util.ASC[q].tobytes(), util.ASC[t].tobytes()
        minlen = min(len(qa), len(ta))
        max_d = min(6000, minlen // 4 if minlen > 20 else minlen)
        for rev, kind in ((0, "extend_fwd"), (1, "extend_rev")):
            x, y = C.c_int(0), C.c_int(0)
            o.nd_oracle_extend(qa, len(qa), ta, len(ta), max_d, 500, 0.1, rev, C.byref(x), C.byref(y))
            jobs.append((kind, qa, ta, max_d, 500, 0.1))
            want.append((None, x.value, y.value, None))
        md = min(6000, max(len(qa), len(ta)) // 5)
        band = 500 if md > 1500 else md // 3
        pos = (C.c_uint * 6)(*([4242] * 6))
        o.nd_oracle_alnpos(qa, len(qa), ta, len(ta), md, band, C.byref(pos))
        jobs.append(("alnpos", qa, ta, md, band, 0.0))
        want.append((int(pos[0] != 4242), None, None, list(pos)))
        m, b = C.c_int(-1), C.c_int(-1)
        o.nd_oracle_ide(qa, len(qa), ta, len(ta), md, band, C.byref(m), C.byref(b))
        jobs.append(("ide", qa, ta, md, band, 0.0))
        want.append((int(m.value >= 0), m.value, b.value, None))
    got = api.ext_batch(jobs)
    n_aln = 0
    for i, ((kind, *_), w, g) in enumerate(zip(jobs, want, got)):
        done, a, b, pos = g
        if kind.startswith("extend"):
            assert (a, b) == (w[1], w[2]), (i, kind)
        elif kind == "ide":
            assert done == w[0] and (not done or (a, b) == (w[1], w[2])), (i, kind)
        else:
            assert done == w[0] and (not done or pos == w[3]), (i, kind)
            n_aln += done
    assert n_aln > 100
    # the single-call exports (lib/align.h:51-58 signatures; V / D ignored)
    P = C.c_void_p
    lib.extend_fwd.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, P, P, C.c_int, C.c_int, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.alnpos.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, P, P, C.c_int, C.c_int, C.POINTER(C.c_uint * 6)]
    lib.ide.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, P, P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for f in (lib.extend_fwd, lib.alnpos, lib.ide):
        f.restype = None
    for i in (0, 3, 6):
        kind, qa, ta, md, band, f = jobs[i * 4 + 0]
        x, y = C.c_int(-5), C.c_int(-5)
        lib.extend_fwd(qa, len(qa), ta, len(ta), None, None, md, band, 0.1, C.byref(x), C.byref(y))
        assert (x.value, y.value) == (want[i * 4][1], want[i * 4][2])
        kind, qa, ta, md, band, f = jobs[i * 4 + 2]
        pos = (C.c_uint * 6)(*([4242] * 6))
        lib.alnpos(qa, len(qa), ta, len(ta), None, None, md, band, C.byref(pos))
        assert list(pos) == want[i * 4 + 2][3]
        m, b = C.c_int(-1), C.c_int(-1)
        lib.ide(qa, len(qa), ta, len(ta), None, None, md, band, C.byref(m), C.byref(b))
        assert (m.value, b.value) == (want[i * 4 + 3][1], want[i * 4 + 3][2])
