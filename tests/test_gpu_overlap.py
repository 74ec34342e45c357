"""MI355X overlap engine (libndgpu_overlap.so, through its C ABI) against the overlap oracle and the golden
`.ovl` files of the compiled reference.  Bit-exact at every stage: minimizers, index arrays, occurrence
threshold, sorted anchors (including the reference sort's order of equal keys), chain-DP arrays, records, bytes."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import mm_util as M  # noqa: E402
from make_overlap_golden import CASES, SETS  # noqa: E402
from test_overlap_oracle import case_kwargs, max_occ_of, opt_overrides  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(HERE, "golden", "overlap")


@pytest.fixture(scope="module")
def olib(oracle_lib):
    return M.bind(oracle_lib)


@pytest.fixture(scope="module")
def sets():
    from nextdenovo_amd import overlap
    out = {}
    for k in SETS:
        p = os.path.join(GOLD, k + ".2bit")
        out[k] = (overlap.ReadSet.from_2bit(p), M.load_set(p))
    return out


def dev_opt(preset, dual, extra=()):
    from nextdenovo_amd import overlap
    o = overlap.preset(preset)
    if dual:
        o.no_dual = 0
    kw = case_kwargs(extra)
    if "mid_occ_frac" in kw:
        o.mid_occ_frac = kw["mid_occ_frac"]
    if "mid_occ" in kw:
        o.mid_occ = kw["mid_occ"]
    if kw.get("mode3"):
        o.mode = 3
    if "--dvt" in extra:
        o.dvt = 1
    o.max_occ = max_occ_of(extra)
    for name, v in opt_overrides(extra).items():
        setattr(o, "min_chain_score" if name == "min_sc" else name, v)
    return o


def oracle_sketch_all(olib, oset, w, k, hpc, rid_is_index):
    ids, lens, codes, off = oset
    xs, ys, offs = [], [], [0]
    for i in range(ids.size):
        mv = M.sketch(olib, codes[int(off[i]): int(off[i]) + int(lens[i])], w, k, i if rid_is_index else 0, hpc)
        xs.append(mv["x"])
        ys.append(mv["y"])
        offs.append(offs[-1] + mv.size)
    return np.concatenate(xs), np.concatenate(ys), np.asarray(offs, dtype=np.uint64)


@pytest.mark.parametrize("preset", ["ava-ont", "ava-pb", "ava-hifi"])
@pytest.mark.parametrize("rid_is_index", [False, True])
def test_sketch_matches_oracle(olib, sets, preset, rid_is_index):
    from nextdenovo_amd import overlap
    o = overlap.preset(preset)
    for k in (("hseed", "hpart", "seed") if preset == "ava-hifi" else ("seed", "part")):
        rs, oset = sets[k]
        x, y, off = overlap.sketch(o, rs, rid_is_index)
        ex, ey, eoff = oracle_sketch_all(olib, oset, o.w, o.k, o.hpc, rid_is_index)
        assert np.array_equal(off, eoff)
        assert np.array_equal(x, ex) and np.array_equal(y, ey)


@pytest.mark.parametrize("preset", ["ava-ont", "ava-pb", "ava-hifi"])
def test_index_matches_oracle(olib, sets, preset):
    from nextdenovo_amd import overlap
    o = overlap.preset(preset)
    rs, (ids, lens, codes, off) = sets["hseed" if preset == "ava-hifi" else "seed"]
    ix = olib.nd_mm_index_build(ids.size, M.ptr(codes), M.ptr(off), M.ptr(lens), M.ptr(ids), o.w, o.k, o.hpc)
    nk, nm = olib.nd_mm_index_keys(ix), olib.nd_mm_index_n(ix)
    ekey, estart, epos = np.zeros(nk, np.uint64), np.zeros(nk + 1, np.int64), np.zeros(nm, np.uint64)
    olib.nd_mm_index_dump(ix, M.ptr(ekey), M.ptr(estart), M.ptr(epos))
    with overlap.Index(o, rs) as dix:
        key, start, pos = dix.dump()
        assert np.array_equal(key, ekey) and np.array_equal(start.astype(np.int64), estart) and np.array_equal(pos, epos)
        for f in (1e-4, 2e-4, 2e-3, 0.05):
            assert dix.mid_occ(f) == olib.nd_mm_index_mid_occ(ix, np.float32(f))
    olib.nd_mm_index_free(ix)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_ovl_bytes_match_reference_golden(sets, case):
    from nextdenovo_amd import overlap
    tag, preset, t, q, dual, extra = case
    if "-I" in extra:
        pytest.skip("multi-part index runs are covered through the stage CLI test")
    with open(os.path.join(GOLD, tag + ".ovl"), "rb") as f:
        want = f.read()
    o = dev_opt(preset, dual, extra)
    with overlap.Index(o, sets[t][0]) as ix:
        mid = o.mid_occ if o.mid_occ > 0 else ix.mid_occ()
        recs = ix.map(sets[q][0], mid)
        got = overlap.encode(recs, np.zeros(2, dtype=np.uint32))
        st = ix.stats()
    assert got == want
    assert st["anchors"] > 0 and st["chains"] > 0
    assert (st["rechained"] > 0) == (o.max_occ > 0)   # -f FLOAT,INT: some reads of those fixtures chain nothing at the first threshold


@pytest.mark.parametrize("preset,dual,tq", [("ava-ont", True, ("seed", "part")), ("ava-ont", False, ("seed", "seed")),
                                            ("ava-pb", False, ("seed", "seed")), ("ava-hifi", True, ("hseed", "hpart"))])
def test_anchors_and_chain_arrays_match_oracle(olib, sets, preset, dual, tq, monkeypatch):
    """Sorted anchors (tie order of the reference sort included) and the DP's f[] / p[] per query read."""
    from nextdenovo_amd import overlap
    monkeypatch.setenv("NDGPU_OVL_BATCH_ANCHORS", "100000000")
    o = dev_opt(preset, dual)
    oo = M.preset(preset, dual)
    trs, (tid, tl, tc, to) = sets[tq[0]]
    qrs, (qid, ql, qc, qo) = sets[tq[1]]
    ix = olib.nd_mm_index_build(tid.size, M.ptr(tc), M.ptr(to), M.ptr(tl), M.ptr(tid), oo.w, oo.k, oo.hpc)
    mid = olib.nd_mm_index_mid_occ(ix, np.float32(1e-4 if preset == "ava-hifi" else 2e-4))
    n_tie = 0
    with overlap.Index(o, trs) as dix:
        assert dix.mid_occ() == mid
        dix.map(qrs, mid)
        for i in range(qid.size):
            ax, ay, f, p = dix.debug_anchors(i)
            mv = M.sketch(olib, qc[int(qo[i]): int(qo[i]) + int(ql[i])], oo.w, oo.k, 0, oo.hpc)
            a = np.zeros(max(1, mv.size * mid), dtype=M.MM128)
            n = olib.nd_mm_seeds(ix, C.byref(oo), str(int(qid[i])).encode(), int(ql[i]), mid, M.ptr(mv), mv.size, M.ptr(a), 1)
            a = a[:n].copy()
            assert ax.size == n
            assert np.array_equal(ax, a["x"]) and np.array_equal(ay, a["y"]), "anchors of query %d" % i
            n_tie += int((a["x"][1:] == a["x"][:-1]).sum()) if n > 1 else 0
            if n:
                ef, ep = np.zeros(n, np.int32), np.zeros(n, np.int32)
                u = np.zeros(n, np.uint64)
                nb = C.c_int64(0)
                olib.nd_mm_chain(C.byref(oo), n, M.ptr(a), M.ptr(u), C.byref(nb), M.ptr(ef), M.ptr(ep))
                assert np.array_equal(f, ef) and np.array_equal(p, ep), "chain DP of query %d" % i
        st = dix.stats()
    olib.nd_mm_index_free(ix)
    if preset == "ava-ont":
        assert n_tie > 0 and st["tie_reads"] > 0  # the exact-replay path really ran


@pytest.mark.parametrize("profile,preset", [("ont", "ava-ont"), ("clr", "ava-pb"), ("hifi", "ava-hifi")])
def test_live_set_many_batches(olib, profile, preset, monkeypatch, tmp_path):
    """A fresh seeded read set, mapped in several small batches, against the oracle's whole-run bytes."""
    from nextdenovo_amd import overlap, synth
    monkeypatch.setenv("NDGPU_OVL_BATCH_ANCHORS", "20000")
    g = synth.make_genome(120000, seed=77, n_repeats=5, repeat_len=2000)
    rs = synth.simulate_reads(g, 20, profile, seed=78)
    n = len(rs.seqs)
    ids = np.arange(n, dtype=np.uint32)
    lens = np.asarray([s.size for s in rs.seqs], dtype=np.uint32)
    words = [synth.pack_2bit_msb(s) for s in rs.seqs]
    woff = np.zeros(n, dtype=np.uint64)
    woff[1:] = np.cumsum([w.size for w in words])[:-1]
    dset = overlap.ReadSet(ids, lens, np.concatenate(words), woff)
    codes = np.concatenate(rs.seqs).astype(np.uint8)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
    oset = (ids, lens, codes, off)
    for dual in (False, True):
        want, mid = M.step1(olib, M.preset(preset, dual), oset, oset, **case_kwargs((), preset))
        o = dev_opt(preset, dual)
        monkeypatch.setenv("NDGPU_OVL_LANES", "3" if dual else "1")   # the batches one after the other / three at a time (a stream each)
        with overlap.Index(o, dset) as ix:
            assert ix.mid_occ() == mid
            recs = ix.map(dset, mid)
            assert ix.stats()["batches"] >= 3
        got = overlap.encode(recs, np.zeros(2, dtype=np.uint32))
        assert len(want) > 5000 and got == want


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_stage_cli_writes_reference_bytes(case, tmp_path):
    """`python -m nextdenovo_amd.minimap2_nd` with the reference's own command line."""
    from nextdenovo_amd import minimap2_nd
    tag, preset, t, q, dual, extra = case
    out = str(tmp_path / "o.ovl")
    argv = ["--step", "1"] + (["--dual=yes"] if dual else []) + ["-t", "8", "-x", preset, *extra,
                                                                  os.path.join(GOLD, t + ".2bit"), os.path.join(GOLD, q + ".2bit"), "-o", out]
    assert minimap2_nd.run(argv) == 0
    with open(os.path.join(GOLD, tag + ".ovl"), "rb") as f:
        want = f.read()
    with open(out, "rb") as f:
        assert f.read() == want


def _adversarial_reads(seed=11):
    rng = np.random.default_rng(seed)
    reads = []
    for t in range(260):
        mode = t % 5
        n = int(rng.integers(1, 150)) if t % 3 else int(rng.integers(900, 2600))   # around and across the 1024-symbol tile
        if mode == 0:
            c = rng.integers(0, 4, n)
        elif mode == 1:
            c = rng.integers(0, 2, n)
        elif mode == 2:
            u = rng.integers(0, 4, int(rng.integers(1, 7)))
            c = np.tile(u, n // u.size + 1)[:n]
        elif mode == 3:
            c = rng.integers(0, 4, n)
            c[rng.integers(0, n, n // 3)] = 0
        else:
            parts = [np.full(int(rng.choice([1, 1, 2, 3, 40, 130, 260, 300])), int(rng.integers(0, 4))) for _ in range(int(rng.integers(3, 40)))]
            c = np.concatenate(parts)
        reads.append(c.astype(np.uint8))
    return reads


@pytest.mark.parametrize("k,w,hpc", [(15, 5, 0), (19, 5, 1), (5, 1, 0), (7, 8, 1), (11, 17, 0), (3, 2, 1), (27, 64, 0), (14, 5, 0), (6, 3, 1),
                                     (51, 51, 1), (51, 51, 0), (33, 4, 1), (63, 7, 1), (35, 64, 0), (47, 1, 1),
                                     (29, 5, 1), (30, 4, 0), (31, 9, 1), (34, 3, 0), (40, 6, 1), (36, 8, 0), (62, 2, 0), (65, 7, 1), (70, 5, 0),
                                     (100, 11, 0), (127, 2, 1)])
def test_sketch_adversarial_reads(olib, k, w, hpc):
    """Low-complexity / tandem / homopolymer-heavy / very short reads: the position-parallel K1 (odd k) and the
    sequential K1 (even k) both reproduce the window automaton, first-window quirks included; k > 32 is the two-word k-mer
    of ava-hifi, whose homopolymer-compressed span comes out of the reference's wrapped 32-slot run queue.  Every other k above 28
    (even: the tandem reads hold k-mers that equal their reverse complement; 29..31; three and four words) is the sequential K1
    with the k-mer in four words."""
    from nextdenovo_amd import overlap, synth
    reads = _adversarial_reads()
    lens = np.asarray([r.size for r in reads], dtype=np.uint32)
    words = [synth.pack_2bit_msb(r) for r in reads]
    woff = np.zeros(len(reads), dtype=np.uint64)
    woff[1:] = np.cumsum([x.size for x in words])[:-1]
    rs = overlap.ReadSet(np.arange(len(reads), dtype=np.uint32), lens, np.concatenate(words), woff)
    o = overlap.preset("ava-ont", k=k, w=w, hpc=hpc)
    x, y, off = overlap.sketch(o, rs, True)
    for i, r in enumerate(reads):
        mv = M.sketch(olib, r, w, k, i, hpc)
        lo, hi = int(off[i]), int(off[i + 1])
        assert hi - lo == mv.size and np.array_equal(x[lo:hi], mv["x"]) and np.array_equal(y[lo:hi], mv["y"]), \
            "read %d (len %d)" % (i, r.size)
