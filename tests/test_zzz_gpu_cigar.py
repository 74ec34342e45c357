"""`minimap2-nd --step 1 -c` on the MI355X (ndgpu_ovl_map_cigar: mm_align_skeleton as batches of dynamic-programming problems on
the device, csrc/ovl_cigar.cpp + csrc/ksw2_kernels.hip) against the bytes the compiled reference writes: the golden `.ovl` files
of tests/golden/make_cigar_golden.py (plain read sets and a set with insertions / inversions that z-drop, split chains and bring
up the inversion alignment), the command line, the chains themselves against the overlap oracle, and -- where oracle/_ref
travelled -- a fresh read set against the reference binary run on the spot.  (Named to run last: these paths were finished after the
round's GPU budget was spent and have run under the kernel interpreter only.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import mm_util as M  # noqa: E402
import make_cigar_golden as G  # noqa: E402

pytestmark = pytest.mark.gpu


def dev_opts(preset, dual, extra):
    from nextdenovo_amd import overlap
    o = overlap.preset(preset)
    if dual:
        o.no_dual = 0
    if "--dvt" in extra:
        o.dvt = 1
    ao = overlap.aln_opt()
    if "-z" in extra:
        z = extra[extra.index("-z") + 1].split(",")
        ao.zdrop, ao.zdrop_inv = int(z[0]), int(z[-1])
    if "-s" in extra:
        ao.min_dp_max = int(extra[extra.index("-s") + 1])
    if "-O" in extra:   # minimap2/main.c:356-358: one value sets both gap pieces
        v = extra[extra.index("-O") + 1].split(",")
        ao.q, ao.q2 = int(v[0]), int(v[-1])
    if "-E" in extra:
        v = extra[extra.index("-E") + 1].split(",")
        ao.e, ao.e2 = int(v[0]), int(v[-1])
    if "--cap-sw-mem" in extra:
        ao.max_sw_mat = int(extra[extra.index("--cap-sw-mem") + 1])
    if "-f" in extra:   # -f INT,INT: the occurrence threshold and the re-chaining one (main.c:337-343)
        head, _, tail = extra[extra.index("-f") + 1].partition(",")
        o.mid_occ, o.max_occ = int(float(head) + .499), int(float(tail) + .499) if tail else 0
    return o, ao


def run_case(case):
    from nextdenovo_amd import overlap
    tag, preset, t, q, dual, extra = case
    o, ao = dev_opts(preset, dual, extra)
    T, Q = overlap.ReadSet.from_2bit(G.set_path(t)), overlap.ReadSet.from_2bit(G.set_path(q))
    with overlap.Index(o, T) as ix:
        recs, st = ix.map_cigar(T, Q, o.mid_occ if o.mid_occ > 0 else ix.mid_occ(), ao, want_stats=True)
    return overlap.encode(recs, np.zeros(2, dtype=np.uint32)), st


_API_CASES = [c for c in G.CASES_C if "-I" not in c[5]]


@pytest.mark.parametrize("case", _API_CASES, ids=[c[0] for c in _API_CASES])
def test_cigar_bytes_match_reference_golden(case):
    with open(os.path.join(G.OUT, case[0] + ".ovl"), "rb") as f:
        want = f.read()
    got, st = run_case(case)
    assert got == want
    assert st["chains"] > 0 and st["first_pass"] > st["chains"] and st["cells"] > 0
    if ".sv" in case[0]:  # the rearranged reads: z-drops, second passes, splits, inversion tests; aligned inversions in most sets
        assert st["second_pass"] > 50 and st["splits"] > 50 and st["inversion_tests"] > 50, st


@pytest.mark.parametrize("case", [G.CASES_C[0], G.CASES_C[7], G.CASES_C[8], G.CASES_C[10]], ids=[G.CASES_C[i][0] for i in (0, 7, 8, 10)])
def test_cigar_cli_writes_reference_bytes(case, tmp_path):
    """`python -m nextdenovo_amd.minimap2_nd --step 1 -c ...` with the reference's own command line (-z, -s included)."""
    from nextdenovo_amd import minimap2_nd
    tag, preset, t, q, dual, extra = case
    out = str(tmp_path / "o.ovl")
    argv = ["--step", "1"] + (["--dual=yes"] if dual else []) + ["-t", "8", "-x", preset, *extra, G.set_path(t), G.set_path(q), "-o", out]
    assert minimap2_nd.run(argv) == 0
    with open(os.path.join(G.OUT, tag + ".ovl"), "rb") as f:
        want = f.read()
    with open(out, "rb") as f:
        assert f.read() == want


def check_chains_against_oracle(olib, preset, dual, t, q):
    """ndgpu_ovl_map_chains: every read's hits (strand, target, a[] offset, count, score, hash) in hit order and the coordinates the
    chained anchors give == mm_gen_regs of the oracle (oracle/mm_oracle.c: nd_mm_map_read)."""
    from nextdenovo_amd import overlap
    o = overlap.preset(preset)
    if dual:
        o.no_dual = 0
    T, Q = overlap.ReadSet.from_2bit(G.set_path(t)), overlap.ReadSet.from_2bit(G.set_path(q))
    tids, tlens, tcodes, toff = M.load_set(G.set_path(t))
    qids, qlens, qcodes, qoff = M.load_set(G.set_path(q))
    oo = M.preset(preset, dual)
    ix = olib.nd_mm_index_build(tids.size, M.ptr(tcodes), M.ptr(toff), M.ptr(tlens), M.ptr(tids), oo.w, oo.k, oo.hpc)
    mid = olib.nd_mm_index_mid_occ(ix, np.float32(2e-4))
    with overlap.Index(o, T) as dix:
        assert dix.mid_occ() == mid
        ch, cnt, ax, ay, a_off = dix.map_chains(Q, mid)
    assert cnt.sum() == ch.size and a_off[-1] == ax.size
    c0 = 0
    regs = np.zeros(4096, dtype=M.REG)
    n_checked = 0
    for i in range(qids.size):
        n = olib.nd_mm_map_read(ix, C.byref(oo), mid, int(qids[i]), M.ptr(qcodes[int(qoff[i]):]), int(qlens[i]), M.ptr(regs), regs.size)
        assert n == int(cnt[i])
        x, y = ax[int(a_off[i]):int(a_off[i + 1])], ay[int(a_off[i]):int(a_off[i + 1])]
        for k in range(n):
            h, r = ch[c0 + k], regs[k]
            assert (int(h["rev"]), int(h["qname"]), int(h["qs"]), int(h["qe"]), int(h["tname"]), int(h["ts"])) == \
                (int(r["rev"]), int(r["rid"]), int(r["as_"]), int(r["cnt"]), int(r["score"]), int(r["hash"]))
            first, last = int(h["qs"]), int(h["qs"]) + int(h["qe"]) - 1
            span = int(y[first] >> np.uint64(32)) & 0xff
            assert int(x[first]) >> 63 == int(r["rev"]) and (int(x[first]) << 1 & 0xffffffffffffffff) >> 33 == int(r["rid"])
            assert max(0, (int(x[first]) & 0xffffffff) + 1 - span) == int(r["rs"]) and (int(x[last]) & 0xffffffff) + 1 == int(r["re"])
            n_checked += 1
        c0 += n
    olib.nd_mm_index_free(ix)
    assert n_checked > 500


@pytest.mark.parametrize("preset,dual,t,q", [("ava-ont", True, "seed", "part"), ("ava-pb", False, "sv", "sv")])
def test_chains_match_oracle(oracle_lib, preset, dual, t, q):
    check_chains_against_oracle(M.bind(oracle_lib), preset, dual, t, q)


def test_fresh_reads_against_the_reference_binary(tmp_path):
    """A read set nobody has seen (other genome, other seed, deeper, longer rearrangements) against oracle/_ref/minimap2-nd -c."""
    ref = os.path.join(M.REFDIR, "minimap2-nd")
    if not (os.path.exists(ref) and os.path.exists(os.path.join(M.REFDIR, "seq_dump"))):
        pytest.skip("oracle/_ref (the compiled reference) did not travel")
    from nextdenovo_amd import overlap, synth
    rng = np.random.default_rng(5)
    g = synth.make_genome(70000, seed=31, n_repeats=5, repeat_len=2200)
    rs = synth.simulate_reads(g, 30, "ont", seed=32, mu=9.1, sigma=0.45, min_len=2500)
    seqs = []
    for n, s in enumerate(rs.seqs):
        s = s.copy()
        if n % 3 == 1 and s.size > 6000:
            p = int(s.size * 0.4)
            s = np.concatenate([s[:p], rng.integers(0, 4, int(rng.integers(400, 1500))).astype(np.uint8), s[p:]])
        if n % 3 == 2 and s.size > 6000:
            p, ln = int(s.size * 0.55), int(rng.integers(600, 1800))
            s[p:p + ln] = synth.revcomp_codes(s[p:p + ln])
        seqs.append(s)
    seed, part = M.dump_reads(str(tmp_path / "w"), [synth.codes_to_ascii(s) for s in seqs], seed_cutoff=9000)
    for preset, dual in (("ava-ont", True), ("ava-pb", False)):
        want = M.ref_step1(seed, part if dual else seed, str(tmp_path / "ref.ovl"), preset, dual, ("-c",), threads=16)
        o, ao = dev_opts(preset, dual, ())
        T = overlap.ReadSet.from_2bit(seed)
        Q = overlap.ReadSet.from_2bit(part) if dual else T
        with overlap.Index(o, T) as ix:
            recs, st = ix.map_cigar(T, Q, o.mid_occ if o.mid_occ > 0 else ix.mid_occ(), ao, want_stats=True)
        assert overlap.encode(recs, np.zeros(2, dtype=np.uint32)) == want
        assert st["splits"] > 0


@pytest.mark.parametrize("hq", [False, True])
def test_sort_in_seed_ranges_equals_the_sort_at_once(monkeypatch, hq):
    """The out-of-core form of `ovl_sort` (also new at the end of round 3, hence in this last file): tests/test_gpu_ovlsort.py."""
    import test_gpu_ovlsort as GS
    GS.check_sort_in_seed_ranges_equals_the_sort_at_once(monkeypatch, hq)


def test_ksw_ll_matches_reference_vectors(oracle_lib):
    """ksw_ll_kernel (the inversion test's local-alignment score) against the compiled reference's vectors and the oracle."""
    import test_zz_gpu_ksw2 as GK
    from nextdenovo_amd import overlap
    GK.check_ll(overlap.load(), oracle_lib)


def test_device_equals_oracle_on_unseen_reads(oracle_lib):
    """The device path against the oracle's restatement of the -c path (oracle/cigar_oracle.c) on reads neither has a golden file
    for -- needs no compiled reference on the box."""
    from nextdenovo_amd import overlap, synth
    olib = M.bind(oracle_lib)
    rng = np.random.default_rng(9)
    g = synth.make_genome(30000, seed=41, n_repeats=3, repeat_len=1500)
    rs = synth.simulate_reads(g, 18, "ont", seed=42, mu=8.8, sigma=0.4, min_len=2500)
    seqs = []
    for n, s in enumerate(rs.seqs):
        s = s.copy()
        if n % 3 == 1 and s.size > 5000:
            p = int(s.size * 0.45)
            s = np.concatenate([s[:p], rng.integers(0, 4, int(rng.integers(500, 1000))).astype(np.uint8), s[p:]])
        if n % 3 == 2 and s.size > 5000:
            p, ln = int(s.size * 0.5), int(rng.integers(700, 1200))
            s[p:p + ln] = synth.revcomp_codes(s[p:p + ln])
        seqs.append(s)
    n = len(seqs)
    ids = np.arange(1, n + 1, dtype=np.uint32)
    lens = np.asarray([s.size for s in seqs], dtype=np.uint32)
    words = [synth.pack_2bit_msb(s) for s in seqs]
    woff = np.zeros(n, dtype=np.uint64)
    woff[1:] = np.cumsum([w.size for w in words])[:-1]
    R = overlap.ReadSet(ids, lens, np.concatenate(words), woff)
    codes = np.concatenate(seqs).astype(np.uint8)
    off = np.zeros(n, dtype=np.uint64)
    off[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
    oset = (ids, lens, codes, off)
    for preset in ("ava-ont", "ava-pb"):
        o, ao = dev_opts(preset, False, ())
        with overlap.Index(o, R) as ix:
            mid = ix.mid_occ()
            recs, st = ix.map_cigar(R, R, mid, ao, want_stats=True)
        want, _ = M.step1_cigar(olib, M.preset(preset, False), M.aln_opt(), oset, oset, mid_occ=mid)
        assert overlap.encode(recs, np.zeros(2, dtype=np.uint32)) == want
        assert st["splits"] > 0 and len(want) > 1000
