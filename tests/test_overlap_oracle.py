"""The overlap oracle (oracle/mm_oracle.c) against (1) the committed golden `.ovl` files written by the compiled
reference `minimap2-nd --step 1` and (2), when oracle/_ref is present, the reference binary run live on a fresh
seeded read set.  Byte-for-byte."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import mm_util as M  # noqa: E402
from make_overlap_golden import CASES, CASES_M3, SETS  # noqa: E402

GOLD = os.path.join(HERE, "golden", "overlap")


@pytest.fixture(scope="module")
def lib(oracle_lib):
    return M.bind(oracle_lib)


@pytest.fixture(scope="module")
def sets():
    return {k: M.load_set(os.path.join(GOLD, k + ".2bit")) for k in SETS}


def case_kwargs(extra, preset=None):
    kw = {}
    if preset == "ava-hifi":
        kw["mid_occ_frac"] = 1e-4  # options.c:108
    if "-f" in extra:
        x = float(extra[extra.index("-f") + 1].partition(",")[0])
        if x < 1.0:
            kw["mid_occ_frac"] = x
        else:
            kw["mid_occ"] = int(x + .499)
    if "-I" in extra:
        from nextdenovo_amd.minimap2_nd import parse_num
        kw["batch_size"] = parse_num(extra[extra.index("-I") + 1])
    if "--mode" in extra and extra[extra.index("--mode") + 1] == "3":
        kw["mode3"] = True
    return kw


def opt_overrides(extra):
    """-k / -w / -n / -m / --minlen of a case's argv as fields of the oracle's option struct."""
    kw = {}
    for flag, name in (("-k", "k"), ("-w", "w"), ("-n", "min_cnt"), ("-m", "min_sc"), ("--minlen", "minlen")):
        if flag in extra:
            kw[name] = int(extra[extra.index(flag) + 1])
    return kw


def max_occ_of(extra):
    """-f FLOAT,INT (main.c:343): the re-chaining threshold, 0 when the option has no second number."""
    tail = extra[extra.index("-f") + 1].partition(",")[2] if "-f" in extra else ""
    return int(float(tail) + .499) if tail else 0


@pytest.mark.parametrize("case", CASES + CASES_M3, ids=[c[0] for c in CASES + CASES_M3])
def test_oracle_matches_golden_ovl(lib, sets, case):
    tag, preset, t, q, dual, extra = case
    with open(os.path.join(GOLD, tag + ".ovl"), "rb") as f:
        want = f.read()
    got, _ = M.step1(lib, M.preset(preset, dual, dvt=1 if "--dvt" in extra else 0, max_occ=max_occ_of(extra), **opt_overrides(extra)), sets[t], sets[q],
                     **case_kwargs(extra, preset))
    assert got == want


def test_golden_set_has_equal_coordinate_anchors(lib, sets):
    """The fixture must exercise the tie order of the reference's unstable radix sort."""
    ids, lens, codes, off = sets["seed"]
    opt = M.preset("ava-ont", False)
    ix = lib.nd_mm_index_build(ids.size, M.ptr(codes), M.ptr(off), M.ptr(lens), M.ptr(ids), opt.w, opt.k, opt.hpc)
    mid = lib.nd_mm_index_mid_occ(ix, np.float32(2e-4))
    dup = 0
    for i in range(ids.size):
        mv = M.sketch(lib, codes[int(off[i]): int(off[i]) + int(lens[i])], opt.w, opt.k)
        a = np.zeros(max(1, mv.size * mid), dtype=M.MM128)
        n = lib.nd_mm_seeds(ix, C.byref(opt), str(int(ids[i])).encode(), int(lens[i]), mid, M.ptr(mv), mv.size, M.ptr(a), 1)
        x = a["x"][:n]
        assert (np.diff(x.astype(np.int64).view(np.uint64)) >= 0).all() if n > 1 else True
        dup += int((x[1:] == x[:-1]).sum())
    lib.nd_mm_index_free(ix)
    assert dup > 0


@pytest.mark.skipif(not os.path.exists(os.path.join(M.REFDIR, "minimap2-nd")), reason="oracle/_ref not built")
@pytest.mark.parametrize("profile,preset", [("ont", "ava-ont"), ("clr", "ava-pb"), ("hifi", "ava-hifi")])
def test_oracle_matches_live_reference(lib, profile, preset):
    from nextdenovo_amd import synth
    g = synth.make_genome(50000, seed=21, n_repeats=4, repeat_len=1200)
    rs = synth.simulate_reads(g, 18, profile, seed=22)
    wd = tempfile.mkdtemp(prefix="ndmm")
    seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in rs.seqs], seed_cutoff=8000)
    S, P = M.load_set(seed), M.load_set(part)
    for t, q, dual in ((seed, part, True), (seed, seed, False)):
        want = M.ref_step1(t, q, os.path.join(wd, "o.ovl"), preset, dual)
        got, _ = M.step1(lib, M.preset(preset, dual), S, S if q == seed else P, **case_kwargs((), preset))
        assert len(want) > 1000 and got == want


@pytest.mark.skipif(not os.path.exists(os.path.join(M.REFDIR, "minimap2-nd")), reason="oracle/_ref not built")
@pytest.mark.parametrize("extra", [(), ("--dvt",), ("-f", "40")])
def test_oracle_mode3_end_extension_matches_live_reference(lib, sets, extra):
    """`--step 1 --mode 3` (HiFi: nd_extend_ends stretches every hit into the unaligned read ends with extend_rev / extend_fwd
    before the step-1 filter, minimap2/map.c:385-482, 919-928): the oracle's `.ovl` against the compiled reference's, on the
    committed HiFi read sets, both strands, with and without the dovetail pre-filter."""
    wd = tempfile.mkdtemp(prefix="ndm3")
    for t, q, dual in (("hseed", "hseed", False), ("hseed", "hpart", True)):
        want = M.ref_step1(os.path.join(GOLD, t + ".2bit"), os.path.join(GOLD, q + ".2bit"), os.path.join(wd, "o.ovl"), "ava-hifi", dual,
                           extra=("--mode", "3") + tuple(extra))
        plain = M.ref_step1(os.path.join(GOLD, t + ".2bit"), os.path.join(GOLD, q + ".2bit"), os.path.join(wd, "p.ovl"), "ava-hifi", dual,
                            extra=tuple(extra))
        kw = case_kwargs(extra, "ava-hifi")
        got, _ = M.step1(lib, M.preset("ava-hifi", dual, dvt=1 if "--dvt" in extra else 0), sets[t], sets[q], mode3=True, **kw)
        assert len(want) > 5000 and got == want
        assert want != plain   # the extension really moves coordinates on this set


@pytest.mark.skipif(not os.path.exists(os.path.join(M.REFDIR, "minimap2-nd")), reason="oracle/_ref not built")
@pytest.mark.parametrize("mode,genome,depth", [(0, 60000, 28), (2, 60000, 28), (2, 30000, 150), (1, 60000, 28)], ids=["mode0", "mode2", "mode2-deep", "mode1"])
@pytest.mark.parametrize("preset,extra", [("ava-ont", ("-k", "17", "-w", "17", "--minlen", "1000", "--maxhan1", "2000")),
                                          ("ava-pb", ("-k", "17", "-w", "10", "--minlen", "700", "--maxhan1", "1500", "--maxhan2", "300"))])
def test_oracle_step2_matches_live_reference(lib, preset, extra, mode, genome, depth):
    """`minimap2-nd --step 2` on corrected reads (the cns_align command of nextDenovo:356-366): per-target marking of the hits, the
    re-alignment of the marked candidates with the short k-mer sketch (the default --mode 2, written as nextDenovo writes it:
    without --mode; fewer than 200 candidates per read: every candidate against a one-read index of the query; `mode2-deep`: 200
    and more: the query against batches of cn candidates) or none (--mode 0), the length / identity / block-length filters, the
    dovetail / contained filter (oracle/step2_oracle.c) and the 10-field encoder -- `.ovl` and `.bl` byte for byte against the
    compiled reference."""
    from nextdenovo_amd import synth
    import refpipe
    if preset == "ava-pb" and (depth > 100 or mode == 1):
        pytest.skip("the batched form and --mode 1 are pinned live with the ava-ont options (and on the committed golden runs)")
    g = synth.make_genome(genome, seed=61, n_repeats=3 if genome > 20000 else 0, repeat_len=1500)
    rs = synth.simulate_reads(g, depth, "hifi", seed=62, mu=8.6, sigma=0.35, min_len=2500)
    seqs = list(rs.seqs)
    rng = np.random.default_rng(5)
    for t in range(25):                      # short reads inside longer ones: contained verdicts
        a = int(rng.integers(0, len(seqs)))
        if seqs[a].size > 3000:
            s0 = int(rng.integers(0, seqs[a].size - 2600))
            seqs.append(seqs[a][s0:s0 + 2600].copy())
    wd = tempfile.mkdtemp(prefix="nds2")
    half = len(seqs) // 2
    files, sets = [], []
    for tag, lo, hi in (("a", 0, half), ("b", half, len(seqs))):
        p = os.path.join(wd, tag + ".fasta")
        with open(p, "w") as f:
            for i in range(lo, hi):
                f.write(">%d %d 0.99\n%s\n" % (i + 1, seqs[i].size, synth.codes_to_ascii(seqs[i]).decode()))
        files.append(p)
        ids = np.arange(lo + 1, hi + 1, dtype=np.uint32)
        lens = np.asarray([seqs[i].size for i in range(lo, hi)], dtype=np.uint32)
        off = np.zeros(hi - lo, dtype=np.uint64)
        off[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
        sets.append((ids, lens, np.concatenate([seqs[i] for i in range(lo, hi)]).astype(np.uint8), off))
    out = os.path.join(wd, "o.ovl")
    cmd = [os.path.join(M.REFDIR, "minimap2-nd"), "--step", "2", *(("--mode", str(mode)) if mode != 2 else ()), "--dual=yes", "-t", "3", "-x", preset,
           *extra, files[0], files[1], files[0], "-o", out]
    refpipe.run(cmd)
    want, want_bl = open(out, "rb").read(), open(out + ".bl").read()
    kw = {}
    for k_, name in (("-k", "k"), ("-w", "w"), ("--minlen", "minlen"), ("--maxhan1", "maxhan1"), ("--maxhan2", "maxhan2")):
        if k_ in extra:
            kw[name] = int(extra[extra.index(k_) + 1])
    got, got_bl = M.step2(lib, M.preset(preset, True, **kw), sets[0], [sets[1], sets[0]], mode, cn=50 if mode == 1 else 20)
    assert len(want) > (3000 if depth < 100 else 1000) and got == want
    assert got_bl == want_bl and want_bl.count("\n") > 20
    if mode:
        cnt = (C.c_int64 * 2)()
        lib.nd_mm_step2_counters(cnt)
        assert cnt[0] + cnt[1] > 20 and (cnt[1] > 20 if depth > 100 else True), list(cnt)   # both forms of the re-alignment ran
        plain, _ = M.step2(lib, M.preset(preset, True, **kw), sets[0], [sets[1], sets[0]], 0)
        assert plain != got   # the re-alignment really changes records on this set


def _fasta_set(path):
    """FASTA[.gz] with numeric names -> (ids, lens, codes, off) as the oracle wants them."""
    import gzip
    ids, seqs = [], []
    with (gzip.open(path, "rt") if path.endswith(".gz") else open(path)) as f:
        for line in f:
            if line.startswith(">"):
                ids.append(int(line[1:].split()[0]))
                seqs.append([])
            else:
                seqs[-1].append(line.strip())
    code = np.full(256, 0, dtype=np.uint8)
    for i, c in enumerate(b"ACGT"):
        code[c] = i
    arrs = [code[np.frombuffer("".join(s).encode(), dtype=np.uint8)] for s in seqs]
    lens = np.asarray([a.size for a in arrs], dtype=np.uint32)
    off = np.zeros(len(arrs), dtype=np.uint64)
    off[1:] = np.cumsum(lens.astype(np.uint64))[:-1]
    return (np.asarray(ids, dtype=np.uint32), lens, np.concatenate(arrs).astype(np.uint8), off)


@pytest.mark.parametrize("tag", ["ont", "pb", "ont.m2", "pb.m2", "deep.m2", "ont.m1", "deep.m1", "ont.rechain", "ont.rechain.m2"]
                         + (["deep.rechain.m1"] if os.environ.get("NDGPU_SLOW_TESTS") else []))   # (20 s; all three modes of the re-chaining fixtures matched when they were made)
def test_oracle_step2_matches_golden(lib, tag):
    """The step-2 oracle on the committed fixtures of the compiled reference (tests/golden/step2; `.m2`: the command without --mode,
    i.e. with the re-alignment): what the GPU tests of the device path compare with on a box that has no reference."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_step2_golden import CASES as S0, CASES_M1, CASES_M2, CASES_RECHAIN, OUT
    argv = dict(S0 + CASES_M2 + CASES_M1 + CASES_RECHAIN)[tag]
    kw, f_kw = {}, {}
    if "-f" in argv:   # (-f INT,INT: the fixed threshold and the re-chaining one)
        f_kw, kw["max_occ"] = {"mid_occ_fixed": int(argv[argv.index("-f") + 1].partition(",")[0])}, max_occ_of(argv)
    for k_, name in (("-k", "k"), ("-w", "w"), ("--minlen", "minlen"), ("--maxhan1", "maxhan1"), ("--maxhan2", "maxhan2")):
        if k_ in argv:
            kw[name] = int(argv[argv.index(k_) + 1])
    if tag.startswith("deep"):
        a = _fasta_set(os.path.join(OUT, "c.fa.gz"))
        qs = [a]
    else:
        a, b = _fasta_set(os.path.join(OUT, "a.fa.gz")), _fasta_set(os.path.join(OUT, "b.fa.gz"))
        qs = [b, a]
    mode = 2 if tag.endswith(".m2") else 1 if tag.endswith(".m1") else 0
    got, got_bl = M.step2(lib, M.preset(argv[argv.index("-x") + 1], True, **kw), a, qs, mode, cn=50 if mode == 1 else 20, **f_kw)
    lib.nd_mm_step2_big_maps.restype = C.c_int64
    assert lib.nd_mm_step2_big_maps() == 0   # (no mapping of these fixtures has the 100,000 anchors mm_chain_dp_nextdenovo thins: see below)
    assert got == open(os.path.join(OUT, tag + ".ovl"), "rb").read()
    assert got_bl == open(os.path.join(OUT, tag + ".ovl.bl")).read()


def test_oracle_step2_anchor_thinning(lib, monkeypatch):
    """mm_chain_dp_nextdenovo (minimap2/chain.c:185-226): --mode 1 chains its one-read-index mappings through it, and a mapping with
    more than 100,000 anchors loses the anchors of crowded target positions before the DP.  tests/golden/step2/tandem.* is a read set
    across a tandem array whose mappings have 300,000 and more; the compiled reference's bytes are reproduced WITH the restated
    thinning and not without it (five of six random arrays behaved like this one when the fixture was chosen)."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_step2_golden import CASES_THIN, OUT
    tag, argv = CASES_THIN[0]
    a = _fasta_set(os.path.join(OUT, "tandem.fa.gz"))
    opt = M.preset("ava-ont", True, k=17, w=17, minlen=1000, maxhan1=2000)
    want, want_bl = open(os.path.join(OUT, tag + ".ovl"), "rb").read(), open(os.path.join(OUT, tag + ".ovl.bl")).read()
    got, got_bl = M.step2(lib, opt, a, [a], 1, cn=50, mid_occ_fixed=1000)
    lib.nd_mm_step2_big_maps.restype = C.c_int64
    assert lib.nd_mm_step2_big_maps() >= 5
    assert got == want and got_bl == want_bl
    monkeypatch.setenv("ND_ORACLE_NO_THINNING", "1")
    got0, got_bl0 = M.step2(lib, opt, a, [a], 1, cn=50, mid_occ_fixed=1000)
    assert (got0, got_bl0) != (want, want_bl)


def test_cigar_oracle_matches_golden_ovl(lib):
    """`--step 1 -c`: the oracle's restatement of mm_align_skeleton (oracle/cigar_oracle.c, scalar ksw2 / ksw_ll kernels) against the
    compiled reference's bytes on the rearranged reads -- z-drops, second passes, chain splits, inversion tests, aligned inversions,
    the homopolymer-compressed sketch, --dvt.  (All nine golden runs match; one is in the CPU suite, ~30 s.)"""
    import make_cigar_golden as G
    tag, preset, t, q, dual, extra = [c for c in G.CASES_C if c[0] == "pb.sv.dvt.c"][0]
    with open(os.path.join(G.OUT, tag + ".ovl"), "rb") as f:
        want = f.read()
    got, _ = M.step1_cigar(lib, M.preset(preset, dual, dvt=1), M.aln_opt(), M.load_set(G.set_path(t)), M.load_set(G.set_path(q)))
    assert got == want


@pytest.mark.skipif(not os.environ.get("NDGPU_SLOW_TESTS"), reason="~40 s; the kernel-level equivalence is in test_oracle_ksw2.py")
def test_cigar_oracle_one_gap_piece(lib):
    """`-c -O 4 -E 2`: the reference binary aligned this golden run with ksw_extz2_sse; the oracle's two-piece kernel with equal
    pieces reproduces its bytes."""
    import make_cigar_golden as G
    tag, preset, t, q, dual, extra = [c for c in G.CASES_C if c[0] == "ont.sv.O4E2.c"][0]
    with open(os.path.join(G.OUT, tag + ".ovl"), "rb") as f:
        want = f.read()
    got, _ = M.step1_cigar(lib, M.preset(preset, dual), M.aln_opt(q=4, e=2, q2=4, e2=2), M.load_set(G.set_path(t)), M.load_set(G.set_path(q)))
    assert got == want
