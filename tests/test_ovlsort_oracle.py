"""ovl_sort oracle (oracle/ovlsort_oracle.c): (1) the whole oracle chain -- overlap oracle -> sort oracle -- on the
committed stage fixture reproduces the reference's `sorted.ovl` / `.bl` (tests/golden/stage, written by the real
seq_dump -> minimap2-nd -> ovl_sort chain); (2) live against oracle/_ref/ovl_sort when it is built."""
import os
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mm_util as M  # noqa: E402
import os_util as O  # noqa: E402

STAGE = os.path.join(HERE, "golden", "stage")


@pytest.fixture(scope="module")
def libs(oracle_lib):
    return M.bind(oracle_lib), O.bind(oracle_lib)


def stage_raw_files(mlib):
    """The two raw .ovl inputs of the stage fixture (seed x part --dual=yes, seed x seed), from the overlap oracle."""
    from nextdenovo_amd import ovl
    S = M.load_set(os.path.join(STAGE, "input.seed.001.2bit"))
    P = M.load_set(os.path.join(STAGE, "input.part.001.2bit"))
    a, _ = M.step1(mlib, M.preset("ava-ont", True), S, P)
    b, _ = M.step1(mlib, M.preset("ava-ont", False), S, S)
    out = []
    for blob in (a, b):
        with tempfile.NamedTemporaryFile(suffix=".ovl") as f:
            f.write(blob)
            f.flush()
            out.append(ovl.decode_ovl(f.name))
    return out


def test_oracle_chain_matches_stage_fixture(libs):
    mlib, olib = libs
    raws = stage_raw_files(mlib)
    sl, mn = O.read_idx(os.path.join(STAGE, ".input.seed.001.idx"))
    blob, bl, _ = O.oracle_sort(olib, raws, sl, mn, max_bin_cov=40)
    with open(os.path.join(STAGE, "input.seed.001.sorted.ovl"), "rb") as f:
        assert blob == f.read()
    with open(os.path.join(STAGE, "input.seed.001.sorted.ovl.bl")) as f:
        assert bl == f.read()


@pytest.mark.skipif(not os.path.exists(os.path.join(O.REFDIR, "ovl_sort")), reason="oracle/_ref not built")
@pytest.mark.parametrize("k", [40, 22])
def test_oracle_matches_live_reference(libs, k):
    from nextdenovo_amd import ovl, synth
    mlib, olib = libs
    rng = np.random.default_rng(4)
    g = synth.make_genome(70000, seed=35, n_repeats=3, repeat_len=2000)
    rs = synth.simulate_reads(g, 45, "ont", seed=36)
    seqs = list(rs.seqs)
    for t in range(12):  # chimeric reads exercise the trimming / 'k' paths
        a, b = rng.integers(0, len(seqs), 2)
        y = synth.revcomp_codes(seqs[b]) if t % 2 else seqs[b]
        seqs.append(np.concatenate([seqs[a][: max(1500, seqs[a].size // 2)], y[: max(1500, y.size // 2)]]))
    wd = tempfile.mkdtemp(prefix="ndos")
    seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in seqs], seed_cutoff=7000)
    files = []
    if part:
        M.ref_step1(seed, part, os.path.join(wd, "a.ovl"), "ava-ont", True)
        files.append(os.path.join(wd, "a.ovl"))
    M.ref_step1(seed, seed, os.path.join(wd, "b.ovl"), "ava-ont", False)
    files.append(os.path.join(wd, "b.ovl"))
    idx = os.path.join(wd, "db", ".input.seed.001.idx")
    want, want_bl = O.ref_sort(wd, idx, files, k=k)
    sl, mn = O.read_idx(idx)
    blob, bl, _ = O.oracle_sort(olib, [ovl.decode_ovl(f) for f in files], sl, mn, max_bin_cov=k)
    assert len(want) > 10000 and blob == want and bl == want_bl


def _hq_inputs(profile, preset, chimeras, seed_cutoff, extra=()):
    """Reads + compiled-reference step-1 .ovl files for the -H tests."""
    from nextdenovo_amd import synth
    rng = np.random.default_rng(14)
    g = synth.make_genome(80000, seed=45, n_repeats=4, repeat_len=2500)
    kw = dict(mu=9.0, sigma=0.3, min_len=3000) if profile == "hifi" else {}
    rs = synth.simulate_reads(g, 35, profile, seed=46, **kw)
    seqs = list(rs.seqs)
    for t in range(chimeras):
        a, b = rng.integers(0, len(seqs), 2)
        y = synth.revcomp_codes(seqs[b]) if t % 2 else seqs[b]
        seqs.append(np.concatenate([seqs[a][: max(2500, seqs[a].size // 2)], y[: max(2500, y.size // 2)]]))
    wd = tempfile.mkdtemp(prefix="ndosh")
    seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in seqs], seed_cutoff=seed_cutoff)
    files = []
    if part:
        M.ref_step1(seed, part, os.path.join(wd, "a.ovl"), preset, True, extra=extra)
        files.append(os.path.join(wd, "a.ovl"))
    M.ref_step1(seed, seed, os.path.join(wd, "b.ovl"), preset, False, extra=extra)
    files.append(os.path.join(wd, "b.ovl"))
    return wd, files, os.path.join(wd, "db", ".input.seed.001.idx")


@pytest.mark.skipif(not os.path.exists(os.path.join(O.REFDIR, "ovl_sort")), reason="oracle/_ref not built")
@pytest.mark.parametrize("profile,preset,k,flank", [("hifi", "ava-hifi", 28, None), ("hifi", "ava-hifi", 6, 120), ("ont", "ava-ont", 30, None)])
def test_oracle_hq_variant_matches_live_reference(libs, profile, preset, k, flank):
    """`ovl_sort -H` (encode_ovl_filter_hq, del_repeat_alns, check_chimer_hq, the identity rule of contained reads):
    sorted.ovl and .bl of the compiled reference, byte for byte, on HiFi and on noisy reads with chimeras."""
    from nextdenovo_amd import ovl
    mlib, olib = libs
    wd, files, idx = _hq_inputs(profile, preset, 14, 8500 if profile == "hifi" else 7000, extra=("-f", "700") if profile == "hifi" else ())
    want, want_bl = O.ref_sort(wd, idx, files, k=k, flank=flank, hq=True)
    sl, mn = O.read_idx(idx)
    blob, bl, _ = O.oracle_sort(olib, [ovl.decode_ovl(f) for f in files], sl, mn, max_bin_cov=k, flank=flank or 300, hq=True)
    assert len(want) > 10000 and blob == want and bl == want_bl
    plain, plain_bl = O.ref_sort(wd, idx, files, k=k, flank=flank)
    assert plain != want            # the variant really differs on this input
