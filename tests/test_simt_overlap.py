"""The overlap library's kernel LOGIC on a machine without a GPU (see test_simt_kernels.py and tests/simt): its unmodified HIP
sources under the lane-accurate interpreter, rocPRIM's device primitives replaced by host stand-ins, against the compiled
reference's golden files and the oracles.  The bodies are the GPU tests' own (imported from test_gpu_overlap / test_gpu_ovlsort),
run with nextdenovo_amd.overlap bound to the interpreted library for the duration of a test."""
import ctypes as C
import os
import shutil
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(HERE, "simt"))
import mm_util as M  # noqa: E402
import test_gpu_overlap as GO  # noqa: E402  (gpu-marked as a module; its functions are called from here)
import test_gpu_ovlsort as GS  # noqa: E402


@pytest.fixture(scope="module")
def simt_libs():
    import build_simt
    os.environ.setdefault("NDGPU_CONTEXTS", "1")
    return C.CDLL(build_simt.build_overlap()), C.CDLL(build_simt.build())


@pytest.fixture()
def interpreted(simt_libs, monkeypatch):
    from nextdenovo_amd import api, overlap
    monkeypatch.setattr(overlap, "_lib", overlap._bind(simt_libs[0]))
    monkeypatch.setattr(api, "_LIB", api._bind(simt_libs[1]))
    return simt_libs


@pytest.fixture(scope="module")
def olib(oracle_lib):
    return M.bind(oracle_lib)


@pytest.fixture()
def sets(interpreted):
    from nextdenovo_amd import overlap
    out = {}
    for k in GO.SETS:
        p = os.path.join(GO.GOLD, k + ".2bit")
        out[k] = (overlap.ReadSet.from_2bit(p), M.load_set(p))
    return out


from make_overlap_golden import CASES_M3  # noqa: E402

_FAST_CASES = [c for c in GO.CASES + CASES_M3 if "-I" not in c[5] and (c[0] != "hifi.sxp.dual.k40" or os.environ.get("NDGPU_SLOW_TESTS"))]   # (k40: 24 s here; k70 covers the long k-mers, both run on the GPU)


@pytest.mark.parametrize("case", _FAST_CASES, ids=[c[0] for c in _FAST_CASES])
def test_ovl_bytes_match_reference_golden(sets, case):
    """K1 ... K6 end to end: the `.ovl` bytes of the compiled reference's minimap2-nd --step 1."""
    GO.test_ovl_bytes_match_reference_golden(sets, case)


def test_batches_side_by_side(interpreted, olib, monkeypatch, tmp_path):
    """A read set mapped in many small batches, one after the other and three at a time (NDGPU_OVL_LANES: a host thread and a stream per
    lane): the oracle's bytes either way.  The index's minimizer lookup goes through its open-addressing table from the first map call
    on here (NDGPU_OVL_HASH; by default an index builds the table when its second map call arrives)."""
    monkeypatch.setenv("NDGPU_OVL_HASH", "1")
    GO.test_live_set_many_batches(olib, "ont", "ava-ont", monkeypatch, tmp_path)


_CLI_CASES = [c for c in GO.CASES + CASES_M3 if "-I" in c[5] or "--mode" in c[5] or "," in "".join(c[5])]   # (, : -f FLOAT,INT)


@pytest.mark.parametrize("case", _CLI_CASES, ids=[c[0] for c in _CLI_CASES])
def test_stage_cli_writes_reference_bytes(interpreted, case, tmp_path, monkeypatch):
    """The command line itself, for the multi-part index runs and `--mode 3` (HiFi end extension, several extension launches)."""
    monkeypatch.setenv("NDGPU_OVL_EXT_SCRATCH", "20000")
    GO.test_stage_cli_writes_reference_bytes(case, tmp_path)


def test_step2_mode0_cli_and_abi(interpreted, tmp_path):
    """`--step 2 --mode 0` (corrected reads): K5's per-target marking and record filters, the host's dovetail / contained filter,
    10-field encoder and `.bl` table -- the compiled reference's bytes, through the command line (FASTA.gz input, multi-part
    index, one file against itself) and through the C ABI."""
    import test_zz_gpu_step2 as S2
    for i, (tag, argv) in enumerate(S2.CASES):
        if tag not in ("ont", "hifi.self"):   # (the k = 17 cases chain many anchors: 20-30 s each under the interpreter)
            continue
        d = tmp_path / str(i)
        d.mkdir()
        S2.run_case(tag, argv, d)
    S2.test_step2_records_and_verdicts(tmp_path)


def test_sketch_and_index_match_oracle(olib, sets):
    GO.test_sketch_matches_oracle(olib, sets, "ava-ont", True)
    GO.test_index_matches_oracle(olib, sets, "ava-pb")


@pytest.mark.parametrize("k,w,hpc", [(30, 4, 0), (31, 9, 1), (40, 6, 1), (70, 5, 0), (127, 2, 1), (51, 51, 1)])
def test_long_kmer_sketch(interpreted, olib, k, w, hpc, monkeypatch):
    """k > 28 outside the position-parallel kernels' range (and k = 51 sent there by hand): the sequential K1 with the k-mer in up
    to four words, on the low-complexity reads (palindromic even k-mers included)."""
    monkeypatch.setenv("NDGPU_OVL_SEQ_SKETCH", "1")
    GO.test_sketch_adversarial_reads(olib, k, w, hpc)


def test_jobs_of_a_rank_fused_and_side_by_side(interpreted, monkeypatch):
    """stage.Shard.overlaps with 4 seed files, the rank of seed file 2 on its own (no hand-over: 2 mirror jobs against the indexes of
    seed files 0 and 1, the part job and the jobs (2, 2), (2, 3) against its own): the jobs with one index and one set of options in ONE
    map call cut apart at the file boundaries, the calls against different indexes on threads of their own -- the records of every
    job equal to the job-by-job loop's."""
    from nextdenovo_amd import stage, synth
    g = synth.make_genome(60000, seed=5, n_repeats=3, repeat_len=1500)
    rs = synth.simulate_reads(g, 14, "ont", seed=6, mu=7.9, sigma=0.5, min_len=500)
    words, off, lens = synth.pack_db(rs)
    outs = {}
    for mode in ("serial", "grouped"):
        if mode == "serial":
            monkeypatch.setenv("NDGPU_STAGE_SERIAL", "1")
        else:
            monkeypatch.delenv("NDGPU_STAGE_SERIAL")
        sh = stage.Shard(words, off, lens, preset="ava-ont", seed_cutoff=2500, read_cutoff=500, n_seed_files=4, sort_k=20)
        try:
            assert len(sh.part_ids) >= 1 and all(x.size > 5 for x in sh.seed_ids)
            jobs = sh.jobs_of(2)
            assert len(jobs) == 4 + len(sh.part_ids) and sum(1 for j in jobs if j[1] != 2) == 2
            outs[mode] = sh.overlaps(2)
            assert sh.backend.calls == (len(jobs) if mode == "serial" else 4)   # own index: {part jobs, (2, 3)} and (2, 2); two mirrors
        finally:
            sh.close()
    assert len(outs["serial"]) == len(outs["grouped"]) and sum(r.size for r in outs["serial"]) > 200
    for a, b in zip(outs["serial"], outs["grouped"]):
        assert a.size == b.size and a.tobytes() == b.tobytes()


def test_options_drawn_at_random_match_oracle():
    """tools/fuzz_overlap_options.py: preset, k, w, -n, -m, -f INT[,INT], --dual, --mode 3 and the batch size drawn at random on small
    read sets with repeats and tandem arrays; the device library's `.ovl` bytes against the oracle's.  (A process of its own: the tool
    binds nextdenovo_amd.overlap to the interpreted library for good.)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "fuzz_overlap_options.py"), "5", "3"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "3 cases, 0 bad" in r.stdout


def test_anchors_and_chain_arrays_match_oracle(olib, sets, monkeypatch):
    GO.test_anchors_and_chain_arrays_match_oracle(olib, sets, "ava-ont", True, ("seed", "part"), monkeypatch)


@pytest.mark.parametrize("descending", [0, 1])
def test_overlap_then_sort_matches_reference_sorted_ovl(interpreted, descending):
    """ovl_sort's device path (radix sorts, one wavefront per seed for the admission chain, the `.bl` table); once with the lanes
    of a wavefront run lowest first, once highest first (code that leans on lock step without an ND_LOCKSTEP() mark passes under at
    most one of the two)."""
    for lib in interpreted:
        lib.simt_set_lane_order(descending)
    try:
        GS.test_device_overlap_then_sort_matches_reference_sorted_ovl()
    finally:
        for lib in interpreted:
            lib.simt_set_lane_order(0)


def test_whole_stage_from_2bit_to_cns_fasta(interpreted, tmp_path):
    """raw_align x2 -> sort_align -> seed_cns, every kernel of both libraries, in process: `cns.fasta` / `.idx` equal to what the
    reference chain wrote (the GPU test of the same name runs the last command as a child process)."""
    from nextdenovo_amd import minimap2_nd, nextcorrect, ovl_sort
    d = str(tmp_path / "w")
    shutil.copytree(GS.STAGE, d)
    seed, part = os.path.join(d, "input.seed.001.2bit"), os.path.join(d, "input.part.001.2bit")
    o0, o1 = os.path.join(d, "raw0.ovl"), os.path.join(d, "raw1.ovl")
    assert minimap2_nd.run(["--step", "1", "--dual=yes", "-t", "8", "-x", "ava-ont", seed, part, "-o", o0]) == 0
    assert minimap2_nd.run(["--step", "1", "-I", "3G", "-t", "8", "-x", "ava-ont", seed, seed, "-o", o1]) == 0
    fofn = os.path.join(d, "ovl.fofn")
    with open(fofn, "w") as f:
        f.write(o0 + "\n" + o1 + "\n")
    so = os.path.join(d, "mine.sorted.ovl")
    assert ovl_sort.run(["-m", "2g", "-t", "4", "-k", "40", "-i", os.path.join(d, ".input.seed.001.idx"), "-o", so, fofn]) == 0
    assert open(so, "rb").read() == GS._golden("input.seed.001.sorted.ovl")
    assert open(so + ".bl", "rb").read() == GS._golden("input.seed.001.sorted.ovl.bl")
    idxs = os.path.join(d, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(d)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(d, n) + "\n")
    out = os.path.join(d, "cns.fasta")
    nextcorrect.cli(["-f", idxs, "-i", so, "-r", "ont", "-p", "4", "-min_len_seed", "1250", "-o", out])
    assert open(out, "rb").read() == GS._golden("cns.default.fasta", gz=True)
    assert open(out + ".idx", "rb").read() == GS._golden("cns.default.fasta.idx", gz=True)


@pytest.mark.parametrize("tag", ["ont.m2", "hifi.self.m2", "ont.m1", "ont.rechain"] + (["pb.m2", "ont.I.m2", "deep.m2", "ont.rechain.m2", "deep.rechain.m1", "tandem.m1"] if os.environ.get("NDGPU_SLOW_TESTS") else []))
def test_step2_with_the_realignment(interpreted, tmp_path, tag):
    """`--step 2` as nextDenovo writes it (no --mode: every marked candidate mapped again with the short k-mer sketch -- hits per
    read, wanted-target lists and nameless units through the seed kernels, provisional records out of K5, the bookkeeping on the
    host): the compiled reference's `.ovl` / `.bl` bytes through the command line."""
    import test_zz_gpu_step2 as S2
    S2.run_case(tag, dict(S2.CASES_M2 + S2.CASES_M1 + S2.CASES_RECHAIN + S2.CASES_THIN)[tag], tmp_path)


_C_FAST = ("pb.sv.dvt.c",) + (("ont.sv.O4E2.c", "ont.sxp.dual.f6r300.c") if os.environ.get("NDGPU_SLOW_TESTS") else ())   # (~70 s each under the interpreter; all ten golden runs run on the GPU; O4E2 = one gap piece)


@pytest.mark.parametrize("tag", _C_FAST)
def test_step1_with_base_level_alignment(interpreted, tag):
    """`--step 1 -c` (mm_align_skeleton as batches of device problems: the ksw2 extension kernel in all its roles, the ksw_ll
    kernel of the inversion test, K5 handing out chains): the compiled reference's bytes on the rearranged reads -- z-drops,
    second passes, chain splits, inversion tests and aligned inversions all occur, with the homopolymer-compressed sketch and --dvt."""
    import test_zzz_gpu_cigar as GC
    case = [c for c in GC.G.CASES_C if c[0] == tag][0]
    GC.test_cigar_bytes_match_reference_golden(case)


def test_chains_of_the_base_level_alignment_match_oracle(interpreted, olib):
    import test_zzz_gpu_cigar as GC
    GC.check_chains_against_oracle(olib, "ava-ont", True, "seed", "part")


def test_seed_files_through_the_stage_pipeline(interpreted, tmp_path, monkeypatch):
    """`correct_stage` over three seed files: the next seed file's sort + pile admission during this one's consensus and two consensus
    calls in flight (stage.StagePipeline; the contexts serve the older call first) give the files that one seed file after the other
    (NDGPU_STAGE_SERIAL=1) gives."""
    from nextdenovo_amd import correct_stage, synth
    import refpipe
    g = synth.make_genome(32000, seed=51, n_repeats=2, repeat_len=1200)
    rs = synth.simulate_reads(g, 24, "ont", seed=52, mu=8.5, sigma=0.4, min_len=900)
    fa = os.path.join(str(tmp_path), "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(x) for x in rs.seqs])
    fofn = os.path.join(str(tmp_path), "input.fofn")
    with open(fofn, "w") as f:
        f.write(fa + "\n")
    outs = {}
    for mode in ("line", "serial"):
        if mode == "serial":
            monkeypatch.setenv("NDGPU_STAGE_SERIAL", "1")
        else:
            monkeypatch.delenv("NDGPU_STAGE_SERIAL", raising=False)
        d = os.path.join(str(tmp_path), "db_" + mode)
        out = os.path.join(str(tmp_path), "cns_" + mode)
        assert correct_stage.run(["--fofn", fofn, "--read-cutoff", "500", "--seed-cutoff", "4k", "--seed-cutfiles", "3", "-d", d, "-x", "ava-ont",
                                  "-k", "24", "-r", "ont", "-min_len_seed", "2000", "-p", "4", "-o", out]) == 0
        outs[mode] = {n[len("cns_" + mode):]: open(os.path.join(str(tmp_path), n), "rb").read()
                      for n in sorted(os.listdir(str(tmp_path))) if n.startswith("cns_" + mode + ".")}
    assert len(outs["line"]) == 6 and outs["line"] == outs["serial"]          # three .fasta + three .idx
    assert sum(len(v) for k, v in outs["line"].items() if k.endswith(".fasta")) > 60000


def test_sort_in_seed_ranges_equals_the_sort_at_once(interpreted, monkeypatch):
    """The out-of-core form of the overlap sort (tests/test_gpu_ovlsort.py) under the interpreter."""
    GS.check_sort_in_seed_ranges_equals_the_sort_at_once(monkeypatch, False)


def test_every_failed_device_operation_surfaces(interpreted, monkeypatch):
    """DESIGN section 7b (round 3): on a device short of memory an all-vs-all job once returned records built from half the anchors
    and no error.  The primitives' return values were dropped, and a failing primitive clears the runtime's sticky error.  Here
    every checked device operation of an index build + map + sort (block-pool allocation, rocPRIM primitive, kernel launch) is
    made to fail in turn (NDGPU_OVL_FAIL_AT): the call must raise MemoryError -- never return a different number of records --
    and the next, undisturbed call must give the undisturbed result."""
    from nextdenovo_amd import overlap
    dset, _ = None, None
    p = os.path.join(GO.GOLD, GO.SETS[0] + ".2bit")
    dset = overlap.ReadSet.from_2bit(p)
    lens = np.asarray(dset.lens, dtype=np.uint32)

    def job():
        with overlap.Index(overlap.preset("ava-ont"), dset) as ix:
            raw = ix.map(dset, ix.mid_occ())
        srt, bl, _ = overlap.sort_overlaps([raw], lens, int(lens.min()), 40, 300)
        return raw, srt, bl

    monkeypatch.delenv("NDGPU_OVL_FAIL_AT", raising=False)
    want = job()
    assert want[0].shape[0] > 100
    # fail the 1st, 2nd, ... checked operation of a job until a job gets through (the k-th operation does not exist)
    k = 1
    while True:
        monkeypatch.setenv("NDGPU_OVL_FAIL_AT", str(k))   # (a new value restarts the library's count)
        try:
            got = job()
        except MemoryError:
            k += 1 if k < 12 else 3   # (every operation of the first dozen, every third after: the job has ~110)
            continue
        assert all(np.array_equal(a, b) for a, b in zip(got[:2], want[:2])) and got[2] == want[2]
        break
    assert k > 60, k   # (the first operation that does not exist lies beyond the job's ~110)
    monkeypatch.delenv("NDGPU_OVL_FAIL_AT")
    again = job()
    assert all(np.array_equal(a, b) for a, b in zip(again[:2], want[:2])) and again[2] == want[2]
