"""MI355X overlap sort / filter (ndgpu_ovl_sort, the `ovl_sort` step) and the whole correction stage chained on the
device: .2bit reads -> overlap -> sort -> consensus, against the reference chain's committed outputs."""
import gzip
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import mm_util as M  # noqa: E402
import os_util as O  # noqa: E402

pytestmark = pytest.mark.gpu
STAGE = os.path.join(HERE, "golden", "stage")


def _golden(name, gz=False):
    p = os.path.join(STAGE, name + (".gz" if gz else ""))
    with (gzip.open(p, "rb") if gz else open(p, "rb")) as f:
        return f.read()


def _device_raw(dual_pairs):
    """GPU overlap runs of the stage fixture: [(target, query, dual)] -> list of record arrays."""
    from nextdenovo_amd import overlap
    sets = {k: overlap.ReadSet.from_2bit(os.path.join(STAGE, "input.%s.001.2bit" % k)) for k in ("seed", "part")}
    out = []
    for t, q, dual in dual_pairs:
        o = overlap.preset("ava-ont")
        if dual:
            o.no_dual = 0
        with overlap.Index(o, sets[t]) as ix:
            out.append(ix.map(sets[q], ix.mid_occ()))
    return out


def test_device_overlap_then_sort_matches_reference_sorted_ovl():
    from nextdenovo_amd import overlap, ovl_sort
    files = _device_raw([("seed", "part", True), ("seed", "seed", False)])
    sl, mn = ovl_sort.read_idx(os.path.join(STAGE, ".input.seed.001.idx"))
    recs, bl, st = overlap.sort_overlaps(files, sl, mn, 40, 300)
    assert overlap.encode(recs, np.zeros(2, dtype=np.uint32)) == _golden("input.seed.001.sorted.ovl")
    assert "".join("%d %s\n" % x for x in bl).encode() == _golden("input.seed.001.sorted.ovl.bl")
    assert st["seeds"] > 10 and st["kept"] == recs.size


@pytest.mark.parametrize("k", [40, 18])
def test_sort_matches_oracle_on_chimeric_set(oracle_lib, k):
    """Deeper set with glued (chimeric) reads: every admission / trimming branch, against the sort oracle."""
    from nextdenovo_amd import overlap, synth
    olib = O.bind(oracle_lib)
    rng = np.random.default_rng(4)
    g = synth.make_genome(120000, seed=45, n_repeats=5, repeat_len=2500)
    rs = synth.simulate_reads(g, 55, "ont", seed=46)
    seqs = list(rs.seqs)
    for t in range(30):
        a, b = rng.integers(0, len(seqs), 2)
        y = synth.revcomp_codes(seqs[b]) if t % 2 else seqs[b]
        seqs.append(np.concatenate([seqs[a][: max(1500, seqs[a].size // 2)], y[: max(1500, y.size // 2)]]))
    n = len(seqs)
    ids = np.arange(n, dtype=np.uint32)
    lens = np.asarray([s.size for s in seqs], dtype=np.uint32)
    words = [synth.pack_2bit_msb(s) for s in seqs]
    woff = np.zeros(n, dtype=np.uint64)
    woff[1:] = np.cumsum([w.size for w in words])[:-1]
    allr = overlap.ReadSet(ids, lens, np.concatenate(words), woff)
    is_seed = lens >= 7000
    order = np.concatenate([np.flatnonzero(is_seed), np.flatnonzero(~is_seed)])
    seeds = overlap.ReadSet(ids[is_seed], lens[is_seed], allr.words, woff[is_seed])
    parts = overlap.ReadSet(ids[~is_seed], lens[~is_seed], allr.words, woff[~is_seed])
    files = []
    for q, dual in ((parts, True), (seeds, False)):
        o = overlap.preset("ava-ont")
        o.no_dual = 0 if dual else 1
        with overlap.Index(o, seeds) as ix:
            files.append(ix.map(q, ix.mid_occ()))
    seed_len = np.where(is_seed, lens, 0).astype(np.uint32)
    mn = int(lens[is_seed].min())
    recs, bl, st = overlap.sort_overlaps(files, seed_len, mn, k, 300)
    raws = [np.stack([f[c] for c in ("qname", "rev", "qs", "qe", "tname", "ts", "te", "match")], axis=1) for f in files]
    want, want_bl, _ = O.oracle_sort(olib, raws, seed_len, mn, max_bin_cov=k)
    assert len(want) > 50000
    assert overlap.encode(recs, np.zeros(2, dtype=np.uint32)) == want
    assert "".join("%d %s\n" % x for x in bl) == want_bl
    assert any(kind == "c" for _, kind in bl)


def test_whole_stage_on_device_from_2bit_to_cns_fasta(tmp_path):
    """The three stage command lines chained (raw_align x2 -> sort_align -> seed_cns), all on the MI355X engines:
    `cns.fasta` / `.idx` equal what the reference chain wrote."""
    from nextdenovo_amd import minimap2_nd, ovl_sort
    d = str(tmp_path / "w")
    shutil.copytree(STAGE, d)
    seed, part = os.path.join(d, "input.seed.001.2bit"), os.path.join(d, "input.part.001.2bit")
    o0, o1 = os.path.join(d, "raw0.ovl"), os.path.join(d, "raw1.ovl")
    assert minimap2_nd.run(["--step", "1", "--dual=yes", "-t", "8", "-x", "ava-ont", seed, part, "-o", o0]) == 0
    assert minimap2_nd.run(["--step", "1", "-I", "3G", "-t", "8", "-x", "ava-ont", seed, seed, "-o", o1]) == 0
    fofn = os.path.join(d, "ovl.fofn")
    with open(fofn, "w") as f:
        f.write(o0 + "\n" + o1 + "\n")
    so = os.path.join(d, "mine.sorted.ovl")
    assert ovl_sort.run(["-m", "2g", "-t", "4", "-k", "40", "-i", os.path.join(d, ".input.seed.001.idx"), "-o", so, fofn]) == 0
    assert open(so, "rb").read() == _golden("input.seed.001.sorted.ovl")
    assert open(so + ".bl", "rb").read() == _golden("input.seed.001.sorted.ovl.bl")
    idxs = os.path.join(d, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(d)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(d, n) + "\n")
    out = os.path.join(d, "cns.fasta")
    cmd = [sys.executable, "-m", "nextdenovo_amd.nextcorrect", "-f", idxs, "-i", so, "-r", "ont", "-p", "4", "-min_len_seed", "1250",
           "-o", out]
    r = subprocess.run(cmd, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(out, "rb").read() == _golden("cns.default.fasta", gz=True)
    assert open(out + ".idx", "rb").read() == _golden("cns.default.fasta.idx", gz=True)


@pytest.mark.skipif(not all(os.path.exists(os.path.join(O.REFDIR, n)) for n in ("minimap2-nd", "ovl_sort", "seq_dump")),
                    reason="oracle/_ref not built")
def test_workload_scale_chain_matches_reference_binaries(tmp_path):
    """A 0.8 Mb / 45x ONT set (2.7k reads, 36 Mb: the read-length mix of config 2): the compiled reference's
    minimap2-nd --step 1 and ovl_sort (many threads) against the device's, file for file."""
    from nextdenovo_amd import minimap2_nd, ovl_sort, synth
    g = synth.make_genome(800000, seed=91)
    rs = synth.simulate_reads(g, 45, "ont", seed=92)
    wd = str(tmp_path)
    seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in rs.seqs], seed_cutoff=12000)
    idx = os.path.join(wd, "db", ".input.seed.001.idx")
    jobs = [(seed, part, True, "0"), (seed, seed, False, "1")]
    ref_files, my_files = [], []
    for t, q, dual, tag in jobs:
        r = os.path.join(wd, "ref.%s.ovl" % tag)
        M.ref_step1(t, q, r, "ava-ont", dual, threads=32)
        m = os.path.join(wd, "mine.%s.ovl" % tag)
        argv = ["--step", "1"] + (["--dual=yes"] if dual else []) + ["-t", "8", "-x", "ava-ont", t, q, "-o", m]
        assert minimap2_nd.run(argv) == 0
        assert os.path.getsize(r) > 100000 and open(m, "rb").read() == open(r, "rb").read()
        ref_files.append(r)
        my_files.append(m)
    want, want_bl = O.ref_sort(wd, idx, ref_files, k=40, threads=8)
    fofn = os.path.join(wd, "mine.fofn")
    with open(fofn, "w") as f:
        f.write("\n".join(my_files) + "\n")
    so = os.path.join(wd, "mine.sorted.ovl")
    assert ovl_sort.run(["-m", "2g", "-t", "8", "-k", "40", "-i", idx, "-o", so, fofn]) == 0
    assert len(want) > 500000 and open(so, "rb").read() == want
    assert open(so + ".bl").read() == want_bl


@pytest.mark.skipif(not all(os.path.exists(os.path.join(O.REFDIR, n)) for n in ("minimap2-nd", "ovl_sort", "seq_dump", "nextcorrect.so",
                                                                                "ovlseq.so")), reason="oracle/_ref not built")
def test_hifi_stage_chain_matches_reference(tmp_path):
    """HiFi reads through the whole correction stage as nextDenovo runs it for `read_type = hifi` (raw_align with
    `-x ava-hifi -f seed_depth*20`, lib/config_parser.py:46-47,212; sort_align; seed_cns with `-r hifi -max_lq_length 1000`):
    the compiled reference programs against the three device command lines -- raw .ovl files, sorted.ovl, .bl and every
    corrected record."""
    import refpipe
    import util
    from nextdenovo_amd import minimap2_nd, ovl_sort, synth
    g = synth.make_genome(90000, seed=31, n_repeats=3, repeat_len=3000)
    rs = synth.simulate_reads(g, 30, "hifi", seed=32, mu=9.0, sigma=0.25, min_len=4000)
    wd = str(tmp_path)
    fa = os.path.join(wd, "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    idxs, so_ref = refpipe.run_overlap_chain(wd, fa, seed_cutoff=8500, preset="ava-hifi", extra=("-f", "800"), sort_depth=28)
    db, ra = os.path.join(wd, "db"), os.path.join(wd, "ra")
    seed, part = os.path.join(db, "input.seed.001.2bit"), os.path.join(db, "input.part.001.2bit")
    mine = []
    for t, q, dual, tag in ((seed, part, True, "0"), (seed, seed, False, "1")):
        ref_file = os.path.join(ra, "input.seed.001.2bit.%s.ovl" % tag)
        if not os.path.exists(ref_file):
            continue
        m = os.path.join(wd, "mine.%s.ovl" % tag)
        argv = ["--step", "1"] + (["--dual=yes"] if dual else ["-I", "3G"]) + ["-t", "8", "-x", "ava-hifi", "-f", "800", t, q, "-o", m]
        assert minimap2_nd.run(argv) == 0
        assert os.path.getsize(ref_file) > 20000 and open(m, "rb").read() == open(ref_file, "rb").read()
        mine.append(m)
    assert len(mine) == 2
    fofn = os.path.join(wd, "mine.fofn")
    with open(fofn, "w") as f:
        f.write("\n".join(mine) + "\n")
    so = os.path.join(wd, "mine.sorted.ovl")
    assert ovl_sort.run(["-m", "2g", "-t", "4", "-k", "28", "-i", os.path.join(db, ".input.seed.001.idx"), "-o", so, fofn]) == 0
    assert open(so, "rb").read() == open(so_ref, "rb").read()
    assert open(so + ".bl").read() == open(so_ref + ".bl").read()
    bl = {int(line.split()[0]) for line in open(so_ref + ".bl") if line.strip()}
    # reference consensus of every pile nextcorrect.py would form (lib/nextcorrect.py:92-143,183-199,236)
    rfn, rfr = util.bind_correct(refpipe.ref_cns())
    want = {}
    for seed_name, seqs, st, en, mal, _ in refpipe.read_piles(idxs, so_ref, min_len_seed=4250, blacklist=bl):
        a = util.call_correct(rfn, rfr, dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=min(en[0] // 2, 1000),
                                             read_type=3, fast=0, split=0))
        if a[0] >= 4250 and a[1] >= 0.8:
            want[int(seed_name)] = (a[0], "%f" % a[1], a[2])
    assert len(want) > 30
    out = os.path.join(wd, "cns.fasta")
    cmd = [sys.executable, "-m", "nextdenovo_amd.nextcorrect", "-f", idxs, "-i", so, "-r", "hifi", "-p", "4", "-max_lq_length", "1000",
           "-min_len_seed", "4250", "-o", out]
    r = subprocess.run(cmd, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = {}
    with open(out, "rb") as f:
        lines = f.read().split(b"\n")
    for h, s_ in zip(lines[0::2], lines[1::2]):
        name, ln, ide = h[1:].split()
        got[int(name)] = (int(ln), ide.decode(), s_)
    assert got == want


def test_fused_stage_equals_file_chain_golden(tmp_path):
    """`correct_stage` (overlap -> sort -> consensus with nothing on disk in between) on the stage fixture: the same
    cns.fasta / .idx the reference chain wrote through its files; --keep reproduces the intermediate files too."""
    from nextdenovo_amd import correct_stage
    keep = str(tmp_path / "keep")
    out = str(tmp_path / "cns")
    assert correct_stage.run(["-d", STAGE, "-x", "ava-ont", "-k", "40", "-r", "ont", "-min_len_seed", "1250", "-p", "4",
                              "--keep", keep, "-o", out]) == 0
    assert open(out + ".001.fasta", "rb").read() == _golden("cns.default.fasta", gz=True)
    assert open(out + ".001.fasta.idx", "rb").read() == _golden("cns.default.fasta.idx", gz=True)
    assert open(os.path.join(keep, "input.seed.001.sorted.ovl"), "rb").read() == _golden("input.seed.001.sorted.ovl")
    assert open(os.path.join(keep, "input.seed.001.sorted.ovl.bl"), "rb").read() == _golden("input.seed.001.sorted.ovl.bl")


@pytest.mark.skipif(not all(os.path.exists(os.path.join(O.REFDIR, n)) for n in ("minimap2-nd", "ovl_sort", "seq_dump")),
                    reason="oracle/_ref not built")
def test_fused_stage_two_seed_files_matches_reference_programs(tmp_path):
    """Two seed files + a part file: the five raw_align jobs of nextDenovo:426-467 (the seed1 x seed2 job serves both
    sorts through its symlink), one sort and one consensus run per seed file.  The fused command's --keep files against
    the compiled reference programs run job by job, its cns files against the file-based device command on the
    reference's sorted.ovl."""
    import refpipe
    from nextdenovo_amd import correct_stage, synth
    g = synth.make_genome(70000, seed=41, n_repeats=3, repeat_len=1500)
    rs = synth.simulate_reads(g, 32, "ont", seed=42, mu=8.7, sigma=0.45, min_len=900)
    wd = str(tmp_path)
    fa = os.path.join(wd, "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    fofn = os.path.join(wd, "input.fofn")
    with open(fofn, "w") as f:
        f.write(fa + "\n")
    db = os.path.join(wd, "db")
    os.makedirs(db)
    R = lambda n: os.path.join(O.REFDIR, n)  # noqa: E731
    refpipe.run([R("seq_dump"), "-f", "500", "-s", "5000", "-b", "2g", "-n", "2", "-d", db, fofn])
    s1, s2, p1 = (os.path.join(db, n) for n in ("input.seed.001.2bit", "input.seed.002.2bit", "input.part.001.2bit"))
    assert os.path.getsize(s2) > 1000 and os.path.getsize(p1) > 1000
    ra = os.path.join(wd, "ra")
    os.makedirs(ra)
    jobs = [(0, s1, p1, True, None), (1, s1, s1, False, "3G"), (2, s1, s2, True, "3G"), (3, s2, p1, True, None), (4, s2, s2, False, "3G")]
    ref_ovl = {}
    for k, t, q, dual, batch in jobs:
        o = os.path.join(ra, "%s.%d.ovl" % (os.path.basename(t), k))
        cmd = [R("minimap2-nd"), "--step", "1"] + (["-I", batch] if batch else []) + (["--dual=yes"] if dual else []) + \
            ["-t", "8", "-x", "ava-ont", t, q, "-o", o]
        refpipe.run(cmd)
        ref_ovl[k] = o
    keep = os.path.join(wd, "keep")
    out = os.path.join(wd, "cns")
    assert correct_stage.run(["-d", db, "-x", "ava-ont", "-k", "30", "-r", "ont", "-min_len_seed", "2500", "-p", "4", "--keep", keep,
                              "-o", out]) == 0
    for k, t, q, dual, batch in jobs:
        mine = os.path.join(keep, os.path.basename(ref_ovl[k]))
        assert os.path.getsize(ref_ovl[k]) > 5000 and open(mine, "rb").read() == open(ref_ovl[k], "rb").read(), "job %d" % k
    idxs = os.path.join(wd, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(db)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(db, n) + "\n")
    for tag, ks in (("001", (0, 1, 2)), ("002", (2, 3, 4))):
        with open(os.path.join(ra, "in%s.fofn" % tag), "w") as f:
            f.write("\n".join(ref_ovl[k] for k in ks) + "\n")
        refpipe.run([R("ovl_sort"), "-m", "2g", "-t", "4", "-k", "30", "-i", os.path.join(db, ".input.seed.%s.idx" % tag),
                     "-o", "ref.%s.sorted.ovl" % tag, "in%s.fofn" % tag], cwd=ra)
        so_ref = os.path.join(ra, "ref.%s.sorted.ovl" % tag)
        so = os.path.join(keep, "input.seed.%s.sorted.ovl" % tag)
        assert os.path.getsize(so_ref) > 10000 and open(so, "rb").read() == open(so_ref, "rb").read()
        assert open(so + ".bl").read() == open(so_ref + ".bl").read()
        file_out = os.path.join(wd, "file.%s.fasta" % tag)
        cmd = [sys.executable, "-m", "nextdenovo_amd.nextcorrect", "-f", idxs, "-i", so_ref, "-r", "ont", "-p", "4", "-min_len_seed", "2500",
               "-o", file_out]
        r = subprocess.run(cmd, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        assert os.path.getsize(file_out) > 50000
        assert open(out + ".%s.fasta" % tag, "rb").read() == open(file_out, "rb").read()
        assert open(out + ".%s.fasta.idx" % tag, "rb").read() == open(file_out + ".idx", "rb").read()


def test_fused_stage_from_fasta(tmp_path):
    """`correct_stage --fofn`: db_split (seq_dump with the device packing) + the fused stage in one command == the fused stage on
    a directory the same reads were dumped into beforehand."""
    from nextdenovo_amd import correct_stage, seq_dump, synth
    import refpipe
    g = synth.make_genome(40000, seed=51, n_repeats=2, repeat_len=1200)
    rs = synth.simulate_reads(g, 30, "ont", seed=52, mu=8.4, sigma=0.4, min_len=700)
    wd = str(tmp_path)
    fa = os.path.join(wd, "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    fofn = os.path.join(wd, "input.fofn")
    with open(fofn, "w") as f:
        f.write("reads.fa\n")
    common = ["-x", "ava-ont", "-k", "25", "-r", "ont", "-min_len_seed", "2000", "-p", "4"]
    a, b = os.path.join(wd, "a"), os.path.join(wd, "b")
    assert correct_stage.run(["--fofn", fofn, "--read-cutoff", "500", "--seed-cutoff", "4k", "--seed-cutfiles", "2", "-d", a, "-o",
                              os.path.join(wd, "one")] + common) == 0
    assert seq_dump.run(["-f", "500", "-s", "4k", "-b", "0", "-n", "2", "-d", b, fofn]) == 0
    assert correct_stage.run(["-d", b, "-o", os.path.join(wd, "two")] + common) == 0
    for tag in ("001", "002"):
        x = open(os.path.join(wd, "one.%s.fasta" % tag), "rb").read()
        assert len(x) > 20000 and x == open(os.path.join(wd, "two.%s.fasta" % tag), "rb").read()


@pytest.mark.skipif(not all(os.path.exists(os.path.join(O.REFDIR, n)) for n in ("minimap2-nd", "ovl_sort", "seq_dump")),
                    reason="oracle/_ref not built")
@pytest.mark.parametrize("profile,preset,k,flank", [("hifi", "ava-hifi", 28, None), ("hifi", "ava-hifi", 6, 120), ("ont", "ava-ont", 30, None)])
def test_sort_hq_variant_matches_reference(tmp_path, profile, preset, k, flank):
    """`ovl_sort -H` on the device (ndgpu_ovl_sort_hq through the stage command line) against the compiled reference
    `ovl_sort -H`: sorted.ovl and .bl, byte for byte (HiFi reads, and noisy reads with chimeras for the 'k' path)."""
    from test_ovlsort_oracle import _hq_inputs
    from nextdenovo_amd import ovl_sort
    wd, files, idx = _hq_inputs(profile, preset, 14, 8500 if profile == "hifi" else 7000, extra=("-f", "700") if profile == "hifi" else ())
    want, want_bl = O.ref_sort(wd, idx, files, k=k, flank=flank, hq=True)
    fofn = os.path.join(wd, "mine.fofn")
    with open(fofn, "w") as f:
        f.write("\n".join(files) + "\n")
    so = os.path.join(wd, "mine.sorted.ovl")
    argv = ["-H", "-m", "2g", "-t", "4", "-k", str(k)] + (["-l", str(flank)] if flank else []) + ["-i", idx, "-o", so, fofn]
    assert ovl_sort.run(argv) == 0
    assert len(want) > 10000 and open(so, "rb").read() == want
    assert open(so + ".bl").read() == want_bl


def _random_step1_files(seed=3, n_files=5, per_file=2500, n_ids=400):
    """Step-1 records nobody mapped: random but well-formed (spans inside the reads, both directions), 45 % of the ids are not
    seeds -- so the "5 misses then stop" rule of a file (util/ovl_sort.c:975-1003) cuts in at different records for the two sides."""
    from nextdenovo_amd import overlap
    rng = np.random.default_rng(seed)
    lens = rng.integers(3000, 30000, n_ids).astype(np.uint32)
    seed_len = np.where(rng.random(n_ids) < 0.55, lens, 0).astype(np.uint32)
    files = []
    for f in range(n_files):
        r = np.zeros(per_file, dtype=overlap.REC)
        # early records of a file hit seeds more often, so that the fifth miss falls somewhere in the middle of it
        pool_hit, pool_any = np.flatnonzero(seed_len), np.arange(n_ids)
        for k in range(per_file):
            q = int(rng.choice(pool_hit if rng.random() < (0.97 if k < per_file // 2 else 0.6) else pool_any))
            t = int(rng.choice(pool_hit if rng.random() < (0.97 if k < per_file // 3 else 0.6) else pool_any))
            ql, tl = int(lens[q]), int(lens[t])
            span = int(rng.integers(300, min(ql, tl) - 10))
            qs, ts = int(rng.integers(0, ql - span)), int(rng.integers(0, tl - span))
            r[k] = (int(rng.integers(0, 2)), q, qs, qs + span, t, ts, ts + span + int(rng.integers(-20, 20)) if ts + span + 20 < tl else ts + span,
                    int(span * rng.uniform(0.2, 0.9)))
        files.append(r)
    return files, seed_len, int(lens[seed_len > 0].min())


def check_sort_in_seed_ranges_equals_the_sort_at_once(monkeypatch, hq):   # (run from tests/test_zzz_gpu_cigar.py and tests/test_simt_overlap.py)
    """The out-of-core form of the sort (raw records through the device in pieces, seeds in id ranges: what `ovl_sort -m` smaller than
    the data does with temporary files) == the sort at once: records, order and `.bl`, with pieces that cut files where the miss
    counters stand between 0 and 5, and on the stage fixture's own overlaps."""
    from nextdenovo_amd import overlap, ovl_sort
    cases = [_random_step1_files() + (40,)]
    if not hq:
        sl, mn = ovl_sort.read_idx(os.path.join(STAGE, ".input.seed.001.idx"))
        cases.append((_device_raw([("seed", "part", True), ("seed", "seed", False)]), sl, mn, 40))
    for files, seed_len, mn, k in cases:
        monkeypatch.delenv("NDGPU_OVLSORT_PIECE_RECORDS", raising=False)
        monkeypatch.delenv("NDGPU_OVLSORT_RANGE_CANDIDATES", raising=False)
        recs, bl, st = overlap.sort_overlaps(files, seed_len, mn, k, 300, hq=hq)
        assert st["ranges"] == 1 and st["kept"] == recs.size and recs.size > 100
        for piece, rng_cap in ((257, 900), (1000000, 2000), (61, 10 ** 9)):
            monkeypatch.setenv("NDGPU_OVLSORT_PIECE_RECORDS", str(piece))
            monkeypatch.setenv("NDGPU_OVLSORT_RANGE_CANDIDATES", str(rng_cap))
            r2, bl2, st2 = overlap.sort_overlaps(files, seed_len, mn, k, 300, hq=hq)
            assert np.array_equal(r2, recs) and bl2 == bl
            assert (st2["candidates"], st2["seeds"], st2["kept"]) == (st["candidates"], st["seeds"], st["kept"])
            assert st2["ranges"] > (1 if rng_cap < 10 ** 9 else 0)
