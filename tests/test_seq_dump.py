"""`seq_dump` drop-in (nextdenovo_amd/seq_dump.py): the kseq-faithful parser on the CPU, the whole command (2-bit packing on the
device through `ndgpu_pack_2bit`) against the compiled reference `seq_dump` on the GPU box."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REFDIR = os.path.join(os.path.dirname(HERE), "oracle", "_ref")


def _make_inputs(d):
    """Three input files that exercise the parser: multi-line FASTA with lower case / N / CRLF / empty lines, FASTQ whose
    quality lines start with '@' and '+', a gzip'd FASTA; lengths on both sides of the two cut-offs."""
    rng = np.random.default_rng(5)

    def seq(n, alphabet="ACGT"):
        return "".join(rng.choice(list(alphabet), n))

    recs1 = [seq(n) for n in (50, 700, 1200, 1203, 5000, 4999, 2333, 16, 17, 9000, 640, 15999)]
    recs1[3] = recs1[3][:400] + "NNNN" + recs1[3][404:600].lower() + "RYK" + recs1[3][603:]
    recs1[5] = "N" + recs1[5][1:-1] + "N"
    p1 = os.path.join(d, "a.fa")
    with open(p1, "w", newline="") as f:
        f.write("junk before the first header\n")
        for i, s in enumerate(recs1):
            f.write(">r%d some comment\n" % i)
            w = (60, 80, 10 ** 6)[i % 3]
            for k in range(0, len(s), w):
                f.write(s[k:k + w] + ("\r\n" if i % 4 == 1 else "\n"))
            if i % 5 == 0:
                f.write("\n")
    recs2 = [seq(n) for n in (800, 3100, 450, 7000, 2999, 3000)]
    p2 = os.path.join(d, "b.fq")
    with open(p2, "w") as f:
        for i, s in enumerate(recs2):
            q = "".join(rng.choice(list("@+>IJK#5"), len(s)))
            q = ("@" if i % 2 == 0 else "+") + q[1:]
            f.write("@q%d\n%s\n+q%d\n" % (i, s, i))
            if i == 3:   # quality over two lines
                f.write(q[:100] + "\n" + q[100:] + "\n")
            else:
                f.write(q + "\n")
    recs3 = [seq(n, "ACGTacgtN") for n in (2500, 999, 1000, 60000)]
    p3 = os.path.join(d, "c.fa.gz")
    with gzip.open(p3, "wt") as f:
        for i, s in enumerate(recs3):
            f.write(">g%d\n%s\n" % (i, s))
    fofn = os.path.join(d, "input.fofn")
    with open(fofn, "w") as f:
        f.write("# comment line\n%s\nb.fq\n\nc.fa.gz\n" % p1)   # absolute and fofn-relative paths
    return fofn, [recs1, recs2, recs3]


def test_parser_follows_kseq(tmp_path):
    from nextdenovo_amd import seq_dump
    fofn, recs = _make_inputs(str(tmp_path))
    for name, want in zip(("a.fa", "b.fq", "c.fa.gz"), recs):
        buf, got = seq_dump.read_records(str(tmp_path / name))
        assert [l for _, l in got] == [len(s) for s in want]
        for (s0, l), s in zip(got, want):
            assert buf[s0:s0 + l].tobytes().decode() == s
    # quality lines are read until they cover the sequence, whatever they start with; a record whose quality stays short
    # ends the file (kseq_read returns -2)
    bad = tmp_path / "bad.fq"
    bad.write_text("@a\nACGT\n+\nIIII\n@b\nACGTACGT\n+\nIII\n@c\nAC\n+\nII\n")
    _, got = seq_dump.read_records(str(bad))
    assert [l for _, l in got] == [4, 8]          # "III" + "@c" + "AC" + "+" = 8 quality characters
    bad.write_text("@a\nACGT\n+\nIIII\n@b\nACGTACGT\n+\nIII\n")
    _, got = seq_dump.read_records(str(bad))
    assert [l for _, l in got] == [4]
    assert seq_dump.parse_num("1k") == 1000 and seq_dump.parse_num("2.5g") == 2500000000 and seq_dump.parse_num("750") == 750


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "seq_dump")), reason="oracle/_ref not built")
def test_parser_lengths_match_reference_idx(tmp_path):
    """The compiled reference seq_dump on the same files: ids, offsets and lengths of its .idx files (no device needed)."""
    from nextdenovo_amd import seq_dump
    fofn, _ = _make_inputs(str(tmp_path))
    ref = tmp_path / "ref"
    subprocess.run([os.path.join(REFDIR, "seq_dump"), "-f", "1k", "-s", "3k", "-b", "6k", "-n", "2", "-d", str(ref), fofn], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lens = {}
    for n in os.listdir(ref):
        if n.endswith(".idx"):
            for line in open(ref / n):
                i, _, ln = line.split("\t")
                lens[int(i)] = int(ln)
    mine = []
    for name in ("a.fa", "b.fq", "c.fa.gz"):
        _, got = seq_dump.read_records(str(tmp_path / name))
        mine += [l for _, l in got if 1000 <= l < 1000000]
    assert [lens[i] for i in sorted(lens)] == mine and len(mine) >= 12


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "seq_dump")), reason="oracle/_ref not built")
@pytest.mark.parametrize("argv", [("-f", "1k", "-s", "3k", "-b", "6k", "-n", "2"), ("-f", "500", "-s", "1001", "-b", "0", "-n", "3"),
                                  ("-f", "16", "-s", "2.5k", "-b", "1g", "-n", "1")])
def test_seq_dump_files_equal_reference(tmp_path, argv):
    from nextdenovo_amd import seq_dump
    fofn, _ = _make_inputs(str(tmp_path))
    ref, mine = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([os.path.join(REFDIR, "seq_dump"), *argv, "-d", ref, fofn], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    assert seq_dump.run([*argv, "-d", mine, fofn]) == 0
    names = sorted(os.listdir(ref))
    assert names == sorted(os.listdir(mine)) and len(names) >= 4
    for n in names:
        assert open(os.path.join(ref, n), "rb").read() == open(os.path.join(mine, n), "rb").read(), n


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "seq_stat")), reason="oracle/_ref not built")
@pytest.mark.parametrize("argv", [("-f", "1k", "-g", "20k", "-d", "3"), ("-f", "500", "-g", "5k", "-d", "45"), ("-f", "16", "-g", "100k", "-d", "30", "-a"),
                                  ("-f", "2k", "-g", "1m", "-d", "40"), ()])
def test_seq_stat_report_equals_reference(tmp_path, argv):
    """The db_stat report (histogram, N-stat table, suggested seed cut-off) against the compiled reference seq_stat."""
    from nextdenovo_amd import seq_stat
    fofn, _ = _make_inputs(str(tmp_path))
    rng = np.random.default_rng(9)
    big = tmp_path / "many.fa"          # enough reads for several histogram bins and every N-stat row
    with open(big, "w") as f:
        for i, n in enumerate(np.clip(rng.lognormal(8.3, 0.6, 2600).astype(int), 30, 60000)):
            f.write(">m%d\n%s\n" % (i, "ACGT" * (int(n) // 4) + "A" * (int(n) % 4)))
    with open(fofn, "a") as f:
        f.write("many.fa\n")
    ref, mine = str(tmp_path / "ref.stat"), str(tmp_path / "mine.stat")
    subprocess.run([os.path.join(REFDIR, "seq_stat"), *argv, "-o", ref, fofn], check=True, stderr=subprocess.DEVNULL)
    assert seq_stat.run([*argv, "-o", mine, fofn]) == 0
    assert open(mine, "rb").read() == open(ref, "rb").read() and os.path.getsize(ref) > 500


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "seq_dump")), reason="compiled reference not built")
@pytest.mark.parametrize("n_seed,block", [(1, 0), (3, 0), (2, 20000)])
def test_in_memory_dealing_matches_reference_seq_dump(tmp_path, n_seed, block):
    """nextdenovo_amd.stage.deal (what bench.py --gpus N and the sharded stage use to give every rank its seed file) against the
    files the compiled reference seq_dump writes: which read id lands in which seed / part file (util/seq_dump.c:74-114)."""
    from nextdenovo_amd import stage
    rng = np.random.default_rng(11)
    lens = [int(x) for x in rng.integers(300, 9000, 60)] + [999, 1000, 1001, 2999, 3000, 3001]
    fa = str(tmp_path / "reads.fa")
    with open(fa, "w") as f:
        for i, n in enumerate(lens):
            f.write(">r%d\n%s\n" % (i, "".join(rng.choice(list("ACGT"), n))))
    fofn = str(tmp_path / "input.fofn")
    open(fofn, "w").write(fa + "\n")
    out = str(tmp_path / "out")
    os.makedirs(out)
    cmd = [os.path.join(REFDIR, "seq_dump"), "-f", "1000", "-s", "3000", "-n", str(n_seed), "-d", out]
    if block:
        cmd += ["-b", str(block)]
    subprocess.run(cmd + [fofn], check=True, capture_output=True)

    def ids_of(path):
        return [int(l.split()[0]) for l in open(path)] if os.path.exists(path) else []

    kept, seed_files, parts = stage.deal(np.asarray(lens), 1000, 3000, n_seed, block_size=block)
    assert kept.tolist() == [i for i, n in enumerate(lens) if n >= 1000]
    for k in range(n_seed):
        assert seed_files[k].tolist() == ids_of(os.path.join(out, ".input.seed.%03d.idx" % (k + 1))), k
    ref_parts = []
    k = 1
    while os.path.exists(os.path.join(out, ".input.part.%03d.idx" % k)):
        ref_parts.append(ids_of(os.path.join(out, ".input.part.%03d.idx" % k)))
        k += 1
    assert [p.tolist() for p in parts] == [p for p in ref_parts if p]


def test_native_reader_equals_python_parser_and_streams(tmp_path):
    """ndgpu_fastx_* (csrc/fastx_reader.cpp) against the Python restatement of kseq_read on the parser fixtures, on files built to
    hit the corners (CR-only lines, header-only records, '+' / '@' inside quality, truncated tails, junk before the first
    header, empty files), on a random mix -- and in small chunks: the records are the same whatever the chunk size, and a read
    longer than the chunk is handed over once the buffer has room."""
    from nextdenovo_amd import seq_dump
    fofn, _ = _make_inputs(str(tmp_path))
    cases = [str(tmp_path / n) for n in ("a.fa", "b.fq", "c.fa.gz")]
    rng = np.random.default_rng(8)
    texts = [b"", b"\n\n", b"no header at all\nACGT\n", b">only_header", b">h\n", b">h\n\n\nACGT\n\n>g\nAC\r\nGT\r\n", b">a\r\nA\r\n\r\nC\r\n",
             b"@q\nACGT\n+\nII\nII\n@r\nAC\n+\n@+\n", b"@q\nACGT\n+", b"@q\nACGT\n+\n", b"@q\nACGT\n+\nIII", b">a b c\tx\nAC GT\n>b\n@\n",
             b"junk\n>x\nAAAA\n+\nIIII\n>y\nCC\n", b">a\n" + b"ACGT" * 5000 + b"\n>b\nA\n", b">\nACG\n", b"@\nAC\n+\nII\n"]
    for k in range(12):
        parts = []
        for _ in range(int(rng.integers(1, 9))):
            n = int(rng.integers(0, 400))
            s = "".join(rng.choice(list("ACGTNacgt"), n))
            w = int(rng.integers(1, 90))
            nl = "\r\n" if rng.random() < 0.3 else "\n"
            body = nl.join(s[i:i + w] for i in range(0, n, w))
            if rng.random() < 0.5:
                parts.append(">r%d c\n%s%s" % (k, body, nl if rng.random() < 0.9 else ""))
            else:
                q = "".join(rng.choice(list("@+>I#5"), n if rng.random() < 0.85 else max(0, n - 1)))
                parts.append("@r%d\n%s%s+\n%s%s" % (k, body, nl, q, nl))
        texts.append("".join(parts).encode())
    for i, t in enumerate(texts):
        p = str(tmp_path / ("t%d.fx" % i))
        open(p, "wb").write(t)
        cases.append(p)
    ref_exe = os.path.join(REFDIR, "seq_dump")
    for ci, p in enumerate(cases):
        b0, r0 = seq_dump.read_records_py(p)
        want = [b0[s:s + l].tobytes() for s, l in r0]
        if os.path.exists(ref_exe):  # the compiled reference on the same file: the lengths its .idx lists, in order
            d = str(tmp_path / ("ref%d" % ci))
            f1 = str(tmp_path / ("one%d.fofn" % ci))
            open(f1, "w").write(p + "\n")
            subprocess.run([ref_exe, "-f", "1", "-s", "999999", "-n", "1", "-d", d, f1], check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
            ref_len = []
            k = 1
            while os.path.exists(os.path.join(d, ".input.part.%03d.idx" % k)):
                ref_len += [int(l.split("\t")[2]) for l in open(os.path.join(d, ".input.part.%03d.idx" % k))]
                k += 1
            assert ref_len == [len(w) for w in want if 1 <= len(w) < 999999], (p, ref_len)
        for cb, cr in ((1 << 20, 1 << 10), (257, 3), (64, 1)):
            got = []
            for buf, off, ln in seq_dump.iter_chunks(p, cb, cr):
                got += [buf[int(o):int(o) + int(l)].tobytes() for o, l in zip(off, ln)]
            assert got == want, (p, cb, cr, len(got), len(want))


@pytest.mark.skipif(not os.path.exists(os.path.join(REFDIR, "seq_dump")), reason="oracle/_ref not built")
@pytest.mark.parametrize("argv", [("-f", "1k", "-s", "3k", "-b", "6k", "-n", "2"), ("-f", "500", "-s", "1001", "-b", "0", "-n", "3")])
@pytest.mark.parametrize("n_files", [1, 5])
def test_command_file_layout_with_a_stand_in_packer(tmp_path, argv, monkeypatch, n_files):
    """The command's file logic -- ids, .idx offsets, seed dealing, part splitting, chunked reading -- without a GPU: the device
    packer is replaced by a numpy one (test-only; exact for ACGT reads) and the files are compared with the compiled reference's."""
    from nextdenovo_amd import overlap, seq_dump, synth
    rng = np.random.default_rng(21)
    paths = []
    for k_file in range(n_files):   # several files (some gzipped): read side by side by the reader threads, consumed in order
        fa = tmp_path / ("r%d.fa%s" % (k_file, ".gz" if k_file % 2 else ""))
        with (gzip.open(fa, "wt") if k_file % 2 else open(fa, "w")) as f:
            for i in range(40 if n_files == 1 else 12):
                n = int(rng.integers(200, 9000))
                s = "".join(rng.choice(list("ACGT"), n))
                f.write(">r%d_%d\n" % (k_file, i) + "\n".join(s[k:k + 70] for k in range(0, n, 70)) + "\n")
        paths.append(str(fa))
    fofn = tmp_path / "in.fofn"
    fofn.write_text("\n".join(paths) + "\n")

    def fake_pack(buf, a_off, lens):
        code = np.zeros(256, dtype=np.uint8)
        for ch, v in zip(b"ACGT", range(4)):
            code[ch] = v
        words, woff, at = [], [], 0
        for o, l in zip(a_off, lens):
            w = synth.pack_2bit_msb(code[buf[int(o):int(o) + int(l)]])
            words.append(w)
            woff.append(at)
            at += w.size
        return np.concatenate(words), np.asarray(woff, dtype=np.uint64)

    monkeypatch.setattr(overlap, "pack_2bit", fake_pack)
    monkeypatch.setattr(seq_dump, "CHUNK_BASES", 20000)   # several chunks per file
    monkeypatch.setattr(seq_dump, "CHUNK_RECS", 7)
    ref, mine = str(tmp_path / "ref"), str(tmp_path / "mine")
    subprocess.run([os.path.join(REFDIR, "seq_dump"), *argv, "-d", ref, str(fofn)], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    assert seq_dump.run([*argv, "-d", mine, str(fofn)]) == 0
    names = sorted(os.listdir(ref))
    assert names == sorted(os.listdir(mine)) and len(names) >= 4
    for n in names:
        assert open(os.path.join(ref, n), "rb").read() == open(os.path.join(mine, n), "rb").read(), n
