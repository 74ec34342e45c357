"""Host-side consensus logic of the product (consensus.cpp, poa.cpp, readdb.cpp) on CPU:
the engine runs with the oracle plugged in as alignment backend (tests/csrc/host_harness.cpp)
and must reproduce the reference's nextCorrect() byte for byte."""
import ctypes as C

import numpy as np
import pytest

import refpipe
import util


def test_golden_piles(host_harness):
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    piles = util.load_piles()
    assert len(piles) >= 10
    seen = set()
    for i, p in enumerate(piles):
        ln, ide, seq = util.call_correct(fn, fr, p)
        assert ln == p["exp_len"], i
        seen.add((p["read_type"], p["fast"], p["split"]))
        if ln > 4:
            assert seq == p["exp_seq"], i
            assert np.float32(ide) == np.float32(p["exp_ide"]), i
    assert {(1, 0, 0), (2, 0, 0), (3, 0, 0), (1, 1, 0), (1, 0, 1)} <= seen


def test_poa_golden(host_harness):
    host_harness.ndtest_poa.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int]
    for seqs, exp in util.load_poa():
        arr = (C.c_char_p * len(seqs))(*seqs)
        out = C.create_string_buffer(20000)
        n = host_harness.ndtest_poa(arr, len(seqs), out, 20000)
        assert n >= 0 and out.raw[:n] == exp


def test_edge_piles(host_harness):
    """Seed-only pile, tiny pile, pile below min_len_aln: same outcome codes as the reference
    conventions (len 2 = uncorrectable, lib/nextcorrect.c:1999)."""
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    p = util.load_piles()[0]
    solo = dict(p, seqs=p["seqs"][:1], aln_start=p["aln_start"][:1], aln_end=p["aln_end"][:1])
    ln, _, _ = util.call_correct(fn, fr, solo)
    assert ln == 2
    two = dict(p, seqs=p["seqs"][:2], aln_start=p["aln_start"][:2], aln_end=p["aln_end"][:2])
    ln, _, _ = util.call_correct(fn, fr, two)
    assert ln == 2
    ln, _, _ = util.call_correct(fn, fr, p, min_len_aln=10 ** 6)
    assert ln == 2


@pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so"), reason="compiled reference not present")
def test_edge_piles_vs_reference(host_harness):
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    rfn, rfr = util.bind_correct(refpipe.ref_cns())
    for p in util.load_piles()[:4]:
        for k in (1, 2, 5, 12):
            sub = dict(p, seqs=p["seqs"][:k], aln_start=p["aln_start"][:k], aln_end=p["aln_end"][:k])
            for fast in (0, 1):
                a = util.call_correct(rfn, rfr, sub, fast=fast)
                b = util.call_correct(fn, fr, sub, fast=fast)
                assert a[0] == b[0]
                if a[0] > 4:
                    assert a[2] == b[2] and np.float32(a[1]) == np.float32(b[1])


@pytest.mark.skipif(not refpipe.have_ref("nextcorrect.so", "minimap2-nd", "seq_dump", "ovl_sort", "ovlseq.so"),
                    reason="compiled reference chain not present")
def test_live_reference_chain(host_harness, tmp_path):
    """Fresh seeded data through the real reference stage chain, every pile compared."""
    from nextdenovo_amd import synth
    fn, fr = util.bind_correct(host_harness, "ndtest_correct", "ndtest_free")
    rfn, rfr = util.bind_correct(refpipe.ref_cns())
    g = synth.make_genome(30000, seed=21, n_repeats=0)
    rs = synth.simulate_reads(g, 30, "ont", seed=22, mu=8.0, sigma=0.4, min_len=1000)
    fa = str(tmp_path / "reads.fa")
    refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in rs.seqs])
    idxs, so = refpipe.run_overlap_chain(str(tmp_path), fa, seed_cutoff=2500)
    n = 0
    for seed, seqs, st, en, mal, recs in refpipe.read_piles(idxs, so, min_len_seed=1250):
        p = dict(seqs=seqs, aln_start=st, aln_end=en, max_aln=mal, max_lq=min(en[0] // 2, 10000), read_type=1,
                 fast=0, split=0)
        a = util.call_correct(rfn, rfr, p)
        b = util.call_correct(fn, fr, p)
        assert a[0] == b[0] and (a[0] <= 4 or (a[2] == b[2] and np.float32(a[1]) == np.float32(b[1]))), seed
        n += 1
    assert n > 20


def test_readdb_windows():
    """ReadDb (fwd + rc pools) reproduces getseq/subbit_ semantics (lib/bseq.c:241-255)."""
    from nextdenovo_amd import synth
    rng = np.random.default_rng(5)
    reads = [rng.integers(0, 4, int(n), dtype=np.uint8) for n in (1, 15, 16, 17, 33, 1000, 4097)]
    rs = synth.ReadSet()
    rs.seqs = reads
    words, off, lens = synth.pack_db(rs)
    # python model of the device packing: LSB-first words, fwd then rc per read
    for r, codes in enumerate(reads):
        w = words[int(off[r]):int(off[r]) + (codes.size + 15) // 16]
        back = np.stack([(w >> np.uint32(30 - 2 * i)) & np.uint32(3) for i in range(16)], axis=1).reshape(-1)
        assert np.array_equal(back[:codes.size].astype(np.uint8), codes)
        assert np.array_equal(synth.revcomp_codes(synth.revcomp_codes(codes)), codes)
