"""`minimap2-nd --step 2 --mode 0` on the MI355X (corrected reads: per-target marking and the record filters in K5, the dovetail /
contained filter, the 10-field encoder and the `.bl` table on the host) through the command line, against the `.ovl` / `.bl` files of
the compiled reference (tests/golden/step2, written by tests/golden/make_step2_golden.py).  (Named to run after the other GPU
tests.)"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_step2_golden import CASES, CASES_M1, CASES_M2, CASES_RECHAIN, CASES_THIN, OUT as GOLD, files_of  # noqa: E402

pytestmark = pytest.mark.gpu


def run_case(tag, argv, tmp_path):
    from nextdenovo_amd import minimap2_nd
    out = str(tmp_path / "o.ovl")
    # `.m2`: the command as nextDenovo writes it (no --mode: the re-alignment); `.m1`: --mode 1
    mode = () if tag.endswith(".m2") else ("--mode", "1") if tag.endswith(".m1") else ("--mode", "0")
    assert minimap2_nd.run(["--step", "2", *mode, "-t", "3", *argv, *[os.path.join(GOLD, f) for f in files_of(tag)], "-o", out]) == 0
    with open(os.path.join(GOLD, tag + ".ovl"), "rb") as f:
        want = f.read()
    with open(out, "rb") as f:
        got = f.read()
    assert want[:2] == b"\x00\xff" and len(want) > 300
    assert got == want
    with open(os.path.join(GOLD, tag + ".ovl.bl")) as f, open(out + ".bl") as g:
        assert g.read() == f.read()


_ALL = CASES + CASES_M2 + CASES_M1 + CASES_RECHAIN + CASES_THIN   # (tandem.m1: mappings beyond 100,000 anchors -- the anchor thinning of mm_chain_dp_nextdenovo)


@pytest.mark.parametrize("tag,argv", _ALL, ids=[c[0] for c in _ALL])
def test_step2_cli_writes_reference_bytes(tag, argv, tmp_path):
    run_case(tag, argv, tmp_path)


def test_step2_records_and_verdicts(tmp_path):
    """Through the C ABI: the device's 10-field records carry both read lengths and an identity <= 10000, and the host filter
    drops some of them (the fixture has contained reads)."""
    from nextdenovo_amd import minimap2_nd, overlap
    a = minimap2_nd.load_reads(os.path.join(GOLD, "a.fa.gz"))
    b = minimap2_nd.load_reads(os.path.join(GOLD, "b.fa.gz"))
    assert int(a.ids[0]) == 1 and np.all(np.diff(a.ids.astype(np.int64)) == 1)   # names are numbers, in file order
    opt = minimap2_nd.build_opt(minimap2_nd.parse_argv("--step 2 --mode 0 --dual=yes -x ava-ont -k 17 -w 17 --minlen 1000 --maxhan1 2000 a b -o x".split()))
    with overlap.Index(opt, a) as ix, overlap.Step2Filter() as flt:
        recs = ix.map2(b, ix.mid_occ())
        blob, kept = flt.feed(recs, opt.maxhan1, opt.maxhan2, want_verdicts=True)
        bl = flt.bl()
    assert recs.size > 50 and 0 < int(kept.sum()) < recs.size
    lens = dict(zip(a.ids.tolist(), a.lens.tolist()))
    assert all(lens[int(t)] == int(l) for t, l in zip(recs["tname"], recs["tlen"]))
    assert int(recs["identity"].max()) <= 10000 and np.all(recs["qe"] > recs["qs"])
    assert len(blob) > 100 and bl.count("\n") > 10
    with pytest.raises(RuntimeError):
        overlap.Index(opt, a).map(b, 10)   # 8-field entry refuses --step 2 options
