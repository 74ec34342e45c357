"""`minimap2-nd --step 1 --mode 3` on the MI355X (HiFi: chain ends trimmed, every hit extended into the unaligned read ends,
minimap2/map.c:340-482, 919-928) against the `.ovl` files of the compiled reference, through the C ABI and through the command
line.  (Named to run after the other GPU tests.)"""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import test_gpu_overlap as GO  # noqa: E402
from make_overlap_golden import CASES_M3  # noqa: E402
from test_gpu_overlap import sets  # noqa: E402,F401  (fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", CASES_M3, ids=[c[0] for c in CASES_M3])
def test_mode3_ovl_bytes_match_reference_golden(sets, case):  # noqa: F811
    GO.test_ovl_bytes_match_reference_golden(sets, case)


@pytest.mark.parametrize("case", CASES_M3[:2], ids=[c[0] for c in CASES_M3[:2]])
def test_mode3_stage_cli_writes_reference_bytes(case, tmp_path, monkeypatch):
    monkeypatch.setenv("NDGPU_OVL_EXT_SCRATCH", "50000")   # several extension launches
    GO.test_stage_cli_writes_reference_bytes(case, tmp_path)


def test_mode3_extension_really_runs(sets):  # noqa: F811
    import numpy as np
    from nextdenovo_amd import overlap
    o = GO.dev_opt("ava-hifi", False, ("--mode", "3"))
    with overlap.Index(o, sets["hseed"][0]) as ix:
        recs = ix.map(sets["hseed"][0], ix.mid_occ())
        st = ix.stats()
    assert st["ext_problems"] > 1000 and st["ext_launches"] >= 1 and recs.size > 5000
    plain = GO.dev_opt("ava-hifi", False, ())
    with overlap.Index(plain, sets["hseed"][0]) as ix:
        base = ix.map(sets["hseed"][0], ix.mid_occ())
    assert recs.size > base.size and not np.array_equal(recs[:base.size], base)
