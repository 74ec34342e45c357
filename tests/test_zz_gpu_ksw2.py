"""The two-piece affine-gap extension kernel on the MI355X (csrc/ksw2_kernels.hip: `ksw_extd2_sse` with the reference's signature,
`ndgpu_ksw_extd2_batch`) against the vectors the compiled reference function produced (tests/golden/ksw2.npz) and against the
oracle on targets longer than the kernel's LDS budget.  (Named to run after the other GPU tests.)"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ksw_util as K  # noqa: E402
from test_oracle_ksw2 import golden, golden_mid  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ovl_lib():
    from nextdenovo_amd import overlap
    return overlap.load()


def check_golden_single_calls(lib, stride=1):
    ps, want = golden()
    for i in range(0, len(ps), stride):
        p = ps[i]
        got = K.call_sse(lib, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        assert got == want[i], i


def check_golden_batch(lib):
    ps, want = golden()
    got = K.call_batch(lib, ps)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, i


def check_long(lib, oracle_lib):
    ps = K.long_problems()
    got = K.call_batch(lib, ps)
    for i, p in enumerate(ps):
        want = K.call_oracle(oracle_lib, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"])
        assert got[i] == want, i
        assert want[0][0] > 1000 or want[0][1] or want[0][8] > 1000   # a real alignment (max or end-to-end score), or a z-drop


def test_ksw_extd2_sse_matches_reference_vectors(ovl_lib):
    check_golden_single_calls(ovl_lib)


def test_ksw_batch_matches_reference_vectors(ovl_lib):
    check_golden_batch(ovl_lib)


def test_ksw_long_targets_match_oracle(ovl_lib, oracle_lib, monkeypatch):
    check_long(ovl_lib, oracle_lib)
    monkeypatch.setenv("NDGPU_KSW_SCRATCH_GB", "0.001")   # sub-batches of one or two problems
    check_golden_batch(ovl_lib)


def check_mid(lib, stride=1):
    ps, want = golden_mid()
    ps, want = ps[::stride], want[::stride]
    got = K.call_batch(lib, ps)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, i
    for i in range(0, len(ps), 8 * stride if stride == 1 else 1):   # and through the single-call entry
        p = ps[i]
        assert K.call_sse(lib, p["q"], p["t"], p["mat"], *p["gaps"], p["w"], p["zdrop"], p["end_bonus"], p["flag"]) == want[i], i


def test_ksw_large_lds_tier_matches_reference_vectors(ovl_lib):
    """Targets of 1,025 .. 4,096 bases run in the kernel's large LDS tier (44 KB of dynamic LDS per wavefront)."""
    check_mid(ovl_lib)


def check_ll(lib, oracle_lib=None, stride=1):
    """ndgpu_ksw_ll_batch (ksw_ll_kernel: the SSE schedule of ksw_ll_i16, lanes 0..7 = the stripes) against the compiled reference's
    vectors, in one batch; with an oracle also on problems the vectors do not hold."""
    want = np.load(os.path.join(HERE, "golden", "ksw_ll.npz"))["res"]
    ps = K.ll_problems()
    idx = list(range(0, len(ps), stride))
    got = K.call_ll_batch(lib, [ps[i] for i in idx])
    for i, g in zip(idx, got):
        assert g == tuple(int(x) for x in want[i]), i
    if oracle_lib is not None:
        more = K.ll_problems(seed=23, n=300)[::stride]
        for p, g in zip(more, K.call_ll_batch(lib, more)):
            assert g == K.call_ll_oracle(oracle_lib, p)

