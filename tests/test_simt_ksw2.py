"""The ksw2 kernel's LOGIC without a GPU: csrc/ksw2_kernels.hip under the lane-accurate interpreter (tests/simt) against the vectors
of the compiled reference function and the oracle -- the bodies of tests/test_zz_gpu_ksw2.py."""
import ctypes as C
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "simt"))
import test_zz_gpu_ksw2 as G  # noqa: E402


@pytest.fixture(scope="module")
def simt_ovl():
    import build_simt
    return C.CDLL(build_simt.build_overlap())


@pytest.mark.parametrize("schedule", [0, 1, 6])
def test_reference_vectors_single_calls(simt_ovl, schedule):
    simt_ovl.simt_set_schedule(schedule)
    G.check_golden_single_calls(simt_ovl, stride=1 if schedule == 0 else 3)
    simt_ovl.simt_set_schedule(0)


def test_reference_vectors_one_batch(simt_ovl):
    G.check_golden_batch(simt_ovl)


def test_long_targets(simt_ovl, oracle_lib):
    G.check_long(simt_ovl, oracle_lib)


def test_large_lds_tier(simt_ovl):
    """A sample of the 1,025 .. 4,096-base targets (the kernel's large LDS tier) through the interpreter."""
    G.check_mid(simt_ovl, stride=17)


def test_local_alignment_score_of_the_inversion_test(simt_ovl, oracle_lib):
    """ksw_ll_kernel through the interpreter: a sample of the compiled reference's ksw_ll_i16 vectors and of oracle-checked problems."""
    G.check_ll(simt_ovl, oracle_lib, stride=4)
