"""The alternative forms of the alignment kernels (csrc/ond_kernels.hip) against the defaults on the device -- NDGPU_K8A=wave: a
wavefront per alignment in the traceback; NDGPU_K7=pair: two alignments per wavefront in the forward pass -- the same read set
through the whole chain, records identical; the kernel times are printed for the A/B (`pytest -s`).  (Named to run after the other
GPU tests; the switches are read once per process, hence the child processes.)"""
import pytest

from test_gpu_robust import _driver

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def default_run():
    return _driver({})[0]


def test_wave_form_traceback_gives_identical_records(default_run):
    want = default_run
    got, _ = _driver({"NDGPU_K8A": "wave"})
    assert len(want["digests"]) > 50 and got["digests"] == want["digests"]
    print("traceback ms: lane-per-alignment %.1f, wavefront-per-alignment %.1f; forward %.1f / %.1f"
          % (want["stats"]["traceback_ms"], got["stats"]["traceback_ms"], want["stats"]["forward_ms"], got["stats"]["forward_ms"]))


def test_pair_form_forward_gives_identical_records(default_run):
    want = default_run
    got, _ = _driver({"NDGPU_K7": "pair"})
    assert len(want["digests"]) > 50 and got["digests"] == want["digests"]
    both, _ = _driver({"NDGPU_K7": "pair", "NDGPU_K8A": "wave"})
    assert both["digests"] == want["digests"]
    print("forward ms: wavefront per alignment %.1f, two per wavefront %.1f; with both alternative forms: forward %.1f, traceback %.1f (default %.1f)"
          % (want["stats"]["forward_ms"], got["stats"]["forward_ms"], both["stats"]["forward_ms"], both["stats"]["traceback_ms"],
             want["stats"]["traceback_ms"]))
