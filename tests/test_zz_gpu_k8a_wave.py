"""The wavefront-per-alignment form of the traceback kernel (NDGPU_K8A=wave, csrc/ond_kernels.hip) against the default
lane-per-alignment kernel on the device: the same read set through the whole chain, records identical; the traceback times of both
are printed for the A/B (`pytest -s`).  (Named to run after the other GPU tests; the switch is read once per process, hence the
child processes.)"""
import pytest

from test_gpu_robust import _driver

pytestmark = pytest.mark.gpu


def test_wave_form_traceback_gives_identical_records():
    want, _ = _driver({})
    got, _ = _driver({"NDGPU_K8A": "wave"})
    assert len(want["digests"]) > 50 and got["digests"] == want["digests"]
    print("traceback ms: lane-per-alignment %.1f, wavefront-per-alignment %.1f; forward %.1f / %.1f"
          % (want["stats"]["traceback_ms"], got["stats"]["traceback_ms"], want["stats"]["forward_ms"], got["stats"]["forward_ms"]))
