import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in sources)


@pytest.fixture(scope="session")
def oracle_lib():
    """oracle/libndoracle.so (plain-C restatement; test infrastructure)."""
    import ctypes as C
    so = os.path.join(ROOT, "oracle", "libndoracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("ond_oracle.c", "ond_ext_oracle.c", "msa_oracle.c", "mm_oracle.c", "ovlsort_oracle.c", "step2_oracle.c", "ksw2_oracle.c", "nd_oracle.h")]
    if _stale(so, srcs):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)
    return C.CDLL(so)


@pytest.fixture(scope="session")
def host_harness():
    """Product host engine + oracle aligner backend (tests/csrc/host_harness.cpp), CPU only."""
    import ctypes as C
    so = os.path.join(HERE, "csrc", "libndhost_test.so")
    csrc = os.path.join(ROOT, "nextdenovo_amd", "csrc")
    srcs = [os.path.join(HERE, "csrc", "host_harness.cpp"), os.path.join(csrc, "consensus.cpp"),
            os.path.join(csrc, "poa.cpp"), os.path.join(csrc, "readdb.cpp"), os.path.join(csrc, "nd_host.h"),
            os.path.join(ROOT, "oracle", "ond_oracle.c"), os.path.join(ROOT, "oracle", "msa_oracle.c")]
    if _stale(so, srcs):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-o", so] + [s for s in srcs if not s.endswith(".h")]
        subprocess.run(cmd, check=True)
    return C.CDLL(so)


@pytest.fixture(scope="session")
def native_lib():
    """The shipped library (HIP).  Loading works without a GPU; compute calls do not."""
    from nextdenovo_amd import api
    return api.load()
