#!/usr/bin/env python
"""Run `-m gpu` test files against the interpreted libraries (tests/simt) instead of a GPU: a dry run of the GPU tests' own
logic and of kernel changes before GPU minutes are spent on them.  Tests that start child processes (which load the real
library) or that work at BASELINE sizes are not meant for this; name the files / tests to run as for pytest:

    python tools/gpu_tests_interpreted.py tests/test_gpu_overlap.py -k "not live_set"
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests", "simt")):
    sys.path.insert(0, p)
os.environ.setdefault("NDGPU_CONTEXTS", "1")
import build_simt  # noqa: E402
import pytest  # noqa: E402
from nextdenovo_amd import api, overlap  # noqa: E402

overlap._lib = overlap._bind(C.CDLL(build_simt.build_overlap()))
api._LIB = api._bind(C.CDLL(build_simt.build()))
sys.exit(pytest.main(["-p", "no:cacheprovider", "-q", *sys.argv[1:]]))
