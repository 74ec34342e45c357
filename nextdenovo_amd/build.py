"""Build the in-tree native library with hipcc for gfx950.

    python -m nextdenovo_amd.build

produces nextdenovo_amd/libndgpu_nextcorrect.so (HIP kernels + host engine + C ABI).
hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU
box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libndgpu_nextcorrect.so")
SOURCES = ["ond_kernels.hip", "msa_kernels.hip", "device_runtime.hip", "consensus.cpp", "poa.cpp", "readdb.cpp", "capi.cpp"]
HEADERS = ["nd_device.h", "nd_host.h", "nd_runtime.h", os.path.join("..", "..", "include", "ndgpu_nextcorrect.h")]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
           "-Wall", "-Wno-unused-function", "-o", LIB] + os.environ.get("NDGPU_CXXFLAGS", "").split() + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
