"""Build the in-tree native library with hipcc for gfx950.

    python -m nextdenovo_amd.build

produces nextdenovo_amd/libndgpu_nextcorrect.so (correction stage: HIP kernels + host engine + C ABI)
and nextdenovo_amd/libndgpu_overlap.so (overlap stage: `minimap2-nd --step 1` path).
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
SOURCES = ["ond_kernels.hip", "msa_kernels.hip", "lq_kernels.hip", "ext_kernels.hip", "device_runtime.hip", "consensus.cpp", "poa.cpp", "readdb.cpp", "capi.cpp"]
HEADERS = ["nd_device.h", "nd_host.h", "nd_runtime.h", os.path.join("..", "..", "include", "ndgpu_nextcorrect.h")]
OVL_LIB = os.path.join(HERE, "libndgpu_overlap.so")
OVL_SOURCES = ["ovl_kernels.hip", "ovl_engine.hip", "ovlsort_kernels.hip", "ovlsort_engine.hip", "fastx_reader.cpp", "ovl_step2.cpp", "ovl_cigar.cpp", "ksw2_kernels.hip"]
OVL_HEADERS = ["ovl_device.h", os.path.join("..", "..", "include", "ndgpu_overlap.h")]


def _stale(lib=None, deps=None) -> bool:
    lib = lib or LIB
    deps = deps or (SOURCES + HEADERS)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    for lib, srcs, hdrs, exports in ((LIB, SOURCES, HEADERS, "exports_nextcorrect.map"), (OVL_LIB, OVL_SOURCES, OVL_HEADERS, "exports_overlap.map")):
        if not force and not _stale(lib, srcs + hdrs + [exports]):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
               "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-Wl,--version-script=" + os.path.join(CSRC, exports), "-o", lib] + os.environ.get("NDGPU_CXXFLAGS", "").split() + [os.path.join(CSRC, f) for f in srcs] + (["-lz", "-ldl"] if lib == OVL_LIB else [])
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
