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
OVL_SOURCES = ["ovl_kernels.hip", "ovl_engine.hip", "ovlsort_kernels.hip", "ovlsort_engine.hip", "fastx_reader.cpp", "pinflate.cpp", "ovl_step2.cpp", "ovl_cigar.cpp", "ksw2_kernels.hip"]
OVL_HEADERS = ["ovl_device.h", os.path.join("..", "..", "include", "ndgpu_overlap.h")]


def _stale(lib=None, deps=None) -> bool:
    lib = lib or LIB
    deps = deps or (SOURCES + HEADERS)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """One object per source (nextdenovo_amd/_obj/, compiled side by side, recompiled only when the source or any header is newer),
    then one link per library."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    obj_dir = os.path.join(HERE, "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    extra = os.environ.get("NDGPU_CXXFLAGS", "").split()
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function", "-fvisibility=hidden"] + extra
    stamp = os.path.join(obj_dir, "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(flags):
        force = True
    all_headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", f) for f in ("ndgpu_nextcorrect.h", "ndgpu_overlap.h")]
    t_hdr = max(os.path.getmtime(h) for h in all_headers)
    jobs, plan = [], []
    for lib, srcs, hdrs, exports in ((LIB, SOURCES, HEADERS, "exports_nextcorrect.map"), (OVL_LIB, OVL_SOURCES, OVL_HEADERS, "exports_overlap.map")):
        objs, relink = [], force or not os.path.exists(lib)
        for f in srcs:
            src, obj = os.path.join(CSRC, f), os.path.join(obj_dir, f + ".o")
            objs.append(obj)
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), t_hdr):
                cmd = [hipcc, *flags, "-x", "hip", "-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                jobs.append((f, subprocess.Popen(cmd)))
                relink = True
            elif os.path.exists(lib) and os.path.getmtime(obj) > os.path.getmtime(lib):
                relink = True
        if not relink and os.path.getmtime(os.path.join(CSRC, exports)) > os.path.getmtime(lib):
            relink = True
        plan.append((lib, objs, exports, relink))
    bad = [f for f, p in jobs if p.wait() != 0]
    if bad:
        raise RuntimeError("hipcc failed for: " + ", ".join(bad))
    for lib, objs, exports, relink in plan:
        if not relink:
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-pthread", "-Wl,--version-script=" + os.path.join(CSRC, exports),
               "-o", lib, *objs] + (["-lz", "-ldl"] if lib == OVL_LIB else [])
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
