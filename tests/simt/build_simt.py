"""TEST INFRASTRUCTURE ONLY.  Builds tests/simt/_build/libnextcorrect_simt.so: the consensus library's own sources
(nextdenovo_amd/csrc, unmodified) compiled with g++ against the lane-accurate interpreter in tests/simt/include, so that the
kernels' logic can be run against the oracle on a machine without a GPU.  Nothing under nextdenovo_amd/ knows this file
exists; only tests load it."""
from __future__ import annotations

import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "nextdenovo_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libnextcorrect_simt.so")
OVL_LIB = os.path.join(OUT_DIR, "liboverlap_simt.so")
SOURCES = ["ond_kernels.hip", "msa_kernels.hip", "lq_kernels.hip", "ext_kernels.hip", "device_runtime.hip", "consensus.cpp", "poa.cpp", "readdb.cpp",
           "capi.cpp"]
OVL_SOURCES = ["ovl_kernels.hip", "ovl_engine.hip", "ovlsort_kernels.hip", "ovlsort_engine.hip", "fastx_reader.cpp", "pinflate.cpp", "ovl_step2.cpp", "ovl_cigar.cpp", "ksw2_kernels.hip"]


def _stale(lib) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "simt_runtime.cpp"),
                                                                   os.path.join(HERE, "include", "hip", "hip_runtime.h"),
                                                                   os.path.join(HERE, "include", "rocprim", "rocprim.hpp")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False) -> str:
    return _build(LIB, SOURCES, [], force)


def build_overlap(force: bool = False) -> str:
    return _build(OVL_LIB, OVL_SOURCES, ["-lz", "-ldl"], force)


def _build(lib, sources, libs, force) -> str:
    out_dir = OUT_DIR
    flags = ["-std=c++17", "-O2", "-g", "-fPIC", "-pthread", "-I", os.path.join(HERE, "include"), "-I", CSRC, "-w"]
    san = os.environ.get("SIMT_SANITIZE")  # e.g. "undefined": shifts, signed overflow, misaligned / null accesses inside the kernels
    if san:                                # (a build of its own under _build/<sanitizer>/; load it with libubsan preloaded)
        out_dir = os.path.join(OUT_DIR, san.replace(",", "_"))
        flags += ["-fsanitize=" + san, "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]
        lib = os.path.join(out_dir, os.path.basename(lib))
        libs = list(libs) + ["-fsanitize=" + san]
    if not force and not _stale(lib):
        return lib
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    procs = []
    for src in sources + ["simt_runtime.cpp"]:
        path = os.path.join(HERE if src == "simt_runtime.cpp" else CSRC, src)
        if src.endswith(".hip"):
            # the one textual change: GCN inline assembly (memory-ordering waits) has no host meaning and is blanked
            text = open(path).read()
            text, n_asm = re.subn(r'asm volatile\("s_[^;]*;', ";", text)
            # and the workgroup's dynamically sized LDS array is the interpreter's per-workgroup buffer
            text = re.sub(r'extern __shared__ (\w+) (\w+)\[\];', r'\1 *\2 = (\1 *)simt::dynamic_lds();', text)
            path = os.path.join(out_dir, src + ".cpp")
            with open(path, "w") as f:
                f.write('#line 1 "%s"\n' % os.path.join(CSRC, src) + text)
        obj = os.path.join(out_dir, src + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen(["g++", *flags, "-x", "c++", "-c", path, "-o", obj], stderr=subprocess.PIPE, text=True)))
    for src, p in procs:
        _, err = p.communicate()
        if p.returncode:
            raise RuntimeError("simt build of %s failed:\n%s" % (src, err[-6000:]))
    subprocess.check_call(["g++", "-shared", "-pthread", "-o", lib, *objs, *libs])
    return lib


if __name__ == "__main__":
    print(build(force=True))
    print(build_overlap(force=True))
