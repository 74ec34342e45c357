#!/usr/bin/env python
"""A/B of a kernel CANDIDATE without touching the product sources (whose hash the committed counter figures are tied to):
the candidate is the product's csrc/ond_kernels.hip with named pieces replaced (exact-text substitution, asserted), written to
build_variants/<name>/ (git-ignored; travels to the GPU box), compiled for gfx950 and linked with the product's other objects.

    python tools/kernel_candidate.py build <name>          build_variants/<name>/libndgpu_nextcorrect.so (hipcc; no GPU needed)
    python tools/kernel_candidate.py simt <name>           the same source under the kernel interpreter: golden piles + random piles
                                                            against the compiled reference (CPU; needs oracle/_ref)
    python tools/kernel_candidate.py isa <name>            VALU / SALU / memory instruction counts of K7 and K8a, product vs candidate
    python tools/kernel_candidate.py bench <name> [args]   bench.py with the candidate library in the product library's place (GPU box)
    python tools/kernel_candidate.py chain <name> [out]    tools/chain_latency.py (the isolated edit step) with the candidate library (GPU box)

Candidates:
  snake32   the match run of a cell compared 32 bases per round (three + three words, two funnel shifts each, one 64-bit count) in
            K7 / K7w (the snake) and K8a (the run walked back) instead of 64 (five + five words, four shifts, a four-way cascade):
            DESIGN.md section 7c.1 -- the run averages ~10 equal bases, a lane of ~33 meets 32 equal bases in ~3 % of the steps.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "nextdenovo_amd", "csrc")

SNAKE64_BODY = '''    for (;;) {
        int rem = q_len - x;
        const int rt = t_len - y;
        rem = rt < rem ? rt : rem;
        if (rem <= 0) break;
        const Bases64 a = fetch64_rel(qp, q_sh + (uint32_t)x);
        const Bases64 b = fetch64_rel(tp, t_sh + (uint32_t)y);
        const uint32_t d0 = a.w[0] ^ b.w[0], d1 = a.w[1] ^ b.w[1], d2 = a.w[2] ^ b.w[2], d3 = a.w[3] ^ b.w[3];
        int m = d0 ? (__builtin_ctz(d0) >> 1) : d1 ? 16 + (__builtin_ctz(d1) >> 1) : d2 ? 32 + (__builtin_ctz(d2) >> 1) : d3 ? 48 + (__builtin_ctz(d3) >> 1) : 64;
        m = m < rem ? m : rem;
        x += m;
        y += m;
        if (m < 64) break;
    }
    return x;
'''
SNAKE32_BODY = '''    for (;;) {
        int rem = q_len - x;
        const int rt = t_len - y;
        rem = rt < rem ? rt : rem;
        if (rem <= 0) break;
        const uint64_t a = fetch32(qp, q_sh + (uint32_t)x), b = fetch32(tp, t_sh + (uint32_t)y);
        const uint64_t dd = a ^ b;
        int m = dd ? (__builtin_ctzll(dd) >> 1) : 32;
        m = m < rem ? m : rem;
        x += m;
        y += m;
        if (m < 32) break;
    }
    return x;
'''
FETCH32 = '''// (candidate snake32) 32 bases starting at base `pos` of a sequence as one 64-bit word: three words read, two funnel shifts
__device__ __forceinline__ uint64_t fetch32(const uint32_t *__restrict__ seq, uint64_t pos) {
    const uint32_t *__restrict__ p = seq + (pos >> 4);
    const uint32_t s = (uint32_t)(pos & 15u) * 2u;
    const uint32_t v0 = p[0], v1 = p[1], v2 = p[2];
    const uint32_t lo = (uint32_t)((((uint64_t)v1 << 32) | v0) >> s), hi = (uint32_t)((((uint64_t)v2 << 32) | v1) >> s);
    return ((uint64_t)hi << 32) | lo;
}

'''
SNAKE_DECL = "__device__ __forceinline__ int snake64(const uint32_t *__restrict__ qp, const uint32_t *__restrict__ tp, uint32_t q_sh, uint32_t t_sh,"
K8A_RUN64 = '''            const int n = avail < 64 ? avail : 64;
            const Bases64 a = fetch64_abs(qp, q_off + (uint64_t)(uint32_t)(x - n + 1));
            const Bases64 b = fetch64_abs(tp, t_off + (uint64_t)(uint32_t)(yy - n + 1));
            int m = n;  // bases [0, n) of the fetch are the run's candidates, the last one first
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                const int nb = n - 16 * i;  // candidates in word i
                if (nb <= 0) continue;
                uint32_t diff = a.w[i] ^ b.w[i];
                if (nb < 16) diff &= (1u << (2 * nb)) - 1u;
                if (diff) {
                    m = n - 1 - (16 * i + ((31 - __builtin_clz(diff)) >> 1));
                    break;
                }
            }
'''
K8A_RUN32 = '''            const int n = avail < 32 ? avail : 32;
            uint64_t diff = fetch32(qp, q_off + (uint64_t)(uint32_t)(x - n + 1)) ^ fetch32(tp, t_off + (uint64_t)(uint32_t)(yy - n + 1));
            if (n < 32) diff &= (1ull << (2 * n)) - 1ull;   // bases [0, n) of the fetch are the run's candidates, the last one first
            const int m = diff ? n - 1 - ((63 - __builtin_clzll(diff)) >> 1) : n;
'''

CANDIDATES = {
    "snake32": [(SNAKE64_BODY, SNAKE32_BODY), (SNAKE_DECL, FETCH32 + SNAKE_DECL), (K8A_RUN64, K8A_RUN32)],
}


def variant_dir(name):
    d = os.path.join(ROOT, "build_variants", name)
    os.makedirs(d, exist_ok=True)
    return d


def candidate_source(name):
    text = open(os.path.join(CSRC, "ond_kernels.hip")).read()
    for old, new in CANDIDATES[name]:
        assert text.count(old) == 1, "the product source no longer holds the piece this candidate replaces:\n" + old[:200]
        text = text.replace(old, new)
    path = os.path.join(variant_dir(name), "ond_kernels.hip")
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
    return path


def build(name):
    from nextdenovo_amd import build as B
    B.build()   # the product's objects (nextdenovo_amd/_obj)
    src = candidate_source(name)
    d = variant_dir(name)
    hipcc = "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function", "-fvisibility=hidden", "-I", CSRC]
    obj = os.path.join(d, "ond_kernels.hip.o")
    subprocess.run([hipcc, *flags, "-x", "hip", "-c", src, "-o", obj], check=True)
    objs = [obj if f == "ond_kernels.hip" else os.path.join(B.HERE, "_obj", f + ".o") for f in B.SOURCES]
    lib = os.path.join(d, "libndgpu_nextcorrect.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "--hip-link", "-shared", "-fPIC", "-pthread",
                    "-Wl,--version-script=" + os.path.join(CSRC, "exports_nextcorrect.map"), "-o", lib, *objs], check=True)
    return lib


def isa(name):
    import re
    d = variant_dir(name)
    out = {}
    for tag, src in (("product", os.path.join(CSRC, "ond_kernels.hip")), (name, candidate_source(name))):
        s_path = os.path.join(d, tag + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-x", "hip", "--cuda-device-only", "-S",
                        src, "-o", s_path], check=True)
        cur, counts = None, {}
        for line in open(s_path):
            m = re.match(r"^(_ZN\w+):", line)
            if m:
                cur = m.group(1)
                counts[cur] = {"valu": 0, "salu": 0, "mem": 0, "lds": 0}
                continue
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                cur = None
            ins = line.strip().split(" ")[0] if line.startswith("\t") and not line.startswith("\t.") and not line.startswith("\t;") else ""
            if cur and ins:
                kind = "valu" if ins.startswith("v_") else "salu" if ins.startswith("s_") else "lds" if ins.startswith("ds_") else "mem" if ins.startswith(("global_", "buffer_", "flat_", "scratch_")) else None
                if kind:
                    counts[cur][kind] += 1
        out[tag] = {k: v for k, v in counts.items() if "ond_" in k}
    for tag, c in out.items():
        for k, v in c.items():
            print("%-8s %-60s %s" % (tag, k[:60], v))


def simt(name):
    """Golden piles and random piles through the interpreted build of the candidate against the compiled reference."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
    import build_simt
    src = candidate_source(name)
    # the interpreter's build takes the library's sources from CSRC: give it a CSRC of links with the one file exchanged
    d = os.path.join(variant_dir(name), "csrc")
    os.makedirs(d, exist_ok=True)
    for f in os.listdir(CSRC):
        p = os.path.join(d, f)
        if os.path.lexists(p):
            os.remove(p)
        os.symlink(src if f == "ond_kernels.hip" else os.path.join(CSRC, f), p)
    inc = os.path.join(ROOT, "build_variants", "include")   # (the sources include "../../include/...")
    if not os.path.lexists(inc):
        os.symlink(os.path.join(ROOT, "include"), inc)
    build_simt.CSRC = d
    build_simt.OUT_DIR = os.path.join(variant_dir(name), "simt")
    build_simt.LIB = os.path.join(build_simt.OUT_DIR, "libnextcorrect_simt.so")
    lib = build_simt._build(build_simt.LIB, build_simt.SOURCES, [], True)
    print("interpreted candidate:", lib)
    env = dict(os.environ, NDGPU_SIMT="1", NDGPU_SIMT_LIB=lib)
    rc = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_consensus.py"), "7301", "6"], env=env).returncode
    if rc == 0:   # the interpreter tests of the consensus kernels (golden piles, single alignments, forced paths) on the candidate
        import pytest
        rc = int(pytest.main([os.path.join(ROOT, "tests", "test_simt_kernels.py"), "-x", "-q", "-p", "no:cacheprovider"]))
    return rc


def bench(name, argv):
    from nextdenovo_amd import build as B
    lib = os.path.join(variant_dir(name), "libndgpu_nextcorrect.so")
    if not os.path.exists(lib):
        sys.exit("build it first (here): python tools/kernel_candidate.py build " + name)
    B.LIB = lib   # api.load() binds to this file
    import runpy
    sys.argv = ["bench.py"] + argv
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")


def chain(name, argv):
    """tools/chain_latency.py (one long alignment on an idle device: the isolated edit step) with the candidate library."""
    from nextdenovo_amd import build as B
    B.LIB = os.path.join(variant_dir(name), "libndgpu_nextcorrect.so")
    import runpy
    sys.argv = ["chain_latency.py"] + argv
    runpy.run_path(os.path.join(ROOT, "tools", "chain_latency.py"), run_name="__main__")


if __name__ == "__main__":
    what, name = sys.argv[1], sys.argv[2]
    if what == "build":
        print(build(name))
    elif what == "isa":
        isa(name)
    elif what == "simt":
        sys.exit(simt(name))
    elif what == "bench":
        bench(name, sys.argv[3:])
    elif what == "chain":
        chain(name, sys.argv[3:])
