#!/usr/bin/env python
"""What the drop-in of INTEGRATION.md section 1 delivers: the reference's own stage driver (oracle/_ref/driver/nextcorrect.py,
unmodified, `-p P` forked workers calling nextCorrect() once per seed) on a 0.8 Mb / 45x ONT stage (2.7k reads, 36 Mb) with
 (a) the product library installed as nextcorrect.so  (b) the same driver with integration/nextcorrect_ndgpu_batch.patch and
NDGPU_BATCH=1  (c) the reference's own nextcorrect.so on the host cores -- corrected bases per second of wall time each, process
start and DB load included, and whether the three cns.fasta hold the same records.  Run on the GPU box:
    python tools/dropin_rate.py gpurun_out/<tag>/dropin_rate.json
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mm_util as M  # noqa: E402
from nextdenovo_amd import build, minimap2_nd, ovl_sort, synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def records(path):
    out, lines = {}, open(path).read().splitlines()
    for i in range(0, len(lines) - 1, 2):
        out[lines[i].split()[0]] = (lines[i], lines[i + 1])
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "dropin_rate.json"
    wd = tempfile.mkdtemp(prefix="nddropin")
    g = synth.make_genome(800000, seed=91)
    rs = synth.simulate_reads(g, 45, "ont", seed=92)
    seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in rs.seqs], seed_cutoff=12000)
    db = os.path.dirname(seed)
    o0, o1 = os.path.join(wd, "raw0.ovl"), os.path.join(wd, "raw1.ovl")
    assert minimap2_nd.run(["--step", "1", "--dual=yes", "-t", "8", "-x", "ava-ont", seed, part, "-o", o0]) == 0
    assert minimap2_nd.run(["--step", "1", "-I", "3G", "-t", "8", "-x", "ava-ont", seed, seed, "-o", o1]) == 0
    fofn = os.path.join(wd, "ovl.fofn")
    with open(fofn, "w") as f:
        f.write(o0 + "\n" + o1 + "\n")
    so = os.path.join(wd, "sorted.ovl")
    assert ovl_sort.run(["-m", "2g", "-t", "4", "-k", "40", "-i", os.path.join(db, ".input.seed.001.idx"), "-o", so, fofn]) == 0
    idxs = os.path.join(wd, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(db)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(db, n) + "\n")
    cores = os.cpu_count() or 1
    res = {"stage": "0.8 Mb genome, 45x ONT-profile reads (%d reads), seed_cutoff 12k; stage files written by the device chain" % len(rs.seqs),
           "host_cores": cores}
    want = None
    for name, lib_so, patched, env, p in (("reference_library_host_cores", os.path.join(REF, "nextcorrect.so"), False, {}, min(cores, 64)),
                                          ("product_library_unmodified_driver", build.LIB, False, {}, 16),
                                          ("product_library_batch_patch", build.LIB, True, {"NDGPU_BATCH": "1"}, 4)):
        lib = os.path.join(wd, "lib_" + name)
        shutil.copytree(os.path.join(REF, "driver"), lib)
        shutil.copy(lib_so, os.path.join(lib, "nextcorrect.so"))
        if patched:
            subprocess.run(["patch", "-s", os.path.join(lib, "nextcorrect.py"), os.path.join(ROOT, "integration", "nextcorrect_ndgpu_batch.patch")], check=True)
        out = os.path.join(wd, name + ".cns.fasta")
        cmd = [sys.executable, os.path.join(lib, "nextcorrect.py"), "-f", idxs, "-i", so, "-r", "ont", "-p", str(p), "-o", out]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=3000, env=dict(os.environ, **env))
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            res[name] = {"error": r.stderr[-500:]}
            continue
        got = records(out)
        if want is None:
            want = got
        bases = sum(len(s) for _, s in got.values())
        res[name] = {"workers_p": p, "seeds": len(got), "corrected_bases": bases, "wall_s": dt, "bases_per_s": bases / dt, "same_records_as_reference": got == want}
        print(name, json.dumps(res[name]), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
