"""The drop-in the north star describes: the reference's OWN stage driver, unmodified (oracle/_ref/driver/nextcorrect.py + kit.py,
copied there by oracle/Makefile, beside its stock ovlseq.so), with the product library installed as `nextcorrect.so` next to it --
exactly what INTEGRATION.md section 1 tells a user to do.  `-p 4` forks four workers after the CDLL (lib/nextcorrect.py:56,232);
they share the GPU.  Then the same driver with integration/nextcorrect_ndgpu_batch.patch applied and NDGPU_BATCH=1 (section 2).
Both must write the cns.fasta / .idx the reference wrote with its own nextcorrect.so (tests/golden/stage)."""
import gzip
import os
import shutil
import subprocess
import sys
import time

import pytest

import refpipe
import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(util.HERE)
DRIVER = os.path.join(refpipe.REFDIR, "driver")
STAGE = os.path.join(util.GOLD, "stage")


def _install(tmp_path, patched):
    from nextdenovo_amd import build
    lib = str(tmp_path / "lib")
    shutil.copytree(DRIVER, lib)
    shutil.copy(build.LIB, os.path.join(lib, "nextcorrect.so"))
    if patched:
        subprocess.run(["patch", "-s", os.path.join(lib, "nextcorrect.py"), os.path.join(ROOT, "integration", "nextcorrect_ndgpu_batch.patch")],
                       check=True)
    d = str(tmp_path / "stage")
    shutil.copytree(STAGE, d)
    idxs = os.path.join(d, "idxs.fofn")
    with open(idxs, "w") as f:
        for n in sorted(os.listdir(d)):
            if n.startswith(".input.") and n.endswith(".idx"):
                f.write(os.path.join(d, n) + "\n")
    return lib, d, idxs, os.path.join(d, "input.seed.001.sorted.ovl")


def _records(text):
    """{seed: (header, sequence)} of a cns.fasta"""
    out, lines = {}, text.splitlines()
    for i in range(0, len(lines) - 1, 2):
        out[lines[i].split()[0]] = (lines[i], lines[i + 1])
    return out


def _golden(name):
    with gzip.open(os.path.join(STAGE, name + ".gz"), "rb") as f:
        return f.read().decode()


@pytest.mark.skipif(not os.path.exists(os.path.join(DRIVER, "nextcorrect.py")), reason="reference driver did not travel")
@pytest.mark.parametrize("patched,env,extra,gold", [
    (False, {}, [], "cns.default.fasta"),
    (False, {}, ["-b"], "cns.nobl.fasta"),
    (True, {"NDGPU_BATCH": "1"}, [], "cns.default.fasta"),
    (True, {"NDGPU_BATCH": "1"}, ["-b"], "cns.nobl.fasta"),
    (True, {}, [], "cns.default.fasta"),            # the patch leaves the stock path alone when the switch is off
], ids=["unmodified", "unmodified-b", "batch", "batch-b", "patched-switch-off"])
def test_reference_driver_on_product_library(tmp_path, patched, env, extra, gold):
    lib, d, idxs, so = _install(tmp_path, patched)
    out = os.path.join(d, "cns.fasta")
    cmd = [sys.executable, os.path.join(lib, "nextcorrect.py"), "-f", idxs, "-i", so, "-r", "ont", "-p", "4", "-min_len_seed", "1250",
           "-o", out] + extra
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
    dt = time.perf_counter() - t0
    assert r.returncode == 0, r.stderr[-3000:]
    got, want = _records(open(out).read()), _records(_golden(gold))
    assert got == want and len(want) > 20                       # every record: header (id, len, %f identity) and bases
    idx_got = sorted(open(out + ".idx").read().splitlines(), key=lambda l: int(l.split()[0]))
    idx_want = sorted(_golden(gold + ".idx").splitlines(), key=lambda l: int(l.split()[0]))
    assert [l.split()[0] for l in idx_got] == [l.split()[0] for l in idx_want]   # (offsets depend on the order workers finish in)
    bases = sum(len(s) for _, s in got.values())
    print("\n[drop-in %s%s] %d seeds, %d corrected bases in %.2f s (process start, DB load and HIP init included): %.0f bases/s"
          % ("NDGPU_BATCH=1 " if env else "", "patched" if patched else "unmodified", len(got), bases, dt, bases / dt))
