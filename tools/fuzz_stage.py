#!/usr/bin/env python
"""The whole stage on random read sets: the reference chain run program by program (seq_dump -> minimap2-nd --step 1 per job -> ovl_sort
-> the reference's OWN driver lib/nextcorrect.py on its own nextcorrect.so: oracle/_ref) against the device stage in one command
(nextdenovo_amd.correct_stage: overlap -> sort -> consensus with nothing on disk in between): every record of cns.fasta.
    python tools/fuzz_stage.py SEED N        on a GPU box, or with NDGPU_SIMT=1 under the kernel interpreter (minutes per set)"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refpipe  # noqa: E402
from nextdenovo_amd import synth  # noqa: E402


def records(path):
    out, lines = {}, open(path).read().splitlines()
    for i in range(0, len(lines) - 1, 2):
        out[lines[i].split()[0]] = (lines[i], lines[i + 1])
    return out


def main():
    seed, n_sets = int(sys.argv[1]), int(sys.argv[2])
    if os.environ.get("NDGPU_SIMT"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
        import build_simt
        from nextdenovo_amd import api, overlap
        os.environ.setdefault("NDGPU_CONTEXTS", "2")
        overlap._lib = overlap._bind(C.CDLL(build_simt.build_overlap()))
        api._LIB = api._bind(C.CDLL(build_simt.build()))
    from nextdenovo_amd import correct_stage
    rng = np.random.default_rng(seed)
    import shutil
    drv = tempfile.mkdtemp(prefix="refdriver")   # the reference's driver next to the reference's own library
    for n in ("nextcorrect.py", "kit.py", "ovlseq.so"):
        shutil.copy(os.path.join(refpipe.REFDIR, "driver", n), drv)
    shutil.copy(os.path.join(refpipe.REFDIR, "nextcorrect.so"), drv)
    driver = os.path.join(drv, "nextcorrect.py")
    bad = 0
    for it in range(n_sets):
        prof = str(rng.choice(["ont", "ont", "clr"]))
        preset, rtype = ("ava-ont", "ont") if prof == "ont" else ("ava-pb", "clr")
        g = synth.make_genome(int(rng.integers(25000, 60000)), seed=int(rng.integers(1, 10 ** 6)), n_repeats=int(rng.integers(0, 4)),
                              repeat_len=int(rng.integers(800, 2500)))
        rs = synth.simulate_reads(g, float(rng.uniform(22, 40)), prof, seed=int(rng.integers(1, 10 ** 6)), mu=float(rng.uniform(8.4, 9.0)),
                                  sigma=float(rng.uniform(0.35, 0.55)), min_len=800)
        seqs = list(rs.seqs)
        for t in range(int(rng.integers(0, 8))):   # a few glued (chimeric) reads: blacklist verdicts, trimmed piles
            a, b = rng.integers(0, len(seqs), 2)
            seqs.append(np.concatenate([seqs[a][: max(1200, seqs[a].size // 2)], synth.revcomp_codes(seqs[b])[: max(1200, seqs[b].size // 2)]]))
        wd = tempfile.mkdtemp(prefix="fstage")
        fa = os.path.join(wd, "reads.fa")
        refpipe.write_fasta(fa, [synth.codes_to_ascii(s) for s in seqs])
        seed_cutoff, k = int(rng.choice([3000, 5000, 7000])), int(rng.choice([20, 30, 40]))
        min_len_seed = seed_cutoff // 2
        extra = ["-b"] if rng.random() < 0.25 else []
        if rng.random() < 0.25:
            extra += ["-s"]
        t0 = time.time()
        try:
            idxs, sorted_ovl = refpipe.run_overlap_chain(wd, fa, seed_cutoff, preset=preset, sort_depth=k)
            ref_out = os.path.join(wd, "ref.cns.fasta")
            r = subprocess.run([sys.executable, driver, "-f", idxs, "-i", sorted_ovl, "-r", rtype, "-p", "4", "-min_len_seed", str(min_len_seed),
                                "-o", ref_out] + extra, capture_output=True, text=True, timeout=1800)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-300:])
        except Exception as e:  # noqa: BLE001
            print(it, "reference chain failed:", str(e)[:200], flush=True)
            continue
        out = os.path.join(wd, "dev")
        rc = correct_stage.run(["-d", os.path.join(wd, "db"), "-x", preset, "-k", str(k), "-r", rtype, "-min_len_seed", str(min_len_seed), "-p", "4",
                                "-o", out] + extra)
        got, want = records(out + ".001.fasta"), records(ref_out)
        ok = rc == 0 and got == want
        note = ""
        if rc == 0 and not ok:
            # Equal sort keys in DIFFERENT raw files (duplicated read segments): the reference's merge takes the block with the lower
            # index (util/ovl_sort.c:883-893) and its reader threads take their blocks in the order they get to their first record
            # (:933-936) -- a race.  The device keeps file order, which is what the reference does on an idle machine.  Sort again;
            # if the reference does not reproduce its own file, compare against the chain rerun from the new one.
            ra = os.path.dirname(sorted_ovl)
            for attempt in range(3):
                again = os.path.join(ra, "again%d.sorted.ovl" % attempt)
                subprocess.run([os.path.join(refpipe.REFDIR, "ovl_sort"), "-m", "2g", "-t", "4", "-k", str(k), "-i", os.path.join(wd, "db", ".input.seed.001.idx"),
                                "-o", os.path.basename(again), "input.fofn"], cwd=ra, check=True, capture_output=True)
                if open(again, "rb").read() == open(sorted_ovl, "rb").read():
                    continue
                ref2 = os.path.join(wd, "ref2.cns.fasta")
                subprocess.run([sys.executable, driver, "-f", idxs, "-i", again, "-r", rtype, "-p", "4", "-min_len_seed", str(min_len_seed),
                                "-o", ref2] + extra, capture_output=True, text=True, timeout=1800, check=True)
                if records(ref2) == got:
                    ok, note = True, "(the reference's sort does not reproduce itself on this set: cross-file tie race; equal to its rerun)"
                break
        bad += not ok
        print(it, "equal" if ok else "DIFFER", prof, "seed_cutoff", seed_cutoff, "k", k, extra, "records", len(got), len(want), "%.0fs" % (time.time() - t0),
              note if ok else wd, flush=True)
    print("mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
