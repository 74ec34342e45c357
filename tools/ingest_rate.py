#!/usr/bin/env python
"""Read-ingestion throughput on the host cores (no GPU involved): N gzipped FASTQ files through the native streaming reader, one
after the other (what the reference's seq_dump does, util/seq_dump.c:60-72) and side by side (iter_chunks_files), + Python's gzip
as a yardstick.  python tools/ingest_rate.py [out.json]"""
import gzip
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nextdenovo_amd import seq_dump  # noqa: E402

N_FILES, BASES_PER_FILE = 8, 48_000_000
wd = tempfile.mkdtemp(prefix="ndingest")
rng = np.random.default_rng(3)
paths = []
for k in range(N_FILES):
    p = os.path.join(wd, "reads%d.fastq.gz" % k)
    with gzip.open(p, "wb", compresslevel=4) as f:
        done = 0
        i = 0
        while done < BASES_PER_FILE:
            n = int(rng.integers(5000, 40000))
            s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), n).tobytes()
            f.write(b"@r%d_%d\n%s\n+\n%s\n" % (k, i, s, b"I" * n))
            done += n
            i += 1
    paths.append(p)
total = N_FILES * BASES_PER_FILE
res = {"files": N_FILES, "bases": total, "compressed_bytes": sum(os.path.getsize(p) for p in paths), "host_cores": os.cpu_count()}


def consume(it):
    n = 0
    for chunk in it:
        n += int(chunk[2].sum())
    return n


for name, thr in (("native_reader_one_file_at_a_time", 1), ("native_reader_files_side_by_side", 0)):
    t0 = time.perf_counter()
    n = consume(seq_dump.iter_chunks_files(paths, threads=thr))
    dt = time.perf_counter() - t0
    res[name] = {"threads": thr or min(N_FILES, os.cpu_count() or 1, 8), "seconds": dt, "mbases_per_s": n / dt / 1e6}
    print(name, res[name], flush=True)
t0 = time.perf_counter()
with gzip.open(paths[0], "rb") as f:
    while f.read(1 << 24):
        pass
res["python_gzip_inflate_only_one_file_s"] = time.perf_counter() - t0
print(res["python_gzip_inflate_only_one_file_s"])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
