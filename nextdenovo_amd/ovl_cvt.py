#!/usr/bin/env python
"""`ovl_cvt` for the 8-field overlap files of the correction stage (util/ovl_cvt.c): `-m 1 file.ovl` prints the records as tab
separated text, `-m 0 file.txt` packs such text into `.ovl` bytes on stdout.  The 10-field files of the assembly stage (`--step 2`,
header 00 FF, lib/ovl.c:70-107) are outside this engine and are refused.

    python -m nextdenovo_amd.ovl_cvt -m 1 input.seed.001.sorted.ovl > sorted.txt
"""
from __future__ import annotations

import os
import sys

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nextdenovo_amd import ovl, overlap  # noqa: E402


def run(argv, out=None) -> int:
    import getopt
    opts, args = getopt.getopt(argv, "m:")
    mode = int(dict(opts).get("-m", "0"))
    if len(args) < 1:
        sys.stderr.write("Usage: ovl_cvt [-m 0|1] input\n")
        return 1
    out = out or sys.stdout.buffer
    if mode:
        with open(args[0], "rb") as f:
            head = f.read(2)
        if head == b"\x00\xff":
            raise SystemExit("[ERROR] 10-field (--step 2) overlap files are not handled by this engine")
        recs = ovl.decode_ovl(args[0])
        out.write("".join("%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n" % tuple(int(x) for x in r) for r in recs).encode())
    else:
        rows = []
        with open(args[0]) as f:
            for line in f:
                p = line.split("\t")
                if len(p) > 8:
                    raise SystemExit("[ERROR] 10-field (--step 2) overlap text is not handled by this engine")
                if len(p) == 8:
                    rows.append([int(x) for x in p])
        a = np.asarray(rows, dtype=np.uint32).reshape(-1, 8)
        a[:, 1] &= 0xff  # %hhu
        out.write(overlap.encode(overlap.from_decoded(a), np.zeros(2, dtype=np.uint32)))
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
