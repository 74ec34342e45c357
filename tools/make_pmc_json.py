#!/usr/bin/env python
"""profiles/pmc_traffic.json from a tools/rocprof_pmc_summary.py table: per modelled kernel the FETCH_SIZE / WRITE_SIZE KB per
launch, each tied to the sources its kernel lives in when the passes ran (bench.py refuses a figure once those sources change).
usage: make_pmc_json.py <pmc_summary.txt> <bench steps the passes ran, warm-up included> <config> [source note]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    table, steps, config = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
    note = sys.argv[4] if len(sys.argv) > 4 else os.path.relpath(table, ROOT)
    keys = ("ond_forward_kernel", "ond_traceback_kernel", "tb_walk_kernel", "count_links_kernel", "score_seg_kernel<96")
    kernels = {}
    acc = {}   # a kernel's instantiations together (K7 with and without checkpoints share one HIP-event bracket and one model in bench.py)
    for ln in open(table):
        mt = re.match(r"^(\S.*?)\s+(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", ln)
        if not mt:
            continue
        for k in keys:
            if mt.group(1).startswith(k):
                name = k.split("<")[0]
                a = acc.setdefault(name, [0, 0.0, 0.0])
                calls = int(mt.group(2))
                a[0] += calls
                a[1] += calls * float(mt.group(3))
                a[2] += calls * float(mt.group(4))
    for name, (calls, fkb, wkb) in acc.items():
        kernels[name] = {"launches_per_step": calls / steps, "fetch_kb_per_launch": fkb / calls, "write_kb_per_launch": wkb / calls,
                         "source_sha16": bench.kernel_source_sha16(name)}
    out = {"config": config, "source": note, "kernels": kernels}
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
