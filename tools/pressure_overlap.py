"""The overlap stage on a device that is short of memory (a config-3 run of round 3 returned records built from
half the anchors, silently).  One all-vs-all job (index + map + ovl_sort) of a config-2-like read set is run on the idle device,
then again and again while a block of device memory taken by this script leaves less and less free: every run must either give the
idle run's records or raise MemoryError.  Run on a GPU box:
    python tools/pressure_overlap.py [genome_size] [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch

    import chain_util
    from nextdenovo_amd import overlap, synth
    gsize = float(sys.argv[1]) if len(sys.argv) > 1 else 4.6e6
    rs = chain_util.make_set(gsize, 50)
    words, word_off, lens = synth.pack_db(rs)
    dset = overlap.ReadSet(np.arange(len(rs), dtype=np.uint32), lens, words, word_off)

    def job():
        with overlap.Index(overlap.preset("ava-ont"), dset) as ix:
            raw = ix.map(dset, ix.mid_occ())
        srt, bl, _ = overlap.sort_overlaps([raw], lens.astype(np.uint32), int(lens.min()), 40, 300)
        return raw, srt, bl

    want = job()
    _live, _cached, peak = overlap.pool_bytes()
    rows = []
    print("idle: %d raw records, %d sorted, pool peak %.2f GB" % (want[0].shape[0], want[1].shape[0], peak / 2 ** 30), flush=True)
    for free_gb in (32, 16, 12, 8, 6, 5, 4, 3.5, 3, 2.5, 2, 1.5, 1, 0.75, 0.5, 0.25):
        overlap.trim()
        torch.cuda.empty_cache()
        free_b, total_b = torch.cuda.mem_get_info()
        take = free_b - int(free_gb * 2 ** 30)
        if take <= 0:
            continue
        hog = torch.empty(take, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        left = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        try:
            got = job()
            same = bool(all(np.array_equal(a, b) for a, b in zip(got[:2], want[:2])) and got[2] == want[2])
            verdict = "identical" if same else "DIFFERENT: %d raw / %d sorted records" % (got[0].shape[0], got[1].shape[0])
        except MemoryError as e:
            verdict = "MemoryError"
        except Exception as e:  # any other failure is loud too, but say which
            verdict = "%s: %s" % (type(e).__name__, str(e)[:120])
        rows.append({"free_gb": left / 2 ** 30, "verdict": verdict, "s": time.perf_counter() - t0})
        print("free %.2f GB -> %s (%.2f s)" % (left / 2 ** 30, verdict, rows[-1]["s"]), flush=True)
        del hog
    out = {"genome": gsize, "reads": len(rs), "idle_records": int(want[0].shape[0]), "pool_peak_gb": peak / 2 ** 30, "runs": rows,
           "silent_differences": sum(1 for r in rows if r["verdict"].startswith("DIFFERENT"))}
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps(out))
    return 1 if out["silent_differences"] else 0


if __name__ == "__main__":
    sys.exit(main())
