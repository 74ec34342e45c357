"""One-paragraph summary of a bench.py line: python tools/bench_brief.py FILE [FILE ...]"""
import json
import sys

for p in sys.argv[1:]:
    try:
        d = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(" %s: %r" % (p, e))
        continue
    sm = d.get("step_ms", {})
    print(" %s: ms_per_step %.1f median %s min %s max %s | value %.1f M | cns %.1f ovl %.1f sort %.1f asm %.1f fasta %.1f" % (
        p.split("/")[-1], d["ms_per_step"], sm.get("median"), sm.get("min"), sm.get("max"), d["value"] / 1e6, d["consensus_ms_per_step"],
        d["overlap"]["ms_per_step"], d["overlap"]["sort"]["ms_per_step"], d["overlap"]["pile_assembly_ms_per_step"],
        d.get("fasta_write", {}).get("ms_per_step", 0.0)))
    print("   steps %s" % sm.get("list"))
    print("   kernel_ms/step %s" % {k: round(v / d["steps"], 1) for k, v in d["kernel_ms"].items()})
    c = d["counters"]
    print("   traceback in segments: tasks %s walkers %s fallbacks %s" % (c.get("tb_tasks"), c.get("tb_walkers"), c.get("tb_fallbacks")))
    print("   lq rounds %s declined %s jobs %s repairs %s | host %s" % (c.get("lq_rounds"), c.get("lq_declined"), c.get("lq_jobs"), c.get("lq_repairs"),
                                                                        {k: d.get("host", {}).get(k) for k in ("cpu_model", "cpu_count", "host_threads", "before", "after")}))
    if "cpu_baseline" in d and d["cpu_baseline"]:
        cb = d["cpu_baseline"]
        print("   cpu_baseline %.2f M/s on %d cores (wall %.1f s), per core %.0f -> x cores %.2f M/s | parity %s" % (
            cb["value"] / 1e6, cb["cores"], cb.get("wall_s", 0), cb.get("per_core_measured", 0), cb.get("per_core_measured_x_cores", 0) / 1e6,
            {k: d.get("parity", {}).get(k) for k in ("piles", "mismatch")}))
    r = d["roofline"]
    print("   roofline %s: alg %.1f MB/launch, %.2f ms/launch, frac %.5f, traffic %s" % (r["kernel"], r["alg_bytes_per_launch"] / 1e6, r["avg_launch_ms"], r["frac"], r.get("traffic")))
