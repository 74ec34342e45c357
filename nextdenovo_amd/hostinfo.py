"""The CPUs this process can actually have (the host-side mirror of effective_cpus() in csrc/consensus.cpp): the smaller of the
hardware threads, the scheduling affinity and the cgroup CPU quota.  A container on a 256-thread host with a quota of 16 CPUs that
starts 64 worker processes gets 16 CPUs' worth of time and is throttled as a whole for the rest of every period."""
from __future__ import annotations

import math
import os


def cgroup_cpu_quota(root: str = "/sys/fs/cgroup"):
    """CPUs the cgroup grants (float), or None when there is no quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cfs_period_us)."""
    try:
        with open(os.path.join(root, "cpu.max")) as f:
            a, b = f.read().split()[:2]
        return None if a == "max" else float(a) / float(b)
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            q = float(f.read().strip())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            p = float(f.read().strip())
        return q / p if q > 0 and p > 0 else None
    except (OSError, ValueError):
        return None


def effective_cpus(root: str = "/sys/fs/cgroup") -> int:
    if os.environ.get("NDGPU_HOST_CPUS"):
        return max(1, int(os.environ["NDGPU_HOST_CPUS"]))
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    q = cgroup_cpu_quota(root)
    if q:
        n = min(n, max(1, int(math.ceil(q - 1e-9))))
    return max(1, n)


def throttle_stat():
    """(nr_periods, nr_throttled, throttled_usec) of this cgroup, or None."""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            kv = {}
            with open(path) as f:
                for ln in f:
                    parts = ln.split()
                    if len(parts) == 2:
                        kv[parts[0]] = int(parts[1])
            if "nr_throttled" in kv:
                return kv.get("nr_periods", 0), kv["nr_throttled"], kv.get("throttled_usec", kv.get("throttled_time", 0) // 1000)
        except (OSError, ValueError):
            continue
    return None
