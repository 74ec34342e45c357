"""The CPUs the process can have (nextdenovo_amd/hostinfo.py; the C++ side is effective_cpus() in csrc/consensus.cpp): the GPU boxes of
the pool are 256-thread hosts whose containers have a cgroup quota of 16 CPUs, and thread pools sized by os.cpu_count() got the whole
process throttled."""
import os

from nextdenovo_amd import hostinfo


def _tree(tmp_path, v2=None, v1=None):
    root = tmp_path / "cg"
    (root / "cpu").mkdir(parents=True)
    if v2 is not None:
        (root / "cpu.max").write_text(v2)
    if v1 is not None:
        (root / "cpu" / "cpu.cfs_quota_us").write_text(str(v1[0]))
        (root / "cpu" / "cpu.cfs_period_us").write_text(str(v1[1]))
    return str(root)


def test_quota_of_both_cgroup_versions(tmp_path, monkeypatch):
    monkeypatch.delenv("NDGPU_HOST_CPUS", raising=False)
    assert hostinfo.cgroup_cpu_quota(_tree(tmp_path / "a", v2="1600000 100000\n")) == 16.0      # what the MI355X boxes say
    assert hostinfo.cgroup_cpu_quota(_tree(tmp_path / "b", v2="max 100000\n")) is None
    assert hostinfo.cgroup_cpu_quota(_tree(tmp_path / "c", v1=(250000, 100000))) == 2.5
    assert hostinfo.cgroup_cpu_quota(_tree(tmp_path / "d", v1=(-1, 100000))) is None                # cgroup v1: no quota
    assert hostinfo.cgroup_cpu_quota(str(tmp_path / "nowhere")) is None
    have = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)))
    assert hostinfo.effective_cpus(_tree(tmp_path / "e", v2="max 100000")) == have
    assert hostinfo.effective_cpus(_tree(tmp_path / "f", v1=(250000, 100000))) == min(have, 3)      # a fraction of a CPU counts as one more
    assert hostinfo.effective_cpus(_tree(tmp_path / "g", v2="100000 100000")) == 1
    monkeypatch.setenv("NDGPU_HOST_CPUS", "5")
    assert hostinfo.effective_cpus(_tree(tmp_path / "h", v2="100000 100000")) == 5                  # the override
