"""pytest wiring: marker registration + import paths.

``-m "not gpu"`` = oracle vs golden vectors / HF cross-check / host logic / C-ABI symbol
checks (no GPU).  ``-m gpu`` = HIP-path parity through the C ABI on a real MI355X.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "sesameai-tts_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _cpu_quota():
    """CPUs this container may use per scheduling period (cgroup v2 cpu.max), or None (bench.cpu_quota says why it matters)."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        return None


def pytest_sessionstart(session):
    # the oracle's torch CPU ops on more OpenMP threads than the container's CPU quota get the process throttled (GPU boxes of this
    # pool: 128 threads against a quota of 16): cap the team, the results do not depend on it
    q = _cpu_quota()
    if q:
        import torch
        if torch.get_num_threads() > q:
            torch.set_num_threads(q)
