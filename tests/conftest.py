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
