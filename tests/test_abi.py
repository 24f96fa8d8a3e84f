"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
the headers under include/ declare (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sesameai-tts_amd", "lib", "libcsm_hip.so")


def declared_symbols():
    syms = []
    inc = os.path.join(ROOT, "include")
    for h in sorted(os.listdir(inc)):
        text = open(os.path.join(inc, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms += re.findall(r"\b((?:csm|mimi)_[a-z0-9_]+)\s*\(", text)
    return sorted(set(syms))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_headers_declare_something():
    syms = declared_symbols()
    assert "csm_frame_step" in syms and "csm_prefill" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libcsm_hip.so does not export: {missing}"


def test_binding_covers_every_declared_symbol(lib):
    from sesameai import _abi
    bound = set(_abi.SIGNATURES) | set(_abi.MIMI_SIGNATURES)
    assert set(declared_symbols()) <= bound, sorted(set(declared_symbols()) - bound)


def test_create_rejects_bad_config_without_touching_the_gpu(lib):
    from sesameai import _abi
    cfg = _abi.CsmConfig()
    cfg.backbone = _abi.CsmLlamaDims(2, 8, 2, 500, 1024, 256, 1e-5)       # dim % 512 != 0
    cfg.decoder = _abi.CsmLlamaDims(2, 4, 2, 512, 1024, 256, 1e-5)
    cfg.text_vocab, cfg.audio_vocab, cfg.n_codebooks = 1000, 2051, 32
    w = _abi.CsmWeights()
    h = ctypes.c_void_p(None)
    rc = _abi.lib.csm_create(ctypes.byref(cfg), ctypes.byref(w), 1, 8, 8, ctypes.byref(h))
    assert rc == -1 and b"dims" in _abi.lib.csm_last_error(None)
    rc = _abi.lib.csm_create(None, None, 1, 8, 8, ctypes.byref(h))
    assert rc == -1


def test_plain_c_host_builds_against_the_header_and_the_library():
    """examples/c_host/csm_c_host.c and mimi_c_host.c are the C ABI used from plain C (gcc -std=c11, include/csm_hip.h, -lcsm_hip): it must compile and
    link without a GPU; tests/test_c_host_gpu.py runs it beside the Python host."""
    import subprocess
    d = os.path.join(ROOT, "examples", "c_host")
    r = subprocess.run(["make", "-B", "-C", d], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(os.path.join(d, "csm_c_host")) and os.path.exists(os.path.join(d, "mimi_c_host"))
