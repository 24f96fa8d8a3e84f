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


def test_a_mistyped_switch_is_reported_once_on_stderr():
    """CSM_* / MIMI_* switches are read with getenv at create time; a name no switch reads would silently leave the default in force.  The first
    csm_create / mimi_create of a process lists such names once (include/csm_hip.h csm_warn_unknown_switches)."""
    import subprocess, sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from sesameai import _abi; h = ctypes.c_void_p(None);"
            "_abi.lib.csm_create(None, None, 1, 8, 8, ctypes.byref(h)); _abi.lib.csm_create(None, None, 1, 8, 8, ctypes.byref(h))") % os.path.join(ROOT, "sesameai-tts_amd")
    env = dict(os.environ, CSM_PERSITS="0", MIMI_KSPLITT="4", CSM_PERSIST="1", CSM_QUIET="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stderr.splitlines() if "no switch reads" in ln]
    assert len(lines) == 1 and "CSM_PERSITS" in lines[0] and "MIMI_KSPLITT" in lines[0] and "CSM_PERSIST " not in lines[0] + " " and "CSM_QUIET" not in lines[0]
