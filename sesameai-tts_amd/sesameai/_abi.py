"""ctypes binding of libcsm_hip.so (include/csm_hip.h, include/csm_hip_ops.h, include/mimi_hip.h).

The product path has NO fallback: if the HIP library is missing this module raises at import
time, and every call checks the returned status and raises with ``csm_last_error``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: libcsm_hip.so must bind to the HIP runtime torch has already loaded (its bundled
#                         libamdhip64); loading ours before torch puts two runtimes in the process ("no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
# CSM_HIP_TIMELINE=1: the build with the persistent decoder's debug stamps compiled in (make -C csrc timeline)
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib",
                        "libcsm_hip_timeline.so" if os.environ.get("CSM_HIP_TIMELINE", "0") == "1" else "libcsm_hip.so")
if os.environ.get("CSM_HIP_LIB"):           # an explicitly named build of the library (A/B runs of kernel variants)
    LIB_PATH = os.environ["CSM_HIP_LIB"]

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the gfx950 extension first "
        f"(python -c 'import __graft_entry__ as g; g.build()' or make -C sesameai-tts_amd/csrc). "
        f"There is no CPU fallback.")

lib = C.CDLL(LIB_PATH)

CSM_MAX_LAYERS = 32
ERRORS = {-1: "CSM_E_INVALID", -2: "CSM_E_HIP", -3: "CSM_E_STATE", -4: "CSM_E_TOO_LONG"}


class CsmLlamaDims(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_heads", C.c_int32), ("n_kv_heads", C.c_int32),
                ("dim", C.c_int32), ("ffn", C.c_int32), ("max_seq", C.c_int32), ("norm_eps", C.c_float)]


class CsmConfig(C.Structure):
    _fields_ = [("backbone", CsmLlamaDims), ("decoder", CsmLlamaDims),
                ("text_vocab", C.c_int32), ("audio_vocab", C.c_int32), ("n_codebooks", C.c_int32)]


class CsmLayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wq", "wk", "wv", "wo", "w1", "w2", "w3", "sa_norm", "mlp_norm")]


class CsmWeights(C.Structure):
    _fields_ = [("text_emb", C.c_void_p), ("audio_emb", C.c_void_p),
                ("bb", CsmLayerWeights * CSM_MAX_LAYERS), ("bb_norm", C.c_void_p),
                ("dec", CsmLayerWeights * CSM_MAX_LAYERS), ("dec_norm", C.c_void_p),
                ("projection", C.c_void_p), ("c0_head", C.c_void_p), ("audio_head_t", C.c_void_p),
                ("bb_rope", C.c_void_p), ("dec_rope", C.c_void_p),
                ("fp8", C.c_int32),
                ("bb8", CsmLayerWeights * CSM_MAX_LAYERS), ("bb8s", CsmLayerWeights * CSM_MAX_LAYERS),
                ("dec8", CsmLayerWeights * CSM_MAX_LAYERS), ("dec8s", CsmLayerWeights * CSM_MAX_LAYERS),
                ("c0_head8", C.c_void_p), ("c0_head8s", C.c_void_p),
                ("audio_head8", C.c_void_p), ("audio_head8s", C.c_void_p)]


_vp, _i, _f, _l, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_long, C.c_uint64

# name -> (restype, argtypes); every symbol the headers declare
SIGNATURES = {
    # include/csm_hip.h
    "csm_create": (_i, [C.POINTER(CsmConfig), C.POINTER(CsmWeights), _i, _i, _i, C.POINTER(_vp)]),
    "csm_destroy": (None, [_vp]),
    "csm_last_error": (C.c_char_p, [_vp]),
    "csm_reset": (_i, [_vp, _vp]),
    "csm_seed": (_i, [_vp, _u64, _vp]),
    "csm_prefill": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "csm_depth": (_i, [_vp, _i, _f, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "csm_frame_step": (_i, [_vp, _i, _f, _i, _i, _vp]),
    "csm_copy_frame": (_i, [_vp, _i, _vp, _vp]),
    "csm_set_step_inputs": (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    "csm_generate_frame_s1": (_i, [_vp, _vp, _vp, _vp, _i, _f, _i, _vp, _vp]),
    "csm_reset_slots": (_i, [_vp, _vp, _i, _vp]),
    "csm_prefill_slot": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp]),
    "csm_refill_supported": (_i, [_vp, _i]),
    "csm_refill_begin": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp]),
    "csm_refill_advance": (_i, [_vp, _i, _vp]),
    "csm_broadcast_weights": (_i, [_vp, C.c_size_t, _vp, _i, _vp]),
    "csm_num_frames": (_i, [_vp]),
    "csm_read_frames": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "csm_frames_dev": (_vp, [_vp]),
    "csm_last_h_dev": (_vp, [_vp]),
    "csm_bytes_per_frame": (C.c_double, [_vp, _i, C.c_double]),
    "csm_describe": (_i, [_vp, C.c_char_p, _i]),
    "csm_warn_unknown_switches": (None, []),
    # include/csm_hip_ops.h
    "csm_op_gemv": (_i, [_i, _i, _i, _i, _vp, _l, _l, _vp, _f, _vp, _vp, _vp, _vp, _vp, _l, _vp, _l, _i,
                         _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "csm_op_attn": (_i, [_i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csm_op_attn_oproj": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "csm_op_embed_sum": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "csm_op_sample": (_i, [_i, _i, _i, _vp, _f, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "csm_debug_persist_stamps": (_i, [_vp, _vp, _i]),
    "csm_debug_fast_paths": (_i, [_vp]),
    "csm_debug_graph_captures": (_i, [_vp]),
    "csm_debug_time_kernels": (_i, [_vp, _i, _i, _f, _i, _vp, _vp]),
}

# include/mimi_hip.h (bound when the symbols are present; tests/test_abi.py requires them)
MIMI_SIGNATURES = {
    "mimi_create": (_i, [_vp, _vp, _i, _i, C.POINTER(_vp)]),
    "mimi_destroy": (None, [_vp]),
    "mimi_last_error": (C.c_char_p, [_vp]),
    "mimi_decode": (_i, [_vp, _vp, _i, _i, _l, _l, _vp, _i, _vp]),
    "mimi_decode_strided": (_i, [_vp, _vp, _i, _i, _l, _l, _l, _vp, _i, _vp]),
    "mimi_reset_stream": (_i, [_vp, _vp]),
    "mimi_encode": (_i, [_vp, _vp, _l, _l, _i, _vp, _vp]),
}

for _name, (_res, _args) in list(SIGNATURES.items()) + list(MIMI_SIGNATURES.items()):
    _fn = getattr(lib, _name, None)
    if _fn is None:
        if _name in SIGNATURES:
            raise ImportError(f"{LIB_PATH} does not export {_name}")
        continue
    _fn.restype = _res
    _fn.argtypes = _args


class CsmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


def check(code: int, handle=None, mimi: bool = False) -> None:
    if code == 0:
        return
    fn = lib.mimi_last_error if mimi else lib.csm_last_error
    msg = fn(handle)
    raise CsmError(code, msg.decode() if msg else "")
