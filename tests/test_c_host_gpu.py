"""The C ABI driven from plain C (examples/c_host/csm_c_host.c: gcc, the HIP runtime and include/csm_hip.h -- no Python, no torch in
the process) against the Python host on the same weights: the drop-in boundary is the shared library, not the shim."""
import ctypes as C
import os
import struct
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "examples", "c_host", "csm_c_host")
MIMI_HOST = os.path.join(ROOT, "examples", "c_host", "mimi_c_host")


def _write_blob(path, model, tokens, mask):
    """config, the prompt and every tensor of CsmWeights in declaration order (include/csm_hip.h), each as int64 byte count + data."""
    def raw(t):
        t = t.detach().cpu().contiguous()
        return t.view(torch.uint8).numpy().tobytes() if t.dtype != torch.uint8 else t.numpy().tobytes()

    layer = ("attn.q_proj.weight", "attn.k_proj.weight", "attn.v_proj.weight", "attn.output_proj.weight",
             "mlp.w1.weight", "mlp.w2.weight", "mlp.w3.weight", "sa_norm.scale", "mlp_norm.scale")
    names = ["text_embeddings.weight", "audio_embeddings.weight"]
    names += [f"backbone.layers.{i}.{n}" for i in range(model.bb.num_layers) for n in layer] + ["backbone.norm.scale"]
    names += [f"decoder.layers.{i}.{n}" for i in range(model.dec.num_layers) for n in layer] + ["decoder.norm.scale"]
    names += ["projection.weight", "codebook0_head.weight", "audio_head_t", "bb_rope", "dec_rope"]
    with open(path, "wb") as f:
        f.write(b"CSMB")
        f.write(bytes(model._cfg_struct()))
        f.write(struct.pack("<i", tokens.shape[0]))
        f.write(raw(tokens.to(torch.int32))); f.write(raw(mask.to(torch.uint8)))
        for n in names:
            b = raw(model._w[n])
            f.write(struct.pack("<q", len(b))); f.write(b)


def test_plain_c_host_produces_the_python_hosts_frames(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(HOST):
        r = subprocess.run(["make", "-C", os.path.dirname(HOST)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    from sesameai.models import Model, csm_tiny_args, synthetic_state_dict
    m = Model(csm_tiny_args(), synthetic_state_dict(csm_tiny_args(), seed=1234), max_frames=32, max_prefill_rows=64)
    m.setup_caches(1)
    g = torch.Generator().manual_seed(8)
    S, n = 12, 9
    tok = torch.zeros(S, 33, dtype=torch.long); msk = torch.zeros(S, 33, dtype=torch.bool)
    tok[:5, 32] = torch.randint(0, 1000, (5,), generator=g); msk[:5, 32] = True
    tok[5:, :32] = torch.randint(0, 2048, (S - 5, 32), generator=g); msk[5:, :32] = True
    blob = str(tmp_path / "tiny.blob")
    _write_blob(blob, m, tok, msk)
    assert C.sizeof(type(m._cfg_struct())) == 2 * 28 + 12
    # the Python host: the same calls through the shim
    m.reset_caches(); m.seed(7)
    m.prefill_prompt(tok.unsqueeze(0), msk.unsqueeze(0))
    m.depth(1, 0.9, 50, commit=True)
    for _ in range(n - 1):
        m.step(1, 0.9, 50)
    want, eos = m.read_frames(1)
    # the C host, a process of its own (the system HIP runtime, no torch)
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",)}
    r = subprocess.run([HOST, blob, str(n)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # (RCCL may print a version banner on stdout when the host makes its communicator: keep the host's own lines)
    lines = [ln for ln in r.stdout.strip().splitlines() if ln[:1].isdigit() or ln[:1] == "-" or ln.startswith(("eos_at", "replicas"))]
    got = torch.tensor([[int(x) for x in ln.split()] for ln in lines[:n]], dtype=torch.int32)
    assert lines[n] == f"eos_at {int(eos[0])}" and lines[n + 1].startswith("replicas 1 broadcast_bytes ")
    assert got.shape == (n, 32) and torch.equal(got, want[:, 0]), "the plain-C host and the Python host disagree"


def test_plain_c_codec_host_decodes_the_python_hosts_pcm(tmp_path):
    """examples/c_host/mimi_c_host.c: include/mimi_hip.h from plain C.  The blob carries an image of the MimiWeights struct the shim
    built (pointers = the shim's device addresses) and the tensors behind them; the C host re-points every field at its own copies,
    decodes a 10-frame chunk three times (launch chain, capture, graph replay -- it checks they agree) and writes the PCM: the same
    bytes as MimiCodec.decode in this process."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    if not os.path.exists(MIMI_HOST):
        r = subprocess.run(["make", "-C", os.path.dirname(MIMI_HOST)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    from sesameai.mimi import MimiCodec, mimi_tiny_args, synthetic_state_dict
    sd = synthetic_state_dict(mimi_tiny_args(), seed=4321)
    codec = MimiCodec(mimi_tiny_args(), sd, max_frames=16)
    n0 = len(codec._keep)
    cfg, w = codec._build(sd)                                      # a second image of the structs, with the tensors behind it in _keep[n0:]
    tensors = codec._keep[n0:]
    T = 10
    codes = torch.randint(0, 2048, (1, 32, T), generator=torch.Generator().manual_seed(31))
    blob, out = str(tmp_path / "codec.blob"), str(tmp_path / "out.pcm")
    with open(blob, "wb") as f:
        f.write(b"MIMB")
        f.write(struct.pack("<ii", C.sizeof(cfg), C.sizeof(w)))
        f.write(bytes(cfg)); f.write(bytes(w))
        f.write(struct.pack("<i", len(tensors)))
        for t in tensors:
            b = t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()
            f.write(struct.pack("<Qq", t.data_ptr(), len(b))); f.write(b)
        f.write(struct.pack("<i", T))
        f.write(codes[0].to(torch.int32).contiguous().numpy().tobytes())
    want = codec.decode(codes)[0, 0].cpu()
    r = subprocess.run([MIMI_HOST, blob, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    import numpy as np
    got = torch.from_numpy(np.fromfile(out, dtype=np.float32))
    assert got.shape == want.shape and torch.equal(got, want), f"max |d| {(got - want).abs().max().item():.3e}"


def test_weight_broadcast_on_a_communicator_the_caller_owns():
    """csm_broadcast_weights (include/csm_hip.h): the library links no RCCL and creates no communicator -- it resolves
    ncclBroadcast from the instance already in the process.  Here the caller is this test: a one-rank ncclComm_t made through ctypes on
    the librccl torch has loaded (RTLD_NOLOAD: nothing new is loaded), one in-place broadcast of a 64 MB blob on a side stream, the
    bytes intact; a null communicator is CSM_E_INVALID with a message.  (N ranks: examples/c_host/csm_c_host.c with CSM_C_HOST_GPUS=N,
    and torch.distributed's own broadcast on the Python side.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sesameai import _abi

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    rccl = C.CDLL("librccl.so.1", mode=4 | 2)                                   # RTLD_NOLOAD | RTLD_NOW
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    uid, comm = UniqueId(), C.c_void_p()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        blob = torch.randint(-2**31, 2**31 - 1, (1 << 24,), dtype=torch.int32, device="cuda")
        want = blob.clone()
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        _abi.check(_abi.lib.csm_broadcast_weights(blob.data_ptr(), blob.numel() * 4, comm, 0, side.cuda_stream))
        side.synchronize()
        assert torch.equal(blob, want)
        rc = _abi.lib.csm_broadcast_weights(blob.data_ptr(), 16, None, 0, side.cuda_stream)
        assert rc == -1 and b"communicator" in _abi.lib.csm_last_error(None)
    finally:
        rccl.ncclCommDestroy(comm)
        C.CDLL(None).fflush(None)          # RCCL's version banner sits in C stdio's buffer: out now, into this test's captured output, not after pytest's summary line
