"""Mimi codec (decode and encode) on gfx950 -- the object the reference keeps in
``Generator._audio_tokenizer`` (moshi ``loaders.get_mimi``, sesameai/generator.py:52-57).

Surface used by the reference: ``.decode(codes[B,32,T]) -> [B,1,1920*T]`` (generator.py:116,299,
tts_service.py:245), ``.sample_rate`` (generator.py:59), ``.set_num_codebooks(32)`` (:55) and
``.encode(wav[B,1,n]) -> [B,32,ceil(n/1920)]`` (:86; voice prompts).
All arithmetic runs in libcsm_hip.so (include/mimi_hip.h); this file only re-lays-out the
checkpoint tensors into the tap-major form the kernels stream and owns the device memory.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _abi
from ._abi import check, lib

MAX_TR, MAX_ST = 16, 8


class _MimiConfig(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
                ("n_codebooks", C.c_int32), ("n_semantic", C.c_int32), ("tr_layers", C.c_int32),
                ("tr_heads", C.c_int32), ("tr_ffn", C.c_int32), ("tr_context", C.c_int32),
                ("rope_theta", C.c_float), ("norm_eps", C.c_float), ("n_stages", C.c_int32),
                ("ratios", C.c_int32 * MAX_ST), ("n_filters", C.c_int32), ("kernel", C.c_int32),
                ("last_kernel", C.c_int32), ("res_kernel", C.c_int32)]


class _MimiConv(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("c_in", C.c_int32), ("c_out", C.c_int32),
                ("taps", C.c_int32), ("phases", C.c_int32)]


class _MimiTrLayer(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ln1_w", "ln1_b", "in_proj", "out_proj", "ls1",
                                           "ln2_w", "ln2_b", "lin1", "lin2", "ls2")]


class _MimiWeights(C.Structure):
    _fields_ = [("codebooks", C.c_void_p), ("proj_first", C.c_void_p), ("proj_rest", C.c_void_p),
                ("rope_freqs", C.c_void_p), ("upsample", C.c_void_p), ("tr", _MimiTrLayer * MAX_TR),
                ("conv_in", _MimiConv), ("up", _MimiConv * MAX_ST), ("res1", _MimiConv * MAX_ST),
                ("res2", _MimiConv * MAX_ST), ("conv_out", _MimiConv),
                ("has_encoder", C.c_int32), ("enc_conv_in_w", C.c_void_p), ("enc_conv_in_b", C.c_void_p),
                ("enc_res1", _MimiConv * MAX_ST), ("enc_res2", _MimiConv * MAX_ST), ("enc_down", _MimiConv * MAX_ST),
                ("enc_conv_out", _MimiConv), ("enc_tr", _MimiTrLayer * MAX_TR), ("downsample", C.c_void_p),
                ("in_proj_first", C.c_void_p), ("in_proj_rest", C.c_void_p), ("codebook_sqnorm", C.c_void_p)]


lib.mimi_create.argtypes = [C.POINTER(_MimiConfig), C.POINTER(_MimiWeights), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
lib.mimi_encode.restype = C.c_int
lib.mimi_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
lib.mimi_decode_strided.restype = C.c_int
lib.mimi_decode_strided.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_long, C.c_long,
                                    C.c_void_p, C.c_int, C.c_void_p]


@dataclass(frozen=True)
class MimiArgs:
    hidden: int = 512
    codebook_size: int = 2048
    codebook_dim: int = 256
    num_codebooks: int = 32
    num_semantic: int = 1
    tr_layers: int = 8
    tr_heads: int = 8
    tr_ffn: int = 2048
    tr_context: int = 250
    rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    n_filters: int = 64
    ratios: Tuple[int, ...] = (8, 6, 5, 4)
    kernel: int = 7
    last_kernel: int = 3
    res_kernel: int = 3
    compress: int = 2
    sample_rate: int = 24000

    @property
    def hop(self) -> int:
        return 2 * int(math.prod(self.ratios))


def mimi_tiny_args() -> MimiArgs:
    return MimiArgs(hidden=128, codebook_dim=64, tr_layers=2, tr_heads=2, tr_ffn=256, tr_context=6)


def state_dict_layout(s: MimiArgs) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Canonical decode-side tensors (torch conv layouts: Conv1d [out,in,k], ConvTranspose1d
    [in,out,k]) with the kind of random init used for synthetic weights."""
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    for k in range(s.num_codebooks):
        out.append((f"rvq.{k}.embedding_sum", (s.codebook_size, s.codebook_dim), "normal"))
        out.append((f"rvq.{k}.cluster_usage", (s.codebook_size,), "usage"))
    out.append(("rvq_first.output_proj.weight", (s.hidden, s.codebook_dim, 1), "conv"))
    out.append(("rvq_rest.output_proj.weight", (s.hidden, s.codebook_dim, 1), "conv"))
    out.append(("upsample.convtr.weight", (s.hidden, 1, 4), "conv"))
    d = s.hidden
    for i in range(s.tr_layers):
        L = f"transformer.{i}"
        out += [(f"{L}.norm1.weight", (d,), "ones"), (f"{L}.norm1.bias", (d,), "small"),
                (f"{L}.in_proj_weight", (3 * d, d), "linear"), (f"{L}.out_proj.weight", (d, d), "linear"),
                (f"{L}.layer_scale_1.scale", (d,), "scale"),
                (f"{L}.norm2.weight", (d,), "ones"), (f"{L}.norm2.bias", (d,), "small"),
                (f"{L}.linear1.weight", (s.tr_ffn, d), "linear"), (f"{L}.linear2.weight", (d, s.tr_ffn), "linear"),
                (f"{L}.layer_scale_2.scale", (d,), "scale")]
    c = s.n_filters * 2 ** len(s.ratios)
    out += [("seanet.conv_in.weight", (c, d, s.kernel), "conv"), ("seanet.conv_in.bias", (c,), "small")]
    for j, r in enumerate(s.ratios):
        out += [(f"seanet.up.{j}.convtr.weight", (c, c // 2, 2 * r), "conv"), (f"seanet.up.{j}.convtr.bias", (c // 2,), "small")]
        c //= 2
        h = c // s.compress
        out += [(f"seanet.up.{j}.res.conv1.weight", (h, c, s.res_kernel), "conv"), (f"seanet.up.{j}.res.conv1.bias", (h,), "small"),
                (f"seanet.up.{j}.res.conv2.weight", (c, h, 1), "conv"), (f"seanet.up.{j}.res.conv2.bias", (c,), "small")]
    out += [("seanet.conv_out.weight", (1, c, s.last_kernel), "conv"), ("seanet.conv_out.bias", (1,), "small")]
    return out


def encoder_state_dict_layout(s: MimiArgs) -> List[Tuple[str, Tuple[int, ...], str]]:
    """Encode-side tensors (SEANet encoder, encoder transformer, stride-2 downsample, RVQ input projections)."""
    out: List[Tuple[str, Tuple[int, ...], str]] = []
    c, d = s.n_filters, s.hidden
    out += [("enc.conv_in.weight", (c, 1, s.kernel), "conv"), ("enc.conv_in.bias", (c,), "small")]
    for j, r in enumerate(reversed(s.ratios)):
        h = c // s.compress
        out += [(f"enc.down.{j}.res.conv1.weight", (h, c, s.res_kernel), "conv"), (f"enc.down.{j}.res.conv1.bias", (h,), "small"),
                (f"enc.down.{j}.res.conv2.weight", (c, h, 1), "conv"), (f"enc.down.{j}.res.conv2.bias", (c,), "small"),
                (f"enc.down.{j}.conv.weight", (2 * c, c, 2 * r), "conv"), (f"enc.down.{j}.conv.bias", (2 * c,), "small")]
        c *= 2
    out += [("enc.conv_out.weight", (d, c, s.last_kernel), "conv"), ("enc.conv_out.bias", (d,), "small")]
    for i in range(s.tr_layers):
        L = f"enc_transformer.{i}"
        out += [(f"{L}.norm1.weight", (d,), "ones"), (f"{L}.norm1.bias", (d,), "small"),
                (f"{L}.in_proj_weight", (3 * d, d), "linear"), (f"{L}.out_proj.weight", (d, d), "linear"),
                (f"{L}.layer_scale_1.scale", (d,), "scale"),
                (f"{L}.norm2.weight", (d,), "ones"), (f"{L}.norm2.bias", (d,), "small"),
                (f"{L}.linear1.weight", (s.tr_ffn, d), "linear"), (f"{L}.linear2.weight", (d, s.tr_ffn), "linear"),
                (f"{L}.layer_scale_2.scale", (d,), "scale")]
    out += [("downsample.conv.weight", (d, d, 4), "conv"),
            ("rvq_first.input_proj.weight", (s.codebook_dim, d, 1), "conv"),
            ("rvq_rest.input_proj.weight", (s.codebook_dim, d, 1), "conv")]
    return out


def synthetic_state_dict(s: MimiArgs, seed: int = 4321, encoder: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded random fp32 weights of the true shapes (no checkpoint can be downloaded here):
    codebooks N(0,1), convs/linears U(+-sqrt(3/fan_in)) so activations stay O(1).  The encode-side
    tensors come from a second generator (seed + 1)."""
    w: Dict[str, torch.Tensor] = {}
    _fill(w, state_dict_layout(s), torch.Generator(device="cpu").manual_seed(seed))
    if encoder:
        _fill(w, encoder_state_dict_layout(s), torch.Generator(device="cpu").manual_seed(seed + 1))
    return w


def _fill(w: Dict[str, torch.Tensor], names, g: torch.Generator) -> None:
    for name, shp, kind in names:
        if kind == "normal":
            t = torch.randn(shp, generator=g)
        elif kind == "usage":
            t = 0.5 + torch.rand(shp, generator=g)
        elif kind == "ones":
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif kind == "small":
            t = 0.05 * torch.randn(shp, generator=g)
        elif kind == "scale":
            t = 0.3 + 0.1 * torch.rand(shp, generator=g)
        else:
            if name.endswith("convtr.weight"):
                fan_in = 2 * (1 if name.startswith("upsample") else shp[0])
            elif kind == "conv":
                fan_in = shp[1] * shp[2]
            else:
                fan_in = shp[1]
            t = (torch.rand(shp, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in)
        w[name] = t.float()


def from_moshi_state_dict(sd: Dict[str, torch.Tensor], s: MimiArgs) -> Dict[str, torch.Tensor]:
    """Maps a moshi ``tokenizer-*.safetensors`` (kyutai/moshiko, loaders.MIMI_NAME) state dict to the
    canonical names above: the decode side always, the encode side (SEANet encoder, encoder
    transformer, stride-2 downsample, RVQ input projections -- what ``Segment.audio`` voice prompts
    need, sesameai/generator.py:86) when the checkpoint holds it.  Written from moshi 0.2.2's module
    tree (SEANetEncoder/Decoder ``model`` index lists, ``ProjectedTransformer``, ``SplitResidualVectorQuantizer``);
    no checkpoint can be downloaded here, so it is pinned by a name/shape round trip
    (tests/test_host_logic.py::test_moshi_checkpoint_name_map_round_trip)."""
    out: Dict[str, torch.Tensor] = {}
    for k in range(s.num_codebooks):
        src = ("quantizer.rvq_first.vq.layers.0" if k == 0 else f"quantizer.rvq_rest.vq.layers.{k - 1}") + "._codebook"
        out[f"rvq.{k}.embedding_sum"] = sd[f"{src}.embedding_sum"]
        out[f"rvq.{k}.cluster_usage"] = sd[f"{src}.cluster_usage"]
    out["rvq_first.output_proj.weight"] = sd["quantizer.rvq_first.output_proj.weight"]
    out["rvq_rest.output_proj.weight"] = sd["quantizer.rvq_rest.output_proj.weight"]
    out["upsample.convtr.weight"] = sd["upsample.convtr.convtr.convtr.weight"]
    for i in range(s.tr_layers):
        p, L = f"decoder_transformer.transformer.layers.{i}", f"transformer.{i}"
        out[f"{L}.norm1.weight"], out[f"{L}.norm1.bias"] = sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"]
        out[f"{L}.norm2.weight"], out[f"{L}.norm2.bias"] = sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"]
        out[f"{L}.in_proj_weight"] = sd[f"{p}.self_attn.in_proj_weight"]
        out[f"{L}.out_proj.weight"] = sd[f"{p}.self_attn.out_proj.weight"]
        out[f"{L}.linear1.weight"], out[f"{L}.linear2.weight"] = sd[f"{p}.linear1.weight"], sd[f"{p}.linear2.weight"]
        out[f"{L}.layer_scale_1.scale"], out[f"{L}.layer_scale_2.scale"] = sd[f"{p}.layer_scale_1.scale"], sd[f"{p}.layer_scale_2.scale"]
    # SEANetDecoder.model: [conv_in, (ELU, convtr, resblock) x4, ELU, conv_out]
    out["seanet.conv_in.weight"], out["seanet.conv_in.bias"] = sd["decoder.model.0.conv.conv.weight"], sd["decoder.model.0.conv.conv.bias"]
    for j in range(len(s.ratios)):
        ct, rb = 2 + 3 * j, 3 + 3 * j
        out[f"seanet.up.{j}.convtr.weight"] = sd[f"decoder.model.{ct}.convtr.convtr.weight"]
        out[f"seanet.up.{j}.convtr.bias"] = sd[f"decoder.model.{ct}.convtr.convtr.bias"]
        out[f"seanet.up.{j}.res.conv1.weight"] = sd[f"decoder.model.{rb}.block.1.conv.conv.weight"]
        out[f"seanet.up.{j}.res.conv1.bias"] = sd[f"decoder.model.{rb}.block.1.conv.conv.bias"]
        out[f"seanet.up.{j}.res.conv2.weight"] = sd[f"decoder.model.{rb}.block.3.conv.conv.weight"]
        out[f"seanet.up.{j}.res.conv2.bias"] = sd[f"decoder.model.{rb}.block.3.conv.conv.bias"]
    last = 2 + 3 * len(s.ratios)
    out["seanet.conv_out.weight"], out["seanet.conv_out.bias"] = sd[f"decoder.model.{last}.conv.conv.weight"], sd[f"decoder.model.{last}.conv.conv.bias"]
    if "encoder.model.0.conv.conv.weight" not in sd:
        return out
    # SEANetEncoder.model: [conv_in, (resblock, ELU, strided conv) x4, ELU, conv_out]
    out["enc.conv_in.weight"], out["enc.conv_in.bias"] = sd["encoder.model.0.conv.conv.weight"], sd["encoder.model.0.conv.conv.bias"]
    for j in range(len(s.ratios)):
        rb, dn = 1 + 3 * j, 3 + 3 * j
        out[f"enc.down.{j}.res.conv1.weight"] = sd[f"encoder.model.{rb}.block.1.conv.conv.weight"]
        out[f"enc.down.{j}.res.conv1.bias"] = sd[f"encoder.model.{rb}.block.1.conv.conv.bias"]
        out[f"enc.down.{j}.res.conv2.weight"] = sd[f"encoder.model.{rb}.block.3.conv.conv.weight"]
        out[f"enc.down.{j}.res.conv2.bias"] = sd[f"encoder.model.{rb}.block.3.conv.conv.bias"]
        out[f"enc.down.{j}.conv.weight"] = sd[f"encoder.model.{dn}.conv.conv.weight"]
        out[f"enc.down.{j}.conv.bias"] = sd[f"encoder.model.{dn}.conv.conv.bias"]
    out["enc.conv_out.weight"], out["enc.conv_out.bias"] = sd[f"encoder.model.{last}.conv.conv.weight"], sd[f"encoder.model.{last}.conv.conv.bias"]
    for i in range(s.tr_layers):
        p, L = f"encoder_transformer.transformer.layers.{i}", f"enc_transformer.{i}"
        out[f"{L}.norm1.weight"], out[f"{L}.norm1.bias"] = sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"]
        out[f"{L}.norm2.weight"], out[f"{L}.norm2.bias"] = sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"]
        out[f"{L}.in_proj_weight"] = sd[f"{p}.self_attn.in_proj_weight"]
        out[f"{L}.out_proj.weight"] = sd[f"{p}.self_attn.out_proj.weight"]
        out[f"{L}.linear1.weight"], out[f"{L}.linear2.weight"] = sd[f"{p}.linear1.weight"], sd[f"{p}.linear2.weight"]
        out[f"{L}.layer_scale_1.scale"], out[f"{L}.layer_scale_2.scale"] = sd[f"{p}.layer_scale_1.scale"], sd[f"{p}.layer_scale_2.scale"]
    out["downsample.conv.weight"] = sd["downsample.conv.conv.conv.weight"]
    out["rvq_first.input_proj.weight"] = sd["quantizer.rvq_first.input_proj.weight"]
    out["rvq_rest.input_proj.weight"] = sd["quantizer.rvq_rest.input_proj.weight"]
    return out


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class MimiCodec:
    def __init__(self, args: MimiArgs, state_dict: Optional[Dict[str, torch.Tensor]] = None, device: str = "cuda",
                 max_frames: int = 1125):
        if not torch.cuda.is_available():
            raise RuntimeError("MimiCodec (MI355X build) needs a ROCm GPU: there is no CPU fallback")
        self.args = args
        self.device = torch.device(device)
        self.sample_rate = args.sample_rate
        self.frame_rate = 12.5
        self.max_frames = max_frames
        sd = state_dict if state_dict is not None else synthetic_state_dict(args)
        for name, shp, _ in state_dict_layout(args):
            if name not in sd or tuple(sd[name].shape) != tuple(shp):
                raise ValueError(f"Mimi checkpoint: tensor {name} missing or not of shape {shp}")
        self._keep: List[torch.Tensor] = []
        self._h = C.c_void_p(None)
        cfg, w = self._build(sd)
        with torch.cuda.device(self.device):
            check(lib.mimi_create(C.byref(cfg), C.byref(w), max_frames, 0, C.byref(self._h)), None, mimi=True)

    @classmethod
    def from_pretrained(cls, path: Optional[str], device: str = "cuda", **kw) -> "MimiCodec":
        if not path:
            return cls(MimiArgs(), None, device=device, **kw)
        from safetensors.torch import load_file
        return cls(MimiArgs(), from_moshi_state_dict(load_file(path), MimiArgs()), device=device, **kw)

    # -- weight re-layout (host-side plumbing) -----------------------------------------------------
    def _dev(self, t: torch.Tensor) -> int:
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        self._keep.append(t)
        return t.data_ptr()

    def _conv(self, w: torch.Tensor, b: Optional[torch.Tensor]) -> _MimiConv:
        """Conv1d [out,in,k] -> [1][k][out][in] (tap j reads x[t + j - (k-1)])."""
        co, ci, k = w.shape
        return _MimiConv(self._dev(w.permute(2, 0, 1).unsqueeze(0)), self._dev(b) if b is not None else None, ci, co, k, 1)

    def _convtr(self, w: torch.Tensor, b: Optional[torch.Tensor], stride: int) -> _MimiConv:
        """ConvTranspose1d [in,out,2s], stride s -> [s phases][2 taps][out][in]:
        out[t*s+p] = x[t] . w[:,:,p] + x[t-1] . w[:,:,p+s]."""
        ci, co, k = w.shape
        assert k == 2 * stride
        wt = w.permute(2, 1, 0)                                    # [k][out][in]
        packed = torch.stack([wt[:stride], wt[stride:]], dim=1)     # [s][2][out][in]
        return _MimiConv(self._dev(packed), self._dev(b) if b is not None else None, ci, co, 2, stride)

    def _build(self, sd: Dict[str, torch.Tensor]):
        s = self.args
        cfg = _MimiConfig(s.hidden, s.codebook_size, s.codebook_dim, s.num_codebooks, s.num_semantic, s.tr_layers,
                          s.tr_heads, s.tr_ffn, s.tr_context, s.rope_theta, s.norm_eps, len(s.ratios))
        for j, r in enumerate(s.ratios):
            cfg.ratios[j] = r
        cfg.n_filters, cfg.kernel, cfg.last_kernel, cfg.res_kernel = s.n_filters, s.kernel, s.last_kernel, s.res_kernel
        w = _MimiWeights()
        books = torch.stack([sd[f"rvq.{k}.embedding_sum"].float() / sd[f"rvq.{k}.cluster_usage"].float().clamp(min=1e-5)[:, None]
                             for k in range(s.num_codebooks)])
        w.codebooks = self._dev(books)
        w.proj_first = self._dev(sd["rvq_first.output_proj.weight"][:, :, 0].t())
        w.proj_rest = self._dev(sd["rvq_rest.output_proj.weight"][:, :, 0].t())
        hd = s.hidden // s.tr_heads
        w.rope_freqs = self._dev(torch.exp(torch.arange(hd // 2, dtype=torch.float32) * (-math.log(s.rope_theta) * 2 / hd)))
        up = sd["upsample.convtr.weight"][:, 0, :]                  # [C][4]
        w.upsample = self._dev(torch.stack([torch.stack([up[:, 0], up[:, 2]]), torch.stack([up[:, 1], up[:, 3]])]))
        for i in range(s.tr_layers):
            L = f"transformer.{i}"
            w.tr[i] = _MimiTrLayer(*[self._dev(sd[f"{L}.{n}"]) for n in (
                "norm1.weight", "norm1.bias", "in_proj_weight", "out_proj.weight", "layer_scale_1.scale",
                "norm2.weight", "norm2.bias", "linear1.weight", "linear2.weight", "layer_scale_2.scale")])
        w.conv_in = self._conv(sd["seanet.conv_in.weight"], sd["seanet.conv_in.bias"])
        for j, r in enumerate(s.ratios):
            w.up[j] = self._convtr(sd[f"seanet.up.{j}.convtr.weight"], sd[f"seanet.up.{j}.convtr.bias"], r)
            w.res1[j] = self._conv(sd[f"seanet.up.{j}.res.conv1.weight"], sd[f"seanet.up.{j}.res.conv1.bias"])
            w.res2[j] = self._conv(sd[f"seanet.up.{j}.res.conv2.weight"], sd[f"seanet.up.{j}.res.conv2.bias"])
        w.conv_out = self._conv(sd["seanet.conv_out.weight"], sd["seanet.conv_out.bias"])
        self.has_encoder = all(n in sd for n, _, _ in encoder_state_dict_layout(s))
        w.has_encoder = int(self.has_encoder)
        if self.has_encoder:
            w.enc_conv_in_w = self._dev(sd["enc.conv_in.weight"][:, 0, :].t())          # [taps][C]
            w.enc_conv_in_b = self._dev(sd["enc.conv_in.bias"])
            for j, r in enumerate(reversed(s.ratios)):
                w.enc_res1[j] = self._conv(sd[f"enc.down.{j}.res.conv1.weight"], sd[f"enc.down.{j}.res.conv1.bias"])
                w.enc_res2[j] = self._conv(sd[f"enc.down.{j}.res.conv2.weight"], sd[f"enc.down.{j}.res.conv2.bias"])
                w.enc_down[j] = self._conv(sd[f"enc.down.{j}.conv.weight"], sd[f"enc.down.{j}.conv.bias"])   # taps = 2r, stride r
            w.enc_conv_out = self._conv(sd["enc.conv_out.weight"], sd["enc.conv_out.bias"])
            for i in range(s.tr_layers):
                L = f"enc_transformer.{i}"
                w.enc_tr[i] = _MimiTrLayer(*[self._dev(sd[f"{L}.{n}"]) for n in (
                    "norm1.weight", "norm1.bias", "in_proj_weight", "out_proj.weight", "layer_scale_1.scale",
                    "norm2.weight", "norm2.bias", "linear1.weight", "linear2.weight", "layer_scale_2.scale")])
            w.downsample = self._dev(sd["downsample.conv.weight"].permute(2, 0, 1))       # [4][out][in]
            w.in_proj_first = self._dev(sd["rvq_first.input_proj.weight"][:, :, 0])       # [cd][d]
            w.in_proj_rest = self._dev(sd["rvq_rest.input_proj.weight"][:, :, 0])
            w.codebook_sqnorm = self._dev(books.pow(2).sum(-1))                           # [K][2048]
        return cfg, w

    def decode_weight_bytes(self) -> int:
        """fp32 bytes one decode call reads whatever its length: every decode-side tensor except the codebooks, of which a call
        touches only the rows it looks up (32 x 1 KB per frame: activations, not weights).  bench.py's `mimi.roofline`."""
        import math as _m
        return sum(4 * _m.prod(shp) for name, shp, _ in state_dict_layout(self.args) if not name.startswith("rvq."))

    # -- reference surface ----------------------------------------------------------------------------
    def set_num_codebooks(self, n: int) -> None:
        if n != self.args.num_codebooks:
            raise ValueError(f"this codec is built for {self.args.num_codebooks} codebooks")

    @torch.inference_mode()
    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        """wav (B,1,n) fp32 @ 24 kHz -> codes (B,32,ceil(n/1920)) int64 (generator.py:86)."""
        if not self.has_encoder:
            raise RuntimeError("this MimiCodec was built without encoder weights; pass Segment.audio_codes instead")
        assert wav.dim() == 3 and wav.shape[1] == 1, "wav must be (B, 1, n)"
        B, _, n = wav.shape
        T = -(-n // self.args.hop)
        x = wav.to(device=self.device, dtype=torch.float32).contiguous()
        codes = torch.empty(B, self.args.num_codebooks, T, dtype=torch.int32, device=self.device)
        check(lib.mimi_encode(self._h, x.data_ptr(), n, x.stride(0), B, codes.data_ptr(), _stream()), self._h, mimi=True)
        return codes.long()

    def _run(self, codes: torch.Tensor, stateful: bool) -> torch.Tensor:
        assert codes.dim() == 3 and codes.shape[1] == self.args.num_codebooks, "codes must be (B, 32, T)"
        B, K, T = codes.shape
        c = codes.to(device=self.device, dtype=torch.int32)
        pcm = torch.empty(B, 1, self.args.hop * T, dtype=torch.float32, device=self.device)
        check(lib.mimi_decode_strided(self._h, c.data_ptr(), B, T, c.stride(0), c.stride(1), c.stride(2), pcm.data_ptr(),
                                      int(stateful), _stream()), self._h, mimi=True)
        return pcm

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes (B,32,T) int -> (B,1,1920*T) fp32; stateless (generator.py:116,299)."""
        return self._run(codes, stateful=False)

    def reset_stream(self) -> None:
        check(lib.mimi_reset_stream(self._h, _stream()), self._h, mimi=True)

    @torch.inference_mode()
    def decode_stream(self, codes: torch.Tensor) -> torch.Tensor:
        """Stateful streaming decode (B == 1): successive calls continue one stream, and the
        concatenated output equals ``decode`` of the concatenated codes.  Call reset_stream() first."""
        return self._run(codes, stateful=True)

    def __del__(self):
        try:
            if self._h:
                lib.mimi_destroy(self._h)
        except Exception:
            pass
