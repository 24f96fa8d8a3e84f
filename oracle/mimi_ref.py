"""CPU oracle for the Mimi codec DECODE path  --  TEST INFRASTRUCTURE ONLY.

Restates what ``self._audio_tokenizer.decode(codes)`` computes at the reference call sites
sesameai/generator.py:116,299 and tts_service.py:245 (``moshi==0.2.2`` ``MimiModel.decode``,
requirements.txt:6 -- not vendored, not installed here).  fp32 throughout (the reference
never casts Mimi: sesameai/generator.py:53).  Structure per SURVEY.md App. A.3:

    codes (B,32,T) -> split RVQ lookup-sum + two 1x1 output projections -> (B,512,T)
      -> depthwise ConvTranspose1d k4 s2 (causal trim)                  -> (B,512,2T)
      -> 8-layer causal transformer (window 250, interleaved RoPE, LayerScale)
      -> SEANet decoder (ratios 8,6,5,4; ELU; causal zero-padded convs)  -> (B,1,1920T)

PARITY STATUS: unpinned by the reference (no tests/fixtures there); pinned against the
independent ``transformers.models.mimi`` port (tests/test_oracle_vs_hf.py) and the golden
vectors in tests/golden/.  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class MimiShape:
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
    layer_scale_init: float = 0.01
    n_filters: int = 64
    ratios: Tuple[int, ...] = (8, 6, 5, 4)
    kernel: int = 7
    last_kernel: int = 3
    res_kernel: int = 3
    compress: int = 2
    sample_rate: int = 24000
    frame_rate: float = 12.5

    @property
    def hop(self) -> int:          # samples per code frame
        return 2 * int(math.prod(self.ratios))


def mimi_full() -> MimiShape:
    return MimiShape()


def mimi_tiny() -> MimiShape:
    """Same topology and SEANet widths, small transformer / RVQ (second-scale CPU tests); hop
    stays 1920, head_dim stays 64, every channel count stays a multiple of 32."""
    return MimiShape(hidden=128, codebook_size=2048, codebook_dim=64, tr_layers=2, tr_heads=2,
                     tr_ffn=256, tr_context=6, n_filters=64)


# ----------------------------------------------------------------------------------------
def weight_names(s: MimiShape) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, init kind).  Names follow the moshi module tree loosely; conv weights
    are torch layout: Conv1d [out,in,k], ConvTranspose1d [in,out,k]."""
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
                (f"{L}.in_proj_weight", (3 * d, d), "linear"),
                (f"{L}.out_proj.weight", (d, d), "linear"),
                (f"{L}.layer_scale_1.scale", (d,), "scale"),
                (f"{L}.norm2.weight", (d,), "ones"), (f"{L}.norm2.bias", (d,), "small"),
                (f"{L}.linear1.weight", (s.tr_ffn, d), "linear"),
                (f"{L}.linear2.weight", (d, s.tr_ffn), "linear"),
                (f"{L}.layer_scale_2.scale", (d,), "scale")]
    c = s.n_filters * 2 ** len(s.ratios)
    out += [("seanet.conv_in.weight", (c, d, s.kernel), "conv"), ("seanet.conv_in.bias", (c,), "small")]
    for j, r in enumerate(s.ratios):
        out += [(f"seanet.up.{j}.convtr.weight", (c, c // 2, 2 * r), "conv"),
                (f"seanet.up.{j}.convtr.bias", (c // 2,), "small")]
        c //= 2
        h = c // s.compress
        out += [(f"seanet.up.{j}.res.conv1.weight", (h, c, s.res_kernel), "conv"),
                (f"seanet.up.{j}.res.conv1.bias", (h,), "small"),
                (f"seanet.up.{j}.res.conv2.weight", (c, h, 1), "conv"),
                (f"seanet.up.{j}.res.conv2.bias", (c,), "small")]
    out += [("seanet.conv_out.weight", (1, c, s.last_kernel), "conv"), ("seanet.conv_out.bias", (1,), "small")]
    return out


def encoder_weight_names(s: MimiShape) -> List[Tuple[str, Tuple[int, ...], str]]:
    """ENCODE side (voice prompts: sesameai/generator.py:86): SEANet encoder, encoder transformer,
    stride-2 downsample, RVQ input projections.  Kept in a separate list (and a separate RNG
    stream) so the committed decode goldens stay valid."""
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


def make_weights(s: MimiShape, seed: int = 4321, encoder: bool = False) -> Dict[str, torch.Tensor]:
    """Seeded synthetic fp32 weights: codebooks N(0,1), convs/linears Kaiming-uniform-like
    (U[-sqrt(3/fan_in), +]) so activations keep O(1) scale through the stack.  ``encoder=True``
    adds the encode-side tensors from a second generator (seed + 1)."""
    w: Dict[str, torch.Tensor] = {}
    _fill(w, weight_names(s), torch.Generator(device="cpu").manual_seed(seed))
    if encoder:
        _fill(w, encoder_weight_names(s), torch.Generator(device="cpu").manual_seed(seed + 1))
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
        elif kind == "scale":   # LayerScale: real checkpoints are O(0.01..1); keep branches alive
            t = 0.3 + 0.1 * torch.rand(shp, generator=g)
        else:
            if name.endswith("convtr.weight"):      # [in, out/groups, k], k = 2*stride everywhere:
                in_per_group = 1 if name.startswith("upsample") else shp[0]
                fan_in = 2 * in_per_group           # each output sample sees in/groups * k/stride taps
            elif kind == "conv":
                fan_in = shp[1] * shp[2]
            else:
                fan_in = shp[1]
            bound = math.sqrt(3.0 / fan_in)
            t = (torch.rand(shp, generator=g) * 2 - 1) * bound
        w[name] = t.float()


# ----------------------------------------------------------------------------------------
def causal_conv1d(x: torch.Tensor, w: torch.Tensor, b, dilation: int = 1) -> torch.Tensor:
    """stride-1 causal conv: left-pad (k-1)*dilation zeros (StreamingConv1d, pad_mode constant)."""
    k = w.shape[-1]
    return F.conv1d(F.pad(x, ((k - 1) * dilation, 0)), w, b, dilation=dilation)


def causal_convtr1d(x: torch.Tensor, w: torch.Tensor, b, stride: int, groups: int = 1) -> torch.Tensor:
    """causal transposed conv: full output then drop the last k - stride samples."""
    k = w.shape[-1]
    y = F.conv_transpose1d(x, w, b, stride=stride, groups=groups)
    return y[..., : y.shape[-1] - (k - stride)]


def rvq_decode(s: MimiShape, w: Dict[str, torch.Tensor], codes: torch.Tensor) -> torch.Tensor:
    """codes (B,K,T) int -> (B,hidden,T).  Codes >= codebook_size raise, as F.embedding does."""
    def emb(k):
        return w[f"rvq.{k}.embedding_sum"] / w[f"rvq.{k}.cluster_usage"].clamp(min=1e-5)[:, None]
    first = sum(F.embedding(codes[:, k].long(), emb(k)) for k in range(s.num_semantic))
    rest = sum(F.embedding(codes[:, k].long(), emb(k)) for k in range(s.num_semantic, codes.shape[1]))
    q = F.conv1d(first.transpose(1, 2), w["rvq_first.output_proj.weight"])
    if codes.shape[1] > s.num_semantic:
        q = q + F.conv1d(rest.transpose(1, 2), w["rvq_rest.output_proj.weight"])
    return q


def _rope_interleaved(x: torch.Tensor, pos: torch.Tensor, theta: float) -> torch.Tensor:
    """x (B,H,T,hd); rotate pairs (2i,2i+1) by pos * theta^(-2i/hd)."""
    hd = x.shape[-1]
    freqs = torch.exp(torch.arange(hd // 2, dtype=torch.float32) * (-math.log(theta) * 2 / hd))
    ang = pos.float()[:, None] * freqs[None, :]                    # (T, hd/2)
    c, sn = torch.cos(ang), torch.sin(ang)
    xr, xi = x[..., 0::2], x[..., 1::2]
    out = torch.stack([xr * c - xi * sn, xr * sn + xi * c], dim=-1)
    return out.flatten(-2)


def transformer(s: MimiShape, w: Dict[str, torch.Tensor], x: torch.Tensor, offset: int = 0,
                prefix: str = "transformer") -> torch.Tensor:
    """x (B,T,d) -> (B,T,d); causal with context window (key j visible to query i iff
    0 <= i-j < context)."""
    B, T, d = x.shape
    H = s.tr_heads
    hd = d // H
    pos = torch.arange(offset, offset + T)
    delta = pos[:, None] - pos[None, :]
    allowed = (delta >= 0) & (delta < s.tr_context)
    for i in range(s.tr_layers):
        L = f"{prefix}.{i}"
        h = F.layer_norm(x, (d,), w[f"{L}.norm1.weight"], w[f"{L}.norm1.bias"], s.norm_eps)
        qkv = F.linear(h, w[f"{L}.in_proj_weight"]).view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = _rope_interleaved(q, pos, s.rope_theta)
        k = _rope_interleaved(k, pos, s.rope_theta)
        att = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
        att = att.masked_fill(~allowed, float("-inf")).softmax(-1)
        a = (att @ v).transpose(1, 2).reshape(B, T, d)
        x = x + w[f"{L}.layer_scale_1.scale"] * F.linear(a, w[f"{L}.out_proj.weight"])
        h = F.layer_norm(x, (d,), w[f"{L}.norm2.weight"], w[f"{L}.norm2.bias"], s.norm_eps)
        h = F.linear(F.gelu(F.linear(h, w[f"{L}.linear1.weight"])), w[f"{L}.linear2.weight"])
        x = x + w[f"{L}.layer_scale_2.scale"] * h
    return x


def seanet_decode(s: MimiShape, w: Dict[str, torch.Tensor], x: torch.Tensor) -> torch.Tensor:
    x = causal_conv1d(x, w["seanet.conv_in.weight"], w["seanet.conv_in.bias"])
    for j, r in enumerate(s.ratios):
        x = causal_convtr1d(F.elu(x), w[f"seanet.up.{j}.convtr.weight"], w[f"seanet.up.{j}.convtr.bias"], r)
        y = causal_conv1d(F.elu(x), w[f"seanet.up.{j}.res.conv1.weight"], w[f"seanet.up.{j}.res.conv1.bias"])
        y = causal_conv1d(F.elu(y), w[f"seanet.up.{j}.res.conv2.weight"], w[f"seanet.up.{j}.res.conv2.bias"])
        x = x + y
    return causal_conv1d(F.elu(x), w["seanet.conv_out.weight"], w["seanet.conv_out.bias"])


@torch.inference_mode()
def decode(s: MimiShape, w: Dict[str, torch.Tensor], codes: torch.Tensor) -> torch.Tensor:
    """codes (B,K,T) -> pcm (B,1,hop*T) fp32."""
    x = rvq_decode(s, w, codes)
    x = causal_convtr1d(x, w["upsample.convtr.weight"], None, 2, groups=s.hidden)
    x = transformer(s, w, x.transpose(1, 2)).transpose(1, 2)
    return seanet_decode(s, w, x)


@torch.inference_mode()
def decode_stateless_chunks(s: MimiShape, w: Dict[str, torch.Tensor], codes: torch.Tensor,
                            chunk: int = 10) -> torch.Tensor:
    """generate_stream semantics (sesameai/generator.py:111-117,186-203): every ``chunk``
    frames are decoded independently (all streaming state dropped at the seams)."""
    T = codes.shape[-1]
    return torch.cat([decode(s, w, codes[..., t:t + chunk]) for t in range(0, T, chunk)], dim=-1)


# ----------------------------------------------------------------------------------------
# ENCODE (moshi MimiModel.encode as called at sesameai/generator.py:86; structure checked against
# transformers.models.mimi in tests/test_oracle_vs_hf.py)
# ----------------------------------------------------------------------------------------
def _strided_causal_conv(x: torch.Tensor, w: torch.Tensor, b, stride: int, pad_mode: str = "constant") -> torch.Tensor:
    """StreamingConv1d with stride: left pad k - stride, right pad so that ceil(len/stride) frames come out."""
    k = w.shape[-1]
    n = x.shape[-1]
    pad_total = k - stride
    n_frames = math.ceil((n - k + pad_total) / stride + 1) - 1
    extra = max(0, n_frames * stride + k - pad_total - n)
    x = F.pad(x, (pad_total, extra), mode=pad_mode)
    return F.conv1d(x, w, b, stride=stride)


@torch.inference_mode()
def encode_latent(s: MimiShape, w: Dict[str, torch.Tensor], wav: torch.Tensor) -> torch.Tensor:
    """wav (B,1,n) -> pre-quantisation embeddings (B, hidden, T), T = ceil(n / hop)."""
    x = causal_conv1d(wav, w["enc.conv_in.weight"], w["enc.conv_in.bias"])
    for j, r in enumerate(reversed(s.ratios)):
        y = causal_conv1d(F.elu(x), w[f"enc.down.{j}.res.conv1.weight"], w[f"enc.down.{j}.res.conv1.bias"])
        y = causal_conv1d(F.elu(y), w[f"enc.down.{j}.res.conv2.weight"], w[f"enc.down.{j}.res.conv2.bias"])
        x = x + y
        x = _strided_causal_conv(F.elu(x), w[f"enc.down.{j}.conv.weight"], w[f"enc.down.{j}.conv.bias"], r)
    x = causal_conv1d(F.elu(x), w["enc.conv_out.weight"], w["enc.conv_out.bias"])
    x = transformer(s, w, x.transpose(1, 2), prefix="enc_transformer").transpose(1, 2)
    return _strided_causal_conv(x, w["downsample.conv.weight"], None, 2, pad_mode="replicate")


def _rvq_encode(res: torch.Tensor, books: List[torch.Tensor]) -> List[torch.Tensor]:
    """res (B,T,D); nearest-centroid residual quantisation (torch.cdist + argmin, first index on ties)."""
    out = []
    for e in books:
        idx = torch.cdist(res.reshape(1, -1, res.shape[-1]), e[None], p=2)[0].argmin(dim=-1).view(res.shape[:-1])
        out.append(idx)
        res = res - F.embedding(idx, e)
    return out


@torch.inference_mode()
def encode(s: MimiShape, w: Dict[str, torch.Tensor], wav: torch.Tensor) -> torch.Tensor:
    """wav (B,1,n) fp32 @ 24 kHz -> codes (B, num_codebooks, ceil(n/1920)) int64.  Split RVQ: the
    semantic codebook and the acoustic stack both quantise the SAME embeddings, each behind its own
    1x1 input projection."""
    z = encode_latent(s, w, wav)
    def emb(k):
        return w[f"rvq.{k}.embedding_sum"] / w[f"rvq.{k}.cluster_usage"].clamp(min=1e-5)[:, None]
    first = F.conv1d(z, w["rvq_first.input_proj.weight"]).transpose(1, 2)
    rest = F.conv1d(z, w["rvq_rest.input_proj.weight"]).transpose(1, 2)
    codes = _rvq_encode(first, [emb(k) for k in range(s.num_semantic)])
    codes += _rvq_encode(rest, [emb(k) for k in range(s.num_semantic, s.num_codebooks)])
    return torch.stack(codes, dim=1)
