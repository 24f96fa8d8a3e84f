"""MI355X-native ``Model`` with the reference's surface (reference: sesameai/models.py).

Same names and call semantics as the reference's ``Model`` -- ``setup_caches``,
``reset_caches``, ``generate_frame(tokens, tokens_mask, input_pos, temperature, topk)`` ->
``(B, 32) int32`` -- but every op runs in hand-written gfx950 kernels behind the C ABI of
libcsm_hip.so (include/csm_hip.h).  PyTorch is used only for device memory and streams.
There is no CPU / eager fallback: without the HIP library or a GPU this module raises.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _abi
from ._abi import lib, check


# ----------------------------------------------------------------------------------------
# shapes (reference: sesameai/models.py:10-45 llama3_2_1B / llama3_2_100M / FLAVORS)
# ----------------------------------------------------------------------------------------
@dataclass(frozen=True)
class LlamaFlavor:
    num_layers: int
    num_heads: int
    num_kv_heads: int
    embed_dim: int
    intermediate_dim: int
    max_seq_len: int = 2048
    norm_eps: float = 1e-5
    rope_base: float = 500_000.0
    scale_factor: float = 32.0

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


FLAVORS: Dict[str, LlamaFlavor] = {
    "llama-1B": LlamaFlavor(16, 32, 8, 2048, 8192),
    "llama-100M": LlamaFlavor(4, 8, 2, 1024, 8192),
    # small stand-ins with the same head dims (64 / 128), used by the test-suite
    "llama-tiny-bb": LlamaFlavor(2, 8, 2, 512, 1024, max_seq_len=256),
    "llama-tiny-dec": LlamaFlavor(2, 4, 2, 512, 1024, max_seq_len=256),
    # the tiny backbone with the real 2048-position cache: long-context paths (split-K decode attention over ~1700 keys,
    # 1334-row prompts) against the live oracle in seconds
    "llama-tiny-bb-2k": LlamaFlavor(2, 8, 2, 512, 1024, max_seq_len=2048),
}


@dataclass
class ModelArgs:
    backbone_flavor: str
    decoder_flavor: str
    text_vocab_size: int
    audio_vocab_size: int
    audio_num_codebooks: int


def csm_1b_args() -> ModelArgs:
    return ModelArgs("llama-1B", "llama-100M", 128_256, 2051, 32)


def csm_tiny_args() -> ModelArgs:
    return ModelArgs("llama-tiny-bb", "llama-tiny-dec", 1000, 2051, 32)


def csm_tiny_2k_args() -> ModelArgs:
    return ModelArgs("llama-tiny-bb-2k", "llama-tiny-dec", 1000, 2051, 32)


# ----------------------------------------------------------------------------------------
# host-side tables and weights
# ----------------------------------------------------------------------------------------
def llama3_rope_theta(f: LlamaFlavor) -> torch.Tensor:
    """Llama3ScaledRoPE's per-pair frequencies (fp32): base^(-2i/hd), long wavelengths divided by the scale factor."""
    hd = f.head_dim
    theta = 1.0 / (f.rope_base ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd))
    old_len, lo, hi = 8192, 1.0, 4.0
    scaled = []
    for fr in theta.tolist():
        wl = 2 * math.pi / fr
        if wl < old_len / hi:
            scaled.append(fr)
        elif wl > old_len / lo:
            scaled.append(fr / f.scale_factor)
        else:
            smooth = (old_len / wl - lo) / (hi - lo)
            scaled.append((1 - smooth) * fr / f.scale_factor + smooth * fr)
    return torch.tensor(scaled, dtype=theta.dtype)


def llama3_rope_table(f: LlamaFlavor) -> torch.Tensor:
    """[max_seq][hd/2][2] (cos, sin) of torchtune's Llama3ScaledRoPE (low/high freq factors 1/4,
    old context 8192), built in fp32 on the host and rounded to bf16 like the reference's
    ``model.to(dtype=bf16)`` rounds the registered buffer (sesameai/generator.py:343)."""
    theta = llama3_rope_theta(f)
    idx = torch.einsum("i,j->ij", torch.arange(f.max_seq_len, dtype=theta.dtype), theta).float()
    return torch.stack([torch.cos(idx), torch.sin(idx)], dim=-1).to(torch.bfloat16)


def state_dict_layout(args: ModelArgs) -> List[Tuple[str, Tuple[int, ...]]]:
    """Tensor names/shapes of a CSM checkpoint (module tree of sesameai/models.py:110-118 with
    torchtune attribute names), in a fixed order."""
    bb, dec = FLAVORS[args.backbone_flavor], FLAVORS[args.decoder_flavor]
    out: List[Tuple[str, Tuple[int, ...]]] = [
        ("text_embeddings.weight", (args.text_vocab_size, bb.embed_dim)),
        ("audio_embeddings.weight", (args.audio_vocab_size * args.audio_num_codebooks, bb.embed_dim)),
    ]
    for pfx, s in (("backbone", bb), ("decoder", dec)):
        hd = s.head_dim
        for i in range(s.num_layers):
            L = f"{pfx}.layers.{i}"
            out += [(f"{L}.attn.q_proj.weight", (s.num_heads * hd, s.embed_dim)),
                    (f"{L}.attn.k_proj.weight", (s.num_kv_heads * hd, s.embed_dim)),
                    (f"{L}.attn.v_proj.weight", (s.num_kv_heads * hd, s.embed_dim)),
                    (f"{L}.attn.output_proj.weight", (s.embed_dim, s.num_heads * hd)),
                    (f"{L}.mlp.w1.weight", (s.intermediate_dim, s.embed_dim)),
                    (f"{L}.mlp.w2.weight", (s.embed_dim, s.intermediate_dim)),
                    (f"{L}.mlp.w3.weight", (s.intermediate_dim, s.embed_dim)),
                    (f"{L}.sa_norm.scale", (s.embed_dim,)),
                    (f"{L}.mlp_norm.scale", (s.embed_dim,))]
        out.append((f"{pfx}.norm.scale", (s.embed_dim,)))
    out += [("projection.weight", (dec.embed_dim, bb.embed_dim)),
            ("codebook0_head.weight", (args.audio_vocab_size, bb.embed_dim)),
            ("audio_head", (args.audio_num_codebooks - 1, dec.embed_dim, args.audio_vocab_size))]
    return out


def synthetic_state_dict(args: ModelArgs, seed: int = 1234, std: float = 0.02, flavour: str = "bench") -> Dict[str, torch.Tensor]:
    """Random-init weights of the true shapes (no checkpoint can be downloaded here):
    N(0, std^2) fp32 -> bf16 from one seeded CPU generator in ``state_dict_layout`` order,
    norm scales = 1.  ``flavour="decisive"``: the same draws re-arranged by
    ``_decisive_checkpoint`` into a checkpoint whose greedy decisions are far from ties."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shp in state_dict_layout(args):
        if name.endswith(".scale"):
            t = torch.ones(shp, dtype=torch.float32)
        else:
            t = torch.empty(shp, dtype=torch.float32).normal_(0.0, std, generator=g)
        sd[name] = t.to(torch.bfloat16)
    if flavour == "decisive":
        return _decisive_checkpoint(args, sd, seed)
    if flavour.split(":")[0] == "decisive_copy":
        f = flavour.split(":")
        if len(f) not in (1, 3):
            raise ValueError(f"decisive_copy flavour is 'decisive_copy' or 'decisive_copy:<layer>:<lag>', got {flavour!r}")
        n_layers = FLAVORS[args.backbone_flavor].num_layers
        return _decisive_copy_checkpoint(args, sd, seed, int(f[1]) if len(f) == 3 else n_layers // 2, int(f[2]) if len(f) == 3 else 3)
    if flavour != "bench":
        raise ValueError(f"unknown synthetic flavour {flavour!r}")
    return sd


def _decisive_checkpoint(args: ModelArgs, sd: Dict[str, torch.Tensor], seed: int, last_gain: float = 4.0) -> Dict[str, torch.Tensor]:
    """A synthetic checkpoint that talks: random weights give near-uniform logits, so greedy decoding is a sequence of near-ties and
    two correct bf16 implementations drift apart within a frame.  Here each head reads back one embedding: codebook0_head's rows are
    the LAST codebook's embedding rows in a seeded order (the backbone's input is the sum of a frame's 32 embeddings, reference
    sesameai/models.py:156-157, so frame t's last code names frame t+1's first), audio_head[i-1]'s columns are
    projection @ (codebook i-1's embedding rows) in a seeded order (the decoder's input at step i is the projected embedding of code i-1,
    models.py:170-181), text rows carry one last-codebook entry so the prompt frame decides as clearly, embeddings are 8x (the last
    codebook 32x) and the o- / down-projections 0.5x so a row's residual stream keeps its embedding in front of what the layers add.
    Every op still runs on every shape; every greedy decision is one logit several units above the rest, so free-running greedy codes
    are a checkable statement (tests/test_decisive_gpu.py; same construction, written independently, in oracle/csm_ref.py)."""
    bf = torch.bfloat16
    V, ncb = args.audio_vocab_size, args.audio_num_codebooks
    live = min(2048, V)                                     # Mimi's codebook size: CSM's extra logit rows stay random and small
    pg = torch.Generator(device="cpu").manual_seed(seed * 1_000_003 + 17)
    order = [torch.randperm(live, generator=pg) for _ in range(ncb)]
    out = dict(sd)
    for name in sd:
        if name.endswith(("attn.output_proj.weight", "mlp.w2.weight")):
            out[name] = (sd[name].float() * 0.5).to(bf)
    audio = (sd["audio_embeddings.weight"].float() * 8.0).to(bf)
    lo = (ncb - 1) * V
    audio[lo:lo + live] = (audio[lo:lo + live].float() * last_gain).to(bf)
    out["audio_embeddings.weight"] = audio
    named = audio[lo:lo + live]
    which = (torch.arange(args.text_vocab_size) * 40503) % live
    out["text_embeddings.weight"] = (sd["text_embeddings.weight"].float() * 8.0 + named[which].float()).to(bf)
    c0 = sd["codebook0_head.weight"].clone()
    c0[:live] = (named[order[0]].float() / (8.0 * last_gain)).to(bf)
    out["codebook0_head.weight"] = c0
    out["audio_head"] = _chained_audio_heads(args, sd["audio_head"], audio, sd["projection.weight"], order)
    return out


def _chained_audio_heads(args: ModelArgs, heads: torch.Tensor, audio: torch.Tensor, projection: torch.Tensor, order) -> torch.Tensor:
    """audio_head[i-1][:, v] = projection @ (codebook i-1's embedding row order[i][v]) / 8: step i of the depth decoder names the code that
    step i-1 fed it (models.py:170-181)."""
    V, ncb = args.audio_vocab_size, args.audio_num_codebooks
    live = min(2048, V)
    proj_t = projection.double().t()                            # fp64 products, rounded once: the same bits on every host
    heads = heads.clone()
    for i in range(1, ncb):
        src = audio[(i - 1) * V:(i - 1) * V + live][order[i]].double()
        heads[i - 1, :, :live] = ((src @ proj_t).t() / 8.0).float().to(torch.bfloat16)
    return heads


def _decisive_copy_checkpoint(args: ModelArgs, sd: Dict[str, torch.Tensor], seed: int, layer: int, lag: int) -> Dict[str, torch.Tensor]:
    """The decisive checkpoint with the frame-to-frame decision moved INTO one backbone layer's attention, so that the free-running
    greedy codes depend on what the KV cache holds (the plain decisive checkpoint is memoryless: frame t+1 follows from frame t's last
    code through one row's residual stream, whatever attention does).  Every backbone row carries a large tag on 16 coordinates (added
    to codebook 0's embedding rows and to the text rows: a row holds exactly one of the two); layer ``layer``'s q / k read nothing but
    the tag, into the 8 fastest rotary pairs, the key side turned by ``lag`` positions, so the softmax is one-hot on the row ``lag``
    back; its v / o projections carry a random image of THAT row into the current one, tall enough to lead the residual stream, and
    codebook0_head's rows are the images of the last codebook's embedding rows: c0(t+1) names the last code of the row ``lag`` back, read
    through the RoPE'd cached K and the cached V (reference path: sesameai/models.py:154-158 inside the loop of generator.py:283-294).
    The projection loses the tag direction (the depth decoder never sees it) and the decoder's chain of heads is the decisive one.
    Same construction as oracle/csm_ref.py decisive_copy_weights, written on its own; tests assert tensor equality."""
    bf = torch.bfloat16
    bb = FLAVORS[args.backbone_flavor]
    V, ncb, d, hd = args.audio_vocab_size, args.audio_num_codebooks, bb.embed_dim, bb.head_dim
    nh, nkv = bb.num_heads, bb.num_kv_heads
    if not (0 <= layer < bb.num_layers and 1 <= lag < bb.max_seq_len):
        raise ValueError(f"decisive_copy: layer {layer} / lag {lag} outside the backbone's {bb.num_layers} layers / {bb.max_seq_len} positions")
    TAG, PAIRS, SHARP, LAST, LOGIT = 256.0, 8, 16.0, 8.0, 4.0
    live = min(2048, V)
    out = _decisive_checkpoint(args, sd, seed, last_gain=LAST)
    pg = torch.Generator(device="cpu").manual_seed(seed * 1_000_003 + 17)
    order = [torch.randperm(live, generator=pg) for _ in range(ncb)]
    tg = torch.Generator(device="cpu").manual_seed(seed * 1_000_003 + 29)
    tag = torch.zeros(d, dtype=torch.float64)
    tag[torch.arange(16) * (d // 16) + 3] = (torch.randint(0, 2, (16,), generator=tg).double() * 2 - 1) / 4.0       # unit vector
    strip = lambda m: m - (m @ tag)[:, None] * tag[None, :]                    # rows made orthogonal to the tag
    out["projection.weight"] = strip(sd["projection.weight"].double()).float().to(bf)
    out["audio_head"] = _chained_audio_heads(args, sd["audio_head"], out["audio_embeddings.weight"], out["projection.weight"], order)
    audio = out["audio_embeddings.weight"].clone()
    audio[:V] = (audio[:V].double() + TAG * tag).float().to(bf)
    out["audio_embeddings.weight"] = audio
    out["text_embeddings.weight"] = (out["text_embeddings.weight"].double() + TAG * tag).float().to(bf)
    # tag . rmsnorm(row): the row's mean square is the tag's plus 31 embeddings at 8 x 0.02 and one at 8 x that
    row_rms = math.sqrt(TAG * TAG / d + (8.0 * 0.02) ** 2 * ((ncb - 1) + LAST * LAST))
    seen = TAG / row_rms
    theta = llama3_rope_theta(bb).double()[:PAIRS]
    drop = float((1.0 - torch.cos(theta)).sum())                              # what a neighbour of the target row loses, per unit of q.k
    amp = math.sqrt(SHARP * math.sqrt(hd) / (drop * seen * seen))
    Lp = f"backbone.layers.{layer}.attn."
    q = torch.zeros(nh, hd, d, dtype=torch.float64)
    k = torch.zeros(nkv, hd, d, dtype=torch.float64)
    for j in range(PAIRS):
        q[:, 2 * j] = amp * tag
        k[:, 2 * j] = amp * math.cos(lag * float(theta[j])) * tag
        k[:, 2 * j + 1] = amp * math.sin(lag * float(theta[j])) * tag
    out[Lp + "q_proj.weight"] = q.reshape(nh * hd, d).float().to(bf)
    out[Lp + "k_proj.weight"] = k.reshape(nkv * hd, d).float().to(bf)
    out[Lp + "v_proj.weight"] = strip(sd[Lp + "v_proj.weight"].double()).float().to(bf)
    wv, wo = out[Lp + "v_proj.weight"].double(), sd[Lp + "output_proj.weight"].double()
    rows = out["audio_embeddings.weight"][(ncb - 1) * V:(ncb - 1) * V + live][order[0]].double()
    image = (rows @ wv.t()).view(live, nkv, 1, hd).expand(live, nkv, nh // nkv, hd).reshape(live, nh * hd) @ wo.t()
    size = image.norm(dim=1, keepdim=True)
    gain = 2.0 ** round(math.log2(TAG * row_rms / float(size.mean())))        # the image stands as tall as the tag; a power of two
    out[Lp + "output_proj.weight"] = (wo * gain).float().to(bf)
    c0 = sd["codebook0_head.weight"].clone()
    c0[:live] = (image / size * (LOGIT * math.sqrt(5.0) / math.sqrt(d))).float().to(bf)
    out["codebook0_head.weight"] = c0
    return out


MATRIX_SUFFIXES = ("q_proj.weight", "k_proj.weight", "v_proj.weight", "output_proj.weight", "w1.weight", "w2.weight", "w3.weight")


def quantize_fp8_rows(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """[N][K] -> (e4m3 bytes [N][K] uint8, scale [N] fp32, dequantised bf16 [N][K]).  OCP e4m3fn with one
    POWER-OF-TWO scale per output row, 2^ceil(log2(max|row| / 448)): e4m3 has 3 mantissa bits, so
    byte * scale is exactly representable in bf16 and the fp8 decode stream computes the same dot
    products as a bf16 model holding the dequantised weights."""
    wf = w.float()
    mx = wf.abs().amax(dim=1).clamp(min=1e-30)
    scale = torch.exp2(torch.ceil(torch.log2(mx / 448.0)))
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    deq = (q.float() * scale[:, None]).to(torch.bfloat16)
    return q.view(torch.uint8), scale.float(), deq


def fp8_weight_set(args: ModelArgs, sd: Dict[str, torch.Tensor]):
    """Quantises every projection matrix of both stacks + the heads.  Returns (state dict with the
    matrices replaced by their dequantised bf16 values, {name: (bytes, scales)})."""
    out, q8 = dict(sd), {}
    for name in list(sd):
        if name.endswith(MATRIX_SUFFIXES) and (name.startswith("backbone.layers") or name.startswith("decoder.layers")):
            q, s, deq = quantize_fp8_rows(sd[name])
            out[name] = deq; q8[name] = (q, s)
    q, s, deq = quantize_fp8_rows(sd["codebook0_head.weight"])
    out["codebook0_head.weight"] = deq; q8["codebook0_head.weight"] = (q, s)
    ah = sd["audio_head"]                                       # [31][d][V]: quantise per logit row = per (i, v)
    aht = ah.transpose(1, 2).contiguous()                       # [31][V][d]
    q, s, deq = quantize_fp8_rows(aht.reshape(-1, aht.shape[-1]))
    out["audio_head"] = deq.view_as(aht).transpose(1, 2).contiguous()
    q8["audio_head_t"] = (q.view(aht.shape), s.view(aht.shape[0], aht.shape[1]))
    return out, q8


def from_hf_state_dict(args: ModelArgs, hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Converts a ``transformers`` ``CsmForConditionalGeneration`` checkpoint (the HF-format
    ``sesame/csm-1b`` repo) to the reference's torchtune layout.  The HF port uses half-split RoPE,
    so each head's q/k rows were permuted by ``cat(arange(0,hd,2), arange(1,hd,2))`` at conversion;
    this applies the inverse permutation."""
    def unperm(w: torch.Tensor, n_heads: int, hd: int) -> torch.Tensor:
        perm = torch.cat([torch.arange(0, hd, 2), torch.arange(1, hd, 2)])
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(hd)
        return w.view(n_heads, hd, -1)[:, inv, :].reshape(n_heads * hd, -1)

    out: Dict[str, torch.Tensor] = {
        "text_embeddings.weight": hf["embed_text_tokens.weight"],
        "audio_embeddings.weight": hf["backbone_model.embed_tokens.embed_audio_tokens.weight"],
        "projection.weight": hf["depth_decoder.model.inputs_embeds_projector.weight"],
        "codebook0_head.weight": hf["lm_head.weight"],
        "audio_head": hf["depth_decoder.codebooks_head.weight"],
        "backbone.norm.scale": hf["backbone_model.norm.weight"],
        "decoder.norm.scale": hf["depth_decoder.model.norm.weight"],
    }
    for pfx, hpfx, f in (("backbone", "backbone_model", FLAVORS[args.backbone_flavor]),
                         ("decoder", "depth_decoder.model", FLAVORS[args.decoder_flavor])):
        for i in range(f.num_layers):
            L, Hh = f"{pfx}.layers.{i}", f"{hpfx}.layers.{i}"
            out[f"{L}.attn.q_proj.weight"] = unperm(hf[f"{Hh}.self_attn.q_proj.weight"], f.num_heads, f.head_dim)
            out[f"{L}.attn.k_proj.weight"] = unperm(hf[f"{Hh}.self_attn.k_proj.weight"], f.num_kv_heads, f.head_dim)
            out[f"{L}.attn.v_proj.weight"] = hf[f"{Hh}.self_attn.v_proj.weight"]
            out[f"{L}.attn.output_proj.weight"] = hf[f"{Hh}.self_attn.o_proj.weight"]
            out[f"{L}.mlp.w1.weight"] = hf[f"{Hh}.mlp.gate_proj.weight"]
            out[f"{L}.mlp.w3.weight"] = hf[f"{Hh}.mlp.up_proj.weight"]
            out[f"{L}.mlp.w2.weight"] = hf[f"{Hh}.mlp.down_proj.weight"]
            out[f"{L}.sa_norm.scale"] = hf[f"{Hh}.input_layernorm.weight"]
            out[f"{L}.mlp_norm.scale"] = hf[f"{Hh}.post_attention_layernorm.weight"]
    return out


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class Model:
    """Drop-in for the reference ``Model`` (sesameai/models.py:99-203)."""

    def __init__(self, config: ModelArgs, state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 device: str = "cuda", max_frames: int = 2048, max_prefill_rows: int = 2048,
                 weights_dtype: str = "bf16"):
        if not torch.cuda.is_available():
            raise RuntimeError("sesameai (MI355X build) needs a ROCm GPU: there is no CPU fallback")
        self.config = config
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:      # a concrete index: tensors report theirs ("cuda:0")
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.bb, self.dec = FLAVORS[config.backbone_flavor], FLAVORS[config.decoder_flavor]
        if state_dict is None:
            state_dict = synthetic_state_dict(config)
        if weights_dtype not in ("bf16", "fp8"):
            raise ValueError("weights_dtype must be 'bf16' or 'fp8'")
        self.weights_dtype = weights_dtype
        self._q8: Dict[str, torch.Tensor] = {}
        if weights_dtype == "fp8":     # decode step streams e4m3 bytes; prefill / batch keep the (dequantised) bf16
            state_dict, q8 = fp8_weight_set(config, state_dict)
            for name, (q, sc) in q8.items():
                self._q8[name] = q.to(self.device).contiguous()
                self._q8[name + ".scale"] = sc.to(self.device).contiguous()
        layout = dict(state_dict_layout(config))
        missing = [k for k in layout if k not in state_dict]
        if missing:
            raise KeyError(f"checkpoint is missing tensors: {missing[:4]}...")
        self._w: Dict[str, torch.Tensor] = {}
        for name, shp in layout.items():
            t = state_dict[name]
            if tuple(t.shape) != tuple(shp):
                raise ValueError(f"{name}: expected shape {shp}, got {tuple(t.shape)}")
            self._w[name] = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
        # device-native layouts: K-major audio_head -> one contiguous row per logit
        self._w["audio_head_t"] = self._w["audio_head"].transpose(1, 2).contiguous()
        self._w["bb_rope"] = llama3_rope_table(self.bb).to(self.device).contiguous()
        self._w["dec_rope"] = llama3_rope_table(self.dec).to(self.device).contiguous()
        self._max_frames = max_frames
        self._max_prefill_rows = max_prefill_rows
        self._h = C.c_void_p(None)
        self._max_batch = 0
        self._seeded = False
        # prefix-KV reuse: the reference re-runs the whole 900-1550-row voice prompt
        # for every sentence (tts_service.py:191-207); here the backbone KV of the previous prompt is
        # kept and only the rows after the longest common prefix are prefilled again.
        self.prefix_reuse = True
        self._kv_prompt: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.last_prefill_rows = 0

    # -- reference-compatible construction helpers ------------------------------------------
    @classmethod
    def from_pretrained(cls, path: str, device: str = "cuda", **kw) -> "Model":
        """Loads ``model.safetensors`` of a local ``sesame/csm-1b`` snapshot directory (the
        reference pulls it from the hub with PyTorchModelHubMixin, sesameai/generator.py:338;
        there is no network here)."""
        import os
        from safetensors.torch import load_file
        f = path if path.endswith(".safetensors") else os.path.join(path, "model.safetensors")
        sd = load_file(f)
        if "backbone_model.norm.weight" in sd:          # transformers-format checkpoint
            sd = from_hf_state_dict(csm_1b_args(), sd)
        return cls(csm_1b_args(), sd, device=device, **kw)

    def parameters(self):
        return iter(self._w.values())

    def weight_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._w.values())

    # -- caches ---------------------------------------------------------------------------------
    def _cfg_struct(self) -> _abi.CsmConfig:
        def dims(f: LlamaFlavor) -> _abi.CsmLlamaDims:
            return _abi.CsmLlamaDims(f.num_layers, f.num_heads, f.num_kv_heads, f.embed_dim,
                                     f.intermediate_dim, f.max_seq_len, f.norm_eps)
        return _abi.CsmConfig(dims(self.bb), dims(self.dec), self.config.text_vocab_size,
                              self.config.audio_vocab_size, self.config.audio_num_codebooks)

    def _weights_struct(self) -> _abi.CsmWeights:
        w = _abi.CsmWeights()
        p = lambda n: self._w[n].data_ptr()
        w.text_emb, w.audio_emb = p("text_embeddings.weight"), p("audio_embeddings.weight")
        for pfx, arr, f in (("backbone", w.bb, self.bb), ("decoder", w.dec, self.dec)):
            for i in range(f.num_layers):
                L = f"{pfx}.layers.{i}"
                arr[i] = _abi.CsmLayerWeights(
                    p(f"{L}.attn.q_proj.weight"), p(f"{L}.attn.k_proj.weight"), p(f"{L}.attn.v_proj.weight"),
                    p(f"{L}.attn.output_proj.weight"), p(f"{L}.mlp.w1.weight"), p(f"{L}.mlp.w2.weight"),
                    p(f"{L}.mlp.w3.weight"), p(f"{L}.sa_norm.scale"), p(f"{L}.mlp_norm.scale"))
        w.bb_norm, w.dec_norm = p("backbone.norm.scale"), p("decoder.norm.scale")
        w.projection, w.c0_head = p("projection.weight"), p("codebook0_head.weight")
        w.audio_head_t, w.bb_rope, w.dec_rope = p("audio_head_t"), p("bb_rope"), p("dec_rope")
        w.fp8 = int(self.weights_dtype == "fp8")
        if w.fp8:
            q = lambda n: self._q8[n].data_ptr()
            for pfx, arr, arrs, f in (("backbone", w.bb8, w.bb8s, self.bb), ("decoder", w.dec8, w.dec8s, self.dec)):
                for i in range(f.num_layers):
                    L = f"{pfx}.layers.{i}"
                    names = [f"{L}.attn.q_proj.weight", f"{L}.attn.k_proj.weight", f"{L}.attn.v_proj.weight",
                             f"{L}.attn.output_proj.weight", f"{L}.mlp.w1.weight", f"{L}.mlp.w2.weight", f"{L}.mlp.w3.weight"]
                    arr[i] = _abi.CsmLayerWeights(*[q(n) for n in names], None, None)
                    arrs[i] = _abi.CsmLayerWeights(*[q(n + ".scale") for n in names], None, None)
            w.c0_head8, w.c0_head8s = q("codebook0_head.weight"), q("codebook0_head.weight.scale")
            w.audio_head8, w.audio_head8s = q("audio_head_t"), q("audio_head_t.scale")
        return w

    def setup_caches(self, max_batch_size: int) -> None:
        """reference: Model.setup_caches (sesameai/models.py:120-130)."""
        if self._h:
            lib.csm_destroy(self._h)
            self._h = C.c_void_p(None)
        cfg, w = self._cfg_struct(), self._weights_struct()
        rows = max(self._max_prefill_rows, 2 * max_batch_size)
        with torch.cuda.device(self.device):
            check(lib.csm_create(C.byref(cfg), C.byref(w), max_batch_size, rows, self._max_frames, C.byref(self._h)))
        self._max_batch = max_batch_size
        self._kv_prompt = None

    def caches_are_enabled(self) -> bool:
        return bool(self._h)

    def _on_device(self):
        """Every C-ABI call runs with this model's GPU current (kernels, streams and allocations are per device)."""
        return torch.cuda.device(self.device)

    def reset_caches(self) -> None:
        """reference: Model.reset_caches (sesameai/models.py:186-188)."""
        self._require()
        with self._on_device():
            check(lib.csm_reset(self._h, _stream_ptr()), self._h)

    def seed(self, seed: int) -> None:
        self._require()
        with self._on_device():
            check(lib.csm_seed(self._h, seed, _stream_ptr()), self._h)
        self._seeded = True
        self._seed_value = int(seed)

    def _to_dev(self, x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """A prompt tensor on the device in the dtype the C ABI reads.  A host tensor is cast by ONE single-threaded numpy pass straight
        into a pinned staging buffer the model keeps, then copied asynchronously: 4x fewer bytes than the int64 original, no pinning
        of the caller's pages -- and no torch CPU kernel.  The last point is what round 4 measured (tools/dbg/prefill_wall_diag.py):
        a 1,334-row prompt is 44,022 elements, above torch's 32,768-element grain, so ``x.to(torch.int32)`` becomes an OpenMP region
        on every hardware thread torch sees (128 here); in a container with a CPU quota (16 on the GPU boxes of this pool) the
        spinning team burns the quota and the cgroup throttles the process for the rest of the period: every third 6 ms prefill took
        ~85 ms.  (Callers that run their own large torch CPU ops in such a container want OMP_NUM_THREADS <= the quota.)"""
        if x.device.type != "cpu" or self.device.type != "cuda":
            return x.to(device=self.device, dtype=dtype).contiguous()
        import numpy as np
        out = torch.empty(x.shape, dtype=dtype, device=self.device)
        n = out.numel() * out.element_size()
        if n == 0:
            return out
        n_al = (n + 255) // 256 * 256
        st = getattr(self, "_stage", None)
        if st is None or st["buf"].numel() < n_al:
            if st is not None:
                st["done"].synchronize()
            st = self._stage = {"buf": torch.empty(max(n_al, 1 << 20), dtype=torch.uint8, pin_memory=True), "off": 0,
                                "done": torch.cuda.Event()}
        if st["off"] + n_al > st["buf"].numel():
            st["done"].synchronize()                     # the copies issued from the buffer so far have left it
            st["off"] = 0
        piece = st["buf"][st["off"]: st["off"] + n].view(dtype).view(x.shape)
        st["off"] += n_al
        np.copyto(piece.numpy(), x.detach().numpy(), casting="unsafe")
        with self._on_device():
            out.copy_(piece, non_blocking=True)
            st["done"].record()
        return out

    def _check_positions(self, input_pos: torch.Tensor) -> None:
        """Positions must lie in [0, max_seq_len): checked here when that costs no device sync; positions that only
        exist on the device are checked by the kernels (device flag -> CSM_E_TOO_LONG from read_frames)."""
        if input_pos.device.type == "cpu" and input_pos.numel():
            lo, hi = int(input_pos.min()), int(input_pos.max())
            if lo < 0 or hi >= self.bb.max_seq_len:
                raise ValueError(f"input_pos outside [0, {self.bb.max_seq_len}): [{lo}, {hi}]")

    def _require(self) -> None:
        assert self._h, "backbone caches are not enabled"     # reference: models.py:153

    def __del__(self):
        try:
            if self._h:
                lib.csm_destroy(self._h)
        except Exception:
            pass

    # -- the frame ----------------------------------------------------------------------------
    def prefill(self, tokens: torch.Tensor, tokens_mask: torch.Tensor, input_pos: torch.Tensor,
                _keeps_prompt_prefix: bool = False) -> None:
        self._require()
        if not _keeps_prompt_prefix:
            self._kv_prompt = None          # rows written at caller-chosen positions: the cached prefix is unknown
        b, s, _ = tokens.shape
        if b * s > max(self._max_prefill_rows, 2 * self._max_batch):
            raise ValueError(f"prompt of {b}x{s} rows exceeds max_prefill_rows={self._max_prefill_rows}")
        self._check_positions(input_pos)
        t = self._to_dev(tokens, torch.int32)
        m = self._to_dev(tokens_mask, torch.uint8)
        p = self._to_dev(input_pos, torch.int32)
        with self._on_device():
            check(lib.csm_prefill(self._h, t.data_ptr(), m.data_ptr(), p.data_ptr(), b, s, int(_keeps_prompt_prefix), _stream_ptr()), self._h)

    def prefill_prompt(self, tokens: torch.Tensor, tokens_mask: torch.Tensor) -> int:
        """Prefill a prompt that starts at position 0 (B,S,33), reusing the cached backbone KV of the
        longest common row prefix with the previous prompt.  Returns the number of rows actually run.
        The KV entries of the reused rows were produced by the same deterministic kernels from the
        same inputs at the same positions, so the result is bit-identical to a full prefill."""
        b, s, _ = tokens.shape
        if s > self.bb.max_seq_len:
            raise ValueError(f"prompt of {s} rows exceeds max_seq_len={self.bb.max_seq_len}")
        t = self._to_dev(tokens, torch.int32)
        m = self._to_dev(tokens_mask, torch.uint8).to(torch.bool)
        t = torch.where(m, t, torch.zeros_like(t))                    # masked slots do not matter
        start = 0
        if self.prefix_reuse and self._kv_prompt is not None and self._kv_prompt[0].shape[0] == b:
            ot, om = self._kv_prompt
            n = min(ot.shape[1], s)
            same = ((t[:, :n] == ot[:, :n]) & (m[:, :n] == om[:, :n])).all(dim=2).all(dim=0)
            start = int(same.to(torch.int32).cumprod(0).sum().item())
            start = min(start, s - 1)                                 # at least the last row runs (it yields last_h)
        pos = torch.arange(start, s, device=self.device, dtype=torch.int32).unsqueeze(0).repeat(b, 1)
        self.prefill(t[:, start:], m[:, start:], pos, _keeps_prompt_prefix=True)
        self._kv_prompt = (t, m)
        self.last_prefill_rows = s - start
        return s - start

    def reset_slots(self, slots) -> None:
        """Position 0 and "no EOS yet" for the listed batch slots; the other slots are untouched (csm_reset_slots)."""
        self._require()
        arr = (C.c_int32 * len(slots))(*[int(s) for s in slots])
        with self._on_device():
            check(lib.csm_reset_slots(self._h, arr, len(slots), _stream_ptr()), self._h)

    def refill_slot(self, slot: int, tokens: torch.Tensor, tokens_mask: torch.Tensor, temperature: float, topk: int) -> torch.Tensor:
        """A new prompt (S,33) starting at position 0 into batch slot ``slot`` of a live batch: backbone prefill into the slot's
        caches, depth pass, the new utterance's frame 0 staged as the slot's next input.  Returns frame 0 (32,) int32 on the
        device.  The other slots keep generating undisturbed (bit-identical frames); see include/csm_hip.h csm_prefill_slot."""
        self._require()
        s = tokens.shape[0]
        if s >= self.bb.max_seq_len or s > max(self._max_prefill_rows, 2 * self._max_batch):
            raise ValueError(f"prompt of {s} rows exceeds the limits (max_seq_len {self.bb.max_seq_len}, max_prefill_rows {self._max_prefill_rows})")
        self._kv_prompt = None                                   # slot 0's cached prompt prefix no longer describes the caches
        t = self._to_dev(tokens, torch.int32)
        m = self._to_dev(tokens_mask, torch.uint8)
        p = torch.arange(s, device=self.device, dtype=torch.int32)
        out = torch.empty(self.config.audio_num_codebooks, dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib.csm_prefill_slot(self._h, int(slot), t.data_ptr(), m.data_ptr(), p.data_ptr(), s, 1, float(temperature), int(topk),
                                       out.data_ptr(), _stream_ptr()), self._h)
        return out

    def refill_begin(self, slot: int, tokens: torch.Tensor, tokens_mask: torch.Tensor) -> None:
        """Starts a refill BESIDE the frame loop (csm_refill_begin): the prompt (S,33) is embedded and the slot parked; its layers run
        a few at a time through ``refill_advance`` between frame steps, so the other slots never wait for a whole prompt.  One
        refill at a time per model."""
        self._require()
        s = tokens.shape[0]
        if s >= self.bb.max_seq_len or s > max(self._max_prefill_rows, 2 * self._max_batch):
            raise ValueError(f"prompt of {s} rows exceeds the limits (max_seq_len {self.bb.max_seq_len}, max_prefill_rows {self._max_prefill_rows})")
        self._kv_prompt = None
        t = self._to_dev(tokens, torch.int32)
        m = self._to_dev(tokens_mask, torch.uint8)
        p = torch.arange(s, device=self.device, dtype=torch.int32)
        self._refill_keep = (t, m, p)                             # the position array is read by every advance call
        with self._on_device():
            check(lib.csm_refill_begin(self._h, int(slot), t.data_ptr(), m.data_ptr(), p.data_ptr(), s, _stream_ptr()), self._h)

    def refill_advance(self, max_layers: int) -> bool:
        """Up to ``max_layers`` more backbone layers of the pending refill.  True when the prompt is complete: the NEXT frame step
        yields the new utterance's frame 0 in the slot's row (csm_refill_advance)."""
        with self._on_device():
            rc = lib.csm_refill_advance(self._h, int(max_layers), _stream_ptr())
        if rc < 0:
            check(rc, self._h)
        if rc == 1:
            self._refill_keep = None
        return rc == 1

    def supports_refill_beside_the_loop(self, batch: Optional[int] = None) -> bool:
        """Whether frame steps of ``batch`` rows (default: the handle's max batch) honour a refill beside the loop -- the engine's own
        predicate (csm_refill_supported: the matrix-core decode path, which CSM_WIDE / CSM_WIDE_MIN can move or switch off)."""
        if not self._h:
            return False
        return bool(lib.csm_refill_supported(self._h, int(self._max_batch if batch is None else batch)))

    def depth(self, batch: int, temperature: float, topk: int, *, forced: Optional[torch.Tensor] = None,
              noise: Optional[torch.Tensor] = None, want_logits: bool = False, commit: bool = True):
        """c0 head + 31 decoder steps on the current backbone state -> (B,32) int32 [, logits]."""
        self._require()
        out = torch.empty(batch, self.config.audio_num_codebooks, dtype=torch.int32, device=self.device)
        logits = None
        if want_logits:
            logits = torch.empty(self.config.audio_num_codebooks, batch, self.config.audio_vocab_size,
                                 dtype=torch.bfloat16, device=self.device)
        f = forced.to(device=self.device, dtype=torch.int32).contiguous() if forced is not None else None
        n = noise.to(device=self.device, dtype=torch.bfloat16).contiguous() if noise is not None else None
        with self._on_device():
            check(lib.csm_depth(self._h, batch, float(temperature), int(topk),
                                f.data_ptr() if f is not None else None, out.data_ptr(),
                                logits.data_ptr() if logits is not None else None,
                                n.data_ptr() if n is not None else None, int(commit), _stream_ptr()), self._h)
        return (out, logits) if want_logits else out

    def step(self, batch: int, temperature: float, topk: int, use_graph: bool = True) -> None:
        """One continuing frame from on-device state (hipGraph replay); no host sync."""
        with self._on_device():
            check(lib.csm_frame_step(self._h, batch, float(temperature), int(topk), int(use_graph), _stream_ptr()), self._h)

    def last_frame(self, batch: int) -> torch.Tensor:
        out = torch.empty(batch, self.config.audio_num_codebooks, dtype=torch.int32, device=self.device)
        with self._on_device():
            check(lib.csm_copy_frame(self._h, batch, out.data_ptr(), _stream_ptr()), self._h)
        return out

    def read_frames(self, batch: int, first: int = 0, n: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(frames [n][B][32] int32 CPU, eos_at [B] int32 CPU) -- synchronises."""
        total = lib.csm_num_frames(self._h)
        n = total - first if n is None else n
        frames = torch.empty(max(n, 0), batch, self.config.audio_num_codebooks, dtype=torch.int32)
        eos = torch.empty(batch, dtype=torch.int32)
        with self._on_device():
            check(lib.csm_read_frames(self._h, batch, first, max(n, 0), frames.data_ptr() if n > 0 else None,
                                      eos.data_ptr(), _stream_ptr()), self._h)
        return frames, eos

    def num_frames(self) -> int:
        return lib.csm_num_frames(self._h)

    def frames_device_ptr(self) -> int:
        return lib.csm_frames_dev(self._h)

    @torch.inference_mode()
    def generate_frame(self, tokens: torch.Tensor, tokens_mask: torch.Tensor, input_pos: torch.Tensor,
                       temperature: float, topk: int) -> torch.Tensor:
        """reference: Model.generate_frame (sesameai/models.py:132-184).

        tokens (B,S,33), tokens_mask (B,S,33), input_pos (B,S) -> (B,32) int32 on the device.
        S > 1 (a prompt) runs prefill + depth; S == 1 stages the caller's row and replays the
        captured frame-step graph."""
        self._require()
        b, s, _ = tokens.size()
        if s > 1:
            ar = torch.arange(s, device=input_pos.device, dtype=input_pos.dtype)
            if bool((input_pos == ar.unsqueeze(0)).all()):
                self.prefill_prompt(tokens, tokens_mask)
            else:
                self._kv_prompt = None
                self.prefill(tokens, tokens_mask, input_pos)
            return self.depth(b, temperature, topk, commit=True)
        # S == 1 steps append beyond the prompt (the reference loop), so the cached prefix stays valid.  The reference's own
        # tensors -- int64 tokens / positions, bool mask, on this GPU (tts_service.py:229-241) -- go straight to ONE C-ABI call
        # (csm_generate_frame_s1): no dtype-conversion kernels, no device context, no separate copy-out; positions are checked
        # by the staging kernel (device flag -> CSM_E_TOO_LONG at the next read_frames), or here when they live on the host.
        dev = self.device
        if not (tokens.dtype is torch.int64 and tokens.device == dev and tokens.is_contiguous()):
            tokens = tokens.to(device=dev, dtype=torch.int64).contiguous()
        if not (tokens_mask.dtype is torch.bool and tokens_mask.device == dev and tokens_mask.is_contiguous()):
            tokens_mask = tokens_mask.to(device=dev, dtype=torch.bool).contiguous()
        if not (input_pos.dtype is torch.int64 and input_pos.device == dev and input_pos.is_contiguous()):
            self._check_positions(input_pos)
            input_pos = input_pos.to(device=dev, dtype=torch.int64).contiguous()
        out = torch.empty((b, self.config.audio_num_codebooks), dtype=torch.int32, device=dev)
        rc = lib.csm_generate_frame_s1(self._h, tokens.data_ptr(), tokens_mask.data_ptr(), input_pos.data_ptr(), b,
                                       temperature, topk, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        if rc:
            check(rc, self._h)
        return out

    def fast_paths(self) -> int:
        """bit mask of the all-CU launches this handle runs (include/csm_hip_ops.h csm_debug_fast_paths)."""
        return int(lib.csm_debug_fast_paths(self._h))

    def describe(self) -> str:
        """One line of text: which kernels this handle runs and which CSM_* / MIMI_* switches are set (csm_describe)."""
        self._require()
        n = lib.csm_describe(self._h, None, 0)
        buf = C.create_string_buffer(n + 1)
        lib.csm_describe(self._h, buf, n + 1)
        return buf.value.decode()

    def graph_captures(self) -> int:
        """frame-step graphs captured since setup_caches (include/csm_hip_ops.h csm_debug_graph_captures)."""
        return int(lib.csm_debug_graph_captures(self._h))

    def bytes_per_frame(self, batch: int, p_mean: float) -> float:
        return lib.csm_bytes_per_frame(self._h, batch, float(p_mean))
