"""Pins the oracle (oracle/csm_ref.py, oracle/mimi_ref.py) against the independent
transformers.models.{csm,mimi} ports present in this image.
fp32 on both sides, tolerance 1e-5 relative to the tensor's scale."""
import warnings

import pytest
import torch

csm_mod = pytest.importorskip("transformers.models.csm.modeling_csm")

from oracle import csm_ref as C
from oracle import mimi_ref as M
from oracle.hf_map import build_hf_csm, build_hf_mimi


@pytest.fixture(scope="module")
def tiny_fp32():
    shape = C.csm_tiny()
    w32 = {k: v.float() for k, v in C.make_weights(shape, norm_jitter=0.1).items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bb, dd, heads = build_hf_csm(shape, w32)
    m = C.OracleModel(shape, w32, dtype=torch.float32)
    m.setup_caches(2)
    return shape, w32, m, bb, dd, heads


def test_backbone_matches_hf(tiny_fp32):
    shape, w32, m, bb, _, _ = tiny_fp32
    g = torch.Generator().manual_seed(0)
    B, S = 2, 9
    tok = torch.zeros(B, S, 33, dtype=torch.long)
    msk = torch.zeros(B, S, 33, dtype=torch.bool)
    tok[:, :4, 32] = torch.randint(0, shape.text_vocab_size, (B, 4), generator=g); msk[:, :4, 32] = True
    tok[:, 4:, :32] = torch.randint(0, 2051, (B, 5, 32), generator=g); msk[:, 4:, :32] = True
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    m.reset_caches()
    h = m.embed_frame(tok, msk)
    mine = m.backbone.forward(h, pos, m.backbone_causal_mask[pos, :])
    with torch.no_grad():
        theirs = bb(inputs_embeds=h, use_cache=False).last_hidden_state
    assert (mine - theirs).abs().max() <= 1e-5 * theirs.abs().max()


def test_kv_cached_step_equals_full_recompute(tiny_fp32):
    """property: prefill S then 1-token steps == one full-sequence pass."""
    shape, w32, m, _, _, _ = tiny_fp32
    g = torch.Generator().manual_seed(1)
    S = 12
    h = torch.randn(1, S, shape.backbone.embed_dim, generator=g)
    pos = torch.arange(S).unsqueeze(0)
    m.reset_caches()
    full = m.backbone.forward(h, pos, m.backbone_causal_mask[pos, :])
    m.reset_caches()
    outs = [m.backbone.forward(h[:, :8], pos[:, :8], m.backbone_causal_mask[pos[:, :8], :])]
    for t in range(8, S):
        outs.append(m.backbone.forward(h[:, t:t + 1], pos[:, t:t + 1], m.backbone_causal_mask[pos[:, t:t + 1], :]))
    assert torch.allclose(torch.cat(outs, 1), full, atol=2e-5, rtol=0)


def test_depth_decoder_and_heads_match_hf(tiny_fp32):
    shape, w32, m, _, dd, heads = tiny_fp32
    g = torch.Generator().manual_seed(2)
    B, n = 2, 6
    x = torch.randn(B, n, shape.backbone.embed_dim, generator=g)
    dpos = torch.arange(n).unsqueeze(0).repeat(B, 1)
    m.decoder.reset_caches()
    mine = m.decoder.forward(torch.nn.functional.linear(x, w32["projection.weight"]), dpos,
                             m.decoder_causal_mask[dpos, :])
    with torch.no_grad():
        theirs = dd(inputs_embeds=x, use_cache=False).last_hidden_state
    assert (mine - theirs).abs().max() <= 1e-5 * theirs.abs().max()
    # per-codebook head: h @ audio_head[i-1]  (models.py:176) == HF CsmCodebooksHead
    from transformers.models.csm.modeling_csm import CsmCodebooksHead
    head = CsmCodebooksHead(shape.decoder.embed_dim, shape.audio_num_codebooks, shape.audio_vocab_size)
    with torch.no_grad():
        head.weight.copy_(heads)
        hf_logits = head(theirs[:, 1:], codebook_indices=torch.arange(1, n))
    mine_logits = torch.stack([torch.mm(mine[:, i], w32["audio_head"][i - 1]) for i in range(1, n)], 1)
    assert (mine_logits - hf_logits).abs().max() <= 1e-5 * hf_logits.abs().max()


@pytest.mark.parametrize("which", ["tiny", "full"])
def test_mimi_decode_matches_hf(which):
    s = M.mimi_tiny() if which == "tiny" else M.mimi_full()
    w = M.make_weights(s)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hf = build_hf_mimi(s, w)
    codes = torch.randint(0, s.codebook_size, (2, 32, 12), generator=torch.Generator().manual_seed(1))
    mine = M.decode(s, w, codes)
    with torch.no_grad():
        theirs = hf.decode(codes)[0]
    assert mine.shape == theirs.shape == (2, 1, 12 * 1920)
    assert (mine - theirs).abs().max() <= 1e-5 * theirs.abs().max()


@pytest.mark.parametrize("which", ["tiny", "full"])
def test_mimi_encode_matches_hf(which):
    """ENCODE side (SEANet encoder, encoder transformer, replicate-padded stride-2 downsample,
    split RVQ nearest-centroid search) against transformers' MimiModel.encode."""
    s = M.mimi_tiny() if which == "tiny" else M.mimi_full()
    w = M.make_weights(s, encoder=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hf = build_hf_mimi(s, w)
    wav = torch.randn(2, 1, 1920 * 7 + 333, generator=torch.Generator().manual_seed(1)) * 0.3
    mine = M.encode(s, w, wav)
    with torch.no_grad():
        theirs = hf.encode(wav)[0]
    assert mine.shape == theirs.shape == (2, 32, 8)
    assert (mine == theirs).all(dim=1).float().mean() >= 0.95          # whole frames (fp32 near-ties aside)


def test_mimi_decode_is_causal_and_stateless_chunks_differ():
    """frames >= t never change samples < 1920 t; the reference's stateless
    10-frame chunking (generator.py:111-117) is NOT equal to whole decode after chunk 0."""
    s = M.mimi_tiny()
    w = M.make_weights(s)
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(0, s.codebook_size, (1, 32, 14), generator=g)
    whole = M.decode(s, w, codes)
    pert = codes.clone()
    pert[:, :, 9:] = torch.randint(0, s.codebook_size, (1, 32, 5), generator=g)
    assert torch.equal(M.decode(s, w, pert)[..., : 9 * 1920], whole[..., : 9 * 1920])
    ch = M.decode_stateless_chunks(s, w, codes, 10)
    assert torch.allclose(ch[..., :19200], whole[..., :19200], atol=1e-4)
    assert (ch[..., 19200:] - whole[..., 19200:]).abs().max() > 1e-2


# ----------------------------------------------------------------------------------------
# bf16 pin (VERDICT r1 "oracle pin is fp32-only and tiny-only").
#
# The bf16 oracle and the bf16 HF port (SDPA attention, the op torchtune calls) differ at exactly THREE points, each a
# property of torchtune 0.4.0 that the HF port does not share:
#   1. the RoPE table: torchtune builds the llama3-scaled frequencies in Python floats (double) and rounds the fp32
#      cos/sin buffer to bf16 with the model (sesameai/generator.py:343); HF builds them in fp32 tensor math -- some
#      entries land one bf16 ulp apart;
#   2. RoPE application: torchtune computes x*cos -/+ x*sin in fp32 and rounds once (`.type_as(x)`); HF's eager bf16
#      `(q * cos) + (rotate_half(q) * sin)` rounds each product and the sum;
#   3. the key range: torchtune attends over the whole position-indexed cache under a boolean mask
#      (sesameai/models.py:55-69,154,172), HF over the live keys only -- torch's CPU flash kernel blocks the two
#      differently.
# With those three aligned by the oracle's test-only knobs the two implementations are BIT-IDENTICAL, on the tiny
# stacks and on full-width CSM-1B layers -- every other rounding point (RMSNorm round-before-scale, Linear outputs,
# SDPA, SiLU and the product as separate bf16 ops, residual adds, final norm) is thereby pinned in bf16.  The three
# points themselves are then bounded one by one.
# ----------------------------------------------------------------------------------------
def _hf_rope_table(model, s, n):
    x = torch.zeros(1, n, s.embed_dim, dtype=torch.bfloat16)
    with torch.no_grad():
        cos, sin = model.rotary_emb(x, torch.arange(n).unsqueeze(0))
    return torch.stack([cos[0, :, : s.head_dim // 2], sin[0, :, : s.head_dim // 2]], dim=-1)      # [n][hd/2][2] bf16


def _full_width_shape():
    """CSM-1B layer shapes (2048 / 8192 / 32 heads / 8 KV heads / hd 64 and 1024 / 8192 / 8 / 2 / hd 128), 2 layers per
    stack and a small text vocabulary so the CPU suite stays in its time budget."""
    return C.CsmShape(backbone=C.LlamaShape(2, 32, 8, 2048, 8192), decoder=C.LlamaShape(2, 8, 2, 1024, 8192),
                      text_vocab_size=1000, audio_vocab_size=2051, audio_num_codebooks=32)


@pytest.fixture(scope="module", params=["tiny", "full_width"])
def bf16_pair(request):
    shape = C.csm_tiny() if request.param == "tiny" else _full_width_shape()
    w = C.make_weights(shape, norm_jitter=0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bb, dd, heads = build_hf_csm(shape, w, dtype=torch.bfloat16, attn="sdpa")
    m = C.OracleModel(shape, w)
    m.setup_caches(2)
    return shape, w, m, bb, dd


def _run_both(shape, w, m, bb, dd, seed):
    g = torch.Generator().manual_seed(seed)
    B, S, n = 2, 9, 6
    h = torch.randn(B, S, shape.backbone.embed_dim, generator=g).to(torch.bfloat16)
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    m.reset_caches()
    mine_bb = m.backbone.forward(h, pos, m.backbone_causal_mask[pos, :])
    x = torch.randn(B, n, shape.backbone.embed_dim, generator=g).to(torch.bfloat16)
    dpos = torch.arange(n).unsqueeze(0).repeat(B, 1)
    m.decoder.reset_caches()
    mine_dd = m.decoder.forward(torch.nn.functional.linear(x, w["projection.weight"]), dpos, m.decoder_causal_mask[dpos, :])
    with torch.no_grad():
        hf_bb = bb(inputs_embeds=h, use_cache=False).last_hidden_state
        hf_dd = dd(inputs_embeds=x, use_cache=False).last_hidden_state
    return (mine_bb, hf_bb), (mine_dd, hf_dd)


def test_bf16_oracle_is_bit_identical_to_bf16_hf_once_the_three_torchtune_points_are_aligned(bf16_pair):
    shape, w, m, bb, dd = bf16_pair
    saved = (m.backbone.table, m.decoder.table)
    try:
        C.ROPE_ROUNDING, C.ATTN_KEYS = "hf", "live"
        m.backbone.table = _hf_rope_table(bb, shape.backbone, shape.backbone.max_seq_len)
        m.decoder.table = _hf_rope_table(dd, shape.decoder, shape.audio_num_codebooks)
        for (mine, theirs) in _run_both(shape, w, m, bb, dd, seed=3):
            assert mine.dtype == theirs.dtype == torch.bfloat16
            assert torch.equal(mine, theirs), f"{int((mine != theirs).sum())} of {mine.numel()} hidden-state elements differ"
    finally:
        C.ROPE_ROUNDING, C.ATTN_KEYS = "torchtune", "cache"
        m.backbone.table, m.decoder.table = saved


def test_bf16_oracle_vs_bf16_hf_with_torchtune_semantics_stays_within_the_three_points_reach(bf16_pair):
    """Default knobs (= the reference's torchtune semantics): the hidden states may differ from the HF port, but only by
    what points 1-3 can cause -- a few bf16 ulps of the tensor's scale after 2 layers, most elements identical or 1 ulp."""
    shape, w, m, bb, dd = bf16_pair
    for (mine, theirs) in _run_both(shape, w, m, bb, dd, seed=4):
        scale = theirs.float().abs().max().item()
        d = (mine.float() - theirs.float()).abs()
        ulp_of_scale = scale * 2.0 ** -8
        near = (d <= 2 * theirs.float().abs().clamp(min=scale / 64) * 2.0 ** -8).float().mean().item()
        print(f"bf16 oracle vs bf16 HF, torchtune semantics: max|d| = {d.max().item() / ulp_of_scale:.1f} ulp of the scale, "
              f"{(mine == theirs).float().mean().item():.3f} identical, {near:.3f} within 2 ulp")
        assert d.max().item() <= 8 * ulp_of_scale, (d.max().item(), scale)
        assert near >= 0.5


def test_the_three_torchtune_points_one_by_one():
    s = C.csm_1b().backbone
    # 1. table: the oracle's bf16 table is the correctly rounded fp64 cos/sin of torchtune's (double-built) frequencies
    theta = C.llama3_scaled_rope_theta(s.head_dim, s.rope_base, s.scale_factor)
    ang = torch.arange(s.max_seq_len, dtype=torch.float32)[:, None] * theta[None, :]            # fp32 product, as torchtune's einsum
    exact = torch.stack([torch.cos(ang.double()), torch.sin(ang.double())], -1)
    tab = C.rope_table(s)
    assert tab.dtype == torch.bfloat16
    err = (tab.double() - exact).abs()
    # bf16 round-to-nearest of the fp32 cos/sin: half a bf16 ulp (of the value's binade: up to 2^-8 relative) + fp32 noise
    assert bool((err <= exact.abs() * 2.0 ** -8 + 2.0 ** -20).all()), float((err - exact.abs() * 2.0 ** -8).max())
    # 2. application: one rounding of fp32 math (error <= 1/2 ulp of the result) vs HF's three roundings
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 64, 4, s.head_dim, generator=g).to(torch.bfloat16)
    pos = torch.arange(64).unsqueeze(0) * 31
    xs = x.double().reshape(1, 64, 4, -1, 2); rc = tab[pos].double().view(1, 64, 1, -1, 2)
    want = torch.stack([xs[..., 0] * rc[..., 0] - xs[..., 1] * rc[..., 1], xs[..., 1] * rc[..., 0] + xs[..., 0] * rc[..., 1]], -1).flatten(3)
    tt = C.apply_rope(x, tab, pos)
    half_ulp = want.abs().clamp(min=2.0 ** -20) * 2.0 ** -8      # >= half an ulp of every value's binade
    assert bool(((tt.double() - want).abs() <= half_ulp + 2.0 ** -20).all())
    try:
        C.ROPE_ROUNDING = "hf"
        hf = C.apply_rope(x, tab, pos)
    finally:
        C.ROPE_ROUNDING = "torchtune"
    # HF's form rounds both products and the sum: its error scales with the PRODUCTS, not with the (possibly cancelling) result
    mag = torch.stack([(xs[..., 0] * rc[..., 0]).abs() + (xs[..., 1] * rc[..., 1]).abs(),
                       (xs[..., 1] * rc[..., 0]).abs() + (xs[..., 0] * rc[..., 1]).abs()], -1).flatten(3)
    assert bool(((hf.double() - want).abs() <= 2.1 * mag * 2.0 ** -8 + 2.0 ** -20).all()) and not torch.equal(hf, tt)
    # 3. key range: the whole-cache form and the live-keys form of the same attention differ by at most one ulp
    shape = C.csm_tiny()
    w = C.make_weights(shape, norm_jitter=0.1)
    m = C.OracleModel(shape, w); m.setup_caches(1)
    h = torch.randn(1, 40, shape.backbone.embed_dim, generator=g).to(torch.bfloat16)
    pos = torch.arange(40).unsqueeze(0)
    m.reset_caches(); a = m.backbone.forward(h, pos, m.backbone_causal_mask[pos, :])
    try:
        C.ATTN_KEYS = "live"
        m.reset_caches(); b = m.backbone.forward(h, pos, m.backbone_causal_mask[pos, :])
    finally:
        C.ATTN_KEYS = "cache"
    scale = a.float().abs().max().item()
    assert (a.float() - b.float()).abs().max().item() <= 4 * scale * 2.0 ** -8
    assert (a == b).float().mean().item() >= 0.8
