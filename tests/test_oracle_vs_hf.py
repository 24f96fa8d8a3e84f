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
