"""Secondary cross-check harness: load the oracle's weights into the independent
``transformers.models.csm`` / ``transformers.models.mimi`` ports that ship in this image.

TEST INFRASTRUCTURE ONLY (see oracle/csm_ref.py header).  The HF port is architecture-
equivalent to the reference's torchtune/moshi graph but uses a different weight layout:
half-split RoPE instead of torchtune's interleaved pairs, so each head's q/k rows are
permuted by ``cat(arange(0,hd,2), arange(1,hd,2))``.
"""
from __future__ import annotations

from typing import Dict

import torch

from .csm_ref import CsmShape, LlamaShape


def _rope_params(s: LlamaShape) -> dict:
    return dict(rope_type="llama3", rope_theta=s.rope_base, factor=s.scale_factor,
                low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192)


def _perm_rows(wt: torch.Tensor, n_heads: int, hd: int) -> torch.Tensor:
    """interleaved (torchtune) -> half-split (HF) row order inside each head."""
    perm = torch.cat([torch.arange(0, hd, 2), torch.arange(1, hd, 2)])
    return wt.view(n_heads, hd, -1)[:, perm, :].reshape(n_heads * hd, -1)


def _load_layers(hf_layers, w: Dict[str, torch.Tensor], pfx: str, s: LlamaShape, dtype) -> None:
    hd = s.head_dim
    for i, layer in enumerate(hf_layers):
        L = f"{pfx}.layers.{i}"
        sd = {
            "self_attn.q_proj.weight": _perm_rows(w[f"{L}.attn.q_proj.weight"], s.num_heads, hd),
            "self_attn.k_proj.weight": _perm_rows(w[f"{L}.attn.k_proj.weight"], s.num_kv_heads, hd),
            "self_attn.v_proj.weight": w[f"{L}.attn.v_proj.weight"],
            "self_attn.o_proj.weight": w[f"{L}.attn.output_proj.weight"],
            "mlp.gate_proj.weight": w[f"{L}.mlp.w1.weight"],
            "mlp.up_proj.weight": w[f"{L}.mlp.w3.weight"],
            "mlp.down_proj.weight": w[f"{L}.mlp.w2.weight"],
            "input_layernorm.weight": w[f"{L}.sa_norm.scale"],
            "post_attention_layernorm.weight": w[f"{L}.mlp_norm.scale"],
        }
        layer.load_state_dict({k: v.to(dtype) for k, v in sd.items()}, strict=True)


def build_hf_csm(shape: CsmShape, w: Dict[str, torch.Tensor], dtype=torch.float32, attn: str = "eager"):
    """Returns (backbone: CsmBackboneModel, depth: CsmDepthDecoderModel, heads [31,d,V]).
    ``attn``: "eager" (explicit matmul/softmax; the fp32 cross-check) or "sdpa" (F.scaled_dot_product_attention, the op
    torchtune 0.4.0 calls -- the bf16 cross-check needs the same fused kernel on both sides)."""
    from transformers.models.csm.configuration_csm import CsmConfig, CsmDepthDecoderConfig
    from transformers.models.csm.modeling_csm import CsmBackboneModel, CsmDepthDecoderModel

    bb, dec = shape.backbone, shape.decoder
    dcfg = CsmDepthDecoderConfig(
        num_codebooks=shape.audio_num_codebooks, backbone_hidden_size=bb.embed_dim,
        vocab_size=shape.audio_vocab_size, hidden_size=dec.embed_dim,
        intermediate_size=dec.intermediate_dim, num_hidden_layers=dec.num_layers,
        num_attention_heads=dec.num_heads, num_key_value_heads=dec.num_kv_heads,
        max_position_embeddings=shape.audio_num_codebooks + 1, rms_norm_eps=dec.norm_eps,
        rope_parameters=_rope_params(dec), head_dim=dec.head_dim)
    cfg = CsmConfig(
        num_codebooks=shape.audio_num_codebooks, vocab_size=shape.audio_vocab_size,
        text_vocab_size=shape.text_vocab_size, hidden_size=bb.embed_dim,
        intermediate_size=bb.intermediate_dim, num_hidden_layers=bb.num_layers,
        num_attention_heads=bb.num_heads, num_key_value_heads=bb.num_kv_heads,
        max_position_embeddings=bb.max_seq_len, rms_norm_eps=bb.norm_eps,
        rope_parameters=_rope_params(bb), head_dim=bb.head_dim,
        depth_decoder_config=dcfg.to_dict())
    cfg._attn_implementation = attn
    dcfg._attn_implementation = attn
    with torch.no_grad():
        backbone = CsmBackboneModel(cfg).to(dtype).eval()
        depth = CsmDepthDecoderModel(dcfg).to(dtype).eval()
        _load_layers(backbone.layers, w, "backbone", bb, dtype)
        backbone.norm.weight.copy_(w["backbone.norm.scale"].to(dtype))
        backbone.embed_tokens.embed_audio_tokens.weight.copy_(w["audio_embeddings.weight"].to(dtype))
        _load_layers(depth.layers, w, "decoder", dec, dtype)
        depth.norm.weight.copy_(w["decoder.norm.scale"].to(dtype))
        depth.inputs_embeds_projector.weight.copy_(w["projection.weight"].to(dtype))
        depth.embed_tokens.weight.copy_(w["audio_embeddings.weight"].to(dtype))
    return backbone, depth, w["audio_head"].to(dtype)


# ----------------------------------------------------------------------------------------
# Mimi
# ----------------------------------------------------------------------------------------
def build_hf_mimi(s, w: Dict[str, torch.Tensor]):
    """MimiModel (HF) carrying the oracle's decode-side weights (encoder left random)."""
    from transformers.models.mimi.configuration_mimi import MimiConfig
    from transformers.models.mimi.modeling_mimi import MimiModel

    cfg = MimiConfig(
        hidden_size=s.hidden, num_filters=s.n_filters, upsampling_ratios=list(s.ratios),
        kernel_size=s.kernel, last_kernel_size=s.last_kernel, residual_kernel_size=s.res_kernel,
        compress=s.compress, codebook_size=s.codebook_size, codebook_dim=s.codebook_dim,
        vector_quantization_hidden_dimension=s.codebook_dim, num_quantizers=s.num_codebooks,
        num_semantic_quantizers=s.num_semantic, upsample_groups=s.hidden,
        num_hidden_layers=s.tr_layers, intermediate_size=s.tr_ffn,
        num_attention_heads=s.tr_heads, num_key_value_heads=s.tr_heads,
        head_dim=s.hidden // s.tr_heads, sliding_window=s.tr_context, norm_eps=s.norm_eps,
        rope_parameters=dict(rope_type="default", rope_theta=s.rope_theta), use_cache=False)
    cfg._attn_implementation = "eager"
    m = MimiModel(cfg).float().eval()
    d, H = s.hidden, s.tr_heads
    hd = d // H
    with torch.no_grad():
        q = m.quantizer
        for k in range(s.num_codebooks):
            rvq = q.semantic_residual_vector_quantizer if k < s.num_semantic else q.acoustic_residual_vector_quantizer
            cb = rvq.layers[k if k < s.num_semantic else k - s.num_semantic].codebook
            cb.embed_sum.copy_(w[f"rvq.{k}.embedding_sum"])
            cb.cluster_usage.copy_(w[f"rvq.{k}.cluster_usage"])
            cb._embed = None
        q.semantic_residual_vector_quantizer.output_proj.weight.copy_(w["rvq_first.output_proj.weight"])
        q.acoustic_residual_vector_quantizer.output_proj.weight.copy_(w["rvq_rest.output_proj.weight"])
        m.upsample.conv.weight.copy_(w["upsample.convtr.weight"])
        for i, layer in enumerate(m.decoder_transformer.layers):
            L = f"transformer.{i}"
            wq, wk, wv = w[f"{L}.in_proj_weight"].chunk(3, dim=0)
            layer.self_attn.q_proj.weight.copy_(_perm_rows(wq, H, hd))
            layer.self_attn.k_proj.weight.copy_(_perm_rows(wk, H, hd))
            layer.self_attn.v_proj.weight.copy_(wv)
            layer.self_attn.o_proj.weight.copy_(w[f"{L}.out_proj.weight"])
            layer.input_layernorm.weight.copy_(w[f"{L}.norm1.weight"])
            layer.input_layernorm.bias.copy_(w[f"{L}.norm1.bias"])
            layer.post_attention_layernorm.weight.copy_(w[f"{L}.norm2.weight"])
            layer.post_attention_layernorm.bias.copy_(w[f"{L}.norm2.bias"])
            layer.mlp.fc1.weight.copy_(w[f"{L}.linear1.weight"])
            layer.mlp.fc2.weight.copy_(w[f"{L}.linear2.weight"])
            layer.self_attn_layer_scale.scale.copy_(w[f"{L}.layer_scale_1.scale"])
            layer.mlp_layer_scale.scale.copy_(w[f"{L}.layer_scale_2.scale"])
        dl = m.decoder.layers
        dl[0].conv.weight.copy_(w["seanet.conv_in.weight"]); dl[0].conv.bias.copy_(w["seanet.conv_in.bias"])
        for j in range(len(s.ratios)):
            ct, rb = dl[2 + 3 * j], dl[3 + 3 * j]
            ct.conv.weight.copy_(w[f"seanet.up.{j}.convtr.weight"]); ct.conv.bias.copy_(w[f"seanet.up.{j}.convtr.bias"])
            rb.block[1].conv.weight.copy_(w[f"seanet.up.{j}.res.conv1.weight"])
            rb.block[1].conv.bias.copy_(w[f"seanet.up.{j}.res.conv1.bias"])
            rb.block[3].conv.weight.copy_(w[f"seanet.up.{j}.res.conv2.weight"])
            rb.block[3].conv.bias.copy_(w[f"seanet.up.{j}.res.conv2.bias"])
        last = dl[2 + 3 * len(s.ratios)]
        last.conv.weight.copy_(w["seanet.conv_out.weight"]); last.conv.bias.copy_(w["seanet.conv_out.bias"])
        if "enc.conv_in.weight" in w:          # encode side
            el = m.encoder.layers
            el[0].conv.weight.copy_(w["enc.conv_in.weight"]); el[0].conv.bias.copy_(w["enc.conv_in.bias"])
            for j in range(len(s.ratios)):
                rb, cv = el[1 + 3 * j], el[3 + 3 * j]
                rb.block[1].conv.weight.copy_(w[f"enc.down.{j}.res.conv1.weight"]); rb.block[1].conv.bias.copy_(w[f"enc.down.{j}.res.conv1.bias"])
                rb.block[3].conv.weight.copy_(w[f"enc.down.{j}.res.conv2.weight"]); rb.block[3].conv.bias.copy_(w[f"enc.down.{j}.res.conv2.bias"])
                cv.conv.weight.copy_(w[f"enc.down.{j}.conv.weight"]); cv.conv.bias.copy_(w[f"enc.down.{j}.conv.bias"])
            lastc = el[2 + 3 * len(s.ratios)]
            lastc.conv.weight.copy_(w["enc.conv_out.weight"]); lastc.conv.bias.copy_(w["enc.conv_out.bias"])
            for i, layer in enumerate(m.encoder_transformer.layers):
                L = f"enc_transformer.{i}"
                wq, wk, wv = w[f"{L}.in_proj_weight"].chunk(3, dim=0)
                layer.self_attn.q_proj.weight.copy_(_perm_rows(wq, H, hd))
                layer.self_attn.k_proj.weight.copy_(_perm_rows(wk, H, hd))
                layer.self_attn.v_proj.weight.copy_(wv)
                layer.self_attn.o_proj.weight.copy_(w[f"{L}.out_proj.weight"])
                layer.input_layernorm.weight.copy_(w[f"{L}.norm1.weight"]); layer.input_layernorm.bias.copy_(w[f"{L}.norm1.bias"])
                layer.post_attention_layernorm.weight.copy_(w[f"{L}.norm2.weight"]); layer.post_attention_layernorm.bias.copy_(w[f"{L}.norm2.bias"])
                layer.mlp.fc1.weight.copy_(w[f"{L}.linear1.weight"]); layer.mlp.fc2.weight.copy_(w[f"{L}.linear2.weight"])
                layer.self_attn_layer_scale.scale.copy_(w[f"{L}.layer_scale_1.scale"]); layer.mlp_layer_scale.scale.copy_(w[f"{L}.layer_scale_2.scale"])
            m.downsample.conv.weight.copy_(w["downsample.conv.weight"])
            q.semantic_residual_vector_quantizer.input_proj.weight.copy_(w["rvq_first.input_proj.weight"])
            q.acoustic_residual_vector_quantizer.input_proj.weight.copy_(w["rvq_rest.input_proj.weight"])
    return m
