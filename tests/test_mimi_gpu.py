"""Mimi decode parity: HIP path (sesameai.mimi.MimiCodec -> include/mimi_hip.h) vs the oracle
(oracle/mimi_ref.py) and the committed golden PCM.  fp32 both sides; the only freedom is the
fp32 summation order inside a dot product, so the tolerance is 2e-5 of the clip's peak."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 2e-5


@pytest.fixture(scope="module")
def tiny_codec():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import mimi_ref as M
    from sesameai.mimi import MimiCodec, mimi_tiny_args, synthetic_state_dict
    s = M.mimi_tiny()
    w = M.make_weights(s, seed=4321, encoder=True)
    sd = synthetic_state_dict(mimi_tiny_args(), seed=4321)
    assert set(w) == set(sd) and all(torch.equal(w[k], sd[k]) for k in w), "product and oracle synthetic Mimi weights differ"
    return s, w, MimiCodec(mimi_tiny_args(), sd, max_frames=64)


def _close(got, want, what):
    got, want = got.detach().cpu().float(), want.detach().cpu().float()
    assert got.shape == want.shape, f"{what}: {tuple(got.shape)} vs {tuple(want.shape)}"
    peak = want.abs().max().item()
    err = (got - want).abs().max().item()
    print(f"{what}: max|d|={err:.3g} peak={peak:.3g} rel={err / peak:.2g}")
    assert err <= REL_TOL * peak, f"{what}: max abs err {err} vs peak {peak}"


def test_tiny_decode_vs_oracle_and_golden(tiny_codec):
    from oracle import mimi_ref as M
    s, w, codec = tiny_codec
    gold = torch.load(os.path.join(GOLD, "mimi_tiny.pt"))
    pcm = codec.decode(gold["codes"])
    assert pcm.shape == (1, 1, 1920 * gold["codes"].shape[-1]) and pcm.dtype == torch.float32
    _close(pcm, gold["pcm"], "tiny whole decode vs golden")
    codes = torch.randint(0, 2048, (3, 32, 7), generator=torch.Generator().manual_seed(4))
    _close(codec.decode(codes), M.decode(s, w, codes), "tiny batch-3 decode vs oracle")


def test_stateless_chunks_match_reference_stream_semantics(tiny_codec):
    """generate_stream decodes every 10-frame buffer independently (generator.py:111-117)."""
    s, w, codec = tiny_codec
    gold = torch.load(os.path.join(GOLD, "mimi_tiny.pt"))
    codes = gold["codes"]
    got = torch.cat([codec.decode(codes[..., t:t + 10]) for t in range(0, codes.shape[-1], 10)], dim=-1)
    _close(got[..., ::16], gold["chunks_stride16"], "stateless 10-frame chunks vs golden")


def test_chunk_decodes_replayed_from_the_graph_equal_the_launch_chain(tiny_codec):
    """A stateless decode of <= 32 frames replays its middle (everything between the code lookup and the output convolution) from
    a hipGraph captured at the second decode of that length (mimi_engine.hip decode_one): the first call of a length runs the launch
    chain, the later ones the graph -- same bits, for other codes too, with other lengths in between and from another stream."""
    from oracle import mimi_ref as M
    from sesameai.mimi import MimiCodec, mimi_tiny_args, synthetic_state_dict
    s, w, codec = tiny_codec
    g = torch.Generator().manual_seed(12)
    a7, b7, c10 = (torch.randint(0, 2048, (1, 32, n), generator=g) for n in (7, 7, 10))
    chain = MimiCodec(mimi_tiny_args(), synthetic_state_dict(mimi_tiny_args(), seed=4321), max_frames=64)
    want_a = chain.decode(a7)                                      # a fresh handle's FIRST decode of a length: the launch chain
    chain2 = MimiCodec(mimi_tiny_args(), synthetic_state_dict(mimi_tiny_args(), seed=4321), max_frames=64)
    want_b = chain2.decode(b7)
    codec.decode(a7); codec.decode(a7)                             # the second use captures: from here T = 7 is a replay on this handle
    side = torch.cuda.Stream()
    for _ in range(3):
        assert torch.equal(codec.decode(a7), want_a)
        codec.decode(c10)                                          # another length in between (a graph of its own)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            got_b = codec.decode(b7)
        side.synchronize()
        assert torch.equal(got_b, want_b)
    _close(want_a, M.decode(s, w, a7), "7-frame chunk vs oracle")


def test_stateful_stream_equals_whole_decode(tiny_codec):
    """property: Mimi decode is strictly causal, so a stateful stream in ragged chunks
    (1, 2, 10, 3, ... frames) reproduces the whole-utterance decode."""
    s, w, codec = tiny_codec
    codes = torch.randint(0, 2048, (1, 32, 23), generator=torch.Generator().manual_seed(6))
    whole = codec.decode(codes)
    codec.reset_stream()
    outs, t = [], 0
    for n in (1, 2, 10, 3, 1, 6):
        outs.append(codec.decode_stream(codes[..., t:t + n])); t += n
    stream = torch.cat(outs, dim=-1)
    err = (stream - whole).abs().max().item()
    print(f"stateful stream vs whole: max|d|={err:.3g}")
    assert err <= 1e-5 * whole.abs().max().item()


def test_codes_beyond_codebook_are_clamped(tiny_codec):
    """CSM's audio vocab is 2051 but Mimi's codebooks hold 2048 entries: the reference would
    raise inside F.embedding; the HIP path clamps (documented in include/mimi_hip.h)."""
    s, w, codec = tiny_codec
    codes = torch.randint(0, 2048, (1, 32, 4), generator=torch.Generator().manual_seed(8))
    hi = codes.clone(); hi[0, 5, 2] = 2050
    cl = codes.clone(); cl[0, 5, 2] = 2047
    assert torch.equal(codec.decode(hi), codec.decode(cl))


def _check_codes(got, s, w, wav, what):
    """RVQ codes are integers: bit-exact, except that an fp32 near-tie between two centroids may fall the other way, after
    which the remaining levels of that frame's stack legitimately differ.  So walk the oracle's residual quantiser and
    require, at the FIRST level where a frame's codes part, that the two centroids are equidistant to 1e-4 (relative) from
    the oracle's residual; everything before that level must be identical, and >= 90 % of the frames must match throughout."""
    import torch.nn.functional as F
    from oracle import mimi_ref as M
    got = got.cpu()
    want = M.encode(s, w, wav)
    assert got.shape == want.shape and got.dtype == torch.int64
    z = M.encode_latent(s, w, wav)
    emb = lambda k: w[f"rvq.{k}.embedding_sum"] / w[f"rvq.{k}.cluster_usage"].clamp(min=1e-5)[:, None]
    n_ties, worst = 0, 0.0
    for proj, levels in (("rvq_first.input_proj.weight", range(0, s.num_semantic)), ("rvq_rest.input_proj.weight", range(s.num_semantic, s.num_codebooks))):
        res = F.conv1d(z, w[proj]).transpose(1, 2)                                  # (B, T, D)
        parted = torch.zeros(res.shape[:2], dtype=torch.bool)                         # frames whose stack has already parted
        for k in levels:
            e = emb(k)
            d = torch.cdist(res.reshape(1, -1, res.shape[-1]), e[None], p=2)[0].view(*res.shape[:2], -1)
            idx = d.argmin(dim=-1)
            assert torch.equal(idx, want[:, k]), "oracle walk out of step with M.encode"
            differs = (got[:, k] != idx) & ~parted
            if differs.any():
                dg = torch.gather(d, 2, got[:, k].unsqueeze(-1))[..., 0][differs]
                dw = torch.gather(d, 2, idx.unsqueeze(-1))[..., 0][differs]
                rel = ((dg - dw) / dw.clamp(min=1e-12))
                worst = max(worst, rel.max().item()); n_ties += int(differs.sum())
                assert rel.max().item() <= 1e-4, f"{what}: level {k} picks a centroid {rel.max().item():.3g} farther (relative) than the oracle's"
            parted |= differs
            res = res - F.embedding(idx, e)
    frames_ok = (got == want).all(dim=1).float().mean().item()
    print(f"{what}: frames identical on all levels {frames_ok:.3f}, all codes {(got == want).float().mean().item():.3f}; "
          f"{n_ties} first differences, every one an fp32 near-tie (worst relative distance gap {worst:.2e})")
    assert frames_ok >= 0.9 and (got[:, 0] == want[:, 0]).float().mean().item() >= 0.97


def test_tiny_encode_vs_oracle(tiny_codec):
    """Mimi ENCODE (voice-prompt audio -> codes, generator.py:86): ragged length, batch of 2."""
    from oracle import mimi_ref as M
    s, w, codec = tiny_codec
    wav = torch.randn(2, 1, 1920 * 9 + 777, generator=torch.Generator().manual_seed(12)) * 0.3
    codes = codec.encode(wav)
    assert codes.shape == (2, 32, 10)
    _check_codes(codes, s, w, wav, "tiny encode")
    # encode -> decode round trip runs and has the right length
    pcm = codec.decode(codes)
    assert pcm.shape == (2, 1, 10 * 1920) and torch.isfinite(pcm).all()


def test_full_size_encode_vs_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import mimi_ref as M
    from sesameai.mimi import MimiArgs, MimiCodec, synthetic_state_dict
    s = M.mimi_full()
    w = M.make_weights(s, seed=4321, encoder=True)
    codec = MimiCodec(MimiArgs(), synthetic_state_dict(MimiArgs(), seed=4321), max_frames=32)
    wav = torch.randn(1, 1, 1920 * 20 + 5, generator=torch.Generator().manual_seed(13)) * 0.3
    _check_codes(codec.encode(wav), s, w, wav, "full-size encode")


def test_full_size_decode_vs_golden():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sesameai.mimi import MimiArgs, MimiCodec, synthetic_state_dict
    gold = torch.load(os.path.join(GOLD, "mimi_full.pt"))
    codec = MimiCodec(MimiArgs(), synthetic_state_dict(MimiArgs(), seed=int(gold["weight_seed"])), max_frames=32)
    pcm = codec.decode(gold["codes"])
    _close(pcm[..., ::16], gold["pcm_stride16"], "full-size decode vs golden (every 16th sample)")
    _close(pcm[..., :4096], gold["pcm_head"], "full-size head")
    _close(pcm[..., -4096:], gold["pcm_tail"], "full-size tail")
    got = torch.cat([codec.decode(gold["codes"][..., t:t + 10]) for t in range(0, gold["codes"].shape[-1], 10)], dim=-1)
    _close(got[..., ::16], gold["chunks_stride16"], "full-size stateless chunks")


def test_warm_up_leaves_no_trace_in_what_a_seeded_request_produces(tiny_codec):
    """Generator.warm_up (called by load_csm_1b) runs a synthetic utterance through the streaming and the whole-utterance path so that
    a process's one-time costs are not paid by the first request.  A request that seeds the sampler afterwards produces exactly what it
    produces on a Generator that was never warmed, its whole prompt is prefilled (the synthetic prompt is nobody's prefix), and a
    Generator without a codec skips the warm-up."""
    from sesameai.generator import Generator, Segment
    from sesameai.models import Model, csm_tiny_args
    s, w, codec = tiny_codec
    g = torch.Generator().manual_seed(2)
    ctx = [Segment(speaker=1, text=torch.randint(0, 1000, (6,), generator=g).tolist(), audio_codes=torch.randint(0, 2048, (32, 9), generator=g))]
    text = torch.randint(0, 1000, (5,), generator=g).tolist()
    outs = []
    for warm in (False, True):
        model = Model(csm_tiny_args(), None, max_frames=64, max_prefill_rows=128)
        gen = Generator(model, audio_tokenizer=codec)
        if warm:
            gen.warm_up()
            assert model._kv_prompt is None
        model.seed(5)
        chunks = list(gen.generate_stream(text, 1, ctx, max_audio_length_ms=13 * 80, temperature=0.9, topk=50))
        assert model.last_prefill_rows == 6 + 9 + 1 + 5
        outs.append(torch.cat(chunks))
    assert torch.equal(outs[0], outs[1])
    bare = Generator(Model(csm_tiny_args(), None, max_frames=16, max_prefill_rows=128), audio_tokenizer=None)
    bare.warm_up()                                                    # nothing to decode with: a no-op, not an error


def test_generator_end_to_end_tiny(tiny_codec):
    """Generator.generate / generate_stream keep the reference's shapes: (n*1920,) fp32 audio."""
    from sesameai.generator import Generator, Segment
    from sesameai.models import Model, csm_tiny_args
    s, w, codec = tiny_codec
    model = Model(csm_tiny_args(), None, max_frames=64, max_prefill_rows=128)
    gen = Generator(model, audio_tokenizer=codec)
    g = torch.Generator().manual_seed(1)
    ctx = [Segment(speaker=1, text=torch.randint(0, 1000, (5,), generator=g).tolist(),
                   audio_codes=torch.randint(0, 2048, (32, 6), generator=g))]
    text = torch.randint(0, 1000, (4,), generator=g).tolist()
    model.seed(3)
    audio = gen.generate(text, 1, ctx, max_audio_length_ms=960, temperature=0.9, topk=50)
    assert audio.dim() == 1 and audio.shape[0] == 12 * 1920 and audio.dtype == torch.float32
    assert torch.isfinite(audio).all() and gen.sample_rate == 24000
    model.seed(3)
    chunks = list(gen.generate_stream(text, 1, ctx, max_audio_length_ms=960, temperature=0.9, topk=50))
    assert [c.shape[0] for c in chunks] == [19200, 2 * 1920]
    # same seed -> same codes; first chunk of the stream == first 10 frames decoded statelessly
    first = codec.decode(gen._model.read_frames(1)[0][:10].permute(1, 2, 0).contiguous())
    assert torch.allclose(chunks[0], first.reshape(-1).to(chunks[0].device))
    # it streams: the first buffer is handed out while later frames are still being generated, and the chunks of a
    # seeded run equal the one-shot generation decoded in stateless 10-frame buffers
    model.seed(5)
    it = gen.generate_stream(text, 1, ctx, max_audio_length_ms=2400, temperature=0.9, topk=50)
    c0 = next(it)
    assert gen._model.num_frames() < 30, "generate_stream produced everything before yielding its first chunk"
    rest = list(it)
    assert c0.shape[0] == 19200 and [c.shape[0] for c in rest] == [19200, 19200]
    frames = gen._model.read_frames(1)[0]
    assert frames.shape[0] == 30
    want = codec.decode(frames[10:20].permute(1, 2, 0).contiguous()).reshape(-1)
    assert torch.allclose(rest[0], want.to(rest[0].device))
    with pytest.raises(ValueError, match="Inputs too long"):
        gen.generate(list(range(300)), 1, [], max_audio_length_ms=150_000)
    with pytest.raises(ValueError, match="Inputs too long"):
        next(gen.generate_stream(list(range(300)), 1, [], max_audio_length_ms=150_000))


def test_tts_service_cli_surface_end_to_end_tiny(tiny_codec, tmp_path):
    """The trimmed CLI class (sesameai-tts_amd/tts_service.py) with the reference's method names: voice prompt from a
    WAV file through the GPU Mimi encoder, context caching, sentence-wise export with silences/fades, 24 kHz int16 WAV."""
    import importlib.util
    import wave
    import numpy as np
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from sesameai.generator import Generator, load_llama3_tokenizer
    from dataclasses import replace
    from sesameai.mimi import MimiCodec, mimi_tiny_args
    from sesameai.models import FLAVORS, Model, ModelArgs
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tts_service_amd3", os.path.join(root, "sesameai-tts_amd", "tts_service.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    codec = MimiCodec(mimi_tiny_args(), None, max_frames=400)       # a segment may run its full 30 s = 375 frames
    words = "i'm getting all warmed up for our chatting to begin hello there how are you".replace("'", " ' ").split()
    vocab = {"[UNK]": 0, "<|begin_of_text|>": 1, "<|end_of_text|>": 2, "[": 3, "]": 4, "1": 5, ".": 6, "?": 7, "'": 8}
    for wd in words:
        vocab.setdefault(wd, len(vocab))
    tok = Tokenizer(WordLevel(vocab, unk_token="[UNK]")); tok.pre_tokenizer = Whitespace()
    tok.save(str(tmp_path / "tokenizer.json"))
    # a 0.4 s stereo 16 kHz prompt recording
    sr = 16000
    t = np.arange(int(0.4 * sr)) / sr
    pcm = (np.stack([0.3 * np.sin(2 * np.pi * 220 * t), 0.3 * np.sin(2 * np.pi * 330 * t)], 1) * 32767).astype("<i2")
    with wave.open(str(tmp_path / "prompt.wav"), "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(sr); f.writeframes(pcm.tobytes())
    (tmp_path / "samples.py").write_text(f"alice = {{{str(tmp_path / 'prompt.wav')!r}: 'hello there'}}\n")
    tts = mod.TTS(voice_dir=str(tmp_path))
    FLAVORS["llama-tiny-bb-2k"] = replace(FLAVORS["llama-tiny-bb"], max_seq_len=2048)       # the CLI assumes 2048 positions
    model = Model(ModelArgs("llama-tiny-bb-2k", "llama-tiny-dec", 1000, 2051, 32), None, max_frames=512, max_prefill_rows=128)
    tts.generator = Generator(model, audio_tokenizer=codec, text_tokenizer=load_llama3_tokenizer(str(tmp_path / "tokenizer.json")))
    assert tts.list_voices() == ["alice"]
    tts.load_voice("alice")                                  # encodes the WAV with Mimi, caches the context, warms up
    assert len(tts.cached_context_tokens) == 1 and tts.cached_context_tokens[0].shape[1] == 33
    n_audio_rows = int(tts.cached_context_masks[0][:, 0].sum())
    assert n_audio_rows == 5 + 1                             # ceil(0.4 s * 12.5) = 5 frames + the all-zero EOS frame
    out = str(tmp_path / "out.wav")
    tts.export_wav("hello there. how are you?", out, temperature=0.9, topk=20)
    with wave.open(out, "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate()) == (1, 2, 24000)
        n = f.getnframes()
        data = np.frombuffer(f.readframes(n), dtype="<i2")
    assert n >= 2 * int(0.6 * 24000)                         # two sentences, each with 500 ms lead + 100 ms tail silence
    assert int(np.abs(data[:1000]).max()) == 0 and int(np.abs(data).max()) > 1000


def test_from_pretrained_reads_a_moshi_format_safetensors_file(tmp_path):
    """MimiCodec.from_pretrained (reference: moshi's loaders.get_mimi on the hub file, sesameai/generator.py:340-344):
    a checkpoint written with moshi 0.2.2's tensor names loads through from_moshi_state_dict and decodes / encodes
    exactly like the codec built from the canonical dict."""
    from safetensors.torch import save_file
    from sesameai.mimi import MimiArgs, MimiCodec, synthetic_state_dict
    from test_host_logic import moshi_name
    s = MimiArgs()
    sd = synthetic_state_dict(s, seed=99)
    f = tmp_path / "tokenizer-test.safetensors"
    save_file({moshi_name(n, len(s.ratios)): t.contiguous() for n, t in sd.items()}, str(f))
    a = MimiCodec.from_pretrained(str(f), device="cuda", max_frames=40)
    b = MimiCodec(s, sd, device="cuda", max_frames=40)
    g = torch.Generator().manual_seed(2)
    codes = torch.randint(0, s.codebook_size, (1, s.num_codebooks, 25), generator=g).cuda()
    assert torch.equal(a.decode(codes), b.decode(codes))
    wav = (torch.randn(1, 1, 24000, generator=g) * 0.1).cuda()
    assert torch.equal(a.encode(wav), b.encode(wav))
