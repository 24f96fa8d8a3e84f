"""Host-side logic that needs no GPU: golden-vector checks of the oracle itself, prompt
layout (generator.py:63-109), sampler semantics, and the multi-GPU plumbing on gloo."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import csm_ref as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_prompt_layout_matches_golden_and_reference_rules():
    from oracle.make_golden import toy_prompt
    gold = torch.load(os.path.join(GOLD, "prompt_layout.pt"))
    tok, msk = toy_prompt(C.csm_tiny(), gold["seed"], gold["n_text"], gold["ctx_frames"])
    assert torch.equal(tok, gold["tokens"]) and torch.equal(msk, gold["mask"])
    n_text, T = gold["n_text"], gold["ctx_frames"]
    assert tok.shape == (n_text + T + 1 + n_text, 33)
    assert msk[:n_text, 32].all() and not msk[:n_text, :32].any()          # text rows: only column 32
    audio = slice(n_text, n_text + T + 1)
    assert msk[audio, :32].all() and not msk[audio, 32].any()               # audio rows: columns 0..31
    assert (tok[n_text + T] == 0).all()                                     # the appended all-zero EOS frame


def test_oracle_reproduces_committed_tiny_golden():
    """the golden file is a pure function of the oracle + seeds (guards against drift)."""
    from oracle.make_golden import toy_prompt
    gold = torch.load(os.path.join(GOLD, "tiny_frames.pt"))
    shape = C.csm_tiny()
    w = C.make_weights(shape, seed=int(gold["weight_seed"]))
    tok, msk = toy_prompt(shape, int(gold["prompt_seed"]), 6, 5)
    assert torch.equal(tok, gold["prompt_tokens"])
    m = C.OracleModel(shape, w); m.setup_caches(1)
    tr = C.FrameTrace()
    s = m.generate_frame(tok.unsqueeze(0), msk.unsqueeze(0), torch.arange(tok.size(0)).unsqueeze(0), 1.0, 1, greedy=True, trace=tr)
    assert torch.equal(s[0], gold["codes"][0])
    assert torch.equal(torch.stack(tr.logits, 0)[:, 0], gold["logits"][0])


def test_sampler_semantics_on_golden_cases():
    gold = torch.load(os.path.join(GOLD, "sampler_cases.pt"))
    logits = gold["logits"]
    for case in gold["cases"]:
        out = C.sample_topk(logits, case["topk"], case["temperature"], q=case["noise"], greedy_lowest_index=(case["topk"] == 1))
        assert torch.equal(out[:, 0], case["out"])
    # topk=1 == argmax with the lowest index among ties (row 3 has a 60-way tie at the top, row 4 is flat)
    g = C.sample_topk(logits, 1, 0.8, greedy_lowest_index=True)[:, 0]
    assert int(g[3]) == 100 and int(g[4]) == 0 and int(g[5]) == 7
    # ties at the kth value are all kept: row 6 carries 10 extra copies of its 50th-largest value
    t = logits / 0.9
    kth = torch.topk(t, 50)[0][..., -1, None]
    assert int((t[6] >= kth[6]).sum()) >= 60
    # EOS rule: a frame stops generation iff all 32 codes are zero (generator.py:285)
    assert bool(torch.all(torch.zeros(1, 32, dtype=torch.int32) == 0))


def test_generate_codes_stops_at_eos_and_guards_length():
    shape = C.csm_tiny()
    w = C.make_weights(shape)
    m = C.OracleModel(shape, w); m.setup_caches(1)
    tok, msk = C.build_prompt([([1, 2, 3], None)])
    with pytest.raises(ValueError, match="Inputs too long"):
        C.generate_codes(m, tok, msk, 80 * 254, 0.9, 50, max_seq_len=256)     # 3 >= 256 - 254
    calls = {"n": 0}
    orig = m.generate_frame

    def fake(*a, **k):
        calls["n"] += 1
        out = orig(*a, **k)
        return torch.zeros_like(out) if calls["n"] == 3 else out
    m.generate_frame = fake
    frames = C.generate_codes(m, tok, msk, 800, 0.9, 50, greedy=True, max_seq_len=256)
    assert len(frames) == 2 and calls["n"] == 3


def test_hf_checkpoint_conversion_round_trip():
    """a transformers-format CSM checkpoint (half-split RoPE rows) converts back to the reference layout."""
    pytest.importorskip("transformers.models.csm.modeling_csm")
    import warnings
    from oracle.hf_map import build_hf_csm
    from sesameai.models import csm_tiny_args, from_hf_state_dict
    shape = C.csm_tiny()
    w = {k: v.float() for k, v in C.make_weights(shape, norm_jitter=0.1).items()}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bb, dd, heads = build_hf_csm(shape, w)
    hf = {f"backbone_model.{k}": v for k, v in bb.state_dict().items()}
    hf.update({f"depth_decoder.model.{k}": v for k, v in dd.state_dict().items()})
    hf["embed_text_tokens.weight"] = w["text_embeddings.weight"]
    hf["lm_head.weight"] = w["codebook0_head.weight"]
    hf["depth_decoder.codebooks_head.weight"] = heads
    back = from_hf_state_dict(csm_tiny_args(), hf)
    for k, v in w.items():
        assert torch.equal(back[k].float(), v), k


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "sesameai-tts_amd"))
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=2)
import importlib.util
spec = importlib.util.spec_from_file_location("par", os.path.join(sys.argv[1], "sesameai-tts_amd", "sesameai", "parallel.py"),
                                              submodule_search_locations=None)
# parallel.py imports .models -> ._abi (needs the built library, no GPU); import it as a package member
import sesameai.parallel as par
from sesameai.models import csm_tiny_args, synthetic_state_dict
args = csm_tiny_args()
rank = dist.get_rank()
sd = synthetic_state_dict(args, seed=1234) if rank == 0 else None
got = par.broadcast_state_dict(args, sd, torch.device("cpu"))
ref = synthetic_state_dict(args, seed=1234)
ok = all(torch.equal(got[k], ref[k]) for k in ref)
sh = [list(par.shard_utterances(10, 4, r)) for r in range(4)]
ok = ok and sum(sh, []) == list(range(10)) and sh[0] == [0, 1, 2]
dist.barrier(); dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def test_weight_broadcast_and_sharding_world_size_2_gloo(tmp_path):
    """N>1 path on CPU: one flat-blob broadcast from rank 0, every rank ends with identical
    weights; utterances shard contiguously with no further communication."""
    lib = os.path.join(ROOT, "sesameai-tts_amd", "lib", "libcsm_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r)]) for r in range(2)]
    rcs = [p.wait(timeout=180) for p in procs]
    assert rcs == [0, 0]


def test_wav_loading_mono_and_resample(tmp_path):
    """tts_service.load_audio: stereo int16 44.1 kHz -> mono float 24 kHz (reference _load_audio, tts_service.py:141-168)."""
    import importlib.util
    import wave
    import numpy as np
    spec = importlib.util.spec_from_file_location("tts_service_amd", os.path.join(ROOT, "sesameai-tts_amd", "tts_service.py"))
    lib = os.path.join(ROOT, "sesameai-tts_amd", "lib", "libcsm_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    sr, secs = 44100, 0.5
    t = np.arange(int(sr * secs)) / sr
    left, right = 0.5 * np.sin(2 * np.pi * 440 * t), 0.25 * np.sin(2 * np.pi * 440 * t)
    pcm = (np.stack([left, right], 1) * 32767).astype("<i2")
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(sr); f.writeframes(pcm.tobytes())
    x = mod.load_audio(path, 24000)
    assert x.dim() == 1 and abs(x.shape[0] - 12000) <= 1 and x.dtype == torch.float32
    want = 0.375 * np.sin(2 * np.pi * 440 * np.arange(x.shape[0]) / 24000)
    assert np.abs(x.numpy()[200:-200] - want[200:-200]).max() < 5e-3


def test_watermark_hook_passes_audio_through_without_silentcipher():
    """reference tts_service.py:23 imports these names from sesameai.watermarking; without the third-party model the
    hook must be a no-op, not an ImportError."""
    from sesameai.watermarking import CSM_1B_GH_WATERMARK, load_watermarker, verify, watermark
    wm = load_watermarker("cpu")
    x = torch.linspace(-1, 1, 2400)
    if wm is None:
        y, sr = watermark(wm, x, 24000, CSM_1B_GH_WATERMARK)
        assert sr == 24000 and torch.equal(x, y) and verify(wm, y, sr, CSM_1B_GH_WATERMARK) is False


def test_voice_registry_discovers_pt_prompts_and_samples_module(tmp_path):
    """reference tts_service.py:36-42 builds its voice list from the dicts of `samples.py`; ours also accepts
    pre-tokenised <voice>.pt prompt files."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("tts_service_amd2", os.path.join(ROOT, "sesameai-tts_amd", "tts_service.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    (tmp_path / "samples.py").write_text("alice = {'/tmp/a.wav': 'hello there'}\n_private = 3\n")
    torch.save([("hi", torch.zeros(32, 4, dtype=torch.long))], tmp_path / "bob.pt")
    voices = mod.discover_voices(str(tmp_path))
    assert set(voices) == {"alice", "bob"}
    assert voices["alice"] == {"/tmp/a.wav": "hello there"} and voices["bob"].endswith("bob.pt")
    tts = mod.TTS(voice_dir=str(tmp_path))
    assert tts.list_voices() == ["bob", "alice"] or set(tts.list_voices()) == {"alice", "bob"}
    with pytest.raises(ValueError, match="not found"):
        tts.load_voice("carol")


def test_prompt_assembly_properties_hypothesis():
    """for arbitrary segments the (S,33) prompt keeps the reference's layout rules
    (sesameai/generator.py:63-109): text rows carry the id in column 32 only; audio rows carry 32 codes in columns
    0..31 only; every audio segment ends with an all-zero EOS frame; rows appear in segment order."""
    from hypothesis import given, settings, strategies as st
    from sesameai.generator import Generator, Segment

    gen = Generator.__new__(Generator)
    gen.device, gen._text_tokenizer, gen._audio_tokenizer = torch.device("cpu"), None, None

    seg = st.tuples(st.lists(st.integers(0, 128255), min_size=1, max_size=12), st.integers(1, 9), st.integers(0, 2 ** 31 - 1))

    @settings(max_examples=40, deadline=None)
    @given(st.lists(seg, min_size=0, max_size=4), st.lists(st.integers(0, 128255), min_size=1, max_size=10))
    def check(segments, text):
        ctx, want_rows = [], 0
        for ids, frames, seed in segments:
            codes = torch.randint(0, 2048, (32, frames), generator=torch.Generator().manual_seed(seed))
            ctx.append(Segment(speaker=0, text=ids, audio_codes=codes))
            want_rows += len(ids) + frames + 1
        tok, msk = gen._build_prompt(text, 0, ctx)
        assert tok.shape == msk.shape == (want_rows + len(text), 33) and tok.dtype == torch.long and msk.dtype == torch.bool
        row = 0
        for (ids, frames, seed), s in zip(segments, ctx):
            t = tok[row:row + len(ids)]; m = msk[row:row + len(ids)]
            assert t[:, 32].tolist() == ids and bool(m[:, 32].all()) and not bool(m[:, :32].any()) and int(t[:, :32].abs().sum()) == 0
            row += len(ids)
            a = tok[row:row + frames + 1]; am = msk[row:row + frames + 1]
            assert torch.equal(a[:frames, :32], s.audio_codes.t()) and int(a[frames].abs().sum()) == 0      # EOS frame
            assert bool(am[:, :32].all()) and not bool(am[:, 32].any()) and int(a[:, 32].abs().sum()) == 0
            row += frames + 1
        assert tok[row:, 32].tolist() == text and bool(msk[row:, 32].all()) and not bool(msk[row:, :32].any())

    check()


def test_local_tokenizer_json_gets_the_reference_bos_eos_template(tmp_path):
    """reference generator.py:24-38: the Llama-3 tokenizer with a TemplateProcessing that wraps every text in
    <|begin_of_text|> ... <|end_of_text|>; `_tokenize_text_segment` encodes "[speaker]text" (generator.py:67).
    No hub access here, so a small WordLevel tokenizer.json with the same special tokens stands in for the real file."""
    from tokenizers import Tokenizer
    from tokenizers.models import WordLevel
    from tokenizers.pre_tokenizers import Whitespace
    from sesameai.generator import Generator, load_llama3_tokenizer
    vocab = {"[UNK]": 0, "<|begin_of_text|>": 1, "<|end_of_text|>": 2, "[": 3, "]": 4, "1": 5, "hello": 6, "there": 7}
    tok = Tokenizer(WordLevel(vocab, unk_token="[UNK]")); tok.pre_tokenizer = Whitespace()
    path = str(tmp_path / "tokenizer.json"); tok.save(path)
    assert load_llama3_tokenizer(str(tmp_path / "missing.json")) is None
    t = load_llama3_tokenizer(path)
    assert t.encode("hello there").ids == [1, 6, 7, 2]
    gen = Generator.__new__(Generator)
    gen.device, gen._text_tokenizer, gen._audio_tokenizer = torch.device("cpu"), t, None
    frame, mask = gen._tokenize_text_segment("hello there", 1)
    assert frame[:, 32].tolist() == [1, 3, 5, 4, 6, 7, 2] and bool(mask[:, 32].all()) and not bool(mask[:, :32].any())
    gen._text_tokenizer = None
    with pytest.raises(RuntimeError, match="no text tokenizer"):
        gen._tokenize_text_segment("hello", 1)
