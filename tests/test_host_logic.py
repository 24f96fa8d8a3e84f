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
stats = {}
got = par.broadcast_state_dict(args, sd, torch.device("cpu"), stats=stats)
ref = synthetic_state_dict(args, seed=1234)
ok = all(torch.equal(got[k], ref[k]) for k in ref) and stats["bytes"] > 0 and stats["ms"] > 0
# a second blob whose names / shapes every rank knows (the codec's weights in bench.py): values come from rank 0 only
named = {"a.w": torch.arange(70, dtype=torch.float32).view(7, 10), "b": torch.full((3,), 2.5)}
tmpl = {k: torch.zeros_like(v) for k, v in named.items()}
got2 = par.broadcast_named(named if rank == 0 else None, torch.device("cpu"), template=None if rank == 0 else tmpl)
ok = ok and all(torch.equal(got2[k], named[k]) for k in named)
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


# ----------------------------------------------------------------------------------------
# the frame loop's stop rule (reference: sesameai/generator.py:285 / tts_service.py:228, `torch.all(sample == 0)` -> break)
# on the PRODUCT Generator, driven by a scripted stand-in for the device model
# ----------------------------------------------------------------------------------------
class _ScriptedModel:
    """Stands in for sesameai.models.Model: hands out scripted frames and raises the device-side EOS flag (first
    all-zero frame per sequence) only once that frame has been 'launched', exactly like k_advance."""

    def __init__(self, frames: torch.Tensor):
        self.frames = frames.to(torch.int32)            # [N][B][32]
        self.device = torch.device("cpu")
        self.launched = 0
        self.steps_after_eos = 0

    def setup_caches(self, b): pass
    def reset_caches(self): self.launched = 0
    def prefill_prompt(self, t, m): return t.shape[1]
    def depth(self, B, T, k, commit=True): self.launched += 1
    def step(self, B, T, k, use_graph=True): self.launched += 1
    def num_frames(self): return self.launched

    def read_frames(self, B, first=0, n=None):
        n = self.launched - first if n is None else n
        fr = self.frames[first:first + n, :B].clone()
        zero = (self.frames[:self.launched, :B] == 0).all(dim=2)                     # [launched][B]
        eos = torch.full((B,), -1, dtype=torch.int32)
        for b in range(B):
            idx = torch.nonzero(zero[:, b])
            if idx.numel():
                eos[b] = int(idx[0])
        return fr, eos


class _FakeCodec:
    sample_rate = 24_000
    def decode(self, codes):                                   # (B, 32, T) -> (B, 1, 1920 T), value = frame's first code
        return codes[:, :1, :].float().repeat_interleave(1920, dim=2)


def _scripted(n_total, B, eos_at):
    g = torch.Generator().manual_seed(7)
    fr = torch.randint(1, 2048, (n_total, B, 32), generator=g)
    for b, k in enumerate(eos_at):
        if k is not None:
            fr[k, b] = 0
    return fr


@pytest.mark.parametrize("k", [0, 1, 7, 8, 9, 15, 16, 23, 39])
def test_generator_cuts_at_the_eos_frame_wherever_it_falls_in_a_poll_block(k):
    """EOS at a block start (8, 16), mid-block, in the prompt frame (0) and in the last allowed frame (39): the frames
    handed out are exactly those BEFORE the all-zero frame, like the reference's loop (append comes after the break)."""
    from sesameai.generator import Generator
    script = _scripted(64, 1, [k])
    gen = Generator(_ScriptedModel(script), audio_tokenizer=_FakeCodec())
    prompt = torch.zeros(5, 33, dtype=torch.long); mask = torch.zeros(5, 33, dtype=torch.bool)
    seen = []
    frames = gen.generate_codes(prompt, mask, 40, 0.9, 50, on_frames=lambda f: seen.append(f.shape[0]))
    assert frames.shape == (k, 1, 32) and sum(seen) == k
    assert torch.equal(frames, script[:k].to(torch.int32))
    assert int(gen.last_eos_at[0]) == k
    assert gen._model.launched <= ((k // 8) + 2) * 8 + 1          # at most the block holding EOS + the one already enqueued


def test_generator_runs_to_the_length_limit_without_eos_and_streams_the_same_frames():
    from sesameai.generator import Generator
    script = _scripted(64, 1, [None])
    gen = Generator(_ScriptedModel(script), audio_tokenizer=_FakeCodec())
    prompt = torch.zeros(5, 33, dtype=torch.long); mask = torch.zeros(5, 33, dtype=torch.bool)
    frames = gen.generate_codes(prompt, mask, 25, 0.9, 50)
    assert frames.shape == (25, 1, 32) and int(gen.last_eos_at[0]) == -1
    # generate_stream: 10-frame buffers, the tail buffer shorter; EOS at 23 -> 10 + 10 + 3 frames of audio
    script = _scripted(64, 1, [23])
    gen = Generator(_ScriptedModel(script), audio_tokenizer=_FakeCodec())
    chunks = list(gen.generate_stream([1, 2, 3], 0, [], max_audio_length_ms=40 * 80))
    assert [c.shape[0] for c in chunks] == [19200, 19200, 3 * 1920]
    audio = torch.cat(chunks)
    assert torch.equal(audio[::1920], script[:23, 0, 0].float())          # every frame once, in order, none after EOS
    assert gen.generate([1, 2, 3], 0, [], max_audio_length_ms=40 * 80).shape[0] == 23 * 1920
    # EOS in the very first frame: empty audio, like the reference (generator.py:296-297)
    gen = Generator(_ScriptedModel(_scripted(64, 1, [0])), audio_tokenizer=_FakeCodec())
    assert gen.generate([1, 2, 3], 0, [], max_audio_length_ms=40 * 80).numel() == 0
    assert list(gen.generate_stream([1, 2, 3], 0, [], max_audio_length_ms=40 * 80)) == []


def test_first_stream_chunk_is_decoded_before_more_frames_are_queued_and_the_next_block_before_the_user_gets_it():
    """Time to first audio: the first block is exactly one buffer (frame 0 + 9 steps), nothing else is queued while it is decoded (the
    decode would otherwise squeeze between frame steps whose persistent launches hold every CU), and the second block is queued before
    the first chunk reaches the user (a user who plays each chunk before asking for the next must not starve the GPU)."""
    from sesameai.generator import Generator
    script = _scripted(64, 1, [None])
    model = _ScriptedModel(script)
    events = []
    codec = _FakeCodec()
    dec = codec.decode
    codec.decode = lambda codes: (events.append(("decode", model.launched)), dec(codes))[1]
    gen = Generator(model, audio_tokenizer=codec)
    launched_when_user_got_chunk = []
    for chunk in gen.generate_stream([1, 2, 3], 0, [], max_audio_length_ms=35 * 80):
        launched_when_user_got_chunk.append(model.launched)
    assert events[0] == ("decode", 10)                        # first decode: exactly 10 frames launched, none beyond
    assert launched_when_user_got_chunk[0] == 20              # ... and the next block was queued before the user saw the chunk
    assert events[1] == ("decode", 30)                        # steady state: the next block is queued before the current one is decoded
    assert [e[0] for e in events] == ["decode"] * 4 and launched_when_user_got_chunk[-1] == 35
    # generate_codes (no decode between blocks): same frames, every block is followed at once by the next
    model2 = _ScriptedModel(script)
    gen2 = Generator(model2, audio_tokenizer=_FakeCodec())
    seen = []
    prompt = torch.zeros(5, 33, dtype=torch.long); mask = torch.zeros(5, 33, dtype=torch.bool)
    frames = gen2.generate_codes(prompt, mask, 35, 0.9, 50, on_frames=lambda f: seen.append((f.shape[0], model2.launched)), poll=10)
    assert frames.shape[0] == 35 and seen[0] == (10, 20) and seen[1] == (10, 30)      # no deferral without a stream consumer
    assert not hasattr(gen2, "_release_first_block")                                # the gate travels with the stream, not on the Generator


class _ScriptedSlots:
    """Stands in for sesameai.models.Model under Generator.generate_codes_continuous: every prompt (identified by its first text
    token) has a scripted utterance; slots emit their utterance's next frame per step, all-zero frames once it is over, and the
    EOS word of a slot is the GLOBAL index of its first all-zero frame since its last refill -- the contract of csm_prefill_slot /
    k_advance (include/csm_hip.h)."""

    def __init__(self, scripts, max_batch):
        self.scripts, self.device, self._max_batch = scripts, torch.device("cpu"), max_batch
        self.refills, self.resets = [], []

    def setup_caches(self, b): pass

    def reset_caches(self):
        self.hist, self.slot, self.eos = [], {}, {}

    def num_frames(self): return max(len(self.hist), 1) if self.slot else 0

    def refill_slot(self, slot, tokens, mask, T, k):
        pid = int(tokens[0, 32])
        self.refills.append((slot, pid, self.num_frames()))
        self.slot[slot] = [pid, 1]
        f0 = self.scripts[pid][0].to(torch.int32)
        if not self.hist:
            self.hist.append({})
        self.hist[-1][slot] = f0
        self.eos[slot] = len(self.hist) - 1 if bool((f0 == 0).all()) else -1
        return f0

    def reset_slots(self, slots): self.resets.append(list(slots))

    def step(self, B, T, k, use_graph=True):
        row = {}
        for s_ in range(B):
            pid, cur = self.slot[s_]
            sc = self.scripts[pid]
            f = sc[cur].to(torch.int32) if cur < sc.shape[0] else torch.zeros(32, dtype=torch.int32)
            self.slot[s_][1] = cur + 1
            row[s_] = f
            if self.eos[s_] < 0 and bool((f == 0).all()):
                self.eos[s_] = len(self.hist)
        self.hist.append(row)

    def read_frames(self, B, first=0, n=None):
        fr = torch.stack([torch.stack([self.hist[g][s_] for s_ in range(B)]) for g in range(first, first + n)]) if n else torch.empty(0, B, 32, dtype=torch.int32)
        return fr, torch.tensor([self.eos[s_] for s_ in range(B)], dtype=torch.int32)


def test_continuous_batching_retires_at_eos_and_refills_the_slot():
    """7 utterances of different lengths (EOS after 0, 3, 5, 11, 17 frames, one that runs into the length limit, one that ends
    exactly on a poll boundary) through 3 slots: every utterance comes back complete, cut at ITS all-zero frame like the
    reference's batch-1 loop (generator.py:285), in prompt order; slots are refilled as they free up and retired slots that
    have nothing left to do are rewound."""
    from sesameai.generator import Generator
    g = torch.Generator().manual_seed(3)
    lens = [3, 17, 0, 5, 40, 11, 8]
    scripts = []
    for n in lens:
        sc = torch.randint(1, 2048, (n + 1, 32), generator=g)
        sc[n] = 0                                               # the all-zero EOS frame
        scripts.append(sc)
    model = _ScriptedSlots(scripts, 3)
    gen = Generator(model, audio_tokenizer=_FakeCodec(), max_batch_size=3)
    prompts = []
    for i in range(len(lens)):
        t = torch.zeros(4 + i, 33, dtype=torch.long); t[:, 32] = i
        prompts.append((t, torch.zeros(4 + i, 33, dtype=torch.bool)))
    out = gen.generate_codes_continuous(prompts, 25, 0.9, 50)
    assert len(out) == len(lens)
    for i, n in enumerate(lens):
        want = scripts[i][: min(n, 25)].to(torch.int32)
        assert out[i].shape == want.shape and torch.equal(out[i], want), f"utterance {i} (length {n})"
    assert sorted(pid for _, pid, _ in model.refills) == list(range(len(lens)))          # every prompt was started exactly once
    assert len({slot for slot, _, _ in model.refills}) == 3 and model.resets                # three slots in use; idle ones rewound
    with pytest.raises(ValueError):
        gen.generate_codes_continuous([(torch.zeros(2030, 33, dtype=torch.long), torch.zeros(2030, 33, dtype=torch.bool))], 25, 0.9, 50)


class _ScriptedSlotsBeside:
    """Stands in for sesameai.models.Model under Generator._iter_codes_refilling_beside_the_loop (the contract of csm_refill_begin /
    csm_refill_advance / k_advance's fresh flag, include/csm_hip.h): a prompt needs 16 layers of refill work, handed out a few per call
    between frame steps; until it is complete the slot emits placeholder frames (all 7); the step after completion emits the
    utterance's frame 0 and restarts the slot's EOS word; a retired slot emits placeholders too."""

    class _BB:
        num_layers = 16

    def __init__(self, scripts, max_batch):
        self.scripts, self.device, self._max_batch, self.bb = scripts, torch.device("cpu"), max_batch, self._BB()
        self.advance_calls, self.calls_between_steps, self.resets = [], [], []

    def setup_caches(self, b): pass
    def supports_refill_beside_the_loop(self, batch=None): return True

    def reset_caches(self):
        self.hist, self.cur, self.eos, self.pending, self._since_step = [], {}, {}, None, []

    def num_frames(self): return len(self.hist)

    def refill_begin(self, slot, tokens, mask):
        assert self.pending is None, "one refill at a time"
        self.pending = [slot, int(tokens[0, 32]), 16]
        self.cur[slot] = None                                    # parked

    def refill_advance(self, k):
        self.advance_calls.append(k); self._since_step.append(k)
        self.pending[2] -= k
        if self.pending[2] > 0:
            return False
        slot, pid, _ = self.pending
        self.cur[slot], self.pending = [pid, 0, True], None       # fresh: the next step emits frame 0
        return True

    def reset_slots(self, slots): self.resets.append(list(slots))

    def step(self, B, T, k, use_graph=True):
        self.calls_between_steps.append(self._since_step); self._since_step = []
        row = {}
        for s_ in range(B):
            c = self.cur.get(s_)
            if c is None:
                row[s_] = torch.full((32,), 7, dtype=torch.int32)
                continue
            pid, i, fresh = c
            sc = self.scripts[pid]
            f = sc[i].to(torch.int32) if i < sc.shape[0] else torch.zeros(32, dtype=torch.int32)
            if fresh:
                self.eos[s_] = -1
            if self.eos.get(s_, -1) < 0 and bool((f == 0).all()):
                self.eos[s_] = len(self.hist)
            c[1], c[2] = i + 1, False
            row[s_] = f
        self.hist.append(row)

    def read_frames(self, B, first=0, n=None):
        fr = torch.stack([torch.stack([self.hist[g][s_] for s_ in range(B)]) for g in range(first, first + n)])
        return fr, torch.tensor([self.eos.get(s_, -1) for s_ in range(B)], dtype=torch.int32)


def test_continuous_batching_refills_beside_the_loop_a_few_layers_per_step():
    """Round 4: a retired slot's prompt runs a few backbone layers after each frame step instead of stalling the batch (csm_refill_*).
    9 utterances (EOS after 0..40 frames, one cut by the length limit) through 3 slots: every one comes back complete and cut at
    ITS EOS; while anybody generates, at most ONE bounded refill call sits between two frame steps (the initial fill and the
    all-idle case run whole prompts); placeholder frames of parked / retired slots never leak into a result."""
    from sesameai.generator import Generator
    g = torch.Generator().manual_seed(9)
    lens = [3, 17, 0, 5, 40, 11, 8, 1, 22]
    scripts = []
    for n in lens:
        sc = torch.randint(8, 2048, (n + 1, 32), generator=g); sc[n] = 0
        scripts.append(sc)
    model = _ScriptedSlotsBeside(scripts, 3)
    gen = Generator(model, audio_tokenizer=_FakeCodec(), max_batch_size=3)
    gen.refill_row_layers = 20                                        # 5-row prompts: 4 layers per call -> 4 calls per prompt
    prompts = []
    for i in range(len(lens)):
        t = torch.zeros(5, 33, dtype=torch.long); t[:, 32] = i
        prompts.append((t, torch.zeros(5, 33, dtype=torch.bool)))
    limits = [25, 25, 25, 25, 25, 25, 25, 25, 10]                     # one length limit per request (the last one cuts a 22-frame utterance at 10)
    out = gen.generate_codes_continuous(prompts, limits, 0.9, 50)
    for i, n in enumerate(lens):
        want = scripts[i][: min(n, limits[i])].to(torch.int32)
        assert out[i].shape == want.shape and torch.equal(out[i], want), f"utterance {i} (length {n})"
        assert not bool((out[i] == 7).all(dim=-1).any()), "a placeholder frame leaked into a result"
    for calls in model.calls_between_steps[1:]:                       # refill calls between two frame steps: ONE bounded piece while anybody generates,
        assert sum(k < 16 for k in calls) <= 1, calls                   # (whole prompts, 16 layers per call, only once nobody does)
    assert 4 in model.advance_calls and 16 in model.advance_calls      # bounded pieces beside the loop, whole prompts in the initial fill
    assert model.resets, "retired slots must be rewound while they wait"


def test_continuous_batching_hands_out_each_utterance_as_it_finishes_and_runs_past_the_history_size():
    """results used to be returned only at the end and the total number of frame steps was capped by the engine's
    linear frame history.  ``iter_codes_continuous`` yields (index, frames) when an utterance retires -- the short ones long
    before the long one ends -- and the number of global frame steps may exceed any history size (the engine's history is a
    ring read block by block: here 12 rounds x 30 frames through one slot = 360 global frames)."""
    from sesameai.generator import Generator
    g = torch.Generator().manual_seed(5)
    lens = [30] * 12 + [2, 3]
    scripts = []
    for n in lens:
        sc = torch.randint(1, 2048, (n + 1, 32), generator=g); sc[n] = 0
        scripts.append(sc)
    model = _ScriptedSlots(scripts, 2)
    gen = Generator(model, audio_tokenizer=_FakeCodec(), max_batch_size=2)
    prompts = []
    for i in range(len(lens)):
        t = torch.zeros(3, 33, dtype=torch.long); t[:, 32] = i
        prompts.append((t, torch.zeros(3, 33, dtype=torch.bool)))
    order, steps_when_done = [], {}
    for i, frames in gen.iter_codes_continuous(prompts, 40, 0.9, 50):
        order.append(i); steps_when_done[i] = len(model.hist)
        assert torch.equal(frames, scripts[i][: lens[i]].to(torch.int32)), f"utterance {i}"
    assert sorted(order) == list(range(len(lens)))
    assert len(model.hist) > 150                                     # far more global frame steps than one utterance holds
    assert steps_when_done[order[0]] < steps_when_done[order[-1]] - 100, "results must come out as the run goes, not at its end"


def test_two_interleaved_streams_do_not_release_each_others_blocks_and_an_abandoned_stream_leaves_nothing_behind():
    """the first-block hand-shake was per-Generator state.  It is now an object owned by each stream."""
    from sesameai.generator import Generator
    script = _scripted(64, 1, [None])
    model = _ScriptedModel(script)
    gen = Generator(model, audio_tokenizer=_FakeCodec())
    s1 = gen.generate_stream([1, 2, 3], 0, [], max_audio_length_ms=30 * 80)
    first = next(s1)
    assert first.shape[0] == 19200 and model.launched == 20
    s1.close()                                                       # abandoned after one chunk
    assert not any(k.startswith("_release") for k in vars(gen))      # nothing installed on the Generator
    launched_before = model.launched
    # a fresh stream on the same Generator is unaffected by the abandoned one
    chunks = list(gen.generate_stream([1, 2, 3], 0, [], max_audio_length_ms=30 * 80))
    assert [c.shape[0] for c in chunks] == [19200] * 3 and model.launched == 30 != launched_before


def test_audio_stream_writer_appends_incrementally_and_patches_the_sizes_on_close(tmp_path):
    import struct
    import numpy as np
    from sesameai.generator import AudioStreamWriter
    f = tmp_path / "inc.wav"
    w = AudioStreamWriter(str(f), 24_000)
    w.add_chunk(torch.arange(100, dtype=torch.float32))
    assert f.exists() and w.chunks_written == 1 and w.audio_chunks == []               # on disk already, no list of tensors kept
    w._file.flush()
    raw = f.read_bytes()                                             # while open: "length unknown" placeholders, a WAV players read to its end
    assert struct.unpack("<I", raw[4:8])[0] == 0xFFFFFFFF and struct.unpack("<I", raw[40:44])[0] == 0xFFFFFFFF and len(raw) == 444
    w.add_chunk(torch.arange(100, 150, dtype=torch.bfloat16))
    w.write_file()
    raw = f.read_bytes()
    assert struct.unpack("<I", raw[4:8])[0] == 36 + 600 and struct.unpack("<I", raw[40:44])[0] == 600 and len(raw) == 644
    assert np.array_equal(np.frombuffer(raw[44:], dtype="<f4"), np.arange(150, dtype=np.float32))
    w.write_file()                                                   # idempotent
    # a writer that is dropped without write_file: finalised by the context manager / the destructor, handle closed
    g = tmp_path / "dropped.wav"
    with AudioStreamWriter(str(g), 24_000) as w2:
        w2.add_chunk(torch.ones(10))
    assert struct.unpack("<I", g.read_bytes()[40:44])[0] == 40 and w2._file is None
    h = tmp_path / "collected.wav"
    w3 = AudioStreamWriter(str(h), 24_000)
    w3.add_chunk(torch.ones(7))
    del w3
    import gc
    gc.collect()
    assert struct.unpack("<I", h.read_bytes()[40:44])[0] == 28


def test_rope_table_fp32_tensor_form_vs_the_double_form_over_the_positions_each_stack_reads():
    """torchtune 0.4.0's ``Llama3ScaledRoPE.apply_scaling`` iterates over the elements of an fp32 TENSOR (``for freq in freqs``:
    wavelength, smoothing and the scaled frequency are 0-dim fp32 tensor arithmetic); the oracle (oracle/csm_ref.py:158-175)
    and the product (sesameai/models.py llama3_rope_table) iterate over Python doubles.  Both end as the bf16 table the
    model rounds at ``model.to(bf16)`` (generator.py:343).  The two forms must give the same bf16 table wherever a stack reads
    it: every one of the backbone's 2048 positions (head_dim 64) and the depth decoder's 32 positions (head_dim 128).  Beyond
    those, the head_dim-128 table may differ in a handful of entries (one bf16 ulp, positions the 32-position decoder never
    reaches); the count is printed, not hidden."""
    import math
    from oracle import csm_ref as C
    from sesameai.models import FLAVORS, llama3_rope_table

    def torchtune_form(head_dim, base=500_000.0, scale_factor=32.0, low=1.0, high=4.0, old_len=8192, max_seq=2048):
        freqs = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
        low_wl, high_wl = old_len / low, old_len / high
        new = []
        for freq in freqs:                                        # 0-dim fp32 tensors, as in torchtune
            wavelen = 2 * math.pi / freq
            if wavelen < high_wl:
                new.append(freq)
            elif wavelen > low_wl:
                new.append(freq / scale_factor)
            else:
                smooth = (old_len / wavelen - low) / (high - low)
                new.append((1 - smooth) * freq / scale_factor + smooth * freq)
        theta = torch.tensor(new, dtype=freqs.dtype)
        idx = torch.einsum("i, j -> ij", torch.arange(max_seq, dtype=theta.dtype), theta).float()
        return torch.stack([torch.cos(idx), torch.sin(idx)], dim=-1).to(torch.bfloat16)

    for flavor, shape, reads in (("llama-1B", C.csm_1b().backbone, 2048), ("llama-100M", C.csm_1b().decoder, 32)):
        tt = torchtune_form(FLAVORS[flavor].head_dim)
        oracle = C.rope_table(shape)
        product = llama3_rope_table(FLAVORS[flavor])
        assert torch.equal(oracle, product), f"{flavor}: oracle and product tables differ"
        diff = (tt.view(torch.int16) != oracle.view(torch.int16))
        n_all = int(diff.sum())
        first_pos = int(diff.any(dim=2).any(dim=1).nonzero()[0]) if n_all else None
        print(f"{flavor} (head_dim {FLAVORS[flavor].head_dim}): {n_all} of {diff.numel()} table entries differ between the fp32-tensor and the double form"
              + (f", first at position {first_pos}" if n_all else "") + f"; the stack reads positions 0..{reads - 1}")
        assert not bool(diff[:reads].any()), f"{flavor}: the two forms differ inside the positions the stack reads"
        assert n_all <= 8


def test_generate_streaming_audio_writes_every_chunk_to_one_file(tmp_path, capsys):
    """reference: generate_streaming_audio / AudioStreamWriter (sesameai/generator.py:303-327,349-434): every streamed chunk is
    handed to the writer as it is produced, the file holds all of them in order (mono float32 WAV, what torchaudio.save
    writes for a float tensor), playback without `sounddevice` is switched off with the reference's message."""
    import struct
    import numpy as np
    from sesameai.generator import AudioStreamWriter, Generator, generate_streaming_audio
    script = _scripted(64, 1, [23])
    gen = Generator(_ScriptedModel(script), audio_tokenizer=_FakeCodec())
    out = str(tmp_path / "stream.wav")
    generate_streaming_audio(gen, [1, 2, 3], 0, [], out, max_audio_length_ms=40 * 80, play_audio=True)
    printed = capsys.readouterr().out
    assert "Generated chunk 3" in printed and "Generated chunk 4" not in printed and "Audio generation completed" in printed
    try:
        import sounddevice  # noqa: F401
    except ImportError:
        assert "sounddevice library not found" in printed
    raw = open(out, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt "
    fmt, ch, sr, _, _, bits = struct.unpack("<HHIIHH", raw[20:36])
    assert (fmt, ch, sr, bits) == (3, 1, 24_000, 32)
    pcm = np.frombuffer(raw[44:], dtype="<f4")
    assert pcm.shape[0] == 23 * 1920 and np.array_equal(pcm[::1920], script[:23, 0, 0].float().numpy())
    # the writer alone: thread-safe appends, nothing written when nothing was generated
    w = AudioStreamWriter(str(tmp_path / "empty.wav"), 24_000)
    w.write_file()
    assert not (tmp_path / "empty.wav").exists()


def test_generator_batch_stops_when_every_sequence_has_hit_eos():
    """B = 3 with different stop frames: the loop runs until the LAST sequence's EOS (or the limit) and reports each
    sequence's own stop frame; the caller trims with last_eos_at."""
    from sesameai.generator import Generator
    script = _scripted(64, 3, [5, 19, 12])
    model = _ScriptedModel(script)
    gen = Generator(model, audio_tokenizer=_FakeCodec(), max_batch_size=3)
    prompt = torch.zeros(3, 5, 33, dtype=torch.long); mask = torch.zeros(3, 5, 33, dtype=torch.bool)
    frames = gen.generate_codes(prompt, mask, 60, 0.9, 50)
    assert gen.last_eos_at.tolist() == [5, 19, 12]
    assert 20 <= frames.shape[0] <= 32 and model.launched <= 33              # stopped within a poll block of frame 19
    assert torch.equal(frames[:20], script[:20].to(torch.int32))
    # one sequence never stops -> the length limit ends the batch
    gen = Generator(_ScriptedModel(_scripted(64, 3, [5, None, 12])), audio_tokenizer=_FakeCodec(), max_batch_size=3)
    frames = gen.generate_codes(prompt, mask, 30, 0.9, 50)
    assert frames.shape[0] == 30 and gen.last_eos_at.tolist() == [5, -1, 12]


def moshi_name(n: str, ns: int) -> str:
    """canonical tensor name (sesameai.mimi.state_dict_layout) -> its name in a moshi 0.2.2 Mimi checkpoint, written out from
    moshi's module tree (ns = number of SEANet stages)"""
    p = n.split(".")
    if p[0] == "rvq":
        k = int(p[1]); base = "quantizer.rvq_first.vq.layers.0" if k == 0 else f"quantizer.rvq_rest.vq.layers.{k - 1}"
        return f"{base}._codebook.{p[2]}"
    if n in ("rvq_first.output_proj.weight", "rvq_rest.output_proj.weight", "rvq_first.input_proj.weight", "rvq_rest.input_proj.weight"):
        return "quantizer." + n
    if n == "upsample.convtr.weight": return "upsample.convtr.convtr.convtr.weight"
    if n == "downsample.conv.weight": return "downsample.conv.conv.conv.weight"
    if p[0] in ("transformer", "enc_transformer"):
        pre = ("decoder_transformer" if p[0] == "transformer" else "encoder_transformer") + f".transformer.layers.{p[1]}."
        tail = ".".join(p[2:])
        tail = {"in_proj_weight": "self_attn.in_proj_weight", "out_proj.weight": "self_attn.out_proj.weight"}.get(tail, tail)
        return pre + tail
    if p[0] == "seanet":
        if p[1] == "conv_in": return f"decoder.model.0.conv.conv.{p[2]}"
        if p[1] == "conv_out": return f"decoder.model.{2 + 3 * ns}.conv.conv.{p[2]}"
        j = int(p[2])
        if p[3] == "convtr": return f"decoder.model.{2 + 3 * j}.convtr.convtr.{p[4]}"
        return f"decoder.model.{3 + 3 * j}.block.{1 if p[4] == 'conv1' else 3}.conv.conv.{p[5]}"
    if p[0] == "enc":
        if p[1] == "conv_in": return f"encoder.model.0.conv.conv.{p[2]}"
        if p[1] == "conv_out": return f"encoder.model.{2 + 3 * ns}.conv.conv.{p[2]}"
        j = int(p[2])
        if p[3] == "conv": return f"encoder.model.{3 + 3 * j}.conv.conv.{p[4]}"
        return f"encoder.model.{1 + 3 * j}.block.{1 if p[4] == 'conv1' else 3}.conv.conv.{p[5]}"
    raise AssertionError(n)


def test_moshi_checkpoint_name_map_round_trip():
    """from_moshi_state_dict maps every decode- AND encode-side tensor (name and shape) of a moshi-format Mimi
    checkpoint; the inverse map below is written out independently from moshi 0.2.2's module tree."""
    from sesameai.mimi import MimiArgs, encoder_state_dict_layout, from_moshi_state_dict, mimi_tiny_args, state_dict_layout
    for s in (mimi_tiny_args(), MimiArgs()):
        names = state_dict_layout(s) + encoder_state_dict_layout(s)
        canon = {n: torch.empty(shp, dtype=torch.float32).fill_(float(i)) for i, (n, shp, _) in enumerate(names)}
        ns = len(s.ratios)

        moshi = {moshi_name(n, ns): t for n, t in canon.items()}
        assert len(moshi) == len(canon)
        back = from_moshi_state_dict(moshi, s)
        assert set(back) == set(canon)
        for n, t in canon.items():
            assert back[n].shape == t.shape and torch.equal(back[n], t), n
        # a decode-only checkpoint still maps (no encoder -> Segment.audio raises later, Segment.audio_codes works)
        dec_only = {k: v for k, v in moshi.items() if not k.startswith(("encoder", "downsample")) and "input_proj" not in k}
        assert set(from_moshi_state_dict(dec_only, s)) == {n for n, _, _ in state_dict_layout(s)}


def test_load_csm_1b_refuses_to_fall_back_to_random_weights(monkeypatch):
    from sesameai.generator import load_csm_1b
    for v in ("CSM_MODEL_PATH", "CSM_MIMI_PATH", "CSM_SYNTHETIC"):
        monkeypatch.delenv(v, raising=False)
    with pytest.raises(FileNotFoundError, match="CSM_MODEL_PATH"):
        load_csm_1b("cuda")
    with pytest.raises(FileNotFoundError):
        load_csm_1b("cuda", model_path="/nonexistent/model.safetensors")          # the Mimi checkpoint is required too


def test_tts_service_watermarks_every_clip(monkeypatch):
    """reference tts_service.py: generate_with_context ends in watermark(...) + resample back to the model rate."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tts_service_wm", os.path.join(root, "sesameai-tts_amd", "tts_service.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    calls = []

    class WM:
        def encode_wav(self, wav, rate, key, calc_sdr=False, message_sdr=36):
            calls.append((tuple(wav.shape), rate, list(key)))
            return wav * 0.5, None

    class Gen:
        sample_rate, device = 24_000, torch.device("cpu")
        def _tokenize_text_segment(self, text, speaker): return torch.zeros(3, 33).long(), torch.zeros(3, 33).bool()
        def generate_codes(self, t, m, n, temp, topk): return torch.ones(4, 1, 32, dtype=torch.int32)
        def _decode_frames(self, fr): return torch.ones(fr.shape[0] * 1920)

    tts = mod.TTS(voice_dir="/nonexistent")
    tts.generator, tts.watermarker = Gen(), WM()
    audio = tts.generate_with_context("hi", max_audio_length_ms=1000)
    assert len(calls) == 1 and calls[0][1] == 44100 and calls[0][2] == mod.CSM_1B_GH_WATERMARK
    assert abs(audio.shape[0] - 4 * 1920) <= 2 and abs(float(audio[2000:5000].mean()) - 0.5) < 0.02       # marked audio, back at 24 kHz


def test_tts_say_generates_sentence_by_sentence_reports_rtf_and_survives_a_failing_sentence(tmp_path, capsys):
    """reference: TTS.say (tts_service.py:313-470), generation half: one generate_audio_segment per sentence, the reference's
    ``> sentence ... [Audio: ..s in ..s, RTF: ..x]`` line, a failing sentence replaced by fallback silence, the combined WAV
    written when a filename is given; every segment is offered to ``on_segment`` as soon as it exists (the player hook)."""
    import importlib.util
    import wave
    spec = importlib.util.spec_from_file_location("tts_service_amd3", os.path.join(ROOT, "sesameai-tts_amd", "tts_service.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)

    class Gen:
        sample_rate, device = 24_000, torch.device("cpu")
        def _tokenize_text_segment(self, text, speaker):
            if "boom" in text:
                raise RuntimeError("scripted failure")
            return torch.zeros(3, 33).long(), torch.zeros(3, 33).bool()
        def generate_codes(self, t, m, n, temp, topk): return torch.ones(5, 1, 32, dtype=torch.int32)
        def _decode_frames(self, fr): return torch.linspace(-1, 1, fr.shape[0] * 1920)

    tts = mod.TTS(voice_dir="/nonexistent")
    tts.generator, tts.watermarker = Gen(), None
    got = []
    out = tmp_path / "say.wav"
    tts.say("  First sentence. This one goes boom! And a third?  ", output_filename=str(out), on_segment=lambda pcm, sr: got.append((len(pcm), sr)))
    printed = capsys.readouterr().out
    assert printed.count("[Audio: ") == 2 and "RTF: " in printed and "> First sentence. ... " in printed
    assert "Error generating audio for sentence: This one goes boom!: scripted failure" in printed and "Export complete:" in printed
    seg = 5 * 1920 + 24_000 * 600 // 1000                           # audio + 500 ms lead + 100 ms tail silence
    assert got == [(seg, 24_000), (24_000, 24_000), (seg, 24_000)]  # the failed sentence: 1000 ms of fallback silence
    with wave.open(str(out), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (1, 2, 24_000, 2 * seg + 24_000)
    tts.say("   ", output_filename=None)
    assert "No valid text to process" in capsys.readouterr().out
