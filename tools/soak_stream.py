#!/usr/bin/env python3
"""Soak of the streaming path at CSM-1B size: N utterances through Generator.generate_stream (frame steps queued ahead on the caller's
stream, every 10-frame chunk decoded statelessly by Mimi on the side stream -- the chunk's middle replayed from a hipGraph -- while the
next frames run), under T = 0.9 / top-k 50 sampling with per-utterance seeds and lengths that leave ragged last chunks.  Checked per
utterance: the streamed PCM equals, bit for bit, (a) a second streamed run with the same seed and (b) the same frames decoded chunk by
chunk by ANOTHER Mimi handle after the fact on the caller's stream (nothing overlapping) -- so neither the overlap of the two streams
nor the graph replay changes a sample; codes in range, PCM finite.
    python tools/soak_stream.py [utterances] [max frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch  # noqa: E402
from sesameai.generator import Generator, Segment  # noqa: E402
from sesameai.mimi import MimiArgs, MimiCodec  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
L = int(sys.argv[2]) if len(sys.argv) > 2 else 47
margs = csm_1b_args()
model = Model(margs, synthetic_state_dict(margs, seed=1234), max_frames=128, max_prefill_rows=512)
codec, codec2 = MimiCodec(MimiArgs(), None, max_frames=128), MimiCodec(MimiArgs(), None, max_frames=128)
gen = Generator(model, audio_tokenizer=codec)
g = torch.Generator().manual_seed(17)
jobs = []
for i in range(N):
    n_ctx = int(torch.randint(0, 3, (1,), generator=g))
    ctx = [Segment(speaker=int(torch.randint(0, 2, (1,), generator=g)), text=torch.randint(0, margs.text_vocab_size, (int(torch.randint(5, 40, (1,), generator=g)),), generator=g).tolist(),
                   audio_codes=torch.randint(0, 2048, (32, int(torch.randint(10, 90, (1,), generator=g))), generator=g)) for _ in range(n_ctx)]
    text = torch.randint(0, margs.text_vocab_size, (int(torch.randint(4, 30, (1,), generator=g)),), generator=g).tolist()
    jobs.append((text, ctx, 3 + int(torch.randint(0, L, (1,), generator=g))))
runs = []
for rep in range(2):
    t0 = time.time(); out = []
    for i, (text, ctx, n) in enumerate(jobs):
        model.seed(1000 + i)
        out.append(torch.cat(list(gen.generate_stream(text, 1, ctx, max_audio_length_ms=n * 80.0, temperature=0.9, topk=50))).cpu())
    runs.append(out)
    print(f"run {rep}: {N} streamed utterances of {min(j[2] for j in jobs)}..{max(j[2] for j in jobs)} frames in {time.time() - t0:.1f}s", flush=True)
bad = [i for i in range(N) if not torch.equal(runs[0][i], runs[1][i])]
assert not bad, f"streamed utterances {bad[:8]} differ between two runs with the same seed"
for i, (text, ctx, n) in enumerate(jobs):
    model.seed(1000 + i)
    tok, msk = gen._build_prompt(text, 1, ctx)
    frames = gen.generate_codes(tok, msk, n, 0.9, 50)[:, 0]                    # [n][32] on the host: the same frames, no streaming
    assert frames.shape == (n, 32) and int(frames.min()) >= 0 and int(frames.max()) < margs.audio_vocab_size
    pcm = torch.cat([codec2.decode(frames[t:t + 10].t().unsqueeze(0).contiguous().cuda())[0, 0] for t in range(0, n, 10)]).cpu()
    assert torch.isfinite(pcm).all() and pcm.shape == runs[0][i].shape, (pcm.shape, runs[0][i].shape)
    assert torch.equal(pcm, runs[0][i]), f"utterance {i}: streamed PCM differs from the after-the-fact chunk decode (max |d| {(pcm - runs[0][i]).abs().max():.3e})"
print(f"soak ok: {N} utterances bit-identical across two streamed runs and against the sequential chunk decode of the same frames")
