#!/usr/bin/env python3
"""Soak of the continuously refilled batch (csm_refill_begin / csm_refill_advance, ring history) at CSM-1B size: N prompts of different lengths
through B slots under T = 0.9 / top-k 50 sampling, each with its own length limit (random weights never emit EOS; different limits make the
slots retire at different steps, so prompts are refilled BESIDE the running batch), twice with the same seed -- the refill schedule is a pure
function of the host loop, so every utterance must come back bit-identical in the second run, complete and in range; the number of global frame
steps exceeds the history ring several times.
    python tools/soak_refill.py [B] [prompts] [frames per utterance]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd"))
import torch  # noqa: E402
from sesameai.generator import Generator  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4 * B
L = int(sys.argv[3]) if len(sys.argv) > 3 else 24
margs = csm_1b_args()
m = Model(margs, synthetic_state_dict(margs, seed=1234), max_frames=64, max_prefill_rows=max(256, 2 * B))      # a 64-frame ring: the run wraps it many times
gen = Generator.__new__(Generator)
gen._model, gen._max_batch, gen._eos_poll, gen.device = m, B, 8, m.device
m.setup_caches(B)
g = torch.Generator().manual_seed(5)
prompts = []
for i in range(N):
    S = 30 + int(torch.randint(0, 150, (1,), generator=g))
    t = torch.zeros(S, 33, dtype=torch.long); mk = torch.zeros(S, 33, dtype=torch.bool)
    nt = S // 3
    t[:nt, 32] = torch.randint(0, margs.text_vocab_size, (nt,), generator=g); mk[:nt, 32] = True
    t[nt:, :32] = torch.randint(0, 2048, (S - nt, 32), generator=g); mk[nt:, :32] = True
    prompts.append((t, mk))
limits = [max(2, L // 3 + int(torch.randint(0, L, (1,), generator=g))) for _ in range(N)]
runs = []
for rep in range(2):
    m.seed(99)
    t0 = time.time()
    adv = []
    orig = m.refill_advance
    m.refill_advance = lambda k: (adv.append(k), orig(k))[1]
    out = gen.generate_codes_continuous(prompts, limits, 0.9, 50)
    m.refill_advance = orig
    steps = m.num_frames()
    assert len(out) == N and all(o.shape == (limits[i], 32) for i, o in enumerate(out)), [o.shape for o in out][:5]
    assert all(int(o.min()) >= 0 and int(o.max()) < margs.audio_vocab_size for o in out)
    runs.append(out)
    print(f"run {rep}: {N} utterances of {min(limits)}..{max(limits)} frames through {B} slots in {time.time() - t0:.1f}s, {steps} global frame steps (ring of 64), "
          f"{sum(1 for k in adv if k < 16)} bounded refill calls beside the loop + {sum(1 for k in adv if k >= 16)} whole-prompt calls", flush=True)
bad = [i for i in range(N) if not torch.equal(runs[0][i], runs[1][i])]
assert not bad, f"utterances {bad[:8]} differ between two runs with the same seed"
print("soak ok: both runs bit-identical")
