"""Prefill of one S-row prompt, repeated: wall time per prefill, and (under rocprofv3 --kernel-trace --stats) the
per-kernel picture of the prompt path alone.   python tools/prefill_prof.py [S] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sesameai-tts_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402


def main():
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 190
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    margs = csm_1b_args()
    m = Model(margs, synthetic_state_dict(margs, seed=1234), max_frames=8, max_prefill_rows=max(256, S))
    m.setup_caches(1)
    g = torch.Generator().manual_seed(3)
    tok = torch.zeros(1, S, 33, dtype=torch.long)
    tok[0, :, :32] = torch.randint(0, 2048, (S, 32), generator=g)
    msk = torch.ones(1, S, 33, dtype=torch.bool); msk[0, :, 32] = False
    pos = torch.arange(S).unsqueeze(0)
    for _ in range(2):
        m.reset_caches(); m.prefill(tok, msk, pos)
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):                 # one prefill at a time (a queue of 10 x 100+ launches stalls the runtime for tens of ms)
        t0 = time.perf_counter()
        m.reset_caches(); m.prefill(tok, msk, pos)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    ms = times[len(times) // 2]           # median
    flop = 2.0 * S * (16 * (2048 * 3072 + 2048 * 2048 + 3 * 2048 * 8192))
    print(f"prefill S={S}: {ms:.3f} ms per call (median of {reps}, max {times[-1]:.2f}), {flop / ms / 1e9:.1f} TFLOP/s of projection work "
          f"(G128_MIN_ROWS={os.environ.get('CSM_G128_MIN_ROWS', 'default')})")


if __name__ == "__main__":
    main()
