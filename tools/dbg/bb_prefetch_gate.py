#!/usr/bin/env python3
"""Gate for VERDICT r5 next #2: take the bytes of the batched backbone chain's MLP kernels off the critical path.  At B = 32 the chain
(16 layers x 7 launches) is the one part of the step that is a real HBM stream and runs at 0.23 of peak: per layer q|k|v 9.5, split-key
attention 10.1, o-proj ~5, finisher 4.7 us are latency-bound and leave HBM nearly idle, then gate/up (67 MB, 15.3 us) and down (34 MB, ~11 us)
stream.  CSM_BB_PREFETCH=n forks n blocks onto a second stream at the top of every layer that touch that layer's gate/up/down weights (100 MB)
while the latency-bound kernels run, joined in front of gate/up -- so the MLP kernels find their weights in the 256 MB Infinity Cache.
Measured per setting in a process of its own: (i) backbone-only steps (csm_prefill with S = 1 rows, eager launches, as tools/dbg/bb_two_branches.py),
(ii) whole frame steps through the captured graph (the fork / join become graph branches), both at B = 32, positions 190.."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    sys.path.insert(0, os.path.join(ROOT, "sesameai-tts_amd")); sys.path.insert(0, ROOT)
    import bench
    from types import SimpleNamespace
    from sesameai.models import Model, csm_1b_args, synthetic_state_dict
    args = csm_1b_args()
    sd = synthetic_state_dict(args, seed=1234)
    ba = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
    B, N = 32, 200
    tok, msk = bench.synthetic_prompt(ba, B, args.text_vocab_size, seed0=4000)
    S = tok.shape[1]
    m = Model(args, sd, max_frames=64, max_prefill_rows=B * S)
    m.setup_caches(B)
    m.seed(7)
    pos = torch.arange(S).unsqueeze(0).repeat(B, 1)
    m.prefill(tok, msk, pos)
    row = torch.randint(0, 2048, (B, 1, 33), generator=torch.Generator().manual_seed(1)).cuda()
    rmask = torch.ones(B, 1, 33, dtype=torch.bool); rmask[:, 0, 32] = False
    rmask = rmask.cuda()

    def bb_steps(n):
        for i in range(n):
            m.prefill(row, rmask, torch.full((B, 1), S + (i % 1000), device="cuda", dtype=torch.int64))

    def timed(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / n
    bb_steps(20)
    t_bb = min(timed(bb_steps, N) for _ in range(3))
    m.reset_caches(); m.prefill(tok, msk, pos); m.depth(B, 0.9, 50, commit=True)

    def frames(n):
        for _ in range(n):
            m.step(B, 0.9, 50)
    frames(10)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        m.reset_caches(); m.prefill(tok, msk, pos); m.depth(B, 0.9, 50, commit=True); frames(5)
        torch.cuda.synchronize(); e0.record(); frames(40); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40)
    fr, _ = m.read_frames(B)
    print(f"RESULT prefetch_blocks={os.environ.get('CSM_BB_PREFETCH', '0'):>4s}  backbone-only step {t_bb:.4f} ms   whole frame step (graph) {best:.4f} ms   "
          f"checksum {int(fr.sum())}", flush=True)
    sys.exit(0)

print(__doc__)
for n in ("0", "32", "64", "128", "256", "512", "0"):
    env = dict(os.environ, CSM_BB_PREFETCH=n)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(lines[-1] if lines else f"prefetch_blocks={n}: FAILED rc={r.returncode}\n{r.stderr[-1500:]}", flush=True)
