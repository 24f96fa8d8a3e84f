"""Timeline of the persistent depth-decoder launch (csrc/dec_persist.cuh) on the bench workload: where one decoder
step's time goes, from s_memrealtime stamps of workgroup 100.  Needs the stamped build of the library:
    make -C sesameai-tts_amd/csrc timeline && python tools/persist_timeline.py"""
import ctypes as C
import os
import sys

os.environ["CSM_HIP_TIMELINE"] = "1"      # the library build with the stamps compiled in (make -C sesameai-tts_amd/csrc timeline)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sesameai-tts_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from sesameai import _abi  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402


def main():
    from types import SimpleNamespace
    args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
    margs = csm_1b_args()
    m = Model(margs, synthetic_state_dict(margs, seed=1234), max_frames=64, max_prefill_rows=256)
    m.setup_caches(1); m.seed(1)
    tok, msk = bench.synthetic_prompt(args, 1, margs.text_vocab_size)
    S = tok.shape[1]
    m.prefill(tok, msk, torch.arange(S).unsqueeze(0))
    m.depth(1, 0.9, 50, commit=True)
    assert _abi.lib.csm_debug_persist_stamps(m._h, None, 0) == 0, "this handle does not run the persistent launch"
    for _ in range(6):
        m.step(1, 0.9, 50)
    buf = (C.c_uint64 * (32 * 32 + 4096 + 256))()
    assert _abi.lib.csm_debug_persist_stamps(m._h, buf, 32 * 32 + 4096 + 256) == 0
    allw = torch.tensor(list(buf), dtype=torch.float64)
    t = allw[:1024].view(32, 32) * 0.01          # us
    n_steps = 30
    names = []
    seq = []                                                                       # (label, from, to) within a step
    for l in range(4):
        if l > 0:
            seq.append((f"L{l} rows published(L{l-1}) -> x of q|k|v ready (edge + sa_norm)", (l - 1) * 4 + 3, l * 4 + 0))
            seq.append((f"L{l} x ready -> q/k/v in LDS (q|k|v unit + edge)", l * 4 + 0, l * 4 + 1))
        seq.append((f"L{l} q/k/v ready -> x of MLP ready (attention + o-proj + edge + mlp_norm)", (l * 4 + 1) if l > 0 else None, l * 4 + 2))
        seq.append((f"L{l} x of MLP ready -> rows published (gate/up + split down + partials edge + sum)", l * 4 + 2, l * 4 + 3))
    seq.append(("rows published(L3) -> x of head ready (edge + final norm)", 15, 16))
    seq.append(("  x of head ready -> head wave 2 saw it", 16, 20))
    seq.append(("  head wave 2: rows computed and published", 20, 21))
    seq.append(("  published -> logits in LDS (edge)", 21, 17))
    seq.append(("  logits in LDS -> sampler waves saw them", 17, 22))
    seq.append(("  sampler body", 22, 23))
    seq.append(("  code written -> gather wave saw it", 23, 18))
    seq.append(("code -> next step's table rows in LDS", 18, 19))
    tot = 0.0
    print("persistent depth decoder, workgroup 100, mean over steps 3..28 (us):")
    for label, a, b in seq:
        vals = []
        for s in range(3, n_steps - 1):
            ta = t[s, a] if a is not None else t[s - 1, 19]
            if ta > 0 and t[s, b] > 0:
                vals.append(float(t[s, b] - ta))
        mean = sum(vals) / max(len(vals), 1)
        tot += mean
        print(f"   {label:88s} {mean:6.2f}")
    print(f"   {'sum = one decoder step':88s} {tot:6.2f}")
    raw = allw[:1024].view(32, 32)
    print("   poll passes per sweep (logits, head x, q|k|v [3 layers], partials [4 layers]):",
          [round(float(raw[3:29, 24 + i].mean()) / 6, 2) for i in range(4)], "(per step; 3 q|k|v and 4 partials sweeps per step)")
    pub = allw[1024:1024 + 2048].view(256, 8) * 0.01; rdy = allw[3072:3072 + 256] * 0.01; fxa = allw[3328:3328 + 256] * 0.01
    t0 = float(fxa[fxa > 0].min())
    print(f"   step 5, head phase over all 256 workgroups (us after the first 'x of head ready'): x ready {float(fxa.min()) - t0:.2f}..{float(fxa.max()) - t0:.2f}")
    for w in range(2, 7):
        col = pub[:, w][pub[:, w] > 0]
        if col.numel():
            print(f"      wave {w} published: {float(col.min()) - t0:.2f} .. {float(col.max()) - t0:.2f} (slowest CU {int(pub[:, w].argmax())}, {col.numel()} producers)")
    print(f"      logits in LDS: {float(rdy.min()) - t0:.2f} .. {float(rdy.max()) - t0:.2f}")
    ex = allw[4096:4096 + 256].view(32, 8) * 0.01
    def mean_d(a_, b_, ref=None):
        vals = [float(ex[s_, b_] - (ex[s_, a_] if ref is None else ref[s_])) for s_ in range(3, 29) if ex[s_, b_] > 0]
        return sum(vals) / max(len(vals), 1)
    print("   layer 2 detail:")
    print(f"      q/k/v in LDS -> wave 5 starts its attention head      {mean_d(0, 0, t[:, 2 * 4 + 1]):6.2f}")
    print(f"      wave 5: attention head                                 {mean_d(0, 1):6.2f}")
    print(f"      wave 5: attention done -> o-proj rows published        {mean_d(1, 2):6.2f}")
    print(f"      o-proj published -> x of MLP ready (edge + mlp_norm)   {float(sum(float(t[s_, 2 * 4 + 2] - ex[s_, 2]) for s_ in range(3, 29)) / 26):6.2f}")
    print(f"      x of MLP ready -> wave 0 starts gate/up                {mean_d(3, 3, t[:, 2 * 4 + 2]):6.2f}")
    print(f"      wave 0: 6 (gate, up) pairs + h exchange                {mean_d(3, 4):6.2f}")
    print(f"      wave 0: 3 row blocks of the down partial, published    {mean_d(4, 5):6.2f}")
    print(f"      partials published -> rows published (edge + sum)      {float(sum(float(t[s_, 2 * 4 + 3] - ex[s_, 5]) for s_ in range(3, 29)) / 26):6.2f}")
    sm = allw[4352:4352 + 512].view(32, 16) * 0.01
    names = ["flag seen -> per-thread maxima stored", "barrier + kth bound of the 128 pair maxima", "count, scan, barrier, candidate list, barrier",
             "t = logit / T, exact kth, survivors re-packed (wave 0)", "log-softmax, softmax", "barrier (Exp(1) draws ready) + race"]
    print("   sampler detail (thread 0 of the quad):")
    prev = t[:, 22]
    for i, nm in enumerate(names):
        vals = [float(sm[s_, i] - prev[s_]) for s_ in range(3, 29) if sm[s_, i] > 0]
        print(f"      {nm:58s} {sum(vals) / max(len(vals), 1):6.2f}")
        prev = sm[:, i]
    vals = [float(t[s_, 23] - sm[s_, 5]) for s_ in range(3, 29) if sm[s_, 5] > 0]
    print(f"      {'argmax over the wave, last barrier':58s} {sum(vals) / max(len(vals), 1):6.2f}")
    vals = [float(sm[s_, 7] - sm[s_, 6]) for s_ in range(3, 29) if sm[s_, 7] > 0]
    print(f"      {'sampler body alone, last pass (CSM_PERSIST_TRICKLE=134 runs it twice: warm instruction cache)':58s} {sum(vals) / max(len(vals), 1):6.2f}")
    bl = allw[5312:5312 + 16] * 0.01
    if float(bl[12]) > 0:
        nm = ["entry (wave 0)", "q|k|v pair published (wave 0)", "attention vector in LDS (gather wave)", "o-proj rows published (wave 0)", "h1 in LDS (gather wave)",
              "gate/up done (wave 0)", "gate/up done (gather wave)", "h values exchanged (wave 0)", "W2 slice in LDS (wave 0)", "W2 slice in LDS (gather wave)",
              "partials published (wave 0)", "partials published (gather wave)", "rows written (gather wave)"]
        print("   whole-backbone-layer launch (k_bb_layer), workgroup 100, layer 8, us after entry:")
        for i, n_ in enumerate(nm):
            print(f"      {n_:44s} {float(bl[i] - bl[0]):6.2f}")
    whole = float(t[n_steps - 2, 19] - t[2, 19]) / (n_steps - 4)
    print(f"   step period measured directly: {whole:.2f} us")


if __name__ == "__main__":
    main()
