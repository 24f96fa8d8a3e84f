"""Timeline of the BATCHED persistent depth-decoder launch (csrc/dec_persist_m.cuh): where one decoder step's time goes, from
s_memrealtime stamps of workgroup 100 (gather waves 0..2 and compute wave 0).  Needs the stamped build of the library:
    make -C sesameai-tts_amd/csrc timeline && python tools/persist_m_timeline.py [B]"""
import ctypes as C
import os
import sys

os.environ["CSM_HIP_TIMELINE"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sesameai-tts_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from sesameai import _abi  # noqa: E402
from sesameai.models import Model, csm_1b_args, synthetic_state_dict  # noqa: E402

NW = 32 * 32 + 4096 + 256


def main():
    from types import SimpleNamespace
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    args = SimpleNamespace(ctx_text=40, ctx_frames=125, gen_text=24)
    margs = csm_1b_args()
    m = Model(margs, synthetic_state_dict(margs, seed=1234), max_frames=64, max_prefill_rows=B * 190)
    m.setup_caches(B); m.seed(1)
    tok, msk = bench.synthetic_prompt(args, B, margs.text_vocab_size)
    S = tok.shape[1]
    m.prefill(tok, msk, torch.arange(S).unsqueeze(0).repeat(B, 1))
    m.depth(B, 0.9, 50, commit=True)
    assert _abi.lib.csm_debug_persist_stamps(m._h, None, 0) == 0, "this handle does not run the persistent launch"
    for _ in range(6):
        m.step(B, 0.9, 50)
    buf = (C.c_uint64 * NW)()
    assert _abi.lib.csm_debug_persist_stamps(m._h, buf, NW) == 0
    t = torch.tensor(list(buf)[:30 * 128], dtype=torch.float64).view(30, 128) * 0.01       # us

    def mean(a, b, prev_a=False):
        vals = []
        for s in range(3, 29):
            ta = t[s - 1, a] if prev_a else t[s, a]
            if ta > 0 and t[s, b] > 0:
                vals.append(float(t[s, b] - ta))
        return sum(vals) / max(len(vals), 1)

    print(f"batched persistent depth decoder, B={B}, workgroup 100, mean over steps 3..28 (us)")
    print("  gather waves (time between consecutive events):")
    tot = 0.0
    prev, prev_is_last_step = 35, True          # table rows of the previous step
    for l in range(4):
        ev = []
        if l > 0:
            ev += [(f"L{l} x all-gather + sa_norm in LDS", l * 8 + 0), (f"L{l} q|k|v of my (row, head) swept", l * 8 + 1)]
        ev += [(f"L{l} attention done, published", l * 8 + 2), (f"L{l} attention all-gather in LDS", l * 8 + 3), (f"L{l} h1 all-gather + mlp_norm in LDS", l * 8 + 4),
               (f"L{l} group h gather in LDS", l * 8 + 5), (f"L{l} partials summed, rows published (wave 1)", l * 8 + 6)]
        for name, idx in ev:
            d = mean(prev, idx, prev_is_last_step)
            prev_is_last_step = False
            if idx % 8 == 6:
                print(f"     {name:60s} {d:7.2f}   (after the group h gather)")
                tot += d
                prev = idx
            else:
                print(f"     {name:60s} {d:7.2f}")
                tot += d
                prev = idx
    for name, idx in (("final x all-gather + norm in LDS", 32), ("logits of my row swept", 33), ("sampled", 34), ("next step's table rows in LDS", 35)):
        d = mean(prev, idx)
        print(f"     {name:60s} {d:7.2f}")
        tot += d
        prev = idx
    print(f"     {'sum':60s} {tot:7.2f}")
    print(f"     step period measured directly: {float(t[27, 35] - t[3, 35]) / 24:.2f}")
    print("  compute wave 0 (flag seen -> published) and the wait in front of it:")
    for l in range(4):
        names = ["q|k|v", "o-proj", "gate/up", "down"]
        for i, nm in enumerate(names):
            if l == 0 and i == 0:
                continue
            seen, done = 64 + l * 8 + 2 * i, 64 + l * 8 + 2 * i + 1
            before = (64 + l * 8 + 2 * i - 1) if (l, i) != (0, 1) else None
            wait = mean(before, seen) if before is not None else float("nan")
            print(f"     L{l} {nm:8s} wait {wait:6.2f}   compute + publish {mean(seen, done):6.2f}")
    print(f"     head      wait {mean(64 + 3 * 8 + 7, 96):6.2f}   compute + publish {mean(96, 97):6.2f}")
    # cross-role latencies: publish (compute) -> fill seen etc.
    print("  cross checks (L2): o-proj published -> h1 in LDS", f"{mean(64 + 2 * 8 + 3, 2 * 8 + 4):.2f};",
          "gate/up published -> group h in LDS", f"{mean(64 + 2 * 8 + 5, 2 * 8 + 5):.2f};",
          "down published -> rows published", f"{mean(64 + 2 * 8 + 7, 2 * 8 + 6):.2f};",
          "q|k|v published -> swept", f"{mean(64 + 2 * 8 + 1, 2 * 8 + 1):.2f};",
          "attention published -> all-gather in LDS", f"{mean(2 * 8 + 2, 2 * 8 + 3):.2f};",
          "h1 in LDS -> gate/up flag seen", f"{mean(2 * 8 + 4, 64 + 2 * 8 + 4):.2f}")


if __name__ == "__main__":
    main()
