"""Generates the committed golden vectors under tests/golden/ from the oracle.

TEST INFRASTRUCTURE ONLY.  Run here (CPU, minutes):  python -m oracle.make_golden [--only ...]
Nothing in this script reads /root/reference; the vectors are functions of the oracle code,
the seeds below and torch's CPU generator.

Files (all torch.save'd dicts of small tensors):
  csm1b_frames.pt   CSM-1B shapes, seeded weights (seed 1234), config-1 prompt (16 text tokens):
                    the oracle's greedy trajectory for N frames, and for every frame/codebook
                    the top-8 (value, index) of the bf16 logits, teacher-forced on that
                    trajectory, plus the bf16-vs-fp32 oracle logit gap (the rounding-noise floor
                    any bf16 implementation lives in).
  tiny_frames.pt    same for the tiny shapes with a 2-segment prompt, full logits kept.
  sampler_cases.pt  sample_topk inputs/outputs incl. tie cases and supplied Exp(1) noise.
  prompt_layout.pt  (tokens, mask) of a 2-segment toy prompt (generator.py:63-109 layout).
  mimi_*.pt         Mimi decode of a seeded (1,32,10) code block: whole and stateless chunks.
  csm1b_cfg2.pt     CSM-1B, BASELINE config 2's prompt (bench.py's seeded S=190 prompt: 40 text + 125 audio + EOS + 24 text
                    rows): 6 greedy frames, top-8 logits, margins, bf16-vs-fp32 gap.
  csm1b_cfg3.pt     the same shape at B=32 (bench.py's prompts of utterances 0..31): 2 frames of the batched oracle, top-8
                    logits and margins per (frame, codebook, utterance).
  csm1b_cfg5.pt     BASELINE config 5: fp8-e4m3-dequantised weights (oracle.csm_ref.fp8_dequantized), S=1334 prompt (10
                    segments), 2 frames; and a 1700-row prompt: the prompt frame (p = 1699) and one step frame (p = 1700).
  csm1b_cfg5b.pt    config 5 batched at B = 4 (four different 1334-row prompts), 2 frames of the batched oracle.
  csm1b_cfg5c.pt    config 5 at B = 32 (the batch SURVEY.md 8d names): top-8 logits / codes / margins only, prompts by seed.
  csm1b_decisive.pt CSM-1B with the DECISIVE synthetic checkpoint (oracle.csm_ref.decisive_weights): the oracle's FREE-RUNNING greedy
                    codes -- 64 frames at B = 1 for the 190-row and the 1334-row prompt, 16 frames at B = 32, twelve utterances of mixed
                    lengths for the refilled batch -- each with bf16 and with fp8-dequantised weights, every row's top-1 / top-2 margin
                    asserted >= 4 x the oracle's bf16-vs-fp32 gap on that run; + the Mimi oracle's PCM of the first clip.
  tiny_decisive.pt  the same for the tiny shapes (full logits of the first frames kept).
  csm1b_decisive_copy.pt / tiny_decisive_copy.pt
                    the HISTORY-DEPENDENT decisive checkpoint (oracle.csm_ref.decisive_copy_weights: c0 of a frame names the last code of
                    the row `lag` positions back, read through one backbone layer's cached K / V): free-running greedy codes for the
                    190-row prompt (copy layer 8, lag 3: from frame 4 on the rows read are the ones the frame steps appended), for the
                    1334-row prompt (copy layer 3, lag 700: a key in the middle of the split key range) and at B = 32; bf16 and
                    fp8-dequantised; margin >= 4 x gap on every row, the implied trajectory asserted, and KV faults injected into the
                    oracle (stale rows, a wrong RoPE position, a dropped key range, zeroed prompt rows) asserted to CHANGE the codes.
  csm1b_possweep.pt CSM-1B bench checkpoint, ONE 2046-row prompt cut at S in POSSWEEP_S: for every S the prompt frame and two teacher-forced
                    steps at positions S and S + 1 (top-8 logits, codes, margins, the bf16-vs-fp32 gap), bf16 and fp8-dequantised; plus 64
                    CONSECUTIVE teacher-forced steps after a 740-row prompt (positions 740..803: across the one-launch backbone layer's
                    switch from one CU per head to the key range split over 8 CUs at 768).
"""
from __future__ import annotations

import argparse
import os
import time

import torch

from . import csm_ref as C
from . import mimi_ref as M

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def toy_prompt(shape: C.CsmShape, seed: int, n_text: int, ctx_frames: int):
    g = torch.Generator().manual_seed(seed)
    segs = []
    if ctx_frames:
        ids = torch.randint(0, shape.text_vocab_size, (n_text,), generator=g).tolist()
        codes = torch.randint(0, 2048, (shape.audio_num_codebooks, ctx_frames), generator=g)
        segs.append((ids, codes))
    ids = torch.randint(0, shape.text_vocab_size, (n_text,), generator=g).tolist()
    segs.append((ids, None))
    return C.build_prompt(segs)


@torch.inference_mode()
def frames_golden(shape: C.CsmShape, weights, prompt, n_frames: int, keep_full: bool, with_fp32: bool, quiet: bool = False, w32=None):
    tok, msk = prompt
    m = C.OracleModel(shape, weights)
    m.setup_caches(1)
    m32 = None
    if with_fp32:
        m32 = C.OracleModel(shape, w32 if w32 is not None else {k: v.float() for k, v in weights.items()}, dtype=torch.float32)
        m32.setup_caches(1)
    cur_t, cur_m = tok.unsqueeze(0), msk.unsqueeze(0)
    pos = torch.arange(tok.size(0)).unsqueeze(0)
    codes, top_v, top_i, full, gap, margin = [], [], [], [], [], []
    for f in range(n_frames):
        t0 = time.time()
        tr = C.FrameTrace()
        # greedy frame; the oracle feeds back its own choice == teacher forcing on its trajectory
        s = m.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, trace=tr)
        lg = torch.stack(tr.logits, 0)[:, 0]                     # (32, V) bf16
        v, i = torch.topk(lg.float(), 8, dim=-1)
        codes.append(s[0]); top_v.append(v.to(torch.bfloat16)); top_i.append(i.to(torch.int16))
        margin.append((v[:, 0] - v[:, 1]))
        if keep_full:
            full.append(lg)
        if m32 is not None:
            tr32 = C.FrameTrace()
            m32.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, forced=s, trace=tr32)
            lg32 = torch.stack(tr32.logits, 0)[:, 0]
            gap.append((lg.float() - lg32).abs().max(dim=-1)[0])
        cur_t = torch.cat([s.long(), torch.zeros(1, 1).long()], dim=1).unsqueeze(1)
        cur_m = torch.cat([torch.ones_like(s).bool(), torch.zeros(1, 1).bool()], dim=1).unsqueeze(1)
        pos = pos[:, -1:] + 1
        if not quiet:
            print(f"  frame {f}: {time.time() - t0:.1f}s  c0..3={s[0, :4].tolist()}", flush=True)
    out = dict(prompt_tokens=tok, prompt_mask=msk, codes=torch.stack(codes), top_v=torch.stack(top_v),
               top_i=torch.stack(top_i), margin=torch.stack(margin))
    if keep_full:
        out["logits"] = torch.stack(full)
    if gap:
        out["bf16_vs_fp32_gap"] = torch.stack(gap)
    return out


def bench_prompt(shape: C.CsmShape, seed: int, segments: int = 1, ctx_text: int = 40, ctx_frames: int = 125, gen_text: int = 24):
    """bench.py::synthetic_prompt for one utterance (same generator call order), restated here so the golden files do
    not depend on the benchmark script; tests assert the two agree."""
    g = torch.Generator().manual_seed(seed)
    rows = segments * (ctx_text + ctx_frames + 1) + gen_text
    t = torch.zeros(rows, 33, dtype=torch.long)
    m = torch.zeros(rows, 33, dtype=torch.bool)
    r = 0
    for _ in range(segments):
        t[r:r + ctx_text, 32] = torch.randint(0, shape.text_vocab_size, (ctx_text,), generator=g); m[r:r + ctx_text, 32] = True
        r += ctx_text
        t[r:r + ctx_frames, :32] = torch.randint(0, 2048, (ctx_frames, 32), generator=g); m[r:r + ctx_frames + 1, :32] = True
        r += ctx_frames + 1
    t[r:r + gen_text, 32] = torch.randint(0, shape.text_vocab_size, (gen_text,), generator=g); m[r:r + gen_text, 32] = True
    return t, m


@torch.inference_mode()
def frames_golden_batch(shape: C.CsmShape, weights, toks, msks, n_frames: int):
    """Batched oracle (B utterances of equal length), greedy, teacher-forced on its own trajectory: per frame the codes
    [B][32], top-8 logits [32][B][8] and the top-1/top-2 margin [32][B]."""
    B = toks.shape[0]
    m = C.OracleModel(shape, weights)
    m.setup_caches(B)
    cur_t, cur_m = toks, msks
    pos = torch.arange(toks.size(1)).unsqueeze(0).repeat(B, 1)
    codes, top_v, top_i, margin = [], [], [], []
    for f in range(n_frames):
        t0 = time.time()
        tr = C.FrameTrace()
        s = m.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, trace=tr)
        lg = torch.stack(tr.logits, 0)                           # (32, B, V) bf16
        v, i = torch.topk(lg.float(), 8, dim=-1)
        codes.append(s.clone()); top_v.append(v.to(torch.bfloat16)); top_i.append(i.to(torch.int16)); margin.append(v[..., 0] - v[..., 1])
        cur_t = torch.cat([s.long(), torch.zeros(B, 1).long()], dim=1).unsqueeze(1)
        cur_m = torch.cat([torch.ones_like(s).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
        pos = pos[:, -1:] + 1
        print(f"  batch frame {f}: {time.time() - t0:.1f}s", flush=True)
    return dict(prompt_tokens=toks, prompt_mask=msks, codes=torch.stack(codes), top_v=torch.stack(top_v), top_i=torch.stack(top_i),
                margin=torch.stack(margin))


@torch.inference_mode()
def free_run(shape: C.CsmShape, weights, toks, msks, n_frames: int, what: str):
    """The oracle's FREE-RUNNING greedy loop (generator.py:283-294 with topk = 1) on B equal-length prompts, beside an fp32 oracle that is
    fed the bf16 oracle's codes: codes [n][B][32], the smallest top-1 / top-2 margin and the largest bf16-vs-fp32 logit gap per frame.
    Asserts what the decisive checkpoint is for: every decision's margin >= 4 x the gap, and the trajectory the construction implies."""
    if toks.dim() == 2:
        toks, msks = toks.unsqueeze(0), msks.unsqueeze(0)
    B = toks.shape[0]
    m = C.OracleModel(shape, weights); m.setup_caches(B)
    m32 = C.OracleModel(shape, {k: v.float() for k, v in weights.items()}, dtype=torch.float32); m32.setup_caches(B)
    cur_t, cur_m = toks.long(), msks
    pos = torch.arange(toks.size(1)).unsqueeze(0).repeat(B, 1)
    codes, margins, gaps = [], [], []
    t0 = time.time()
    for f in range(n_frames):
        tr, tr32 = C.FrameTrace(), C.FrameTrace()
        s = m.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, trace=tr)
        m32.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, forced=s, trace=tr32)
        lg, lg32 = torch.stack(tr.logits, 0).float(), torch.stack(tr32.logits, 0)       # [32][B][V]
        top2 = torch.topk(lg, 2, dim=-1)[0]
        margin, gap = (top2[..., 0] - top2[..., 1]), (lg - lg32).abs().amax(dim=-1)
        assert bool((margin >= 4.0 * gap.max()).all()), f"{what} frame {f}: margin {float(margin.min()):.3f} < 4 x gap {float(gap.max()):.4f}"
        assert not bool((s == 0).all(dim=1).any()), "an all-zero (EOS) frame"
        codes.append(s.clone()); margins.append(margin.min()); gaps.append(gap.max())
        cur_t = torch.cat([s.long(), torch.zeros(B, 1).long()], dim=1).unsqueeze(1)
        cur_m = torch.cat([torch.ones_like(s).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
        pos = pos[:, -1:] + 1
    codes = torch.stack(codes)
    print(f"  {what}: {n_frames} frames x {B} in {time.time() - t0:.0f}s, smallest margin {float(min(margins)):.3f}, largest gap "
          f"{float(max(gaps)):.4f} ({float(min(margins)) / float(max(gaps)):.0f} x)", flush=True)
    return dict(codes=codes.to(torch.int16), min_margin=torch.stack(margins), max_gap=torch.stack(gaps))


def decisive_many_prompts(shape: C.CsmShape):
    """Twelve prompts of mixed lengths and per-utterance frame limits for the continuously refilled batch of 8."""
    spec = [(2100 + i, (8 + 5 * i) % 37 + 4, (17 * i) % 90 + 6, 6 + i % 5, 12 + (7 * i) % 23) for i in range(12)]
    return [(bench_prompt(shape, seed, 1, ct, cf, gt), lim) for seed, ct, cf, gt, lim in spec]


def decisive_golden(shape: C.CsmShape, seed: int, full: bool):
    gold = dict(weight_seed=seed)
    w = C.make_weights(shape, seed=seed, flavour="decisive")
    for tag, wts in (("bf16", w), ("fp8", None)):
        if wts is None:
            wts = C.fp8_dequantized(w)
        if full:
            n1, n32 = 64, 16
            p190 = bench_prompt(shape, 2025)
            p1334 = bench_prompt(shape, 5000, segments=10, ctx_text=30, ctx_frames=100)
            b32 = [bench_prompt(shape, 2025 + b) for b in range(32)]
        else:
            n1, n32 = 24, 8
            p190 = toy_prompt(shape, 11, 6, 5)
            p1334 = toy_prompt(shape, 12, 20, 60)
            b32 = [toy_prompt(shape, 100 + b, 6, 5) for b in range(5)]
        for name, (tok, msk) in (("s190", p190), ("s1334", p1334)):
            g = free_run(shape, wts, tok, msk, n1, f"{tag} {name}")
            want = C.decisive_expected_codes(shape, seed, int(tok[-1, 32]), n1)
            assert torch.equal(g["codes"][:, 0].to(torch.int32), want), "the oracle left the trajectory the construction implies"
            g["prompt_rows"] = tok.shape[0]
            gold[f"{tag}_{name}"] = g
        gold[f"{tag}_b32"] = free_run(shape, wts, torch.stack([p[0] for p in b32]), torch.stack([p[1] for p in b32]), n32, f"{tag} b32")
        if tag == "bf16":
            many = []
            for i, ((tok, msk), lim) in enumerate(decisive_many_prompts(shape)):
                g = free_run(shape, wts, tok, msk, lim, f"{tag} many[{i}] S={tok.shape[0]}")
                many.append(g["codes"][:, 0].clone())
            gold["bf16_many"] = many
    return decisive_finish(gold, w, full)


DECISIVE_CHECKSUM_NAMES = ["text_embeddings.weight", "audio_embeddings.weight", "codebook0_head.weight", "audio_head", "projection.weight",
                           "backbone.layers.7.attn.output_proj.weight", "backbone.layers.15.mlp.w2.weight", "decoder.layers.3.mlp.w2.weight"]


def decisive_finish(gold, w, full: bool):
    """What the golden carries beside the codes: checksums of the re-derived tensors (the GPU test builds the checkpoint with the
    product's own generator and must get the same ones) and, full size, the Mimi oracle's PCM of the first clip -- whole
    (generator.py:299) and as the stateless 10-frame chunks generate_stream yields (generator.py:119-210; every 4th sample kept)."""
    names = [n for n in DECISIVE_CHECKSUM_NAMES if n in w]
    gold["weight_checksum_names"] = names
    gold["weight_checksum"] = torch.stack([w[n].view(torch.int16).to(torch.int64).sum() for n in names])    # exact, order-free
    if full:
        s = M.mimi_full()
        mw = M.make_weights(s, seed=4321)
        codes = gold["bf16_s190"]["codes"][:, 0].long().t().unsqueeze(0)            # (1, 32, T)
        gold["pcm_s190"] = M.decode(s, mw, codes)[0, 0].clone()
        gold["pcm_chunks_stride"] = 4
        gold["pcm_s190_chunks"] = M.decode_stateless_chunks(s, mw, codes, 10)[0, 0][::4].clone()
        # the long prompt's clip too (round 6: it was checked against the HIP codec on the oracle's codes -- HIP vs HIP): every 4th sample
        codes = gold["bf16_s1334"]["codes"][:, 0].long().t().unsqueeze(0)
        gold["pcm_s1334_stride4"] = M.decode(s, mw, codes)[0, 0][::4].clone()
        codes = gold["fp8_s1334"]["codes"][:, 0].long().t().unsqueeze(0)
        gold["pcm_fp8_s1334_stride4"] = M.decode(s, mw, codes)[0, 0][::4].clone()
        gold["mimi_weight_seed"] = 4321
    return gold


# ---- the history-dependent decisive checkpoint ----------------------------------------------------------------------------------
COPY_SHORT = "decisive_copy"            # middle layer, lag 3
COPY_FAULTS = ({"kind": "stale"}, {"kind": "shift_rope", "delta": 1}, {"kind": "zero_prompt"})


def copy_long_flavour(shape: C.CsmShape, rows: int) -> str:
    """the long-prompt variant: an early layer, a lag of about half the prompt (a key in the middle of the key range)."""
    return f"decisive_copy:{min(3, shape.backbone.num_layers - 1)}:{max(4, (rows * 21) // 40)}"


@torch.inference_mode()
def copy_fault_check(shape: C.CsmShape, weights, flavour: str, tok, msk, want, n: int = 8):
    """The point of the flavour: faults in the copy layer's KV cache CHANGE the free-running codes (and the unfaulted run is `want`)."""
    layer, lag = C.copy_flavour_params(shape, flavour)
    S = tok.shape[0]
    # (decode-step faults show once a frame reads a row that a step appended: frame lag + 1 on -- the long-lag variant reads prompt rows only)
    faults = [f for f in COPY_FAULTS if f["kind"] == "zero_prompt" or lag + 1 < n] + [{"kind": "drop_keys", "lo": max(0, S - 1 - lag), "hi": S + n - lag}]
    changed = []
    try:
        for fault in faults:
            C.KV_FAULT = dict(fault, stack="backbone", layer=layer)
            m = C.OracleModel(shape, weights); m.setup_caches(1)
            got = torch.cat(C.generate_codes(m, tok, msk, n * 80, 1.0, 1, greedy=True, max_seq_len=shape.backbone.max_seq_len))
            changed.append(int((got != want[:n]).any(dim=1).sum()))
            assert changed[-1] > 0, f"{flavour}: the fault {fault} in the copy layer's cache does not change the codes"
    finally:
        C.KV_FAULT = None
    print(f"  {flavour}: KV faults {[f['kind'] for f in faults]} change {changed} of the first {n} frames", flush=True)
    return torch.tensor(changed)


def decisive_copy_golden(shape: C.CsmShape, seed: int, full: bool):
    gold = dict(weight_seed=seed)
    if full:
        n1, n32 = 64, 16
        p190 = bench_prompt(shape, 2025)
        p1334 = bench_prompt(shape, 5000, segments=10, ctx_text=30, ctx_frames=100)
        b32 = [bench_prompt(shape, 2025 + b) for b in range(32)]
    else:
        n1, n32 = 24, 8
        p190 = toy_prompt(shape, 11, 6, 5)
        p1334 = toy_prompt(shape, 12, 20, 60)
        b32 = [toy_prompt(shape, 100 + b, 6, 5) for b in range(5)]
    long_flavour = copy_long_flavour(shape, p1334[0].shape[0])
    gold["flavours"] = dict(s190=COPY_SHORT, b32=COPY_SHORT, s1334=long_flavour)
    sums = {}
    for flavour, runs in ((COPY_SHORT, (("s190", p190), ("b32", b32))), (long_flavour, (("s1334", p1334),))):
        layer, lag = C.copy_flavour_params(shape, flavour)
        w = C.make_weights(shape, seed=seed, flavour=flavour)
        names = [n for n in DECISIVE_CHECKSUM_NAMES + [f"backbone.layers.{layer}.attn.{t}_proj.weight" for t in ("q", "k", "v", "output")] if n in w]
        sums[flavour] = (names, torch.stack([w[n].view(torch.int16).to(torch.int64).sum() for n in names]))
        for tag in ("bf16", "fp8"):
            wts = w if tag == "bf16" else C.fp8_dequantized(w)
            for name, prompt in runs:
                if name == "b32":
                    toks, msks = torch.stack([p[0] for p in prompt]), torch.stack([p[1] for p in prompt])
                    g = free_run(shape, wts, toks, msks, n32, f"{flavour} {tag} b32")
                    for b in range(toks.shape[0]):
                        want = C.decisive_copy_expected_codes(shape, seed, toks[b], msks[b], n32, lag)
                        assert torch.equal(g["codes"][:, b].to(torch.int32), want), "the batched oracle left the trajectory the construction implies"
                else:
                    tok, msk = prompt
                    g = free_run(shape, wts, tok, msk, n1, f"{flavour} {tag} {name}")
                    want = C.decisive_copy_expected_codes(shape, seed, tok, msk, n1, lag)
                    assert torch.equal(g["codes"][:, 0].to(torch.int32), want), "the oracle left the trajectory the construction implies"
                    g["prompt_rows"] = tok.shape[0]
                    g["faults_changed"] = copy_fault_check(shape, wts, flavour, tok, msk, want)
                gold[f"{tag}_{name}"] = g
        del w
    gold["weight_checksums"] = sums
    return gold


# ---- position sweep on the bench checkpoint -------------------------------------------------------------------------------------------
POSSWEEP_S = (63, 64, 65, 511, 512, 766, 767, 768, 769, 775, 1023, 1024, 1535, 2046)
POSSWEEP_PROMPT = dict(seed=7000, segments=15, ctx_text=30, ctx_frames=100, gen_text=81)          # 15 x 131 + 81 = 2046 rows
CONSEC_S, CONSEC_FRAMES = 740, 64


def possweep_prompt(shape: C.CsmShape):
    a = POSSWEEP_PROMPT
    return bench_prompt(shape, a["seed"], a["segments"], a["ctx_text"], a["ctx_frames"], a["gen_text"])


@torch.inference_mode()
def possweep_golden(shape: C.CsmShape, weights, sizes, consec_s: int, consec_frames: int):
    """For each S: the first S rows of the prompt prefilled in one call (the reference's way: generator.py:283 with the whole prompt),
    the prompt frame, then two teacher-forced steps -- greedy, fed the bf16 oracle's own codes -- at positions S and S + 1 (one step when
    S + 1 is the last position).  Per (S, frame): codes [32], top-8 logits, margin [32], bf16-vs-fp32 gap [32]."""
    tok, msk = possweep_prompt(shape)
    assert tok.shape[0] == 2046
    w32 = {k: v.float() for k, v in weights.items()}
    out = dict(sizes=torch.tensor(sizes), per_size=[])
    for S in list(sizes) + [None]:
        t0 = time.time()
        n_frames = 3 if S is not None else consec_frames + 1
        S_ = S if S is not None else consec_s
        n_frames = min(n_frames, shape.backbone.max_seq_len - S_ + 1)
        g = frames_golden(shape, weights, (tok[:S_], msk[:S_]), n_frames, keep_full=False, with_fp32=True, quiet=True, w32=w32)
        del g["prompt_tokens"], g["prompt_mask"]
        g["rows"] = S_
        print(f"  possweep S={S_}: {n_frames} frames in {time.time() - t0:.0f}s, max gap {float(g['bf16_vs_fp32_gap'].max()):.4f}, "
              f"rows with margin < gap/2: {int((g['margin'] < 0.5 * g['bf16_vs_fp32_gap'].max()).sum())} of {g['margin'].numel()}", flush=True)
        if S is None:
            out["consecutive"] = g
        else:
            out["per_size"].append(g)
    return out


def sampler_cases():
    g = torch.Generator().manual_seed(77)
    V = 2051
    logits = (torch.randn(48, V, generator=g) * 3).to(torch.bfloat16)
    logits[3, 100:160] = logits[3].max()                       # a 60-way tie at the top
    logits[4, :] = 0.5                                         # all equal
    logits[5, 7] = 40.0                                        # one dominant logit
    kth = torch.topk(logits[6].float(), 50)[0][-1]
    logits[6, 2000:2010] = kth.to(torch.bfloat16)              # ties exactly at the kth value
    cases = []
    for (T, k) in ((0.9, 50), (0.7, 30), (1.0, 2051), (0.8, 1), (1.3, 5)):
        q = torch.empty(48, V, dtype=torch.bfloat16).exponential_(1, generator=g)
        out = C.sample_topk(logits, k, T, q=q, greedy_lowest_index=(k == 1))
        cases.append(dict(temperature=T, topk=k, noise=q, out=out[:, 0].clone()))
    return dict(logits=logits, cases=cases)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--frames", type=int, default=12)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    want = lambda n: (not a.only) or (n in a.only.split(","))

    if want("prompt"):
        shape = C.csm_tiny()
        tok, msk = toy_prompt(shape, 5, 4, 3)
        torch.save(dict(tokens=tok, mask=msk, seed=5, n_text=4, ctx_frames=3), os.path.join(OUT, "prompt_layout.pt"))
    if want("sampler"):
        torch.save(sampler_cases(), os.path.join(OUT, "sampler_cases.pt"))
    if want("tiny"):
        shape = C.csm_tiny()
        w = C.make_weights(shape, seed=1234)
        gold = frames_golden(shape, w, toy_prompt(shape, 11, 6, 5), 6, keep_full=True, with_fp32=True)
        gold.update(weight_seed=1234, prompt_seed=11)
        torch.save(gold, os.path.join(OUT, "tiny_frames.pt"))
    if want("mimi"):
        for name, s in (("tiny", M.mimi_tiny()), ("full", M.mimi_full())):
            w = M.make_weights(s, seed=4321)
            codes = torch.randint(0, s.codebook_size, (1, 32, 10 if name == "full" else 23),
                                  generator=torch.Generator().manual_seed(9))
            whole = M.decode(s, w, codes)
            chunks = M.decode_stateless_chunks(s, w, codes, 10)
            # keep every 16th sample of the full-size clip + exact head/tail windows (small file)
            torch.save(dict(codes=codes, weight_seed=4321, pcm=whole if name == "tiny" else None,
                            pcm_stride16=whole[..., ::16].clone(), pcm_head=whole[..., :4096].clone(),
                            pcm_tail=whole[..., -4096:].clone(),
                            chunks_stride16=chunks[..., ::16].clone(),
                            rms=whole.pow(2).mean().sqrt()), os.path.join(OUT, f"mimi_{name}.pt"))
    if want("tinydecisive"):
        torch.save(decisive_golden(C.csm_tiny(), 1234, full=False), os.path.join(OUT, "tiny_decisive.pt"))
    if want("decisive"):
        torch.save(decisive_golden(C.csm_1b(), 1234, full=True), os.path.join(OUT, "csm1b_decisive.pt"))
    if want("tinycopy"):
        torch.save(decisive_copy_golden(C.csm_tiny(), 1234, full=False), os.path.join(OUT, "tiny_decisive_copy.pt"))
    if want("copy"):
        torch.save(decisive_copy_golden(C.csm_1b(), 1234, full=True), os.path.join(OUT, "csm1b_decisive_copy.pt"))
    if a.only == "copyx":
        # more copy layers / lags on the 190-row prompt, added to the existing file: the FIRST layer with lag 1 (layer 0 normalises the embedding
        # sum itself; lag 1 reads the row the immediately preceding frame step appended -- from frame 2 on) and the LAST layer with lag 7 (its
        # finisher applies the stack's final norm)
        shape = C.csm_1b()
        path = os.path.join(OUT, "csm1b_decisive_copy.pt")
        gold = torch.load(path)
        seed = int(gold["weight_seed"])
        tok, msk = bench_prompt(shape, 2025)
        for name, flavour in (("s190_first", "decisive_copy:0:1"), ("s190_last", f"decisive_copy:{shape.backbone.num_layers - 1}:7")):
            layer, lag = C.copy_flavour_params(shape, flavour)
            w = C.make_weights(shape, seed=seed, flavour=flavour)
            names = [n for n in DECISIVE_CHECKSUM_NAMES + [f"backbone.layers.{layer}.attn.{t}_proj.weight" for t in ("q", "k", "v", "output")] if n in w]
            gold["weight_checksums"][flavour] = (names, torch.stack([w[n].view(torch.int16).to(torch.int64).sum() for n in names]))
            gold["flavours"][name] = flavour
            for tag in ("bf16", "fp8"):
                wts = w if tag == "bf16" else C.fp8_dequantized(w)
                g = free_run(shape, wts, tok, msk, 32, f"{flavour} {tag} {name}")
                want_codes = C.decisive_copy_expected_codes(shape, seed, tok, msk, 32, lag)
                assert torch.equal(g["codes"][:, 0].to(torch.int32), want_codes), "the oracle left the trajectory the construction implies"
                g["prompt_rows"] = tok.shape[0]
                g["faults_changed"] = copy_fault_check(shape, wts, flavour, tok, msk, want_codes, n=max(8, lag + 5))
                gold[f"{tag}_{name}"] = g
            del w
        torch.save(gold, path)
    if a.only == "copymany":
        # the twelve utterances of the continuously refilled batch of 8 (decisive_many_prompts) on the copy checkpoint: a refill writes a prompt's K/V
        # into ONE slot of a live batch, and every slot then steps at its own position -- what the free-running codes read back
        shape = C.csm_1b()
        path = os.path.join(OUT, "csm1b_decisive_copy.pt")
        gold = torch.load(path)
        seed = int(gold["weight_seed"])
        _, lag = C.copy_flavour_params(shape, COPY_SHORT)
        w = C.make_weights(shape, seed=seed, flavour=COPY_SHORT)
        many = []
        for i, ((tok, msk), lim) in enumerate(decisive_many_prompts(shape)):
            g = free_run(shape, w, tok, msk, lim, f"{COPY_SHORT} bf16 many[{i}] S={tok.shape[0]}")
            assert torch.equal(g["codes"][:, 0].to(torch.int32), C.decisive_copy_expected_codes(shape, seed, tok, msk, lim, lag))
            many.append(g["codes"][:, 0].clone())
        gold["bf16_many"] = many
        torch.save(gold, path)
    if a.only == "cfg4":
        # BASELINE config 4 = batch 256 sharded over 8 GPUs: 32 utterances per rank, rank r's prompts are bench.py's seeds 4000 + 32 r .. (batch32_leg).
        # No 8-GPU node has been available to this build; what CAN be checked on one GPU is that a shard -- here the LAST rank's, seeds 4224..4255 --
        # computes what the batched oracle computes for those prompts: top-8 logits / codes / margins of 2 teacher-forced frames + the bf16-vs-fp32 gap.
        shape = C.csm_1b()
        w = C.make_weights(shape, seed=1234)
        rank, seed0 = 7, 4000 + 32 * 7
        ps = [bench_prompt(shape, seed0 + b) for b in range(32)]
        toks, msks = torch.stack([p[0] for p in ps]), torch.stack([p[1] for p in ps])
        gold = frames_golden_batch(shape, w, toks, msks, 2)
        B = 32
        gaps = []
        with torch.inference_mode():
            m32 = C.OracleModel(shape, {k: v.float() for k, v in w.items()}, dtype=torch.float32); m32.setup_caches(B)
            cur_t, cur_m = toks, msks
            pos = torch.arange(toks.size(1)).unsqueeze(0).repeat(B, 1)
            mb = C.OracleModel(shape, w); mb.setup_caches(B)
            for f in range(2):
                tr, tr32 = C.FrameTrace(), C.FrameTrace()
                s_ = mb.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, trace=tr)
                assert torch.equal(s_, gold["codes"][f])
                m32.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, forced=s_, trace=tr32)
                gaps.append((torch.stack(tr.logits, 0).float() - torch.stack(tr32.logits, 0)).abs().max(dim=-1)[0])
                cur_t = torch.cat([s_.long(), torch.zeros(B, 1).long()], dim=1).unsqueeze(1)
                cur_m = torch.cat([torch.ones_like(s_).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
                pos = pos[:, -1:] + 1
        gold["bf16_vs_fp32_gap"] = torch.stack(gaps)
        gold.update(weight_seed=1234, rank=rank, prompt_seed=seed0, prompt_checksum=gold["prompt_tokens"].sum(dim=(1, 2)))
        del gold["prompt_tokens"], gold["prompt_mask"]
        torch.save(gold, os.path.join(OUT, "csm1b_cfg4_rank7.pt"))
    if want("possweep"):
        shape = C.csm_1b()
        w = C.make_weights(shape, seed=1234)
        gold = dict(weight_seed=1234, prompt=dict(POSSWEEP_PROMPT), consec_rows=CONSEC_S)
        gold["bf16"] = possweep_golden(shape, w, POSSWEEP_S, CONSEC_S, CONSEC_FRAMES)
        torch.save(gold, os.path.join(OUT, "csm1b_possweep.pt"))                 # (kept if the fp8 half is interrupted)
        gold["fp8"] = possweep_golden(shape, C.fp8_dequantized(w), POSSWEEP_S, CONSEC_S, CONSEC_FRAMES)
        torch.save(gold, os.path.join(OUT, "csm1b_possweep.pt"))
        del w
    if a.only == "decisivefinish":          # re-derive the checksums / PCM of existing files without re-running the trajectories
        for fname, shape, full in (("tiny_decisive.pt", C.csm_tiny(), False), ("csm1b_decisive.pt", C.csm_1b(), True)):
            path = os.path.join(OUT, fname)
            gold = torch.load(path)
            torch.save(decisive_finish(gold, C.make_weights(shape, seed=int(gold["weight_seed"]), flavour="decisive"), full), path)
    if want("csm1b"):
        shape = C.csm_1b()
        t0 = time.time()
        w = C.make_weights(shape, seed=1234)
        print(f"weights: {time.time() - t0:.0f}s", flush=True)
        gold = frames_golden(shape, w, toy_prompt(shape, 2025, 16, 0), a.frames, keep_full=False, with_fp32=True)
        gold.update(weight_seed=1234, prompt_seed=2025)
        torch.save(gold, os.path.join(OUT, "csm1b_frames.pt"))
    for name, fname, fp8 in (("cfg3gap", "csm1b_cfg3.pt", False), ("cfg5cgap", "csm1b_cfg5c.pt", True)):
        if not want(name):
            continue
        # The oracle's own bf16-vs-fp32 logit gap ON THE ROWS OF A BATCHED GOLDEN (32 utterances x 32 codebooks x 2 frames): the gap stored
        # with config 2 / config 5 is a maximum over 192 / 64 rows of ONE utterance, the batched tests compare 2,048 rows -- a maximum over
        # ten times the samples (round 4: tests/test_frame_gpu.py holds the B = 32 paths to 1x THIS gap).  Added to the existing file; its
        # codes must be reproduced on this host.
        shape = C.csm_1b()
        w = C.make_weights(shape, seed=1234)
        if fp8:
            w = C.fp8_dequantized(w)
        path = os.path.join(OUT, fname)
        gold = torch.load(path)
        if "prompt_tokens" in gold:
            toks, msks = gold["prompt_tokens"].long(), gold["prompt_mask"]
        else:
            ps = [bench_prompt(shape, int(gold["prompt_seed"]) + b, segments=10, ctx_text=30, ctx_frames=100) for b in range(gold["codes"].shape[1])]
            toks, msks = torch.stack([p[0] for p in ps]), torch.stack([p[1] for p in ps])
        B = toks.shape[0]
        gaps = []
        with torch.inference_mode():
            mb = C.OracleModel(shape, w); mb.setup_caches(B)
            m32 = C.OracleModel(shape, {k: v.float() for k, v in w.items()}, dtype=torch.float32); m32.setup_caches(B)
            cur_t, cur_m = toks, msks
            pos = torch.arange(toks.size(1)).unsqueeze(0).repeat(B, 1)
            for f in range(gold["codes"].shape[0]):
                t0 = time.time()
                tr, tr32 = C.FrameTrace(), C.FrameTrace()
                s = mb.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, trace=tr)
                assert torch.equal(s, gold["codes"][f]), "the stored trajectory is not reproduced on this host"
                m32.generate_frame(cur_t, cur_m, pos, 1.0, 1, greedy=True, forced=s, trace=tr32)
                lg, lg32 = torch.stack(tr.logits, 0).float(), torch.stack(tr32.logits, 0)
                gaps.append((lg - lg32).abs().max(dim=-1)[0])               # [32 codebooks][B]
                cur_t = torch.cat([s.long(), torch.zeros(B, 1).long()], dim=1).unsqueeze(1)
                cur_m = torch.cat([torch.ones_like(s).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
                pos = pos[:, -1:] + 1
                print(f"  {name} frame {f}: {time.time() - t0:.1f}s max {float(gaps[-1].max()):.4f}", flush=True)
        gold["bf16_vs_fp32_gap"] = torch.stack(gaps)
        torch.save(gold, path)
        del w, mb, m32
    if want("cfg2") or want("cfg3") or want("cfg5") or want("cfg5b") or want("cfg5c"):
        shape = C.csm_1b()
        w = C.make_weights(shape, seed=1234)
        if want("cfg2"):
            gold = frames_golden(shape, w, bench_prompt(shape, 2025), 6, keep_full=False, with_fp32=True)
            gold.update(weight_seed=1234, prompt_seed=2025)
            torch.save(gold, os.path.join(OUT, "csm1b_cfg2.pt"))
        if want("cfg3"):
            ps = [bench_prompt(shape, 2025 + b) for b in range(32)]
            gold = frames_golden_batch(shape, w, torch.stack([p[0] for p in ps]), torch.stack([p[1] for p in ps]), 2)
            gold.update(weight_seed=1234, prompt_seed=2025)
            gold["prompt_tokens"] = gold["prompt_tokens"].to(torch.int32)
            torch.save(gold, os.path.join(OUT, "csm1b_cfg3.pt"))
        if want("cfg5"):
            w8 = C.fp8_dequantized(w)
            del w
            g5 = frames_golden(shape, w8, bench_prompt(shape, 5000, segments=10, ctx_text=30, ctx_frames=100), 2, keep_full=False, with_fp32=True)
            tok, msk = bench_prompt(shape, 5001, segments=12, ctx_text=30, ctx_frames=100, gen_text=128)      # 12 x 131 + 128 = 1700 rows
            gl = frames_golden(shape, w8, (tok, msk), 2, keep_full=False, with_fp32=True)
            gold = dict(weight_seed=1234, s1334=g5, s1700=gl,
                        deq_checksum=torch.stack([w8["backbone.layers.3.mlp.w2.weight"].float().abs().sum(),
                                                  w8["decoder.layers.1.attn.q_proj.weight"].float().abs().sum(),
                                                  w8["audio_head"].float().abs().sum()]))
            for d in (g5, gl):
                d["prompt_tokens"] = d["prompt_tokens"].to(torch.int32)
            torch.save(gold, os.path.join(OUT, "csm1b_cfg5.pt"))
        if want("cfg5b"):
            # BASELINE config 5, batched (SURVEY.md 8d lists B = 32; B = 4 keeps the fixture small): 4 different 1334-row prompts,
            # fp8-dequantised weights, 2 teacher-forced frames of the BATCHED oracle
            w8 = C.fp8_dequantized(w)
            ps = [bench_prompt(shape, 6000 + b, segments=10, ctx_text=30, ctx_frames=100) for b in range(4)]
            gold = frames_golden_batch(shape, w8, torch.stack([p[0] for p in ps]), torch.stack([p[1] for p in ps]), 2)
            gold.update(weight_seed=1234, prompt_seed=6000)
            gold["prompt_tokens"] = gold["prompt_tokens"].to(torch.int32)
            torch.save(gold, os.path.join(OUT, "csm1b_cfg5b.pt"))
        if want("cfg5c"):
            # BASELINE config 5 at the batch SURVEY.md 8d lists for it: B = 32 x the 1334-row prompt (bench.py's config5_b32 leg: seeds
            # 6000..6031), fp8-dequantised weights, 2 teacher-forced frames of the batched oracle.  Only the top-8 logits, the codes and
            # the margins are kept (the prompts are a function of the seeds: the test rebuilds them with bench.synthetic_prompt).
            w8 = C.fp8_dequantized(w)
            ps = [bench_prompt(shape, 6000 + b, segments=10, ctx_text=30, ctx_frames=100) for b in range(32)]
            gold = frames_golden_batch(shape, w8, torch.stack([p[0] for p in ps]), torch.stack([p[1] for p in ps]), 2)
            gold.update(weight_seed=1234, prompt_seed=6000, prompt_checksum=gold["prompt_tokens"].sum(dim=(1, 2)))
            del gold["prompt_tokens"], gold["prompt_mask"]
            torch.save(gold, os.path.join(OUT, "csm1b_cfg5c.pt"))


if __name__ == "__main__":
    main()
