"""Position coverage, at CSM-1B shapes, of the kernels that carry 97 % of a B = 1 frame (VERDICT r5 missing #3): `k_bb_layer<false|true>`
(one backbone layer of a decode step per launch; csrc/bb_block.cuh switches from one CU per head to the key range split over 8 CUs at
768 keys), `k_dec_persist` (codebooks 2..31) and, round 6, `k_dec_first` (codebook 1) exist only for the CSM-1B shape, so no tiny-shape test reaches them, and until round 5 their logits were
compared with the oracle at about ten positions.  Reference arithmetic: sesameai/models.py:154-158 (backbone step over the position-indexed
cache), :160-182 (depth decoder).

Golden `csm1b_possweep.pt` (oracle/make_golden.py --only possweep; bench checkpoint, ONE 2046-row prompt):
  * cut at S in {63, 64, 65, 511, 512, 766, 767, 768, 769, 775, 1023, 1024, 1535, 2046}: the prompt frame (prefill kernels at that row
    count) and teacher-forced steps at positions S and S + 1 (so 767/768/769/770 keys, uneven 8-way splits, and 2046 / 2047 -- the last
    two positions of the cache -- are all compared);
  * 64 CONSECUTIVE teacher-forced steps after a 740-row prompt: positions 740..803, across the switch.
For every (S, frame): all 32 rows of logits (top-8 of the oracle) within 1 x the oracle's own bf16-vs-fp32 gap ON THAT CUT (a maximum over
its 96 rows: 0.088-0.141), greedy picks equal wherever the oracle's top-1 / top-2 margin exceeds 0.75 x the gap of the WHOLE SWEEP (the
maximum over its 3,424 rows, 0.141 bf16 / 0.127 fp8: the near-tie rule of tests/test_frame_gpu.py, whose noise floor is a maximum over a
golden's rows too; measured 0.55 x, see NEAR_TIE below); then the REPLAYED hipGraph step from the same
state: codes equal up to the first near-tie.  bf16 and the fp8-e4m3 weight stream (against the oracle on the dequantised weights)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
# A greedy pick may differ from the oracle's only where the ORACLE's top-1 / top-2 margin is at most NEAR_TIE x the sweep's gap.  Each of the two
# logits may move by 1 x gap, so 2 x is the hard limit; tests/test_frame_gpu.py holds 0.5 x on its ~200-row goldens (measured <= 0.42 x) and 1.0 x on
# its 2,048-row batched ones (measured 0.84 x); these 6,848 rows (bf16 + fp8) measured 0.55 x (fp8, p = 775, margin 0.0703), 0.44 x in bf16.
NEAR_TIE = 0.75


@pytest.fixture(scope="module")
def sweep():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from oracle import csm_ref as C
    from oracle.make_golden import possweep_prompt
    from sesameai.models import csm_1b_args, synthetic_state_dict
    gold = torch.load(os.path.join(GOLD, "csm1b_possweep.pt"))
    tok, msk = possweep_prompt(C.csm_1b())
    assert tok.shape[0] == 2046
    return gold, synthetic_state_dict(csm_1b_args(), seed=int(gold["weight_seed"])), tok, msk


def _row(codes):
    row = torch.zeros(1, 1, 33, dtype=torch.long); row[0, 0, :32] = codes.reshape(-1).long()
    rmask = torch.ones(1, 1, 33, dtype=torch.bool); rmask[0, 0, 32] = False
    return row, rmask


def _sweep_gap(G):
    return max([float(g["bf16_vs_fp32_gap"].max()) for g in G["per_size"]] + [float(G["consecutive"]["bf16_vs_fp32_gap"].max())])


def _compare(m, g, f, noise, what, stats, tie_noise):
    out, logits = m.depth(1, 1.0, 1, forced=g["codes"][f].reshape(1, -1), want_logits=True, commit=False)
    lg = logits[:, 0].float().cpu()
    d = (torch.gather(lg, 1, g["top_i"][f].long()) - g["top_v"][f].float()).abs().max().item()
    stats["worst"] = max(stats["worst"], d / noise)
    assert d <= noise, f"{what}: max|dlogit| {d:.4f} > the oracle's bf16-vs-fp32 gap {noise:.4f}"
    for cb in (out[0].cpu() != g["codes"][f].reshape(-1)).nonzero().flatten().tolist():
        margin = float(g["margin"][f, cb])
        stats["excused"].append((what, cb, margin / tie_noise, margin))
        assert margin <= NEAR_TIE * tie_noise, f"{what} codebook {cb}: greedy pick differs where the oracle's margin is {margin:.4f} = {margin / tie_noise:.2f} x the sweep's gap"
    stats["rows"] += 32


def _graph_step(m, g, f, S, tie_noise, what, stats):
    """the captured frame step on the state the golden frame f was computed from: codes equal the oracle's up to the first near-tie"""
    row, rmask = _row(g["codes"][f - 1])
    got = m.generate_frame(row, rmask, torch.tensor([[S + f - 1]]), 1.0, 1)[0].cpu()
    want = g["codes"][f].reshape(-1).to(got.dtype)
    diff = (got != want).nonzero().flatten()
    if diff.numel():
        first = int(diff[0])
        margin = float(g["margin"][f, first])
        assert margin <= NEAR_TIE * tie_noise, f"{what}: the graph step's codebook {first} differs where the oracle's margin is {margin / tie_noise:.2f} x the sweep's gap"
        stats["graph_rows"] += first
    else:
        stats["graph_rows"] += 32


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_logits_and_picks_across_positions_and_the_key_split_switch(sweep, dtype):
    from sesameai.models import Model, csm_1b_args
    gold, sd, tok, msk = sweep
    G = gold[dtype]
    m = Model(csm_1b_args(), sd, max_frames=80, max_prefill_rows=2048, weights_dtype=dtype)
    m.setup_caches(1)
    assert m.fast_paths() & 1, "the persistent depth decoder (k_dec_persist) is not in charge"
    assert m.fast_paths() & (8 if dtype == "bf16" else 16), "the one-launch backbone layer (k_bb_layer) is not in charge"
    assert m.fast_paths() & 32, "the one-launch first decoder step (k_dec_first) is not in charge"
    stats = dict(worst=0.0, excused=[], rows=0, graph_rows=0)
    tie_noise = _sweep_gap(G)
    for g in G["per_size"]:
        S, nf = int(g["rows"]), g["codes"].shape[0]
        noise = float(g["bf16_vs_fp32_gap"].max())
        m.reset_caches()
        m.prefill_prompt(tok[:S].unsqueeze(0), msk[:S].unsqueeze(0))
        _compare(m, g, 0, noise, f"{dtype} S={S} prompt frame", stats, tie_noise)
        for f in range(1, nf):
            row, rmask = _row(g["codes"][f - 1])
            m.prefill(row, rmask, torch.tensor([[S + f - 1]]))                 # the backbone's decode step at position S + f - 1 (k_bb_layer)
            _compare(m, g, f, noise, f"{dtype} S={S} step at p={S + f - 1}", stats, tie_noise)
        # ... and the replayed graph from the prompt's state
        m.reset_caches()
        m.prefill_prompt(tok[:S].unsqueeze(0), msk[:S].unsqueeze(0))
        m.depth(1, 1.0, 1, forced=g["codes"][0].reshape(1, -1), commit=True)
        for f in range(1, nf):
            _graph_step(m, g, f, S, tie_noise, f"{dtype} S={S} graph step at p={S + f - 1}", stats)
    print(f"\n[possweep] {dtype}: {len(G['per_size'])} cuts {[int(g['rows']) for g in G['per_size']]}: {stats['rows']} logit rows, worst max|dlogit| = "
          f"{stats['worst']:.2f} x the cut's gap; {len(stats['excused'])} picks excused as near-ties (largest oracle margin "
          f"{max([e[3] for e in stats['excused']], default=0.0):.4f} = {max([e[2] for e in stats['excused']], default=0.0):.2f} x the sweep's gap {tie_noise:.4f}); "
          f"graph steps: {stats['graph_rows']} decisions matched")
    assert len(stats["excused"]) <= 0.08 * stats["rows"]


@pytest.mark.parametrize("dtype", ["bf16", "fp8"])
def test_64_consecutive_steps_across_the_key_split_switch(sweep, dtype):
    from sesameai.models import Model, csm_1b_args
    gold, sd, tok, msk = sweep
    g = gold[dtype]["consecutive"]
    S, nf = int(g["rows"]), g["codes"].shape[0]
    assert S == 740 and nf == 65 and S < 768 < S + nf - 1
    noise = float(g["bf16_vs_fp32_gap"].max())
    m = Model(csm_1b_args(), sd, max_frames=80, max_prefill_rows=2048, weights_dtype=dtype)
    m.setup_caches(1)
    assert m.fast_paths() & 1 and m.fast_paths() & (8 if dtype == "bf16" else 16) and m.fast_paths() & 32
    stats = dict(worst=0.0, excused=[], rows=0, graph_rows=0)
    tie_noise = _sweep_gap(gold[dtype])
    m.prefill_prompt(tok[:S].unsqueeze(0), msk[:S].unsqueeze(0))
    _compare(m, g, 0, noise, f"{dtype} prompt frame", stats, tie_noise)
    for f in range(1, nf):
        row, rmask = _row(g["codes"][f - 1])
        m.prefill(row, rmask, torch.tensor([[S + f - 1]]))
        _compare(m, g, f, noise, f"{dtype} step at p={S + f - 1}", stats, tie_noise)
    m.reset_caches()
    m.prefill_prompt(tok[:S].unsqueeze(0), msk[:S].unsqueeze(0))
    m.depth(1, 1.0, 1, forced=g["codes"][0].reshape(1, -1), commit=True)
    for f in range(1, nf):
        _graph_step(m, g, f, S, tie_noise, f"{dtype} graph step at p={S + f - 1}", stats)
    print(f"\n[possweep] {dtype}: 64 consecutive steps p = {S}..{S + nf - 2}: worst max|dlogit| = {stats['worst']:.2f} x gap ({noise:.4f}); "
          f"{len(stats['excused'])} of {stats['rows']} picks excused as near-ties (largest oracle margin {max([e[3] for e in stats['excused']], default=0.0):.4f} = "
          f"{max([e[2] for e in stats['excused']], default=0.0):.2f} x the sweep's gap {tie_noise:.4f}); "
          f"graph steps: {stats['graph_rows']} of {32 * (nf - 1)} decisions matched before a near-tie")
    assert len(stats["excused"]) <= 0.08 * stats["rows"]
    assert stats["graph_rows"] >= 8 * (nf - 1)
