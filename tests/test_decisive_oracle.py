"""CPU side of the decisive synthetic checkpoint (oracle.csm_ref.decisive_weights): the product builds the same tensors with its own
generator, the oracle's free-running greedy codes are the ones the construction implies and the ones committed under tests/golden/,
and every stored trajectory was decided with room to spare (margin >= 4 x the oracle's bf16-vs-fp32 gap; tests/test_decisive_gpu.py
holds the HIP path to those codes bit for bit)."""
import os

import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_product_and_oracle_build_the_same_decisive_checkpoint():
    from oracle import csm_ref as C
    from sesameai.models import csm_tiny_args, synthetic_state_dict
    w = C.make_weights(C.csm_tiny(), seed=1234, flavour="decisive")
    sd = synthetic_state_dict(csm_tiny_args(), seed=1234, flavour="decisive")
    assert set(w) == set(sd) and all(torch.equal(w[k], sd[k]) for k in w)
    base = C.make_weights(C.csm_tiny(), seed=1234)
    same = [k for k in w if torch.equal(w[k], base[k])]
    assert any("q_proj" in k for k in same) and not any(k in same for k in ("codebook0_head.weight", "audio_head", "audio_embeddings.weight"))
    gold = torch.load(os.path.join(GOLD, "tiny_decisive.pt"))
    got = torch.stack([sd[k].view(torch.int16).to(torch.int64).sum() for k in gold["weight_checksum_names"]])
    assert torch.equal(got, gold["weight_checksum"])


def test_oracle_free_run_follows_the_construction_and_the_golden():
    from oracle import csm_ref as C
    from oracle.make_golden import toy_prompt
    shape = C.csm_tiny()
    gold = torch.load(os.path.join(GOLD, "tiny_decisive.pt"))
    tok, msk = toy_prompt(shape, 11, 6, 5)
    for dtype in ("bf16", "fp8"):
        w = C.make_weights(shape, seed=int(gold["weight_seed"]), flavour="decisive")
        if dtype == "fp8":
            w = C.fp8_dequantized(w)
        m = C.OracleModel(shape, w)
        m.setup_caches(1)
        n = 8
        frames = torch.cat(C.generate_codes(m, tok, msk, n * 80, 1.0, 1, greedy=True, max_seq_len=shape.backbone.max_seq_len))
        assert torch.equal(frames, C.decisive_expected_codes(shape, int(gold["weight_seed"]), int(tok[-1, 32]), n))
        assert torch.equal(frames, gold[f"{dtype}_s190"]["codes"][:n, 0].to(torch.int32))


def test_every_stored_decisive_trajectory_was_decided_with_room_to_spare():
    for fname, frames in (("tiny_decisive.pt", 24), ("csm1b_decisive.pt", 64)):
        gold = torch.load(os.path.join(GOLD, fname))
        for key, g in gold.items():
            if isinstance(g, dict) and "min_margin" in g:
                assert float(g["min_margin"].min()) >= 4.0 * float(g["max_gap"].max()), (fname, key)
                assert g["codes"].shape[0] >= (frames if g["codes"].shape[1] == 1 else 8)
                assert int(g["codes"].max()) < 2048 and not bool((g["codes"] == 0).all(dim=2).any())


# ---- the history-dependent form (round 6): the decision is read out of one backbone layer's KV cache ------------------------------
def test_product_and_oracle_build_the_same_copy_checkpoint():
    from oracle import csm_ref as C
    from sesameai.models import csm_tiny_args, synthetic_state_dict
    gold = torch.load(os.path.join(GOLD, "tiny_decisive_copy.pt"))
    for flavour in sorted(set(gold["flavours"].values())):
        w = C.make_weights(C.csm_tiny(), seed=int(gold["weight_seed"]), flavour=flavour)
        sd = synthetic_state_dict(csm_tiny_args(), seed=int(gold["weight_seed"]), flavour=flavour)
        assert set(w) == set(sd) and all(torch.equal(w[k], sd[k]) for k in w), flavour
        names, sums = gold["weight_checksums"][flavour]
        assert torch.equal(torch.stack([sd[k].view(torch.int16).to(torch.int64).sum() for k in names]), sums), flavour


def _tiny_copy_run(shape, w, tok, msk, n):
    from oracle import csm_ref as C
    m = C.OracleModel(shape, w)
    m.setup_caches(1)
    return torch.cat(C.generate_codes(m, tok, msk, n * 80, 1.0, 1, greedy=True, max_seq_len=shape.backbone.max_seq_len))


def test_copy_checkpoint_free_run_follows_the_construction_and_the_golden():
    from oracle import csm_ref as C
    from oracle.make_golden import toy_prompt
    shape = C.csm_tiny()
    gold = torch.load(os.path.join(GOLD, "tiny_decisive_copy.pt"))
    for name, prompt in (("s190", toy_prompt(shape, 11, 6, 5)), ("s1334", toy_prompt(shape, 12, 20, 60))):
        flavour = gold["flavours"][name]
        _, lag = C.copy_flavour_params(shape, flavour)
        w = C.make_weights(shape, seed=int(gold["weight_seed"]), flavour=flavour)
        n = 10
        frames = _tiny_copy_run(shape, w, prompt[0], prompt[1], n)
        assert torch.equal(frames, C.decisive_copy_expected_codes(shape, int(gold["weight_seed"]), prompt[0], prompt[1], n, lag)), name
        assert torch.equal(frames, gold[f"bf16_{name}"]["codes"][:n, 0].to(torch.int32)), name


def test_kv_cache_faults_move_the_copy_trajectory_and_not_the_memoryless_one():
    """What the round-5 decisive checkpoint could not see (VERDICT r5 missing #2) and the copy checkpoint does: faults
    injected into ONE backbone layer's KV cache of the ORACLE -- decode steps that do not append their K/V, a K rotated with the wrong
    position, a dropped key range, prompt rows zeroed after the prompt ran.  The memoryless checkpoint's free-running codes are the same
    under every one of them; the copy checkpoint's change under every one in its copy layer, and under none in another layer (the
    decision is read out of ONE layer's cache: the position sweep of tests/test_possweep_gpu.py holds the other layers' arithmetic)."""
    from oracle import csm_ref as C
    from oracle.make_golden import toy_prompt
    shape = C.csm_tiny()
    tok, msk = toy_prompt(shape, 12, 20, 60)
    S, n = tok.shape[0], 10
    try:
        for flavour in ("decisive", "decisive_copy:1:3"):
            layer, lag = 1, 3
            w = C.make_weights(shape, seed=1234, flavour=flavour)
            C.KV_FAULT = None
            base = _tiny_copy_run(shape, w, tok, msk, n)
            faults = [{"kind": "stale"}, {"kind": "shift_rope", "delta": 1}, {"kind": "shift_write", "delta": -1}, {"kind": "zero_prompt"},
                      {"kind": "drop_keys", "lo": S - 1 - lag, "hi": S + n - lag}]
            for fault in faults:
                for L in (layer, 1 - layer):
                    C.KV_FAULT = dict(fault, stack="backbone", layer=L)
                    moved = not torch.equal(_tiny_copy_run(shape, w, tok, msk, n), base)
                    assert moved == (flavour != "decisive" and L == layer), (flavour, L, fault)
    finally:
        C.KV_FAULT = None


def test_every_stored_copy_trajectory_was_decided_with_room_to_spare_and_saw_the_faults():
    for fname, frames in (("tiny_decisive_copy.pt", 24), ("csm1b_decisive_copy.pt", 32)):
        gold = torch.load(os.path.join(GOLD, fname))
        for key, g in gold.items():
            if isinstance(g, dict) and "min_margin" in g:
                assert float(g["min_margin"].min()) >= 4.0 * float(g["max_gap"].max()), (fname, key)
                assert g["codes"].shape[0] >= (frames if g["codes"].shape[1] == 1 else 8)
                assert int(g["codes"].max()) < 2048 and not bool((g["codes"] == 0).all(dim=2).any())
                if "faults_changed" in g:
                    assert bool((g["faults_changed"] > 0).all()), (fname, key)
