"""Helpers shared by the -m gpu parity tests: they call the HIP path through the C ABI
(sesameai._abi) and compare with the oracle on the same seeded inputs."""
import torch


def stream():
    return torch.cuda.current_stream().cuda_stream


def dev(t, dtype=None):
    t = t.to("cuda")
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous()


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """|a-b| in units of bf16 ULPs (sign-magnitude bit patterns mapped to ordered ints)."""
    def key(x):
        i = x.detach().cpu().to(torch.bfloat16).contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a) - key(b)).abs()


def assert_bf16_close(got: torch.Tensor, want: torch.Tensor, max_ulp: int = 2, min_exact: float = 0.9,
                      what: str = "", abs_floor: float = 0.0):
    """bf16 tensors: every element within ``max_ulp`` bf16 ULPs (or ``abs_floor`` absolute,
    for values straddling zero) and at least ``min_exact`` of them bit-identical.  Differences
    come only from fp32 summation order inside a dot product (HIP lane-strided + shuffles vs
    oneDNN blocks)."""
    got, want = got.detach().cpu(), want.detach().cpu()
    assert got.shape == want.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    ulp = bf16_ulp_diff(got, want)
    absd = (got.float() - want.float()).abs()
    bad = (ulp > max_ulp) & (absd > abs_floor)
    exact = (ulp == 0).float().mean().item()
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())} elements differ by > {max_ulp} ulp "
                                 f"(max ulp {int(ulp.max())}, max abs {absd.max().item():.4g})")
    assert exact >= min_exact, f"{what}: only {exact:.3f} bit-exact (< {min_exact})"
    return exact
