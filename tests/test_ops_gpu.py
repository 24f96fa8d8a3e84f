"""Per-op parity of the gfx950 kernels against the oracle, through the C ABI
(include/csm_hip_ops.h).  Every test compares on the same seeded inputs; bf16 results must
be within 2 bf16 ULP everywhere with >=90 % bit-identical (fp32 summation order is the only
freedom), integer results (sampler indices, embedding sums of <= 2 rows) bit-exact."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import assert_bf16_close, dev, stream


@pytest.fixture(scope="module")
def abi():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from sesameai import _abi
    return _abi


def _ck(abi, code):
    assert code == 0, abi.lib.csm_last_error(None)


def op_gemv(abi, kind, x, w0, *, w1=None, w2=None, norm_scale=None, eps=1e-5, resid=None, out=None, ldo=None,
            normed_out=None, x_row_stride=None, x_row_offset=0, M=None, N=None, nt=0, head_dim=64, nq=0, nkv=0,
            kv_heads=0, smax=0, rows_per_seq=1, pos=None, rope=None, kcache=None, vcache=None):
    K = w0.shape[1]
    M = M if M is not None else x.shape[0]
    N = N if N is not None else w0.shape[0]
    p = lambda t: t.data_ptr() if t is not None else None
    _ck(abi, abi.lib.csm_op_gemv(kind, M, K, N, p(x), x_row_stride if x_row_stride is not None else K, x_row_offset,
                                 p(norm_scale), eps, p(w0), p(w1), p(w2), p(resid), p(out), ldo if ldo is not None else N,
                                 p(normed_out), K, nt, head_dim, nq, nkv, kv_heads, smax, rows_per_seq, p(pos), p(rope),
                                 p(kcache), p(vcache), stream()))
    torch.cuda.synchronize()


def rnd(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("K,N", [(512, 1024), (1024, 2051), (2048, 2048), (8192, 1024)])
@pytest.mark.parametrize("M", [1, 2, 3, 5])
def test_linear_and_residual(abi, K, N, M):
    g = torch.Generator().manual_seed(K + N + M)
    x, w, r = rnd((M, K), g), rnd((N, K), g, 0.02), rnd((M, N), g)
    want = F.linear(x, w)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    op_gemv(abi, 0, dev(x), dev(w), out=out, nt=M & 1)
    assert_bf16_close(out, want, abs_floor=5e-5, what=f"linear K{K} N{N} M{M}")   # near-zero outputs: fp32 order noise ~3e-6
    res = dev(r)
    op_gemv(abi, 1, dev(x), dev(w), out=res, resid=res)            # in place, like the residual stream
    assert_bf16_close(res, want + r, abs_floor=0.016, what="linear+residual")   # y + r can cancel: 1 ulp of y


@pytest.mark.parametrize("K,M", [(512, 1), (1024, 2), (2048, 1), (2048, 4)])
def test_rmsnorm_head_and_side_output(abi, K, M):
    from oracle.csm_ref import rms_norm
    g = torch.Generator().manual_seed(K + M)
    S = 3                                                         # rows per sequence; only the last is used
    x, w = rnd((M * S, K), g, 2.0), rnd((2051, K), g, 0.02)
    scale = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16)
    xn = rms_norm(x.view(M, S, K)[:, -1], scale, 1e-5)
    want = F.linear(xn, w)
    out = torch.zeros(M, 2560, dtype=torch.bfloat16, device="cuda")
    normed = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    op_gemv(abi, 2, dev(x), dev(w), norm_scale=dev(scale), out=out, ldo=2560, normed_out=normed,
            x_row_stride=S * K, x_row_offset=(S - 1) * K, M=M)
    assert_bf16_close(normed, xn, max_ulp=1, min_exact=0.99, what="rmsnorm")
    assert_bf16_close(out[:, :2051], want, what="norm+head")


@pytest.mark.parametrize("d,H,KV,hd", [(2048, 32, 8, 64), (1024, 8, 2, 128), (512, 8, 2, 64), (512, 4, 2, 128)])
@pytest.mark.parametrize("rows_per_seq", [1, 2])
def test_qkv_rope_kvappend(abi, d, H, KV, hd, rows_per_seq):
    from oracle.csm_ref import LlamaShape, apply_rope, rms_norm, rope_table
    g = torch.Generator().manual_seed(d + H + rows_per_seq)
    B, smax = 2, 64
    M = B * rows_per_seq
    s = LlamaShape(1, H, KV, H * hd, 1024, max_seq_len=smax)
    table = rope_table(s)
    x = rnd((M, d), g)
    scale = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16)
    wq, wk, wv = rnd((H * hd, d), g, 0.02), rnd((KV * hd, d), g, 0.02), rnd((KV * hd, d), g, 0.02)
    pos = torch.tensor([[5 + t for t in range(rows_per_seq)], [40 + t for t in range(rows_per_seq)]])
    xn = rms_norm(x, scale, 1e-5).view(B, rows_per_seq, d)
    q = apply_rope(F.linear(xn, wq).view(B, rows_per_seq, H, hd), table, pos)
    k = apply_rope(F.linear(xn, wk).view(B, rows_per_seq, KV, hd), table, pos)
    v = F.linear(xn, wv).view(B, rows_per_seq, KV, hd)
    qout = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(B, KV, smax, hd, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros_like(kc)
    op_gemv(abi, 3, dev(x), dev(wq), w1=dev(wk), w2=dev(wv), norm_scale=dev(scale), out=qout, ldo=H * hd,
            N=(H + 2 * KV) * hd, head_dim=hd, nq=H * hd, nkv=KV * hd, kv_heads=KV, smax=smax,
            rows_per_seq=rows_per_seq, pos=dev(pos.reshape(-1), torch.int32), rope=dev(table), kcache=kc, vcache=vc)
    assert_bf16_close(qout.view(B, rows_per_seq, H, hd), q, abs_floor=0.008, what="q rope")   # x0*c - x1*s can cancel
    for b in range(B):
        for t in range(rows_per_seq):
            p = int(pos[b, t])
            assert_bf16_close(kc[b, :, p], k[b, t], abs_floor=0.008, what="k cache")
            assert_bf16_close(vc[b, :, p], v[b, t], what="v cache")
    written = torch.zeros(B, smax, dtype=torch.bool)
    for b in range(B):
        written[b, pos[b]] = True
    assert float(kc.cpu()[~written[:, None, :, None].expand_as(kc)].abs().max()) == 0.0, "stray KV writes"


@pytest.mark.parametrize("d,ffn,M", [(2048, 8192, 1), (1024, 8192, 2), (512, 1024, 3)])
def test_swiglu(abi, d, ffn, M):
    from oracle.csm_ref import rms_norm
    g = torch.Generator().manual_seed(d + ffn + M)
    x = rnd((M, d), g)
    scale = (1 + 0.1 * torch.randn(d, generator=g)).to(torch.bfloat16)
    w1, w3 = rnd((ffn, d), g, 0.05), rnd((ffn, d), g, 0.05)
    xn = rms_norm(x, scale, 1e-5)
    want = F.silu(F.linear(xn, w1)) * F.linear(xn, w3)
    out = torch.zeros(M, ffn, dtype=torch.bfloat16, device="cuda")
    op_gemv(abi, 4, dev(x), dev(w1), w1=dev(w3), norm_scale=dev(scale), out=out, N=ffn)
    assert_bf16_close(out, want, max_ulp=3, abs_floor=2e-3, what="swiglu")


@pytest.mark.parametrize("M", [16, 33, 190])
def test_wide_mfma_path(abi, M):
    """prefill / batched rows on the matrix cores (mm.cuh): linear, +residual, q/k/v + RoPE + KV
    append and SwiGLU against the same oracle expressions as the GEMV path."""
    from oracle.csm_ref import LlamaShape, apply_rope, rope_table
    g = torch.Generator().manual_seed(M)
    d, H, KV, hd, ffn, smax = 1024, 8, 2, 128, 2048, 256
    x = rnd((M, d), g)
    w, r = rnd((2051, d), g, 0.02), rnd((M, 2051), g)
    out = torch.zeros(M, 2560, dtype=torch.bfloat16, device="cuda")
    op_gemv(abi, 10, dev(x), dev(w), out=out, ldo=2560)
    assert_bf16_close(out[:, :2051], F.linear(x, w), abs_floor=5e-5, what="mfma linear (N=2051 tail)")
    res = dev(r)
    op_gemv(abi, 11, dev(x), dev(w), out=res, resid=res)
    assert_bf16_close(res, F.linear(x, w) + r, abs_floor=0.016, what="mfma linear+residual")   # y + r can cancel: 1 ulp of y
    w1, w3 = rnd((ffn, d), g, 0.05), rnd((ffn, d), g, 0.05)
    act = torch.zeros(M, ffn, dtype=torch.bfloat16, device="cuda")
    op_gemv(abi, 14, dev(x), dev(w1), w1=dev(w3), out=act, N=ffn)
    assert_bf16_close(act, F.silu(F.linear(x, w1)) * F.linear(x, w3), max_ulp=3, abs_floor=2e-3, what="mfma swiglu")
    S = M                                                           # one sequence of M rows at positions 3..3+M-1
    s_ = LlamaShape(1, H, KV, H * hd, ffn, max_seq_len=smax)
    table = rope_table(s_)
    wq, wk, wv = rnd((H * hd, d), g, 0.02), rnd((KV * hd, d), g, 0.02), rnd((KV * hd, d), g, 0.02)
    pos = torch.arange(3, 3 + M).unsqueeze(0)
    q = apply_rope(F.linear(x, wq).view(1, S, H, hd), table, pos)
    k = apply_rope(F.linear(x, wk).view(1, S, KV, hd), table, pos)
    v = F.linear(x, wv).view(1, S, KV, hd)
    qout = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(1, KV, smax, hd, dtype=torch.bfloat16, device="cuda"); vc = torch.zeros_like(kc)
    op_gemv(abi, 13, dev(x), dev(wq), w1=dev(wk), w2=dev(wv), out=qout, ldo=H * hd, N=(H + 2 * KV) * hd, head_dim=hd,
            nq=H * hd, nkv=KV * hd, kv_heads=KV, smax=smax, rows_per_seq=S, pos=dev(pos.reshape(-1), torch.int32),
            rope=dev(table), kcache=kc, vcache=vc)
    assert_bf16_close(qout.view(1, S, H, hd), q, abs_floor=0.008, what="mfma q rope")     # x0*c - x1*s can cancel: 1 ulp of the inputs
    assert_bf16_close(kc[0, :, 3:3 + M].transpose(0, 1), k[0], abs_floor=0.008, what="mfma k cache")
    assert_bf16_close(vc[0, :, 3:3 + M].transpose(0, 1), v[0], abs_floor=5e-5, what="mfma v cache")


@pytest.mark.parametrize("M,K,hd", [(200, 1024, 128), (513, 2048, 64), (130, 512, 64), (129, 8192, 64)])
def test_gemm128_equals_mm32_bitwise(abi, M, K, hd):
    """The 128 x 128 LDS-tiled kernel of long prompts (kinds 20+) must give the SAME BITS as the 32 x 32 kernel
    (kinds 10+): same MFMA k-mapping, same chain order, same association of the four K-quarter partials.  That is
    what keeps a prompt row independent of how many rows share its prefill call (prefix-KV reuse)."""
    from oracle.csm_ref import LlamaShape, rope_table
    g = torch.Generator().manual_seed(M + K)
    H, KV, smax = (8, 2, 640) if hd == 128 else (16, 4, 640)
    x = dev(rnd((M, K), g))
    w = dev(rnd((2051, K), g, 0.02))
    o1 = torch.zeros(M, 2560, dtype=torch.bfloat16, device="cuda"); o2 = torch.zeros_like(o1)
    op_gemv(abi, 10, x, w, out=o1, ldo=2560); op_gemv(abi, 20, x, w, out=o2, ldo=2560)
    assert torch.equal(o1, o2), "linear"
    assert_bf16_close(o2[:, :2051], F.linear(x.cpu(), w.cpu()), abs_floor=2e-4, what="gemm128 linear vs oracle")
    r = rnd((M, 2051), g)
    r1, r2 = dev(r), dev(r)
    op_gemv(abi, 11, x, w, out=r1, resid=r1); op_gemv(abi, 21, x, w, out=r2, resid=r2)
    assert torch.equal(r1, r2), "linear + residual"
    ffn = 1088                                                     # not a multiple of 64: exercises the column tail
    w1, w3 = dev(rnd((ffn, K), g, 0.05)), dev(rnd((ffn, K), g, 0.05))
    a1 = torch.zeros(M, ffn, dtype=torch.bfloat16, device="cuda"); a2 = torch.zeros_like(a1)
    op_gemv(abi, 14, x, w1, w1=w3, out=a1, N=ffn); op_gemv(abi, 24, x, w1, w1=w3, out=a2, N=ffn)
    assert torch.equal(a1, a2), "swiglu"
    table = dev(rope_table(LlamaShape(1, H, KV, H * hd, 1024, max_seq_len=smax)))
    wq, wk, wv = dev(rnd((H * hd, K), g, 0.02)), dev(rnd((KV * hd, K), g, 0.02)), dev(rnd((KV * hd, K), g, 0.02))
    pos = dev(torch.arange(5, 5 + M), torch.int32)
    outs = []
    for kind in (13, 23):
        q = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
        kc = torch.zeros(1, KV, smax, hd, dtype=torch.bfloat16, device="cuda"); vc = torch.zeros_like(kc)
        op_gemv(abi, kind, x, wq, w1=wk, w2=wv, out=q, ldo=H * hd, N=(H + 2 * KV) * hd, head_dim=hd, nq=H * hd,
                nkv=KV * hd, kv_heads=KV, smax=smax, rows_per_seq=M, pos=pos, rope=table, kcache=kc, vcache=vc)
        outs.append((q, kc, vc))
    for t1, t2, what in zip(outs[0], outs[1], ("q", "k cache", "v cache")):
        assert torch.equal(t1, t2), what
    assert outs[1][1].abs().sum() > 0


def test_gemm128_with_256_row_blocks_and_lds_dma_operands_equals_mm32_bitwise():
    """The opt-in 256 x 128 form of k_gemm128 (operands by global_load_lds_dwordx4 into three LDS buffers; off by default because it
    measured slower) must stay bit-identical too.  Its row threshold is read when the library is loaded, so the bitwise test above
    is re-run in a child process with CSM_G256_MIN_ROWS=128: all four shapes then take the 256-row blocks (row tails 200, 513, 130, 129)."""
    import os, subprocess, sys
    env = dict(os.environ, CSM_G256_MIN_ROWS="128")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_ops_gpu.py"), "-q", "-m", "gpu", "-x",
                        "-k", "test_gemm128_equals_mm32_bitwise", "-p", "no:cacheprovider"], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("M,K,hd", [(64, 2048, 64), (96, 2048, 64), (190, 2048, 64), (190, 8192, 64), (131, 1024, 128), (256, 2048, 64)])
def test_prompt_kernels_give_the_same_bits_at_every_row_count(abi, M, K, hd):
    """Prompts of 64..256 rows run k_mmt / k_mmq (several 32 x 32 output tiles per wave, residual projections as four
    K-quarter slabs; kinds 31/33/34) instead of k_mm32 (kinds 11/13/14).  Same MFMA chains, same fold: SAME BITS --
    a prompt row must not depend on how many rows share its prefill call."""
    from oracle.csm_ref import LlamaShape, rope_table
    g = torch.Generator().manual_seed(7 * M + K)
    H, KV, smax = (8, 2, 640) if hd == 128 else (16, 4, 640)
    x = dev(rnd((M, K), g))
    N = 1024
    w = dev(rnd((N, K), g, 0.02))
    r = rnd((M, N), g)
    r1, r2 = dev(r), dev(r)
    op_gemv(abi, 11, x, w, out=r1, resid=r1); op_gemv(abi, 31, x, w, out=r2, resid=r2)
    assert torch.equal(r1, r2), "linear + residual (quarter slabs + finisher)"
    ffn = 1152                                                     # a multiple of 64 (the kernels' column group)
    w1, w3 = dev(rnd((ffn, K), g, 0.05)), dev(rnd((ffn, K), g, 0.05))
    a1 = torch.zeros(M, ffn, dtype=torch.bfloat16, device="cuda"); a2 = torch.zeros_like(a1)
    op_gemv(abi, 14, x, w1, w1=w3, out=a1, N=ffn); op_gemv(abi, 34, x, w1, w1=w3, out=a2, N=ffn)
    assert torch.equal(a1, a2), "swiglu"
    assert a2.float().abs().sum() > 0
    table = dev(rope_table(LlamaShape(1, H, KV, H * hd, 1024, max_seq_len=smax)))
    wq, wk, wv = dev(rnd((H * hd, K), g, 0.02)), dev(rnd((KV * hd, K), g, 0.02)), dev(rnd((KV * hd, K), g, 0.02))
    pos = dev(torch.arange(5, 5 + M), torch.int32)
    outs = []
    for kind in (13, 33):
        q = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
        kc = torch.zeros(1, KV, smax, hd, dtype=torch.bfloat16, device="cuda"); vc = torch.zeros_like(kc)
        op_gemv(abi, kind, x, wq, w1=wk, w2=wv, out=q, ldo=H * hd, N=(H + 2 * KV) * hd, head_dim=hd, nq=H * hd,
                nkv=KV * hd, kv_heads=KV, smax=smax, rows_per_seq=M, pos=pos, rope=table, kcache=kc, vcache=vc)
        outs.append((q, kc, vc))
    for t1, t2, what in zip(outs[0], outs[1], ("q", "k cache", "v cache")):
        assert torch.equal(t1, t2), what
    assert outs[1][1].abs().sum() > 0


@pytest.mark.parametrize("H,KV,hd,nsplit", [(32, 8, 64, 1), (32, 8, 64, 8), (8, 2, 128, 1), (8, 2, 128, 3)])
def test_attention(abi, H, KV, hd, nsplit):
    g = torch.Generator().manual_seed(H + hd + nsplit)
    B, S, smax = 2, 3, 300
    M = B * S
    q = rnd((B, S, H, hd), g)
    kc, vc = rnd((B, KV, smax, hd), g), rnd((B, KV, smax, hd), g)
    pos = torch.tensor([[0, 1, 2], [210, 211, 212]])
    rep = H // KV
    kk = kc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    vv = vc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    mask = torch.arange(smax)[None, None, :] <= pos[:, :, None]                  # (B,S,smax)
    want = F.scaled_dot_product_attention(q.transpose(1, 2), kk, vv, attn_mask=mask[:, None]).transpose(1, 2)
    out = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(M * H * nsplit * ((hd + 2 + 31) // 32 * 32), dtype=torch.float32, device="cuda")      # ATTN_PS(hd): partial rows are whole 128-byte lines
    qd, kd, vd, pd = dev(q), dev(kc), dev(vc), dev(pos.reshape(-1), torch.int32)    # keep the device copies alive
    _ck(abi, abi.lib.csm_op_attn(M, S, H, KV, hd, smax, nsplit, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                 pd.data_ptr(), out.data_ptr(), part.data_ptr(), stream()))
    torch.cuda.synchronize()
    # The oracle here is torch's CPU flash kernel on bf16 (itself approximate), so grade both
    # against exact attention computed in fp64 from the same bf16 inputs: the HIP kernel
    # (fp32 scores / softmax / PV, one bf16 rounding) must be within 1 bf16 ulp of exact and
    # no farther from the oracle than the oracle is from exact (+1 ulp).
    sc = (q.double().transpose(1, 2) @ kk.double().transpose(-1, -2)) / hd ** 0.5
    exact = (sc.masked_fill(~mask[:, None], float("-inf")).softmax(-1) @ vv.double()).transpose(1, 2)
    got = out.cpu().float().view(B, S, H, hd).double()
    err_hip = (got - exact).abs().max().item()
    err_ref = (want.double() - exact).abs().max().item()
    diff = (got - want.double()).abs().max().item()
    print(f"attention H{H} hd{hd} nsplit{nsplit}: |hip-exact|={err_hip:.4g} |oracle-exact|={err_ref:.4g} |hip-oracle|={diff:.4g}")
    ulp = 2.0 ** -8 * max(1.0, exact.abs().max().item())
    assert err_hip <= ulp, f"attention error vs exact {err_hip}"
    assert diff <= err_ref + err_hip + 1e-6


@pytest.mark.parametrize("nsplit", [1, 8])
def test_decode_attention_over_long_key_ranges(abi, nsplit):
    """p in {0, 1, 63, 1500} -- plus the last tile boundary (1023) and the last slot (2047) of the
    2048-position backbone cache, with the B=1 split-K form (8 key ranges merged by their softmax states) and the
    single-block form.  Graded against exact fp64 attention and the oracle's SDPA like test_attention."""
    H, KV, hd, smax = 32, 8, 64, 2048
    g = torch.Generator().manual_seed(1000 + nsplit)
    ps = [0, 1, 63, 1023, 1500, 2047]
    B = len(ps)
    q = rnd((B, 1, H, hd), g)
    kc, vc = rnd((B, KV, smax, hd), g), rnd((B, KV, smax, hd), g)
    pos = torch.tensor(ps)[:, None]
    for b, p in enumerate(ps):                                   # slots past the position hold garbage, NaN bit patterns included
        kc[b, :, p + 1:] = float("nan"); vc[b, :, p + 1:] = float("nan")
    rep = H // KV
    kz, vz = torch.nan_to_num(kc, nan=0.0), torch.nan_to_num(vc, nan=0.0)
    kk = kz.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    vv = vz.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    mask = torch.arange(smax)[None, None, :] <= pos[:, :, None]
    want = F.scaled_dot_product_attention(q.transpose(1, 2), kk, vv, attn_mask=mask[:, None]).transpose(1, 2)
    sc = (q.double().transpose(1, 2) @ kk.double().transpose(-1, -2)) / hd ** 0.5
    exact = (sc.masked_fill(~mask[:, None], float("-inf")).softmax(-1) @ vv.double()).transpose(1, 2)
    out = torch.zeros(B, H * hd, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(B * H * nsplit * ((hd + 2 + 31) // 32 * 32), dtype=torch.float32, device="cuda")
    qd, kd, vd, pd = dev(q), dev(kc), dev(vc), dev(pos.reshape(-1), torch.int32)
    _ck(abi, abi.lib.csm_op_attn(B, 1, H, KV, hd, smax, nsplit, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(),
                                 pd.data_ptr(), out.data_ptr(), part.data_ptr(), stream()))
    torch.cuda.synchronize()
    got = out.cpu().float().view(B, 1, H, hd).double()
    assert torch.isfinite(got).all()
    for b, p in enumerate(ps):
        err_hip = (got[b] - exact[b]).abs().max().item()
        err_ref = (want[b].double() - exact[b]).abs().max().item()
        ulp = 2.0 ** -8 * max(1.0, exact[b].abs().max().item())
        print(f"decode attention p={p} nsplit={nsplit}: |hip-exact|={err_hip:.4g} |oracle-exact|={err_ref:.4g}")
        assert err_hip <= ulp, (p, err_hip)
        assert (got[b] - want[b].double()).abs().max().item() <= err_ref + err_hip + 1e-6, p


@pytest.mark.parametrize("S,start", [(1334, 0), (200, 1500), (48, 2000)])
def test_prompt_flash_attention_over_long_prompts(abi, S, start):
    """k_attn_flash over >= 1334 keys (BASELINE config 5's prompt) and over the tail of the 2048-slot cache: the online
    softmax walks up to 64 key tiles; same grading as the short-prompt test."""
    H, KV, hd, smax = 32, 8, 64, 2048
    g = torch.Generator().manual_seed(S + start)
    q = rnd((1, S, H, hd), g)
    kc, vc = rnd((1, KV, smax, hd), g), rnd((1, KV, smax, hd), g)
    pos = start + torch.arange(S)[None, :]
    kc[0, :, start + S:] = float("nan"); vc[0, :, start + S:] = float("nan")
    rep = H // KV
    kz, vz = torch.nan_to_num(kc, nan=0.0), torch.nan_to_num(vc, nan=0.0)
    kk = kz.unsqueeze(2).expand(1, KV, rep, smax, hd).reshape(1, H, smax, hd)
    vv = vz.unsqueeze(2).expand(1, KV, rep, smax, hd).reshape(1, H, smax, hd)
    mask = torch.arange(smax)[None, None, :] <= pos[:, :, None]
    want = F.scaled_dot_product_attention(q.transpose(1, 2), kk, vv, attn_mask=mask[:, None]).transpose(1, 2)
    sc = (q.double().transpose(1, 2) @ kk.double().transpose(-1, -2)) / hd ** 0.5
    exact = (sc.masked_fill(~mask[:, None], float("-inf")).softmax(-1) @ vv.double()).transpose(1, 2)
    out = torch.zeros(S, H * hd, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(16, dtype=torch.float32, device="cuda")
    qd, kd, vd, pd = dev(q), dev(kc), dev(vc), dev(pos.reshape(-1), torch.int32)
    _ck(abi, abi.lib.csm_op_attn(S, S, H, KV, hd, smax, 0, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), pd.data_ptr(),
                                 out.data_ptr(), part.data_ptr(), stream()))
    torch.cuda.synchronize()
    got = out.cpu().float().view(1, S, H, hd).double()
    assert torch.isfinite(got).all()
    err_hip = (got - exact).abs().max().item()
    err_ref = (want.double() - exact).abs().max().item()
    print(f"flash attention S{S} from {start}: |hip-exact|={err_hip:.4g} |oracle-exact|={err_ref:.4g}")
    ulp = 2.0 ** -8 * max(1.0, exact.abs().max().item())
    assert err_hip <= 1.5 * ulp
    assert (got - want.double()).abs().max().item() <= err_ref + err_hip + 1e-6


@pytest.mark.parametrize("H,KV,S,start", [(32, 8, 77, (0, 5, 100)), (8, 2, 33, (0, 0, 190)), (16, 4, 1, (0, 63, 64)),
                                          (32, 8, 190, (0,) * 32)])        # (config 3's prefill: 1,536 blocks, several per CU)
def test_prompt_flash_attention_vs_exact_and_oracle(abi, H, KV, S, start):
    """attn_flash.cuh (prompt rows, hd 64): S^T = K.Q^T and O^T += V^T.P^T on the matrix cores, fp32 softmax, P rounded
    to bf16 like torch's CPU flash kernel.  Graded like test_attention: against exact fp64 attention of the same bf16
    inputs and against the oracle's SDPA."""
    hd, smax = 64, 320
    g = torch.Generator().manual_seed(H + S)
    B = len(start)
    M = B * S
    q = rnd((B, S, H, hd), g)
    kc, vc = rnd((B, KV, smax, hd), g), rnd((B, KV, smax, hd), g)
    pos = torch.tensor(start)[:, None] + torch.arange(S)[None, :]
    # cache rows that a prompt has not written yet hold arbitrary bits, NaN patterns included: they must not leak
    for b in range(B):
        kc[b, :, int(pos[b].max()) + 1:] = float("nan"); vc[b, :, int(pos[b].max()) + 1:] = float("nan")
    rep = H // KV
    kk = kc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    vv = vc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    mask = torch.arange(smax)[None, None, :] <= pos[:, :, None]
    kz, vz = torch.nan_to_num(kk, nan=0.0), torch.nan_to_num(vv, nan=0.0)
    want = F.scaled_dot_product_attention(q.transpose(1, 2), kz, vz, attn_mask=mask[:, None]).transpose(1, 2)
    sc = (q.double().transpose(1, 2) @ kz.double().transpose(-1, -2)) / hd ** 0.5
    exact = (sc.masked_fill(~mask[:, None], float("-inf")).softmax(-1) @ vz.double()).transpose(1, 2)
    out = torch.zeros(M, H * hd, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(16, dtype=torch.float32, device="cuda")
    qd, kd, vd, pd = dev(q), dev(kc), dev(vc), dev(pos.reshape(-1), torch.int32)
    _ck(abi, abi.lib.csm_op_attn(M, S, H, KV, hd, smax, 0, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), pd.data_ptr(),
                                 out.data_ptr(), part.data_ptr(), stream()))
    torch.cuda.synchronize()
    got = out.cpu().float().view(B, S, H, hd).double()
    assert torch.isfinite(got).all()
    err_hip = (got - exact).abs().max().item()
    err_ref = (want.double() - exact).abs().max().item()
    print(f"flash attention H{H} S{S}: |hip-exact|={err_hip:.4g} |oracle-exact|={err_ref:.4g}")
    ulp = 2.0 ** -8 * max(1.0, exact.abs().max().item())
    assert err_hip <= 1.5 * ulp, f"flash attention error vs exact {err_hip}"          # bf16 P adds up to half an ulp
    assert (got - want.double()).abs().max().item() <= err_ref + err_hip + 1e-6


def test_prompt_flash_attention_rows_do_not_depend_on_the_tiling(abi):
    """The same prompt rows attended in one call of 90 rows, or as the last 37 rows on their own (different 32-row
    tiles, fewer key tiles walked by the early blocks): identical bits.  This is what lets a warm prefill of a few new
    rows reproduce a cold prefill exactly."""
    H, KV, hd, smax, S = 32, 8, 64, 256, 90
    g = torch.Generator().manual_seed(9)
    q = dev(rnd((S, H, hd), g))
    kc, vc = dev(rnd((1, KV, smax, hd), g)), dev(rnd((1, KV, smax, hd), g))
    pos = dev(torch.arange(7, 7 + S), torch.int32)
    part = torch.zeros(16, dtype=torch.float32, device="cuda")

    def run(first, n):
        out = torch.zeros(n, H * hd, dtype=torch.bfloat16, device="cuda")
        qs, ps = q[first:first + n].contiguous(), pos[first:first + n].contiguous()
        _ck(abi, abi.lib.csm_op_attn(n, n, H, KV, hd, smax, 0, qs.data_ptr(), kc.data_ptr(), vc.data_ptr(), ps.data_ptr(),
                                     out.data_ptr(), part.data_ptr(), stream()))
        torch.cuda.synchronize()
        return out

    whole = run(0, S)
    assert torch.equal(whole[53:], run(53, 37))
    assert torch.equal(whole[89:], run(89, 1))
    assert torch.equal(whole[:5], run(0, 5))


@pytest.mark.parametrize("H,KV,rows_per_seq", [(8, 2, 1), (8, 2, 2), (4, 2, 1)])
def test_fused_decoder_attention_oproj(abi, H, KV, rows_per_seq):
    """depth-decoder "SDPA + output_proj + residual" fused into one kernel (hd 128, <= 32 keys)."""
    g = torch.Generator().manual_seed(H + rows_per_seq)
    B, hd, smax = 2, 128, 32
    M, d = B * rows_per_seq, H * hd
    q = rnd((B, rows_per_seq, H, hd), g)
    kc, vc = rnd((B, KV, smax, hd), g), rnd((B, KV, smax, hd), g)
    wo, res = rnd((d, d), g, 0.02), rnd((M, d), g)
    pos = torch.tensor([[0 + t for t in range(rows_per_seq)], [30 + t for t in range(rows_per_seq)]])
    rep = H // KV
    kk = kc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    vv = vc.unsqueeze(2).expand(B, KV, rep, smax, hd).reshape(B, H, smax, hd)
    mask = torch.arange(smax)[None, None, :] <= pos[:, :, None]
    att = F.scaled_dot_product_attention(q.transpose(1, 2), kk, vv, attn_mask=mask[:, None]).transpose(1, 2).reshape(M, d)
    want = F.linear(att, wo) + res
    qd, kd, vd, pd, wd = dev(q), dev(kc), dev(vc), dev(pos.reshape(-1), torch.int32), dev(wo)
    out = dev(res)
    _ck(abi, abi.lib.csm_op_attn_oproj(M, rows_per_seq, H, KV, smax, qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), pd.data_ptr(),
                                       wd.data_ptr(), d, out.data_ptr(), out.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert_bf16_close(out, want, max_ulp=3, min_exact=0.8, abs_floor=0.02, what="fused attn+oproj")


def test_embed_sum(abi):
    from oracle.csm_ref import OracleModel, csm_tiny, make_weights
    shape = csm_tiny()
    w = make_weights(shape)
    m = OracleModel(shape, w)
    g = torch.Generator().manual_seed(3)
    B, S = 2, 5
    tok = torch.zeros(B, S, 33, dtype=torch.long)
    msk = torch.zeros(B, S, 33, dtype=torch.bool)
    tok[:, :2, 32] = torch.randint(0, shape.text_vocab_size, (B, 2), generator=g); msk[:, :2, 32] = True
    tok[:, 2:, :32] = torch.randint(0, 2051, (B, 3, 32), generator=g); msk[:, 2:, :32] = True
    want = m.embed_frame(tok, msk)
    d = shape.backbone.embed_dim
    h = torch.zeros(B * S, d, dtype=torch.bfloat16, device="cuda")
    td, md = dev(tok.view(-1, 33), torch.int32), dev(msk.view(-1, 33), torch.uint8)
    te, ae = dev(w["text_embeddings.weight"]), dev(w["audio_embeddings.weight"])
    _ck(abi, abi.lib.csm_op_embed_sum(B * S, 32, d, 2051, shape.text_vocab_size, td.data_ptr(), md.data_ptr(),
                                      te.data_ptr(), ae.data_ptr(), h.data_ptr(), stream()))
    torch.cuda.synchronize()
    assert_bf16_close(h.view(B, S, d), want, max_ulp=1, min_exact=0.999, what="embed sum")



def _assert_sampler_misses_are_one_ulp_ties(logits, T, k, noise, got, want, what):
    """Where the HIP sampler's pick differs from the oracle's, the oracle's own deciding quantity r = bf16(p / q) of the
    two picks must be within ONE bf16 ulp (the fp32 order of the <= k exp-sums can move a probability by one ulp and
    nothing else is free); anything else is a real disagreement."""
    import torch.nn.functional as F
    from gpu_util import bf16_ulp_diff
    bad = (got != want).nonzero().flatten()
    if bad.numel() == 0:
        return 0
    l = logits[bad] / T
    kth = torch.topk(l, k)[0][..., -1, None]
    probs = F.softmax(F.log_softmax(l.masked_fill(l < kth, -float("inf")), dim=-1), dim=-1)
    r = probs / noise[bad]
    r_got = r[torch.arange(bad.numel()), got[bad].long()]
    r_want = r[torch.arange(bad.numel()), want[bad].long()]
    ulps = bf16_ulp_diff(r_got, r_want)
    assert int(ulps.max()) <= 1, f"{what}: a pick differs by {int(ulps.max())} ulp of p/q (rows {bad.tolist()})"
    assert bool((probs[torch.arange(bad.numel()), got[bad].long()] > 0).all()), f"{what}: picked a removed index"
    return int(bad.numel())

def test_sampler_golden(abi):
    """tests/golden/sampler_cases.pt: oracle sample_topk with supplied Exp(1) noise, incl. ties at
    the top, ties at the kth value, all-equal rows, topk==1 and topk==V."""
    import os
    gold = torch.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_cases.pt"))
    logits = gold["logits"]
    B, V = logits.shape
    lg = torch.zeros(B, 2560, dtype=torch.bfloat16)
    lg[:, :V] = logits
    lg[:, V:] = 99.0                                          # padding must be ignored
    lgd = dev(lg)
    for case in gold["cases"]:
        frame = torch.full((B, 32), -1, dtype=torch.int32, device="cuda")
        nd = dev(case["noise"])
        _ck(abi, abi.lib.csm_op_sample(B, V, 2560, lgd.data_ptr(), case["temperature"], case["topk"],
                                       nd.data_ptr(), None, 7, 32, frame.data_ptr(), stream()))
        torch.cuda.synchronize()
        got = frame[:, 7].cpu()
        agree = (got == case["out"]).float().mean().item()
        # the only freedom is the fp32 order of the exp-sums (<= ~52 terms); it can move one
        # probability by one bf16 ulp.  Greedy is exact.
        need = 1.0 if case["topk"] == 1 else 0.95
        assert agree >= need, f"sampler T={case['temperature']} k={case['topk']}: agreement {agree:.3f}"
        if case["topk"] > 1:
            _assert_sampler_misses_are_one_ulp_ties(logits, case["temperature"], case["topk"], case["noise"], got, case["out"].cpu(),
                                                    f"golden T={case['temperature']} k={case['topk']}")
        assert (frame[:, :7] == -1).all() and (frame[:, 8:] == -1).all()


@pytest.mark.parametrize("V", [2051, 1000, 2056, 2100])
def test_sampler_random_parameters_vs_oracle(abi, V):
    """Random (temperature, top-k) pairs, logits on a coarse bf16 grid (many exact ties at the kth value): the HIP
    sampler must pick the oracle's index given the oracle's Exp(1) noise.  Integer-exact except where one
    probability sits within a bf16 ulp of the decision (fp32 order of <= k exp-sums), as in the golden cases.
    V = 2051 / 2056 / 2100: a second 2,048-logit row with 3 / 8 / 52 live logits; V = 1000: one row."""
    from oracle.csm_ref import sample_topk
    g = torch.Generator().manual_seed(V)
    B, ldl = 64, 2560
    total, agree = 0, 0
    for case in range(24):
        T = float(torch.empty(1).uniform_(0.2, 1.6, generator=g))
        k = int(torch.randint(1, 120, (1,), generator=g)) if case else V           # case 0: top-k = V (nothing removed)
        scale = (0.5, 1.0, 3.0)[case % 3]
        logits = (torch.randn(B, V, generator=g) * scale).to(torch.bfloat16)
        if case % 4 == 1:
            logits = (logits.float() * 4).round().div(4).to(torch.bfloat16)            # quarter-steps: ties everywhere
        noise = torch.empty(B, V).exponential_(1, generator=g).to(torch.bfloat16).clamp_min(1e-30)
        want = sample_topk(logits, k, T, q=noise)[:, 0]
        lg = torch.full((B, ldl), 99.0, dtype=torch.bfloat16); lg[:, :V] = logits
        lgd, nd = dev(lg), dev(noise)
        frame = torch.full((B, 32), -1, dtype=torch.int32, device="cuda")
        _ck(abi, abi.lib.csm_op_sample(B, V, ldl, lgd.data_ptr(), T, k, nd.data_ptr(), None, 3, 32, frame.data_ptr(), stream()))
        torch.cuda.synchronize()
        got = frame[:, 3].cpu()
        assert int(got.min()) >= 0 and int(got.max()) < V
        # every pick must at least be a kept (top-k) index
        kth = torch.topk((logits / T), k)[0][:, -1]
        assert bool(((logits / T)[torch.arange(B), got.long()] >= kth).all()), f"case {case}: picked a removed index"
        total += B; agree += int((got == want).sum())
        if k > 1:
            _assert_sampler_misses_are_one_ulp_ties(logits, T, k, noise, got, want.cpu(), f"case {case} T={T:.3f} k={k}")
    print(f"sampler V={V}: {agree}/{total} picks identical; every other pick is a <= 1-ulp tie of p/q")
    assert agree / total >= 0.95, f"agreement {agree / total:.3f}"


def test_sampler_philox_distribution(abi):
    """without supplied noise the on-device Philox Exp(1) draws must reproduce the softmax
    distribution of the kept logits (statistical parity)."""
    V, B, k, T = 2051, 1024, 5, 1.0
    g = torch.Generator().manual_seed(5)
    row = (torch.randn(V, generator=g) * 2).to(torch.bfloat16)
    lg = torch.zeros(B, 2560, dtype=torch.bfloat16)
    lg[:, :V] = row
    lgd = dev(lg)
    rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    counts = torch.zeros(V)
    for step in range(8):
        rng[1] = step
        frame = torch.zeros(B, 32, dtype=torch.int32, device="cuda")
        # row index b enters the Philox counter, so identical rows draw independent samples
        _ck(abi, abi.lib.csm_op_sample(B, V, 2560, lgd.data_ptr(), T, k, None, rng.data_ptr(), 0, 32, frame.data_ptr(), stream()))
        torch.cuda.synchronize()
        counts += torch.bincount(frame[:, 0].cpu().long(), minlength=V).float()
    top_v, top_i = torch.topk(row.float(), k)
    p = torch.softmax(top_v, -1)
    assert counts.sum() == counts[top_i].sum(), "sampled outside the top-k set"
    freq = counts[top_i] / counts.sum()
    n = counts.sum()
    sigma = torch.sqrt(p * (1 - p) / n)
    assert ((freq - p).abs() <= 5 * sigma + 0.01).all(), f"freq {freq.tolist()} vs p {p.tolist()}"
