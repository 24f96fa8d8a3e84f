// Wide-M path (prompt prefill, batched utterances): the same projections as gemv.cuh but for
// M >= 16 token rows, on the bf16 matrix cores.
//
//   out[M][N] = x[M][K] . W[N][K]^T        (x = already-normalised activations, bf16)
//
// One block = one 32(M) x 32(N) output tile; its 4 waves split K four ways and each runs
// v_mfma_f32_32x32x16_bf16 down its K range with both operands straight from L2/HBM (a block
// streams its 32 weight rows exactly once; x tiles are re-read from L2 by the N/32 blocks of a
// row stripe).  Inside every 64-wide K chunk the k order is permuted so that each lane reads 64
// contiguous bytes of its row (lane half h, step q takes k = 32h + 8q .. +8 for BOTH operands,
// which keeps the MFMA sum over k intact).  The four partial tiles are summed through LDS in a
// fixed order (deterministic) and the epilogues are those of the GEMV path: store / +residual /
// fused q,k,v + interleaved RoPE + KV append / SiLU(gate)*up.
#pragma once
#include "common.cuh"
#include "gemv.cuh"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// RMSNorm of M rows (torchtune rounding: fp32 normalise -> bf16 -> * bf16 scale); one wave per row
__global__ __launch_bounds__(256) void k_rmsnorm_rows(const bf16_t* x, long x_row_stride, long x_row_offset, int M, int K,
                                                      const bf16_t* scale, float eps, bf16_t* out, long out_stride) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const uint4* src = reinterpret_cast<const uint4*>(x + (long)row * x_row_stride + x_row_offset);
    float ss = 0.f;
    for (int c = lane; c < K / 8; c += 64) {
        const uint4 v = src[c];
        float f;
        f = lo2f(v.x); ss += f * f; f = hi2f(v.x); ss += f * f; f = lo2f(v.y); ss += f * f; f = hi2f(v.y); ss += f * f;
        f = lo2f(v.z); ss += f * f; f = hi2f(v.z); ss += f * f; f = lo2f(v.w); ss += f * f; f = hi2f(v.w); ss += f * f;
    }
    ss = wave_sum(ss);
    const float r = 1.0f / sqrtf(ss / (float)K + eps);
    for (int c = lane; c < K / 8; c += 64) {
        const uint4 v = src[c], g = reinterpret_cast<const uint4*>(scale)[c];
        uint4 o;
        o.x = pack_bf(round_bf(lo2f(v.x) * r) * lo2f(g.x), round_bf(hi2f(v.x) * r) * hi2f(g.x));
        o.y = pack_bf(round_bf(lo2f(v.y) * r) * lo2f(g.y), round_bf(hi2f(v.y) * r) * hi2f(g.y));
        o.z = pack_bf(round_bf(lo2f(v.z) * r) * lo2f(g.z), round_bf(hi2f(v.z) * r) * hi2f(g.z));
        o.w = pack_bf(round_bf(lo2f(v.w) * r) * lo2f(g.w), round_bf(hi2f(v.w) * r) * hi2f(g.w));
        reinterpret_cast<uint4*>(out + (long)row * out_stride)[c] = o;
    }
}

__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// GemvArgs is reused: x (row stride x_row_stride), M, w0/w1/w2, N, out/ldo, resid, QKV fields.
// K is a runtime argument (multiple of 256).
template <int EPI, int HD>
__global__ __launch_bounds__(256) void k_mm32(const GemvArgs a, const int K) {
    __shared__ float red[EPI == EPI_SWIGLU ? 2 : 1][4][16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int mrow = min(m0 + r, a.M - 1);
    int nrow = min(n0 + r, a.N - 1);
    const bf16_t* wa;                                     // weight row feeding accumulator 0
    const bf16_t* wb = nullptr;                           // SwiGLU: the matching up-projection row
    if (EPI == EPI_QKV_ROPE) {
        if (nrow < a.nq) wa = a.w0 + (long)nrow * K;
        else if (nrow < a.nq + a.nkv) wa = a.w1 + (long)(nrow - a.nq) * K;
        else wa = a.w2 + (long)(nrow - a.nq - a.nkv) * K;
    } else {
        wa = a.w0 + (long)nrow * K;
        if (EPI == EPI_SWIGLU) wb = a.w1 + (long)nrow * K;
    }
    const bf16_t* xa = a.x + (long)mrow * a.x_row_stride + a.x_row_offset;
    const int kspan = K / 4, kbeg = wave * kspan;
    f32x16_t acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    for (int kc = kbeg; kc < kbeg + kspan; kc += 64) {
        uint4 av[4], bv[4], cv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = *reinterpret_cast<const uint4*>(xa + kc + h * 32 + q * 8);
            bv[q] = *reinterpret_cast<const uint4*>(wa + kc + h * 32 + q * 8);
            if (EPI == EPI_SWIGLU) cv[q] = *reinterpret_cast<const uint4*>(wb + kc + h * 32 + q * 8);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[q]), as_bf16x8(bv[q]), acc0, 0, 0, 0);
            if (EPI == EPI_SWIGLU)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[q]), as_bf16x8(cv[q]), acc1, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        red[0][wave][i][lane] = acc0[i];
        if (EPI == EPI_SWIGLU) red[1][wave][i][lane] = acc1[i];
    }
    __syncthreads();
    // thread (wave g, lane) finishes regs 4g..4g+3 of `lane`: rows 8g + 4h + {0..3}, column n0 + r
    {
#pragma clang fp contract(off)
        const int n = n0 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int reg = wave * 4 + i;
            const int m = m0 + 8 * wave + 4 * h + i;
            float s0 = red[0][0][reg][lane] + red[0][1][reg][lane] + red[0][2][reg][lane] + red[0][3][reg][lane];
            float y = round_bf(s0);
            if (EPI == EPI_QKV_ROPE) {
                // partner column n^1 lives in the neighbouring lane; every lane must take part in the shuffle
                const float other = __shfl_xor(y, 1, WAVE);
                if (m < a.M && n < a.N) {
                    const int p = row_pos(a, m);
                    const int b = m / a.rows_per_seq;
                    if (n < a.nq + a.nkv) {
                        const int e = (n < a.nq ? n : n - a.nq) % HD;
                        const uint32_t cs = reinterpret_cast<const uint32_t*>(a.rope)[(long)p * (HD / 2) + e / 2];
                        const float c = lo2f(cs), s = hi2f(cs);
                        y = (n & 1) ? (y * c + other * s) : (y * c - other * s);
                    }
                    if (n < a.nq) a.out[(long)m * a.ldo + n] = f2bf(y);
                    else {
                        const bool isk = n < a.nq + a.nkv;
                        const int rk = n - a.nq - (isk ? 0 : a.nkv);
                        (isk ? a.kcache : a.vcache)[(((long)b * a.kv_heads + rk / HD) * a.smax + p) * HD + rk % HD] = f2bf(y);
                    }
                }
            } else if (m < a.M && n < a.N) {
                if (EPI == EPI_SWIGLU) {
                    const float u = round_bf(red[1][0][reg][lane] + red[1][1][reg][lane] + red[1][2][reg][lane] + red[1][3][reg][lane]);
                    const float sg = round_bf(y / (1.0f + __expf(-y)));
                    y = sg * u;
                } else if (EPI == EPI_RESID) {
                    y = y + bf2f(a.resid[(long)m * a.ldo + n]);
                }
                a.out[(long)m * a.ldo + n] = f2bf(y);
            }
        }
    }
}
