// Weight-streaming skinny GEMV family for the per-frame step (M = 1..few token rows).
//
// HBM-bound by construction: every weight byte is loaded exactly once per launch with
// 16 B/lane coalesced loads straight into VGPRs (no LDS round trip -- guide: "GEMV / M<=16
// decode weights: load straight to VGPRs, deep unroll, late vmcnt"), all of a wave's loads
// are issued before the activation tile is staged, and the weights stay in registers while
// the kernel walks the M tiles.  One wave owns R consecutive output rows; the K reduction is
// lane-strided (lane l, step i covers k = (64 i + l) * 8 .. +8) and finished with wave
// shuffles.  Activations are staged once per block into LDS as packed bf16 (optionally
// RMS-normalised on the way in) and consumed with v_dot2c_f32_bf16.
//
// Rounding points are the reference's (torchtune 0.4.0 eager bf16):
// fp32 accumulate -> bf16 at every Linear output; RMSNorm rounds before the scale multiply;
// RoPE / SiLU / residual each round once.
#pragma once
#include "common.cuh"

enum { PRO_PLAIN = 0, PRO_NORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV_ROPE = 2, EPI_SWIGLU = 3 };

struct GemvArgs {
    // activations: row m lives at x + m * x_row_stride + x_row_offset (elements), K wide
    const bf16_t* x;
    long x_row_stride, x_row_offset;
    int M;
    // prologue
    const bf16_t* norm_scale;   // PRO_NORM
    float eps;
    bf16_t* normed_out;         // PRO_NORM, optional: block 0 writes normalised row m at + m*normed_stride
    long normed_stride;
    // weights [N][K]
    const bf16_t* w0;           // STORE/RESID: W; QKV: Wq; SWIGLU: W1 (gate)
    const bf16_t* w1;           //                  QKV: Wk; SWIGLU: W3 (up)
    const bf16_t* w2;           //                  QKV: Wv
    int N;                      // output rows (QKV: nq + 2 nkv; SWIGLU: ffn)
    // epilogue
    bf16_t* out;                // row m at out + m*ldo   (QKV: q buffer, ldo = nq)
    long ldo;
    int nt;                     // non-temporal weight loads (streamed-once weights)
    const bf16_t* resid;        // EPI_RESID [M][N] (may alias out)
    // QKV
    int nq, nkv, smax, rows_per_seq, kv_heads;
    const int* pos;             // [M] absolute position of each row
    const bf16_t* rope;         // [max_seq][HD/2][2]
    bf16_t* kcache;             // [B][KV][smax][HD] (this layer)
    bf16_t* vcache;
};

template <int MT, int KITERS, bool NORM>
__device__ __forceinline__ void stage_x(bf16_t* xs, float* red, const GemvArgs& a, int m0) {
    constexpr int K = KITERS * 512;
    constexpr int CHUNKS = K / 8;               // 16-byte chunks per row
    const int tid = threadIdx.x;
    float ss[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        ss[m] = 0.f;
        const bool live = (m0 + m) < a.M;
        const uint4* src = reinterpret_cast<const uint4*>(a.x + (long)(m0 + m) * a.x_row_stride + a.x_row_offset);
        for (int c = tid; c < CHUNKS; c += 256) {
            uint4 v = live ? src[c] : make_uint4(0, 0, 0, 0);
            if (NORM) {
                float f;
                f = lo2f(v.x); ss[m] += f * f; f = hi2f(v.x); ss[m] += f * f;
                f = lo2f(v.y); ss[m] += f * f; f = hi2f(v.y); ss[m] += f * f;
                f = lo2f(v.z); ss[m] += f * f; f = hi2f(v.z); ss[m] += f * f;
                f = lo2f(v.w); ss[m] += f * f; f = hi2f(v.w); ss[m] += f * f;
            }
            reinterpret_cast<uint4*>(xs + m * K)[c] = v;
        }
    }
    if (NORM) {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float s = wave_sum(ss[m]);
            if (lane == 0) red[m * 4 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float tot = red[m * 4 + 0] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3];
            const float r = 1.0f / sqrtf(tot / (float)K + a.eps);
            const uint4* sc = reinterpret_cast<const uint4*>(a.norm_scale);
            for (int c = tid; c < CHUNKS; c += 256) {      // each thread re-reads only its own chunks
                uint4 v = reinterpret_cast<uint4*>(xs + m * K)[c];
                uint4 g = sc[c];
                uint4 o;
                // x32 * rsqrt -> bf16 (type_as) -> * scale (bf16 * bf16 -> bf16)
                o.x = pack_bf(round_bf(lo2f(v.x) * r) * lo2f(g.x), round_bf(hi2f(v.x) * r) * hi2f(g.x));
                o.y = pack_bf(round_bf(lo2f(v.y) * r) * lo2f(g.y), round_bf(hi2f(v.y) * r) * hi2f(g.y));
                o.z = pack_bf(round_bf(lo2f(v.z) * r) * lo2f(g.z), round_bf(hi2f(v.z) * r) * hi2f(g.z));
                o.w = pack_bf(round_bf(lo2f(v.w) * r) * lo2f(g.w), round_bf(hi2f(v.w) * r) * hi2f(g.w));
                reinterpret_cast<uint4*>(xs + m * K)[c] = o;
                if (a.normed_out != nullptr && blockIdx.x == 0 && (m0 + m) < a.M)
                    reinterpret_cast<uint4*>(a.normed_out + (long)(m0 + m) * a.normed_stride)[c] = o;
            }
        }
    }
    __syncthreads();
}

// R = weight rows per wave.  EPI_QKV_ROPE: R == 2 (one interleaved RoPE pair).
// EPI_SWIGLU: R == 2*P, rows [0,P) are gate rows i..i+P-1 and [P,2P) the matching up rows.
template <int MT, int KITERS, int R, int PRO, int EPI, int HD>
__global__ __launch_bounds__(256) void k_gemv(const GemvArgs a) {
    constexpr int K = KITERS * 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + (size_t)MT * K * 2);

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;              // one unit = R weight rows

    // ---- resolve this wave's R weight rows (nullptr = past the end) ------------------------
    const bf16_t* wrow[R];
    int orow[R];                                          // output row index
    if (EPI == EPI_QKV_ROPE) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = unit * R + r;
            orow[r] = row;
            if (row < a.nq) wrow[r] = a.w0 + (long)row * K;
            else if (row < a.nq + a.nkv) wrow[r] = a.w1 + (long)(row - a.nq) * K;
            else if (row < a.N) wrow[r] = a.w2 + (long)(row - a.nq - a.nkv) * K;
            else wrow[r] = nullptr;
        }
    } else if (EPI == EPI_SWIGLU) {
        constexpr int P = R / 2;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            int row = unit * P + r;
            orow[r] = orow[r + P] = row;
            wrow[r] = row < a.N ? a.w0 + (long)row * K : nullptr;
            wrow[r + P] = row < a.N ? a.w1 + (long)row * K : nullptr;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = unit * R + r;
            orow[r] = row;
            wrow[r] = row < a.N ? a.w0 + (long)row * K : nullptr;
        }
    }

    // ---- issue every weight load of this wave up front; they stay in VGPRs ----------------
    uint4 wv[R][KITERS];
    if (a.nt) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < KITERS; ++i)
                wv[r][i] = wrow[r] ? ldg16<true>(reinterpret_cast<const uint4*>(wrow[r]) + i * 64 + lane)
                                   : make_uint4(0, 0, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < KITERS; ++i)
                wv[r][i] = wrow[r] ? ldg16<false>(reinterpret_cast<const uint4*>(wrow[r]) + i * 64 + lane)
                                   : make_uint4(0, 0, 0, 0);
    }

    for (int m0 = 0; m0 < a.M; m0 += MT) {
        if (m0 > 0) __syncthreads();
        stage_x<MT, KITERS, PRO == PRO_NORM>(xs, red, a, m0);

        float acc[MT][R];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
#pragma unroll
        for (int i = 0; i < KITERS; ++i) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const uint4 xv = reinterpret_cast<const uint4*>(xs + m * K)[i * 64 + lane];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[m][r] = dot8(wv[r][i], xv, acc[m][r]);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[m][r] = wave_sum(acc[m][r]);

        // ---- epilogue: lane m finishes token row m0+m -------------------------------------
        {
#pragma clang fp contract(off)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (lane != m || m0 + m >= a.M) continue;
            const long mrow = m0 + m;
            if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (orow[r] >= a.N) continue;
                    float y = round_bf(acc[m][r]);
                    if (EPI == EPI_RESID) y = y + bf2f(a.resid[mrow * a.ldo + orow[r]]);
                    a.out[mrow * a.ldo + orow[r]] = f2bf(y);
                }
            } else if constexpr (EPI == EPI_SWIGLU) {
                constexpr int P = R / 2;
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    if (orow[r] >= a.N) continue;
                    const float g = round_bf(acc[m][r]);
                    const float u = round_bf(acc[m][r + P]);
                    const float s = round_bf(g / (1.0f + __expf(-g)));      // F.silu in bf16
                    a.out[mrow * a.ldo + orow[r]] = f2bf(s * u);
                }
            } else {   // EPI_QKV_ROPE: rows (2i, 2i+1) of one head
                const int row = orow[0];
                if (row >= a.N) continue;
                float v0 = round_bf(acc[m][0]), v1 = round_bf(acc[m][1]);
                int p = a.pos[mrow];
                p = p < 0 ? 0 : (p >= a.smax ? a.smax - 1 : p);     // memory safety; the host guards length
                const int b = (int)(mrow / a.rows_per_seq);
                if (row < a.nq + a.nkv) {                 // q or k: interleaved Llama3-scaled RoPE
                    const int e = (row < a.nq ? row : row - a.nq) % HD;
                    const uint32_t cs = reinterpret_cast<const uint32_t*>(a.rope)[(long)p * (HD / 2) + e / 2];
                    const float c = lo2f(cs), s = hi2f(cs);
                    const float o0 = v0 * c - v1 * s;
                    const float o1 = v1 * c + v0 * s;
                    v0 = o0; v1 = o1;
                }
                const uint32_t packed = pack_bf(v0, v1);
                if (row < a.nq) {
                    *reinterpret_cast<uint32_t*>(a.out + mrow * a.ldo + row) = packed;
                } else {
                    const bool isk = row < a.nq + a.nkv;
                    const int rk = row - a.nq - (isk ? 0 : a.nkv);
                    const int kvh = rk / HD, e = rk % HD;
                    bf16_t* dst = (isk ? a.kcache : a.vcache) + (((long)b * a.kv_heads + kvh) * a.smax + p) * HD + e;
                    *reinterpret_cast<uint32_t*>(dst) = packed;
                }
            }
        }
        }
    }
}
