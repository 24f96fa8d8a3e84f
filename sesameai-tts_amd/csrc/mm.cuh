// Wide-M path (batched utterances from 3 rows, prompt chunks below 256 rows -- longer ones take gemm128.cuh): the same
// projections as gemv.cuh, on the bf16 matrix cores.
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
                                                      const bf16_t* scale, float eps, bf16_t* out, long out_stride, int out_packed) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    if (out_packed) rmsnorm_row_wave(x + (long)row * x_row_stride + x_row_offset, K, scale, eps, out, lane, row);
    else rmsnorm_row_wave(x + (long)row * x_row_stride + x_row_offset, K, scale, eps, out + (long)row * out_stride, lane);
}

// Finishes a split-K residual projection of the wide path and (optionally) applies the next RMSNorm:
//   h[r] = bf16(h[r] + bf16(sum_g slab[g][r]))      (fixed g order: deterministic)
//   xn[i] = RMSNorm(h[r]) * scale                   (skipped when scale == nullptr)
// for output row i, r = i * row_step + row_first (row selection = "last row of every sequence" for the heads).
// One wave per row; the row stays in registers between the two passes (N <= 64 * 8 * NCH).
template <int NCH>
__global__ __launch_bounds__(256) void k_resid_norm(bf16_t* h, const float* slab, int KG, int M, int N, long row_step, long row_first,
                                                    int M_out, const bf16_t* scale, float eps, bf16_t* xn, long xn_stride, int xn_packed) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= M_out) return;
    const long r = (long)i * row_step + row_first;
    float v[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col >= N) continue;
        const uint4 hv = *reinterpret_cast<const uint4*>(h + r * N + col);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        float4 p0[8], p1[8];                                  // all slab loads in flight together (KG <= 8)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int gg = g < KG ? g : 0;
            p0[g] = *reinterpret_cast<const float4*>(slab + ((long)gg * M + r) * N + col);
            p1[g] = *reinterpret_cast<const float4*>(slab + ((long)gg * M + r) * N + col + 4);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {                         // fixed order: deterministic
            if (g < KG) {
                acc[0] += p0[g].x; acc[1] += p0[g].y; acc[2] += p0[g].z; acc[3] += p0[g].w;
                acc[4] += p1[g].x; acc[5] += p1[g].y; acc[6] += p1[g].z; acc[7] += p1[g].w;
            }
        }
        const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float hj = (j & 1) ? hi2f(hw[j >> 1]) : lo2f(hw[j >> 1]);
            v[c][j] = round_bf(hj + round_bf(acc[j]));
            ss += v[c][j] * v[c][j];
        }
        uint4 o;
        o.x = pack_bf(v[c][0], v[c][1]); o.y = pack_bf(v[c][2], v[c][3]); o.z = pack_bf(v[c][4], v[c][5]); o.w = pack_bf(v[c][6], v[c][7]);
        *reinterpret_cast<uint4*>(h + r * N + col) = o;
    }
    if (scale == nullptr) return;
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)N + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = (c * 64 + lane) * 8;
        if (col >= N) continue;
        const uint4 g = *reinterpret_cast<const uint4*>(scale + col);
        uint4 o;
        o.x = pack_bf(round_bf(v[c][0] * rs) * lo2f(g.x), round_bf(v[c][1] * rs) * hi2f(g.x));
        o.y = pack_bf(round_bf(v[c][2] * rs) * lo2f(g.y), round_bf(v[c][3] * rs) * hi2f(g.y));
        o.z = pack_bf(round_bf(v[c][4] * rs) * lo2f(g.z), round_bf(v[c][5] * rs) * hi2f(g.z));
        o.w = pack_bf(round_bf(v[c][6] * rs) * lo2f(g.w), round_bf(v[c][7] * rs) * hi2f(g.w));
        *reinterpret_cast<uint4*>(xn_packed ? xn + xp_off(i, col, N) : xn + (long)i * xn_stride + col) = o;
    }
}

// The same finisher for DECODE steps (a handful of rows): one 256-thread block per row instead of one wave, so a
// lane has KG 16-byte slab loads in flight instead of 4 KG and the row's bytes come through four waves' worth of
// memory pipes -- the kernel is one dependent round trip on remote-L2 slabs, not bandwidth.  Thread t owns columns
// [CPT t, CPT t + CPT).  Sum order over slabs is fixed (deterministic); the sum of squares is reduced in a different
// order than k_resid_norm, which is why prompt rows (bit-identical across prefill sizes) never use this kernel.
template <int CPT>
__global__ __launch_bounds__(256) void k_resid_norm_row(bf16_t* h, const float* slab, int KG, int M, int N, long row_step, long row_first,
                                                        const bf16_t* scale, float eps, bf16_t* xn, long xn_stride, int xn_packed) {
    __shared__ float s_part[4];
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r = (long)i * row_step + row_first;
    const int col = tid * CPT;
    const bool in = col < N;
    float acc[CPT], v[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) acc[j] = 0.f;
    float4 p[8][CPT / 4];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int gg = g < KG ? g : 0;
#pragma unroll
        for (int q = 0; q < CPT / 4; ++q)
            p[g][q] = in ? *reinterpret_cast<const float4*>(slab + ((long)gg * M + r) * N + col + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    uint32_t hw[CPT / 2];
#pragma unroll
    for (int q = 0; q < CPT / 2; ++q) hw[q] = in ? reinterpret_cast<const uint32_t*>(h + r * N + col)[q] : 0u;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g < KG) {
#pragma unroll
            for (int q = 0; q < CPT / 4; ++q) {
                acc[4 * q] += p[g][q].x; acc[4 * q + 1] += p[g][q].y; acc[4 * q + 2] += p[g][q].z; acc[4 * q + 3] += p[g][q].w;
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float hj = (j & 1) ? hi2f(hw[j >> 1]) : lo2f(hw[j >> 1]);
        v[j] = round_bf(hj + round_bf(acc[j]));
        ss += v[j] * v[j];
    }
    if (in) {
#pragma unroll
        for (int q = 0; q < CPT / 2; ++q) reinterpret_cast<uint32_t*>(h + r * N + col)[q] = pack_bf(v[2 * q], v[2 * q + 1]);
    }
    if (scale == nullptr) return;
    uint32_t gw[CPT / 2];
#pragma unroll
    for (int q = 0; q < CPT / 2; ++q) gw[q] = in ? reinterpret_cast<const uint32_t*>(scale + col)[q] : 0u;
    ss = wave_sum(ss);
    if (lane == 0) s_part[wave] = ss;
    __syncthreads();
    ss = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
    const float rs = 1.0f / sqrtf(ss / (float)N + eps);
    if (in) {
#pragma unroll
        for (int q = 0; q < CPT / 2; ++q)
            reinterpret_cast<uint32_t*>(xn + (xn_packed ? xp_off(i, col, N) : (long)i * xn_stride + col))[q] =     // CPT | 8: one piece
                pack_bf(round_bf(v[2 * q] * rs) * lo2f(gw[q]), round_bf(v[2 * q + 1] * rs) * hi2f(gw[q]));
    }
}

__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// Matrix-core operand order for the weight stream ("memory laid out for the hardware"): W [N][K]
// row-major is re-tiled ONCE at load time so that every wave-level load of the wide-M kernels is
// one fully contiguous 1 KB read:  Wt[ntile][chunk][q][lane][8]  with
//   Wt[...] = W[ntile*32 + (lane & 31)][chunk*64 + (lane >> 5)*32 + q*8 + j]
// (lane, q) being exactly the (row, k-step) a lane feeds to v_mfma_f32_32x32x16_bf16 in k_mm32.
// Rows beyond N (N padded to a multiple of 32) are zero.  One thread per 16-byte piece.
__global__ void k_pack_w(const bf16_t* w, int N, int K, bf16_t* wt) {
    const long piece = (long)blockIdx.x * blockDim.x + threadIdx.x;       // ((ntile*(K/64) + chunk)*4 + q)*64 + lane
    const long total = (long)((N + 31) / 32) * (K / 64) * 4 * 64;
    if (piece >= total) return;
    const int lane = (int)(piece & 63), q = (int)((piece >> 6) & 3);
    const long tc = piece >> 8;
    const int chunk = (int)(tc % (K / 64));
    const int ntile = (int)(tc / (K / 64));
    const int n = ntile * 32 + (lane & 31);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < N) v = *reinterpret_cast<const uint4*>(w + (long)n * K + chunk * 64 + (lane >> 5) * 32 + q * 8);
    reinterpret_cast<uint4*>(wt)[piece] = v;
}

// Weight fragments of the wide-M kernels: WT = 0 bf16 (16 bytes = 8 k values per lane), WT = 1 OCP-e4m3 bytes (8 bytes
// per lane; converted to bf16 in registers right before the MFMA, the power-of-two row scale applied to the fp32 sum,
// which is exact -- the result equals the bf16 kernel's on the dequantised weights bit for bit).
template <int WT> struct WFrag { typedef uint4 type; };
template <> struct WFrag<1> { typedef uint2 type; };
template <bool NT> __device__ __forceinline__ uint2 ldg8(const uint2* p) {
    if (NT) {
        typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
        const u32x2v v = __builtin_nontemporal_load(reinterpret_cast<const u32x2v*>(p));
        return make_uint2(v.x, v.y);
    }
    return *p;
}
__device__ __forceinline__ uint4 wld(const uint4* p) { return ldg16<true>(p); }
__device__ __forceinline__ uint2 wld(const uint2* p) { return ldg8<true>(p); }
__device__ __forceinline__ bf16x8_t wfrag_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ bf16x8_t wfrag_bf16x8(const uint2& v) {
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    const bf16x2v a = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v.x, 1.0f, false), b = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v.x, 1.0f, true);
    const bf16x2v c = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v.y, 1.0f, false), d = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v.y, 1.0f, true);
    bf16x8_t o;
    o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1]; o[4] = c[0]; o[5] = c[1]; o[6] = d[0]; o[7] = d[1];
    return o;
}
// fp8 counterpart of k_pack_w: 8-byte pieces, same (ntile, chunk, q, lane) order
__global__ void k_pack_w8(const uint8_t* w, int N, int K, uint2* wt) {
    const long piece = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)((N + 31) / 32) * (K / 64) * 4 * 64;
    if (piece >= total) return;
    const int lane = (int)(piece & 63), q = (int)((piece >> 6) & 3);
    const long tc = piece >> 8;
    const int chunk = (int)(tc % (K / 64));
    const int ntile = (int)(tc / (K / 64));
    const int n = ntile * 32 + (lane & 31);
    uint2 v = make_uint2(0, 0);
    if (n < N) v = *reinterpret_cast<const uint2*>(w + (long)n * K + chunk * 64 + (lane >> 5) * 32 + q * 8);
    wt[piece] = v;
}

// Epilogue of one output element (row m, column n) of the wide-M kernels; s0 = fp32 dot product (s1 = the
// up-projection's for SwiGLU).  Same rounding points as the GEMV path.  QKV: every lane of the wave must call
// this together (the RoPE partner column n^1 sits in the neighbouring lane).
template <int EPI, int HD>
__device__ __forceinline__ void mm_finish(const GemvArgs& a, const int m, const int n, const float s0, const float s1) {
#pragma clang fp contract(off)
    float y = round_bf(s0);
    if (EPI == EPI_QKV_ROPE) {
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
            const float u = round_bf(s1);
            const float sg = round_bf(y / (1.0f + __expf(-y)));
            y = sg * u;
        } else if (EPI == EPI_RESID) {
            y = y + bf2f(a.resid[(long)m * a.ldo + n]);
        }
        a.out[(EPI == EPI_SWIGLU && a.out_packed) ? xp_off(m, n, a.ldo) : (long)m * a.ldo + n] = f2bf(y);
    }
}

// GemvArgs is reused: x (row stride x_row_stride), M, w0/w1/w2, N, out/ldo, resid, QKV fields.
// K is a runtime argument (multiple of 64 * NW).  NW = waves per block = K split (4, or 16 for the
// K = 8192 down projections so that a 32-row stripe of a 16 MB matrix is pulled by 16 waves).
//
// Tile mapping.  mtiles == 0: grid (n tiles, 1 row tile, K groups) -- decode steps of up to 32 streams.
// mtiles >= 2 (batches of 33..511 rows): 1-D grid, XCD-aware.  Workgroups are placed round-robin over the 8 XCDs, each
// with its own 4 MB L2, so block L runs on XCD L % 8.  XCD c owns the n tiles {c, c+8, ...} and runs all `mtiles` row
// tiles of one (n tile, K group) back to back: the weight tile comes from HBM once per XCD and from its L2 for the
// other row tiles (B=128: 11.2 -> 10.4 ms/step, B=256: 22.4 -> 16.2; the opposite assignment, row tiles per XCD,
// changed nothing).
template <int EPI, int HD, int NW, int WT = 0, bool XP = false>
__global__ __launch_bounds__(NW * 64) void k_mm32(const GemvArgs a, const int K, const int mtiles, const int kgroups) {
    __shared__ float red[EPI == EPI_SWIGLU ? 2 : 1][NW][16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    int n0, m0, kz;
    if (mtiles == 0) { n0 = blockIdx.x * 32; m0 = blockIdx.y * 32; kz = blockIdx.z; }
    else {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int mt = j % mtiles, rest = j / mtiles;
        kz = rest % kgroups;
        const int nt = (rest / kgroups) * 8 + xcd;
        if (nt * 32 >= a.N) return;
        n0 = nt * 32; m0 = mt * 32;
    }
    const int mrow = min(m0 + r, a.M - 1);
    // w0/w1/w2 are PACKED (k_pack_w): tile pointer = base + ntile * (K/64) * 4 KB; a wave-level load of
    // step q in chunk c is the contiguous 1 KB at  tile + ((c*4 + q)*64 + lane) * 16 bytes
    typedef typename WFrag<WT>::type wfrag_t;             // 16-byte bf16 or 8-byte fp8 piece
    const wfrag_t* wa;                                    // weight tile feeding accumulator 0
    const wfrag_t* wb = nullptr;                          // SwiGLU: the matching up-projection tile
    const long tile_u4 = (long)(K / 64) * 256;            // pieces per n-tile
    float wscale0 = 1.0f, wscale1 = 1.0f;                 // WT == 1: power-of-two scale of this lane's weight row
    const int nl = min(n0 + r, a.N - 1);
    if (EPI == EPI_QKV_ROPE) {                            // nq, nkv are multiples of 32: a tile never straddles q/k/v
        if (n0 < a.nq) { wa = reinterpret_cast<const wfrag_t*>(a.w0) + (long)(n0 / 32) * tile_u4; if (WT) wscale0 = a.s0[nl]; }
        else if (n0 < a.nq + a.nkv) { wa = reinterpret_cast<const wfrag_t*>(a.w1) + (long)((n0 - a.nq) / 32) * tile_u4; if (WT) wscale0 = a.s1[nl - a.nq]; }
        else { wa = reinterpret_cast<const wfrag_t*>(a.w2) + (long)((n0 - a.nq - a.nkv) / 32) * tile_u4; if (WT) wscale0 = a.s2[nl - a.nq - a.nkv]; }
    } else {
        wa = reinterpret_cast<const wfrag_t*>(a.w0) + (long)(n0 / 32) * tile_u4;
        if (WT) wscale0 = a.s0[nl];
        if (EPI == EPI_SWIGLU) { wb = reinterpret_cast<const wfrag_t*>(a.w1) + (long)(n0 / 32) * tile_u4; if (WT) wscale1 = a.s1[nl]; }
    }
    const bf16_t* xa = a.x + (long)mrow * a.x_row_stride + a.x_row_offset;
    // EPI_SLAB: gridDim.z blocks split K further; each writes its fp32 partial tile to slab[blockIdx.z]
    const int kblk = K / kgroups;
    const int kspan = kblk / NW, kbeg = kz * kblk + wave * kspan, kend = kbeg + kspan;
    f32x16_t acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    if (kspan <= 256) {
        // short K span (decode-step projections at K = 1024): every load of the wave is issued before the
        // first MFMA -- one memory round trip instead of one per chunk
        uint4 av[4][4];
        wfrag_t bv[4][4], cv[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int kc = kbeg + c * 64;
            if (c * 64 < kspan) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    av[c][q] = XP ? reinterpret_cast<const uint4*>(a.x + (long)(m0 / 32) * 32 * a.x_row_stride)[((kc >> 6) * 4 + q) * 64 + lane]
                                  : *reinterpret_cast<const uint4*>(xa + kc + h * 32 + q * 8);
                    bv[c][q] = wld(wa + ((kc >> 6) * 4 + q) * 64 + lane);
                    if (EPI == EPI_SWIGLU) cv[c][q] = wld(wb + ((kc >> 6) * 4 + q) * 64 + lane);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c * 64 < kspan) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[c][q]), wfrag_bf16x8(bv[c][q]), acc0, 0, 0, 0);
                    if (EPI == EPI_SWIGLU)
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[c][q]), wfrag_bf16x8(cv[c][q]), acc1, 0, 0, 0);
                }
            }
        }
    } else {
    // register double buffer: the loads of chunk c+1 are in flight while chunk c feeds the MFMAs
    uint4 av[2][4];
    wfrag_t bv[2][4], cv[2][4];
    auto load = [&](int buf, int kc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[buf][q] = XP ? reinterpret_cast<const uint4*>(a.x + (long)(m0 / 32) * 32 * a.x_row_stride)[((kc >> 6) * 4 + q) * 64 + lane]
                            : *reinterpret_cast<const uint4*>(xa + kc + h * 32 + q * 8);
            bv[buf][q] = wld(wa + ((kc >> 6) * 4 + q) * 64 + lane);
            if (EPI == EPI_SWIGLU) cv[buf][q] = wld(wb + ((kc >> 6) * 4 + q) * 64 + lane);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[buf][q]), wfrag_bf16x8(bv[buf][q]), acc0, 0, 0, 0);
            if (EPI == EPI_SWIGLU)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[buf][q]), wfrag_bf16x8(cv[buf][q]), acc1, 0, 0, 0);
        }
    };
    // No conditional load inside the steady-state loop (round 2): with `if (more) load(...)` in the body hipcc's wait-count
    // pass merges both paths and put s_waitcnt vmcnt(0) in front of the MFMAs -- the chunk in flight was waited for as
    // well, i.e. the double buffer prefetched nothing.  Same MFMA order as before: same bits.
    load(0, kbeg);
    int kc = kbeg;
    for (; kc + 128 < kend; kc += 128) {
        load(1, kc + 64);
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        load(0, kc + 128);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kc + 64 < kend) { load(1, kc + 64); mma(0); mma(1); }
    else mma(0);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        red[0][wave][i][lane] = acc0[i];
        if constexpr (EPI == EPI_SWIGLU) red[1][wave][i][lane] = acc1[i];
    }
    __syncthreads();
    if (wave >= 4) return;                                // waves 0..3 finish the tile
    // thread (wave g, lane) finishes regs 4g..4g+3 of `lane`: rows 8g + 4h + {0..3}, column n0 + r
    {
#pragma clang fp contract(off)
        const int n = n0 + r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int reg = wave * 4 + i;
            const int m = m0 + 8 * wave + 4 * h + i;
            float s0 = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) s0 += red[0][w][reg][lane];       // fixed order: deterministic
            if (WT) s0 *= wscale0;
            if (EPI == EPI_SLAB) {
                if (m < a.M && n < a.N) a.slab[((long)kz * a.M + m) * a.N + n] = s0;
                continue;
            }
            float s1 = 0.f;
            if constexpr (EPI == EPI_SWIGLU) {
#pragma unroll
                for (int w = 0; w < NW; ++w) s1 += red[1][w][reg][lane];
                if (WT) s1 *= wscale1;
            }
            mm_finish<EPI, HD>(a, m, n, s0, s1);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Prompt rows, 64 <= M < 256 (a voice prompt of one segment: ~190 rows).  k_mm32 there moves 1.25 GB per backbone
// layer through the L2s for 122 MB of weights (every 32 x 32 output tile re-reads its x rows and its weight rows:
// x N/32 times, W M/32 times) and is bound by exactly that.  The two kernels below are k_mm32's arithmetic with more
// output tiles per wave, so a fragment fetched once feeds several MFMAs:
//   k_mmt  TM x TN tiles of 32 x 32 per wave, the four waves of a block = the four K quarters (as in k_mm32), fold
//          through LDS tile by tile, epilogue in the kernel (q|k|v + RoPE + KV append, SwiGLU);
//   k_mmq  residual projections: a block = (64 columns, ALL rows, ONE K quarter), its waves split the block's tiles and
//          each chains its quarter alone; the fp32 quarter goes to slab[quarter] and k_resid_norm adds the four in order.
// Both produce k_mm32's bits: per output tile the same MFMA chain per K quarter (k ascending, the same operand
// pieces) and the same fold ((0 + q0) + q1) + q2) + q3 -- rows keep their value whatever kernel or row count
// (tests: test_prompt_kernels_give_the_same_bits_at_every_row_count, test_csm1b_prefix_reuse_bit_identical).
// Weights are the k_pack_w copies; x is row-major or (XP) in operand order like the weights: a wave's x fragment is then
// one contiguous 1 KB read instead of a 64-cache-line gather, which is what bounds these kernels otherwise (measured:
// the row-major forms were SLOWER than k_mm32 at 190 rows, 2.85 vs 2.46 ms per prefill -- fewer, fatter waves, same gathers).
// ---------------------------------------------------------------------------------------------------------------
template <int EPI, int HD>
__device__ __forceinline__ const uint4* mmt_weight_tile(const GemvArgs& a, const int n, const long tile_u4, const int which) {
    if (EPI == EPI_QKV_ROPE) {
        if (n < a.nq) return reinterpret_cast<const uint4*>(a.w0) + (long)(n / 32) * tile_u4;
        if (n < a.nq + a.nkv) return reinterpret_cast<const uint4*>(a.w1) + (long)((n - a.nq) / 32) * tile_u4;
        return reinterpret_cast<const uint4*>(a.w2) + (long)((n - a.nq - a.nkv) / 32) * tile_u4;
    }
    return reinterpret_cast<const uint4*>(which ? a.w1 : a.w0) + (long)(n / 32) * tile_u4;
}

template <int EPI, int HD, int TM, int TN, bool XP, int NBUF>
__global__ __launch_bounds__(256) void k_mmt(const GemvArgs a, const int K, const int mgroups) {
    constexpr bool SW = EPI == EPI_SWIGLU;
    __shared__ float red[SW ? 2 : 1][4][16][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int mg = j % mgroups, ng = (j / mgroups) * 8 + xcd;         // XCD c owns the column groups c, c + 8, ...
    const int n0 = ng * 32 * TN, m0 = mg * 32 * TM;
    if (n0 >= a.N) return;
    const long tile_u4 = (long)(K / 64) * 256;
    const uint4* wa[TN];
    const uint4* wb[TN];
    const bf16_t* xa[TM];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        wa[tn] = mmt_weight_tile<EPI, HD>(a, n0 + 32 * tn, tile_u4, 0);
        wb[tn] = SW ? mmt_weight_tile<EPI, HD>(a, n0 + 32 * tn, tile_u4, 1) : nullptr;
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)       // XP: base of the row tile's operand-order pieces (tiles past the last row re-read the last tile)
        xa[tm] = XP ? a.x + (long)min(m0 / 32 + tm, (a.M - 1) / 32) * 32 * a.x_row_stride
                    : a.x + (long)min(m0 + 32 * tm + r, a.M - 1) * a.x_row_stride + a.x_row_offset;
    const int kspan = K / 4, kbeg = wave * kspan;
    f32x16_t acc0[TM][TN], acc1[SW ? TM : 1][SW ? TN : 1];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc0[tm][tn][i] = 0.f; if (SW) acc1[tm][tn][i] = 0.f; }
    // A ring of NBUF register buffers, each one HALF (MFMA steps 0-1 or 2-3) of a 64-deep chunk: NBUF - 1 halves of loads
    // are in flight while one feeds the MFMAs.  With one block per CU (few, fat waves) that depth is all the latency hiding
    // there is: two buffers left the q|k|v projection at 22 us against k_mm32's 15.
    uint4 av[NBUF][TM][2], bv[NBUF][TN][2], cv[NBUF][SW ? TN : 1][2];
    const int nh = kspan / 32;
    auto load = [&](int buf, int t) {                     // t = half index along this wave's K quarter
        const int kc = kbeg + (t >> 1) * 64;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = (t & 1) * 2 + qq;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                av[buf][tm][qq] = XP ? reinterpret_cast<const uint4*>(xa[tm])[((kc >> 6) * 4 + q) * 64 + lane]
                                     : *reinterpret_cast<const uint4*>(xa[tm] + kc + h * 32 + q * 8);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                bv[buf][tn][qq] = wld(wa[tn] + ((kc >> 6) * 4 + q) * 64 + lane);
                if (SW) cv[buf][tn][qq] = wld(wb[tn] + ((kc >> 6) * 4 + q) * 64 + lane);
            }
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc0[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[buf][tm][qq]), as_bf16x8(bv[buf][tn][qq]), acc0[tm][tn], 0, 0, 0);
                    if (SW) acc1[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[buf][tm][qq]), as_bf16x8(cv[buf][tn][qq]), acc1[tm][tn], 0, 0, 0);
                }
    };
    // (nh is a multiple of NBUF -- checked at launch.  No conditional load inside the steady-state loop: with one, hipcc's
    //  wait-count pass takes the minimum over both paths and the loop waits on vmcnt(0), i.e. prefetches nothing.)
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load(i, i);
    int t0 = 0;
    for (; t0 + 2 * NBUF <= nh; t0 += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            load((u + NBUF - 1) % NBUF, t0 + u + NBUF - 1);
            mma(u);
            __builtin_amdgcn_sched_barrier(0);         // keep load / MFMA groups interleaved (the scheduler would hoist all loads of the body)
        }
    }
    load(NBUF - 1, t0 + NBUF - 1);
#pragma unroll
    for (int u = 0; u < NBUF; ++u) mma(u);
    // fold the four K quarters tile by tile (the same order as k_mm32) and finish
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            if (tm + tn > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                red[0][wave][i][lane] = acc0[tm][tn][i];
                if constexpr (SW) red[1][wave][i][lane] = acc1[tm][tn][i];
            }
            __syncthreads();
            {
#pragma clang fp contract(off)
                const int n = n0 + 32 * tn + r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int reg = wave * 4 + i;
                    const int m = m0 + 32 * tm + 8 * wave + 4 * h + i;
                    float s0 = 0.f;
#pragma unroll
                    for (int w = 0; w < 4; ++w) s0 += red[0][w][reg][lane];
                    float s1 = 0.f;
                    if constexpr (SW) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) s1 += red[1][w][reg][lane];
                    }
                    mm_finish<EPI, HD>(a, m, n, s0, s1);
                }
            }
        }
}

// grid: 8 * ceil(column groups / 8) * 4 blocks; block = (column group of 64, K quarter); wave w: column tile w & 1, row
// tiles (w >> 1) * TM .. + TM of the (up to) 2 TM row tiles, i.e. M <= 64 TM.  slab[quarter][M][N] fp32.
template <int TM, bool XP, int NBUF>
__global__ __launch_bounds__(256) void k_mmq(const GemvArgs a, const int K) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int quarter = j & 3, ng = (j >> 2) * 8 + xcd;
    const int n0 = ng * 64 + 32 * (wave & 1), m0 = (wave >> 1) * 32 * TM;
    if (ng * 64 >= a.N || m0 >= a.M) return;
    const long tile_u4 = (long)(K / 64) * 256;
    const uint4* wa = reinterpret_cast<const uint4*>(a.w0) + (long)(n0 / 32) * tile_u4;
    const bf16_t* xa[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)       // XP: base of the row tile's operand-order pieces (tiles past the last row re-read the last tile)
        xa[tm] = XP ? a.x + (long)min(m0 / 32 + tm, (a.M - 1) / 32) * 32 * a.x_row_stride
                    : a.x + (long)min(m0 + 32 * tm + r, a.M - 1) * a.x_row_stride + a.x_row_offset;
    const int kspan = K / 4, kbeg = quarter * kspan;
    f32x16_t acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tm][i] = 0.f;
    uint4 av[NBUF][TM][2], bv[NBUF][2];
    const int nh = kspan / 32;
    auto load = [&](int buf, int t) {
        const int kc = kbeg + (t >> 1) * 64;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int q = (t & 1) * 2 + qq;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                av[buf][tm][qq] = XP ? reinterpret_cast<const uint4*>(xa[tm])[((kc >> 6) * 4 + q) * 64 + lane]
                                     : *reinterpret_cast<const uint4*>(xa[tm] + kc + h * 32 + q * 8);
            bv[buf][qq] = wld(wa + ((kc >> 6) * 4 + q) * 64 + lane);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(av[buf][tm][qq]), as_bf16x8(bv[buf][qq]), acc[tm], 0, 0, 0);
    };
    // (nh is a multiple of NBUF -- checked at launch.  No conditional load inside the steady-state loop: with one, hipcc's
    //  wait-count pass takes the minimum over both paths and the loop waits on vmcnt(0), i.e. prefetches nothing.)
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i) load(i, i);
    int t0 = 0;
    for (; t0 + 2 * NBUF <= nh; t0 += NBUF) {
#pragma unroll
        for (int u = 0; u < NBUF; ++u) {
            load((u + NBUF - 1) % NBUF, t0 + u + NBUF - 1);
            mma(u);
            __builtin_amdgcn_sched_barrier(0);         // keep load / MFMA groups interleaved (the scheduler would hoist all loads of the body)
        }
    }
    load(NBUF - 1, t0 + NBUF - 1);
#pragma unroll
    for (int u = 0; u < NBUF; ++u) mma(u);
    // accumulator register i of lane (r, h): row 8 (i / 4) + 4 h + (i % 4) of the tile, column r
    const int n = n0 + r;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + 32 * tm + 8 * (i >> 2) + 4 * h + (i & 3);
            if (m < a.M && n < a.N) a.slab[((long)quarter * a.M + m) * a.N + n] = acc[tm][i];
        }
}
