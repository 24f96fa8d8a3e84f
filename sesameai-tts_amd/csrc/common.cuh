// Shared device helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                       // raw bf16 bits
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

// round-to-nearest-even, NaN-preserving (hipcc emits v_cvt_pk_bf16_f32 for the cast)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float round_bf(float f) { return bf2f(f2bf(f)); }

// low / high bf16 of a packed dword as fp32
__device__ __forceinline__ float lo2f(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi2f(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// acc += a.lo*b.lo + a.hi*b.hi  (v_dot2c_f32_bf16: products of bf16 are exact in fp32)
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}
__device__ __forceinline__ float dot8(const uint4& a, const uint4& b, float acc) {
    acc = dot2(a.x, b.x, acc);
    acc = dot2(a.y, b.y, acc);
    acc = dot2(a.z, b.z, acc);
    acc = dot2(a.w, b.w, acc);
    return acc;
}

// Wave64 reductions on the DPP network (no LDS round trip like ds_bpermute): butterflies inside
// each row of 16 lanes (quad_perm xor1, xor2, row_half_mirror, row_mirror), then row_bcast:15 /
// row_bcast:31 carry the row sums upward, so the total lands in row 3 and is broadcast from
// lane 63 through an SGPR.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                                CTRL, ROW_MASK, 0xF, false));
}
// 16 OCP-e4m3 weights (one uint4) x 16 bf16 activations (two uint4): v_cvt_scalef32_pk_bf16_fp8 turns
// two fp8 bytes into a packed bf16 pair in one instruction (exact: e4m3 has 3 mantissa bits), which
// then feeds v_dot2c_f32_bf16 like a bf16 weight would.
__device__ __forceinline__ float dot16_fp8(const uint4& w, const uint4& x0, const uint4& x1, float acc) {
    const uint32_t wd[4] = {w.x, w.y, w.z, w.w};
    const uint32_t xd[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const bf16x2_t lo = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wd[d], 1.0f, false);
        const bf16x2_t hi = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(wd[d], 1.0f, true);
        acc = __builtin_amdgcn_fdot2_f32_bf16(lo, __builtin_bit_cast(bf16x2_t, xd[2 * d]), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(hi, __builtin_bit_cast(bf16x2_t, xd[2 * d + 1]), acc, false);
    }
    return acc;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1, 0xF>(0.f, v);       // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xF>(0.f, v);       // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xF>(0.f, v);      // row_half_mirror
    v += dpp_f<0x140, 0xF>(0.f, v);      // row_mirror
    v += dpp_f<0x142, 0xA>(0.f, v);      // row_bcast:15 -> rows 1,3
    v += dpp_f<0x143, 0xC>(0.f, v);      // row_bcast:31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_f<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_f<0x141, 0xF>(v, v));
    v = fmaxf(v, dpp_f<0x140, 0xF>(v, v));
    v = fmaxf(v, dpp_f<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_f<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// integer min over the wave on the DPP network (same butterfly as wave_max)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<0xB1, 0xF>(v, v));
    v = min(v, dpp_i<0x4E, 0xF>(v, v));
    v = min(v, dpp_i<0x141, 0xF>(v, v));
    v = min(v, dpp_i<0x140, 0xF>(v, v));
    v = min(v, dpp_i<0x142, 0xA>(v, v));
    v = min(v, dpp_i<0x143, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
    v = max(v, dpp_i<0xB1, 0xF>(v, v));
    v = max(v, dpp_i<0x4E, 0xF>(v, v));
    v = max(v, dpp_i<0x141, 0xF>(v, v));
    v = max(v, dpp_i<0x140, 0xF>(v, v));
    v = max(v, dpp_i<0x142, 0xA>(v, v));
    v = max(v, dpp_i<0x143, 0xC>(v, v));
    return __builtin_amdgcn_readlane(v, 63);
}
// exclusive prefix sum over the wave of a small non-negative count (< 2^BITS per lane), bit-sliced: one ballot and one
// mbcnt per bit, no LDS round trips (a __shfl_up scan is six ds_bpermute)
template <int BITS>
__device__ __forceinline__ int wave_excl_scan_small(int cnt) {
    int excl = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const unsigned long long m = __ballot((cnt >> b) & 1);
        excl += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << b;
    }
    return excl;
}

// streamed-once weight load: non-temporal keeps the 1.9 GB backbone stream from evicting
// the depth decoder's 222 MB out of the 256 MB Infinity Cache.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ldg16(const uint4* p) {
    if (NT) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *p;
}

// Philox4x32-10 counter RNG (one call = 4 x 32 random bits)
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
        uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0; key.y += W1;
    }
    return ctr;
}

// RMSNorm of one row by one wave (torchtune rounding: fp32 normalise -> bf16 -> * bf16 scale).  Shared by
// k_rmsnorm_rows (mm.cuh) and the sampler's fused "next decoder input" norm so both give the same bits.
__device__ __forceinline__ long xp_off(int m, int k, long K);
// out_row >= 0: `out` is the base of an operand-order buffer (xp_off) and the row is written as row `out_row` of it
__device__ __forceinline__ void rmsnorm_row_wave(const bf16_t* x, int K, const bf16_t* scale, float eps, bf16_t* out, int lane, int out_row = -1) {
    const uint4* src = reinterpret_cast<const uint4*>(x);
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
        if (out_row >= 0) *reinterpret_cast<uint4*>(out + xp_off(out_row, 8 * c, K)) = o;
        else reinterpret_cast<uint4*>(out)[c] = o;
    }
}

// Activations of the batched DECODE steps are stored in matrix-core operand order (like the packed weights): row m,
// column k of a [rows][K] buffer lives at xp_off(m, k, K).  Inside a 32-row tile the 16-byte piece (chunk, q, h, r)
// is at ((chunk*4 + q)*64 + h*32 + r), i.e. exactly the piece lane (r, h) feeds to v_mfma_f32_32x32x16_bf16 at step q
// of chunk `chunk` -- the consumer's wave load is one contiguous 1 KB read instead of a 64-cache-line gather
// (gate/up 9.25 -> 8.60 us, down 6.61 -> 6.23, o-proj 3.49 -> 3.32 at M = 32).
__device__ __forceinline__ long xp_off(int m, int k, long K) {
    return (long)(m >> 5) * 32 * K + ((((long)(k >> 6) * 4 + ((k & 31) >> 3)) * 64 + ((k >> 5) & 1) * 32 + (m & 31)) << 3) + (k & 7);
}

// floats per split-key partial (o[HD], m, l): rounded up to whole 128-byte lines, so the partials of different (row, KV head) merge
// groups never share a cache line -- an early merger on one XCD cannot pull a line into its L2 that a neighbouring group's producer
// is still writing (the in-kernel merge then does not depend on how sc1 loads treat a line already resident in L2)
#define ATTN_PS(HD_) ((((HD_) + 2) + 31) / 32 * 32)

