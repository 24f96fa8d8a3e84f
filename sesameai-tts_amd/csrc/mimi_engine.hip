// Mimi codec decode on gfx950 (fp32): RVQ gather-sum + output projections, depthwise
// upsample, 8-layer causal transformer (window 250, interleaved RoPE, LayerScale) and the
// SEANet conv decoder.  C ABI in include/mimi_hip.h.
//
// Replaces moshi 0.2.2 MimiModel.decode as called at sesameai/generator.py:116,299 and
// tts_service.py:245 (structure: SURVEY.md App. A.3; checked against oracle/mimi_ref.py).
//
// Layout: every activation is token-major [time][channels] fp32, so the channel (reduction)
// axis is contiguous for loads and the output-channel axis is contiguous for stores.  Every
// linear layer, causal Conv1d and ConvTranspose1d is ONE kernel: a multi-tap GEMM
//     out[t*phases + p][co] = bias[co] + sum_tap sum_ci act(x[t + shift_tap][ci]) * w[p][tap][co][ci]
// on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32, bit-identical to an fmaf chain), one
// wave per 32(time) x 32(channel) tile, operands straight from L2 in 64-byte contiguous pieces
// (the K order inside a 32-wide chunk is permuted so each lane reads 16 consecutive floats).
// A transposed conv with k = 2*stride is `stride` phase-GEMMs with 2 taps each.  Buffers that
// feed a causal op carry `hist` rows of left context in front of row 0: zero for a stateless
// decode, the previous call's tail for a stateful stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/mimi_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
static thread_local std::string g_mimi_err;

#define MAX_TAPS 16

// K-split workspace of a handle (mimi_create): partial tiles [MIMI_KSPLIT][cap_rows][cap_cols] fp32 + one ticket per output tile
#define MIMI_KSPLIT 8                   // default number of K slices (MIMI_KSPLIT=n in the environment: 0 = off, 2 / 4 / 8; measured: profiles/r05/mimi_ksplit_ab.txt)
#define MIMI_KSPLIT_MAX 8
#define MIMI_KSPLIT_MIN_ITERS 64        // deep products only: (taps * C_in / 32) iterations, e.g. the K = 2048 linear (64), the 7-tap 512 -> 1024 conv (112)
struct KSplitWs { float* part; int* ticket; long cap_rows; int cap_cols, n_tickets, ksplit; };

struct GemmArgs {
    const float* x; long ldx;       // A row for output time t, tap j: x + (t * in_stride + shift0 + j * dshift) * ldx
    int in_stride;                  // 1, or the stride of a down-sampling conv
    int edge, row_lo, row_hi;       // edge 0: rows are always in range (left context stored in front of row 0);
                                    // 1: rows outside [row_lo,row_hi) read as zero; 2: clamped (replicate padding)
    int T_in, C_in, C_out;
    const float* w;                 // [phases][taps][C_out][C_in]
    const float* bias;
    int taps, phases;
    int shift0, dshift;             // tap j reads row t * in_stride + shift0 + j * dshift
    int act_out;                    // 1 = exact GELU
    const float* col_scale;         // optional per-output-channel scale (LayerScale)
    const float* resid; long ldr;   // optional residual, indexed like out
    float* out; long ldo;
    // q|k|v projection of the transformer (round 4: was a k_rope_split launch behind every in_proj): C_out = 3 d; channels
    // [0,d) -> rope_q[t] with RoPE, [d,2d) -> rope_k[rope_offset + t] with RoPE, [2d,3d) -> rope_v[rope_offset + t]; head_dim 64,
    // interleaved pairs (the partner channel sits in the neighbouring lane)
    float *rope_q, *rope_k, *rope_v; const float* rope_freqs; int rope_offset, rope_d;
    // K split over blockIdx.z (round 5; phases == 1 only): block (tile, ks) runs iterations [ks, ks + 1) * iters / ksplit, parks its fp32
    // partial tile in kpart[ks][T_in][C_out] with write-through stores, and the LAST of a tile's ksplit blocks to arrive adds the
    // partials in ks order and runs the epilogue.  ksplit is a function of (taps, C_in) alone: streaming == whole decode, bit for bit.
    int ksplit; float* kpart; int* kticket;
};

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expf(v) - 1.0f; }

// One block = one 32 (time) x 32 (channel) output tile; its G32_NW waves split the (tap, 32-wide k chunk) iteration space
// round-robin and add their partial tiles through LDS in a fixed order.  (One wave per tile walked all taps x C_in/32
// chunks serially: at 20 transformer tokens a K = 2048 linear was 64 dependent load->MFMA round trips on 16 blocks.)
// The split depends only on (taps, C_in), never on T, so streaming decode stays bit-identical to whole decode.
#ifndef G32_NW
#define G32_NW 8
#endif
// ELU = the input activation as a compile-time choice.
// Round 4 (tools/dbg/mimi_chunk_timeline.py, profiles/r04/mimi_chunk10_timeline.txt): at 20 transformer tokens every iteration was
// TWO dependent memory round trips -- the tap's row shift came from an indexed array in the argument block (a vector load the
// operand addresses waited for), then the operands -- and every MFMA sat behind a run-time "ELU?" branch.  Now the shift is
// arithmetic (shift0 + j * dshift: every convolution here is an arithmetic progression of taps) and the iteration index is scalar.
// Measured and NOT kept: four iterations' operands requested at once (slower: these launches are bound by the misses 16..64 CUs
// keep in flight and by ~4.7 us of dependent-launch latency each, not by the depth of one wave's queue); the LayerNorm folded
// into the following product (every block normalising its 32 rows itself: 12.7 us against 4.7 + 8.0 for the two launches -- equal);
// the weights re-tiled into MFMA operand order so that every wave load is one contiguous 1 KB (the K = 2048 linear 24 us against
// 19.5 row-major: with one 32-row tile these launches sit on 16..64 CUs and are bound by the fp32 matrix pipe -- 128 dependent
// 64-cycle v_mfma_f32_32x32x2_f32 per wave, two waves per SIMD = 6.8 us -- plus the ~4.7 us of a dependent launch, not by loads).
template <bool ELU>
__global__ __launch_bounds__(64 * G32_NW) void k_gemm32(const GemmArgs a) {
    __shared__ float red[G32_NW][16][64];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int ks = a.ksplit > 1 ? (int)blockIdx.z : 0;
    const int t0 = blockIdx.x * 32, n0 = blockIdx.y * 32, p = a.ksplit > 1 ? 0 : (int)blockIdx.z;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool trow_ok = (t0 + r) < a.T_in;
    const bool nrow_ok = (n0 + r) < a.C_out;
    const int trow = trow_ok ? (t0 + r) : (a.T_in - 1);
    const int nrow = nrow_ok ? (n0 + r) : (a.C_out - 1);
    const int kchunks = a.C_in / 32, iters = a.taps * kchunks;
    const int arow0 = trow * a.in_stride + a.shift0;
    const float* const wrow = a.w + ((long)p * a.taps * a.C_out + nrow) * a.C_in + h * 16;
    const int it_lo = a.ksplit > 1 ? ks * (iters / a.ksplit) : 0, it_hi = a.ksplit > 1 ? it_lo + iters / a.ksplit : iters;
    for (int it = it_lo + wave; it < it_hi; it += G32_NW) {
        const int j = it / kchunks, kc = (it - j * kchunks) * 32;
        int arow = arow0 + j * a.dshift;
        bool use = trow_ok;
        if (a.edge == 1) { use = trow_ok && arow >= a.row_lo && arow < a.row_hi; arow = min(max(arow, a.row_lo), a.row_hi - 1); }
        else if (a.edge == 2) arow = min(max(arow, a.row_lo), a.row_hi - 1);
        const float* xa = a.x + (long)arow * a.ldx + h * 16 + kc;
        const float* wb = wrow + (long)j * a.C_out * a.C_in + kc;
        float4 av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = *reinterpret_cast<const float4*>(xa + q * 4);
            bv[q] = *reinterpret_cast<const float4*>(wb + q * 4);
        }
        float af[16], bf[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            af[q * 4 + 0] = av[q].x; af[q * 4 + 1] = av[q].y; af[q * 4 + 2] = av[q].z; af[q * 4 + 3] = av[q].w;
            bf[q * 4 + 0] = bv[q].x; bf[q * 4 + 1] = bv[q].y; bf[q * 4 + 2] = bv[q].z; bf[q * 4 + 3] = bv[q].w;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float va = af[s];
            if (ELU) va = elu1(va);
            va = use ? va : 0.f;
            const float vb = nrow_ok ? bf[s] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, vb, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    // C/D map of the 32x32 tile: col = lane & 31 (channel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (time);
    // wave g finishes registers 4g .. 4g+3
    const int ch = n0 + r;
    float part4[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.ksplit > 1) {
        // park this K slice's partial tile (waves 0..3: registers 4w..4w+3, as in the epilogue), then take a ticket: the tile's last
        // block reduces.  Write-through (sc1) stores + vmcnt(0) before the ticket, sc1 loads after it: the guide's counter hand-off.
        if (wave < 4 && ch < a.C_out) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int reg = wave * 4 + i;
                const int t = t0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < G32_NW; ++w) sum += red[w][reg][lane];      // fixed order
                if (t < a.T_in) {
                    float* dst = a.kpart + ((long)ks * a.T_in + t) * a.C_out + ch;
                    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(sum) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __shared__ int s_last;
        if (threadIdx.x == 0) {
            int* tk = a.kticket + blockIdx.y * gridDim.x + blockIdx.x;
            const int got = __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = got == a.ksplit - 1;
            if (s_last) __hip_atomic_store(tk, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
        __syncthreads();
        if (!s_last) return;
        if (wave < 4 && ch < a.C_out) {
            float pv[4][MIMI_KSPLIT_MAX];                                            // every load in flight, ONE wait
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int reg = wave * 4 + i;
                const int t = min(t0 + (reg & 3) + 8 * (reg >> 2) + 4 * h, a.T_in - 1);
#pragma unroll
                for (int q = 0; q < MIMI_KSPLIT_MAX; ++q) {
                    pv[i][q] = 0.f;
                    if (q < a.ksplit) {
                        const float* src = a.kpart + ((long)q * a.T_in + t) * a.C_out + ch;
                        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pv[i][q]) : "v"(src) : "memory");
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float acc4 = 0.f;
#pragma unroll
                for (int q = 0; q < MIMI_KSPLIT_MAX; ++q) { asm volatile("" : "+v"(pv[i][q])); if (q < a.ksplit) acc4 += pv[i][q]; }     // slice order: deterministic whoever arrived last
                part4[i] = acc4;
            }
        }
    }
    if (ch >= a.C_out || wave >= 4) return;
    const float bias = a.bias ? a.bias[ch] : 0.f;
    const float cs = a.col_scale ? a.col_scale[ch] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int reg = wave * 4 + i;
        const int t = t0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
        if (t >= a.T_in) continue;
        float sum = 0.f;
        if (a.ksplit > 1) sum = part4[i];
        else {
#pragma unroll
            for (int w = 0; w < G32_NW; ++w) sum += red[w][reg][lane];          // fixed order
        }
        const long orow = (long)t * a.phases + p;
        float v = sum + bias;
        if (a.rope_q) {                                                 // (whole 32-channel tiles: every lane of the half-wave is here)
            const float partner = __shfl_xor(v, 1, 64);
            const int sec = ch / a.rope_d, cc = ch - sec * a.rope_d;
            if (sec == 2) { a.rope_v[(long)(a.rope_offset + t) * a.rope_d + cc] = v; continue; }
            const float ang = (float)(a.rope_offset + t) * a.rope_freqs[(cc & 63) >> 1];
            const float c = cosf(ang), s = sinf(ang);
            const float o = (cc & 1) ? partner * s + v * c : v * c - partner * s;
            if (sec == 0) a.rope_q[(long)t * a.rope_d + cc] = o;
            else a.rope_k[(long)(a.rope_offset + t) * a.rope_d + cc] = o;
            continue;
        }
        if (a.act_out == 1) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        if (a.col_scale) v = cs * v;
        if (a.resid) v = a.resid[orow * a.ldr + ch] + v;
        a.out[orow * a.ldo + ch] = v;
    }
}

// RVQ decode: q_first = emb_0[c0]; q_rest = sum_{k>=1} emb_k[c_k]; out = Wf q_first + Wr q_rest
__global__ __launch_bounds__(256) void k_rvq(const int* codes, long stride_k, long stride_t, int T, int ncb, int nsem,
                                             int cbsize, int cbdim, int hidden, const float* books, const float* pf,
                                             const float* pr, float* out, long ldo) {
    extern __shared__ float q[];                 // [2][cbdim] sums, then [ncb] clamped codes
    const int t = blockIdx.x;
    int* cidx = reinterpret_cast<int*>(q + 2 * cbdim);
    for (int k = threadIdx.x; k < ncb; k += blockDim.x)                  // the frame's codes once, not once per thread
        cidx[k] = min(max(codes[k * stride_k + t * stride_t], 0), cbsize - 1);
    __syncthreads();
    for (int d = threadIdx.x; d < cbdim; d += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
        for (int k = 0; k < ncb; ++k) {                                   // independent row loads, summed in codebook order
            const float v = books[((long)k * cbsize + cidx[k]) * cbdim + d];
            if (k < nsem) s1 += v; else s2 += v;
        }
        q[d] = s1; q[cbdim + d] = s2;
    }
    __syncthreads();
    // blockIdx.y = group of 64 output channels (round 4: with one block per frame a 10-frame chunk was 10 blocks pulling 1 MB of
    // projection weights each, 30 us); the lookup-sum above is repeated per group (32 KB of codebook rows from L2)
    const int c_lo = blockIdx.y * 64, c_hi = min(c_lo + 64, hidden);
    for (int c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int d = 0; d < cbdim; ++d) {                                 // (unroll 32 -- 64 loads in flight -- was measured: 4x SLOWER)
            a1 = fmaf(pf[(long)d * hidden + c], q[d], a1);
            a2 = fmaf(pr[(long)d * hidden + c], q[cbdim + d], a2);
        }
        out[(long)t * ldo + c] = a1 + a2;
    }
}

// depthwise ConvTranspose1d k4 s2, causal: out[2t+p][c] = x[t][c] w[p][0][c] + x[t-1][c] w[p][1][c]
// (+ the K-split tiles' arrival tickets back to zero: they reset themselves, but only in a launch that completes -- the first kernel of every
//  pass puts them at zero so that an aborted launch cannot leave a tile one arrival ahead; no launch of its own)
__global__ void k_upsample(const float* x, long ldx, int T, int C, const float* w, float* out, long ldo, int* tickets, int n_tickets) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tickets != nullptr)
        for (long j = i; j < n_tickets; j += (long)gridDim.x * blockDim.x) tickets[j] = 0;
    if (i >= (long)T * 2 * C) return;
    const int c = (int)(i % C);
    const long n = i / C;
    const long t = n >> 1;
    const int p = (int)(n & 1);
    out[n * ldo + c] = x[t * ldx + c] * w[(p * 2 + 0) * C + c] + x[(t - 1) * ldx + c] * w[(p * 2 + 1) * C + c];
}

// One wave per row; the row is read ONCE into registers (up to 16 values per lane, d <= 1024) with w and b loads in
// flight beside it -- the three-pass version paid three dependent global round trips per call (6.7 us x 16 per decode).
__global__ __launch_bounds__(64) void k_layernorm(const float* x, int d, const float* w, const float* b, float eps, float* out) {
    const long row = blockIdx.x;
    const float* xr = x + row * d;
    const int lane = threadIdx.x;
    float v[16], wv[16], bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        const bool in = i < d;
        v[j] = in ? xr[i] : 0.f; wv[j] = in ? w[i] : 0.f; bv[j] = in ? b[i] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];                     // same per-lane order as before: i = lane, lane + 64, ...
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float dlt = v[j] - mean; if (lane + 64 * j < d) q += dlt * dlt; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = 1.0f / sqrtf(q / d + eps);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        if (i < d) out[row * d + i] = (v[j] - mean) * rstd * wv[j] + bv[j];
    }
}

// causal windowed attention, one wave per (query, head); head_dim 64
__global__ __launch_bounds__(64) void k_mimi_attn(const float* q, const float* kc, const float* vc, int d, int offset,
                                                  int context, float* out) {
    __shared__ float pbuf[1024];
    const int i = blockIdx.x, hh = blockIdx.y, lane = threadIdx.x;
    const int pi = offset + i;
    const int lo = max(0, pi - context + 1);
    const int nk = pi - lo + 1;
    const float* qr = q + (long)i * d + hh * 64;
    float qv[64];
#pragma unroll
    for (int e = 0; e < 64; e += 4) {
        const float4 v = *reinterpret_cast<const float4*>(qr + e);
        qv[e] = v.x; qv[e + 1] = v.y; qv[e + 2] = v.z; qv[e + 3] = v.w;
    }
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) {
        const float* kr = kc + (long)(lo + j) * d + hh * 64;
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 64; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(kr + e);
            s = fmaf(qv[e], v.x, s); s = fmaf(qv[e + 1], v.y, s); s = fmaf(qv[e + 2], v.z, s); s = fmaf(qv[e + 3], v.w, s);
        }
        s *= 0.125f;
        pbuf[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < nk; j += 64) { const float pw = expf(pbuf[j] - mx); pbuf[j] = pw; sum += pw; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    __syncthreads();
    float o = 0.f;
    for (int j = 0; j < nk; ++j) o = fmaf(pbuf[j], vc[(long)(lo + j) * d + hh * 64 + lane], o);
    out[(long)i * d + hh * 64 + lane] = o / sum;
}

// final Conv1d (taps x C_in -> 1 channel) with ELU on the input; one thread per output sample
__global__ void k_conv_out(const float* x, long ldx, long T, int C_in, int taps, const float* w /*[taps][1][C_in]*/,
                           const float* bias, float* pcm) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float acc = bias ? bias[0] : 0.f;
    for (int j = 0; j < taps; ++j) {
        const float* xr = x + (t + j - (taps - 1)) * ldx;
        const float* wr = w + (long)j * C_in;
        for (int c = 0; c < C_in; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float4 u = *reinterpret_cast<const float4*>(wr + c);
            acc = fmaf(elu1(v.x), u.x, acc); acc = fmaf(elu1(v.y), u.y, acc);
            acc = fmaf(elu1(v.z), u.z, acc); acc = fmaf(elu1(v.w), u.w, acc);
        }
    }
    pcm[t] = acc;
}

// slide the left-context window of a [hist + T][C] buffer: new hist rows = last hist rows of (old hist ++ new T rows)
__global__ void k_slide_hist(float* base /*row -hist*/, int hist, long T, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    for (int r = 0; r < hist; ++r) base[(long)r * C + c] = base[((long)r + T) * C + c];   // ascending r: source is always ahead
}

// ---- encode-side kernels ---------------------------------------------------------------
// first encoder conv (1 channel in): out[t][co] = b[co] + sum_k w[k][co] * wav[t - (taps-1) + k]
__global__ void k_enc_conv_in(const float* wav, long n, int taps, int C, const float* w, const float* b, float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const long t = i / C;
    const int co = (int)(i % C);
    float acc = b[co];
    for (int k = 0; k < taps; ++k) {
        const long ti = t - (taps - 1) + k;
        if (ti >= 0) acc = fmaf(wav[ti], w[k * C + co], acc);
    }
    out[i] = acc;
}

// nearest centroid of one RVQ level + residual update: score[t][c] = r_t . e_c (from the GEMM),
// code = argmin_c (|e_c|^2 - 2 score) (first index on ties), r_t -= e_code.   grid = T, block = 256
__global__ __launch_bounds__(256) void k_rvq_pick(const float* score, int ncodes, const float* sqnorm, const float* book, int dim,
                                                  float* resid, int32_t* codes, long code_stride_t) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    __shared__ int pick;
    const int t = blockIdx.x, tid = threadIdx.x;
    float best = INFINITY; int idx = 0x7fffffff;
    for (int c = tid; c < ncodes; c += 256) {
        const float dsc = sqnorm[c] - 2.0f * score[(long)t * ncodes + c];
        if (dsc < best) { best = dsc; idx = c; }             // ascending c per thread: keeps the first minimum
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ob < best || (ob == best && oi < idx)) { best = ob; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int x = 1; x < 4; ++x) if (bv[x] < best || (bv[x] == best && bi[x] < idx)) { best = bv[x]; idx = bi[x]; }
        pick = idx; codes[(long)t * code_stride_t] = idx;
    }
    __syncthreads();
    const int e = pick;
    for (int d = tid; d < dim; d += 256) resid[(long)t * dim + d] -= book[(long)e * dim + d];
}

// ---------------------------------------------------------------------------------------
struct HBuf {            // activation buffer with `hist` rows of left context in front of row 0
    float* base = nullptr;
    int hist = 0, C = 0;
    float* row0() const { return base + (long)hist * C; }
};


struct MimiDecoder {
    MimiConfig cfg;
    MimiWeights w;
    int max_frames;
    long cap_tokens;                    // transformer KV capacity (tokens)
    int offset;                         // tokens already in the KV caches (stateful stream)
    HBuf rvq, a0, s0;                   // rvq out (hist 1), transformer out (hist kernel-1), conv_in out (hist 1)
    HBuf u[MIMI_MAX_STAGES], xj[MIMI_MAX_STAGES];
    float *r1[MIMI_MAX_STAGES];
    float *tok, *ln, *q, *att, *ffn, *kc, *vc;
    KSplitWs ksw = {nullptr, nullptr, 0, 0, 0, 0};
    hipStream_t cap_stream = nullptr;   // graph capture of decode_middle
    hipGraphExec_t mid_exec[65] = {};   // by T (stateless decodes of up to MIMI_GRAPH_MAX_T frames)
    int mid_uses[65] = {};
    std::string err;
};

#define MCHK(h, x)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            char b_[256];                                                                              \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (h) (h)->err = b_; else g_mimi_err = b_;                                                \
            return -2;                                                                                 \
        }                                                                                              \
    } while (0)

static int mfail(MimiDecoder* h, const char* msg) {
    if (h) h->err = msg; else g_mimi_err = msg;
    return -1;
}

static hipError_t alloc_hbuf(HBuf& b, int hist, long rows, int C) {
    b.hist = hist; b.C = C;
    hipError_t e = hipMalloc((void**)&b.base, (size_t)(hist + rows) * C * 4);
    if (e != hipSuccess) return e;
    return hipMemset(b.base, 0, (size_t)(hist + rows) * C * 4);
}

extern "C" void csm_warn_unknown_switches(void);     // csm_engine.hip: names under CSM_ / MIMI_ that no switch reads, once per process

extern "C" int mimi_create(const MimiConfig* cfg, const MimiWeights* w, int max_frames, int reserved, mimi_handle* out) {
    csm_warn_unknown_switches();
    (void)reserved;
    if (!cfg || !w || !out || max_frames < 1) return mfail(nullptr, "mimi_create: null/invalid argument");
    if (cfg->hidden % 32 || cfg->hidden > 1024 || cfg->codebook_dim % 4 || cfg->tr_ffn % 32 || cfg->hidden / cfg->tr_heads != 64)
        return mfail(nullptr, "mimi_create: hidden%32, hidden<=1024, tr_ffn%32 and head_dim==64 required");
    if (cfg->n_stages < 1 || cfg->n_stages > MIMI_MAX_STAGES || cfg->tr_layers > MIMI_MAX_TR_LAYERS || cfg->tr_context > 1024)
        return mfail(nullptr, "mimi_create: too many stages/layers or context > 1024");
    if (cfg->kernel > MAX_TAPS || cfg->res_kernel > MAX_TAPS) return mfail(nullptr, "mimi_create: kernel too wide");
    int c = cfg->n_filters << cfg->n_stages;
    for (int j = 0; j < cfg->n_stages; ++j) { c /= 2; if ((c / 2) % 32) return mfail(nullptr, "mimi_create: SEANet channels must stay multiples of 32"); }
    MimiDecoder* m = new MimiDecoder();
    m->cfg = *cfg; m->w = *w; m->max_frames = max_frames; m->offset = 0;
    const int d = cfg->hidden;
    const long T2 = 2L * max_frames;
    m->cap_tokens = T2;
    MCHK((MimiDecoder*)nullptr, alloc_hbuf(m->rvq, 1, max_frames, d));
    MCHK((MimiDecoder*)nullptr, alloc_hbuf(m->a0, cfg->kernel - 1, T2, d));
    int C = cfg->n_filters << cfg->n_stages;
    MCHK((MimiDecoder*)nullptr, alloc_hbuf(m->s0, 1, T2, C));
    long Tj = T2;
    for (int j = 0; j < cfg->n_stages; ++j) {
        Tj *= cfg->ratios[j]; C /= 2;
        MCHK((MimiDecoder*)nullptr, alloc_hbuf(m->u[j], cfg->res_kernel - 1, Tj, C));
        MCHK((MimiDecoder*)nullptr, hipMalloc((void**)&m->r1[j], (size_t)Tj * (C / 2) * 4));
        const int nh = (j + 1 < cfg->n_stages) ? 1 : cfg->last_kernel - 1;
        MCHK((MimiDecoder*)nullptr, alloc_hbuf(m->xj[j], nh, Tj, C));
    }
#define A4(p, n) MCHK((MimiDecoder*)nullptr, hipMalloc((void**)&(p), (size_t)(n) * 4))
    A4(m->tok, T2 * d); A4(m->ln, T2 * d); A4(m->q, T2 * d); A4(m->att, T2 * d);
    A4(m->ffn, T2 * (cfg->tr_ffn > cfg->codebook_size ? cfg->tr_ffn : cfg->codebook_size));   // also the RVQ score buffer
    A4(m->kc, (long)cfg->tr_layers * m->cap_tokens * d); A4(m->vc, (long)cfg->tr_layers * m->cap_tokens * d);
    {   // K-split workspace (k_gemm32): the deep products' partial tiles and tile tickets.  MIMI_KSPLIT=0 in the environment: no split.
        const char* ev = getenv("MIMI_KSPLIT");
        const int want = ev ? atoi(ev) : MIMI_KSPLIT;
        if (want == 2 || want == 4 || want == 8) {
            m->ksw.ksplit = want;
            const int ccap = cfg->n_filters << cfg->n_stages;                         // widest output of a split product (conv_in); >= hidden
            m->ksw.cap_rows = T2; m->ksw.cap_cols = ccap > d ? ccap : d;
            m->ksw.n_tickets = (int)((T2 + 31) / 32) * ((m->ksw.cap_cols + 31) / 32);
            A4(m->ksw.part, (long)want * m->ksw.cap_rows * m->ksw.cap_cols);
            MCHK((MimiDecoder*)nullptr, hipMalloc((void**)&m->ksw.ticket, (size_t)m->ksw.n_tickets * 4));
            MCHK((MimiDecoder*)nullptr, hipMemset(m->ksw.ticket, 0, (size_t)m->ksw.n_tickets * 4));
        }
    }
#undef A4
    MCHK((MimiDecoder*)nullptr, hipDeviceSynchronize());
    *out = m;
    return 0;
}

extern "C" void mimi_destroy(mimi_handle m) {
    if (!m) return;
    (void)hipFree(m->rvq.base); (void)hipFree(m->a0.base); (void)hipFree(m->s0.base);
    for (int j = 0; j < m->cfg.n_stages; ++j) { (void)hipFree(m->u[j].base); (void)hipFree(m->xj[j].base); (void)hipFree(m->r1[j]); }
    void* ps[] = {m->tok, m->ln, m->q, m->att, m->ffn, m->kc, m->vc, m->ksw.part, m->ksw.ticket};
    for (void* p : ps) (void)hipFree(p);
    for (hipGraphExec_t g : m->mid_exec) if (g) (void)hipGraphExecDestroy(g);
    if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
    delete m;
}

extern "C" const char* mimi_last_error(mimi_handle m) { return m ? m->err.c_str() : g_mimi_err.c_str(); }

// the left-context rows of every activation buffer, zeroed by ONE launch (was 11 hipMemsetAsync per stateless decode)
struct ZeroRegions { float* p[3 + 2 * MIMI_MAX_STAGES]; int n[3 + 2 * MIMI_MAX_STAGES]; int count; };
__global__ void k_zero_regions(const ZeroRegions z) {
    const int reg = blockIdx.y;
    if (reg >= z.count) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < z.n[reg]; i += gridDim.x * blockDim.x) z.p[reg][i] = 0.f;
}

extern "C" int mimi_reset_stream(mimi_handle m, void* stream) {
    if (!m) return -1;
    hipStream_t st = (hipStream_t)stream;
    ZeroRegions z;
    z.count = 0;
    auto add = [&](const HBuf& b) { if (b.hist) { z.p[z.count] = b.base; z.n[z.count] = b.hist * b.C; ++z.count; } };
    add(m->rvq); add(m->a0); add(m->s0);
    for (int j = 0; j < m->cfg.n_stages; ++j) { add(m->u[j]); add(m->xj[j]); }
    if (z.count) {
        hipLaunchKernelGGL(k_zero_regions, dim3(8, z.count), dim3(256), 0, st, z);
        MCHK(m, hipGetLastError());
    }
    m->offset = 0;
    return 0;
}

struct RopeOut { float *q, *k, *v; const float* freqs; int offset, d; };
static hipError_t gemm(hipStream_t st, const float* x, long ldx, long T_in, const float* w, const float* bias, int C_in,
                       int C_out, int taps, int phases, const int* shifts, int elu_in, int act_out, const float* col_scale,
                       const float* resid, long ldr, float* out, long ldo, int in_stride = 1, int edge = 0, int row_lo = 0,
                       int row_hi = 0, const RopeOut* rope = nullptr, const KSplitWs* ksw = nullptr) {
    GemmArgs a;
    memset(&a, 0, sizeof a);
    if (rope) { a.rope_q = rope->q; a.rope_k = rope->k; a.rope_v = rope->v; a.rope_freqs = rope->freqs; a.rope_offset = rope->offset; a.rope_d = rope->d; }
    a.in_stride = in_stride; a.edge = edge; a.row_lo = row_lo; a.row_hi = row_hi;
    a.x = x; a.ldx = ldx; a.T_in = (int)T_in; a.C_in = C_in; a.C_out = C_out; a.w = w; a.bias = bias; a.taps = taps;
    a.phases = phases;
    a.shift0 = shifts[0]; a.dshift = taps > 1 ? shifts[1] - shifts[0] : 0;
    for (int j = 0; j < taps; ++j) if (shifts[j] != a.shift0 + j * a.dshift) return hipErrorInvalidValue;   // (every convolution here is an arithmetic progression of taps)
    a.act_out = act_out; a.col_scale = col_scale; a.resid = resid; a.ldr = ldr; a.out = out; a.ldo = ldo;
    dim3 grid((unsigned)((T_in + 31) / 32), (unsigned)((C_out + 31) / 32), (unsigned)phases);
    // (the choice depends on the product's shape alone, never on T_in: a streamed chunk and the whole clip sum in the same order)
    const int iters = taps * (C_in / 32);
    if (ksw != nullptr && ksw->part != nullptr && phases == 1 && rope == nullptr && iters >= MIMI_KSPLIT_MIN_ITERS && iters % ksw->ksplit == 0) {
        // (the workspace is sized for the handle's max_frames at mimi_create, so a product the decoder accepts always fits; if one ever
        //  does not, it runs unsplit -- same value to fp32 rounding -- instead of failing the decode: ADVICE r5)
        if (T_in <= ksw->cap_rows && C_out <= ksw->cap_cols && (long)grid.x * grid.y <= ksw->n_tickets) {
            a.ksplit = ksw->ksplit; a.kpart = ksw->part; a.kticket = ksw->ticket;
            grid.z = (unsigned)ksw->ksplit;
        }
    }
    if (elu_in) hipLaunchKernelGGL(k_gemm32<true>, grid, dim3(64 * G32_NW), 0, st, a);
    else hipLaunchKernelGGL(k_gemm32<false>, grid, dim3(64 * G32_NW), 0, st, a);
    return hipGetLastError();
}

static hipError_t slide(const HBuf& b, long T, hipStream_t st) {
    if (!b.hist) return hipSuccess;
    hipLaunchKernelGGL(k_slide_hist, dim3((b.C + 255) / 256), dim3(256), 0, st, b.base, b.hist, T, b.C);
    return hipGetLastError();
}

// A decode is three pieces: front (the only kernel that reads the caller's codes), middle (everything between buffers the handle
// owns: up-sampling, transformer, SEANet stages -- ~70 launches whose arguments depend on T and the stream offset only) and back (the
// only kernel that writes the caller's PCM).  The middle of a STATELESS decode (offset 0) is replayed from a hipGraph per T.
static int decode_front(MimiDecoder* m, const int32_t* codes, long stride_k, long stride_t, int T, hipStream_t st) {
    const MimiConfig& c = m->cfg;
    const int d = c.hidden;
    // 1. RVQ lookup-sum + output projections -> rvq [T][d]
    hipLaunchKernelGGL(k_rvq, dim3(T, (d + 63) / 64), dim3(256), (size_t)(2 * c.codebook_dim + c.n_codebooks) * 4, st, codes, stride_k, stride_t, T, c.n_codebooks,
                       c.n_semantic, c.codebook_size, c.codebook_dim, d, m->w.codebooks, m->w.proj_first, m->w.proj_rest,
                       m->rvq.row0(), (long)d);
    MCHK(m, hipGetLastError());
    return 0;
}

static int decode_middle(MimiDecoder* m, int T, hipStream_t st) {
    const MimiConfig& c = m->cfg;
    const int d = c.hidden;
    const long T2 = 2L * T;
    const int zero = 0;
    // (ONE stream per handle: the K-split partial tiles and their tickets are per handle, include/mimi_hip.h)
    // 2. depthwise transposed conv x2 -> tok [2T][d]
    {
        const long n = T2 * d;
        hipLaunchKernelGGL(k_upsample, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->rvq.row0(), (long)d, T, d,
                           m->w.upsample, m->tok, (long)d, m->ksw.ticket, m->ksw.ticket ? m->ksw.n_tickets : 0);
        MCHK(m, hipGetLastError());
    }
    // 3. transformer
    for (int l = 0; l < c.tr_layers; ++l) {
        const MimiTrLayer& L = m->w.tr[l];
        float* kc = m->kc + (long)l * m->cap_tokens * d;
        float* vc = m->vc + (long)l * m->cap_tokens * d;
        hipLaunchKernelGGL(k_layernorm, dim3((unsigned)T2), dim3(64), 0, st, m->tok, d, L.ln1_w, L.ln1_b, c.norm_eps, m->ln);
        const RopeOut ro = {m->q, kc, vc, m->w.rope_freqs, m->offset, d};
        MCHK(m, gemm(st, m->ln, d, T2, L.in_proj, nullptr, d, 3 * d, 1, 1, &zero, 0, 0, nullptr, nullptr, 0, nullptr, 0, 1, 0, 0, 0, &ro));
        hipLaunchKernelGGL(k_mimi_attn, dim3((unsigned)T2, c.tr_heads), dim3(64), 0, st, m->q, kc, vc, d, m->offset, c.tr_context, m->att);
        MCHK(m, gemm(st, m->att, d, T2, L.out_proj, nullptr, d, d, 1, 1, &zero, 0, 0, L.ls1, m->tok, d, m->tok, d));
        hipLaunchKernelGGL(k_layernorm, dim3((unsigned)T2), dim3(64), 0, st, m->tok, d, L.ln2_w, L.ln2_b, c.norm_eps, m->ln);
        MCHK(m, gemm(st, m->ln, d, T2, L.lin1, nullptr, d, c.tr_ffn, 1, 1, &zero, 0, 1, nullptr, nullptr, 0, m->ffn, c.tr_ffn));
        float* dst = (l + 1 < c.tr_layers) ? m->tok : m->a0.row0();
        MCHK(m, gemm(st, m->ffn, c.tr_ffn, T2, L.lin2, nullptr, c.tr_ffn, d, 1, 1, &zero, 0, 0, L.ls2, m->tok, d, dst, d, 1, 0, 0, 0, nullptr, &m->ksw));
        MCHK(m, hipGetLastError());
    }
    // 4. SEANet decoder
    int shifts[MAX_TAPS];
    for (int j = 0; j < c.kernel; ++j) shifts[j] = j - (c.kernel - 1);
    const MimiConv& ci = m->w.conv_in;
    MCHK(m, gemm(st, m->a0.row0(), d, T2, ci.w, ci.bias, ci.c_in, ci.c_out, ci.taps, 1, shifts, 0, 0, nullptr, nullptr, 0,
                 m->s0.row0(), ci.c_out, 1, 0, 0, 0, nullptr, &m->ksw));
    const float* xin = m->s0.row0();
    int Cin = ci.c_out;
    long Tj = T2;
    for (int j = 0; j < c.n_stages; ++j) {
        const MimiConv &up = m->w.up[j], &r1 = m->w.res1[j], &r2 = m->w.res2[j];
        const int tshift[2] = {0, -1};
        // ELU -> ConvTranspose1d (stride phases, 2 taps each)
        MCHK(m, gemm(st, xin, Cin, Tj, up.w, up.bias, up.c_in, up.c_out, 2, up.phases, tshift, 1, 0, nullptr, nullptr, 0,
                     m->u[j].row0(), up.c_out));
        Tj *= up.phases;
        // residual block: u + conv_k1(ELU(conv_k3(ELU(u))))
        for (int k = 0; k < r1.taps; ++k) shifts[k] = k - (r1.taps - 1);
        MCHK(m, gemm(st, m->u[j].row0(), up.c_out, Tj, r1.w, r1.bias, r1.c_in, r1.c_out, r1.taps, 1, shifts, 1, 0, nullptr, nullptr,
                     0, m->r1[j], r1.c_out));
        MCHK(m, gemm(st, m->r1[j], r1.c_out, Tj, r2.w, r2.bias, r2.c_in, r2.c_out, 1, 1, &zero, 1, 0, nullptr, m->u[j].row0(),
                     up.c_out, m->xj[j].row0(), r2.c_out));
        xin = m->xj[j].row0(); Cin = r2.c_out;
    }
    return 0;
}

static int decode_back(MimiDecoder* m, int T, float* pcm, hipStream_t st) {
    const MimiConfig& c = m->cfg;
    long Tj = 2L * T;
    for (int j = 0; j < c.n_stages; ++j) Tj *= m->w.up[j].phases;
    const HBuf& last = m->xj[c.n_stages - 1];
    const MimiConv& co = m->w.conv_out;
    hipLaunchKernelGGL(k_conv_out, dim3((unsigned)((Tj + 255) / 256)), dim3(256), 0, st, last.row0(), (long)last.C, Tj, co.c_in, co.taps, co.w,
                       co.bias, pcm);
    MCHK(m, hipGetLastError());
    return 0;
}

// chunk sizes whose middle is worth a graph (a streaming chunk is 10 frames, the last one of an utterance 1..9): captured at the
// SECOND stateless decode of a T, replayed from then on.  Longer decodes (whole utterances, a different T every time) stay eager.
static const int MIMI_GRAPH_MAX_T = getenv("MIMI_GRAPH_MAX_T") ? atoi(getenv("MIMI_GRAPH_MAX_T")) : 32;

static int decode_one(MimiDecoder* m, const int32_t* codes, long stride_k, long stride_t, int T, float* pcm, hipStream_t st, bool stateless) {
    int rc = decode_front(m, codes, stride_k, stride_t, T, st);
    if (rc) return rc;
    if (stateless && T <= MIMI_GRAPH_MAX_T && T < (int)(sizeof m->mid_uses / sizeof m->mid_uses[0]) && m->offset == 0) {
        if (!m->mid_exec[T] && ++m->mid_uses[T] >= 2) {
            if (!m->cap_stream) MCHK(m, hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking));
            hipGraph_t g = nullptr;
            MCHK(m, hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
            const int rc_mid = decode_middle(m, T, m->cap_stream);
            const hipError_t e2 = hipStreamEndCapture(m->cap_stream, &g);
            if (rc_mid) { if (g) (void)hipGraphDestroy(g); return rc_mid; }
            MCHK(m, e2);
            const hipError_t e3 = hipGraphInstantiate(&m->mid_exec[T], g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            MCHK(m, e3);
        }
        if (m->mid_exec[T]) {
            MCHK(m, hipGraphLaunch(m->mid_exec[T], st));
            return decode_back(m, T, pcm, st);
        }
    }
    rc = decode_middle(m, T, st);
    if (rc) return rc;
    return decode_back(m, T, pcm, st);
}

static int slide_all(MimiDecoder* m, int T, hipStream_t st) {
    const MimiConfig& c = m->cfg;
    long Tj = 2L * T;
    MCHK(m, slide(m->rvq, T, st)); MCHK(m, slide(m->a0, Tj, st)); MCHK(m, slide(m->s0, Tj, st));
    for (int j = 0; j < c.n_stages; ++j) {
        Tj *= c.ratios[j];
        MCHK(m, slide(m->u[j], Tj, st)); MCHK(m, slide(m->xj[j], Tj, st));
    }
    return 0;
}

static long ceil_div(long a, long b) { return (a + b - 1) / b; }

static int transformer_pass(MimiDecoder* m, const MimiTrLayer* layers, long T2, int offset, float* last_out, hipStream_t st) {
    const MimiConfig& c = m->cfg;
    const int d = c.hidden;
    const int zero = 0;
    for (int l = 0; l < c.tr_layers; ++l) {
        const MimiTrLayer& L = layers[l];
        float* kc = m->kc + (long)l * m->cap_tokens * d;
        float* vc = m->vc + (long)l * m->cap_tokens * d;
        hipLaunchKernelGGL(k_layernorm, dim3((unsigned)T2), dim3(64), 0, st, m->tok, d, L.ln1_w, L.ln1_b, c.norm_eps, m->ln);
        const RopeOut ro = {m->q, kc, vc, m->w.rope_freqs, offset, d};
        MCHK(m, gemm(st, m->ln, d, T2, L.in_proj, nullptr, d, 3 * d, 1, 1, &zero, 0, 0, nullptr, nullptr, 0, nullptr, 0, 1, 0, 0, 0, &ro));
        hipLaunchKernelGGL(k_mimi_attn, dim3((unsigned)T2, c.tr_heads), dim3(64), 0, st, m->q, kc, vc, d, offset, c.tr_context, m->att);
        MCHK(m, gemm(st, m->att, d, T2, L.out_proj, nullptr, d, d, 1, 1, &zero, 0, 0, L.ls1, m->tok, d, m->tok, d));
        hipLaunchKernelGGL(k_layernorm, dim3((unsigned)T2), dim3(64), 0, st, m->tok, d, L.ln2_w, L.ln2_b, c.norm_eps, m->ln);
        MCHK(m, gemm(st, m->ln, d, T2, L.lin1, nullptr, d, c.tr_ffn, 1, 1, &zero, 0, 1, nullptr, nullptr, 0, m->ffn, c.tr_ffn));
        float* dst = (l + 1 < c.tr_layers) ? m->tok : last_out;
        MCHK(m, gemm(st, m->ffn, c.tr_ffn, T2, L.lin2, nullptr, c.tr_ffn, d, 1, 1, &zero, 0, 0, L.ls2, m->tok, d, dst, d));
        MCHK(m, hipGetLastError());
    }
    return 0;
}

static int encode_one(MimiDecoder* m, const float* wav, long n, int32_t* codes, long T, hipStream_t st) {
    const MimiConfig& c = m->cfg;
    const MimiWeights& w = m->w;
    const int d = c.hidden, S = c.n_stages;
    const int zero = 0;
    int shifts[MAX_TAPS];
    // SEANet encoder; stage j lives in the decoder's stage buffers of index S-1-j (same time scale and width)
    long L = n;
    int C = c.n_filters;
    {
        const long tot = L * C;
        hipLaunchKernelGGL(k_enc_conv_in, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, wav, L, c.kernel, C,
                           w.enc_conv_in_w, w.enc_conv_in_b, m->u[S - 1].row0());
    }
    for (int j = 0; j < S; ++j) {
        const int bi = S - 1 - j;
        const MimiConv &r1 = w.enc_res1[j], &r2 = w.enc_res2[j], &dn = w.enc_down[j];
        float* x = m->u[bi].row0();
        for (int k = 0; k < r1.taps; ++k) shifts[k] = k - (r1.taps - 1);
        MCHK(m, gemm(st, x, C, L, r1.w, r1.bias, r1.c_in, r1.c_out, r1.taps, 1, shifts, 1, 0, nullptr, nullptr, 0, m->r1[bi], r1.c_out,
                     1, 1, 0, (int)L));
        MCHK(m, gemm(st, m->r1[bi], r1.c_out, L, r2.w, r2.bias, r2.c_in, r2.c_out, 1, 1, &zero, 1, 0, nullptr, x, C, m->xj[bi].row0(), C,
                     1, 1, 0, (int)L));
        // ELU -> strided conv k = 2r: out[t] = sum_k W_k x[t*r + k - r]; zero padding on both sides
        const int r = dn.taps / 2;
        const long Lo = ceil_div(L, r);
        for (int k = 0; k < dn.taps; ++k) shifts[k] = k - r;
        float* out = (j + 1 < S) ? m->u[bi - 1].row0() : m->s0.row0();
        MCHK(m, gemm(st, m->xj[bi].row0(), C, Lo, dn.w, dn.bias, dn.c_in, dn.c_out, dn.taps, 1, shifts, 1, 0, nullptr, nullptr, 0, out,
                     dn.c_out, r, 1, 0, (int)L));
        L = Lo; C = dn.c_out;
    }
    const MimiConv& co = w.enc_conv_out;
    for (int k = 0; k < co.taps; ++k) shifts[k] = k - (co.taps - 1);
    MCHK(m, gemm(st, m->s0.row0(), C, L, co.w, co.bias, co.c_in, co.c_out, co.taps, 1, shifts, 1, 0, nullptr, nullptr, 0, m->tok, d,
                 1, 1, 0, (int)L));
    // encoder transformer (stateless) -> a0
    if (transformer_pass(m, w.enc_tr, L, 0, m->a0.row0(), st)) return -2;
    // stride-2 downsample, replicate padding, no bias -> rvq buffer [T][d]
    const int dshift[4] = {-2, -1, 0, 1};
    MCHK(m, gemm(st, m->a0.row0(), d, T, w.downsample, nullptr, d, d, 4, 1, dshift, 0, 0, nullptr, nullptr, 0, m->rvq.row0(), d,
                 2, 2, 0, (int)L));
    // split RVQ: semantic level on in_proj_first(z), acoustic levels on in_proj_rest(z)
    const int cd = c.codebook_dim;
    float* res_first = m->ln;                   // [T][cd]
    float* res_rest = m->q;                     // [T][cd]
    MCHK(m, gemm(st, m->rvq.row0(), d, T, w.in_proj_first, nullptr, d, cd, 1, 1, &zero, 0, 0, nullptr, nullptr, 0, res_first, cd));
    MCHK(m, gemm(st, m->rvq.row0(), d, T, w.in_proj_rest, nullptr, d, cd, 1, 1, &zero, 0, 0, nullptr, nullptr, 0, res_rest, cd));
    for (int k = 0; k < c.n_codebooks; ++k) {
        float* res = k < c.n_semantic ? res_first : res_rest;
        const float* book = w.codebooks + (long)k * c.codebook_size * cd;
        MCHK(m, gemm(st, res, cd, T, book, nullptr, cd, c.codebook_size, 1, 1, &zero, 0, 0, nullptr, nullptr, 0, m->ffn, c.codebook_size));
        hipLaunchKernelGGL(k_rvq_pick, dim3((unsigned)T), dim3(256), 0, st, m->ffn, c.codebook_size,
                           w.codebook_sqnorm + (long)k * c.codebook_size, book, cd, res, codes + (long)k * T, 1L);
        MCHK(m, hipGetLastError());
    }
    return 0;
}

extern "C" int mimi_encode(mimi_handle m, const float* wav, long n_samples, long stride_b, int B, int32_t* codes, void* stream) {
    if (!m || !wav || !codes || B < 1 || n_samples < 1) return mfail(m, "mimi_encode: bad argument");
    if (!m->w.has_encoder) return mfail(m, "mimi_encode: this codec was created without encoder weights");
    long hop = 2;
    for (int j = 0; j < m->cfg.n_stages; ++j) hop *= m->cfg.ratios[j];
    const long T = ceil_div(n_samples, hop);
    if (T > m->max_frames) return mfail(m, "mimi_encode: audio longer than hop * max_frames");
    hipStream_t st = (hipStream_t)stream;
    for (int b = 0; b < B; ++b) {
        int rc = encode_one(m, wav + (long)b * stride_b, n_samples, codes + (long)b * m->cfg.n_codebooks * T, T, st);
        if (rc) return rc;
    }
    return mimi_reset_stream(m, stream);      // the work buffers were reused: start any later stream from scratch
}

extern "C" int mimi_decode_strided(mimi_handle m, const int32_t* codes, int B, int T, long stride_b, long stride_k, long stride_t,
                                   void* pcm, int stateful, void* stream) {
    if (!m || !codes || !pcm || B < 1 || T < 1) return mfail(m, "mimi_decode: bad argument");
    if (T > m->max_frames) return mfail(m, "mimi_decode: T exceeds max_frames given to mimi_create");
    if (stateful && B != 1) return mfail(m, "mimi_decode: stateful streaming needs B == 1");
    hipStream_t st = (hipStream_t)stream;
    long hop = 2;
    for (int j = 0; j < m->cfg.n_stages; ++j) hop *= m->cfg.ratios[j];
    for (int b = 0; b < B; ++b) {
        if (!stateful) { int rc = mimi_reset_stream(m, stream); if (rc) return rc; }
        if (m->offset + 2L * T > m->cap_tokens) return mfail(m, "mimi_decode: stream longer than max_frames; call mimi_reset_stream");
        int rc = decode_one(m, codes + (long)b * stride_b, stride_k, stride_t, T, (float*)pcm + (long)b * hop * T, st, !stateful);
        if (rc) return rc;
        if (stateful) { rc = slide_all(m, T, st); if (rc) return rc; m->offset += 2 * T; }
    }
    return 0;
}

extern "C" int mimi_decode(mimi_handle m, const int32_t* codes, int B, int T, long stride_b, long stride_k, void* pcm,
                           int stateful, void* stream) {
    return mimi_decode_strided(m, codes, B, T, stride_b, stride_k, 1, pcm, stateful, stream);
}
