// Frame-level glue kernels: masked embedding sum, temperature/top-k sampler (+ embedding of
// the fed-back code), and the per-frame state advance.
#pragma once
#include "common.cuh"
#include <math.h>

// ---------------------------------------------------------------------------------------
// h[m] = sum over the 33 slots of mask * embedding  (sesameai/models.py:155-157,193-203):
// slot cb<32 -> audio_emb[tok + cb*audio_vocab], slot 32 -> text_emb[tok]; fp32 accumulate in
// slot order, one bf16 rounding.  Masked-out slots are skipped (the reference looks them up
// and multiplies by 0).  grid = M rows, block = 256, 8 columns per thread.
// ---------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void k_embed_sum(const int* tokens /*[M][33]*/, const uint8_t* mask /*[M][33]*/,
                                                   const bf16_t* text_emb, const bf16_t* audio_emb,
                                                   int audio_vocab, int text_vocab, int ncb, int d,
                                                   bf16_t* h /*[M][d]*/) {
    __shared__ const bf16_t* rows[64];             // row pointer per slot (nullptr = masked out)
    const int m = blockIdx.x;
    if (threadIdx.x <= ncb && threadIdx.x < 64) {
        const int s = threadIdx.x;
        int t = tokens[(long)m * (ncb + 1) + s];
        const bf16_t* row = nullptr;
        if (mask[(long)m * (ncb + 1) + s]) {
            if (s < ncb) { t = min(max(t, 0), audio_vocab - 1); row = audio_emb + ((long)s * audio_vocab + t) * d; }
            else         { t = min(max(t, 0), text_vocab - 1);  row = text_emb + (long)t * d; }
        }
        rows[s] = row;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d / 8; c += 256) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
        for (int s0 = 0; s0 <= ncb; s0 += 11) {      // 11 independent row loads in flight, summed in slot order
            uint4 v[11];
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                const bf16_t* row = (s0 + u <= ncb) ? rows[s0 + u] : nullptr;
                v[u] = row ? reinterpret_cast<const uint4*>(row)[c] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                acc[0] += lo2f(v[u].x); acc[1] += hi2f(v[u].x); acc[2] += lo2f(v[u].y); acc[3] += hi2f(v[u].y);
                acc[4] += lo2f(v[u].z); acc[5] += hi2f(v[u].z); acc[6] += lo2f(v[u].w); acc[7] += hi2f(v[u].w);
            }
        }
        uint4 o;
        o.x = pack_bf(acc[0], acc[1]); o.y = pack_bf(acc[2], acc[3]);
        o.z = pack_bf(acc[4], acc[5]); o.w = pack_bf(acc[6], acc[7]);
        reinterpret_cast<uint4*>(h + (long)m * d)[c] = o;
    }
}

// ---------------------------------------------------------------------------------------
// sample_topk (sesameai/models.py:72-87), one wave per sequence, everything in registers:
//   t = bf16(logit / T); drop t < kth-largest (ties kept); log_softmax; softmax;
//   argmax(p / Exp(1)) with the first index winning ties.  (Algorithm: see k_sample.)
// The bf16 rounding points are those of torch-CPU's reduced-precision kernels (verified by
// probe): the log-softmax keeps its exp-sum and its log in bf16 and subtracts
// in two bf16 steps; softmax rounds once.  topk == 1 is the deterministic greedy rule
// (lowest index among maxima).  The kth-largest value is found exactly with a two-pass
// radix select over the 16-bit order-preserving keys (256-bin LDS histograms).
// Tail: writes the code into frame[b][cb] and the embedding row of the FED code
// (forced[b][cb] if teacher forcing) to emb_out -- the next decoder input
// (models.py:163,178-180).
// ---------------------------------------------------------------------------------------
#define SAMPLE_MAX_ITERS 8   // supports V <= 8*512

struct SampleArgs {
    const bf16_t* logits;     // [B][ldl]
    int ldl, V;
    float temperature;
    int topk;
    const bf16_t* noise;      // optional [B][V] Exp(1) draws (bf16)
    const uint64_t* rng;      // device {seed, step}
    int codebook;
    const int* forced;        // optional [B][ncb]
    int ncb;
    int* frame;               // [B][ncb]
    const bf16_t* audio_emb;  // [ncb*audio_vocab][d]
    int audio_vocab, d;
    bf16_t* emb_out;          // row b at emb_out + b*emb_stride (elements); may be null
    long emb_stride;
    const bf16_t* xn_scale;   // optional: also write RMSNorm(row) * xn_scale (the next stack's first sa_norm, wide path)
    float xn_eps;
    bf16_t* xn_out;           // row b at xn_out + b*xn_stride
    long xn_stride;
    // optional: layer-0 q | k | v of the next decoder step, one table row per token of this codebook (csm_engine.hip,
    // build_qkv0_table): q -> q_out row b, k/v -> the decoder's layer-0 cache of sequence b at next_pos
    const bf16_t* qkv0;       // [audio_vocab][nq + 2 nkv]
    int nq, nkv, kv_heads, smax, hd, next_pos;
    bf16_t *q_out, *k0, *v0;
};

__device__ __forceinline__ uint32_t order_key(float t) {     // monotone map of a bf16-valued float
    uint32_t u = __float_as_uint(t) >> 16;
    if (u == 0x8000u) u = 0;                                   // -0 == +0
    return (u & 0x8000u) ? (~u & 0xffffu) : (u | 0x8000u);
}

// k-th largest of a wave-distributed key list (each lane holds NC keys, 0 = empty slot): the largest X with
// count(key >= X) >= k (0 if fewer than k keys), by bisection on the 16 key bits with ballots.  One wave runs this
// dependent chain at ~4 ns per instruction and a step is 3 NC + 2 instructions, so callers keep NC as small as they can.
template <int NC>
__device__ __forceinline__ uint32_t kth_largest_key(const uint32_t (&key)[NC], int k) {
    uint32_t prefix = 0;
#pragma unroll
    for (int bit = 15; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) cnt += __popcll(__ballot(key[c] >= cand));
        if (cnt >= k) prefix = cand;
    }
    return prefix;
}

// raw bf16 bits -> order-preserving 16-bit key (no float conversion)
__device__ __forceinline__ uint32_t raw_key(uint32_t bits) {
    if (bits == 0x8000u) bits = 0;
    return (bits & 0x8000u) ? (~bits & 0xffffu) : (bits | 0x8000u);
}

// LDS scratch of the sampler (all in the workgroup's LDS; typed so every access is a ds_ instruction -- the persistent
// depth-decoder kernel must not emit flat loads, which wait on vmcnt(0))
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef __attribute__((address_space(3))) int lds_i32_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;
struct SampleScratch {
    lds_f32_t* cand_t;      // [V] (up to SAMPLE_MAX_ITERS * 512)
    lds_i32_t* cand_i;      // [V]
    lds_u32_t* s_max;       // [256], 16-byte aligned
    lds_f32_t* cand_q;      // [256] Exp(1) draws of the first 256 candidates (may alias s_max: that is dead by then)
    lds_f32_t* s_bv;        // [4]
    lds_i32_t* s_bi;        // [4]
    lds_i32_t* s_n;         // [1]
    lds_i32_t* s_tok;       // [1]
    lds_i32_t* s_wtot;      // [4]
};

// q ~ Exp(1) of vocabulary index idx, rounded to bf16 (torch's exponential_ on a bf16 tensor): the caller's noise row, or
// Philox keyed by (seed, step) at counter (idx, sequence, codebook)
__device__ __forceinline__ float exp1_draw(int idx, const bf16_t* noise_row, uint64_t seed, uint64_t step, int b, int codebook) {
    if (noise_row) return bf2f(noise_row[idx]);
    const uint4 rnd = philox4x32(make_uint4((uint32_t)idx, (uint32_t)b, (uint32_t)codebook, (uint32_t)step),
                                 make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32)));
    const float u = ((float)rnd.x + 0.5f) * 2.3283064365386963e-10f;   // (0,1]
    const float q = round_bf(-logf(u));
    return q > 0.f ? q : 1e-30f;
}

// 4 waves (tid 0..255) sample ONE sequence.  Thread t owns logits [8t, 8t+8) (+ [2048+8t, ..) for V > 2048), handed
// in as w[i][0..3] (packed bf16 pairs; zero beyond the row).  `sync` is the 4-wave barrier (k_sample: __syncthreads;
// the persistent decoder: an LDS counter barrier).
//  1. per-thread max of the RAW bf16 keys -> LDS; the kth largest of the 128 thread-PAIR maxima is a lower bound L of
//     the kth-largest logit (k distinct elements are >= it);
//  2. every element with raw key >= L - margin is a candidate (t = bf16(l/T) is monotone in l, and at most 2*ceil(T)+2
//     neighbouring bf16 inputs can round to one output, hence the margin): appended, undivided, to an LDS list in index
//     order (~k..3k entries);
//  3. wave 0 takes the list two entries per lane, divides by the temperature, finds the exact kth-largest t by bisection
//     (ties kept), re-packs the ~k survivors one per lane and does log-softmax / softmax exactly as torch-CPU rounds
//     them in registers, while waves 1..3 draw the candidates' Exp(1) variates (Philox: ~150 instructions each, off
//     wave 0's critical path); then the race argmax(p / q) on wave 0.  (More than 128 candidates or more than 64
//     survivors -- many ties -- take a generic looped form of the same arithmetic.)
//  4. returns the sampled index (valid in every thread after the final sync).
template <int ITERS, class Sync>
__device__ __forceinline__ int sample_body(const uint32_t (&w)[ITERS][4], int V, float temperature, int topk, const bf16_t* noise_row,
                                           uint64_t seed, uint64_t step, int b, int codebook, const SampleScratch& sc, int tid, Sync sync) {
    const int lane = tid & 63, wave = tid >> 6;
    int best_idx = 0x7fffffff;
    float best = -INFINITY;
    if (topk <= 1) {
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i * 2048 + tid * 8 + j;
                const uint32_t bits = (j & 1) ? (w[i][j >> 1] >> 16) : (w[i][j >> 1] & 0xffffu);
                const float t = round_bf(__uint_as_float(bits << 16) / temperature);
                if (idx < V && t > best) { best = t; best_idx = idx; }
            }
    } else {
        const int k = min(topk, V);
        // order keys of this thread's elements, 0 = not an element (beyond V, NaN); kept in registers for all three passes
        uint32_t key[ITERS][8];
        uint32_t lmax = 0;
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i * 2048 + tid * 8 + j;
                const uint32_t bits = (j & 1) ? (w[i][j >> 1] >> 16) : (w[i][j >> 1] & 0xffffu);
                const bool nan = (bits & 0x7fffu) > 0x7f80u;
                key[i][j] = (idx < V && !nan) ? raw_key(bits) : 0u;
                lmax = max(lmax, key[i][j]);
            }
        sc.s_max[tid] = lmax;
        sync.mark(0);
        sync();
        uint32_t L = 0;
        if (k <= 256) {
            const u32x4_t mv = reinterpret_cast<lds_u32x4_t*>(sc.s_max)[lane];
            if (k <= 128) {      // maxima of thread PAIRS: still k distinct elements >= the kth largest of them, half the bisection work
                const uint32_t mk[2] = {max(mv.x, mv.y), max(mv.z, mv.w)};
                L = kth_largest_key<2>(mk, k);
            } else {
                const uint32_t mk[4] = {mv.x, mv.y, mv.z, mv.w};
                L = kth_largest_key<4>(mk, k);
            }
        }
        sync.mark(1);
        const uint32_t margin = 2u * (uint32_t)ceilf(fmaxf(temperature, 1.0f)) + 2u;
        const uint32_t Lm = L > margin ? L - margin : 1u;             // >= 1: key 0 is "not an element"
        // deterministic compaction in index order: per-thread count -> wave scan -> wave bases
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) cnt += key[i][j] >= Lm ? 1 : 0;
        // (cnt <= 8 * ITERS <= 64: 7 bits)
        const int incl = wave_excl_scan_small<7>(cnt) + cnt;
        if (lane == 63) sc.s_wtot[wave] = incl;
        sync();
        int o = incl - cnt;
        for (int x = 0; x < wave; ++x) o += sc.s_wtot[x];
        if (tid == 255) *sc.s_n = o + cnt;
        // the list holds the UNDIVIDED logit (as a float) and its index; wave 0 divides the ~k survivors, one or two per lane
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (key[i][j] >= Lm) {
                    const uint32_t bits = (j & 1) ? (w[i][j >> 1] >> 16) : (w[i][j >> 1] & 0xffffu);
                    sc.cand_t[o] = __uint_as_float(bits << 16);
                    sc.cand_i[o] = i * 2048 + tid * 8 + j;
                    ++o;
                }
        sync();
        sync.mark(2);
        const int n = *sc.s_n;
        if (wave == 0) {
            bool fast = false;
            if (n <= 128) {
                // ---- two list slots per lane in registers: t = bf16(logit / T), exact kth-largest t (ties kept) ------
                float tv[2]; int ix[2]; uint32_t ck[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = u * 64 + lane;
                    const bool live = c < n;
                    tv[u] = round_bf(sc.cand_t[live ? c : 0] / temperature);
                    ix[u] = sc.cand_i[live ? c : 0] | (c << 16);            // (V <= 16384: the index fits 14 bits)
                    ck[u] = live ? order_key(tv[u]) : 0u;
                }
                const uint32_t kth = n > k ? kth_largest_key<2>(ck, k) : 0u;
                const bool keep0 = ck[0] != 0u && ck[0] >= kth, keep1 = ck[1] != 0u && ck[1] >= kth;
                const unsigned long long m0 = __ballot(keep0), m1 = __ballot(keep1);
                const int kept0 = __popcll(m0), kept = kept0 + __popcll(m1);
                sync.mark(3);
                if (kept <= 64) {
                    // ---- the survivors, one per lane (list order kept): everything below stays in registers ----------
                    fast = true;
                    const int p0 = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
                    const int p1 = kept0 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
                    if (keep0) { sc.cand_t[128 + p0] = tv[0]; sc.cand_i[128 + p0] = ix[0]; }      // (slots 0..127 are still being read by waves 1..3)
                    if (keep1) { sc.cand_t[128 + p1] = tv[1]; sc.cand_i[128 + p1] = ix[1]; }
                    const bool live = lane < kept;
                    const float v = sc.cand_t[128 + (live ? lane : 0)];
                    const int pk = sc.cand_i[128 + (live ? lane : 0)];
                    // log_softmax / softmax with torch-CPU's bf16 rounding points
                    const float mx = wave_max(live ? v : -INFINITY);
                    const float sum = wave_sum(live ? expf(v - mx) : 0.f);
                    const float logsum = round_bf(logf(round_bf(sum)));
                    const float mx2 = round_bf(0.f - logsum);      // log-prob of the max element
                    const float e3 = live ? expf(round_bf(round_bf(v - mx) - logsum) - mx2) : 0.f;
                    const float s2 = wave_sum(e3);
                    sync.mark(4);
                    sync();                                    // the Exp(1) draws of waves 1..3 are in cand_q
                    // argmax(p / q), q ~ Exp(1); first index wins ties
                    const float p = round_bf(e3 / s2);
                    if (live && p > 0.f) { best = round_bf(p / sc.cand_q[pk >> 16]); best_idx = pk & 0xffff; }
                }
            }
            if (!fast) {
                // ---- generic form (more than 128 candidates, or more than 64 survivors through ties) -------------------
                for (int c = lane; c < n; c += 64) sc.cand_t[c] = round_bf(sc.cand_t[c] / temperature);
                uint32_t kth = 0;
                if (n > k) {
                    for (int bit = 15; bit >= 0; --bit) {
                        const uint32_t cnd = kth | (1u << bit);
                        int cnt2 = 0;
                        for (int c = lane; c < ((n + 63) & ~63); c += 64)
                            cnt2 += __popcll(__ballot(c < n && order_key(sc.cand_t[c]) >= cnd));
                        if (cnt2 >= k) kth = cnd;
                    }
                }
                float mx = -INFINITY;
                for (int c = lane; c < n; c += 64) {
                    const float v = sc.cand_t[c];
                    if (order_key(v) >= kth) mx = fmaxf(mx, v);
                }
                mx = wave_max(mx);
                float sum = 0.f;
                for (int c = lane; c < n; c += 64) {
                    const float v = sc.cand_t[c];
                    if (order_key(v) >= kth) sum += expf(v - mx);
                }
                sum = wave_sum(sum);
                const float logsum = round_bf(logf(round_bf(sum)));
                const float mx2 = round_bf(0.f - logsum);
                float s2 = 0.f;
                for (int c = lane; c < n; c += 64) {
                    const float v = sc.cand_t[c];
                    if (order_key(v) >= kth) s2 += expf(round_bf(round_bf(v - mx) - logsum) - mx2);
                }
                s2 = wave_sum(s2);
                sync();
                for (int c = lane; c < n; c += 64) {
                    const float v = sc.cand_t[c];
                    if (order_key(v) < kth) continue;
                    const int idx = sc.cand_i[c];
                    const float p = round_bf(expf(round_bf(round_bf(v - mx) - logsum) - mx2) / s2);
                    if (!(p > 0.f)) continue;
                    const float q = c < 256 ? sc.cand_q[c] : exp1_draw(idx, noise_row, seed, step, b, codebook);
                    const float r = round_bf(p / q);
                    if (r > best || (r == best && idx < best_idx)) { best = r; best_idx = idx; }
                }
            }
            sync.mark(5);
            const float wb = wave_max(best);
            best_idx = wave_min_i(best == wb ? best_idx : 0x7fffffff);
            if (lane == 0) *sc.s_tok = best_idx == 0x7fffffff ? 0 : best_idx;
        } else {
            // waves 1..3 meanwhile: the Exp(1) draw of every listed candidate (a pure function of its index)
            for (int c = (wave - 1) * 64 + lane; c < min(n, 256); c += 192)
                sc.cand_q[c] = exp1_draw(sc.cand_i[c], noise_row, seed, step, b, codebook);
            sync();
        }
        sync();
        return *sc.s_tok;
    }
    // greedy: argmax over the wave (lowest index on ties), then over the 4 waves
    {
        const float wb = wave_max(best);
        best_idx = wave_min_i(best == wb ? best_idx : 0x7fffffff);
        best = wb;
    }
    if (lane == 0) { sc.s_bv[wave] = best; sc.s_bi[wave] = best_idx; }
    sync();
    if (tid == 0) {
        float bv = sc.s_bv[0]; int bi = sc.s_bi[0];
        for (int x = 1; x < 4; ++x)
            if (sc.s_bv[x] > bv || (sc.s_bv[x] == bv && sc.s_bi[x] < bi)) { bv = sc.s_bv[x]; bi = sc.s_bi[x]; }
        if (bi == 0x7fffffff) bi = 0;
        *sc.s_tok = bi;
    }
    sync();
    return *sc.s_tok;
}

struct SyncThreads {
    __device__ __forceinline__ void operator()() const { __syncthreads(); }
    __device__ __forceinline__ void mark(int) const {}      // (the persistent decoder's barrier records a debug timeline here)
};

// One block of 4 waves per sequence (the standalone launch of the chain path).
template <int ITERS>
__global__ __launch_bounds__(256) void k_sample(const SampleArgs a) {
    __shared__ float cand_t[SAMPLE_MAX_ITERS * 512];
    __shared__ int cand_i[SAMPLE_MAX_ITERS * 512];
    __shared__ __attribute__((aligned(16))) uint32_t s_max[256];
    __shared__ float s_bv[4];
    __shared__ int s_bi[4];
    __shared__ int s_n, s_tok, s_wtot[4];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bf16_t* lg = a.logits + (long)b * a.ldl;
    // issued now, consumed by wave 0 three barriers later (a dependent load there would sit on the critical path)
    const uint64_t seed = a.rng ? a.rng[0] : 0, step = a.rng ? a.rng[1] : 0;
    uint32_t w[ITERS][4];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const bool in = (i * 2048 + tid * 8) < a.ldl;
        const uint4 v = in ? reinterpret_cast<const uint4*>(lg + i * 2048)[tid] : make_uint4(0, 0, 0, 0);
        w[i][0] = v.x; w[i][1] = v.y; w[i][2] = v.z; w[i][3] = v.w;
    }
    SampleScratch sc;
    sc.cand_t = (lds_f32_t*)cand_t; sc.cand_i = (lds_i32_t*)cand_i; sc.s_max = (lds_u32_t*)s_max; sc.cand_q = (lds_f32_t*)s_max; sc.s_bv = (lds_f32_t*)s_bv;
    sc.s_bi = (lds_i32_t*)s_bi; sc.s_n = (lds_i32_t*)&s_n; sc.s_tok = (lds_i32_t*)&s_tok; sc.s_wtot = (lds_i32_t*)s_wtot;
    const int tok = sample_body<ITERS>(w, a.V, a.temperature, a.topk, a.noise ? a.noise + (long)b * a.V : nullptr, seed, step, b,
                                       a.codebook, sc, tid, SyncThreads());
    if (tid == 0) a.frame[(long)b * a.ncb + a.codebook] = tok;
    if (a.emb_out) {
        int fed = a.forced ? a.forced[(long)b * a.ncb + a.codebook] : tok;
        fed = min(max(fed, 0), a.audio_vocab - 1);
        const uint4* src = reinterpret_cast<const uint4*>(a.audio_emb + ((long)a.codebook * a.audio_vocab + fed) * a.d);
        uint4* dst = reinterpret_cast<uint4*>(a.emb_out + (long)b * a.emb_stride);
        for (int c = tid; c < a.d / 8; c += 256) dst[c] = src[c];
        if (a.xn_out && wave == 0)
            rmsnorm_row_wave(reinterpret_cast<const bf16_t*>(src), a.d, a.xn_scale, a.xn_eps, a.xn_out + (long)b * a.xn_stride, lane);
        if (a.qkv0) {
            const int ld = a.nq + 2 * a.nkv, per_head = a.hd / 8;
            const uint4* row = reinterpret_cast<const uint4*>(a.qkv0 + (long)fed * ld);
            for (int c = tid; c < ld / 8; c += 256) {
                const uint4 v = row[c];
                if (c < a.nq / 8) reinterpret_cast<uint4*>(a.q_out + (long)b * a.nq)[c] = v;
                else {
                    const int ck = c - a.nq / 8, isv = ck >= a.nkv / 8, cc = isv ? ck - a.nkv / 8 : ck;
                    bf16_t* dstc = (isv ? a.v0 : a.k0) + (((long)b * a.kv_heads + cc / per_head) * a.smax + a.next_pos) * a.hd;
                    reinterpret_cast<uint4*>(dstc)[cc % per_head] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// end-of-frame bookkeeping (generator.py:285-294): history append, EOS flag, next-step inputs
// [frame, 0] with mask [1 x ncb, 0], pos += 1, rng step += 1.   grid = 1, block = 64.
// ---------------------------------------------------------------------------------------
struct AdvanceArgs {
    const int* frame;     // [B][ncb]
    int B, ncb, bstride;  // history is a ring [max_frames][bstride][ncb]: global frame n lives in row n % max_frames
    int* history;
    int* n_frames;        // device counter
    int max_frames;
    int* eos_at;          // [B], -1 until the first all-zero frame
    int* cur_tokens;      // [B][ncb+1]
    uint8_t* cur_mask;    // [B][ncb+1]
    int* cur_pos;         // [B]
    uint64_t* rng;        // {seed, step}
    int* out_frame;       // optional user copy [B][ncb]
    const int* fed;       // optional [B][ncb]: codes fed back instead of frame (teacher forcing)
    int pos_inc;          // 1 after a backbone step consumed cur_pos, 0 after a prefill
    int max_seq;          // backbone positions are [0, max_seq)
    int* overflow;        // device flag: a step ran at a position >= max_seq (csm_read_frames -> CSM_E_TOO_LONG)
    const uint32_t *err0, *err1;   // optional give-up words of the all-CU launches: non-zero -> this frame's codes are invalid, recorded as -1
    int* fresh;           // optional [B]: 1 = this frame is FRAME 0 of an utterance whose prompt was prefilled beside the frame loop
                          // (csm_refill_*): its backbone row of this step was a placeholder, so the position stays, the EOS word restarts;
                          // 2 = parked (the prompt is still running): position held, never tested against max_seq
};

static __global__ __launch_bounds__(256) void k_advance(const AdvanceArgs a) {
    __shared__ int nz[256];                         // per-sequence count of non-zero codes (B <= 256)
    const int n = *a.n_frames;
    const bool bad = (a.err0 != nullptr && *a.err0 != 0u) || (a.err1 != nullptr && *a.err1 != 0u);
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) nz[b] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < a.B * a.ncb; i += blockDim.x) {
        const int b = i / a.ncb, c = i % a.ncb;
        const int v = bad ? -1 : a.frame[i];
        if (bad) const_cast<int*>(a.frame)[i] = -1;          // csm_copy_frame / the next reader sees it too
        if (v != 0) atomicAdd(&nz[b], 1);
        a.history[((long)(n % a.max_frames) * a.bstride + b) * a.ncb + c] = v;
        a.cur_tokens[b * (a.ncb + 1) + c] = a.fed ? a.fed[i] : (v < 0 ? 0 : v);
        a.cur_mask[b * (a.ncb + 1) + c] = 1;
        if (a.out_frame) a.out_frame[i] = v;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
        a.cur_tokens[b * (a.ncb + 1) + a.ncb] = 0;
        a.cur_mask[b * (a.ncb + 1) + a.ncb] = 0;
        const int flag = a.fresh != nullptr ? a.fresh[b] : 0;             // 1 = this frame is the slot's frame 0, 2 = parked (placeholder row)
        if (flag == 1) { a.eos_at[b] = -1; a.fresh[b] = 0; }
        if (nz[b] == 0 && a.eos_at[b] < 0) a.eos_at[b] = n;
        if (flag == 0) {                                                  // a parked / joining slot keeps its position: its row was a placeholder
            if (a.pos_inc && a.cur_pos[b] >= a.max_seq) *a.overflow = 1;  // the step that just ran used this position
            a.cur_pos[b] += a.pos_inc;
        }
    }
    if (threadIdx.x == 0) { *a.n_frames = n + 1; a.rng[1] += 1; }
}

static inline hipError_t launch_sample(const SampleArgs& s, int B, hipStream_t st) {
    switch ((s.V + 2047) / 2048) {
        case 1: hipLaunchKernelGGL(k_sample<1>, dim3(B), dim3(256), 0, st, s); break;
        case 2: hipLaunchKernelGGL(k_sample<2>, dim3(B), dim3(256), 0, st, s); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
