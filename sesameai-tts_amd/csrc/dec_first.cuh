// The FIRST depth-decoder step of a batch-1 frame (codebook 1: positions 0 and 1 of the decoder, reference sesameai/models.py:165-176
// with curr_h = [last_h, embedding of c0]) as ONE launch of 256 workgroups -- dec_persist.cuh's layout, hand-offs and arithmetic with TWO
// rows per phase and no sampler.
//
// Why: codebooks 2..31 run in k_dec_persist at ~70 us per step; the step in front of them ran as 16 launches of the M = 2 GEMV chain
// (q|k|v 5.3, attention + o-proj 9.0, gate/up 10.5, down 7.5 us per layer) + the head GEMV: 134 us -- 5 % of a 2.75 ms frame.  As two
// 1-row steps inside k_dec_persist it would cost two steps' worth of hand-offs (62 + 70 us: nothing gained); as a 2-row step both rows
// share every hand-off, and a launch of its own keeps the two-row code out of k_dec_persist, whose 30 steps stream through the
// instruction caches.  This kernel runs once per frame: nothing in it is trickled across steps.
//
// What it leaves behind is what the chain's cb = 1 step left: the decoder K / V caches of positions 0, 1 for every layer (global), and
// the logits of codebook 1 in the engine's logits row (written by the head units themselves: no logits all-gather -- nothing in this launch
// samples); k_sample then picks c1 and gathers the next step's table rows as before.
//
// Per layer, with r = 0, 1 the two rows (positions):  x[r] = sa_norm(h[r]) -> q|k|v units (RoPE at position r) -> all-gather ->
// attention of row r over keys 0..r (replicated on every workgroup, eight heads on eight waves) -> o-proj units + residual -> all-gather
// -> mlp_norm -> gate/up on the matrix cores (row r is column r of the B operand: the second row costs no instruction) -> SwiGLU ->
// split down projection -> 2 x 256 partials per output row, summed by the row's owner in dec_persist.cuh's fixed order.
// Arithmetic: dec_persist.cuh's functions; the values of row 1 are what a k_dec_persist step at position 1 would compute.
#pragma once
#include "dec_persist.cuh"

struct DecFirstArgs {
    const bf16_t* wsm;                // as DecPersistArgs
    const bf16_t* norms;
    const uint4* w2s;
    const uint4* w13p;
    const bf16_t* dec_norm;
    const bf16_t* head_t;             // [V][1024]: the head of codebook 1
    const bf16_t* rope;               // [max_seq][64][2]
    const bf16_t* hdec;               // [2][1024]: decoder inputs of positions 0 (projection(last_h)) and 1 (projected embedding of c0)
    bf16_t *kc, *vc;                  // decoder caches [L][max_batch][2][32][128]: rows of positions 0, 1 of sequence 0 are WRITTEN
    long kv_layer_stride;
    int V;
    bf16_t* logits;                   // [>= V + 1] bf16: the logits of codebook 1
    dp_u64 *gQ, *gH1, *gH2, *gP;      // granule slots: 8 x 2*768, 8 x 2*512, 8 x 2*512, 2 x 256 x 1024
    uint32_t* err;
    uint32_t* epoch;
    float eps;
    int trickle_sleep, poll_sleep;
};

// LDS image (dynamic shared memory; byte offsets)
#define DF_OFF_K 0                                   // [4][2][32][128] bf16 (positions 0, 1 live; the rest stays zero: dead keys add +0.0)
#define DF_OFF_V 65536
#define DF_OFF_XA 131072                             // [2 rows][1024] bf16
#define DF_OFF_XC (DF_OFF_XA + 4096)
#define DF_OFF_QB (DF_OFF_XC + 4096)
#define DF_OFF_ATT (DF_OFF_QB + 4096)
#define DF_OFF_PS (DF_OFF_ATT + 4096)                // attention P rows: 8 waves x 32 floats
#define DF_OFF_MISC (DF_OFF_PS + 1024)
#define DF_LDS_BYTES (DF_OFF_MISC + 1024)
// misc words
#define DF_M_HL 0        // [2][16] words: this CU's 32 h values of each row
#define DF_M_H0 32       // [2][2]: residual rows 4cu..4cu+3 entering the layer
#define DF_M_H1 36       // [2][2]: after the o-projection
#define DF_M_FXA 40      // flags
#define DF_M_FQ 41
#define DF_M_FXC 42
#define DF_M_ATTN 43     // counters
#define DF_M_CD 44
#define DF_M_ABORT 45
#define DF_M_TILE 64     // [2 rows][4 tiles][16] floats
enum { DF_E_XA = 0, DF_E_Q = 1, DF_E_H1 = 2, DF_E_P = 3, DF_E_H2 = 4 };
__device__ __forceinline__ uint32_t df_tag(uint32_t base, int l, int e) { return base + 1u + (uint32_t)(l * 5 + e); }
#define DF_EPOCH_STEP 32u

// dp_attention_head for at most FOUR keys (positions 0..3; here nk <= 2): that function walks all 32 key slots of the head, 4 per group, and dead
// keys contribute exact zeros (score -inf -> p = 0, "+ 0.0" leaves every running sum's bits unchanged) -- so walking only the first group of four
// gives the same bits with an eighth of the LDS reads and DPP chains (and only the first KB of each head's K / V image has to be zero).
__device__ __forceinline__ void df_attention_head4(const dp_lu4* qb, const dp_lu4* kt, const dp_lu32* vt, dp_lf32* myps, dp_lu32* att, int h, int nk,
                                                   float ascale, int lane) {
    const int grp = lane >> 4, sub = lane & 15;
    const uint4 qa = dp_ldq(qb + h * 16 + sub);
    const uint4 kv = dp_ldq(kt + lane);
    const bool live = grp < nk;
    const float d0 = row16_sum(dot8(qa, kv, 0.f)) * ascale;
    float s0 = live ? d0 : -INFINITY;
    const float mx0 = wave_max(s0);
    s0 = (s0 == -INFINITY) ? 0.f : __expf(s0 - mx0);
    if (sub == 0) myps[grp] = s0;
    const float l0 = wave_sum(s0) * (1.0f / 16.0f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const u32x4_t p4 = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(myps);
    const float pw[4] = {__uint_as_float(p4.x), __uint_as_float(p4.y), __uint_as_float(p4.z), __uint_as_float(p4.w)};
    uint32_t vr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vr[u] = vt[u * 64 + lane];
    float o00 = 0.f, o01 = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) { o00 += pw[u] * lo2f(vr[u]); o01 += pw[u] * hi2f(vr[u]); }
    const float i0 = 1.0f / l0;
    att[h * 64 + lane] = pack_bf(o00 * i0, o01 * i0);
}

// heads `wave` of layer l for both rows (row r attends to keys 0..r), then one arrival on the attention counter
__device__ __forceinline__ void df_attention_wave(char* lds, int wave, int l, int lane) {
    dp_lu32* misc = (dp_lu32*)(lds + DF_OFF_MISC);
    const int kvh = wave >> 2;
#pragma unroll
    for (int r = 0; r < 2; ++r)
        df_attention_head4((const dp_lu4*)(lds + DF_OFF_QB + r * 2048), (const dp_lu4*)(lds + DF_OFF_K + ((l * 2 + kvh) * 32) * 256),
                           (const dp_lu32*)(lds + DF_OFF_V + ((l * 2 + kvh) * 32) * 256), (dp_lf32*)(lds + DF_OFF_PS) + wave * 32,
                           (dp_lu32*)(lds + DF_OFF_ATT + r * 2048), wave, r + 1, 0.08838834764831845f, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(misc + DF_M_ATTN, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// compute wave: dec_persist.cuh's roles (X = waves 0, 1; A = 2, 3, 4: q|k|v unit; B = 5, 6: o-proj unit), its load schedule (a layer's MLP
// weights in thirds behind the layer's three waits, the small-op rows of the next layer behind the second / third), one step, two rows
template <bool HAS_TILE, int NBK>
__device__ __forceinline__ void df_compute_wave(const DecFirstArgs& a, char* lds, const int wave, const unsigned lane, const int cu, const uint32_t base,
                                                const uint32_t ropev) {
    constexpr bool HAS_B = !HAS_TILE;
    const bool is_x = HAS_TILE && wave < 2, is_a = HAS_TILE ? !is_x : wave == 4, is_b = HAS_B && !is_a;
    constexpr int NT = HAS_TILE ? 32 : 0, NCD = NT + NBK * 4, N1 = NCD / 3;
    dp_lu32* misc = (dp_lu32*)(lds + DF_OFF_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DF_M_ABORT);
    const int ts = a.trickle_sleep & 63;
    const int unit = is_a ? cu * 3 + (wave - 2) : cu * 2 + (wave - 5);
    auto my_block = [&](int b) { return wave < 4 ? wave + 4 * b : min((wave + 4) + 3 * b, wave == 6 ? 13 : 15); };
    const int hunit = is_x ? -1 : (wave < 6 ? cu * 4 + (wave - 2) : (cu < 2 ? 1024 + cu : -1));
    const int hrow0 = hunit < 0 ? 0 : 2 * hunit, hrow1 = hunit < 0 ? 0 : min(2 * hunit + 1, a.V - 1);
    uint4 wsa[2][2], wsb[HAS_B ? 2 : 1][2];
    uint4 wt[HAS_TILE ? 32 : 1];
    uint4 wd[NBK][4];
    auto ws_row = [&](int slot, int k) -> const bf16_t* {
        if (slot < DP_NL && !is_x) {
            if (is_a) return a.wsm + ((long)slot * DP_WSM_ROWS + 2 * unit + (k >> 1)) * DP_D;
            return a.wsm + ((long)slot * DP_WSM_ROWS + DP_NQKV + 2 * unit + (k >> 1)) * DP_D;
        }
        return a.head_t + (long)((k >> 1) ? hrow1 : hrow0) * DP_D;
    };
    auto load_wsa = [&](int slot, int k) { wsa[k >> 1][k & 1] = reinterpret_cast<const uint4*>(ws_row(slot, k))[(k & 1) * 64 + lane]; };
    auto load_wsb = [&](int slot, int k) { wsb[HAS_B ? k >> 1 : 0][k & 1] = reinterpret_cast<const uint4*>(ws_row(slot, k))[(k & 1) * 64 + lane]; };
    auto load_cd = [&](int l, int k) {
        if (k < NT) wt[HAS_TILE ? k : 0] = a.w13p[(long)l * DP_W13P_U4 + (((long)cu * 4 + wave) * 32 + k) * 64 + lane];
        else {
            const int kk = k - NT;
            wd[kk >> 2][kk & 3] = a.w2s[(long)l * DP_W2S_U4 + ((long)cu * 4 + (kk & 3)) * 1024 + my_block(kk >> 2) * 64 + lane];
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) load_wsa(0, k);
    if (HAS_B) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_wsb(0, k);
    }
#pragma unroll
    for (int k = 0; k < NCD; ++k) load_cd(0, k);

    for (int l = 0; l < DP_NL; ++l) {
        {   // -- x of the q|k|v units is ready: A waves run their unit on both rows
            if (!dp_wait<N1, false>((dp_lvu32*)(misc + DF_M_FXA), df_tag(base, l, DF_E_XA), ab, a.err, 0xA10u, lane, ts, [&](int k) { load_cd(l, k); })) return;
            if (is_a) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const dp_lu4* xs = (const dp_lu4*)(lds + DF_OFF_XA + r * 2048);
                    const uint4 x0 = dp_ldq(xs + lane), x1 = dp_ldq(xs + 64 + lane);
                    float a0 = dot8(wsa[0][0], x0, 0.f); a0 = dot8(wsa[0][1], x1, a0);
                    float a1 = dot8(wsa[1][0], x0, 0.f); a1 = dot8(wsa[1][1], x1, a1);
                    a0 = wave_sum(a0); a1 = wave_sum(a1);
                    const int row = 2 * unit;
                    const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)ropev, r);          // (cos, sin) of this unit's pair at position r
                    const uint32_t outw = dp_rope_pair(a0, a1, cs, row < 1280);
                    if (lane < DP_NREP) dp_gran_store(a.gQ + lane * 1536 + r * 768 + unit, df_tag(base, l, DF_E_Q), outw);
                }
            }
        }
        {   // -- attention: head `wave`, both rows (the gather wave takes head 7)
            if (!dp_wait<4 + N1, false>((dp_lvu32*)(misc + DF_M_FQ), df_tag(base, l, DF_E_Q), ab, a.err, 0xA20u, lane, ts, [&](int k) {
                    if (k < 4) load_wsa(l + 1 < DP_NL ? l + 1 : DP_NL, k); else load_cd(l, N1 + k - 4);
                })) return;
            df_attention_wave(lds, wave, l, lane);
        }
        if (HAS_B && is_b) {
            // -- o-projection unit + residual, both rows, once the eight attention waves are done
            if (!dp_wait<1, true>((dp_lvu32*)(misc + DF_M_ATTN), 8u * (uint32_t)(l + 1), ab, a.err, 0xA30u, lane, ts, [&](int) {})) return;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const dp_lu4* xs = (const dp_lu4*)(lds + DF_OFF_ATT + r * 2048);
                const uint4 x0 = dp_ldq(xs + lane), x1 = dp_ldq(xs + 64 + lane);
                float a0 = dot8(wsb[0][0], x0, 0.f); a0 = dot8(wsb[0][1], x1, a0);
                float a1 = dot8(wsb[HAS_B ? 1 : 0][0], x0, 0.f); a1 = dot8(wsb[HAS_B ? 1 : 0][1], x1, a1);
                a0 = wave_sum(a0); a1 = wave_sum(a1);
                const uint32_t h0w = *(dp_lvu32*)(misc + DF_M_H0 + 2 * r + (wave - 5));
                const uint32_t outw = dp_resid_pair(a0, a1, h0w);
                if (lane == 0) misc[DF_M_H1 + 2 * r + (wave - 5)] = outw;
                if (lane < DP_NREP) dp_gran_store(a.gH1 + lane * 1024 + r * 512 + unit, df_tag(base, l, DF_E_H1), outw);
            }
        }
        {   // -- the MLP: my (gate, up) pairs of both rows (row r = column r of the B operand) -> h values -> my row blocks of the split down projection
            constexpr int NB4 = HAS_B ? 4 : 0, N4 = NB4 + NCD - 2 * N1;
            if (!dp_wait<N4, false>((dp_lvu32*)(misc + DF_M_FXC), df_tag(base, l, DF_E_H1), ab, a.err, 0xA40u, lane, ts, [&](int k) {
                    if (k < NB4) load_wsb(l + 1 < DP_NL ? l + 1 : DP_NL, k); else load_cd(l, 2 * N1 + k - NB4);
                })) return;
            if (HAS_TILE) {
                const int col = lane & 15;
                const dp_lu4* xq = (const dp_lu4*)(lds + DF_OFF_XC + (col == 1 ? 2048 : 0)) + (lane >> 4);
                dp_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                uint4 xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xa[u] = dp_ldq(xq + 4 * u);
#pragma unroll
                for (int tb = 0; tb < 8; tb += 2) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) xb[u] = dp_ldq(xq + 4 * (4 * (tb + 1) + u));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dp_bf16x8, wt[HAS_TILE ? 4 * tb + u : 0]), __builtin_bit_cast(dp_bf16x8, xa[u]), acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (tb + 2 < 8) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) xa[u] = dp_ldq(xq + 4 * (4 * (tb + 2) + u));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dp_bf16x8, wt[HAS_TILE ? 4 * (tb + 1) + u : 0]), __builtin_bit_cast(dp_bf16x8, xb[u]), acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // column c of the result is row c's: lane l holds tile rows 4 (l >> 4) .. + 3 of column l & 15
                if (col < 2) {
                    dp_lf32* tile = (dp_lf32*)(misc + DF_M_TILE) + col * 64 + wave * 16;
                    tile[(lane >> 4) * 4 + 0] = acc[0]; tile[(lane >> 4) * 4 + 1] = acc[1];
                    tile[(lane >> 4) * 4 + 2] = acc[2]; tile[(lane >> 4) * 4 + 3] = acc[3];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane < 16) {
                    const int r = lane >> 3, i = lane & 7;
                    const dp_lf32* tile = (const dp_lf32*)(misc + DF_M_TILE) + r * 64 + wave * 16;
                    const uint32_t hv = dp_swiglu(tile[i], tile[8 + i]);
                    ((dp_lu16*)(misc + DF_M_HL + 16 * r))[wave * 8 + i] = (unsigned short)hv;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(misc + DF_M_CD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            {
                const uint32_t want = 7u * (uint32_t)(l + 1);
                const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
                for (uint32_t spins = 1; (int32_t)(*(dp_lvu32*)(misc + DF_M_CD) - want) < 0; ++spins)
                    if ((spins & 255u) == 0 && dp_give_up(t0, ab, a.err, 0xA50u, lane)) return;
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                uint4 h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) h[q] = dp_ldq((const dp_lu4*)(misc + DF_M_HL + 16 * r) + q);
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    const int n = my_block(b) * 64 + lane;
                    const float p = dp_down_partial(wd[b], h);
                    dp_gran_store(a.gP + (long)r * (256L * 1024) + ((long)(n >> 2) * 256 + cu) * 4 + (n & 3), df_tag(base, l, DF_E_P), __float_as_uint(p));
                }
            }
        }
    }
    // ---- the head of codebook 1 on row 1: waves 2..6 hold 2 logit rows each (loaded as slot DP_NL during the last layer)
    if (!dp_wait<1, false>((dp_lvu32*)(misc + DF_M_FXA), df_tag(base, DP_NL, DF_E_XA), ab, a.err, 0xA60u, lane, ts, [&](int) {})) return;
    if (hunit >= 0) {
        const dp_lu4* xs = (const dp_lu4*)(lds + DF_OFF_XA + 2048);
        const uint4 x0 = dp_ldq(xs + lane), x1 = dp_ldq(xs + 64 + lane);
        uint4 r00 = wsa[0][0], r01 = wsa[0][1], r10 = wsa[1][0], r11 = wsa[1][1];
        if (HAS_B && is_b) { r00 = wsb[0][0]; r01 = wsb[0][1]; r10 = wsb[HAS_B ? 1 : 0][0]; r11 = wsb[HAS_B ? 1 : 0][1]; }
        float a0 = dot8(r00, x0, 0.f); a0 = dot8(r01, x1, a0);
        float a1 = dot8(r10, x0, 0.f); a1 = dot8(r11, x1, a1);
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        // straight into the engine's logits row: the launch boundary in front of k_sample is the hand-off (k_dec_persist all-gathers its logits
        // because it samples in the launch); the odd vocabulary's last unit carries one logit, its second half reads as 0
        if (lane == 0) reinterpret_cast<uint32_t*>(a.logits)[hunit] = (2 * hunit + 1 < a.V) ? pack_bf(a0, a1) : (pack_bf(a0, a1) & 0xffffu);
    }
}

#ifdef CSM_DEC_FIRST_HERE
static __global__ __launch_bounds__(512) void k_dec_first(const DecFirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    const unsigned lane = threadIdx.x & 63;
    dp_lu32* misc = (dp_lu32*)(lds + DF_OFF_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DF_M_ABORT);
    uint32_t ropev = 0;
    if (wave >= 2 && wave <= 4 && lane < 2) {                     // (cos, sin) of the pair a q|k|v wave rotates, positions 0 and 1
        const int row = 2 * (cu * 3 + (wave - 2));
        const int e = (row < 1024 ? row : row - 1024) % DP_HD;
        ropev = reinterpret_cast<const uint32_t*>(a.rope)[lane * (DP_HD / 2) + e / 2];
    }
    for (int i = threadIdx.x; i < 256; i += 512) misc[i] = 0;
    // the first four key slots (1 KB) of each (layer, KV head) image of K and of V: positions 2, 3 stay zero (finite: a dead key's exact-zero
    // probability times it adds +0.0); the x / q / attention buffers are written before they are read
    for (int i = threadIdx.x; i < 16 * 64; i += 512) dp_stq((dp_lu4*)(lds + (i >> 6 & 1 ? DF_OFF_V : DF_OFF_K) + (i >> 7) * 8192) + (i & 63), make_uint4(0, 0, 0, 0));
    __syncthreads();
    const uint32_t base = dp_sload32(a.epoch);
    if (wave == 7) {
        // ------------------------------------------------------------------------------------------------ gather wave
        __builtin_amdgcn_s_setprio(2);
        const int rep = cu % DP_NREP, ln = (int)lane;
        const dp_u64 *rgQ = a.gQ + rep * 1536, *rgH1 = a.gH1 + rep * 1024, *rgH2 = a.gH2 + rep * 1024;
        const dp_u64* rgP = a.gP + (long)cu * 1024;
        for (int l = 0; l < DP_NL; ++l) {
            {   // rows entering the layer -> sa_norm -> xA (layer 0: the decoder's input rows from memory)
                const uint4 g0 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l) * DP_D)[ln], g1 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l) * DP_D)[64 + ln];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    dp_lu32* xw = (dp_lu32*)(lds + DF_OFF_XA + r * 2048);
                    if (l == 0) {
                        const uint4* src = reinterpret_cast<const uint4*>(a.hdec + (long)r * DP_D);
                        const uint4 v0 = src[ln], v1 = src[64 + ln];
                        const uint32_t hw = ln < 2 ? reinterpret_cast<const uint32_t*>(a.hdec + (long)r * DP_D)[2 * cu + ln] : 0u;
                        dp_stq((dp_lu4*)xw + ln, v0); dp_stq((dp_lu4*)xw + 64 + ln, v1);
                        if (ln < 2) misc[DF_M_H0 + 2 * r + ln] = hw;
                    } else {
                        uint32_t v[8];
                        if (!dp_sweep<4>(rgH2 + r * 512, 512, df_tag(base, l - 1, DF_E_H2), v, ln, ab, a.err, 0xB00u + l, a.poll_sleep)) return;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { xw[2 * (j * 64 + ln)] = v[2 * j]; xw[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                    }
                    dp_norm_in_lds((dp_lu4*)xw, g0, g1, a.eps, ln);
                }
                dp_flag((dp_lvu32*)(misc + DF_M_FXA), df_tag(base, l, DF_E_XA));
            }
            {   // q | k | v of both rows -> q buffers, K / V rows of positions 0, 1 (workgroup 0 also files them in the decoder caches)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    uint32_t v[12];
                    if (!dp_sweep<6>(rgQ + r * 768, 768, df_tag(base, l, DF_E_Q), v, ln, ab, a.err, 0xB10u + l, a.poll_sleep)) return;
                    dp_lu32* qw = (dp_lu32*)(lds + DF_OFF_QB + r * 2048);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { qw[2 * (j * 64 + ln)] = v[2 * j]; qw[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                    const int kvh = ln >> 5, wd_ = 2 * (ln & 31);
                    dp_lu32* kr = (dp_lu32*)(lds + DF_OFF_K + ((l * 2 + kvh) * 32 + r) * 256);
                    dp_lu32* vr = (dp_lu32*)(lds + DF_OFF_V + ((l * 2 + kvh) * 32 + r) * 256);
                    kr[wd_] = v[8]; kr[wd_ + 1] = v[9]; vr[wd_] = v[10]; vr[wd_ + 1] = v[11];
                    if (cu == 0) {
                        const long off = (long)l * a.kv_layer_stride + ((long)kvh * 32 + r) * DP_HD;       // elements
                        uint32_t* kg = reinterpret_cast<uint32_t*>(a.kc + off);
                        uint32_t* vg = reinterpret_cast<uint32_t*>(a.vc + off);
                        kg[wd_] = v[8]; kg[wd_ + 1] = v[9]; vg[wd_] = v[10]; vg[wd_ + 1] = v[11];
                    }
                }
                dp_flag((dp_lvu32*)(misc + DF_M_FQ), df_tag(base, l, DF_E_Q));
            }
            df_attention_wave(lds, 7, l, ln);
            {   // rows after the o-projection -> mlp_norm -> xC
                const uint4 g0 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l + 1) * DP_D)[ln], g1 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l + 1) * DP_D)[64 + ln];
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    uint32_t v[8];
                    if (!dp_sweep<4>(rgH1 + r * 512, 512, df_tag(base, l, DF_E_H1), v, ln, ab, a.err, 0xB20u + l, a.poll_sleep)) return;
                    dp_lu32* xw = (dp_lu32*)(lds + DF_OFF_XC + r * 2048);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { xw[2 * (j * 64 + ln)] = v[2 * j]; xw[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                    dp_norm_in_lds((dp_lu4*)xw, g0, g1, a.eps, ln);
                }
                dp_flag((dp_lvu32*)(misc + DF_M_FXC), df_tag(base, l, DF_E_H1));
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) {   // the 256 down-projection partials of my 4 rows -> sum + residual -> the layer's output rows
                uint32_t v[16];
                if (!dp_sweep<8>(rgP + (long)r * (256L * 1024), 1024, df_tag(base, l, DF_E_P), v, ln, ab, a.err, 0xB30u + l, a.poll_sleep)) return;
                float t0_, t1_;
                dp_reduce_partials(v, t0_, t1_);
                const uint32_t h1w = *(dp_lvu32*)(misc + DF_M_H1 + 2 * r + (ln & 1));
                float y0, y1;
                {
#pragma clang fp contract(off)
                    y0 = round_bf(t0_) + lo2f(h1w); y1 = round_bf(t1_) + hi2f(h1w);
                }
                const uint32_t pair = pack_bf(y0, y1);
                if (ln < 2) misc[DF_M_H0 + 2 * r + ln] = pair;
                if (ln < 2 * DP_NREP) dp_gran_store(a.gH2 + (ln >> 1) * 1024 + r * 512 + 2 * cu + (ln & 1), df_tag(base, l, DF_E_H2), pair);
            }
        }
        {   // row 1 of the stack's output -> final norm -> x of the head
            uint32_t v[8];
            const uint4 g0 = reinterpret_cast<const uint4*>(a.dec_norm)[ln], g1 = reinterpret_cast<const uint4*>(a.dec_norm)[64 + ln];
            if (!dp_sweep<4>(rgH2 + 512, 512, df_tag(base, DP_NL - 1, DF_E_H2), v, ln, ab, a.err, 0xB40u, a.poll_sleep)) return;
            dp_lu32* xw = (dp_lu32*)(lds + DF_OFF_XA + 2048);
#pragma unroll
            for (int j = 0; j < 4; ++j) { xw[2 * (j * 64 + ln)] = v[2 * j]; xw[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
            dp_norm_in_lds((dp_lu4*)xw, g0, g1, a.eps, ln);
            dp_flag((dp_lvu32*)(misc + DF_M_FXA), df_tag(base, DP_NL, DF_E_XA));
        }
        if (cu == 0 && lane == 0) __hip_atomic_store(a.epoch, base + DF_EPOCH_STEP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (wave < 4) df_compute_wave<true, 2>(a, lds, wave, lane, cu, base, ropev);
    else df_compute_wave<false, 3>(a, lds, wave, lane, cu, base, ropev);
}
#endif

// the launch, from the translation unit that holds the kernel (csm_dec_first.hip)
hipError_t csm_launch_dec_first(const DecFirstArgs& p, hipStream_t st);
const void* csm_dec_first_kernel();
