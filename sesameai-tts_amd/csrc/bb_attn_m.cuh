// Attention block of ONE backbone layer for a BATCHED decode step (2..32 utterances, one row each) as ONE launch:
//   RMSNorm -> q|k|v projections -> RoPE -> KV append -> attention over keys [0, p_b] -> output projection + residual
// (sesameai/models.py:154-158 through torchtune's TransformerSelfAttentionLayer; CSM-1B backbone shape: d 2048, 32 heads / 8 KV
// heads of 64).  It replaces four launches of the batched chain -- q|k|v (9.5 us at 32 rows), split-K attention (8.8), its
// merge (4.8) and the o-projection's split-K slabs (8.0) -- with the machinery of the batched persistent depth decoder
// (dec_persist_m.cuh): 256 workgroups, projections split over the workgroups' output columns on v_mfma_f32_16x16x32_bf16 (weights
// = A operand: 12 q|k|v rows and 8 o-proj rows per workgroup, 80 KB, requested at entry and held in registers), two interleaved
// 16-row halves, flag-free self-validating exchange buffers, waves 0..3 compute / 4..7 gather.
//   * the layer's input rows are read straight from the residual stream the previous launch left (no exchange: a kernel boundary
//     lies between) and normalised by every workgroup;
//   * workgroup c = (utterance c >> 3, KV head c & 7) owns that row's attention for the four query heads of the KV group: keys
//     0..p-1 stream from the HBM cache (four waves x eight key slots, fp32 online softmax: bb_block.cuh's arithmetic, a key's K row
//     feeds four heads), key p arrives with q through the exchange; wave h folds head h;
//   * the attention vectors travel to every workgroup (row-major, 512 contiguous bytes per owner), the o-projection adds the
//     residual and writes the stream in place.
// Exchange buffers exist twice; launch l uses set l & 1 and re-poisons its own slots of the other set at entry (the previous
// launch, which used it, has ended; the next one will find it poisoned) -- 16 layers, an even number, so a frame step ends where
// it began and the captured graph replays unchanged.
#pragma once
#include "dec_persist_m.cuh"
#include "bb_block.cuh"

#define BM_D 2048
#define BM_Q_BYTES (256 * 32 * 12 * 2)            // q|k|v: [256 workgroups][32 rows][12 columns] bf16
#define BM_A_BYTES (32 * 2048 * 2)                // attention output, row-major
#define BM_SET_BYTES (BM_Q_BYTES + BM_A_BYTES)
#define BM_XCHG_BYTES (2 * BM_SET_BYTES)
#define BM_KR 8                                   // K (and V) rows per lane per round: 4 waves x 8 slots x 8 = 256 keys (12 spilled 39 registers)

// LDS image (bytes)
#define BM_L_XB 0                                 // [2 halves] 64 KB: activations in B-operand order (256 pieces x 16 rows x 16 bytes)
#define BM_L_RED 131072                           // [2 halves][4 waves][4][64] f32
#define BM_L_NORM (BM_L_RED + 8192)               // sa_norm scale, 4 KB
#define BM_L_ROPE (BM_L_NORM + 4096)              // [32 rows][6 pairs] (cos, sin) of my q|k|v pairs at the rows' positions
#define BM_L_Q (BM_L_ROPE + 768)                  // my row's q of the 4 heads (512 B) | k_new (128) | v_new (128)
#define BM_L_PART (BM_L_Q + 768)                  // [5 partials][4 heads][66] f32
#define BM_L_MISC (BM_L_PART + 5 * 4 * 66 * 4)
#define BM_LDS_BYTES (BM_L_MISC + 256)
static_assert(BM_LDS_BYTES <= 163840, "LDS image exceeds 160 KB");
#define BM_M_FILL 0       // [2 halves][2 fills]
#define BM_M_CDONE 4      // [2]
#define BM_M_RED 6        // [2]
#define BM_M_ABORT 8
#define BM_M_CNT 9        // attention partial arrivals
#define BM_M_BAR 10

struct BbAttnMArgs {
    const bf16_t *wq, *wk, *wv, *wo, *sa_norm;
    const bf16_t* rope;                   // [max_seq][32][2]
    bf16_t* h;                            // [M][2048] residual stream, updated in place
    const bf16_t* xn;                     // optional: sa_norm(h) already computed by the previous launch (the batched chain's finisher): row-major
    int xn_packed;                        //           [M][2048], or (xn_packed) in k_mm32's operand order (common.cuh xp_off)
    bf16_t *kc, *vc;                      // this layer's caches [max_batch][8][smax][64]
    const int* pos;                       // [M] device: position of each row's step
    int smax, M;
    float eps;
    char* xchg;                           // BM_XCHG_BYTES
    int set;                              // exchange set of this launch (layer & 1)
    uint32_t* err;
    int poll_sleep;
};

__device__ __forceinline__ bool bm_poll3(const char* p0, const char* p1, const char* p2, uint32_t (&x)[3], int lane, dp_lvu32* ab, uint32_t* err, uint32_t code, int poll_sleep) {
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t pass = 1;; ++pass) {
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(x[0]) : "v"(p0) : "memory");
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(x[1]) : "v"(p1) : "memory");
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(x[2]) : "v"(p2) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]));
        if (__all(x[0] != 0xffffffffu && x[1] != 0xffffffffu && x[2] != 0xffffffffu)) return true;
        if ((pass & 15u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
        for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
}

template <int NH>
__global__ __launch_bounds__(512) void k_bb_attn_m(const BbAttnMArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    const int lane = threadIdx.x & 63;
    dp_lu32* misc = (dp_lu32*)(lds + BM_L_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + BM_M_ABORT);
    char* const setp = a.xchg + a.set * BM_SET_BYTES;                      // this launch's q|k|v and attention buffers
    char* const other = a.xchg + (a.set ^ 1) * BM_SET_BYTES;
    const int ob = cu >> 3, kvh = cu & 7;                                  // the (row, KV head) this workgroup owns
    const int ohf = ob < a.M ? (ob >> 4) : -1;
    if (threadIdx.x < 64) misc[threadIdx.x] = 0;
    // ---- re-poison my slots of the other set (the next launch polls them) ----
    for (int i = threadIdx.x; i < 32 * 6; i += 512) dm_st4(other + cu * 768 + i * 4, 0xffffffffu);
    if (ob < 32) for (int i = threadIdx.x; i < 128; i += 512) dm_st4(other + BM_Q_BYTES + ob * 4096 + kvh * 512 + i * 4, 0xffffffffu);
    for (int i = threadIdx.x; i < 256; i += 512) dp_stq((dp_lu4*)(lds + BM_L_NORM) + i, reinterpret_cast<const uint4*>(a.sa_norm)[i]);
    if (threadIdx.x >= 256 && threadIdx.x < 256 + 32 * 6) {
        const int i = threadIdx.x - 256, b = i / 6, u = i % 6;
        const int R = 12 * cu + 2 * u;                                     // row of [q; k; v]: q 0..2047, k 2048..2559, v 2560..3071
        const int p = min(max(a.pos[min(b, a.M - 1)], 0), a.smax - 1);
        ((dp_lu32*)(lds + BM_L_ROPE))[i] = reinterpret_cast<const uint32_t*>(a.rope)[(long)p * 32 + (R % 64) / 2];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    if (wave < 4) {
        // ------------------------------------------------------------------------------------------------ compute wave
        const int w = wave, g4 = lane >> 4, bl = lane & 15, r16 = lane & 15;
        uint4 S[16], S2[16];
        {
            const int R = 12 * cu + min(r16, 11);
            const bf16_t* wr = R < 2048 ? a.wq + (long)R * BM_D : R < 2560 ? a.wk + (long)(R - 2048) * BM_D : a.wv + (long)(R - 2560) * BM_D;
            const bf16_t* wo = a.wo + (long)(8 * cu + min(r16, 7)) * BM_D;
#pragma unroll
            for (int k = 0; k < 16; ++k) S[k] = ldg16<true>(reinterpret_cast<const uint4*>(wr + 32 * (16 * w + k) + 8 * g4));
#pragma unroll
            for (int k = 0; k < 16; ++k) S2[k] = ldg16<true>(reinterpret_cast<const uint4*>(wo + 32 * (16 * w + k) + 8 * g4));
        }
        // my 4 residual columns of my row in each half (lanes g4 < 2: columns 8 cu + 4 g4 ..); named registers, not an array indexed by the
        // half: that went to scratch
        uint2 hres0, hres1;
        hres0 = *reinterpret_cast<const uint2*>(a.h + (long)min(bl, a.M - 1) * BM_D + 8 * cu + 4 * (g4 & 1));
        hres1 = *reinterpret_cast<const uint2*>(a.h + (long)min(16 + bl, a.M - 1) * BM_D + 8 * cu + 4 * (g4 & 1));
        auto fold = [&](int hf, const dm_f32x4& acc, float (&v)[4]) -> bool {
            dp_lf32* red = (dp_lf32*)(lds + BM_L_RED) + hf * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(w * 4 + i) * 64 + lane] = acc[i];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            uint32_t old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(misc + BM_M_RED + hf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old);
            if ((old & 3u) != 3u) return false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = red[i * 64 + lane];
#pragma unroll
                for (int ww = 1; ww < 4; ++ww) s += red[(ww * 4 + i) * 64 + lane];
                v[i] = s;
            }
            return true;
        };
#pragma unroll 1
        for (int hf = 0; hf < NH; ++hf) {
            // ---- q|k|v of my 12 columns -> RoPE -> exchange (+ K / V cache rows for later steps) ----
            if (!dm_wait_ge((dp_lvu32*)(misc + BM_M_FILL + hf * 2), 4u, ab, a.err, 0xD10u, lane)) return;
            dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            dm_mma<16>(S, lds + BM_L_XB + hf * 65536, 16 * w, lane, acc);
            dm_arrive(misc + BM_M_CDONE + hf, lane);
            float v[4];
            if (fold(hf, acc, v)) {
                const int b = 16 * hf + bl;
                if (g4 < 3 && b < a.M) {
                    const int R = 12 * cu + 4 * g4;                          // rows R..R+3: pairs (R, R+1), (R+2, R+3)
                    const dp_lu32* rp = (const dp_lu32*)(lds + BM_L_ROPE) + b * 6 + 2 * g4;
                    const uint32_t o0 = dp_rope_pair(v[0], v[1], rp[0], R < 2560), o1 = dp_rope_pair(v[2], v[3], rp[1], R + 2 < 2560);
                    dm_sst8(setp, (uint32_t)(cu * 768 + b * 24 + g4 * 8), o0, o1);
                    if (R >= 2048) {                                        // k / v rows also go to the cache (a 4-row group never straddles q|k or k|v: 2048, 2560 are multiples of 4)
                        const int p = min(max(a.pos[b], 0), a.smax - 1);
                        const int rk = R < 2560 ? R - 2048 : R - 2560;
                        bf16_t* dst = (R < 2560 ? a.kc : a.vc) + (((long)b * 8 + rk / 64) * a.smax + p) * 64 + rk % 64;
                        *reinterpret_cast<uint2*>(dst) = make_uint2(o0, o1);
                    }
                }
            }
        }
#pragma unroll 1
        for (int hf = 0; hf < NH; ++hf) {
            // ---- o-projection of my 8 columns + residual -> the stream ----
            if (!dm_wait_ge((dp_lvu32*)(misc + BM_M_FILL + hf * 2 + 1), 4u, ab, a.err, 0xD20u, lane)) return;
            dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            dm_mma<16>(S2, lds + BM_L_XB + hf * 65536, 16 * w, lane, acc);
            float v[4];
            if (fold(hf, acc, v)) {
                const int b = 16 * hf + bl;
                if (g4 < 2 && b < a.M) {
                    const uint2 hr = hf == 0 ? hres0 : hres1;
                    const uint32_t p0 = dp_resid_pair(v[0], v[1], hr.x), p1 = dp_resid_pair(v[2], v[3], hr.y);
                    *reinterpret_cast<uint2*>(a.h + (long)b * BM_D + 8 * cu + 4 * g4) = make_uint2(p0, p1);
                }
            }
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------------- gather wave
    __builtin_amdgcn_s_setprio(2);
    const int gw = wave - 4;
    const int ps = a.poll_sleep;
    const int slot = lane >> 3, e8 = lane & 7;
    int p_own = 0;
    uint4 kr[BM_KR], vr[BM_KR];
    const bf16_t *kb = nullptr, *vb = nullptr;
    // ---- the layer's input rows -> activation buffer (rows 4 gw .. + 3 of each half; lane = (row, piece group)): the normalised rows the
    //      previous launch left if there are any (every workgroup normalising all 32 rows itself costs ~3 us per half), else residual
    //      stream -> RMSNorm here ----
#pragma unroll 1
    for (int hf = 0; hf < NH; ++hf) {
        const int row = gw * 4 + (lane & 3), pg = lane >> 2;
        const int b = min(16 * hf + row, a.M - 1);
        dp_lu4* xb = (dp_lu4*)(lds + BM_L_XB + hf * 65536);
        uint4 x[16];
        if (a.xn != nullptr) {
            if (a.xn_packed) {
                // piece k8 = 16 j + pg of row b sits at ((chunk * 4 + q) * 64 + hbit * 32 + (b & 31)) with chunk = k8 >> 3, hbit = (k8 >> 2) & 1, q = k8 & 3
                const uint4* src = reinterpret_cast<const uint4*>(a.xn + (long)(b >> 5) * 32 * BM_D) + (b & 31);
#pragma unroll
                for (int j = 0; j < 16; ++j) { const int k8 = j * 16 + pg; x[j] = src[((k8 >> 3) * 4 + (k8 & 3)) * 64 + ((k8 >> 2) & 1) * 32]; }
            } else {
                const uint4* src = reinterpret_cast<const uint4*>(a.xn + (long)b * BM_D);
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = src[j * 16 + pg];
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) dp_stq(xb + (j * 16 + pg) * 16 + row, x[j]);
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(a.h + (long)b * BM_D);
#pragma unroll
            for (int j = 0; j < 16; ++j) x[j] = src[j * 16 + pg];
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) ss += dp_chunk_ss(x[j]);
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) ss += __shfl_xor(ss, o, 64);
            const float r = 1.0f / sqrtf(ss / (float)BM_D + a.eps);
            const dp_lu4* g = (const dp_lu4*)(lds + BM_L_NORM);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                asm volatile("" : "+v"(x[j].x), "+v"(x[j].y), "+v"(x[j].z), "+v"(x[j].w));
                dp_stq(xb + (j * 16 + pg) * 16 + row, dp_chunk_norm(x[j], dp_ldq(g + j * 16 + pg), r));
            }
        }
        dm_arrive(misc + BM_M_FILL + hf * 2, lane);
    }
    // ---- attention of my (row, KV head): q of 4 heads + the step's own k / v from the exchange, keys 0..p-1 from the cache ----
    if (ohf >= 0) {
        // K / V rows of keys 0..p-1, requested now (they do not depend on this step; the q|k|v exchange takes longer than they do to arrive):
        // keys 32 j + 8 gw + slot, 16-byte piece e8 of each 128-byte row.  (Requested before the normalisation above they spilled: 96 registers
        // beside its 64.)
        p_own = min(max((int)dp_sload32(a.pos + ob), 0), a.smax - 1);
        kb = a.kc + ((long)ob * 8 + kvh) * a.smax * 64;
        vb = a.vc + ((long)ob * 8 + kvh) * a.smax * 64;
#pragma unroll
        for (int j = 0; j < BM_KR; ++j) {
            const int key = min(32 * j + 8 * gw + slot, max(p_own - 1, 0));
            kr[j] = *reinterpret_cast<const uint4*>(kb + (long)key * 64 + e8 * 8);
            vr[j] = *reinterpret_cast<const uint4*>(vb + (long)key * 64 + e8 * 8);
        }
        if (gw == 0) {
            // pair u of [q; k; v] lives at workgroup u / 6, slot u % 6: q pairs 128 kvh + (0..127), k pair 1024 + 32 kvh + (0..31), v pair 1280 + 32 kvh + ..
            const char* qb = setp + ob * 24;
            const int u0 = 128 * kvh + lane, u1 = 128 * kvh + 64 + lane, u2 = lane < 32 ? 1024 + 32 * kvh + lane : 1280 + 32 * kvh + (lane - 32);
            uint32_t x[3];
            if (!bm_poll3(qb + (u0 / 6) * 768 + (u0 % 6) * 4, qb + (u1 / 6) * 768 + (u1 % 6) * 4, qb + (u2 / 6) * 768 + (u2 % 6) * 4, x, lane, ab, a.err, 0xD30u, ps)) return;
            dp_lu32* ql = (dp_lu32*)(lds + BM_L_Q);
            ql[lane] = x[0]; ql[64 + lane] = x[1]; ql[128 + lane] = x[2];          // q (128 words) | k_new (32) | v_new (32)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        DpQuadSync bar{(dp_lvu32*)(misc + BM_M_BAR), ab, a.err, lane, nullptr, nullptr};
        uint32_t phase = 0;
        bar.phase = &phase;
        bar();
        uint4 qv[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) qv[h] = dp_ldq((const dp_lu4*)(lds + BM_L_Q) + h * 8 + e8);
        float mx[4], l[4], o[4][8];
#pragma unroll
        for (int h = 0; h < 4; ++h) { mx[h] = -INFINITY; l[h] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[h][i] = 0.f; }
        for (int k0 = 0; k0 < p_own; k0 += 32 * BM_KR) {
            if (k0 > 0) {
#pragma unroll
                for (int j = 0; j < BM_KR; ++j) {
                    const int key = min(k0 + 32 * j + 8 * gw + slot, p_own - 1);
                    kr[j] = *reinterpret_cast<const uint4*>(kb + (long)key * 64 + e8 * 8);
                    vr[j] = *reinterpret_cast<const uint4*>(vb + (long)key * 64 + e8 * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < BM_KR; ++j) {
                const bool live = k0 + 32 * j + 8 * gw + slot < p_own;
                const float vf[8] = {lo2f(vr[j].x), hi2f(vr[j].x), lo2f(vr[j].y), hi2f(vr[j].y), lo2f(vr[j].z), hi2f(vr[j].z), lo2f(vr[j].w), hi2f(vr[j].w)};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    float s = bb_sum8(dot8(qv[h], kr[j], 0.f)) * 0.125f;
                    s = live ? s : -INFINITY;
                    const float mn = fmaxf(mx[h], s);
                    const float corr = (mx[h] == -INFINITY) ? 0.f : __expf(mx[h] - mn);
                    const float pw = live ? __expf(s - mn) : 0.f;
                    l[h] = l[h] * corr + pw;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[h][i] = o[h][i] * corr + pw * vf[i];
                    mx[h] = live ? mn : mx[h];
                }
            }
        }
        // merge the 8 key slots of the wave (lanes sharing e8), per head
        dp_lf32* part = (dp_lf32*)(lds + BM_L_PART);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                const float mo = __shfl_xor(mx[h], off, WAVE), lo = __shfl_xor(l[h], off, WAVE);
                const float mn = fmaxf(mx[h], mo);
                const float c0 = (mx[h] == -INFINITY) ? 0.f : __expf(mx[h] - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
                l[h] = l[h] * c0 + lo * c1;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float oo = __shfl_xor(o[h][i], off, WAVE); o[h][i] = o[h][i] * c0 + oo * c1; }
                mx[h] = mn;
            }
            if (slot == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) part[(gw * 4 + h) * 66 + e8 * 8 + i] = o[h][i];
                if (e8 == 0) { part[(gw * 4 + h) * 66 + 64] = mx[h]; part[(gw * 4 + h) * 66 + 65] = l[h]; }
            }
        }
        if (gw == 3) {
            // the step's own key (position p): a fifth partial with a single key, per head
            const uint4 kn = dp_ldq((const dp_lu4*)(lds + BM_L_Q + 512) + e8), vn = dp_ldq((const dp_lu4*)(lds + BM_L_Q + 640) + e8);
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float s = bb_sum8(dot8(qv[h], kn, 0.f)) * 0.125f;
                if (slot == 1) {
                    part[(16 + h) * 66 + e8 * 8 + 0] = lo2f(vn.x); part[(16 + h) * 66 + e8 * 8 + 1] = hi2f(vn.x);
                    part[(16 + h) * 66 + e8 * 8 + 2] = lo2f(vn.y); part[(16 + h) * 66 + e8 * 8 + 3] = hi2f(vn.y);
                    part[(16 + h) * 66 + e8 * 8 + 4] = lo2f(vn.z); part[(16 + h) * 66 + e8 * 8 + 5] = hi2f(vn.z);
                    part[(16 + h) * 66 + e8 * 8 + 6] = lo2f(vn.w); part[(16 + h) * 66 + e8 * 8 + 7] = hi2f(vn.w);
                    if (e8 == 0) { part[(16 + h) * 66 + 64] = s; part[(16 + h) * 66 + 65] = 1.0f; }
                }
            }
        }
        bar();
        {
            // wave gw folds head gw: lane = output dimension, the 5 partials in order
            float Mx = -INFINITY, L = 0.f, O = 0.f;
#pragma unroll
            for (int pw_ = 0; pw_ < 5; ++pw_) {
                const int idx = (pw_ * 4 + gw) * 66;
                const float mw = part[idx + 64], lw = part[idx + 65], ow = part[idx + lane];
                const float mn = fmaxf(Mx, mw);
                const float c0 = (Mx == -INFINITY) ? 0.f : __expf(Mx - mn), c1 = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
                L = L * c0 + lw * c1; O = O * c0 + ow * c1; Mx = mn;
            }
            const float y = O / L;
            const float yn = __shfl_xor(y, 1, WAVE);
            if ((lane & 1) == 0)
                dm_st4(setp + BM_Q_BYTES + ob * 4096 + (4 * kvh + gw) * 128 + (lane >> 1) * 4, pack_bf(y, yn));
        }
    }
    // ---- every workgroup: the attention vectors of all rows -> activation buffer ----
#pragma unroll 1
    for (int hf = 0; hf < NH; ++hf) {
        if (!dm_wait_ge((dp_lvu32*)(misc + BM_M_CDONE + hf), 4u, ab, a.err, 0xD40u, lane)) return;
        const int rowc = min(16 * hf + gw * 4 + (lane & 3), a.M - 1), pg = lane >> 2;
        if (!dm_sweep_mat<256, 256>(setp + BM_Q_BYTES, (uint32_t)(rowc * 4096 + pg * 16), lds + BM_L_XB + hf * 65536, gw, lane, ab, a.err, 0xD50u, ps)) return;
        dm_arrive(misc + BM_M_FILL + hf * 2 + 1, lane);
    }
}
