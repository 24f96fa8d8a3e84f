// ALL backbone layers of a batch-1 decode step as ONE launch (round 5): k_bb_layer's body (bb_block.cuh) inside a loop over the layers, with
// the launch boundary between two layers replaced by one more granule all-gather (the 2048-long residual row: every CU's gather wave owns eight
// of its elements) and the NEXT layer's weight requests issued as soon as the registers / LDS that will hold them are free -- right after the
// wave has read layer l's W2 slice out of LDS, ~6 us before layer l ends and ~9 us before a new launch could have issued them -- so HBM keeps
// streaming through the last hand-offs of a layer and the first ones of the next (in the 16-launch form it idles there).
//   * the next layer's q | k | v row pair and o-projection rows of a wave travel by global_load_lds into the wave's OWN 16 KB of the W2 area
//     (dead between the down projection and the next one); the part of the next W2 slice they displace is asked for after they are consumed
//     (after q | k | v for waves 4, 5; after the o-projection for waves 0..3: 14 us before the down projection needs it).  In registers -- where
//     k_bb_layer holds them -- they would have to live across the loop's back edge next to the (gate, up) pairs: 64 VGPRs the kernel does not have,
//   * the K / V rows of an attention CU or the first three (gate, up) pairs: registers, as in k_bb_layer.
// The arithmetic, its order and every rounding are k_bb_layer's: the two forms produce the same bits (tests compare them).
#pragma once
#include <cstddef>
#include <type_traits>
#include "bb_block.cuh"

struct BbStackLayer {                     // one per layer, in device memory, read through the scalar cache
    const void *wq, *wk, *wv, *wo, *w1, *w3;          // bf16 rows (F8: e4m3 bytes)
    const uint4* w2t;                                  // k_bb_retile_w2(_fp8)'s pieces of this layer
    const bf16_t *sa_norm, *mlp_norm;
    bf16_t *kc, *vc;
    const float *sq, *sk, *sv, *so, *s1, *s3, *s2;    // F8: per-output-row scales
};
struct BbStackArgs {
    const BbStackLayer* layers;
    int n_layers;
    const bf16_t* rope;
    bf16_t* h;
    const int* pos;
    int smax;
    float eps;
    dp_u64 *gQ, *gA, *gS, *gH, *gP, *gX;  // gX: [8][1024] the residual row between two layers
    uint32_t *err, *epoch;
    int poll_sleep;
    dp_u64* stamps;
};

#define BS_L_X (BL_LDS_BYTES)             // 4096: the layer's input row
#define BS_L_G1 (BL_LDS_BYTES + 4096)     // 4096: sa_norm's scale
#define BS_L_HL2 (BL_LDS_BYTES + 8192)    // 64: the second copy of the CU's 32 h values (odd layers)
#define BS_LDS_BYTES (BL_LDS_BYTES + 8192 + 64)
#define BS_M_FX 6
#define BS_TAGS 8                         // tags per layer: 1 Q, 2 A, 3 S, 4 H, 5 P, 6 X

// The table's pointers come out of scalar loads as integers: give them their address space (global; the scales: constant, so that a
// uniform index is a scalar load) -- as generic pointers every weight load would be a flat_load with a 64-bit address per load.
typedef __attribute__((address_space(4))) const unsigned long long dp_cu64;
typedef __attribute__((address_space(4))) const float bs_cf32;
typedef __attribute__((address_space(1))) const char bs_gc;
typedef __attribute__((address_space(1))) const u32x4_t bs_gu4;
#define BS_U64(l_, field_) (*(dp_cu64*)((unsigned long long)a.layers + (unsigned long long)(l_) * sizeof(BbStackLayer) + offsetof(BbStackLayer, field_)))
#define BS_G(l_, field_) ((bs_gc*)BS_U64(l_, field_))
#define BS_SCALE(l_, field_, i_) (*((bs_cf32*)BS_U64(l_, field_) + (i_)))
// 16 bytes at (uniform base) + (32-bit unsigned lane offset): global_load_dwordx4 v, v_off, s[base] offset:imm
__device__ __forceinline__ uint4 bs_ldg(bs_gc* base, unsigned off) {
    const u32x4_t v = __builtin_nontemporal_load((bs_gu4*)(base + off));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint4 bs_ld(bs_gc* base, unsigned off) { const u32x4_t v = *(bs_gu4*)(base + off); return make_uint4(v.x, v.y, v.z, v.w); }

template <int N> __device__ __forceinline__ void bs_wait_vm() { static_assert(N >= 0 && N < 64, "vmcnt is 6 bits"); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
#define BS_DEAD(u4_) asm volatile("" : "=v"((u4_).x), "=v"((u4_).y), "=v"((u4_).z), "=v"((u4_).w))
#ifdef DP_TIMELINE
#define BS_STAMP(i_, cond_) do { if (a.stamps != nullptr && l == 8 && cu == 100 && lane == 0 && (cond_)) a.stamps[i_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BS_STAMP(i_, cond_) do { } while (0)
#endif

// CREDIT > 0: the next-layer requests of a wave are paced -- before each one the wave waits until at most CREDIT of its vector-memory
// operations are outstanding (0: all at once)
template <bool F8, int CREDIT>
__global__ __launch_bounds__(512) void k_bb_stack(const BbStackArgs a) {
    using paced_t = std::integral_constant<bool, true>; using burst_t = std::integral_constant<bool, false>;
    constexpr int NW = F8 ? 2 : 4;                                      // 16-byte weight pieces per lane per 2048-long row
    constexpr int EB = F8 ? 1 : 2;                                      // bytes per weight
    extern __shared__ __attribute__((aligned(16))) char lds[];          // BS_LDS_BYTES
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    int lane = threadIdx.x & 63;
    const int nl = a.n_layers;
    auto xi = [&](int q) -> int { return F8 ? 2 * ((q >> 1) * 64 + lane) + (q & 1) : q * 64 + lane; };      // index of a lane's q-th 16-byte activation piece
    dp_lu32* misc = (dp_lu32*)(lds + BB_L_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + BB_M_ABORT);
    if (threadIdx.x < 16) misc[threadIdx.x] = 0;
    // ---- layer 0's input row and the two norm scales -> LDS (later layers: the gather wave puts them there) ---------------------------
    {
        const int t = threadIdx.x;
        if (t < 256) {
            const uint4 x = reinterpret_cast<const uint4*>(a.h)[t], g2 = bs_ld(BS_G(0, mlp_norm), 16u * t);
            dp_stq((dp_lu4*)(lds + BS_L_X) + t, x); dp_stq((dp_lu4*)(lds + BL_L_G2) + t, g2);
        } else {
            dp_stq((dp_lu4*)(lds + BS_L_G1) + (t - 256), bs_ld(BS_G(0, sa_norm), 16u * (t - 256)));
        }
    }
    // ---- per-wave constants ------------------------------------------------------------------------------------------------------------
    const int orow = 8 * cu + 2 * (wave & 3);                         // waves 0..3: output rows orow, orow + 1
    const int pair = 6 * cu + wave;                                   // waves 0..5: rows 2 pair, 2 pair + 1 of [q; k; v]
    const int R0 = 2 * (wave < 6 ? pair : 0);
    const int rq = R0 < 2048 ? R0 : R0 < 2560 ? R0 - 2048 : R0 - 2560;  // row inside its matrix
    const uint32_t base = dp_sload32(a.epoch);
    const int p = min(max((int)dp_sload32(a.pos), 0), a.smax - 1);
    const int e0 = R0 % BB_HD;
    const uint32_t cs = dp_sload32(reinterpret_cast<const uint32_t*>(a.rope) + (long)p * (BB_HD / 2) + e0 / 2);
    const int nsplit = p >= BB_KMAX ? 8 : 1;
    const bool attn_cu = cu < BB_NH * nsplit;
    const int head = cu % BB_NH, split = cu / BB_NH, kvh = head / (BB_NH / BB_NKV);
    const int chunk = (p + nsplit - 1) / nsplit, k_lo = split * chunk, k_hi = min(p, k_lo + chunk);
    int slot = lane >> 3, e8 = lane & 7;                              // (recomputed per layer, like every lane-derived value)

    float sc0 = 1.f, sc1 = 1.f, so0 = 1.f, so1 = 1.f;                   // F8: the rows' scales
    uint4 buf[24], gu3[8];
    // the request forms (L = the layer whose weights are asked for)
    // this wave's 16 KB of the W2 area: 16 pieces of 1 KB (lane l's 16 bytes at 16 l of a piece)
    char* const slice = lds + BL_L_W2 + wave * 16 * 1024;
    auto dma = [&](bs_gc* src, int piece, auto paced) {                  // 1 KB: 64 lanes x 16 bytes at src + 16 lane -> piece
        if constexpr (decltype(paced)::value && CREDIT > 0) bs_wait_vm<CREDIT>();
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + 16u * lane), (void __attribute__((address_space(3)))*)(slice + piece * 1024), 16, 0, 0);
    };
    // q | k | v row pair of layer L -> pieces [0, 2 NW) (row r, chunk i at r NW + i); o-projection rows -> pieces [2 NW, 4 NW)
    auto dma_qkv = [&](int L, auto paced) {
        // (a select between three complete addresses, as in k_bb_layer)
        bs_gc* wr = R0 < 2048 ? BS_G(L, wq) + (long)R0 * BB_D * EB : R0 < 2560 ? BS_G(L, wk) + (long)(R0 - 2048) * BB_D * EB : BS_G(L, wv) + (long)(R0 - 2560) * BB_D * EB;
        if (F8) {
            if (R0 < 2048) { sc0 = BS_SCALE(L, sq, rq); sc1 = BS_SCALE(L, sq, rq + 1); }
            else if (R0 < 2560) { sc0 = BS_SCALE(L, sk, rq); sc1 = BS_SCALE(L, sk, rq + 1); }
            else { sc0 = BS_SCALE(L, sv, rq); sc1 = BS_SCALE(L, sv, rq + 1); }
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) { dma(wr + 1024 * i, i, paced); dma(wr + BB_D * EB + 1024 * i, NW + i, paced); }
    };
    auto dma_wo = [&](int L, auto paced) {
        bs_gc* wop = BS_G(L, wo) + (long)orow * BB_D * EB;
        if (F8) { so0 = BS_SCALE(L, so, orow); so1 = BS_SCALE(L, so, orow + 1); }
#pragma unroll
        for (int i = 0; i < NW; ++i) { dma(wop + 1024 * i, 2 * NW + i, paced); dma(wop + BB_D * EB + 1024 * i, 3 * NW + i, paced); }
    };
    auto load_gu = [&](bs_gc* w1p, bs_gc* w3p, int i, int gq, int c, auto paced) -> uint4 {       // pair i of this wave, gq 0 = gate row, 1 = up row, chunk c
        const long prow = 32L * cu + 4 * wave + i;
        if constexpr (decltype(paced)::value && CREDIT > 0) bs_wait_vm<CREDIT>();
        return bs_ldg((gq ? w3p : w1p) + prow * BB_D * EB, 16u * lane + 1024u * c);
    };
    auto req_buf = [&](int L, auto paced) {                                         // (gate, up) pieces of pairs 0..2 of this wave, in the order (pair, gate | up, chunk)
        bs_gc *w1p = BS_G(L, w1), *w3p = BS_G(L, w3);
#pragma unroll
        for (int q = 0; q < 6 * NW; ++q) buf[q] = load_gu(w1p, w3p, q / (2 * NW), (q / NW) & 1, q % NW, paced);
    };
    auto req_gu3 = [&](int L) {
        bs_gc *w1p = BS_G(L, w1), *w3p = BS_G(L, w3);
#pragma unroll
        for (int q = 0; q < 2 * NW; ++q) gu3[q] = load_gu(w1p, w3p, 3, q / NW, q % NW, burst_t{});
    };
    auto req_kv = [&](int L, int k0, auto paced) {                                  // K / V rows of keys k0 .. of this CU's range (slots past the last key re-read it: weight 0)
        bs_gc* kb = BS_G(L, kc) + (long)kvh * a.smax * BB_HD * 2;
        bs_gc* vb = BS_G(L, vc) + (long)kvh * a.smax * BB_HD * 2;
#pragma unroll
        for (int j = 0; j < BB_KMAX / 64; ++j) {
            const int key = min(k0 + 64 * j + 8 * wave + slot, max(k_hi - 1, 0));
            if constexpr (decltype(paced)::value && CREDIT > 0) bs_wait_vm<CREDIT>();
            buf[j] = bs_ld(kb, (unsigned)key * (BB_HD * 2) + 16u * e8);
            if constexpr (decltype(paced)::value && CREDIT > 0) bs_wait_vm<CREDIT>();
            buf[12 + j] = bs_ld(vb, (unsigned)key * (BB_HD * 2) + 16u * e8);
        }
    };
    auto req_w2_lds = [&](int L, int q_lo, int q_hi, auto paced) {                  // pieces [q_lo, q_hi) of this wave's W2 slice of layer L (piece (row block rb, k chunk kc) at rb NW + kc)
        bs_gc* w2t = BS_G(L, w2t);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the wave's reads of what the pieces held are done)
#pragma unroll
        for (int q = 0; q < 4 * NW; ++q)
            if (q >= q_lo && q < q_hi) dma(w2t + 16 * (((long)cu * NW + (q % NW)) * BB_D + 256 * wave + 64 * (q / NW)), q, paced);
    };
    // the requests of layer L that do not wait for anything of layer L, in the order they are consumed.  VMEM operations a wave issues AFTER its
    // q | k | v pieces here: the K / V rows or (gate, up) pieces (waves < 7) and, waves 4..6, the W2 pieces -- bs_wait_rows() counts on exactly that
    // CREDIT < 0: only the rows and the K / V rows travel ahead; the MLP share is asked for after the layer's q | k | v pair is published (req_bulk)
    constexpr bool LATE = CREDIT < 0;
    auto req_bulk = [&](int L, auto paced) {
        if (wave < 7 && !attn_cu) req_buf(L, paced);
        if (wave == 4 || wave == 5) req_w2_lds(L, 2 * NW, 4 * NW, paced);
        if (wave == 6) req_w2_lds(L, 0, 4 * NW, paced);
    };
    auto req_layer = [&](int L, auto paced) {
        if (wave < 4) dma_wo(L, paced);
        if (wave < 6) dma_qkv(L, paced);
        if (wave < 7 && attn_cu) req_kv(L, k_lo, paced);
        if (!LATE) req_bulk(L, paced);
    };
    // ---- layer 0's requests, in the order they are consumed ------------------------------------------------------------------------------
    if (wave == 7 && attn_cu) req_kv(0, k_lo, burst_t{});
    req_layer(0, burst_t{});
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // X / G1 / G2 / misc are in LDS (a bare barrier: the weight loads stay in flight)

    for (int l = 0; l < nl; ++l) {
        const uint32_t tb = base + (uint32_t)(BS_TAGS * l);
        const uint32_t tagQ = tb + 1u, tagA = tb + 2u, tagS = tb + 3u, tagH = tb + 4u, tagP = tb + 5u, tagX = tb + 6u;
        const uint32_t cnt_target = 8u * (uint32_t)(l + 1);
        // (every lane-derived address of the body is recomputed per layer: hoisted out of the loop they are ~100 64-bit values, which spill)
        asm volatile("" : "+v"(lane));
        lane &= 63;
        slot = lane >> 3; e8 = lane & 7;
        // ---- the layer's input row: layers > 0 wait for the gather wave's all-gather (end of the loop body) -------------------------
        if (l > 0 && wave != 7 && !bb_wait_flag((dp_lvu32*)(misc + BS_M_FX), tb, ab, a.err, 0xC0Bu, lane)) return;
        BS_STAMP(13, wave == 0);
        uint32_t hres = 0;
        if (wave < 4) hres = ((const dp_lu32*)(lds + BS_L_X))[orow >> 1];
        // ---- q | k | v pair of this wave: RMSNorm in registers -> dot -> RoPE -> granule (8 replicas) + KV cache ----------------------
        if (wave < 6) {
            uint4 xn[4];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { xn[i] = dp_ldq((const dp_lu4*)(lds + BS_L_X) + xi(i)); ss += dp_chunk_ss(xn[i]); }
            ss = wave_sum(ss);
            const float r = 1.0f / sqrtf(ss / (float)BB_D + a.eps);
#pragma unroll
            for (int i = 0; i < 4; ++i) xn[i] = dp_chunk_norm(xn[i], dp_ldq((const dp_lu4*)(lds + BS_L_G1) + xi(i)), r);
            // this wave's row pair is in its slice once at most the loads issued after it are outstanding
            if (LATE) { if (attn_cu) bs_wait_vm<24>(); else bs_wait_vm<0>(); }
            else if (wave < 4) { if (attn_cu) bs_wait_vm<24>(); else bs_wait_vm<6 * NW>(); }
            else { if (attn_cu) bs_wait_vm<24 + 2 * NW>(); else bs_wait_vm<8 * NW>(); }
            BS_STAMP(0, wave == 0);                         // (a store: after the counted wait)
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint4 w0 = dp_ldq((const dp_lu4*)(slice + i * 1024) + lane), w1 = dp_ldq((const dp_lu4*)(slice + (NW + i) * 1024) + lane);
                if (F8) { a0 = dot16_fp8(w0, xn[2 * i], xn[2 * i + 1], a0); a1 = dot16_fp8(w1, xn[2 * i], xn[2 * i + 1], a1); }
                else { a0 = dot8(w0, xn[i], a0); a1 = dot8(w1, xn[i], a1); }
            }
            a0 = wave_sum(a0) * sc0; a1 = wave_sum(a1) * sc1;
            const uint32_t outw = dp_rope_pair(a0, a1, cs, R0 < 2560);
            if (lane < DP_NREP) dp_gran_store(a.gQ + lane * BB_NQKV_PAIRS + pair, tagQ, outw);
            BS_STAMP(1, wave == 0);
            if (R0 >= 2048 && lane == 0) {
                const int rk = R0 < 2560 ? R0 - 2048 : R0 - 2560;            // row inside k or v: KV head rk / 64, element rk % 64
                const unsigned long long dst = (R0 < 2560 ? BS_U64(l, kc) : BS_U64(l, vc)) + 2ull * (((long)(rk / BB_HD) * a.smax + p) * BB_HD + rk % BB_HD);
                *(__attribute__((address_space(1))) uint32_t*)dst = outw;
            }
        }
        if (LATE) req_bulk(l, burst_t{});
        if (wave == 4 || wave == 5) req_w2_lds(l, 0, 2 * NW, burst_t{});          // (the pieces the row pair occupied)
        // (the fourth pairs' registers carry nothing over from the previous layer: every wave asks for them again before it reads them -- said
        //  here because the two requests sit under `wave < 7` and `wave == 7`, which the register allocator does not see as exhaustive)
#pragma unroll
        for (int q = 0; q < 8; ++q) BS_DEAD(gu3[q]);
        if (wave < 7) req_gu3(l);
        // ---- attention (CUs 0..31, or all of them from BB_KMAX keys on) ----------------------------------------------------------------
        if (attn_cu) {
            dp_lf32* part = (dp_lf32*)(lds + BB_L_PART);
            if (wave == 7) {
                // q of this head (pairs 32 head ..), k_new / v_new of KV head kvh (pairs 1024 + 32 kvh .., 1280 + 32 kvh ..)
                const dp_u64* rg = a.gQ + (cu % DP_NREP) * BB_NQKV_PAIRS;
                const int i0 = lane < 32 ? 32 * head + lane : 1024 + 32 * kvh + (lane - 32);
                const int i1 = 1280 + 32 * kvh + (lane & 31);
                const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
                uint32_t v0, v1;
                for (;;) {
                    const dp_u64 x0 = dp_gran_load(rg + i0), x1 = dp_gran_load(rg + i1);
                    v0 = (uint32_t)x0; v1 = (uint32_t)x1;
                    if (__all((uint32_t)(x0 >> 32) == tagQ && (uint32_t)(x1 >> 32) == tagQ)) break;
                    if (dp_give_up(t0, ab, a.err, 0xC01u, lane)) return;
                    for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                }
                ((dp_lu32*)(lds + BB_L_Q))[lane] = v0;                       // q (words 0..31) | k_new (32..63)
                if (lane < 32) ((dp_lu32*)(lds + BB_L_Q))[64 + lane] = v1;    // v_new
                dp_flag((dp_lvu32*)(misc + BB_M_FQ), tagQ);
            } else if (!bb_wait_flag((dp_lvu32*)(misc + BB_M_FQ), tagQ, ab, a.err, 0xC02u, lane)) return;
            const uint4 qv = dp_ldq((const dp_lu4*)(lds + BB_L_Q) + e8);
            float mx = -INFINITY, lsum = 0.f, o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = 0.f;
            // this CU's keys in rounds of BB_KMAX: round 0 is the set requested a layer ahead (with the 8-way split it is the only one)
            for (int k0 = k_lo; k0 < k_hi; k0 += BB_KMAX) {
                if (k0 > k_lo) req_kv(l, k0, burst_t{});
#pragma unroll
                for (int j = 0; j < BB_KMAX / 64; ++j) {
                    const bool live = k0 + 64 * j + 8 * wave + slot < k_hi;
                    float s = bb_sum8(dot8(qv, buf[j], 0.f)) * 0.125f;
                    s = live ? s : -INFINITY;
                    const float mn = fmaxf(mx, s);
                    const float corr = (mx == -INFINITY) ? 0.f : __expf(mx - mn);
                    const float pw = live ? __expf(s - mn) : 0.f;
                    lsum = lsum * corr + pw;
                    o[0] = o[0] * corr + pw * lo2f(buf[12 + j].x); o[1] = o[1] * corr + pw * hi2f(buf[12 + j].x);
                    o[2] = o[2] * corr + pw * lo2f(buf[12 + j].y); o[3] = o[3] * corr + pw * hi2f(buf[12 + j].y);
                    o[4] = o[4] * corr + pw * lo2f(buf[12 + j].z); o[5] = o[5] * corr + pw * hi2f(buf[12 + j].z);
                    o[6] = o[6] * corr + pw * lo2f(buf[12 + j].w); o[7] = o[7] * corr + pw * hi2f(buf[12 + j].w);
                    mx = live ? mn : mx;
                }
            }
            // merge the 8 key slots of the wave (lanes sharing e8)
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                const float mo = __shfl_xor(mx, off, WAVE), lo = __shfl_xor(lsum, off, WAVE);
                const float mn = fmaxf(mx, mo);
                const float c0 = (mx == -INFINITY) ? 0.f : __expf(mx - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
                lsum = lsum * c0 + lo * c1;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float oo = __shfl_xor(o[i], off, WAVE); o[i] = o[i] * c0 + oo * c1; }
                mx = mn;
            }
            if (slot == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) part[wave * 66 + e8 * 8 + i] = o[i];
                if (e8 == 0) { part[wave * 66 + 64] = mx; part[wave * 66 + 65] = lsum; }
            }
            if (wave == 7) {
                // the step's own key (position p): one more partial with a single key
                const uint4 kn = dp_ldq((const dp_lu4*)(lds + BB_L_Q + 128) + e8), vn = dp_ldq((const dp_lu4*)(lds + BB_L_Q + 256) + e8);
                const float s = bb_sum8(dot8(qv, kn, 0.f)) * 0.125f;
                if (slot == 1) {                                             // (key range 0 carries it; the others add an empty partial)
                    part[8 * 66 + e8 * 8 + 0] = lo2f(vn.x); part[8 * 66 + e8 * 8 + 1] = hi2f(vn.x);
                    part[8 * 66 + e8 * 8 + 2] = lo2f(vn.y); part[8 * 66 + e8 * 8 + 3] = hi2f(vn.y);
                    part[8 * 66 + e8 * 8 + 4] = lo2f(vn.z); part[8 * 66 + e8 * 8 + 5] = hi2f(vn.z);
                    part[8 * 66 + e8 * 8 + 6] = lo2f(vn.w); part[8 * 66 + e8 * 8 + 7] = hi2f(vn.w);
                    if (e8 == 0) { part[8 * 66 + 64] = split == 0 ? s : -INFINITY; part[8 * 66 + 65] = split == 0 ? 1.0f : 0.f; }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(misc + BB_M_CNT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave >= 1 && wave < 7) req_buf(l, burst_t{});          // K / V are consumed: the registers take this wave's first three (gate, up) pairs
            if (wave == 0) {
                const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
                for (uint32_t spins = 1; *(dp_lvu32*)(misc + BB_M_CNT) < cnt_target; ++spins) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((spins & 255u) == 0 && dp_give_up(t0, ab, a.err, 0xC03u, lane)) return;
                }
                asm volatile("" ::: "memory");
                // lane = output dimension: fold the 9 partials in order
                float M = -INFINITY, L = 0.f, O = 0.f;
#pragma unroll
                for (int w = 0; w < 9; ++w) {
                    const float mw = part[w * 66 + 64], lw = part[w * 66 + 65], ow = part[w * 66 + lane];
                    const float mn = fmaxf(M, mw);
                    const float c0 = (M == -INFINITY) ? 0.f : __expf(M - mn), c1 = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
                    L = L * c0 + lw * c1; O = O * c0 + ow * c1; M = mn;
                }
                if (nsplit > 1) {
                    dp_u64* mine = a.gS + ((long)head * 8 + split) * 72;
                    if (split > 0) {
                        // a later key range: hand (o, m, l) to the head's CU
                        dp_gran_store(mine + lane, tagS, __float_as_uint(O));
                        if (lane == 0) dp_gran_store(mine + 64, tagS, __float_as_uint(M));
                        if (lane == 1) dp_gran_store(mine + 65, tagS, __float_as_uint(L));
                    } else {
                        // the head's CU: fold ranges 1..7 in order
                        for (int r = 1; r < 8; ++r) {
                            const dp_u64* src = a.gS + ((long)head * 8 + r) * 72;
                            const dp_u64 t1 = __builtin_amdgcn_s_memrealtime();
                            dp_u64 xo, xm, xl;
                            for (;;) {
                                xo = dp_gran_load(src + lane); xm = dp_gran_load(src + 64); xl = dp_gran_load(src + 65);
                                if (__all((uint32_t)(xo >> 32) == tagS && (uint32_t)(xm >> 32) == tagS && (uint32_t)(xl >> 32) == tagS)) break;
                                if (dp_give_up(t1, ab, a.err, 0xC06u, lane)) return;
                                for (int z = 0; z < a.poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
                            }
                            const float mw = __uint_as_float((uint32_t)xm), lw = __uint_as_float((uint32_t)xl), ow = __uint_as_float((uint32_t)xo);
                            const float mn = fmaxf(M, mw);
                            const float c0 = (M == -INFINITY) ? 0.f : __expf(M - mn), c1 = (mw == -INFINITY) ? 0.f : __expf(mw - mn);
                            L = L * c0 + lw * c1; O = O * c0 + ow * c1; M = mn;
                        }
                    }
                }
                if (split == 0) {
                    const float y = O / L;
                    const float yn = __shfl_xor(y, 1, WAVE);
                    if ((lane & 1) == 0) {
                        const uint32_t pw = pack_bf(y, yn);
#pragma unroll
                        for (int rep = 0; rep < DP_NREP; ++rep) dp_gran_store(a.gA + rep * 1024 + 32 * head + (lane >> 1), tagA, pw);
                    }
                }
            }
        }
        if (attn_cu && wave == 0) req_buf(l, burst_t{});               // (wave 0 had the fold and the publishing to do first)
        // ---- every CU: the attention vector -> o-projection rows (waves 0..3: two each) + residual -> h1 granules ---------------
        if (wave == 7) {
            uint32_t v[16];
            if (!dp_sweep<8>(a.gA + (cu % DP_NREP) * 1024, 1024, tagA, v, lane, ab, a.err, 0xC04u, a.poll_sleep)) return;
#pragma unroll
            for (int j = 0; j < 8; ++j) { ((dp_lu32*)(lds + BB_L_ATT))[2 * (j * 64 + lane)] = v[2 * j]; ((dp_lu32*)(lds + BB_L_ATT))[2 * (j * 64 + lane) + 1] = v[2 * j + 1]; }
            dp_flag((dp_lvu32*)(misc + BB_M_FATT), tagA);
            BS_STAMP(2, true);
            // the gather wave's own share of the MLP weights: only now -- its sweeps wait on vmcnt(0), and loads issued earlier would have put
            // the chip's whole stream in front of the attention hand-off
            req_buf(l, burst_t{});
            req_gu3(l);
            req_w2_lds(l, 0, 4 * NW, burst_t{});
        } else if (!bb_wait_flag((dp_lvu32*)(misc + BB_M_FATT), tagA, ab, a.err, 0xC05u, lane)) return;
        if (wave < 4) {
            const dp_lu4* xs = (const dp_lu4*)(lds + BB_L_ATT);
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint4 wo = dp_ldq((const dp_lu4*)(slice + (2 * NW + i) * 1024) + lane), wo1 = dp_ldq((const dp_lu4*)(slice + (3 * NW + i) * 1024) + lane);
                if (F8) { const uint4 x0 = dp_ldq(xs + xi(2 * i)), x1 = dp_ldq(xs + xi(2 * i + 1)); a0 = dot16_fp8(wo, x0, x1, a0); a1 = dot16_fp8(wo1, x0, x1, a1); }
                else { const uint4 x = dp_ldq(xs + i * 64 + lane); a0 = dot8(wo, x, a0); a1 = dot8(wo1, x, a1); }
            }
            a0 = wave_sum(a0) * so0; a1 = wave_sum(a1) * so1;
            const uint32_t outw = dp_resid_pair(a0, a1, hres);
            if (lane < DP_NREP) dp_gran_store(a.gH + lane * 1024 + 4 * cu + wave, tagH, outw);
            BS_STAMP(3, wave == 0);
            req_w2_lds(l, 0, 4 * NW, burst_t{});            // the whole W2 slice of this layer: the rows it displaced are consumed
        }
        // ---- the MLP: h1 -> mlp_norm -> this CU's 32 (gate, up) pairs -> 32 h values -> its 32-column slice of W2 -> partials ----
        if (wave == 7) {
            // (two half sweeps: this wave holds its 48 weight pieces in flight here, eight more loads at once do not fit the registers)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                uint32_t v[8];
                if (!dp_sweep<4>(a.gH + (cu % DP_NREP) * 1024 + hf * 512, 512, tagH, v, lane, ab, a.err, 0xC07u, a.poll_sleep)) return;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ((dp_lu32*)(lds + BL_L_H1))[2 * ((hf * 4 + j) * 64 + lane)] = v[2 * j]; ((dp_lu32*)(lds + BL_L_H1))[2 * ((hf * 4 + j) * 64 + lane) + 1] = v[2 * j + 1]; }
            }
            dp_flag((dp_lvu32*)(misc + BL_M_FH), tagH);
            BS_STAMP(4, true);
        } else if (!bb_wait_flag((dp_lvu32*)(misc + BL_M_FH), tagH, ab, a.err, 0xC08u, lane)) return;
        {
            const dp_lu4* hs = (const dp_lu4*)(lds + BL_L_H1);
            dp_lu16* hl = (dp_lu16*)(lds + ((l & 1) ? BS_L_HL2 : BL_L_HL));
            uint4 x2[4];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { x2[i] = dp_ldq(hs + xi(i)); ss += dp_chunk_ss(x2[i]); }
            ss = wave_sum(ss);
            const float r = 1.0f / sqrtf(ss / (float)BB_D + a.eps);
#pragma unroll
            for (int i = 0; i < 4; ++i) x2[i] = dp_chunk_norm(x2[i], dp_ldq((const dp_lu4*)(lds + BL_L_G2) + xi(i)), r);
            // pair i: gate row in [i * 2 NW + 0..NW-1], up row in [i * 2 NW + NW..] of buf (i < 3) / gu3 (i = 3)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float ag = 0.f, au = 0.f;
#pragma unroll
                for (int c = 0; c < NW; ++c) {
                    const uint4 wg = i < 3 ? buf[i * 2 * NW + c] : gu3[c], wu = i < 3 ? buf[i * 2 * NW + NW + c] : gu3[NW + c];
                    if (F8) { ag = dot16_fp8(wg, x2[2 * c], x2[2 * c + 1], ag); au = dot16_fp8(wu, x2[2 * c], x2[2 * c + 1], au); }
                    else { ag = dot8(wg, x2[c], ag); au = dot8(wu, x2[c], au); }
                }
                ag = wave_sum(ag); au = wave_sum(au);
                if (F8) { const long prow = 32L * cu + 4 * wave + i; ag *= BS_SCALE(l, s1, prow); au *= BS_SCALE(l, s3, prow); }
                const uint32_t hv1 = dp_swiglu(ag, au);
                if (lane == 0) hl[4 * wave + i] = (unsigned short)hv1;
            }
            BS_STAMP(5, wave == 0); BS_STAMP(6, wave == 7);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(misc + BL_M_CD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            {
                const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
                for (uint32_t spins = 1; *(dp_lvu32*)(misc + BL_M_CD) < cnt_target; ++spins)
                    if ((spins & 255u) == 0 && dp_give_up(t0, ab, a.err, 0xC09u, lane)) return;
                asm volatile("" ::: "memory");
            }
            uint4 hk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) hk[q] = dp_ldq((const dp_lu4*)hl + q);
            BS_STAMP(7, wave == 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the W2 pieces are in LDS (they were issued ~10 us ago)
            BS_STAMP(8, wave == 0); BS_STAMP(9, wave == 7);
            float pacc[4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const dp_lu4* wl = (const dp_lu4*)(lds + BL_L_W2) + (wave * 16 + rb * NW) * 64 + lane;
                if (F8) {       // two 16-byte pieces = the CU's 32 columns (the row's scale is applied by the row's owner, to the sum)
                    pacc[rb] = dot16_fp8(dp_ldq(wl), hk[0], hk[1], 0.f);
                    pacc[rb] = dot16_fp8(dp_ldq(wl + 64), hk[2], hk[3], pacc[rb]);
                } else {
                    pacc[rb] = dot8(dp_ldq(wl), hk[0], 0.f);
                    pacc[rb] = dot8(dp_ldq(wl + 64), hk[1], pacc[rb]);
                    pacc[rb] = dot8(dp_ldq(wl + 128), hk[2], pacc[rb]);
                    pacc[rb] = dot8(dp_ldq(wl + 192), hk[3], pacc[rb]);
                }
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const int row = 256 * wave + 64 * rb + lane;
                dp_gran_store(a.gP + ((long)(row >> 3) * 256 + cu) * 8 + (row & 7), tagP, __float_as_uint(pacc[rb]));
            }
            // the next layer: everything that waits for nothing of it (the W2 slice in LDS has been read, the (gate, up) registers too).  The
            // gather wave of an attention CU asks for its K / V rows here as well: they land while the partials travel; its (gate, up) pairs and
            // W2 slice wait for the attention hand-off as in layer 0.
            // (BS_DEAD: the old contents are dead on every path -- the requests sit under wave conditions, and without the statement the
            //  register allocator carries the previous layer's values around the loop for the waves that do not ask)
#pragma unroll
            for (int q = 0; q < 24; ++q) BS_DEAD(buf[q]);
            if (l + 1 < nl) {
                if (wave == 7 && attn_cu) req_kv(l + 1, k_lo, burst_t{});
                req_layer(l + 1, paced_t{});
            }
            BS_STAMP(15, wave == 0);
        }
        BS_STAMP(10, wave == 0); BS_STAMP(11, wave == 7);
        // ---- the rows' owner (gather wave): 256 partials per row in fixed order + residual -> the next layer's input row ------------------
        if (wave == 7) {
            // the next layer's norm scales: straight into LDS (no registers), in front of the sweeps, whose vmcnt(0) covers them.  (Every wave
            // of this CU is past its last read of the current scales: they all counted down BL_M_CD, after mlp_norm.)
            if (l + 1 < nl) {
                bs_gc *s1p = BS_G(l + 1, sa_norm), *s2p = BS_G(l + 1, mlp_norm);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(s1p + 1024 * i + 16u * lane), (void __attribute__((address_space(3)))*)(lds + BS_L_G1 + i * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(s2p + 1024 * i + 16u * lane), (void __attribute__((address_space(3)))*)(lds + BL_L_G2 + i * 1024), 16, 0, 0);
                }
            }
            // load j of lane l: granules 2 (64 j + l), + 1 = producer 16 j + (l >> 2), rows 2 (l & 3), 2 (l & 3) + 1 of this CU's eight;
            // four quarter sweeps (4 loads each), summed in load order
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                uint32_t v[8];
                if (!dp_sweep<4>(a.gP + (long)cu * 2048 + qt * 512, 512, tagP, v, lane, ab, a.err, 0xC0Au, a.poll_sleep)) return;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (qt == 0 && j == 0) { s0 = __uint_as_float(v[0]); s1 = __uint_as_float(v[1]); }
                    else { s0 += __uint_as_float(v[2 * j]); s1 += __uint_as_float(v[2 * j + 1]); }
                }
            }
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
            const uint32_t h1w = ((const dp_lu32*)(lds + BL_L_H1))[4 * cu + (lane & 3)];          // rows 8 cu + 2 (lane & 3), + 1 of h1
            if (F8) {
                const __attribute__((address_space(1))) float* s2p = (const __attribute__((address_space(1))) float*)BS_U64(l, s2);
                s0 *= s2p[8 * cu + 2 * (lane & 3)]; s1 *= s2p[8 * cu + 2 * (lane & 3) + 1];
            }
            const uint32_t outw = dp_resid_pair(s0, s1, h1w);
            if (l + 1 == nl) {
                if (lane < 4) *reinterpret_cast<uint32_t*>(a.h + 8 * cu + 2 * lane) = outw;
            } else {
                if (lane < 4 * DP_NREP) dp_gran_store(a.gX + (lane >> 2) * 1024 + 4 * cu + (lane & 3), tagX, outw);
                // (two half sweeps: the next layer's weights are in flight in this wave's registers)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    uint32_t v[8];
                    if (!dp_sweep<4>(a.gX + (cu % DP_NREP) * 1024 + hf * 512, 512, tagX, v, lane, ab, a.err, 0xC0Cu, a.poll_sleep)) return;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ((dp_lu32*)(lds + BS_L_X))[2 * ((hf * 4 + j) * 64 + lane)] = v[2 * j]; ((dp_lu32*)(lds + BS_L_X))[2 * ((hf * 4 + j) * 64 + lane) + 1] = v[2 * j + 1]; }
                }
                dp_flag((dp_lvu32*)(misc + BS_M_FX), tb + (uint32_t)BS_TAGS);      // = the next layer's tb
                BS_STAMP(14, true);
            }
        }
        BS_STAMP(12, wave == 7);
    }
    if (cu == 0 && threadIdx.x == 0) *a.epoch = base + (uint32_t)(BS_TAGS * nl);
}
