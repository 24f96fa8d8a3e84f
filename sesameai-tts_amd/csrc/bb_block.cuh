// Attention block of ONE backbone layer for a batch-1 decode step as ONE launch (round 2):
//   RMSNorm -> q|k|v projections -> RoPE -> KV append -> attention over keys [0, p] -> output projection + residual
// (sesameai/models.py:154-158 through torchtune's TransformerSelfAttentionLayer), CSM-1B backbone shape: d 2048, 32 heads /
// 8 KV heads of 64.  It replaces three launches of the chain (k_gemv q|k|v 4.7 us, k_attn split-K 4.0, k_gemv merge +
// o-proj 6.4, and the two gaps between them) with the hand-off machinery of the persistent depth decoder
// (dec_persist.cuh): 256 workgroups, one per CU;
//   * every wave issues ALL its weight loads at entry (a CU's slice is 6 q|k|v row pairs + 8 o-proj rows = 80 KB: it
//     fits the register file, so the whole 21 MB of the block is in flight ~0.3 us after launch),
//   * the 1536 (even, odd) q|k|v row pairs are one per wave (RoPE needs exactly that pair): 6 per CU, published as 8-byte
//     {tag, bf16 pair} granules in 8 replicas; k / v pairs also go to the KV cache for later steps,
//   * attention: CU h < 32 = head h, keys 0..p-1 from the cache (prefetched at entry: they do not depend on this step),
//     key p from the granules; 8 waves x 8 key slots, fp32 online softmax (k_attn's arithmetic), merged through LDS,
//   * the 2048 attention outputs travel as granules to every CU; wave w of CU c owns output row 8c + w.
// Bounded spins (dp_give_up) -> *err.  Any position: from BB_KMAX keys on a head's key range is split over 8 CUs (all 256
// CUs attend) and the head's CU folds their (o, m, l) partials in order -- one more hand-off, taken only where it pays.
// Batches and the fp8 decode stream use the chain.
#pragma once
#include "dec_persist.cuh"

#define BB_D 2048
#define BB_HD 64
#define BB_NH 32
#define BB_NKV 8
#define BB_KMAX 768                       // keys per round: 12 K (and V) loads per lane, 8 waves x 8 slots x 12
#define BB_NQKV_PAIRS 1536

struct BbBlockArgs {
    const bf16_t *wq, *wk, *wv, *wo, *sa_norm;
    const bf16_t* rope;                   // [max_seq][32][2]
    bf16_t* h;                            // [2048] residual stream, updated in place
    bf16_t *kc, *vc;                      // this layer's cache [8][smax][64]
    const int* pos;                       // device: position of this step
    int smax;
    float eps;
    dp_u64 *gQ, *gA;                      // [8][1536], [8][1024] granules
    dp_u64* gS;                           // [32 heads][8 key ranges][72]: (o[64], m, l) partials of the long-context mode (fp32 payloads)
    uint32_t *err, *epoch;
    int poll_sleep;
};

#define BB_L_Q 0                          // LDS bytes: q head (128) | k_new (128) | v_new (128)
#define BB_L_ATT 512                      // 4096: the attention output vector
#define BB_L_PART 4608                    // 9 partials x (64 + 2) floats
#define BB_L_MISC 7168                    // flags / counters
#define BB_LDS_BYTES 7232
#define BB_M_FQ 0
#define BB_M_FATT 1
#define BB_M_CNT 2
#define BB_M_ABORT 3

__device__ __forceinline__ float bb_sum8(float v) {           // sum over the 8 lanes of a key slot
    v += dpp_f<0xB1, 0xF>(0.f, v);
    v += dpp_f<0x4E, 0xF>(0.f, v);
    v += dpp_f<0x141, 0xF>(0.f, v);
    return v;
}

__device__ __forceinline__ bool bb_wait_flag(dp_lvu32* f, uint32_t tag, dp_lvu32* ab, uint32_t* err, uint32_t code, int lane) {
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t spins = 1; *f != tag; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}

__global__ __launch_bounds__(512) void k_bb_attn_block(const BbBlockArgs a) {
    __shared__ __attribute__((aligned(16))) char lds[BB_LDS_BYTES];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    const int lane = threadIdx.x & 63;
    dp_lu32* misc = (dp_lu32*)(lds + BB_L_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + BB_M_ABORT);
    if (threadIdx.x < 16) misc[threadIdx.x] = 0;
    // ---- everything that does not depend on the step's position: issued now, in the order it is consumed ---------------
    uint4 hv[4], g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hv[i] = reinterpret_cast<const uint4*>(a.h)[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = reinterpret_cast<const uint4*>(a.sa_norm)[i * 64 + lane];
    const int orow = 8 * cu + wave;
    const uint32_t hres2 = dp_sload32(a.h + (orow & ~1));
    const bf16_t hres = (bf16_t)((orow & 1) ? hres2 >> 16 : hres2 & 0xffffu);
    const int pair = 6 * cu + wave;                                   // waves 0..5: rows 2 pair, 2 pair + 1 of [q; k; v]
    const int R0 = 2 * (wave < 6 ? pair : 0);
    const bf16_t* wr = R0 < 2048 ? a.wq + (long)R0 * BB_D : R0 < 2560 ? a.wk + (long)(R0 - 2048) * BB_D : a.wv + (long)(R0 - 2560) * BB_D;
    uint4 w0[4], w1[4], wo[4];
    if (wave < 6) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { w0[i] = ldg16<true>(reinterpret_cast<const uint4*>(wr) + i * 64 + lane); w1[i] = ldg16<true>(reinterpret_cast<const uint4*>(wr + BB_D) + i * 64 + lane); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) wo[i] = ldg16<true>(reinterpret_cast<const uint4*>(a.wo + (long)orow * BB_D) + i * 64 + lane);
    const uint32_t base = dp_sload32(a.epoch);
    const int p = min(max((int)dp_sload32(a.pos), 0), a.smax - 1);
    // (cos, sin) of this wave's pair at position p
    const int e0 = R0 % BB_HD;
    const uint32_t cs = dp_sload32(reinterpret_cast<const uint32_t*>(a.rope) + (long)p * (BB_HD / 2) + e0 / 2);
    // attention CUs: K / V rows of keys 0..p-1 of this head's KV group.  Load j of wave w: keys 64 j + 8 w + (lane >> 3),
    // 16-byte piece lane & 7 of each 128-byte row (8 rows = one contiguous 1 KB per wave load)
    // Short contexts: head h = CU h walks all its keys (one hand-off less).  From BB_KMAX keys on the range is split over 8 CUs
    // per head (CU h + 32 r: keys [r chunk, (r + 1) chunk)), their (o, m, l) partials go to CU h, which folds them in order.
    const int nsplit = p >= BB_KMAX ? 8 : 1;
    const bool attn_cu = cu < BB_NH * nsplit;
    const int head = cu % BB_NH, split = cu / BB_NH;
    const int chunk = (p + nsplit - 1) / nsplit, k_lo = split * chunk, k_hi = min(p, k_lo + chunk);
    const int slot = lane >> 3, e8 = lane & 7;
    uint4 kr[BB_KMAX / 64], vr[BB_KMAX / 64];
    if (attn_cu) {
        const int kvh = head / (BB_NH / BB_NKV);
        const bf16_t* kb = a.kc + (long)kvh * a.smax * BB_HD;
        const bf16_t* vb = a.vc + (long)kvh * a.smax * BB_HD;
#pragma unroll
        for (int j = 0; j < BB_KMAX / 64; ++j) {
            // (slots past the last key re-read the last one: finite values, weight 0; with no key in range the round loop does not run)
            const int key = min(k_lo + 64 * j + 8 * wave + slot, max(k_hi - 1, 0));
            kr[j] = *reinterpret_cast<const uint4*>(kb + (long)key * BB_HD + e8 * 8);
            vr[j] = *reinterpret_cast<const uint4*>(vb + (long)key * BB_HD + e8 * 8);
        }
    }
    // (misc zeroed.  A bare s_barrier: __syncthreads() would also wait for every load issued above)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const uint32_t tagQ = base + 1u, tagA = base + 2u, tagS = base + 3u;

    // ---- RMSNorm of the whole row, per wave, in registers (chunk i * 64 + lane = elements 8 (i * 64 + lane) ..) ---------
    uint4 xn[4];
    {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += dp_chunk_ss(hv[i]);
        ss = wave_sum(ss);
        const float r = 1.0f / sqrtf(ss / (float)BB_D + a.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) xn[i] = dp_chunk_norm(hv[i], g[i], r);
    }
    // ---- q | k | v pair of this wave -> RoPE -> granule (8 replicas) + KV cache ----------------------------------------
    if (wave < 6) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a0 = dot8(w0[i], xn[i], a0); a1 = dot8(w1[i], xn[i], a1); }
        a0 = wave_sum(a0); a1 = wave_sum(a1);
        const uint32_t outw = dp_rope_pair(a0, a1, cs, R0 < 2560);
        if (lane < DP_NREP) dp_gran_store(a.gQ + lane * BB_NQKV_PAIRS + pair, tagQ, outw);
        if (R0 >= 2048 && lane == 0) {
            const int rk = R0 < 2560 ? R0 - 2048 : R0 - 2560;            // row inside k or v: KV head rk / 64, element rk % 64
            bf16_t* dst = (R0 < 2560 ? a.kc : a.vc) + ((long)(rk / BB_HD) * a.smax + p) * BB_HD + rk % BB_HD;
            *reinterpret_cast<uint32_t*>(dst) = outw;
        }
    }
    // ---- attention (CUs 0..31) -----------------------------------------------------------------------------------------
    if (attn_cu) {
        const int kvh = head / (BB_NH / BB_NKV);
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
        float mx = -INFINITY, l = 0.f, o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
        // this CU's keys in rounds of BB_KMAX: round 0 is the set prefetched at entry (with the 8-way split it is the only one)
        for (int k0 = k_lo; k0 < k_hi; k0 += BB_KMAX) {
            if (k0 > k_lo) {
                const bf16_t* kb = a.kc + (long)kvh * a.smax * BB_HD;
                const bf16_t* vb = a.vc + (long)kvh * a.smax * BB_HD;
#pragma unroll
                for (int j = 0; j < BB_KMAX / 64; ++j) {
                    const int key = min(k0 + 64 * j + 8 * wave + slot, k_hi - 1);
                    kr[j] = *reinterpret_cast<const uint4*>(kb + (long)key * BB_HD + e8 * 8);
                    vr[j] = *reinterpret_cast<const uint4*>(vb + (long)key * BB_HD + e8 * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < BB_KMAX / 64; ++j) {
                const bool live = k0 + 64 * j + 8 * wave + slot < k_hi;
                float s = bb_sum8(dot8(qv, kr[j], 0.f)) * 0.125f;
                s = live ? s : -INFINITY;
                const float mn = fmaxf(mx, s);
                const float corr = (mx == -INFINITY) ? 0.f : __expf(mx - mn);
                const float pw = live ? __expf(s - mn) : 0.f;
                l = l * corr + pw;
                o[0] = o[0] * corr + pw * lo2f(vr[j].x); o[1] = o[1] * corr + pw * hi2f(vr[j].x);
                o[2] = o[2] * corr + pw * lo2f(vr[j].y); o[3] = o[3] * corr + pw * hi2f(vr[j].y);
                o[4] = o[4] * corr + pw * lo2f(vr[j].z); o[5] = o[5] * corr + pw * hi2f(vr[j].z);
                o[6] = o[6] * corr + pw * lo2f(vr[j].w); o[7] = o[7] * corr + pw * hi2f(vr[j].w);
                mx = live ? mn : mx;
            }
        }
        // merge the 8 key slots of the wave (lanes sharing e8)
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float mo = __shfl_xor(mx, off, WAVE), lo = __shfl_xor(l, off, WAVE);
            const float mn = fmaxf(mx, mo);
            const float c0 = (mx == -INFINITY) ? 0.f : __expf(mx - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
            l = l * c0 + lo * c1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float oo = __shfl_xor(o[i], off, WAVE); o[i] = o[i] * c0 + oo * c1; }
            mx = mn;
        }
        if (slot == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) part[wave * 66 + e8 * 8 + i] = o[i];
            if (e8 == 0) { part[wave * 66 + 64] = mx; part[wave * 66 + 65] = l; }
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
        if (wave == 0) {
            const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
            for (uint32_t spins = 1; *(dp_lvu32*)(misc + BB_M_CNT) < 8u; ++spins) {
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
    // ---- every CU: the attention vector -> o-projection row 8 cu + wave + residual ----------------------------------------
    if (wave == 7) {
        uint32_t v[16];
        if (!dp_sweep<8>(a.gA + (cu % DP_NREP) * 1024, 1024, tagA, v, lane, ab, a.err, 0xC04u, a.poll_sleep)) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ((dp_lu32*)(lds + BB_L_ATT))[2 * (j * 64 + lane)] = v[2 * j]; ((dp_lu32*)(lds + BB_L_ATT))[2 * (j * 64 + lane) + 1] = v[2 * j + 1]; }
        dp_flag((dp_lvu32*)(misc + BB_M_FATT), tagA);
    } else if (!bb_wait_flag((dp_lvu32*)(misc + BB_M_FATT), tagA, ab, a.err, 0xC05u, lane)) return;
    {
        const dp_lu4* xs = (const dp_lu4*)(lds + BB_L_ATT);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = dot8(wo[i], dp_ldq(xs + i * 64 + lane), acc);
        acc = wave_sum(acc);
        if (lane == 0) {
#pragma clang fp contract(off)
            a.h[orow] = f2bf(round_bf(acc) + bf2f(hres));
        }
    }
    if (cu == 0 && threadIdx.x == 0) *a.epoch = base + 4u;
}


// ---------------------------------------------------------------------------------------------------------------
// The WHOLE backbone layer of a batch-1 decode step as one launch: the attention block above, then
//   h1 (granules) -> mlp_norm -> gate / up -> SiLU * up -> down projection (split over the CUs' column slices) + residual.
// A CU's share of the MLP weights -- 32 (gate, up) pairs and a 32-column slice of W2, 384 KB -- fits its register file next
// to the attention block's 80 KB, so waves 0..6 issue those loads at entry as well: the 100 MB of the MLP stream from HBM
// while the attention block runs its hand-offs (in the chain HBM idles through them, then two more launches stream the MLP).
// An attention CU keeps its K / V rows in the registers of its first three pairs until its attention is done.  The gather wave
// (7) loads its share only after the attention hand-off: its sweeps wait on vmcnt(0).
// ---------------------------------------------------------------------------------------------------------------
struct BbLayerArgs {
    const bf16_t *wq, *wk, *wv, *wo, *sa_norm;
    const bf16_t* rope;
    bf16_t* h;
    bf16_t *kc, *vc;
    const int* pos;
    int smax;
    float eps;
    dp_u64 *gQ, *gA;
    dp_u64* gS;
    uint32_t *err, *epoch;
    int poll_sleep;
    const bf16_t *w1, *w3, *mlp_norm;     // gate / up [8192][2048] row-major
    const uint4* w2t;                     // W2 re-tiled [256 cu][4 k chunks][2048 rows] 16-byte pieces (k_bb_retile_w2)
    dp_u64 *gH, *gP;                      // [8][1024] h1 granules; [256 owners][256 producers][8 rows] fp32 partials
    dp_u64* stamps;                       // timeline build: 16 s_memrealtime stamps of workgroup 100 (layer 8)
    const float *sq, *sk, *sv, *so, *s1, *s3, *s2;   // F8: per-output-row scales of the e4m3 matrices (wq.. then point to bytes, w2t to k_bb_retile_w2_fp8's pieces)
};
#ifdef DP_TIMELINE
#define BL_STAMP(i_, cond_) do { if (a.stamps != nullptr && cu == 100 && lane == 0 && (cond_)) a.stamps[i_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BL_STAMP(i_, cond_) do { } while (0)
#endif
#define BL_L_H1 7232                      // 4096: h1 (the residual stream after the attention block)
#define BL_L_HL 11328                     // 64: this CU's 32 h values
#define BL_L_G2 11392                     // 4096: mlp_norm's scale
#define BL_L_W2 15488                     // 131072: the CU's W2 slice, [wave][row block][k chunk][64 lanes] 16-byte pieces
#define BL_LDS_BYTES (15488 + 131072)
#define BL_M_FH 4
#define BL_M_CD 5

// e4m3 W2 [2048 rows][8192] bytes -> [256 cu][2 k chunks][2048 rows] 16-byte pieces (workgroup cu's 32 columns, 16 at a time)
__global__ void k_bb_retile_w2_fp8(const uint8_t* w2, uint4* w2t) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 256L * 2 * BB_D) return;
    const int n = (int)(i % BB_D), q = (int)((i / BB_D) % 2), c = (int)(i / (2 * BB_D));
    w2t[i] = *reinterpret_cast<const uint4*>(w2 + (long)n * 8192 + c * 32 + q * 16);
}

__global__ void k_bb_retile_w2(const bf16_t* w2, uint4* w2t) {       // w2 [2048 rows][8192]
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 256L * 4 * BB_D) return;
    const int n = (int)(i % BB_D), q = (int)((i / BB_D) % 4), c = (int)(i / (4 * BB_D));
    w2t[i] = *reinterpret_cast<const uint4*>(w2 + (long)n * 8192 + c * 32 + q * 8);
}

// F8: the seven weight matrices arrive as OCP-e4m3 bytes with one power-of-two fp32 scale per output row (BASELINE config 5): a lane's
// 16-byte piece holds 16 k values instead of 8, so the activations are kept in the matching order (piece q of a lane = elements
// 16 ((q >> 1) * 64 + lane) + 8 (q & 1) ..), the dot products run through v_cvt_scalef32_pk_bf16_fp8 + v_dot2c_f32_bf16 (dot16_fp8) and the
// row scale multiplies the fp32 sum (exact).  61 MB per layer instead of 122.
template <bool F8>
__global__ __launch_bounds__(512) void k_bb_layer(const BbLayerArgs a) {
    constexpr int NW = F8 ? 2 : 4;                                      // 16-byte weight pieces per lane per 2048-long row
    extern __shared__ __attribute__((aligned(16))) char lds[];          // BL_LDS_BYTES (dynamic: 143 KB)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    const int lane = threadIdx.x & 63;
    auto xi = [&](int q) -> int { return F8 ? 2 * ((q >> 1) * 64 + lane) + (q & 1) : q * 64 + lane; };      // index of a lane's q-th 16-byte activation piece
    dp_lu32* misc = (dp_lu32*)(lds + BB_L_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + BB_M_ABORT);
    if (threadIdx.x < 16) misc[threadIdx.x] = 0;
    BL_STAMP(0, wave == 0);
    // ---- everything that does not depend on the step's position: issued now, in the order it is consumed ---------------
    uint4 hv[4], g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) hv[i] = reinterpret_cast<const uint4*>(a.h)[xi(i)];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[i] = reinterpret_cast<const uint4*>(a.sa_norm)[xi(i)];
    const int orow = 8 * cu + 2 * (wave & 3);                         // waves 0..3: output rows orow, orow + 1
    const uint32_t hres = dp_sload32(a.h + orow);
    const int pair = 6 * cu + wave;                                   // waves 0..5: rows 2 pair, 2 pair + 1 of [q; k; v]
    const int R0 = 2 * (wave < 6 ? pair : 0);
    constexpr int EB = F8 ? 1 : 2;                                      // bytes per weight
    const int rq = R0 < 2048 ? R0 : R0 < 2560 ? R0 - 2048 : R0 - 2560;  // row inside its matrix
    // (this exact form -- a select between three complete addresses.  Selecting the base first and adding the row offset after it, the
    //  same addresses, made the launch 2 us slower: 32.2 against 29.9 us per layer, A/B on one box, round 3)
    const char* wr = R0 < 2048 ? (const char*)a.wq + (long)R0 * BB_D * EB : R0 < 2560 ? (const char*)a.wk + (long)(R0 - 2048) * BB_D * EB
                                                                                      : (const char*)a.wv + (long)(R0 - 2560) * BB_D * EB;
    float sc0 = 1.f, sc1 = 1.f, so0 = 1.f, so1 = 1.f;                   // F8: the rows' scales
    if (F8 && wave < 6) { const float* sp = R0 < 2048 ? a.sq : R0 < 2560 ? a.sk : a.sv; sc0 = sp[rq]; sc1 = sp[rq + 1]; }
    if (F8 && wave < 4) { so0 = a.so[orow]; so1 = a.so[orow + 1]; }
    uint4 w0[NW], w1[NW], wo[NW], wo1[NW];
    if (wave < 6) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { w0[i] = ldg16<true>(reinterpret_cast<const uint4*>(wr) + i * 64 + lane); w1[i] = ldg16<true>(reinterpret_cast<const uint4*>(wr + BB_D * EB) + i * 64 + lane); }
    }
    if (wave < 4) {                                     // o-projection rows: BEFORE the MLP weights in the queue (issued after them these 32 KB
                                                        // arrived last, ~16 us in, and held the whole layer up)
#pragma unroll
        for (int i = 0; i < NW; ++i) { wo[i] = ldg16<true>(reinterpret_cast<const uint4*>((const char*)a.wo + (long)orow * BB_D * EB) + i * 64 + lane); wo1[i] = ldg16<true>(reinterpret_cast<const uint4*>((const char*)a.wo + (long)(orow + 1) * BB_D * EB) + i * 64 + lane); }
    }
    const uint32_t base = dp_sload32(a.epoch);
    const int p = min(max((int)dp_sload32(a.pos), 0), a.smax - 1);
    // (cos, sin) of this wave's pair at position p
    const int e0 = R0 % BB_HD;
    const uint32_t cs = dp_sload32(reinterpret_cast<const uint32_t*>(a.rope) + (long)p * (BB_HD / 2) + e0 / 2);
    // attention CUs: K / V rows of keys 0..p-1 of this head's KV group.  Load j of wave w: keys 64 j + 8 w + (lane >> 3),
    // 16-byte piece lane & 7 of each 128-byte row (8 rows = one contiguous 1 KB per wave load)
    // Short contexts: head h = CU h walks all its keys (one hand-off less).  From BB_KMAX keys on the range is split over 8 CUs
    // per head (CU h + 32 r: keys [r chunk, (r + 1) chunk)), their (o, m, l) partials go to CU h, which folds them in order.
    const int nsplit = p >= BB_KMAX ? 8 : 1;
    const bool attn_cu = cu < BB_NH * nsplit;
    const int head = cu % BB_NH, split = cu / BB_NH;
    const int chunk = (p + nsplit - 1) / nsplit, k_lo = split * chunk, k_hi = min(p, k_lo + chunk);
    const int slot = lane >> 3, e8 = lane & 7;
    // buf: K / V rows on an attention CU (until its attention is done), else the first three (gate, up) pairs of this wave
    uint4 buf[24], gu3[8];
    auto load_gu = [&](int i, int gq, int c) -> uint4 {                 // pair i of this wave, gq 0 = gate row, 1 = up row, chunk c (F8: c < 2)
        const long prow = 32L * cu + 4 * wave + i;
        return ldg16<true>(reinterpret_cast<const uint4*>((const char*)(gq ? a.w3 : a.w1) + prow * BB_D * EB) + c * 64 + lane);
    };
    // all (gate, up) pieces of pairs 0..2 of this wave into buf, in the order (pair, gate | up, chunk)
#define load_buf() do { _Pragma("unroll") for (int q = 0; q < 6 * NW; ++q) buf[q] = load_gu(q / (2 * NW), (q / NW) & 1, q % NW); } while (0)
    if (attn_cu) {
        const int kvh = head / (BB_NH / BB_NKV);
        const bf16_t* kb = a.kc + (long)kvh * a.smax * BB_HD;
        const bf16_t* vb = a.vc + (long)kvh * a.smax * BB_HD;
#pragma unroll
        for (int j = 0; j < BB_KMAX / 64; ++j) {
            // (slots past the last key re-read the last one: finite values, weight 0; with no key in range the round loop does not run)
            const int key = min(k_lo + 64 * j + 8 * wave + slot, max(k_hi - 1, 0));
            buf[j] = *reinterpret_cast<const uint4*>(kb + (long)key * BB_HD + e8 * 8);
            buf[12 + j] = *reinterpret_cast<const uint4*>(vb + (long)key * BB_HD + e8 * 8);
        }
    } else if (wave < 7) load_buf();
    // (misc zeroed.  A bare s_barrier: __syncthreads() would also wait for every load issued above)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const uint32_t tagQ = base + 1u, tagA = base + 2u, tagS = base + 3u, tagH = base + 4u, tagP = base + 5u;

    // ---- RMSNorm of the whole row, per wave, in registers (chunk i * 64 + lane = elements 8 (i * 64 + lane) ..) ---------
    uint4 xn[4];
    {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) ss += dp_chunk_ss(hv[i]);
        ss = wave_sum(ss);
        const float r = 1.0f / sqrtf(ss / (float)BB_D + a.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) xn[i] = dp_chunk_norm(hv[i], g[i], r);
    }
    // ---- q | k | v pair of this wave -> RoPE -> granule (8 replicas) + KV cache ----------------------------------------
    if (wave < 6) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (F8) { a0 = dot16_fp8(w0[i], xn[2 * i], xn[2 * i + 1], a0); a1 = dot16_fp8(w1[i], xn[2 * i], xn[2 * i + 1], a1); }
            else { a0 = dot8(w0[i], xn[i], a0); a1 = dot8(w1[i], xn[i], a1); }
        }
        a0 = wave_sum(a0) * sc0; a1 = wave_sum(a1) * sc1;
        const uint32_t outw = dp_rope_pair(a0, a1, cs, R0 < 2560);
        if (lane < DP_NREP) dp_gran_store(a.gQ + lane * BB_NQKV_PAIRS + pair, tagQ, outw);
        BL_STAMP(1, wave == 0);
        if (R0 >= 2048 && lane == 0) {
            const int rk = R0 < 2560 ? R0 - 2048 : R0 - 2560;            // row inside k or v: KV head rk / 64, element rk % 64
            bf16_t* dst = (R0 < 2560 ? a.kc : a.vc) + ((long)(rk / BB_HD) * a.smax + p) * BB_HD + rk % BB_HD;
            *reinterpret_cast<uint32_t*>(dst) = outw;
        }
    }
    // W2 slice of this wave (16 KB): straight into LDS (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16 l), no
    // registers -- with it in VGPRs the kernel spilled 41 dwords per lane.  Piece (row block rb, k chunk kc) at index rb * 4 + kc.
    auto load_w2_lds = [&]() {
#pragma unroll
        for (int q = 0; q < 4 * NW; ++q)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(a.w2t + ((long)cu * NW + (q % NW)) * BB_D + 256 * wave + 64 * (q / NW) + lane),
                                             (void __attribute__((address_space(3)))*)(lds + BL_L_W2 + ((wave * 16 + q) * 64) * 16), 16, 0, 0);
    };
    if (wave < 7) {
#pragma unroll
        for (int q = 0; q < 2 * NW; ++q) gu3[q] = load_gu(3, q / NW, q % NW);
        load_w2_lds();
    }
    if (wave == 6) {                                    // mlp_norm's scale -> LDS (16 registers per wave otherwise, held for 10 us)
#pragma unroll
        for (int i = 0; i < 4; ++i) dp_stq((dp_lu4*)(lds + BL_L_G2) + i * 64 + lane, reinterpret_cast<const uint4*>(a.mlp_norm)[i * 64 + lane]);
    }
    // ---- attention (CUs 0..31) -----------------------------------------------------------------------------------------
    if (attn_cu) {
        const int kvh = head / (BB_NH / BB_NKV);
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
        float mx = -INFINITY, l = 0.f, o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
        // this CU's keys in rounds of BB_KMAX: round 0 is the set prefetched at entry (with the 8-way split it is the only one)
        for (int k0 = k_lo; k0 < k_hi; k0 += BB_KMAX) {
            if (k0 > k_lo) {
                const bf16_t* kb = a.kc + (long)kvh * a.smax * BB_HD;
                const bf16_t* vb = a.vc + (long)kvh * a.smax * BB_HD;
#pragma unroll
                for (int j = 0; j < BB_KMAX / 64; ++j) {
                    const int key = min(k0 + 64 * j + 8 * wave + slot, k_hi - 1);
                    buf[j] = *reinterpret_cast<const uint4*>(kb + (long)key * BB_HD + e8 * 8);
                    buf[12 + j] = *reinterpret_cast<const uint4*>(vb + (long)key * BB_HD + e8 * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < BB_KMAX / 64; ++j) {
                const bool live = k0 + 64 * j + 8 * wave + slot < k_hi;
                float s = bb_sum8(dot8(qv, buf[j], 0.f)) * 0.125f;
                s = live ? s : -INFINITY;
                const float mn = fmaxf(mx, s);
                const float corr = (mx == -INFINITY) ? 0.f : __expf(mx - mn);
                const float pw = live ? __expf(s - mn) : 0.f;
                l = l * corr + pw;
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
            const float mo = __shfl_xor(mx, off, WAVE), lo = __shfl_xor(l, off, WAVE);
            const float mn = fmaxf(mx, mo);
            const float c0 = (mx == -INFINITY) ? 0.f : __expf(mx - mn), c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
            l = l * c0 + lo * c1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float oo = __shfl_xor(o[i], off, WAVE); o[i] = o[i] * c0 + oo * c1; }
            mx = mn;
        }
        if (slot == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) part[wave * 66 + e8 * 8 + i] = o[i];
            if (e8 == 0) { part[wave * 66 + 64] = mx; part[wave * 66 + 65] = l; }
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
        if (wave >= 1 && wave < 7) load_buf();          // K / V are consumed: the registers take this wave's first three (gate, up) pairs
        if (wave == 0) {
            const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
            for (uint32_t spins = 1; *(dp_lvu32*)(misc + BB_M_CNT) < 8u; ++spins) {
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
    if (attn_cu && wave == 0) load_buf();               // (wave 0 had the fold and the publishing to do first)
    // ---- every CU: the attention vector -> o-projection rows (waves 0..3: two each) + residual -> h1 granules ---------------
    if (wave == 7) {
        uint32_t v[16];
        if (!dp_sweep<8>(a.gA + (cu % DP_NREP) * 1024, 1024, tagA, v, lane, ab, a.err, 0xC04u, a.poll_sleep)) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ((dp_lu32*)(lds + BB_L_ATT))[2 * (j * 64 + lane)] = v[2 * j]; ((dp_lu32*)(lds + BB_L_ATT))[2 * (j * 64 + lane) + 1] = v[2 * j + 1]; }
        dp_flag((dp_lvu32*)(misc + BB_M_FATT), tagA);
        BL_STAMP(2, true);
        // the gather wave's own share of the MLP weights: only now -- its sweeps wait on vmcnt(0), and loads issued at entry
        // would have put the whole 100 MB stream of the chip in front of the attention hand-off
        load_buf();
#pragma unroll
        for (int q = 0; q < 2 * NW; ++q) gu3[q] = load_gu(3, q / NW, q % NW);
        load_w2_lds();
    } else if (!bb_wait_flag((dp_lvu32*)(misc + BB_M_FATT), tagA, ab, a.err, 0xC05u, lane)) return;
    if (wave < 4) {
        const dp_lu4* xs = (const dp_lu4*)(lds + BB_L_ATT);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (F8) { const uint4 x0 = dp_ldq(xs + xi(2 * i)), x1 = dp_ldq(xs + xi(2 * i + 1)); a0 = dot16_fp8(wo[i], x0, x1, a0); a1 = dot16_fp8(wo1[i], x0, x1, a1); }
            else { const uint4 x = dp_ldq(xs + i * 64 + lane); a0 = dot8(wo[i], x, a0); a1 = dot8(wo1[i], x, a1); }
        }
        a0 = wave_sum(a0) * so0; a1 = wave_sum(a1) * so1;
        const uint32_t outw = dp_resid_pair(a0, a1, hres);
        if (lane < DP_NREP) dp_gran_store(a.gH + lane * 1024 + 4 * cu + wave, tagH, outw);
        BL_STAMP(3, wave == 0);
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
        BL_STAMP(4, true);
    } else if (!bb_wait_flag((dp_lvu32*)(misc + BL_M_FH), tagH, ab, a.err, 0xC08u, lane)) return;
    {
        const dp_lu4* hs = (const dp_lu4*)(lds + BL_L_H1);
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
            if (F8) { const long prow = 32L * cu + 4 * wave + i; ag *= a.s1[prow]; au *= a.s3[prow]; }
            const uint32_t hv1 = dp_swiglu(ag, au);
            if (lane == 0) ((dp_lu16*)(lds + BL_L_HL))[4 * wave + i] = (unsigned short)hv1;
        }
        BL_STAMP(5, wave == 0); BL_STAMP(6, wave == 7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(misc + BL_M_CD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        {
            const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
            for (uint32_t spins = 1; *(dp_lvu32*)(misc + BL_M_CD) < 8u; ++spins)
                if ((spins & 255u) == 0 && dp_give_up(t0, ab, a.err, 0xC09u, lane)) return;
            asm volatile("" ::: "memory");
        }
        uint4 hk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) hk[q] = dp_ldq((const dp_lu4*)(lds + BL_L_HL) + q);
        BL_STAMP(7, wave == 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the W2 pieces are in LDS (they were issued ~10 us ago)
        BL_STAMP(8, wave == 0); BL_STAMP(9, wave == 7);
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const int row = 256 * wave + 64 * rb + lane;
            const dp_lu4* wl = (const dp_lu4*)(lds + BL_L_W2) + (wave * 16 + rb * NW) * 64 + lane;
            float pacc;
            if (F8) {       // two 16-byte pieces = the CU's 32 columns (the row's scale is applied by the row's owner, to the sum)
                pacc = dot16_fp8(dp_ldq(wl), hk[0], hk[1], 0.f);
                pacc = dot16_fp8(dp_ldq(wl + 64), hk[2], hk[3], pacc);
            } else {
                pacc = dot8(dp_ldq(wl), hk[0], 0.f);
                pacc = dot8(dp_ldq(wl + 64), hk[1], pacc);
                pacc = dot8(dp_ldq(wl + 128), hk[2], pacc);
                pacc = dot8(dp_ldq(wl + 192), hk[3], pacc);
            }
            dp_gran_store(a.gP + ((long)(row >> 3) * 256 + cu) * 8 + (row & 7), tagP, __float_as_uint(pacc));
        }
    }
    BL_STAMP(10, wave == 0); BL_STAMP(11, wave == 7);
    // ---- the rows' owner (gather wave): 256 partials per row in fixed order + residual -> h ------------------------------------
    if (wave == 7) {
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
        if (lane < 4) {
            const uint32_t h1w = ((const dp_lu32*)(lds + BL_L_H1))[4 * cu + lane];          // rows 8 cu + 2 lane, + 1 of h1
            if (F8) { s0 *= a.s2[8 * cu + 2 * lane]; s1 *= a.s2[8 * cu + 2 * lane + 1]; }
            *reinterpret_cast<uint32_t*>(a.h + 8 * cu + 2 * lane) = dp_resid_pair(s0, s1, h1w);
        }
    }
    BL_STAMP(12, wave == 7);
    if (cu == 0 && threadIdx.x == 0) *a.epoch = base + 8u;
}
