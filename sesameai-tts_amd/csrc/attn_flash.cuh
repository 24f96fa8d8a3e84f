// Prompt-prefill attention on the matrix cores (head_dim 64, GQA group of 4): the attention of every PROMPT row.
//
// attn.cuh gives each (row, kv head) its own block that walks keys [0, pos]: right for decode steps, but over a
// prompt every row re-reads its whole key range from L2 (1.8 GB per layer at 1,334 rows, as long as all the
// prompt's GEMMs together).  Here a block owns 32 consecutive rows of one sequence for one KV head; its four waves
// are the four query heads of the group and share each 32-key K/V tile through LDS.
//
// Per wave, per key tile (all on v_mfma_f32_32x32x16_bf16):
//   S^T[key][query] = K . Q^T            A = K fragments (ds_read_b128 from a swizzled row-major tile), B = Q (registers)
//     -> the 32 scores of a query sit in ONE lane pair (lane r: 16 keys, lane r+32: the other 16), so the online
//        softmax is in-register plus one half-swap; scores, max, exp and the running sum stay fp32
//   O^T[d][query]  += V^T . P^T          B = P^T = the accumulator registers of S^T, rounded to bf16, with NO lane
//        movement (an accumulator tile is a valid operand of an MFMA that sums over its row index); A = V^T by
//        ds_read_b64_tr_b16 transposed reads of the row-major V tile
// This is the arithmetic of torch's CPU flash kernel that the oracle follows (fp32 scores / softmax, probabilities
// rounded to bf16 for the P.V product).  A row's result depends only on its own q, position and keys -- not on which
// rows share its tile or how many key tiles the block walks (a fully masked tile is an exact no-op) -- so prompt rows
// keep their bits however a prompt is cut into prefill calls (prefix-KV reuse).
//
// Round 5 experiment, kept behind -DAF_NG=2 (shipped: AF_NG = 1, the walk above) -- the key range of a block split over NG wave groups.
// A block's walk is one dependent chain per wave (~1.26 us per 32-key tile: softmax VALU + MFMA issue of a single wave per SIMD), and the
// launch lasts as long as its LONGEST block: 42 tiles = 53 us per layer at 1,334 rows while the causal triangle idles half the chip.
// With NG > 1, group g of a block (4 waves = the 4 query heads, its own double-buffered K/V tiles in LDS) walks the 64-key tile PAIRS g,
// g + NG, ... -- a fixed function of the absolute key index, so a row's arithmetic still depends on nothing but its own q, position and
// keys -- and group 0 folds the groups' (m, l, O) states in group order at the end.  A group that has not met a visible key carries
// m = -inf, l = 0, O = 0; the guards below make its updates and the fold exact.  Measured (profiles/r05/flash_key_groups_ab.txt): NG = 2
// is -4.8 % on a 1,334-row prefill and +1..2 % on 190-row and 32-prompt prefills (80 KB of LDS, 512 threads, 159 VGPRs: ONE block per CU
// where three of the 256-thread blocks fit), and one long-context logit lands an ulp further from the oracle's.  Not worth a second
// canonical order of the prompt attention.
#pragma once
#include "attn.cuh"
#include "mm.cuh"

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

#define AF_KLD 64            // K tile row = 64 d (128 B), 16-byte segments XOR-swizzled like gemm128
#define AF_VLD 96            // V tile row stride 192 B: the 4 rows of a transposed read land on 4 distinct bank quarters

#ifndef AF_NG
#define AF_NG 1               // key groups per block (256 threads each).  Measured (profiles/r05/flash_key_groups_ab.txt): AF_NG = 2 takes a 1,334-row
                              // prefill from 5.99 to 5.70 ms (-4.8 %) but costs 1-2 % at 190 rows and at 32 prompts (one 512-thread block per CU instead of
                              // three 256-thread ones) and moves a long-context logit by one more ulp: not shipped; -DAF_NG=2 builds it for the A/B
#endif
#define AF_GRP_BYTES (2 * 64 * AF_KLD * 2 + 2 * 64 * AF_VLD * 2)      // one group's K + V double buffers: 40 KB

template <int HD, int NG>
__global__ __launch_bounds__(256 * NG) void k_attn_flash(const AttnArgs a) {
    static_assert(HD == 64, "k_attn_flash: head_dim 64");
    // (round 4: a buffer holds TWO 32-key tiles -- one global round trip, one LDS write and one barrier per 64 keys; the tiles are
    //  still consumed one after the other with the same online-softmax update, so the bits are those of the 32-key walk)
    __shared__ __align__(16) char af_lds[NG * AF_GRP_BYTES];
    const int grp = threadIdx.x >> 8;                                  // key group of this wave
    bf16_t (*Ks)[64 * AF_KLD] = reinterpret_cast<bf16_t (*)[64 * AF_KLD]>(af_lds + grp * AF_GRP_BYTES);
    bf16_t (*Vs)[64 * AF_VLD] = reinterpret_cast<bf16_t (*)[64 * AF_VLD]>(af_lds + grp * AF_GRP_BYTES + 2 * 64 * AF_KLD * 2);
    const int tid = threadIdx.x & 255, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    // block order (speed only): the KV heads of one row tile next to each other, a sequence's row tiles from the LAST one down -- the
    // blocks with the longest key walks are dispatched first and the short ones fill in behind them (one 512-thread block per CU: at
    // 1,334 rows 336 blocks meet 256 CUs, and in ascending order the 80 longest walks would start after everyone else had finished)
    const int G = a.H / a.KV;
    const int groups = (a.rows_per_seq + 31) / 32;
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int kvh = lin % a.KV, xi = lin / a.KV;
    const int b = xi / groups, r0 = (groups - 1 - xi % groups) * 32;
    const int nr = min(32, a.rows_per_seq - r0);                       // valid query rows of this tile
    const int m_base = b * a.rows_per_seq + r0;
    const int mq = m_base + min(r, nr - 1);                            // this lane's query row (padding repeats the last)
    const int pq = min(max(a.pos[mq], 0), a.smax - 1);
    const int pmax = (int)wave_max((float)pq);                          // same for the 4 waves (same 32 queries)
    const int npairs = pmax / 64 + 1;                                   // pairs of 32-key tiles
    const int nit = (npairs + NG - 1) / NG;                             // iterations: group g takes pairs g, g + NG, ... (one past the end = a null pair)
    const bf16_t* kb = a.kcache + ((long)b * a.KV + kvh) * a.smax * HD;
    const bf16_t* vb = a.vcache + ((long)b * a.KV + kvh) * a.smax * HD;

    // staging: thread t moves segment (t & 7) of key row (t >> 3) of both tiles
    const int skey = tid >> 3, sseg = tid & 7;
    u32x4_t kreg[2], vreg[2];
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    // (t = index of a PAIR of 32-key tiles; rows skey and skey + 32 of the pair share (row >> 1) & 7, so one swizzled segment serves both)
#define AF_GLOAD(t)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                     \
        const int j = (t) * 64 + 32 * u + skey;                                                         \
        const long off = (long)min(j, pmax) * HD + sseg * 8;                                            \
        kreg[u] = *reinterpret_cast<const u32x4_t*>(kb + off);                                          \
        vreg[u] = *reinterpret_cast<const u32x4_t*>(vb + off);                                          \
        if (j > pmax) { kreg[u] = zero4; vreg[u] = zero4; }   /* never-written cache rows may hold NaN bit patterns */ \
    }
#define AF_LWRITE(buf)                                                                                  \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                     \
        *reinterpret_cast<u32x4_t*>(&Ks[buf][(32 * u + skey) * AF_KLD + ((sseg ^ ((skey >> 1) & 7)) * 8)]) = kreg[u]; \
        *reinterpret_cast<u32x4_t*>(&Vs[buf][(32 * u + skey) * AF_VLD + sseg * 8]) = vreg[u];           \
    }

    for (int g = wave; g < G; g += 4) {
        const int hq = kvh * G + g;
        // Q fragments (B operand): lane (r = query, h) holds q[32h + 8s .. +8] of its row for k-step s = 0..3
        u32x4_t qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
            qf[s] = *reinterpret_cast<const u32x4_t*>(a.q + ((long)mq * a.H + hq) * HD + 32 * h + 8 * s);
        f32x16_t o0, o1;                                               // O^T rows d = 0..31 / 32..63, column = query
#pragma unroll
        for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
        float mrun = -INFINITY, lrun = 0.f;

        __syncthreads();                                               // previous head's readers are done with LDS
        AF_GLOAD(grp)
        AF_LWRITE(0)
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const int tp = it * NG + grp;                              // this group's pair of this iteration (>= npairs: a null pair)
            AF_GLOAD(tp + NG)                                          // in flight while this pair of tiles is computed (past pmax: zeros)
#pragma unroll
            for (int u = 0; u < 2; ++u) {                              // (a tile past pmax is all zeros and fully masked: an exact no-op)
            const int t = 2 * tp + u;
            const bf16_t* Kt = Ks[it & 1] + u * 32 * AF_KLD;
            const bf16_t* Vt = Vs[it & 1] + u * 32 * AF_VLD;
            // ---- S^T = K . Q^T ------------------------------------------------------------------------
            f32x16_t sacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) sacc[i] = 0.f;
            const int sw = (r >> 1) & 7;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const u32x4_t kf = *reinterpret_cast<const u32x4_t*>(Kt + r * AF_KLD + (((4 * h + s) ^ sw) * 8));
                sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(kf), as_bf16x8(qf[s]), sacc, 0, 0, 0);
            }
            // ---- online softmax over this lane pair's 32 keys -------------------------------------------
            float p[16];
            float tmax = -INFINITY;
            // (round 4, measured and removed: skipping the causal mask on tiles below every row's position and the rescale of O while no
            //  row's maximum moves -- wave-uniform branches, same bits, 53.5 us before and after at 1,334 rows: not VALU-bound; what is left
            //  is one K/V round trip per 64 keys with a single pair of tiles in flight per block)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = t * 32 + 8 * (i >> 2) + 4 * h + (i & 3);  // key of accumulator register i
                p[i] = (j <= pq) ? __fmul_rn(sacc[i], a.scale) : -INFINITY;   // (explicit roundings: see below)
                tmax = fmaxf(tmax, p[i]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, WAVE));
            const float mnew = fmaxf(mrun, tmax);                       // -inf while this group has not met a visible key (group 0: never, key 0)
            const bool none = mnew == -INFINITY;
            const float corr = none ? 1.0f : __expf(__fsub_rn(mrun, mnew));   // 0 on the first visible tile
            float psum = 0.f;
#pragma unroll
            // (score * scale and its distance to the maximum are rounded separately, the running sum is ONE fma: without the explicit forms
            //  hipcc contracts `s * scale - m` into an fma or not depending on the surrounding code -- the 64-key restructuring of round 4
            //  moved one element per ~500 rows by an ulp against the 32-key build; with them both builds give the same bits: tools/dbg/flash_ab.py)
            for (int i = 0; i < 16; ++i) { p[i] = none ? 0.f : __expf(__fsub_rn(p[i], mnew)); psum = __fadd_rn(psum, p[i]); }
            psum = __fadd_rn(psum, __shfl_xor(psum, 32, WAVE));
            lrun = fmaf(lrun, corr, psum);
            mrun = mnew;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= corr; o1[i] *= corr; }
            // ---- O^T += V^T . P^T ---------------------------------------------------------------------
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4_t pf;                                             // P^T fragment: registers 8s .. 8s+7 as bf16
                pf[0] = pack_bf(p[8 * s + 0], p[8 * s + 1]); pf[1] = pack_bf(p[8 * s + 2], p[8 * s + 3]);
                pf[2] = pack_bf(p[8 * s + 4], p[8 * s + 5]); pf[3] = pack_bf(p[8 * s + 6], p[8 * s + 7]);
                // V^T fragment of lane (r = d, h): element j <- key 16s + 8(j>>2) + 4h + (j&3): two transposed reads
                // of 4 keys x 16 d per 16-lane group; lane 4q+p of a group addresses key row q, columns 4p..4p+3
                const int grp_d = 16 * ((lane >> 4) & 1), qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const bf16_t* base = Vt + (16 * s + 4 * h + qq) * AF_VLD + 32 * dt + grp_d + 4 * pp;
                    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4_t __attribute__((address_space(3)))*)(base));
                    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (s16x4_t __attribute__((address_space(3)))*)(base + 8 * AF_VLD));
                    u32x4_t vf;
                    const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
                    vf[0] = l2[0]; vf[1] = l2[1]; vf[2] = h2[0]; vf[3] = h2[1];
                    if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf), as_bf16x8(pf), o0, 0, 0, 0);
                    else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(vf), as_bf16x8(pf), o1, 0, 0, 0);
                }
            }
            }
            AF_LWRITE((it + 1) & 1)
            __syncthreads();
        }
        // ---- fold the key groups' states into group 0's, in group order: m = max, O and l rescaled by exp(m_g - m) (0 for a group
        //      without a visible key).  Group g > 0 parks (O^T, m, l) of its wave in LDS, lane-major (conflict-free), over the K/V tiles.
        if (NG > 1) {
            float* park = reinterpret_cast<float*>(af_lds);            // [NG - 1][4 waves][34][64 lanes] fp32 = 34,816 B per group <= AF_GRP_BYTES
            if (grp > 0) {
                float* mine = park + ((grp - 1) * 4 + wave) * 34 * 64 + lane;
#pragma unroll
                for (int i = 0; i < 16; ++i) { mine[i * 64] = o0[i]; mine[(16 + i) * 64] = o1[i]; }
                mine[32 * 64] = mrun; mine[33 * 64] = lrun;
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int gg = 1; gg < NG; ++gg) {
                    const float* theirs = park + ((gg - 1) * 4 + wave) * 34 * 64 + lane;
                    const float mg = theirs[32 * 64], lg = theirs[33 * 64];
                    const float mnew = fmaxf(mrun, mg);                 // finite: group 0 has seen key 0
                    const float ca = __expf(__fsub_rn(mrun, mnew)), cb = mg == -INFINITY ? 0.f : __expf(__fsub_rn(mg, mnew));
                    lrun = fmaf(lrun, ca, __fmul_rn(lg, cb));
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        o0[i] = fmaf(o0[i], ca, __fmul_rn(theirs[i * 64], cb));
                        o1[i] = fmaf(o1[i], ca, __fmul_rn(theirs[(16 + i) * 64], cb));
                    }
                    mrun = mnew;
                }
            }
        }
        // ---- out[query][hq][d] = O^T / l: registers 4g4 .. 4g4+3 are 4 consecutive d -> one 8-byte store ---------
        if (grp == 0 && r < nr) {
            const float inv = 1.0f / lrun;
            // (out_packed: matrix-core operand order for the o-projection, common.cuh xp_off -- 8-byte halves of its 16-byte pieces)
            bf16_t* dst = a.out + ((long)(m_base + r) * a.H + hq) * HD;
            const long KO = (long)a.H * HD;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2_t w0, w1;
                w0[0] = pack_bf(o0[4 * g4] * inv, o0[4 * g4 + 1] * inv); w0[1] = pack_bf(o0[4 * g4 + 2] * inv, o0[4 * g4 + 3] * inv);
                w1[0] = pack_bf(o1[4 * g4] * inv, o1[4 * g4 + 1] * inv); w1[1] = pack_bf(o1[4 * g4 + 2] * inv, o1[4 * g4 + 3] * inv);
                if (a.out_packed) {
                    *reinterpret_cast<u32x2_t*>(a.out + xp_off(m_base + r, hq * HD + 8 * g4 + 4 * h, KO)) = w0;
                    *reinterpret_cast<u32x2_t*>(a.out + xp_off(m_base + r, hq * HD + 32 + 8 * g4 + 4 * h, KO)) = w1;
                } else {
                    *reinterpret_cast<u32x2_t*>(dst + 8 * g4 + 4 * h) = w0;
                    *reinterpret_cast<u32x2_t*>(dst + 32 + 8 * g4 + 4 * h) = w1;
                }
            }
        }
    }
#undef AF_GLOAD
#undef AF_LWRITE
}
