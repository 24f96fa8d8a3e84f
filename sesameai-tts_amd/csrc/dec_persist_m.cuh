// Depth-decoder steps 2..31 of one frame for M = 2..32 batched utterances as ONE persistent launch (the CSM-1B decoder
// shape).  reference: sesameai/models.py:165-182.  The batch-1 launch (dec_persist.cuh) replicates the whole attention on
// every workgroup and splits the down projection 256 ways; neither scales with the row count (K/V of 32 rows is 1 MB per
// layer, the split-K partials would be 32 MB per layer), so this kernel is laid out differently:
//
//   * every projection is split over the 256 workgroups' OUTPUT columns and runs on v_mfma_f32_16x16x32_bf16 with the
//     weights as the A operand (16 weight rows) and 16 utterances as the B operand's columns: M <= 16 costs exactly the
//     matrix instructions of the batch-1 launch, M <= 32 twice that (MT = 1 / 2 row tiles);
//   * workgroup c = (utterance b = c >> 3, head h = c & 7) owns that (row, head)'s attention and its K/V cache (64 KB of
//     LDS for the four layers), and samples row b (all 8 owners of a row sample alike, so no token broadcast to them);
//   * the MLP is a 16 x 16 grid: workgroup c = 16 j + g produces 32 ffn columns; group g (16 workgroups, one XCD under
//     round-robin placement -- speed only) gathers its 512 columns of h and workgroup (g, j) multiplies them into 64
//     output columns; the 16 partial sums of a column are added in fixed order by the column's owner (deterministic);
//   * the residual stream stays distributed: workgroup c owns columns 4c..4c+3 of every row.
//
// Exchange: activations cross workgroups through global buffers written with sc1 (write-through) stores and read with
// sc1 loads.  No flags and no tags: every buffer is filled with 0xFF by a memset node in front of the launch, a payload
// dword 0xFFFFFFFF cannot occur (a bf16 pair of two all-ones NaNs / an fp32 NaN with every mantissa bit / a negative
// token -- producers clear the lowest bit of such a NaN, dm_clean), so a consumer re-reads its 16-byte pieces until no dword is 0xFFFFFFFF -- the data is its own flag, at 1x the
// payload bytes (8-byte {tag, value} granules would move 128 KB per workgroup per all-gather at 32 rows).  Buffers rotate
// three deep per edge; a producer re-poisons, with the store that publishes instance n, its slots of the buffer instance
// n + 2 will use (read last for instance n - 1, a whole layer ago).
//
// Waves 0..3 compute (weights trickled into VGPRs while they wait on LDS counters, dec_persist.cuh's dp_wait), waves
// 4..7 gather (no weight loads in flight, so their polls never queue behind the stream), run the attention, the
// owner-side sums and the sampler.  Every spin is bounded (s_memrealtime) and ends in *err.
//
// More than 16 rows run as TWO independent halves (rows 0..15, 16..31) whose phases interleave in every wave's program
// order: A's phase k, B's phase k, A's phase k + 1, ...  A step is a chain of ~26 cross-workgroup exchanges of 3-4 us each
// with ~1.5 us of work between them; handled as one 32-row problem the work doubles and the waits stay (measured: 181 us per
// step against 88 us at 16 rows), interleaved, one half's exchange is in flight while the other half's rows are normalised,
// multiplied and published.  The halves share the weights in registers and nothing else (own counters, own buffer rows).
#pragma once
#include "dec_persist.cuh"

#define DM_R 3
#define DM_X_BYTES (32 * 1024 * 2)
#define DM_Q_BYTES (32 * 1536 * 2)
#define DM_HG_BYTES (16 * 32 * 512 * 2)
#define DM_P_BYTES (16 * 16 * 32 * 64 * 4)
#define DM_L_BYTES (257 * 512)                    // logits: [256 workgroups + the tail block][32 rows][8] bf16
#define DM_T_BYTES 128
#define DM_OFF_X 0
#define DM_OFF_A (DM_OFF_X + DM_R * DM_X_BYTES)
#define DM_OFF_H1 (DM_OFF_A + DM_R * DM_X_BYTES)
#define DM_OFF_Q (DM_OFF_H1 + DM_R * DM_X_BYTES)
#define DM_OFF_HG (DM_OFF_Q + DM_R * DM_Q_BYTES)
#define DM_OFF_P (DM_OFF_HG + DM_R * DM_HG_BYTES)
#define DM_OFF_L (DM_OFF_P + DM_R * DM_P_BYTES)
#define DM_OFF_T (DM_OFF_L + DM_R * DM_L_BYTES)
#define DM_XCHG_BYTES (DM_OFF_T + DM_R * DM_T_BYTES)
#define DM_W13M_U4 (256L * 4 * 32 * 64)           // 16-byte pieces per layer of the packed W1 | W3
#define DM_W2M_U4 (256L * 4 * 16 * 64)            // ... of the packed W2

// LDS image (bytes)
#define DM_L_K 0                                  // [4 layers][32 pos][128] bf16 of this workgroup's (row, kv head)
#define DM_L_V 32768
#define DM_L_XB 65536                             // activations in B-operand order, 32 KB per half: piece (p, row) at (p*16 + row) * 16
#define DM_L_RED 131072                           // small-op partial sums [2 halves][4 waves][4][64] f32
#define DM_L_MISC 139264
#define DM_L_HRES (DM_L_MISC + 1024)              // [32][4] bf16: my residual columns entering the layer
#define DM_L_HRES1 (DM_L_HRES + 256)              // ... after the o-projection
#define DM_L_QB (DM_L_HRES1 + 256)                // q of my (row, head)
#define DM_L_PS (DM_L_QB + 256)                   // attention probabilities
#define DM_L_ROPE (DM_L_PS + 128)                 // [32 pos][3 pairs] (cos, sin) of my q|k|v pairs
#define DM_L_SMAX (DM_L_ROPE + 384)               // sampler: 256 u32
#define DM_L_NORM (DM_L_SMAX + 1024)              // [9][1024] bf16: sa/mlp norms of the 4 layers, final norm
#define DM_L_SSQ (DM_L_NORM + 9 * 2048)           // [2 sweep parities][2 halves][4 gather waves][16 rows] f32 partial sums of squares
#define DM_LDS_BYTES (DM_L_SSQ + 1024)
#define DM_L_CANDT 0                              // sampler scratch: offsets into the owner's half of the activation buffer
#define DM_L_CANDI (DP_CAND_SLOTS * 4)
static_assert(DM_LDS_BYTES <= 163840, "LDS image exceeds 160 KB");
// misc words
#define DM_M_FILL 0      // [2 halves] gather waves: +1 each per fill of the half's activation buffer
#define DM_M_CDONE 2     // [2] compute waves: +1 each once a phase has read it
#define DM_M_RED 4       // [2] small-op arrivals
#define DM_M_FT 6        // [2] residual rows of step s are in HRES when >= s + 1
#define DM_M_ABORT 22
#define DM_M_BAR 23      // sampler quad barrier
#define DM_M_SBV 8
#define DM_M_SBI 12
#define DM_M_SN 16
#define DM_M_STOK 17
#define DM_M_SWTOT 18
#define DM_M_RNG 24
#define DM_M_SARG 28

struct DecPersistMArgs {
    const bf16_t* wsm;                // [4 layers][2560 rows][1024] (as DecPersistArgs)
    const bf16_t* norms;              // [4][2][1024]
    const uint4* w2m;                 // [4] x k_dm_pack_down
    const uint4* w13m;                // [4] x k_dm_pack_gateup
    const bf16_t* dec_norm;
    const bf16_t* head_t;             // [ncb-1][V][1024]
    const bf16_t* rope;               // [max_seq][64][2]
    const bf16_t* proj_emb;           // [ncb*V][1024]
    const bf16_t* qkv0_tab;           // [(ncb-2)*V][1536]
    const bf16_t *hdec, *qd;          // [M][1024] decoder input rows / layer-0 q of step cb_first
    const bf16_t *kc, *vc;            // decoder caches [L][max_batch][2][32][128]
    long kv_layer_stride;
    float temperature; int topk;
    const bf16_t* noise;              // optional [ncb][M][V]
    const uint64_t* rng;
    const int* forced;                // optional [M][ncb]
    int V, ncb, M;
    int* frame;                       // [M][ncb]
    bf16_t* logits_out;               // optional [ncb][M][V]
    int cb_first, cb_last;
    char* xchg;                       // DM_XCHG_BYTES, 0xFF-filled before the launch
    uint32_t* err;
    float eps;
    int trickle_sleep, poll_sleep;
    dp_u64* stamps;                   // optional debug timeline (timeline build only): workgroup 100, [step][128]
};
#define DM_STAMP(i_) do { if (DP_STAMPS(a) != nullptr && cu == 100 && lane == 0) DP_STAMPS(a)[s * 128 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)

// A payload dword must never BE the poison: an all-ones bf16 NaN pair / fp32 NaN (non-finite weights or an overflow upstream) is published
// with its lowest bit cleared -- still a NaN in both halves, so non-finite activations reach the logits like in the reference instead of
// turning into a consumer that spins until its 50 ms bound.  Two VALU instructions per published dword.
__device__ __forceinline__ uint32_t dm_clean(uint32_t v) { return v == 0xffffffffu ? 0xfffffffeu : v; }
__device__ __forceinline__ bool dm_valid(const u32x4_t& v) { return v.x != 0xffffffffu && v.y != 0xffffffffu && v.z != 0xffffffffu && v.w != 0xffffffffu; }
// saddr + 32-bit voffset forms: the buffer base is wave-uniform (an SGPR pair), a lane carries ONE 32-bit offset per access
// pattern and the piece index rides in the instruction's immediate -- full 64-bit per-lane addresses for every buffer of
// every edge were hoisted out of the step loop by the compiler and spilled (640 bytes of scratch per lane)
template <int OFF> __device__ __forceinline__ void dm_lds16(u32x4_t& x, const char* sbase, uint32_t voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(x) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ void dm_sst4(char* sbase, uint32_t voff, uint32_t v) { asm volatile("global_store_dword %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory"); }
__device__ __forceinline__ void dm_sst8(char* sbase, uint32_t voff, uint32_t a, uint32_t b) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    u32x2v v; v.x = a; v.y = b;
    asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dm_sst16(char* sbase, uint32_t voff, const u32x4_t& v) { asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(voff), "v"(v), "s"(sbase) : "memory"); }
template <int J, int NL, int STRIDE> struct DmIssue {
    static __device__ __forceinline__ void go(u32x4_t (&x)[NL], const char* sbase, uint32_t voff, uint32_t done) {
        if (!((done >> J) & 1u)) {
            if constexpr (J * STRIDE < 4096) dm_lds16<J * STRIDE>(x[J], sbase, voff);
            else dm_lds16<0>(x[J], sbase + (long)J * STRIDE, voff);
        }
        if constexpr (J + 1 < NL) DmIssue<J + 1, NL, STRIDE>::go(x, sbase, voff, done);
    }
};
// Re-read NL 16-byte pieces at sbase + voff + j * STRIDE until none of their dwords is the poison (see dm_poll)
template <int NL, int STRIDE>
__device__ __forceinline__ bool dm_poll_s(const char* sbase, uint32_t voff, u32x4_t (&x)[NL], int lane, dp_lvu32* ab, uint32_t* err, uint32_t code, int poll_sleep) {
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t done = 0;
    for (uint32_t pass = 1;; ++pass) {
        DmIssue<0, NL, STRIDE>::go(x, sbase, voff, done);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            asm volatile("" : "+v"(x[j]));
            if (__all(dm_valid(x[j]))) done |= 1u << j;
        }
        done = __builtin_amdgcn_readfirstlane(done);
        if (done == (NL >= 32 ? 0xffffffffu : ((1u << NL) - 1u))) return true;
        if ((pass & 15u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
        for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ void dm_st4(void* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void dm_st8(void* p, uint32_t a, uint32_t b) {
    typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
    u32x2v v; v.x = a; v.y = b;
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void dm_st16(void* p, const u32x4_t& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// Re-read NL 16-byte pieces (addresses p[j]) until none of their dwords is the poison.  A load whose 64 lanes were all
// valid is not issued again (uniform branch, no per-lane predicate).
template <int NL>
__device__ __forceinline__ bool dm_poll(const char* const (&p)[NL], u32x4_t (&x)[NL], int lane, dp_lvu32* ab, uint32_t* err, uint32_t code, int poll_sleep) {
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t done = 0;
    for (uint32_t pass = 1;; ++pass) {
#pragma unroll
        for (int j = 0; j < NL; ++j)
            if (!((done >> j) & 1u)) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[j]) : "v"(p[j]) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            asm volatile("" : "+v"(x[j]));
            if (__all(dm_valid(x[j]))) done |= 1u << j;
        }
        done = __builtin_amdgcn_readfirstlane(done);
        if (done == (NL >= 32 ? 0xffffffffu : ((1u << NL) - 1u))) return true;
        if ((pass & 15u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
        for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
}

// one dword per lane
__device__ __forceinline__ bool dm_poll4(const char* p, uint32_t& x, int lane, dp_lvu32* ab, uint32_t* err, uint32_t code, int poll_sleep) {
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t pass = 1;; ++pass) {
        asm volatile("global_load_dword %0, %1, off sc1" : "=v"(x) : "v"(p) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(x));
        if (__all(x != 0xffffffffu)) return true;
        if ((pass & 15u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
        for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ bool dm_wait_ge(dp_lvu32* f, uint32_t want, dp_lvu32* ab, uint32_t* err, uint32_t code, int lane) {
    if ((int32_t)(*f - want) >= 0) return true;
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t spins = 1; (int32_t)(*f - want) < 0; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ void dm_arrive(dp_lu32* ctr, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// All-gather of 16 rows x (NP * 8) columns of one half into its activation buffer (B-operand order), by the four gather waves: wave gw
// takes rows 4 gw .. + 3 of the half, lane = (row, piece group pg of 16), load j = piece 16 j + pg of the row.  The caller gives the
// lane's byte offset of piece pg of its row and the byte stride between a lane's pieces (buffer layouts differ per edge).
template <int NP, int STRIDE>
__device__ __forceinline__ bool dm_sweep_mat(const char* buf, uint32_t voff, char* xb_half, int gw, int lane, dp_lvu32* ab, uint32_t* err, uint32_t code, int poll_sleep) {
    constexpr int NL = NP / 16;
    const int row = gw * 4 + (lane & 3), pg = lane >> 2;
    u32x4_t x[NL];
    if (!dm_poll_s<NL, STRIDE>(buf, voff, x, lane, ab, err, code, poll_sleep)) return false;
    dp_lu4* xb = (dp_lu4*)xb_half;
#pragma unroll
    for (int j = 0; j < NL; ++j) xb[(j * 16 + pg) * 16 + row] = x[j];
    return true;
}

// All-gather of a half's rows of the residual stream, published as [256 workgroups][32 rows][4 columns] (a producer's 256 bytes are two
// whole cache lines: with a row-major matrix 16 workgroups write 8 bytes each into every line, and the edge took 10-15 us instead of
// 4), + RMSNorm with scale row `norm` of the LDS copy (fp32 normalise -> bf16 -> * bf16 scale, torchtune's rounding points).  Wave gw
// takes the 64 workgroups 64 gw .. + 63 (columns 256 gw .. + 255) of ALL 16 rows: lane = (row pair of the half, workgroup lane >> 3),
// load j = workgroup 64 gw + 8 j + (lane >> 3) -- 8 lanes read one whole 128-byte line, a line is read once per workgroup (with the rows
// split over the waves instead, every wave touched every line for 32 bytes and the sweep was request-bound: 6.5 us against 2.7 for the
// same bytes row-major).  The rows' sums of squares meet in LDS: per-wave partials, the four gather waves' barrier, fixed-order sum.
template <class Sync>
__device__ __forceinline__ bool dm_sweep_cols(const char* buf, int hf, int M, int norm, float eps, char* lds, int gw, int lane, dp_lvu32* ab, uint32_t* err,
                                              uint32_t code, int poll_sleep, const Sync& sync, int parity) {
    const int pr = lane & 7, cl = lane >> 3;
    const int prc = min(8 * hf + pr, (M - 1) >> 1);
    u32x4_t x[8];
    if (!dm_poll_s<8, 8 * 256>(buf, (uint32_t)((gw * 64 + cl) * 256 + prc * 16), x, lane, ab, err, code, poll_sleep)) return false;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float f;
        f = lo2f(x[j].x); s0 += f * f; f = hi2f(x[j].x); s0 += f * f; f = lo2f(x[j].y); s0 += f * f; f = hi2f(x[j].y); s0 += f * f;
        f = lo2f(x[j].z); s1 += f * f; f = hi2f(x[j].z); s1 += f * f; f = lo2f(x[j].w); s1 += f * f; f = hi2f(x[j].w); s1 += f * f;
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    // (two buffers by sweep parity: a wave may write the next sweep's partials while a slower one still reads this sweep's)
    dp_lf32* ssq = (dp_lf32*)(lds + DM_L_SSQ) + (parity * 2 + hf) * 64;                 // [4 waves][16 rows]
    if (lane < 8) { ssq[gw * 16 + 2 * pr] = s0; ssq[gw * 16 + 2 * pr + 1] = s1; }
    sync();
    const float t0 = ((ssq[2 * pr] + ssq[16 + 2 * pr]) + ssq[32 + 2 * pr]) + ssq[48 + 2 * pr];
    const float t1 = ((ssq[2 * pr + 1] + ssq[16 + 2 * pr + 1]) + ssq[32 + 2 * pr + 1]) + ssq[48 + 2 * pr + 1];
    const float r0 = 1.0f / sqrtf(t0 / 1024.0f + eps), r1 = 1.0f / sqrtf(t1 / 1024.0f + eps);
    const dp_lu32* g = (const dp_lu32*)(lds + DM_L_NORM + norm * 2048);
    typedef __attribute__((address_space(3))) unsigned long long dm_lu64;
    char* xb = lds + DM_L_XB + hf * 32768;
    const int ra = 2 * pr, rb = 2 * pr + 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        asm volatile("" : "+v"(x[j]));            // unpack again here: fp32 copies kept from the sum-of-squares pass spill
        const int c = gw * 64 + j * 8 + cl;
        const uint32_t g0 = g[2 * c], g1 = g[2 * c + 1];
        const uint32_t a0 = pack_bf(round_bf(lo2f(x[j].x) * r0) * lo2f(g0), round_bf(hi2f(x[j].x) * r0) * hi2f(g0));
        const uint32_t a1 = pack_bf(round_bf(lo2f(x[j].y) * r0) * lo2f(g1), round_bf(hi2f(x[j].y) * r0) * hi2f(g1));
        const uint32_t b0 = pack_bf(round_bf(lo2f(x[j].z) * r1) * lo2f(g0), round_bf(hi2f(x[j].z) * r1) * hi2f(g0));
        const uint32_t b1 = pack_bf(round_bf(lo2f(x[j].w) * r1) * lo2f(g1), round_bf(hi2f(x[j].w) * r1) * hi2f(g1));
        // columns 4c..4c+3 = half (c & 1) of piece c >> 1
        *(dm_lu64*)(xb + (((c >> 1) * 16 + ra) * 16 + (c & 1) * 8)) = ((unsigned long long)a1 << 32) | a0;
        *(dm_lu64*)(xb + (((c >> 1) * 16 + rb) * 16 + (c & 1) * 8)) = ((unsigned long long)b1 << 32) | b0;
    }
    return true;
}

typedef __attribute__((ext_vector_type(4))) float dm_f32x4;

// NT k steps of a 16-row weight tile (A operand) against one half's 16 rows in its activation buffer (B operand)
template <int NT>
__device__ __forceinline__ void dm_mma(const uint4 (&wf)[NT], const char* xb_half, int t0, int lane, dm_f32x4& acc) {
    // B fragments four k steps at a time, the next four reads in flight behind the current four matrix ops (left alone, the scheduler
    // hoists every read of the phase in front of the first matrix op)
    const dp_lu4* xb = (const dp_lu4*)xb_half + (t0 * 4 + (lane >> 4)) * 16 + (lane & 15);
    uint4 xa[4], xn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xa[k] = dp_ldq(xb + k * 64);
#pragma unroll
    for (int kb = 0; kb < NT; kb += 4) {
        if (kb + 4 < NT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) xn[k] = dp_ldq(xb + (kb + 4 + k) * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dp_bf16x8, wf[kb + k]), __builtin_bit_cast(dp_bf16x8, xa[k]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 4 < NT) {
#pragma unroll
            for (int k = 0; k < 4; ++k) xa[k] = xn[k];
        }
    }
}

// attention of my (row, head) over keys 0..nk-1 of layer l: dec_persist.cuh's dp_attention_head with the head's own q
__device__ __forceinline__ uint32_t dm_attention(char* lds, int l, int nk, int lane) {
    const dp_lu4* qb = (const dp_lu4*)(lds + DM_L_QB);
    const dp_lu4* kt = (const dp_lu4*)(lds + DM_L_K + l * 8192);
    const dp_lu32* vt = (const dp_lu32*)(lds + DM_L_V + l * 8192);
    dp_lf32* myps = (dp_lf32*)(lds + DM_L_PS);
    const int grp = lane >> 4, sub = lane & 15;
    const uint4 qa = dp_ldq(qb + sub);
    uint4 kv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) kv[i] = dp_ldq(kt + i * 64 + lane);
    float s0[8], mx0 = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool live = (4 * i + grp) < nk;
        const float d0 = row16_sum(dot8(qa, kv[i], 0.f)) * 0.08838834764831845f;
        s0[i] = live ? d0 : -INFINITY;
        mx0 = fmaxf(mx0, s0[i]);
    }
    mx0 = wave_max(mx0);
    float l0 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s0[i] = (s0[i] == -INFINITY) ? 0.f : __expf(s0[i] - mx0);
        l0 += s0[i];
        if (sub == 0) myps[4 * i + grp] = s0[i];
    }
    l0 = wave_sum(l0) * (1.0f / 16.0f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float o00 = 0.f, o01 = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < 8; ++t4) {
        const u32x4_t p4 = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(myps + 4 * t4);
        const float pw[4] = {__uint_as_float(p4.x), __uint_as_float(p4.y), __uint_as_float(p4.z), __uint_as_float(p4.w)};
        uint32_t vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vr[u] = vt[(4 * t4 + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) { o00 += pw[u] * lo2f(vr[u]); o01 += pw[u] * hi2f(vr[u]); }
    }
    const float i0 = 1.0f / l0;
    return pack_bf(o00 * i0, o01 * i0);
}

// ---------------------------------------------------------------------------------------------------------------
// compute wave w (0..3); NH = halves (1: up to 16 rows, 2: up to 32)
// ---------------------------------------------------------------------------------------------------------------
template <int NH>
__device__ __forceinline__ void dm_compute_wave(const DecPersistMArgs& a, char* lds, const int w, const int lane, const int cu) {
    dp_lu32* misc = (dp_lu32*)(lds + DM_L_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DM_M_ABORT);
    dp_lvu32* fill = (dp_lvu32*)(misc + DM_M_FILL);
    const int ts = a.trickle_sleep & 63;
    const int g4 = lane >> 4, bl = lane & 15, r16 = lane & 15;
    const int gj = cu >> 4, gg = cu & 15;
    const int mpad = (a.M + 1) & ~1;          // the residual stream travels in row pairs: an odd M publishes one (finite, unused) row more
    uint4 S[8], W[32], D[8];
    // A-operand fragments of the small ops: lane = (weight row r16 of the 16-row tile, k quarter g4); wave w takes k steps 8w..8w+7
    auto small_row = [&](int slot, int cb) -> const bf16_t* {      // slot 0..3: q|k|v of layer slot; 4..7: o-proj of layer slot-4; 8: head of cb
        if (slot < 4) return a.wsm + ((long)slot * DP_WSM_ROWS + 6 * cu + min(r16, 5)) * DP_D;
        if (slot < 8) return a.wsm + ((long)(slot - 4) * DP_WSM_ROWS + DP_NQKV + 4 * cu + min(r16, 3)) * DP_D;
        const int hr = r16 < 8 ? 8 * cu + r16 : (cu == 0 ? min(2048 + r16 - 8, a.V - 1) : 8 * cu + 7);
        return a.head_t + ((long)(cb - 1) * a.V + hr) * DP_D;
    };
    auto load_s = [&](int slot, int cb, int k) { S[k] = *reinterpret_cast<const uint4*>(small_row(slot, cb) + 32 * (8 * w + k) + 8 * g4); };
    auto load_gu = [&](int l, int k) { W[k] = a.w13m[(long)l * DM_W13M_U4 + (((long)cu * 4 + w) * 32 + k) * 64 + lane]; };
    auto load_dn = [&](int l, int k) {
        const uint4 v = a.w2m[(long)l * DM_W2M_U4 + (((long)cu * 4 + w) * 16 + k) * 64 + lane];
        if (k < 8) D[k & 7] = v; else W[(k - 8) & 31] = v;
    };
    // last-arriver fold of the four waves' K-quarter partial tiles of one half; true in the wave that arrived last, with the sums in v
    auto fold = [&](int hf, const dm_f32x4& acc, float (&v)[4]) -> bool {
        dp_lf32* red = (dp_lf32*)(lds + DM_L_RED) + hf * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(w * 4 + i) * 64 + lane] = acc[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(misc + DM_M_RED + hf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
    // the second half's wait: plain poll of its fill counter (the phase's weights are in registers already)
    auto wait_half = [&](int hf, uint32_t want, uint32_t code) -> bool { return dm_wait_ge(fill + hf, want, ab, a.err, code, lane); };
    uint32_t kp = 0;
    const int n_steps = a.cb_last - a.cb_first + 1;
    for (int s = 0; s < n_steps; ++s) {
        const int cb = a.cb_first + s;
        for (int l = 0; l < DP_NL; ++l) {
            const int n = s * DP_NL + l;
            const int nq = s * (DP_NL - 1) + (l > 0 ? l - 1 : 0);      // the q|k|v edge has no instance at layer 0: its own dense count
            char* const xq = a.xchg + DM_OFF_Q + (nq % DM_R) * DM_Q_BYTES, *const xq2 = a.xchg + DM_OFF_Q + ((nq + 2) % DM_R) * DM_Q_BYTES;
            char* const xh = a.xchg + DM_OFF_H1 + (n % DM_R) * DM_X_BYTES, *const xh2 = a.xchg + DM_OFF_H1 + ((n + 2) % DM_R) * DM_X_BYTES;
            char* const xg = a.xchg + DM_OFF_HG + (n % DM_R) * DM_HG_BYTES, *const xg2 = a.xchg + DM_OFF_HG + ((n + 2) % DM_R) * DM_HG_BYTES;
            char* const xp = a.xchg + DM_OFF_P + (n % DM_R) * DM_P_BYTES, *const xp2 = a.xchg + DM_OFF_P + ((n + 2) % DM_R) * DM_P_BYTES;
            {   // ---- q|k|v of my 6 columns (layers 1..3; layer 0's come from the table) ----
                if (l > 0) ++kp;
                if (!dp_wait<8, true>(fill, l > 0 ? 4u * kp : 0u, ab, a.err, 0xA10u, lane, ts, [&](int k) { load_s(l, cb, k); })) return;
                if (w == 0) DM_STAMP(64 + l * 8 + 0);
                if (l > 0) {
#pragma unroll 1
                    for (int hf = 0; hf < NH; ++hf) {
                        if (hf > 0 && !wait_half(hf, 4u * kp, 0xA11u)) return;
                        dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                        dm_mma<8>(S, lds + DM_L_XB + hf * 32768, 8 * w, lane, acc);
                        dm_arrive(misc + DM_M_CDONE + hf, lane);
                        float v[4];
                        if (fold(hf, acc, v)) {
                            // lane (g4, b): rows 4 g4 .. + 3 of my 6: g4 = 0 -> pairs 0, 1; g4 = 1 -> pair 2
                            const dp_lu32* rp = (const dp_lu32*)(lds + DM_L_ROPE) + cb * 3;
                            const int b = 16 * hf + bl;
                            if (g4 < 2 && b < a.M) {
                                const int u0 = 2 * g4;
                                const uint32_t o0 = dp_rope_pair(v[0], v[1], rp[u0], 2 * (3 * cu + u0) < 1280);
                                dm_sst4(xq, (uint32_t)(b * 3072 + (6 * cu + 2 * u0) * 2), dm_clean(o0));
                                dm_sst4(xq2, (uint32_t)(b * 3072 + (6 * cu + 2 * u0) * 2), 0xffffffffu);
                                if (g4 == 0) {
                                    const uint32_t o1 = dp_rope_pair(v[2], v[3], rp[1], 2 * (3 * cu + 1) < 1280);
                                    dm_sst4(xq, (uint32_t)(b * 3072 + (6 * cu + 2) * 2), dm_clean(o1));
                                    dm_sst4(xq2, (uint32_t)(b * 3072 + (6 * cu + 2) * 2), 0xffffffffu);
                                }
                            }
                        }
                    }
                }
            }
            if (w == 0) DM_STAMP(64 + l * 8 + 1);
            {   // ---- o-projection of my 4 columns + residual ----
                ++kp;
                if (!dp_wait<24, true>(fill, 4u * kp, ab, a.err, 0xA20u, lane, ts, [&](int k) { if (k < 8) load_s(4 + l, cb, k); else load_gu(l, k - 8); })) return;
                if (w == 0) DM_STAMP(64 + l * 8 + 2);
#pragma unroll 1
                for (int hf = 0; hf < NH; ++hf) {
                    if (hf > 0 && !wait_half(hf, 4u * kp, 0xA21u)) return;
                    dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    dm_mma<8>(S, lds + DM_L_XB + hf * 32768, 8 * w, lane, acc);
                    dm_arrive(misc + DM_M_CDONE + hf, lane);
                    float v[4];
                    if (fold(hf, acc, v)) {
                        if (l == 0 && !dm_wait_ge((dp_lvu32*)(misc + DM_M_FT + hf), (uint32_t)(s + 1), ab, a.err, 0xA25u, lane)) return;
                        const int b = 16 * hf + bl;
                        if (g4 == 0 && b < mpad) {
                            const dp_lu32* hr = (const dp_lu32*)(lds + DM_L_HRES) + 2 * b;
                            const uint32_t p0 = dp_resid_pair(v[0], v[1], hr[0]), p1 = dp_resid_pair(v[2], v[3], hr[1]);
                            dp_lu32* h1 = (dp_lu32*)(lds + DM_L_HRES1) + 2 * b;
                            h1[0] = p0; h1[1] = p1;
                            dm_sst8(xh, (uint32_t)(cu * 256 + b * 8), dm_clean(p0), dm_clean(p1));
                            dm_sst8(xh2, (uint32_t)(cu * 256 + b * 8), 0xffffffffu, 0xffffffffu);
                        }
                    }
                }
            }
            if (w == 0) DM_STAMP(64 + l * 8 + 3);
            {   // ---- gate / up of my 32 pairs: tile w = pairs 8w..8w+7 as rows (g, u, g, u, ...) ----
                ++kp;
                if (!dp_wait<24, true>(fill, 4u * kp, ab, a.err, 0xA30u, lane, ts, [&](int k) { if (k < 16) load_gu(l, 16 + k); else load_dn(l, k - 16); })) return;
                if (w == 0) DM_STAMP(64 + l * 8 + 4);
#pragma unroll 1
                for (int hf = 0; hf < NH; ++hf) {
                    if (hf > 0 && !wait_half(hf, 4u * kp, 0xA31u)) return;
                    dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    dm_mma<32>(W, lds + DM_L_XB + hf * 32768, 0, lane, acc);
                    dm_arrive(misc + DM_M_CDONE + hf, lane);
                    const int b = 16 * hf + bl;
                    if (b < a.M) {
                        const uint32_t hv = dp_swiglu(acc[0], acc[1]) | (dp_swiglu(acc[2], acc[3]) << 16);
                        const uint32_t off = (uint32_t)(gg * 32768 + (gj * 4 + w) * 512 + b * 16 + g4 * 4);      // [g][piece 4 j + w][32 rows][8 columns]: a wave's 16 rows are 256 contiguous bytes
                        dm_sst4(xg, off, dm_clean(hv));
                        dm_sst4(xg2, off, 0xffffffffu);
                    }
                }
            }
            if (w == 0) DM_STAMP(64 + l * 8 + 5);
            {   // ---- down projection: my 64 output columns over my group's 512 ffn columns; tile w = columns 16w..16w+15 ----
                ++kp;
                if (!dp_wait<8, true>(fill, 4u * kp, ab, a.err, 0xA40u, lane, ts, [&](int k) { load_dn(l, 8 + k); })) return;
                if (w == 0) DM_STAMP(64 + l * 8 + 6);
#pragma unroll 1
                for (int hf = 0; hf < NH; ++hf) {
                    if (hf > 0 && !wait_half(hf, 4u * kp, 0xA41u)) return;
                    dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    dm_mma<8>(D, lds + DM_L_XB + hf * 32768, 0, lane, acc);
                    {
                        uint4 W8[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) W8[k] = W[k];
                        dm_mma<8>(W8, lds + DM_L_XB + hf * 32768, 8, lane, acc);
                    }
                    dm_arrive(misc + DM_M_CDONE + hf, lane);
                    const int b = 16 * hf + bl;
                    if (b < a.M) {
                        const uint32_t off = (uint32_t)((((((gj * 16 + gg) * 4 + w) * 32 + b) * 16) + 4 * g4) * 4);      // [j][g][wave][32 rows][16 columns] f32: a wave's 16 rows are 1 KB contiguous
                        u32x4_t o; o.x = dm_clean(__float_as_uint(acc[0])); o.y = dm_clean(__float_as_uint(acc[1])); o.z = dm_clean(__float_as_uint(acc[2])); o.w = dm_clean(__float_as_uint(acc[3]));
                        u32x4_t ff; ff.x = ff.y = ff.z = ff.w = 0xffffffffu;
                        dm_sst16(xp, off, o);
                        dm_sst16(xp2, off, ff);
                    }
                }
            }
            if (w == 0) DM_STAMP(64 + l * 8 + 7);
        }
        {   // ---- head of codebook cb: my 8 logit rows (+ the tail rows on workgroup 0) ----
            ++kp;
            if (!dp_wait<8, true>(fill, 4u * kp, ab, a.err, 0xA50u, lane, ts, [&](int k) { load_s(8, cb, k); })) return;
            if (w == 0) DM_STAMP(96);
            char* const xl = a.xchg + DM_OFF_L + (s % DM_R) * DM_L_BYTES, *const xl2 = a.xchg + DM_OFF_L + ((s + 2) % DM_R) * DM_L_BYTES;
#pragma unroll 1
            for (int hf = 0; hf < NH; ++hf) {
                if (hf > 0 && !wait_half(hf, 4u * kp, 0xA51u)) return;
                dm_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                dm_mma<8>(S, lds + DM_L_XB + hf * 32768, 8 * w, lane, acc);
                dm_arrive(misc + DM_M_CDONE + hf, lane);
                float v[4];
                if (fold(hf, acc, v)) {
                    const int b = 16 * hf + bl;
                    if (b < a.M && (g4 < 2 || cu == 0)) {
                        const uint32_t off = (uint32_t)((g4 < 2 ? cu : 256) * 512 + b * 16 + (g4 & 1) * 8);
                        dm_sst8(xl, off, dm_clean(pack_bf(v[0], v[1])), dm_clean(pack_bf(v[2], v[3])));
                        dm_sst8(xl2, off, 0xffffffffu, 0xffffffffu);
                    }
                }
            }
            if (w == 0) DM_STAMP(97);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gather wave gw (0..3)
// ---------------------------------------------------------------------------------------------------------------
template <int NH>
__device__ __forceinline__ void dm_gather_wave(const DecPersistMArgs& a, char* lds, const int gw, const int lane, const int cu) {
    dp_lu32* misc = (dp_lu32*)(lds + DM_L_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DM_M_ABORT);
    dp_lvu32* cdone = (dp_lvu32*)(misc + DM_M_CDONE);
    const int ob = cu >> 3, oh = cu & 7, kvh = oh >> 2;      // the (row, head) this workgroup owns
    const int ohf = ob < a.M ? (ob >> 4) : -1;                 // ... in which half (-1: no row of this batch)
    const int gj = cu >> 4, gg = cu & 15;
    const int ps = a.poll_sleep;
    const int mpad = (a.M + 1) & ~1;
    uint32_t kf = 0, nsw = 0, quad_phase = 0;          // fills so far, normed sweeps so far
    const DpQuadSync qsync{(dp_lvu32*)(misc + DM_M_BAR), ab, a.err, lane, &quad_phase, nullptr};      // barrier of the four gather waves
    const int n_steps = a.cb_last - a.cb_first + 1;
    // my lane's piece offsets in the row-major attention matrix and in my group's h pieces [64 pieces = 4 j + wave][32 rows][8 columns], per half
    uint32_t voff_a[NH], voff_g[NH];
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
        const int rowc = min(16 * hf + gw * 4 + (lane & 3), a.M - 1), pg = lane >> 2;
        voff_a[hf] = (uint32_t)(rowc * 2048 + pg * 16);
        voff_g[hf] = (uint32_t)(pg * 512 + rowc * 16);
    }
    // fill k of a half's activation buffer may start once that half's phase k - 1 has read it
#define DM_FILL_BEGIN(hf_, code_) do { if (!dm_wait_ge(cdone + (hf_), 4u * (kf - 1), ab, a.err, (code_), lane)) return; } while (0)
#define DM_FILL_END(hf_) dm_arrive(misc + DM_M_FILL + (hf_), lane)
    for (int s = 0; s < n_steps; ++s) {
        const int cb = a.cb_first + s;
        for (int l = 0; l < DP_NL; ++l) {
            const int n = s * DP_NL + l;
            if (l > 0) {
                ++kf; ++nsw;
#pragma unroll 1
                for (int hf = 0; hf < NH; ++hf) {
                    DM_FILL_BEGIN(hf, 0xB10u);
                    if (!dm_sweep_cols(a.xchg + DM_OFF_X + ((n - 1) % DM_R) * DM_X_BYTES, hf, a.M, 2 * l, a.eps, lds, gw, lane, ab, a.err, 0x110u + l, ps, qsync, (int)(nsw & 1u))) return;
                    DM_FILL_END(hf);
                }
                if (gw == 0) DM_STAMP(l * 8 + 0);
                if (ohf >= 0 && gw == 0) {
                    // q of my head, k / v of my kv head, this step's position
                    const int nq = s * (DP_NL - 1) + l - 1;
                    const char* q = a.xchg + DM_OFF_Q + (nq % DM_R) * DM_Q_BYTES + (long)ob * 3072;
                    const int ln = min(lane, 47);
                    const char* p[1] = {ln < 16 ? q + oh * 256 + ln * 16 : (ln < 32 ? q + 2048 + kvh * 256 + (ln - 16) * 16 : q + 2560 + kvh * 256 + (ln - 32) * 16)};
                    u32x4_t x[1];
                    if (!dm_poll<1>(p, x, lane, ab, a.err, 0x210u + l, ps)) return;
                    if (lane < 16) ((dp_lu4*)(lds + DM_L_QB))[lane] = x[0];
                    else if (lane < 32) ((dp_lu4*)(lds + DM_L_K + (l * 32 + cb) * 256))[lane - 16] = x[0];
                    else if (lane < 48) ((dp_lu4*)(lds + DM_L_V + (l * 32 + cb) * 256))[lane - 32] = x[0];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    DM_STAMP(l * 8 + 1);
                }
            }
            if (ohf >= 0 && gw == 0) {
                const uint32_t o = dm_attention(lds, l, cb + 1, lane);
                const long off = (long)ob * 2048 + oh * 256 + lane * 4;
                dm_st4(a.xchg + DM_OFF_A + (n % DM_R) * DM_X_BYTES + off, dm_clean(o));
                dm_st4(a.xchg + DM_OFF_A + ((n + 2) % DM_R) * DM_X_BYTES + off, 0xffffffffu);
                DM_STAMP(l * 8 + 2);
            }
            ++kf;
#pragma unroll 1
            for (int hf = 0; hf < NH; ++hf) {
                DM_FILL_BEGIN(hf, 0xB20u);
                if (!dm_sweep_mat<128, 256>(a.xchg + DM_OFF_A + (n % DM_R) * DM_X_BYTES, voff_a[hf], lds + DM_L_XB + hf * 32768, gw, lane, ab, a.err, 0x310u + l, ps)) return;
                DM_FILL_END(hf);
            }
            if (gw == 0) DM_STAMP(l * 8 + 3);
            ++kf; ++nsw;
#pragma unroll 1
            for (int hf = 0; hf < NH; ++hf) {
                DM_FILL_BEGIN(hf, 0xB30u);
                if (!dm_sweep_cols(a.xchg + DM_OFF_H1 + (n % DM_R) * DM_X_BYTES, hf, a.M, 2 * l + 1, a.eps, lds, gw, lane, ab, a.err, 0x410u + l, ps, qsync, (int)(nsw & 1u))) return;
                DM_FILL_END(hf);
            }
            if (gw == 0) DM_STAMP(l * 8 + 4);
            ++kf;
#pragma unroll 1
            for (int hf = 0; hf < NH; ++hf) {
                DM_FILL_BEGIN(hf, 0xB40u);
                if (!dm_sweep_mat<64, 8192>(a.xchg + DM_OFF_HG + (n % DM_R) * DM_HG_BYTES + (long)gg * 32768, voff_g[hf], lds + DM_L_XB + hf * 32768, gw, lane, ab, a.err, 0x510u + l, ps)) return;
                DM_FILL_END(hf);
            }
            if (gw == 0) DM_STAMP(l * 8 + 5);
            if (gw == 1) {
#pragma unroll 1
                for (int hf = 0; hf < NH; ++hf) {
                    // the 16 partial sums of my 4 columns, every row of the half: lane = (row, quarter of the groups); fixed order
                    const int b = 16 * hf + (lane & 15), bc = min(b, a.M - 1), qt = lane >> 4;
                    const char* pb = a.xchg + DM_OFF_P + (n % DM_R) * DM_P_BYTES;
                    const char* p[4];
                    u32x4_t x[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) p[i] = pb + ((((((long)gj * 16 + (qt * 4 + i)) * 4 + (gg >> 2)) * 32 + bc) * 16) + 4 * (gg & 3)) * 4;
                    if (!dm_poll<4>(p, x, lane, ab, a.err, 0x610u + l, ps)) return;
                    float t[4] = {__uint_as_float(x[0].x), __uint_as_float(x[0].y), __uint_as_float(x[0].z), __uint_as_float(x[0].w)};
#pragma unroll
                    for (int i = 1; i < 4; ++i) { t[0] += __uint_as_float(x[i].x); t[1] += __uint_as_float(x[i].y); t[2] += __uint_as_float(x[i].z); t[3] += __uint_as_float(x[i].w); }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float o1 = __shfl_xor(t[i], 16, 64);
                        const float u = (qt & 1) == 0 ? t[i] + o1 : o1 + t[i];        // (q0 + q1) or (q2 + q3), the same bits in both lanes
                        const float o2 = __shfl_xor(u, 32, 64);
                        t[i] = qt < 2 ? u + o2 : o2 + u;
                    }
                    if (lane < 16 && b < mpad) {
                        const dp_lu32* h1 = (const dp_lu32*)(lds + DM_L_HRES1) + 2 * b;
                        const uint32_t p0 = dp_resid_pair(t[0], t[1], h1[0]), p1 = dp_resid_pair(t[2], t[3], h1[1]);
                        dp_lu32* hr = (dp_lu32*)(lds + DM_L_HRES) + 2 * b;
                        hr[0] = p0; hr[1] = p1;
                        dm_st8(a.xchg + DM_OFF_X + (n % DM_R) * DM_X_BYTES + cu * 256 + b * 8, dm_clean(p0), dm_clean(p1));
                        dm_st8(a.xchg + DM_OFF_X + ((n + 2) % DM_R) * DM_X_BYTES + cu * 256 + b * 8, 0xffffffffu, 0xffffffffu);
                    }
                }
                DM_STAMP(l * 8 + 6);
            }
        }
        {   // the stack's output rows -> final norm -> x of the head
            const int n = s * DP_NL + DP_NL - 1;
            ++kf; ++nsw;
#pragma unroll 1
            for (int hf = 0; hf < NH; ++hf) {
                DM_FILL_BEGIN(hf, 0xB50u);
                if (!dm_sweep_cols(a.xchg + DM_OFF_X + (n % DM_R) * DM_X_BYTES, hf, a.M, 8, a.eps, lds, gw, lane, ab, a.err, 0x710u, ps, qsync, (int)(nsw & 1u))) return;
                DM_FILL_END(hf);
            }
            if (gw == 0) DM_STAMP(32);
        }
        if (ohf >= 0) {
            // logits of my row -> registers (thread tid of the quad owns logits 8 tid .. + 7, tid 0 also the tail piece), sample
            const int tid = gw * 64 + lane;
            const char* lrow = a.xchg + DM_OFF_L + (s % DM_R) * DM_L_BYTES + ob * 16;
            const char* p[2] = {lrow + tid * 512, lrow + (tid == 0 ? 256 : tid) * 512};
            u32x4_t x[2];
            if (!dm_poll<2>(p, x, lane, ab, a.err, 0x810u, ps)) return;
            if (gw == 0) DM_STAMP(33);
            uint32_t wv[2][4] = {{x[0].x, x[0].y, x[0].z, x[0].w}, {0u, 0u, 0u, 0u}};
            if (tid == 0) { wv[1][0] = x[1].x; wv[1][1] = x[1].y; wv[1][2] = x[1].z; wv[1][3] = x[1].w; }
            if (a.logits_out != nullptr && oh == 0) {
                bf16_t* dst = a.logits_out + ((long)cb * a.M + ob) * a.V;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i0 = tid * 8 + j;
                    if (i0 < a.V) dst[i0] = (bf16_t)((j & 1) ? (wv[0][j >> 1] >> 16) : (wv[0][j >> 1] & 0xffffu));
                    const int i1 = 2048 + j;
                    if (tid == 0 && i1 < a.V) dst[i1] = (bf16_t)((j & 1) ? (wv[1][j >> 1] >> 16) : (wv[1][j >> 1] & 0xffffu));
                }
            }
            // the sampler's scratch aliases my half's activation buffer: every phase of this step must have read it
            if (!dm_wait_ge(cdone + ohf, 4u * kf, ab, a.err, 0xB60u, lane)) return;
            char* scr = lds + DM_L_XB + ohf * 32768;
            SampleScratch sc;
            sc.cand_t = (lds_f32_t*)(scr + DM_L_CANDT); sc.cand_i = (lds_i32_t*)(scr + DM_L_CANDI); sc.s_max = (lds_u32_t*)(lds + DM_L_SMAX); sc.cand_q = (lds_f32_t*)(lds + DM_L_SMAX);
            sc.s_bv = (lds_f32_t*)(misc + DM_M_SBV); sc.s_bi = (lds_i32_t*)(misc + DM_M_SBI); sc.s_n = (lds_i32_t*)(misc + DM_M_SN);
            sc.s_tok = (lds_i32_t*)(misc + DM_M_STOK); sc.s_wtot = (lds_i32_t*)(misc + DM_M_SWTOT);
            const DpQuadSync& sync = qsync;
            const uint64_t seed = (uint64_t)misc[DM_M_RNG] | ((uint64_t)misc[DM_M_RNG + 1] << 32), step = (uint64_t)misc[DM_M_RNG + 2] | ((uint64_t)misc[DM_M_RNG + 3] << 32);
            const int sV = (int)misc[DM_M_SARG], sK = (int)misc[DM_M_SARG + 2];
            const float sT = __uint_as_float(misc[DM_M_SARG + 1]);
            const bool sN = misc[DM_M_SARG + 3] != 0u;
            const int tok = sample_body<2>(wv, sV, sT, sK, sN ? a.noise + ((long)cb * a.M + ob) * sV : nullptr, seed, step, ob, cb, sc, tid, sync);
            if (*ab) return;
            if (gw == 0) DM_STAMP(34);
            int fed = a.forced ? a.forced[(long)ob * a.ncb + cb] : tok;
            fed = min(max(fed, 0), a.V - 1);
            if (gw == 0) {
                if (oh == 0 && lane == 0) {
                    a.frame[(long)ob * a.ncb + cb] = tok;
                    dm_st4(a.xchg + DM_OFF_T + (s % DM_R) * DM_T_BYTES + ob * 4, (uint32_t)fed);
                    dm_st4(a.xchg + DM_OFF_T + ((s + 2) % DM_R) * DM_T_BYTES + ob * 4, 0xffffffffu);
                }
                if (cb + 1 < a.ncb) {
                    // the next step's layer-0 q / k / v of my (row, head): one table row per fed token
                    const bf16_t* qrow = a.qkv0_tab + ((long)(cb - 1) * a.V + fed) * DP_NQKV;
                    const int ln = min(lane, 47);
                    const bf16_t* src = ln < 16 ? qrow + oh * 128 + ln * 8 : (ln < 32 ? qrow + 1024 + kvh * 128 + (ln - 16) * 8 : qrow + 1280 + kvh * 128 + (ln - 32) * 8);
                    const uint4 v = *reinterpret_cast<const uint4*>(src);
                    if (lane < 16) dp_stq((dp_lu4*)(lds + DM_L_QB) + lane, v);
                    else if (lane < 32) dp_stq((dp_lu4*)(lds + DM_L_K + (cb + 1) * 256) + lane - 16, v);
                    else if (lane < 48) dp_stq((dp_lu4*)(lds + DM_L_V + (cb + 1) * 256) + lane - 32, v);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    DM_STAMP(35);
                }
            }
        }
        if (cb + 1 < a.ncb && gw == 2) {
            // every row's fed token -> my 4 residual columns of the next step's input rows, half by half
#pragma unroll 1
            for (int hf = 0; hf < NH; ++hf) {
                const int b = 16 * hf + (lane & 15), bc = min(b, a.M - 1);
                uint32_t tw;
                if (!dm_poll4(a.xchg + DM_OFF_T + (s % DM_R) * DM_T_BYTES + bc * 4, tw, lane, ab, a.err, 0x910u, ps)) return;
                const int tk = min(max((int)tw, 0), a.V - 1);
                typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
                const u32x2v hv = *reinterpret_cast<const u32x2v*>(a.proj_emb + ((long)cb * a.V + tk) * DP_D + 4 * cu);
                if (lane < 16 && b < a.M) { dp_lu32* hr = (dp_lu32*)(lds + DM_L_HRES) + 2 * b; hr[0] = hv.x; hr[1] = hv.y; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) misc[DM_M_FT + hf] = (uint32_t)(s + 2);
            }
            DM_STAMP(36);
        }
    }
#undef DM_FILL_BEGIN
#undef DM_FILL_END
}

template <int NH>
__global__ __launch_bounds__(512) void k_dec_persist_m(const DecPersistMArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    const int lane = threadIdx.x & 63;
    dp_lu32* misc = (dp_lu32*)(lds + DM_L_MISC);
    const int ob = cu >> 3, oh = cu & 7, kvh = oh >> 2;
    for (int i = threadIdx.x; i < 256 + 128; i += 512) misc[i] = 0;      // misc words, HRES, HRES1 (rows >= M must be finite)
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) dp_stq((dp_lu4*)lds + i, make_uint4(0, 0, 0, 0));     // K / V: dead slots must be finite
    __syncthreads();
    {
        if (ob < a.M) {
            const int npos = a.cb_first + 1;
            for (int i = threadIdx.x; i < DP_NL * npos * 16; i += 512) {
                const int c = i & 15, pos = (i >> 4) % npos, l = (i >> 4) / npos;
                const long src = (long)l * a.kv_layer_stride + (((long)ob * 2 + kvh) * 32 + pos) * DP_HD;
                dp_stq((dp_lu4*)(lds + DM_L_K + (l * 32 + pos) * 256) + c, reinterpret_cast<const uint4*>(a.kc + src)[c]);
                dp_stq((dp_lu4*)(lds + DM_L_V + (l * 32 + pos) * 256) + c, reinterpret_cast<const uint4*>(a.vc + src)[c]);
            }
            if (threadIdx.x < 16) dp_stq((dp_lu4*)(lds + DM_L_QB) + threadIdx.x, reinterpret_cast<const uint4*>(a.qd + (long)ob * 1024 + oh * 128)[threadIdx.x]);
        }
        if (threadIdx.x >= 64 && threadIdx.x < 64 + 2 * a.M) {
            const int i = threadIdx.x - 64;                                       // dword i & 1 of row i >> 1
            ((dp_lu32*)(lds + DM_L_HRES))[i] = reinterpret_cast<const uint32_t*>(a.hdec + (long)(i >> 1) * DP_D + 4 * cu)[i & 1];
        }
        if (threadIdx.x >= 128 && threadIdx.x < 128 + 96) {
            const int i = threadIdx.x - 128, pos = i / 3, u = i % 3;
            const int row = 2 * (3 * cu + u), e = (row < 1024 ? row : row - 1024) % DP_HD;
            ((dp_lu32*)(lds + DM_L_ROPE))[i] = reinterpret_cast<const uint32_t*>(a.rope)[pos * (DP_HD / 2) + e / 2];
        }
        if (threadIdx.x >= 256 && threadIdx.x < 260) misc[DM_M_RNG + threadIdx.x - 256] = a.rng ? reinterpret_cast<const uint32_t*>(a.rng)[threadIdx.x - 256] : 0u;
        if (threadIdx.x == 320) {
            misc[DM_M_SARG] = (uint32_t)a.V; misc[DM_M_SARG + 1] = __float_as_uint(a.temperature); misc[DM_M_SARG + 2] = (uint32_t)a.topk; misc[DM_M_SARG + 3] = a.noise != nullptr;
            misc[DM_M_FT] = 1u; misc[DM_M_FT + 1] = 1u;
        }
        for (int i = threadIdx.x; i < 9 * 128; i += 512) {
            const int nrm = i >> 7, c = i & 127;
            const bf16_t* src = nrm < 8 ? a.norms + (long)nrm * DP_D : a.dec_norm;
            dp_stq((dp_lu4*)(lds + DM_L_NORM + nrm * 2048) + c, reinterpret_cast<const uint4*>(src)[c]);
        }
    }
    __syncthreads();
#ifndef DM_ONLY_ROLE
#define DM_ONLY_ROLE 0            // (register-pressure probes: 1 = compute waves only, 2 = gather waves only)
#endif
    if (wave < 4) { if (DM_ONLY_ROLE != 2) dm_compute_wave<NH>(a, lds, wave, lane, cu); }
    else if (DM_ONLY_ROLE != 1) {
        __builtin_amdgcn_s_setprio(2);
        dm_gather_wave<NH>(a, lds, wave - 4, lane, cu);
    }
}

// W1, W3 [8192][1024] -> [256 cu][4 tiles][32 k steps][64 lanes] A-operand pieces: tile q of workgroup cu = pairs cu*32 + 8q .. + 7 as
// rows (gate, up, gate, up, ...), so lane group g4 of the 16 x 16 result holds (gate, up) of pairs 2 g4 and 2 g4 + 1
__global__ void k_dm_pack_gateup(const bf16_t* w1, const bf16_t* w3, uint4* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DM_W13M_U4) return;
    const int lane = (int)(i & 63), t = (int)((i >> 6) & 31), q = (int)((i >> 11) & 3), c = (int)(i >> 13);
    const int r = lane & 15, pair = c * 32 + 8 * q + (r >> 1);
    const bf16_t* w = (r & 1) ? w3 : w1;
    out[i] = *reinterpret_cast<const uint4*>(w + (long)pair * DP_D + 32 * t + 8 * (lane >> 4));
}
// W2 [1024][8192] -> [256 cu][4 tiles][16 k steps][64 lanes]: workgroup cu = 16 j + g, tile q = output rows 64 j + 16 q .. + 15, k step t =
// the 32 ffn columns of workgroup 16 t + g (the order its group gathers h in)
__global__ void k_dm_pack_down(const bf16_t* w2, uint4* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DM_W2M_U4) return;
    const int lane = (int)(i & 63), t = (int)((i >> 6) & 15), q = (int)((i >> 10) & 3), c = (int)(i >> 12);
    const int j = c >> 4, g = c & 15;
    const int row = 64 * j + 16 * q + (lane & 15), col = 32 * (16 * t + g) + 8 * (lane >> 4);
    out[i] = *reinterpret_cast<const uint4*>(w2 + (long)row * DP_FFN + col);
}
