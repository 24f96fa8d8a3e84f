// Depth-decoder steps 2..31 of ONE frame as ONE persistent launch (batch 1, the CSM-1B decoder shape: 4 layers,
// d 1024, 8 heads / 2 KV heads of 128, ffn 8192, 32 positions).   reference: sesameai/models.py:165-182.
//
// Why (measured, tools/microbench/persist*_bench.hip, profiles/r02/): as a chain of launches a decoder layer costs
// 4 seams x ~4.6 us because the HBM/Infinity-Cache stream of a kernel cannot start before the previous kernel has
// ended; the weights do not depend on the activations, so here every workgroup (one per CU, 256 of them) keeps the
// NEXT use of its weight rows in flight in VGPRs while the dependency chain runs, and the seams become 8-byte
// {tag, payload} granule all-gathers polled by one wave per CU that has no weight loads outstanding.
//
//   wave 7      "gather": sweeps the granules of the previous op (relaxed agent-scope loads, 8 copies so that only 32
//               CUs poll a line), applies the RMSNorm the consumer needs, writes LDS, raises an LDS flag; it also owns
//               rows 4cu..4cu+3 of the residual stream: sums the 256 down-projection partials of those rows in a fixed
//               order, adds the residual and publishes them; after sampling it fetches the next step's table rows.
//   waves 0..6  "compute": fixed weight rows per wave; the loads of a wave's next use are issued ONE 1 KB wave load at
//               a time between polls of the LDS flag the wave waits on anyway (static order -> exact vmcnt(N) waits).
//               waves 0-3: one 16-row tile of the gate/up projection (8 (gate, up) pairs on v_mfma_f32_16x16x32_bf16,
//                          weights pre-packed in operand order) + 2 row blocks of the split down projection
//               waves 4-6: 3 / 3 / 2 row blocks of the split down projection
//               waves 2-4: + one q|k|v unit (2 rows, RoPE) + one head unit
//               waves 5,6: + one o-proj unit (2 rows, + residual) + head unit (5) / tail rows (6)
//               all 8 waves: one attention head each over the LDS-resident K/V;  waves 4-7 ("quad"): the sampler.
//               Every wave of an instantiation issues the same load sequence (dp_compute_wave); the per-layer weights are
//               four buffers with a constant layer stride (DecPersistArgs), addressed by arithmetic.
//
// Arithmetic: every dot product, RoPE, attention, RMSNorm, residual and the sampler use the chain path's code or its
// exact lane/k mapping (k_gemv, stage_attn, stage_x, sample_body), so those values are bit-identical to the launch
// chain.  TWO things differ, both only in fp32 summation order: the gate/up dot products run on the matrix cores
// (16 rows x 32 k per instruction instead of a lane-strided dot2 chain), and the down projection is split over the
// 256 workgroups' 32-column slices (its input never leaves the CU that produced it) and summed by the row owner in a
// fixed order.  Deterministic; bf16 rounding points unchanged.  Every spin is bounded (s_memrealtime); on a timeout the kernel sets *err and
// leaves, and csm_read_frames reports it.
#pragma once
#include "gemv.cuh"
#include "sampler.cuh"

typedef unsigned long long dp_u64;
#define DP_NB 256                     // workgroups = CUs of an MI355X
#ifndef DP_NREP
#define DP_NREP 8                     // granule copies (one per XCD under round-robin placement: the 32 workgroups of an XCD poll the same lines;
                                      //  round 4 A/B in the product: 4 copies k_dec_persist 2,144 us, 8: 2,075, 16: 2,221)
#endif
#define DP_TIMEOUT 5000000ull         // 50 ms of s_memrealtime (100 MHz)
#define DP_NL 4
#define DP_D 1024
#define DP_FFN 8192
#define DP_HD 128
#define DP_NQKV 1536
#define DP_WSM_ROWS 2560              // q|k|v rows + o-proj rows of a layer
#define DP_W2S_U4 (256L * 4 * 1024)    // 16-byte pieces per layer of the re-tiled W2
#define DP_W13P_U4 (256L * 4 * 32 * 64)   // ... of the packed W1 | W3
#define DP_LSLOTS 1088                // logit granules per copy (1026 used)

typedef __attribute__((address_space(3))) uint32_t dp_lu32;
typedef __attribute__((address_space(3))) volatile uint32_t dp_lvu32;
typedef __attribute__((address_space(3))) u32x4_t dp_lu4;
typedef __attribute__((address_space(3))) unsigned short dp_lu16;
typedef __attribute__((address_space(3))) float dp_lf32;
// One uniform dword through the SCALAR cache (s_load_dword), whatever the compiler concludes about aliasing.  A plain `*p` of a
// wave-uniform address becomes s_load_dword only when hipcc can prove that nothing in the kernel wrote the location before; when it
// cannot it emits global_load_dword, whose result comes back IN ORDER behind every weight load the wave has already issued -- the
// backbone layer then sees its position / epoch / residual words several us late (measured: 32.2 instead of 29.9 us per launch,
// flipped by an unrelated edit).  Use only for words no workgroup of the SAME launch writes before this read.
typedef __attribute__((address_space(4))) const uint32_t dp_cu32;
__device__ __forceinline__ uint32_t dp_sload32(const void* p) {
    return *(dp_cu32*)(unsigned long long)p;
}
__device__ __forceinline__ uint4 dp_ldq(const dp_lu4* p) { const u32x4_t v = *p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void dp_stq(dp_lu4* p, const uint4& v) { u32x4_t t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; *p = t; }

struct DecPersistArgs {
    // Per-layer weights live in FOUR buffers with a constant layer stride, so the kernel addresses them by arithmetic.  (With
    // per-layer pointers in this struct, a run-time layer index into it made hipcc copy the whole argument block to scratch.)
    const bf16_t* wsm;                // [4 layers][2560 rows][1024]: rows 0..1535 = wq | wk | wv, 1536..2559 = wo
    const bf16_t* norms;              // [4][2][1024]: sa_norm | mlp_norm
    const uint4* w2s;                 // [4] x W2 re-tiled [256 cu][4 k chunks][1024 rows] 16-byte pieces (k_dp_retile_w2)
    const uint4* w13p;                // [4] x W1 | W3 in matrix-core operand order [256 cu][4 tiles][32 k steps][64 lanes] (k_dp_pack_gateup)
    const bf16_t* dec_norm;
    const bf16_t* head_t;             // [ncb-1][V][1024]
    const bf16_t* rope;               // [max_seq][64][2]
    const bf16_t* proj_emb;           // [ncb*V][1024]
    const bf16_t* qkv0_tab;           // [(ncb-2)*V][1536]
    const bf16_t *hdec, *qd;          // decoder input row / layer-0 q of step cb_first (written by the chain's cb = 1 step)
    const bf16_t *kc, *vc;            // decoder caches [L][1][2][32][128] (positions < cb_first of every layer, cb_first of layer 0)
    long kv_layer_stride;
    float temperature; int topk;
    const bf16_t* noise;              // optional [ncb][1][V]
    const uint64_t* rng;
    const int* forced;                // optional [ncb]
    int V, ncb;
    int* frame;                       // [ncb]
    bf16_t* logits_out;               // optional [ncb][1][V]
    int cb_first, cb_last;
    dp_u64 *gQ, *gH1, *gH2, *gL, *gP; // granule slots: 8x768, 8x512, 8x512, 8x1088, 256x1024
    uint32_t* err;
    uint32_t* epoch;
    float eps;
    int trickle_sleep, poll_sleep;
    int fault;                        // timeline build only (CSM_PERSIST_FAULT=1): see DP_FAULT
    dp_u64* stamps;                   // optional (csm_debug_persist_stamps): s_memrealtime of workgroup 100's gather wave, [step][32]
};

// LDS image (dynamic shared memory; byte offsets)
// Debug timeline (tools/persist_timeline.py): compiled in only with -DDP_TIMELINE (libcsm_hip_timeline.so).  In the product
// build every stamp folds away -- ~170 sites whose stores otherwise put waits and SGPR pressure into the compute waves.
#ifdef DP_TIMELINE
#define DP_STAMPS(a_) ((a_).stamps)
#define DP_FAULT(a_) ((a_).fault)            // fault injection (tests): one workgroup withholds one hand-off granule
#else
#define DP_STAMPS(a_) ((dp_u64*)nullptr)
#define DP_FAULT(a_) 0
#endif
#define DP_OFF_K 0                                   // [4][2][32][128] bf16
#define DP_OFF_V 65536
#define DP_OFF_U 131072                              // layers: xA | xC | q | att (2 KB each); sampling: cand_t | cand_i (8448 B each)
#define DP_OFF_XA (DP_OFF_U)
#define DP_OFF_XC (DP_OFF_U + 2048)
#define DP_OFF_QB (DP_OFF_U + 4096)
#define DP_OFF_ATT (DP_OFF_U + 6144)
#define DP_CAND_SLOTS 2112                           // entries of each sampler candidate list (csm_create admits audio_vocab <= this)
#define DP_OFF_CANDT (DP_OFF_U)
#define DP_OFF_CANDI (DP_OFF_U + DP_CAND_SLOTS * 4)
#define DP_OFF_LOGITS (DP_OFF_U + 16896)             // 2560 bf16
#define DP_OFF_SMAX (DP_OFF_LOGITS + 5120)           // 256 u32
#define DP_OFF_PS (DP_OFF_SMAX + 1024)               // attention P rows: 8 waves x 32 floats
#define DP_OFF_MISC (DP_OFF_PS + 1024)
#define DP_LDS_BYTES (DP_OFF_MISC + 512)
static_assert(DP_OFF_CANDI + DP_CAND_SLOTS * 4 <= DP_OFF_LOGITS, "sampler candidate lists overrun the LDS logits");
// misc words
#define DP_M_HL 0        // 16 words: this CU's 32 h values
#define DP_M_H0 16       // 2 words: residual rows 4cu..4cu+3 entering the layer
#define DP_M_H1 18       // 2 words: after the o-projection
#define DP_M_FXA 20      // flags
#define DP_M_FQ 21
#define DP_M_FXC 22
#define DP_M_FLG 23
#define DP_M_FTOK 24
#define DP_M_ATTN 25     // counters
#define DP_M_CD 26
#define DP_M_BAR 27
#define DP_M_ABORT 28
#define DP_M_TOK 29
#define DP_M_SBV 32      // sampler: 4 floats, 4 ints, n, tok, 4 wave totals
#define DP_M_SBI 36
#define DP_M_SN 40
#define DP_M_STOK 41
#define DP_M_SWTOT 42
#define DP_M_RNG 112     // 4 words: Philox {seed, step} of this frame (read once at kernel start: a global load at sampling time would
                         // wait behind every weight load the wave has in flight)
#define DP_M_SARG 116    // 4 words: V, temperature (bits), top-k, 1 if a noise tensor was given -- the sampler's scalars, staged once
#define DP_M_TILE 48     // 4 tiles x 16 floats: the gate | up sums of a tile on their way to the SwiGLU lanes

enum { DP_E_Q = 0, DP_E_H1 = 1, DP_E_P = 2, DP_E_H2 = 3, DP_E_L = 4 };
__device__ __forceinline__ uint32_t dp_tag(uint32_t base, int s, int l, int e) { return base + 1u + (uint32_t)((s * DP_NL + l) * 5 + e); }

__device__ __forceinline__ void dp_gran_store(dp_u64* p, uint32_t tag, uint32_t val) {
    __hip_atomic_store(p, ((dp_u64)tag << 32) | val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ dp_u64 dp_gran_load(const dp_u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool dp_give_up(dp_u64 t0, dp_lvu32* ab, uint32_t* err, uint32_t code, int lane) {
    if (*ab) return true;
    if (__builtin_amdgcn_s_memrealtime() - t0 > DP_TIMEOUT) {
        *ab = 1;
        if (lane == 0) atomicCAS(err, 0u, code);
        return true;
    }
    return false;
}

// wait until *f == tag (or, with `at_least`, *f >= tag) while issuing N loads one at a time in a static order
template <int N, bool AT_LEAST, class Issue>
__device__ __forceinline__ bool dp_wait(dp_lvu32* f, uint32_t tag, dp_lvu32* ab, uint32_t* err, uint32_t code, int lane, int sleep_units, Issue issue) {
    auto hit = [&]() { const uint32_t v = *f; return AT_LEAST ? (int32_t)(v - tag) >= 0 : v == tag; };
    // No branch on `seen` between the loads (a branch there gets jump-threaded into one clone of the remaining load
    // sequence per step): the sleep length is data (0 once the flag has been seen) and the LDS poll always runs.
    int seen = (int)hit();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        issue(k);
        const int nap = seen ? 0 : sleep_units;
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(1);
        seen |= (int)hit();
    }
    seen = __builtin_amdgcn_readfirstlane(seen);
    if (!seen) {
        const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (uint32_t spins = 1; !hit(); ++spins)
            if ((spins & 255u) == 0 && dp_give_up(t0, ab, err, code, lane)) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}

// gather wave: sweep the granules of one edge until every tag matches.  16-byte `sc1` loads (two granules per lane
// per instruction: a poll pass costs ~0.1 us per load instruction, so half the instructions of 8-byte polls), all
// issued before the one wait; load j of lane l covers granules 2 (j*64 + l) and the next one, clamped to the last
// valid pair (no per-load predicate: hipcc would serialise predicated loads).  v[2j], v[2j+1] = their payloads.
template <int NL>
__device__ __forceinline__ bool dp_sweep(const dp_u64* g, int n_valid, uint32_t tag, uint32_t (&v)[2 * NL], int lane, dp_lvu32* ab, uint32_t* err,
                                         uint32_t code, int poll_sleep, dp_u64* passes = nullptr) {
    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
    const int last_pair = (n_valid - 1) >> 1;
    for (;;) {
        if (passes && lane == 0) *passes += 1;
        u32x4_t x[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const dp_u64* p = g + 2 * min(j * 64 + lane, last_pair);
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(x[j]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            asm volatile("" : "+v"(x[j]));                 // the values exist only behind the wait above
            v[2 * j] = x[j].x; v[2 * j + 1] = x[j].z;
            const bool second = 2 * min(j * 64 + lane, last_pair) + 1 < n_valid;
            ok &= x[j].y == tag && (x[j].w == tag || !second);
        }
        if (__all(ok)) return true;
        if (dp_give_up(t0, ab, err, code, lane)) return false;
        for (int z = 0; z < poll_sleep; ++z) __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ void dp_flag(dp_lvu32* flag, uint32_t tag) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *flag = tag;
}

// stage_x<1, 2, true>'s RMSNorm arithmetic by one wave on a 1024-vector in LDS, in place (thread t < 128 of that
// kernel owns 16-byte chunk t: lane l here owns chunks l (its wave 0) and 64 + l (its wave 1))
__device__ __forceinline__ float dp_chunk_ss(const uint4& v) {
    float ss = 0.f, f;
    f = lo2f(v.x); ss += f * f; f = hi2f(v.x); ss += f * f;
    f = lo2f(v.y); ss += f * f; f = hi2f(v.y); ss += f * f;
    f = lo2f(v.z); ss += f * f; f = hi2f(v.z); ss += f * f;
    f = lo2f(v.w); ss += f * f; f = hi2f(v.w); ss += f * f;
    return ss;
}
__device__ __forceinline__ uint4 dp_chunk_norm(const uint4& v, const uint4& g, float r) {
    uint4 o;
    o.x = pack_bf(round_bf(lo2f(v.x) * r) * lo2f(g.x), round_bf(hi2f(v.x) * r) * hi2f(g.x));
    o.y = pack_bf(round_bf(lo2f(v.y) * r) * lo2f(g.y), round_bf(hi2f(v.y) * r) * hi2f(g.y));
    o.z = pack_bf(round_bf(lo2f(v.z) * r) * lo2f(g.z), round_bf(hi2f(v.z) * r) * hi2f(g.z));
    o.w = pack_bf(round_bf(lo2f(v.w) * r) * lo2f(g.w), round_bf(hi2f(v.w) * r) * hi2f(g.w));
    return o;
}
__device__ __forceinline__ void dp_norm_in_lds(dp_lu4* xs, const uint4& g0, const uint4& g1, float eps, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint4 v0 = dp_ldq(xs + lane), v1 = dp_ldq(xs + 64 + lane);
    const float s0 = wave_sum(dp_chunk_ss(v0)), s1 = wave_sum(dp_chunk_ss(v1));
    const float tot = s0 + s1 + 0.f + 0.f;
    const float r = 1.0f / sqrtf(tot / 1024.0f + eps);
    dp_stq(xs + lane, dp_chunk_norm(v0, g0, r));
    dp_stq(xs + 64 + lane, dp_chunk_norm(v1, g1, r));
}

__device__ __forceinline__ uint32_t dp_resid_pair(float a0, float a1, uint32_t hw) {
#pragma clang fp contract(off)
    const float y0 = round_bf(a0) + lo2f(hw), y1 = round_bf(a1) + hi2f(hw);
    return pack_bf(y0, y1);
}
__device__ __forceinline__ uint32_t dp_swiglu(float ag, float au) {
#pragma clang fp contract(off)
    const float g = round_bf(ag), u = round_bf(au);
    const float s = round_bf(g / (1.0f + __expf(-g)));
    return (uint32_t)f2bf(s * u);
}
__device__ __forceinline__ uint32_t dp_rope_pair(float a0, float a1, uint32_t cs, bool rotate) {
#pragma clang fp contract(off)
    float v0 = round_bf(a0), v1 = round_bf(a1);
    if (rotate) {
        const float c = lo2f(cs), s = hi2f(cs);
        const float o0 = v0 * c - v1 * s;
        const float o1 = v1 * c + v0 * s;
        v0 = o0; v1 = o1;
    }
    return pack_bf(v0, v1);
}
__device__ __forceinline__ float dp_down_partial(const uint4 (&w)[4], const uint4 (&h)[4]) {
    float acc = dot8(w[0], h[0], 0.f);
    acc = dot8(w[1], h[1], acc);
    acc = dot8(w[2], h[2], acc);
    return dot8(w[3], h[3], acc);
}
// owner-side sum of the 256 partials of 4 rows (granule index producer*4 + row): lane l holds rows 2 (l & 1) and
// 2 (l & 1) + 1 of producers j*32 + (l >> 1), j = 0..7 (v[2j], v[2j+1]); sequential over j, then butterflies over the
// lanes of equal parity.  Returns the two row totals of this lane's parity in every lane.
__device__ __forceinline__ void dp_reduce_partials(const uint32_t (&v)[16], float& r0, float& r1) {
    float s0 = __uint_as_float(v[0]), s1 = __uint_as_float(v[1]);
#pragma unroll
    for (int j = 1; j < 8; ++j) { s0 += __uint_as_float(v[2 * j]); s1 += __uint_as_float(v[2 * j + 1]); }
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); }
    r0 = s0; r1 = s1;
}

// stage_attn (gemv.cuh) for one row and ONE head, with q / K / V in LDS (that kernel walks two heads of a KV group
// per wave; per head the lane layout and the operation order are the same): lane l holds 16-byte pieces of keys
// 4i + l/16 (i = 0..7) at element offset 8 (l % 16).  Eight waves take one head each.
// NG = 4-key groups walked (8 = all 32 slots).  Groups beyond the live keys contribute exact zeros, so any NG with 4 NG >= nk gives the same bits
// (DP_ATTN_GROUPS build: dp_attention_wave picks the smallest of 1 / 2 / 4 / 8 for the step's key count).
template <int NG = 8>
__device__ __forceinline__ void dp_attention_head(const dp_lu4* qb, const dp_lu4* kt, const dp_lu32* vt, dp_lf32* myps, dp_lu32* att,
                                                  int h, int nk, float ascale, int lane) {
    // stage_attn skips whole 4-key groups beyond nk by uniform branches; here every group is walked (its LDS reads and
    // DPP chains then overlap instead of running one group after the other) and dead keys contribute exact zeros:
    // score -inf -> p = 0, v read as 0 -> "+ 0.0" leaves every running sum's bits unchanged.
    const int grp = lane >> 4, sub = lane & 15;
    const uint4 qa = dp_ldq(qb + h * 16 + sub);
    uint4 kv[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) kv[i] = dp_ldq(kt + i * 64 + lane);
    float s0[NG], mx0 = -INFINITY;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const bool live = (4 * i + grp) < nk;
        const float d0 = row16_sum(dot8(qa, kv[i], 0.f)) * ascale;
        s0[i] = live ? d0 : -INFINITY;
        mx0 = fmaxf(mx0, s0[i]);
    }
    mx0 = wave_max(mx0);
    float l0 = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        s0[i] = (s0[i] == -INFINITY) ? 0.f : __expf(s0[i] - mx0);
        l0 += s0[i];
        if (sub == 0) myps[4 * i + grp] = s0[i];
    }
    l0 = wave_sum(l0) * (1.0f / 16.0f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float o00 = 0.f, o01 = 0.f;
#pragma unroll
    for (int t4 = 0; t4 < NG; ++t4) {                        // 4 keys at a time: one broadcast read of their p, four reads of their v
        const u32x4_t p4 = *reinterpret_cast<const __attribute__((address_space(3))) u32x4_t*>(myps + 4 * t4);
        const float pw[4] = {__uint_as_float(p4.x), __uint_as_float(p4.y), __uint_as_float(p4.z), __uint_as_float(p4.w)};
        uint32_t vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vr[u] = vt[(4 * t4 + u) * 64 + lane];
#pragma unroll
        for (int u = 0; u < 4; ++u) { o00 += pw[u] * lo2f(vr[u]); o01 += pw[u] * hi2f(vr[u]); }
    }
    const float i0 = 1.0f / l0;
    att[h * 64 + lane] = pack_bf(o00 * i0, o01 * i0);
}
// head `wave` of layer l, keys 0..cb, then one arrival on the attention counter
__device__ __forceinline__ void dp_attention_wave(char* lds, int wave, int l, int cb, int lane) {
    dp_lu32* misc = (dp_lu32*)(lds + DP_OFF_MISC);
    const int kvh = wave >> 2;
    const dp_lu4* qb = (const dp_lu4*)(lds + DP_OFF_QB);
    const dp_lu4* kt = (const dp_lu4*)(lds + DP_OFF_K + ((l * 2 + kvh) * 32) * 256);
    const dp_lu32* vt = (const dp_lu32*)(lds + DP_OFF_V + ((l * 2 + kvh) * 32) * 256);
    dp_lf32* ps = (dp_lf32*)(lds + DP_OFF_PS) + wave * 32;
    dp_lu32* att = (dp_lu32*)(lds + DP_OFF_ATT);
#ifdef DP_ATTN_GROUPS
    // the step's cb + 1 live keys sit in the first ceil((cb + 1) / 4) groups: walk 1 / 2 / 4 / 8 of them (uniform branch; same bits, see dp_attention_head)
    if (cb < 4) dp_attention_head<1>(qb, kt, vt, ps, att, wave, cb + 1, 0.08838834764831845f, lane);
    else if (cb < 8) dp_attention_head<2>(qb, kt, vt, ps, att, wave, cb + 1, 0.08838834764831845f, lane);
    else if (cb < 16) dp_attention_head<4>(qb, kt, vt, ps, att, wave, cb + 1, 0.08838834764831845f, lane);
    else
#endif
    dp_attention_head<8>(qb, kt, vt, ps, att, wave, cb + 1, 0.08838834764831845f, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(misc + DP_M_ATTN, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 4-wave barrier of the quad (waves 2..5) on an LDS counter: phase-numbered, bounded
struct DpQuadSync {
    dp_lvu32* ctr; dp_lvu32* ab; uint32_t* err; int lane; uint32_t* phase; dp_u64* stamp;
    __device__ __forceinline__ void mark(int i) const { if (stamp != nullptr && lane == 0) stamp[i] = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void operator()() const {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint32_t want = 4u * (++*phase);
        if (lane == 0) __hip_atomic_fetch_add((dp_lu32*)ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (uint32_t spins = 1; (int32_t)(*ctr - want) < 0; ++spins) {
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 255u) == 0 && dp_give_up(t0, ab, err, 0xB00u, lane)) break;
        }
        asm volatile("" ::: "memory");
    }
};

// ---------------------------------------------------------------------------------------------------------------
// compute wave.  Roles: X = waves 0,1;  A = waves 2,3,4 (q|k|v unit; 2,3,4 are in the quad);  B = waves 5,6 (o-proj
// unit; 5 is in the quad).
// ---------------------------------------------------------------------------------------------------------------
// one sampling step by the 4 waves of the quad (qw = 0..3): sample_body on the logits in LDS; returns the code
template <bool DEBUG_OUT>
__device__ __forceinline__ int dp_sample_step(const DecPersistArgs& a, char* lds, int qw, int lane, int cu, int cb, int s, uint32_t* quad_phase) {
    dp_lu32* misc = (dp_lu32*)(lds + DP_OFF_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DP_M_ABORT);
    const int tid = qw * 64 + lane;
    if (DP_STAMPS(a) != nullptr && cu == 100 && tid == 0) DP_STAMPS(a)[s * 32 + 22] = __builtin_amdgcn_s_memrealtime();
    const dp_lu4* lg = (const dp_lu4*)(lds + DP_OFF_LOGITS);
    uint32_t w[2][4];
    {
        const uint4 v0 = dp_ldq(lg + tid);
        w[0][0] = v0.x; w[0][1] = v0.y; w[0][2] = v0.z; w[0][3] = v0.w;
        const uint4 v1 = tid < 64 ? dp_ldq(lg + 256 + tid) : make_uint4(0, 0, 0, 0);
        w[1][0] = v1.x; w[1][1] = v1.y; w[1][2] = v1.z; w[1][3] = v1.w;
    }
    if (DEBUG_OUT && a.logits_out != nullptr && cu == 0) {       // (gather wave only: its stores would put a vmcnt(0) in a compute wave's path)
        for (int i = lane; i < a.V; i += 64) a.logits_out[(long)cb * a.V + i] = ((const dp_lu16*)lg)[i];
    }
    SampleScratch sc;
    sc.cand_t = (lds_f32_t*)(lds + DP_OFF_CANDT); sc.cand_i = (lds_i32_t*)(lds + DP_OFF_CANDI); sc.s_max = (lds_u32_t*)(lds + DP_OFF_SMAX); sc.cand_q = (lds_f32_t*)(lds + DP_OFF_SMAX);
    sc.s_bv = (lds_f32_t*)(misc + DP_M_SBV); sc.s_bi = (lds_i32_t*)(misc + DP_M_SBI); sc.s_n = (lds_i32_t*)(misc + DP_M_SN);
    sc.s_tok = (lds_i32_t*)(misc + DP_M_STOK); sc.s_wtot = (lds_i32_t*)(misc + DP_M_SWTOT);
    DpQuadSync sync{(dp_lvu32*)(misc + DP_M_BAR), ab, a.err, lane, quad_phase, DP_STAMPS(a) != nullptr && cu == 100 && qw == 0 ? DP_STAMPS(a) + 4352 + s * 16 : nullptr};
    const uint64_t seed = (uint64_t)misc[DP_M_RNG] | ((uint64_t)misc[DP_M_RNG + 1] << 32), step = (uint64_t)misc[DP_M_RNG + 2] | ((uint64_t)misc[DP_M_RNG + 3] << 32);
    // (the sampler's scalars from LDS: as kernel arguments they are re-read from the kernarg segment here, a scalar-cache miss per step)
    const int sV = (int)misc[DP_M_SARG], sK = (int)misc[DP_M_SARG + 2];
    const float sT = __uint_as_float(misc[DP_M_SARG + 1]);
    const bool sN = misc[DP_M_SARG + 3] != 0u;
    int tok = 0;
#ifdef DP_TIMELINE
    const int reps = (a.trickle_sleep & 128) ? 2 : 1;          // timeline experiment: a second pass of the sampler on the same logits
#else
    constexpr int reps = 1;
#endif
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        sync.mark(6);
        tok = sample_body<2>(w, sV, sT, sK, sN ? a.noise + (long)cb * sV : nullptr, seed, step, 0, cb, sc, tid, sync);
        sync.mark(7);
    }
    if (DP_STAMPS(a) != nullptr && cu == 100 && tid == 0) DP_STAMPS(a)[s * 32 + 23] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && cu == 0) a.frame[cb] = tok;
    return tok;
}

typedef __attribute__((ext_vector_type(8))) __bf16 dp_bf16x8;
typedef __attribute__((ext_vector_type(4))) float dp_f32x4;
// Two instantiations only -- <true, 2>: waves 0..3 (a gate/up tile + 2 row blocks), <false, 3>: waves 4..6 (3 row blocks, the
// sampler) -- with the small-op role a RUNTIME property of the wave (the kernel is instruction-cache bound: five instantiations were
// 131 KB of code, and sharing one between waves 5 and 6 alone was worth 1.3 % of the frame).  So that hipcc's wait counts stay exact
// every wave of an instantiation issues the SAME sequence of loads: a role that has no use for a load of the sequence re-reads an
// address it reads anyway (cache hit, no HBM bytes).
//   role X (waves 0, 1): no small op;  A (2, 3, 4): q|k|v unit, rows in wsa;  B (5, 6): o-proj unit, rows in wsb.
#ifndef DP_ROLES
#define DP_ROLES 4              // instantiations of the compute wave: 4 = every role its own (measured best), 3 / 2 = roles by wave index at run time
#endif
template <bool HAS_TILE, int NBK, int TROLE>      // TROLE (tile waves): 0 = role by wave index at run time, 1 = X only, 2 = A only
__device__ __forceinline__ void dp_compute_wave(const DecPersistArgs& a, char* lds, const int wave, const unsigned lane, const int cu, const uint32_t base,
                                                const uint32_t ropev) {
    constexpr bool HAS_B = !HAS_TILE;                 // the instantiation carries o-proj waves
    const bool is_x = HAS_TILE && (TROLE == 1 || (TROLE == 0 && wave < 2)), is_a = HAS_TILE ? !is_x : (TROLE == 1 || (TROLE == 0 && wave == 4)), is_b = HAS_B && !is_a;
    constexpr int NT = HAS_TILE ? 32 : 0, NCD = NT + NBK * 4, N1 = NCD / 3;
    dp_lu32* misc = (dp_lu32*)(lds + DP_OFF_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DP_M_ABORT);
    const int ts = a.trickle_sleep & 63;
    const int unit = is_a ? cu * 3 + (wave - 2) : cu * 2 + (wave - 5);            // q|k|v unit (768) or o-proj unit (512); X: unused
    // row blocks (64 rows) of the split down projection: waves 0-3 own {w, w + 4}; 4: {8, 11, 14}; 5: {9, 12, 15}; 6: {10, 13, 13 again}
    auto my_block = [&](int b) { return wave < 4 ? wave + 4 * b : min((wave + 4) + 3 * b, wave == 6 ? 13 : 15); };
    // head rows of this wave: waves 2..5 own unit cu*4 + (wave - 2); wave 6 of CU 0 / 1 owns the tail units 1024 / 1025
    const int hunit = is_x ? -1 : (wave < 6 ? cu * 4 + (wave - 2) : (cu < 2 ? 1024 + cu : -1));
    const int hrow0 = hunit < 0 ? 0 : 2 * hunit, hrow1 = hunit < 0 ? 0 : min(2 * hunit + 1, a.V - 1);
    uint4 wsa[2][2], wsb[HAS_B ? 2 : 1][2];
    uint4 wt[HAS_TILE ? 32 : 1];
    uint4 wd[NBK][4];

    // rows of the small op slot `slot`: 0..3 = this wave's layer op, 4 = the head of codebook step cb.  (X waves, and a slot
    // that is not this role's, read the head rows / the role's own rows: a valid address, nothing more.)
    auto ws_row = [&](int slot, int cb, int k) -> const bf16_t* {
        if (slot < DP_NL && !is_x) {
            if (is_a) {
                const int row = 2 * unit + (k >> 1);
                return a.wsm + ((long)slot * DP_WSM_ROWS + row) * DP_D;
            }
            return a.wsm + ((long)slot * DP_WSM_ROWS + DP_NQKV + 2 * unit + (k >> 1)) * DP_D;
        }
        return a.head_t + ((long)(cb - 1) * a.V + ((k >> 1) ? hrow1 : hrow0)) * DP_D;
    };
    auto load_wsa = [&](int slot, int cb, int k) { wsa[k >> 1][k & 1] = reinterpret_cast<const uint4*>(ws_row(slot, cb, k))[(k & 1) * 64 + lane]; };
    auto load_wsb = [&](int slot, int cb, int k) { wsb[HAS_B ? k >> 1 : 0][k & 1] = reinterpret_cast<const uint4*>(ws_row(slot, cb, k))[(k & 1) * 64 + lane]; };
    auto load_cd = [&](int l, int k) {
        if (k < NT) wt[HAS_TILE ? k : 0] = a.w13p[(long)l * DP_W13P_U4 + (((long)cu * 4 + wave) * 32 + k) * 64 + lane];
        else {
            const int kk = k - NT;
            wd[kk >> 2][kk & 3] = a.w2s[(long)l * DP_W2S_U4 + ((long)cu * 4 + (kk & 3)) * 1024 + my_block(kk >> 2) * 64 + lane];
        }
    };
    // first uses: A's q|k|v rows of layer 1 (layer 0's q|k|v come from the table), B's o-proj rows of layer 0
#pragma unroll
    for (int k = 0; k < 4; ++k) load_wsa(1, a.cb_first, k);
    if (HAS_B) {
#pragma unroll
        for (int k = 0; k < 4; ++k) load_wsb(0, a.cb_first, k);
    }
#pragma unroll
    for (int k = 0; k < NCD; ++k) load_cd(0, k);

    uint32_t quad_phase = 0;
    const int n_steps = a.cb_last - a.cb_first + 1;
    for (int s = 0; s < n_steps; ++s) {
        const int cb = a.cb_first + s;                 // this step's token sits at position cb; keys 0..cb
        for (int l = 0; l < DP_NL; ++l) {
            const int it = s * DP_NL + l;
            dp_lu32* att = (dp_lu32*)(lds + DP_OFF_ATT);
            {
                // -- x of the q|k|v units is ready (layers 1..3; layer 0's row comes from the table): A waves run their unit; first third
                //    of this layer's MLP weights meanwhile
                const uint32_t tg = dp_tag(base, s, l - 1, DP_E_H2);
                if (!dp_wait<N1, false>((dp_lvu32*)(misc + DP_M_FXA), l > 0 ? tg : *(dp_lvu32*)(misc + DP_M_FXA), ab, a.err, 0x910u, lane, ts,
                                        [&](int k) { load_cd(l, k); })) return;
                if (is_a && l > 0) {
                    const dp_lu4* xs = (const dp_lu4*)(lds + DP_OFF_XA);
                    const uint4 x0 = dp_ldq(xs + lane), x1 = dp_ldq(xs + 64 + lane);
                    float a0 = dot8(wsa[0][0], x0, 0.f); a0 = dot8(wsa[0][1], x1, a0);
                    float a1 = dot8(wsa[1][0], x0, 0.f); a1 = dot8(wsa[1][1], x1, a1);
                    a0 = wave_sum(a0); a1 = wave_sum(a1);
                    const int row = 2 * unit;                              // q rows 0..1023, k 1024..1279, v 1280..1535
                    const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)ropev, cb);   // (cos, sin) of this unit's pair at position cb
                    const uint32_t outw = dp_rope_pair(a0, a1, cs, row < 1280);
                    if (lane < DP_NREP) dp_gran_store(a.gQ + lane * 768 + unit, dp_tag(base, s, l, DP_E_Q), outw);
                }
            }
            {
                // -- attention: head `wave` over keys 0..cb (the gather wave takes head 7); the A waves' next q|k|v rows (or head rows)
                //    and the second third of this layer's MLP weights meanwhile
                if (!dp_wait<4 + N1, false>((dp_lvu32*)(misc + DP_M_FQ), dp_tag(base, s, l, DP_E_Q), ab, a.err, 0x920u, lane, ts, [&](int k) {
                        if (k < 4) load_wsa(l + 1 < DP_NL ? l + 1 : DP_NL, cb, k); else load_cd(l, N1 + k - 4);
                    })) return;
                const bool st5 = DP_STAMPS(a) != nullptr && cu == 100 && lane == 0 && wave == 5 && l == 2;
                if (st5) DP_STAMPS(a)[4096 + s * 8 + 0] = __builtin_amdgcn_s_memrealtime();
                dp_attention_wave(lds, wave, l, cb, lane);
                if (st5) DP_STAMPS(a)[4096 + s * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            }
            if (HAS_B && is_b) {
                // -- o-projection unit + residual, once the eight attention waves are done
                if (!dp_wait<1, true>((dp_lvu32*)(misc + DP_M_ATTN), 8u * (uint32_t)(it + 1), ab, a.err, 0x930u, lane, ts, [&](int) {})) return;
                const dp_lu4* xs = (const dp_lu4*)att;
                const uint4 x0 = dp_ldq(xs + lane), x1 = dp_ldq(xs + 64 + lane);
                float a0 = dot8(wsb[0][0], x0, 0.f); a0 = dot8(wsb[0][1], x1, a0);
                float a1 = dot8(wsb[HAS_B ? 1 : 0][0], x0, 0.f); a1 = dot8(wsb[HAS_B ? 1 : 0][1], x1, a1);
                a0 = wave_sum(a0); a1 = wave_sum(a1);
                const uint32_t h0w = *(dp_lvu32*)(misc + DP_M_H0 + (wave - 5));
                const uint32_t outw = dp_resid_pair(a0, a1, h0w);
                if (lane == 0) misc[DP_M_H1 + (wave - 5)] = outw;
                if (lane < DP_NREP && !(DP_FAULT(a) && s == 3 && l == 1 && cu == 17))
                    dp_gran_store(a.gH1 + lane * 512 + unit, dp_tag(base, s, l, DP_E_H1), outw);
                if (DP_STAMPS(a) != nullptr && cu == 100 && lane == 0 && wave == 5 && l == 2) DP_STAMPS(a)[4096 + s * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            }
            {
                // -- the MLP: my (gate, up) pairs -> h values -> LDS -> my row blocks of the split down projection
                // (the B waves' next o-proj rows, or head rows, and the last third of the MLP weights meanwhile)
                constexpr int NB4 = HAS_B ? 4 : 0, N4 = NB4 + NCD - 2 * N1;
                if (!dp_wait<N4, false>((dp_lvu32*)(misc + DP_M_FXC), dp_tag(base, s, l, DP_E_H1), ab, a.err, 0x940u, lane, ts, [&](int k) {
                        if (k < NB4) load_wsb(l + 1 < DP_NL ? l + 1 : DP_NL, cb, k); else load_cd(l, 2 * N1 + k - NB4);
                    })) return;
                const bool st0 = DP_STAMPS(a) != nullptr && cu == 100 && lane == 0 && wave == 0 && l == 2;
                if (st0) DP_STAMPS(a)[4096 + s * 8 + 3] = __builtin_amdgcn_s_memrealtime();
                if (HAS_TILE) {
                    // tile `wave`: rows 0..7 = gate rows of pairs 8 wave .. 8 wave + 7, rows 8..15 = their up rows; k step t covers
                    // k = 32t .. 32t+31: lane l feeds A[row l & 15][8 (l >> 4) + j] and, as every column of B, x[32t + 8 (l >> 4) + j]
                    const dp_lu4* xs = (const dp_lu4*)(lds + DP_OFF_XC);
                    dp_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    // x fragments four k steps at a time, the next four in flight behind the current four's matrix ops
                    // (one ds_read + wait per step would expose the LDS latency 32 times)
                    const dp_lu4* xq = xs + (lane >> 4);
                    uint4 xa[4], xb[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) xa[u] = dp_ldq(xq + 4 * u);
#pragma unroll
                    for (int tb = 0; tb < 8; tb += 2) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) xb[u] = dp_ldq(xq + 4 * (4 * (tb + 1) + u));
                        __builtin_amdgcn_sched_barrier(0);            // keep the four reads ahead of the four matrix ops
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
                    // every column equals column 0: lane l holds rows 4 (l >> 4) .. + 3 -> LDS -> lane i < 8 pairs gate i with up i
                    dp_lf32* tile = (dp_lf32*)(misc + DP_M_TILE) + wave * 16;
                    if ((lane & 15) == 0) {
                        tile[(lane >> 4) * 4 + 0] = acc[0]; tile[(lane >> 4) * 4 + 1] = acc[1];
                        tile[(lane >> 4) * 4 + 2] = acc[2]; tile[(lane >> 4) * 4 + 3] = acc[3];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane < 8) {
                        const uint32_t hv = dp_swiglu(tile[lane], tile[8 + lane]);
                        ((dp_lu16*)(misc + DP_M_HL))[wave * 8 + lane] = (unsigned short)hv;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(misc + DP_M_CD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                {
                    const uint32_t want = 7u * (uint32_t)(it + 1);
                    const dp_u64 t0 = __builtin_amdgcn_s_memrealtime();
                    for (uint32_t spins = 1; (int32_t)(*(dp_lvu32*)(misc + DP_M_CD) - want) < 0; ++spins)
                        if ((spins & 255u) == 0 && dp_give_up(t0, ab, a.err, 0x950u, lane)) return;
                    asm volatile("" ::: "memory");
                }
                if (st0) DP_STAMPS(a)[4096 + s * 8 + 4] = __builtin_amdgcn_s_memrealtime();
                uint4 h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) h[q] = dp_ldq((const dp_lu4*)(misc + DP_M_HL) + q);
#pragma unroll
                for (int b = 0; b < NBK; ++b) {
                    const int n = my_block(b) * 64 + lane;
                    const float p = dp_down_partial(wd[b], h);
                    dp_gran_store(a.gP + ((long)(n >> 2) * 256 + cu) * 4 + (n & 3), dp_tag(base, s, l, DP_E_P), __float_as_uint(p));
                }
                if (st0) DP_STAMPS(a)[4096 + s * 8 + 5] = __builtin_amdgcn_s_memrealtime();
            }
        }
        // ---- the head of codebook cb: waves 2..6 hold 2 logit rows each (ws slot 4), x = dec_norm(h)
        {
            if (!dp_wait<1, false>((dp_lvu32*)(misc + DP_M_FXA), dp_tag(base, s, DP_NL - 1, DP_E_H2), ab, a.err, 0x960u, lane, ts, [&](int) {})) return;
            const bool st2 = DP_STAMPS(a) != nullptr && cu == 100 && lane == 0 && wave == 2;
            if (st2) DP_STAMPS(a)[s * 32 + 20] = __builtin_amdgcn_s_memrealtime();
            if (hunit >= 0) {
                const dp_lu4* xs = (const dp_lu4*)(lds + DP_OFF_XA);
                const uint4 x0 = dp_ldq(xs + lane), x1 = dp_ldq(xs + 64 + lane);
                // the head rows sit in the role's own slot (A: wsa, B: wsb)
                uint4 r00 = wsa[0][0], r01 = wsa[0][1], r10 = wsa[1][0], r11 = wsa[1][1];
                if (HAS_B && is_b) { r00 = wsb[0][0]; r01 = wsb[0][1]; r10 = wsb[HAS_B ? 1 : 0][0]; r11 = wsb[HAS_B ? 1 : 0][1]; }
                float a0 = dot8(r00, x0, 0.f); a0 = dot8(r01, x1, a0);
                float a1 = dot8(r10, x0, 0.f); a1 = dot8(r11, x1, a1);
                a0 = wave_sum(a0); a1 = wave_sum(a1);
                if (lane < DP_NREP) dp_gran_store(a.gL + lane * DP_LSLOTS + hunit, dp_tag(base, s, DP_NL - 1, DP_E_L), pack_bf(a0, a1));
            }
            if (st2) DP_STAMPS(a)[s * 32 + 21] = __builtin_amdgcn_s_memrealtime();
            if (DP_STAMPS(a) != nullptr && s == 5 && lane == 0 && hunit >= 0) DP_STAMPS(a)[1024 + cu * 8 + wave] = __builtin_amdgcn_s_memrealtime();
            // next use of the small-op rows: the B waves' layer-0 o-proj of the next step (the A waves reload theirs in layer 0)
            if (HAS_B) {
#pragma unroll
                for (int k = 0; k < 4; ++k) load_wsb(0, cb, k);
            }
        }
        // ---- the sampler (waves 4, 5, 6 and the gather wave), on every CU alike: each CU needs the code for its table rows
        if (!HAS_TILE) {
            if (!dp_wait<1, false>((dp_lvu32*)(misc + DP_M_FLG), dp_tag(base, s, DP_NL - 1, DP_E_L), ab, a.err, 0x970u, lane, ts, [&](int) {})) return;
            (void)dp_sample_step<false>(a, lds, wave - 4, (int)lane, cu, cb, s, &quad_phase);
            if (*ab) return;
        }
    }
}

// (Emitted only by csm_dec_persist.hip -- a code object of its own: in the engine's 2.6 MB code object the same instructions ran 0.7 %
//  slower, 2090-2105 against 2074-2093 us per launch, three A/B pairs on one box.  csm_engine.hip defines CSM_DEC_PERSIST_ELSEWHERE.)
#ifndef CSM_DEC_PERSIST_ELSEWHERE
static __global__ __launch_bounds__(512) void k_dec_persist(const DecPersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cu = blockIdx.x;
    const unsigned lane = threadIdx.x & 63;
    dp_lu32* misc = (dp_lu32*)(lds + DP_OFF_MISC);
    dp_lvu32* ab = (dp_lvu32*)(misc + DP_M_ABORT);
    // RoPE (cos, sin) of the pair a q|k|v wave (2, 3, 4) rotates, for every position of this launch: lane p holds position
    // p's.  Read here, once: a global load inside the step would sit on the q|k|v edge and, vmcnt being in order, behind
    // every weight load the wave has in flight.
    uint32_t ropev = 0;
    if (wave >= 2 && wave <= 4 && lane < 32) {
        const int row = 2 * (cu * 3 + (wave - 2));
        const int e = (row < 1024 ? row : row - 1024) % DP_HD;
        ropev = reinterpret_cast<const uint32_t*>(a.rope)[lane * (DP_HD / 2) + e / 2];
    }
    // ---- LDS image of the step's starting state (all 512 threads) ----
    for (int i = threadIdx.x; i < 128; i += 512) misc[i] = 0;
    for (int i = threadIdx.x; i < 2560 / 2; i += 512) ((dp_lu32*)(lds + DP_OFF_LOGITS))[i] = 0;
    // every K / V slot starts as zeros: slots past the current position are then finite, and a dead key's exact-zero
    // probability times its (zero or stale-but-finite) value adds +0.0 -- no per-key select in the attention loop
    for (int i = threadIdx.x; i < 131072 / 16; i += 512) dp_stq((dp_lu4*)lds + i, make_uint4(0, 0, 0, 0));
    __syncthreads();
    {   // K/V rows of positions 0..cb_first (rows of later positions are written before they are used)
        const int npos = a.cb_first + 1;
        for (int i = threadIdx.x; i < DP_NL * 2 * npos * 16; i += 512) {
            const int c = i & 15, pos = (i >> 4) % npos, lk = (i >> 4) / npos;           // lk = layer * 2 + kv head
            const long src = (long)(lk >> 1) * a.kv_layer_stride + ((long)(lk & 1) * 32 + pos) * DP_HD;
            dp_stq((dp_lu4*)(lds + DP_OFF_K + (lk * 32 + pos) * 256) + c, reinterpret_cast<const uint4*>(a.kc + src)[c]);
            dp_stq((dp_lu4*)(lds + DP_OFF_V + (lk * 32 + pos) * 256) + c, reinterpret_cast<const uint4*>(a.vc + src)[c]);
        }
        for (int i = threadIdx.x; i < 128; i += 512) dp_stq((dp_lu4*)(lds + DP_OFF_QB) + i, reinterpret_cast<const uint4*>(a.qd)[i]);
        if (threadIdx.x < 2) misc[DP_M_H0 + threadIdx.x] = reinterpret_cast<const uint32_t*>(a.hdec)[2 * cu + threadIdx.x];
        if (threadIdx.x >= 64 && threadIdx.x < 68) misc[DP_M_RNG + threadIdx.x - 64] = a.rng ? reinterpret_cast<const uint32_t*>(a.rng)[threadIdx.x - 64] : 0u;
        if (threadIdx.x == 128) { misc[DP_M_SARG] = (uint32_t)a.V; misc[DP_M_SARG + 1] = __float_as_uint(a.temperature); misc[DP_M_SARG + 2] = (uint32_t)a.topk; misc[DP_M_SARG + 3] = a.noise != nullptr; }
    }
    __syncthreads();
    const uint32_t base = dp_sload32(a.epoch);
    if (wave == 7) {
        // ------------------------------------------------------------------------------------------------ gather wave
        __builtin_amdgcn_s_setprio(2);
        const int rep = cu % DP_NREP, ln = (int)lane;
        const dp_u64 *rgQ = a.gQ + rep * 768, *rgH1 = a.gH1 + rep * 512, *rgH2 = a.gH2 + rep * 512, *rgL = a.gL + rep * DP_LSLOTS;
        const dp_u64* rgP = a.gP + (long)cu * 1024;
        const int n_steps = a.cb_last - a.cb_first + 1;
        uint32_t quad_phase = 0;
        dp_flag((dp_lvu32*)(misc + DP_M_FQ), dp_tag(base, 0, 0, DP_E_Q));          // layer 0's q / k / v of the first step are in place
        const bool st = DP_STAMPS(a) != nullptr && cu == 100 && lane == 0;
#define DP_STAMP(s_, i_) do { if (st) DP_STAMPS(a)[(s_) * 32 + (i_)] = __builtin_amdgcn_s_memrealtime(); } while (0)
        for (int s = 0; s < n_steps; ++s) {
            const int cb = a.cb_first + s;
            for (int l = 0; l < DP_NL; ++l) {
                if (l > 0) {
                    {   // rows of the previous layer -> sa_norm -> xA
                        uint32_t v[8];
                        const uint4 g0 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l) * DP_D)[ln], g1 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l) * DP_D)[64 + ln];
                        if (!dp_sweep<4>(rgH2, 512, dp_tag(base, s, l - 1, DP_E_H2), v, ln, ab, a.err, 0x100u + l, a.poll_sleep)) return;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { ((dp_lu32*)(lds + DP_OFF_XA))[2 * (j * 64 + ln)] = v[2 * j]; ((dp_lu32*)(lds + DP_OFF_XA))[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                        dp_norm_in_lds((dp_lu4*)(lds + DP_OFF_XA), g0, g1, a.eps, ln);
                        dp_flag((dp_lvu32*)(misc + DP_M_FXA), dp_tag(base, s, l - 1, DP_E_H2));
                        DP_STAMP(s, l * 4 + 0);
                    }
                    {   // q | k | v -> q buffer and this position's K / V rows
                        uint32_t v[12];
                        if (!dp_sweep<6>(rgQ, 768, dp_tag(base, s, l, DP_E_Q), v, ln, ab, a.err, 0x200u + l, a.poll_sleep, st ? DP_STAMPS(a) + s * 32 + 26 : nullptr)) return;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { ((dp_lu32*)(lds + DP_OFF_QB))[2 * (j * 64 + ln)] = v[2 * j]; ((dp_lu32*)(lds + DP_OFF_QB))[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                        {   // load 4: granules 512 + 2 ln (k: head ln >> 5), load 5: 640 + 2 ln (v)
                            const int kvh = ln >> 5, wd_ = 2 * (ln & 31);
                            dp_lu32* kr = (dp_lu32*)(lds + DP_OFF_K + ((l * 2 + kvh) * 32 + cb) * 256);
                            dp_lu32* vr = (dp_lu32*)(lds + DP_OFF_V + ((l * 2 + kvh) * 32 + cb) * 256);
                            kr[wd_] = v[8]; kr[wd_ + 1] = v[9]; vr[wd_] = v[10]; vr[wd_ + 1] = v[11];
                        }
                        dp_flag((dp_lvu32*)(misc + DP_M_FQ), dp_tag(base, s, l, DP_E_Q));
                        DP_STAMP(s, l * 4 + 1);
                    }
                }
                dp_attention_wave(lds, 7, l, cb, ln);              // (layer 0: q / k / v were placed by the table fetch)
                {   // rows after the o-projection -> mlp_norm -> xC
                    uint32_t v[8];
                    const uint4 g0 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l + 1) * DP_D)[ln], g1 = reinterpret_cast<const uint4*>(a.norms + (long)(2 * l + 1) * DP_D)[64 + ln];
                    if (!dp_sweep<4>(rgH1, 512, dp_tag(base, s, l, DP_E_H1), v, ln, ab, a.err, 0x300u + l, a.poll_sleep)) return;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ((dp_lu32*)(lds + DP_OFF_XC))[2 * (j * 64 + ln)] = v[2 * j]; ((dp_lu32*)(lds + DP_OFF_XC))[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                    dp_norm_in_lds((dp_lu4*)(lds + DP_OFF_XC), g0, g1, a.eps, ln);
                    dp_flag((dp_lvu32*)(misc + DP_M_FXC), dp_tag(base, s, l, DP_E_H1));
                    DP_STAMP(s, l * 4 + 2);
                }
                {   // the 256 down-projection partials of my 4 rows -> sum + residual -> the layer's output rows
                    uint32_t v[16];
                    if (!dp_sweep<8>(rgP, 1024, dp_tag(base, s, l, DP_E_P), v, ln, ab, a.err, 0x400u + l, a.poll_sleep, st ? DP_STAMPS(a) + s * 32 + 27 : nullptr)) return;
                    float t0_, t1_;
                    dp_reduce_partials(v, t0_, t1_);                 // rows 2 (ln & 1), 2 (ln & 1) + 1 of this CU's four
                    const uint32_t h1w = *(dp_lvu32*)(misc + DP_M_H1 + (ln & 1));
                    float y0, y1;
                    {
#pragma clang fp contract(off)
                        y0 = round_bf(t0_) + lo2f(h1w); y1 = round_bf(t1_) + hi2f(h1w);
                    }
                    const uint32_t pair = pack_bf(y0, y1);          // lane parity 0: rows 0, 1; parity 1: rows 2, 3
                    if (ln < 2) misc[DP_M_H0 + ln] = pair;
                    // copies r = ln >> 1 (lanes 0..15), granule 2cu + (ln & 1)
                    if (ln < 2 * DP_NREP) dp_gran_store(a.gH2 + (ln >> 1) * 512 + 2 * cu + (ln & 1), dp_tag(base, s, l, DP_E_H2), pair);
                    DP_STAMP(s, l * 4 + 3);
                }
            }
            {   // the stack's output rows -> final norm -> x of the head
                uint32_t v[8];
                const uint4 g0 = reinterpret_cast<const uint4*>(a.dec_norm)[ln], g1 = reinterpret_cast<const uint4*>(a.dec_norm)[64 + ln];
                if (!dp_sweep<4>(rgH2, 512, dp_tag(base, s, DP_NL - 1, DP_E_H2), v, ln, ab, a.err, 0x500u, a.poll_sleep, st ? DP_STAMPS(a) + s * 32 + 25 : nullptr)) return;
#pragma unroll
                for (int j = 0; j < 4; ++j) { ((dp_lu32*)(lds + DP_OFF_XA))[2 * (j * 64 + ln)] = v[2 * j]; ((dp_lu32*)(lds + DP_OFF_XA))[2 * (j * 64 + ln) + 1] = v[2 * j + 1]; }
                dp_norm_in_lds((dp_lu4*)(lds + DP_OFF_XA), g0, g1, a.eps, ln);
                dp_flag((dp_lvu32*)(misc + DP_M_FXA), dp_tag(base, s, DP_NL - 1, DP_E_H2));
                DP_STAMP(s, 16);
                if (DP_STAMPS(a) != nullptr && s == 5 && lane == 0) DP_STAMPS(a)[1024 + 2048 + 256 + cu] = __builtin_amdgcn_s_memrealtime();
            }
            {   // logits -> LDS
                uint32_t v[18];
                const int ng = (a.V + 1) / 2;
                if (!dp_sweep<9>(rgL, ng, dp_tag(base, s, DP_NL - 1, DP_E_L), v, ln, ab, a.err, 0x600u, a.poll_sleep, st ? DP_STAMPS(a) + s * 32 + 24 : nullptr)) return;
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const int g0 = 2 * (j * 64 + ln);
                    if (g0 < ng) ((dp_lu32*)(lds + DP_OFF_LOGITS))[g0] = v[2 * j];
                    if (g0 + 1 < ng) ((dp_lu32*)(lds + DP_OFF_LOGITS))[g0 + 1] = v[2 * j + 1];
                }
                if ((a.V & 1) && ln == 0) ((dp_lu16*)(lds + DP_OFF_LOGITS))[a.V] = 0;       // the tail unit's second half is not a logit
                dp_flag((dp_lvu32*)(misc + DP_M_FLG), dp_tag(base, s, DP_NL - 1, DP_E_L));
                DP_STAMP(s, 17);
                if (DP_STAMPS(a) != nullptr && s == 5 && lane == 0) DP_STAMPS(a)[1024 + 2048 + cu] = __builtin_amdgcn_s_memrealtime();
            }
            {   // sample with waves 4..6 (this wave is the quad's fourth), then the code -> the next step's input row and layer-0
                // q / k / v (table rows), like k_sample's tail
                const int tok = dp_sample_step<true>(a, lds, 3, ln, cu, cb, s, &quad_phase);
                if (*ab) return;
                DP_STAMP(s, 18);
                if (cb + 1 < a.ncb) {
                    int fed = a.forced ? a.forced[cb] : tok;
                    fed = min(max(fed, 0), a.V - 1);
                    const uint32_t* hrow = reinterpret_cast<const uint32_t*>(a.proj_emb + ((long)cb * a.V + fed) * DP_D);
                    const uint32_t* qrow = reinterpret_cast<const uint32_t*>(a.qkv0_tab + ((long)(cb - 1) * a.V + fed) * DP_NQKV);
                    uint32_t qv[12];
#pragma unroll
                    for (int j = 0; j < 12; ++j) qv[j] = qrow[j * 64 + ln];
                    const uint32_t hv = ln < 2 ? hrow[2 * cu + ln] : 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) ((dp_lu32*)(lds + DP_OFF_QB))[j * 64 + ln] = qv[j];
#pragma unroll
                    for (int j = 8; j < 12; ++j)
                        ((dp_lu32*)(lds + (j < 10 ? DP_OFF_K : DP_OFF_V) + ((j & 1) * 32 + cb + 1) * 256))[ln] = qv[j];
                    if (ln < 2) misc[DP_M_H0 + ln] = hv;
                    dp_flag((dp_lvu32*)(misc + DP_M_FQ), dp_tag(base, s + 1, 0, DP_E_Q));
                    DP_STAMP(s, 19);
                }
            }
        }
        if (cu == 0 && lane == 0) __hip_atomic_store(a.epoch, base + (uint32_t)(n_steps * DP_NL * 5 + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#if DP_ROLES == 4
    if (wave < 2) dp_compute_wave<true, 2, 1>(a, lds, wave, lane, cu, base, ropev);
    else if (wave < 4) dp_compute_wave<true, 2, 2>(a, lds, wave, lane, cu, base, ropev);
    else if (wave == 4) dp_compute_wave<false, 3, 1>(a, lds, wave, lane, cu, base, ropev);
    else dp_compute_wave<false, 3, 2>(a, lds, wave, lane, cu, base, ropev);
#elif DP_ROLES == 3
    if (wave < 2) dp_compute_wave<true, 2, 1>(a, lds, wave, lane, cu, base, ropev);
    else if (wave < 4) dp_compute_wave<true, 2, 2>(a, lds, wave, lane, cu, base, ropev);
    else dp_compute_wave<false, 3, 0>(a, lds, wave, lane, cu, base, ropev);
#else
    if (wave < 4) dp_compute_wave<true, 2, 0>(a, lds, wave, lane, cu, base, ropev);
    else dp_compute_wave<false, 3, 0>(a, lds, wave, lane, cu, base, ropev);
#endif
}
#endif

// W1, W3 [8192][1024] -> [256 cu][4 tiles][32 k steps][64 lanes] 16-byte operand pieces: tile q of workgroup cu holds the
// gate rows (tile rows 0..7) and up rows (8..15) of pairs cu*32 + 8q .. + 7; lane l of step t: row l & 15, k = 32t + 8 (l >> 4)
static __global__ void k_dp_pack_gateup(const bf16_t* w1, const bf16_t* w3, uint4* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 256L * 4 * 32 * 64) return;
    const int lane = (int)(i & 63), t = (int)((i >> 6) & 31), q = (int)((i >> 11) & 3), c = (int)(i >> 13);
    const int r = lane & 15, pair = c * 32 + 8 * q + (r & 7);
    const bf16_t* w = r < 8 ? w1 : w3;
    out[i] = *reinterpret_cast<const uint4*>(w + (long)pair * DP_D + 32 * t + 8 * (lane >> 4));
}

// W2 [1024][8192] -> [256 cu][4 k chunks][1024 rows] 16-byte pieces (workgroup cu's 32 columns, 8 at a time)
static __global__ void k_dp_retile_w2(const bf16_t* w2, uint4* w2s) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 256L * 4 * 1024) return;
    const int n = (int)(i % 1024), q = (int)((i / 1024) % 4), c = (int)(i / 4096);
    w2s[i] = *reinterpret_cast<const uint4*>(w2 + (long)n * DP_FFN + c * 32 + q * 8);
}

// the launch, from the translation unit that holds the kernel (csm_dec_persist.hip)
hipError_t csm_launch_dec_persist(const DecPersistArgs& p, hipStream_t st);
const void* csm_dec_persist_kernel();           // for hipFuncSetAttribute / the occupancy query
