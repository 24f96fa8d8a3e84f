// Weight-streaming skinny GEMV family for the per-frame step (M = 1..few token rows).
//
// HBM-bound by construction: every weight byte is loaded exactly once per launch with
// 16 B/lane coalesced loads straight into VGPRs (no LDS round trip -- guide: "GEMV / M<=16
// decode weights: load straight to VGPRs, deep unroll, late vmcnt"), all of a wave's loads
// are issued before the activation tile is staged, and the weights stay in registers while
// the kernel walks the M tiles.  One wave owns R consecutive output rows; the K reduction is
// lane-strided (lane l, step i covers k = (64 i + l) * 8 .. +8) and finished with wave
// shuffles.  Activations are staged once per block into LDS as packed bf16 (optionally
// RMS-normalised on the way in) and consumed with v_dot2c_f32_bf16.
//
// Rounding points are the reference's (torchtune 0.4.0 eager bf16):
// fp32 accumulate -> bf16 at every Linear output; RMSNorm rounds before the scale multiply;
// RoPE / SiLU / residual each round once.
#pragma once
#include "common.cuh"

enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_ATTN = 2, PRO_COMBINE = 3 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV_ROPE = 2, EPI_SWIGLU = 3, EPI_SLAB = 4 };

struct GemvArgs {
    // activations: row m lives at x + m * x_row_stride + x_row_offset (elements), K wide
    const bf16_t* x;
    long x_row_stride, x_row_offset;
    int M;
    // prologue
    const bf16_t* norm_scale;   // PRO_NORM
    float eps;
    bf16_t* normed_out;         // PRO_NORM, optional: block 0 writes normalised row m at + m*normed_stride
    long normed_stride;
    // weights [N][K]
    const bf16_t* w0;           // STORE/RESID: W; QKV: Wq; SWIGLU: W1 (gate)
    const bf16_t* w1;           //                  QKV: Wk; SWIGLU: W3 (up)
    const bf16_t* w2;           //                  QKV: Wv
    int N;                      // output rows (QKV: nq + 2 nkv; SWIGLU: ffn)
    // epilogue
    bf16_t* out;                // row m at out + m*ldo   (QKV: q buffer, ldo = nq)
    long ldo;
    int nt;                     // non-temporal weight loads (streamed-once weights)
    const bf16_t* resid;        // EPI_RESID [M][N] (may alias out)
    // QKV
    int nq, nkv, smax, rows_per_seq, kv_heads;
    const int* pos;             // [M] absolute position of each row
    const bf16_t* rope;         // [max_seq][HD/2][2]
    bf16_t* kcache;             // [B][KV][smax][HD] (this layer)
    bf16_t* vcache;
    // PRO_ATTN (depth decoder: hd 128, <= 32 keys): the activation row IS the attention output of
    // q [M][aH*128] over keys [0,pos[m]] of kcache/vcache, computed in the prologue
    const bf16_t* aq;
    int aH;
    float ascale;
    int pos_base;               // used when pos == nullptr
    // PRO_COMBINE (backbone split-K attention): activation row = merged partial softmax states
    const float* part;          // [M][aH][nsplit][64 + 4]: o[64], m, l, pad
    int nsplit;
    // EPI_SLAB (wide-M split-K): fp32 partial sums, slab g of [gridDim.z][M][N]
    float* slab;
    // WT == 1: w0/w1/w2 point to OCP-e4m3 bytes [N][K]; per-output-row power-of-two scales
    const float *s0, *s1, *s2;
    // wide-M decode steps: `out` (SwiGLU) is written in operand order (xp_off)
    int out_packed;
    // MSPLIT instantiations only (table builds at csm_create: M = tens of thousands of rows): rows per blockIdx.y, a multiple of MT
    int m_chunk;
#ifdef GEMV_PF_HOOKS   // tools/microbench/pfchain_bench.hip only (measured: every prefetch form made the chain slower)
    // Optional L2 prefetch of the NEXT launch's weights: workgroups blockIdx.x >= work_blocks (when pf[0].base != nullptr)
    // do no GEMV work; they read the byte regions the next launch's workgroups will stream (region b' of matrix i =
    // [base + b' * bytes, + bytes)), taking the regions of the workgroups that should share their XCD (round-robin
    // placement: b' = blockIdx.x + pf_shift mod 8).  Speed only -- the bytes are read and dropped.
    struct { const char* base; int bytes; int nblocks; } pf[3];
    int pf_shift, work_blocks;
    // Optional: bumped once per launch (workgroup 0) so a concurrent weight streamer can pace itself (streamer.cuh)
    unsigned* progress;
#endif
};

#ifdef GEMV_PF_HOOKS
__device__ __forceinline__ void gemv_prefetch_blocks(const GemvArgs& a) {
    const int p = (int)blockIdx.x - a.work_blocks, P = (int)gridDim.x - a.work_blocks;
    const int r = ((int)blockIdx.x + a.pf_shift) & 7, q = p >> 3, Pq = (P + 7) >> 3;
    uint32_t acc = 0;
#pragma unroll 1
    for (int i = 0; i < 3; ++i) {
        if (!a.pf[i].base) break;
        const int pieces = a.pf[i].bytes >> 12;                   // 4 KB (one 16-byte piece per thread) per step
        for (int b = r + 8 * q; b < a.pf[i].nblocks; b += 8 * Pq) {
            const uint4* src = reinterpret_cast<const uint4*>(a.pf[i].base + (long)b * a.pf[i].bytes) + threadIdx.x;
            for (int j = 0; j < pieces; j += 4) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = (j + u < pieces) ? src[(j + u) * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].w;
            }
        }
    }
    if (acc == 0x9e3779b9u && a.M < 0) a.out[0] = (bf16_t)acc;     // never true: keeps the loads alive
}
#endif

// Depth-decoder attention fused into the output projection's prologue (hd = 128, at most 32
// keys: the decoder cache holds one frame's codebooks, sesameai/models.py:127).  Replaces a
// separate launch per layer per decoder step (124 per frame).  Every block recomputes the
// <= 32 KB of K/V reads from L2; wave w handles heads w, w+4, ...  Two lanes per key for
// q.k (64 elements each), the softmax over the 32 key slots is a wave reduction, then each
// lane accumulates two output elements over the keys.  Rounded to bf16 once, like SDPA.
__device__ __forceinline__ int row_pos(const GemvArgs& a, long mrow) {
    // decoder positions are compile-time per step: pos == nullptr -> pos_base + row-in-sequence,
    // which removes a dependent global load from the critical path
    int p = a.pos ? a.pos[mrow] : a.pos_base + (int)(mrow % a.rows_per_seq);
    return p < 0 ? 0 : (p >= a.smax ? a.smax - 1 : p);
}

// sum over the 16 lanes of a DPP row (result in every lane of the row)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1, 0xF>(0.f, v);
    v += dpp_f<0x4E, 0xF>(0.f, v);
    v += dpp_f<0x141, 0xF>(0.f, v);
    v += dpp_f<0x140, 0xF>(0.f, v);
    return v;
}

// Layout (all loads coalesced and issued up front): the KV head's K tile (32 keys x 128 = 8 KB,
// contiguous in the cache) is fetched as 8 x 1 KB wave loads, so lane l holds 16-byte pieces of
// keys 4i + l/16 (i = 0..7) at element offset 8*(l%16); with the matching 8 elements of q the
// q.k dot is 8 partials per lane, each finished by a 4-step DPP row reduction.  The softmax
// runs on those 8 scores per lane (every key replicated over 16 lanes).  P then goes through
// LDS (32 floats per head) and each lane accumulates two output columns over the keys from
// contiguous 256-byte V rows.  Two heads of one KV group are processed together.
template <int MT, int KITERS>
__device__ __forceinline__ void stage_attn(bf16_t* xs, float* ps /*[4 waves][2][32]*/, const GemvArgs& a, int m0) {
    constexpr int K = KITERS * 512;
    constexpr int HD = 128;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane >> 4, sub = lane & 15;
    const int G = a.aH / a.kv_heads;                   // even (4 for both CSM stacks): a pair of heads shares one KV head
    float* myps = ps + wave * 64;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mrow = m0 + m;
        if (mrow >= a.M) {
            for (int c = threadIdx.x; c < K / 8; c += 256) reinterpret_cast<uint4*>(xs + m * K)[c] = make_uint4(0, 0, 0, 0);
            continue;
        }
        const int p = row_pos(a, mrow);
        const int nk = min(p + 1, 32);
        const int b = mrow / a.rows_per_seq;
        for (int h = wave * 2; h < a.aH; h += 8) {
            const int kvh = h / G;
            const bf16_t* kb = a.kcache + ((long)b * a.kv_heads + kvh) * a.smax * HD;
            const bf16_t* vb = a.vcache + ((long)b * a.kv_heads + kvh) * a.smax * HD;
            const uint4 qa = *reinterpret_cast<const uint4*>(a.aq + ((long)mrow * a.aH + h) * HD + sub * 8);
            const uint4 qb = *reinterpret_cast<const uint4*>(a.aq + ((long)mrow * a.aH + h + 1) * HD + sub * 8);
            // nk (= position + 1) is the same for every lane: whole 4-key groups beyond it are skipped by uniform
            // branches -- on average half of the loads, dots and P.V terms of a decoder step (positions 1..31)
            uint4 kv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (4 * i < nk) kv[i] = reinterpret_cast<const uint4*>(kb)[i * 64 + lane];
            uint32_t vv[32];
#pragma unroll
            for (int t = 0; t < 32; ++t)
                if ((t & ~3) < nk) vv[t] = reinterpret_cast<const uint32_t*>(vb + (long)t * HD)[lane];
            float s0[8], s1[8], mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s0[i] = -INFINITY; s1[i] = -INFINITY;
                if (4 * i < nk) {
                    const bool live = (4 * i + grp) < nk;
                    const float d0 = row16_sum(dot8(qa, kv[i], 0.f)) * a.ascale, d1 = row16_sum(dot8(qb, kv[i], 0.f)) * a.ascale;
                    s0[i] = live ? d0 : -INFINITY;
                    s1[i] = live ? d1 : -INFINITY;
                    mx0 = fmaxf(mx0, s0[i]); mx1 = fmaxf(mx1, s1[i]);
                }
            }
            mx0 = wave_max(mx0); mx1 = wave_max(mx1);
            float l0 = 0.f, l1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (4 * i < nk) {
                    s0[i] = (s0[i] == -INFINITY) ? 0.f : __expf(s0[i] - mx0);
                    s1[i] = (s1[i] == -INFINITY) ? 0.f : __expf(s1[i] - mx1);
                    l0 += s0[i]; l1 += s1[i];
                    if (sub == 0) { myps[4 * i + grp] = s0[i]; myps[32 + 4 * i + grp] = s1[i]; }
                }
            }
            l0 = wave_sum(l0) * (1.0f / 16.0f);          // every key is replicated over the 16 lanes of its row
            l1 = wave_sum(l1) * (1.0f / 16.0f);
            float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
#pragma unroll
            for (int t4 = 0; t4 < 8; ++t4) {            // same-address LDS reads broadcast
                if (4 * t4 < nk) {
                    const float4 pa = *reinterpret_cast<const float4*>(myps + 4 * t4);
                    const float4 pb = *reinterpret_cast<const float4*>(myps + 32 + 4 * t4);
                    const float pav[4] = {pa.x, pa.y, pa.z, pa.w}, pbv[4] = {pb.x, pb.y, pb.z, pb.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // rows beyond nk hold stale cache data: p is exactly 0 there, but guard NaN/Inf bit patterns
                        const uint32_t raw = (4 * t4 + u) < nk ? vv[4 * t4 + u] : 0u;
                        const float v0 = lo2f(raw), v1 = hi2f(raw);
                        o00 += pav[u] * v0; o01 += pav[u] * v1;
                        o10 += pbv[u] * v0; o11 += pbv[u] * v1;
                    }
                }
            }
            const float i0 = 1.0f / l0, i1 = 1.0f / l1;
            reinterpret_cast<uint32_t*>(xs + m * K + h * HD)[lane] = pack_bf(o00 * i0, o01 * i0);
            reinterpret_cast<uint32_t*>(xs + m * K + (h + 1) * HD)[lane] = pack_bf(o10 * i1, o11 * i1);
        }
    }
    __syncthreads();
}

// Merge of the split-K attention partials (k_attn with nsplit > 1) fused into the output
// projection's prologue: thread t owns one 16-byte chunk (8 columns) of one head (hd = 64).
template <int MT, int KITERS>
__device__ __forceinline__ void stage_combine(bf16_t* xs, const GemvArgs& a, int m0) {
    constexpr int K = KITERS * 512;
    constexpr int HD = 64, PS = ATTN_PS(HD);
    const int tid = threadIdx.x;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int mrow = m0 + m;
        for (int c = tid; c < K / 8; c += 256) {
            uint4 o = make_uint4(0, 0, 0, 0);
            if (mrow < a.M) {
                const int h = c / (HD / 8), e0 = (c % (HD / 8)) * 8;
                const float* src = a.part + ((long)mrow * a.aH + h) * a.nsplit * PS;
                float mx = -INFINITY;
                for (int sp = 0; sp < a.nsplit; ++sp) mx = fmaxf(mx, src[sp * PS + HD]);
                float num[8], den = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) num[i] = 0.f;
                for (int sp = 0; sp < a.nsplit; ++sp) {
                    const float ms = src[sp * PS + HD];
                    const float wgt = (ms == -INFINITY) ? 0.f : __expf(ms - mx);
                    const float4 v0 = *reinterpret_cast<const float4*>(src + sp * PS + e0);
                    const float4 v1 = *reinterpret_cast<const float4*>(src + sp * PS + e0 + 4);
                    num[0] += wgt * v0.x; num[1] += wgt * v0.y; num[2] += wgt * v0.z; num[3] += wgt * v0.w;
                    num[4] += wgt * v1.x; num[5] += wgt * v1.y; num[6] += wgt * v1.z; num[7] += wgt * v1.w;
                    den += wgt * src[sp * PS + HD + 1];
                }
                const float inv = 1.0f / den;
                o.x = pack_bf(num[0] * inv, num[1] * inv); o.y = pack_bf(num[2] * inv, num[3] * inv);
                o.z = pack_bf(num[4] * inv, num[5] * inv); o.w = pack_bf(num[6] * inv, num[7] * inv);
            }
            reinterpret_cast<uint4*>(xs + m * K)[c] = o;
        }
    }
    __syncthreads();
}

template <int MT, int KITERS, bool NORM>
__device__ __forceinline__ void stage_x(bf16_t* xs, float* red, const GemvArgs& a, int m0) {
    constexpr int K = KITERS * 512;
    constexpr int CHUNKS = K / 8;               // 16-byte chunks per row
    constexpr int CPT = (CHUNKS + 255) / 256;   // chunks per thread
    const int tid = threadIdx.x;
    if constexpr (!NORM) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool live = (m0 + m) < a.M;
            const uint4* src = reinterpret_cast<const uint4*>(a.x + (long)(m0 + m) * a.x_row_stride + a.x_row_offset);
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = tid + i * 256;
                if (c < CHUNKS) reinterpret_cast<uint4*>(xs + m * K)[c] = live ? src[c] : make_uint4(0, 0, 0, 0);
            }
        }
    } else {
        // all global loads (activation chunks AND the norm scale) are issued before the
        // reduction; the chunks stay in registers across it, so LDS is written exactly once
        uint4 xv[MT][CPT], g[CPT];
        float ss[MT];
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int c = tid + i * 256;
            g[i] = c < CHUNKS ? reinterpret_cast<const uint4*>(a.norm_scale)[c] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const bool live = (m0 + m) < a.M;
            const uint4* src = reinterpret_cast<const uint4*>(a.x + (long)(m0 + m) * a.x_row_stride + a.x_row_offset);
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = tid + i * 256;
                xv[m][i] = (live && c < CHUNKS) ? src[c] : make_uint4(0, 0, 0, 0);
            }
        }
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ss[m] = 0.f;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const uint4 v = xv[m][i];
                float f;
                f = lo2f(v.x); ss[m] += f * f; f = hi2f(v.x); ss[m] += f * f;
                f = lo2f(v.y); ss[m] += f * f; f = hi2f(v.y); ss[m] += f * f;
                f = lo2f(v.z); ss[m] += f * f; f = hi2f(v.z); ss[m] += f * f;
                f = lo2f(v.w); ss[m] += f * f; f = hi2f(v.w); ss[m] += f * f;
            }
            const float sw = wave_sum(ss[m]);
            if (lane == 0) red[m * 4 + wave] = sw;
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float tot = red[m * 4 + 0] + red[m * 4 + 1] + red[m * 4 + 2] + red[m * 4 + 3];
            const float r = 1.0f / sqrtf(tot / (float)K + a.eps);
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int c = tid + i * 256;
                if (c >= CHUNKS) continue;
                const uint4 v = xv[m][i];
                uint4 o;
                // x32 * rsqrt -> bf16 (type_as) -> * scale (bf16 * bf16 -> bf16)
                o.x = pack_bf(round_bf(lo2f(v.x) * r) * lo2f(g[i].x), round_bf(hi2f(v.x) * r) * hi2f(g[i].x));
                o.y = pack_bf(round_bf(lo2f(v.y) * r) * lo2f(g[i].y), round_bf(hi2f(v.y) * r) * hi2f(g[i].y));
                o.z = pack_bf(round_bf(lo2f(v.z) * r) * lo2f(g[i].z), round_bf(hi2f(v.z) * r) * hi2f(g[i].z));
                o.w = pack_bf(round_bf(lo2f(v.w) * r) * lo2f(g[i].w), round_bf(hi2f(v.w) * r) * hi2f(g[i].w));
                reinterpret_cast<uint4*>(xs + m * K)[c] = o;
                if (a.normed_out != nullptr && blockIdx.x == 0 && (m0 + m) < a.M)
                    reinterpret_cast<uint4*>(a.normed_out + (long)(m0 + m) * a.normed_stride)[c] = o;
            }
        }
    }
    __syncthreads();
}

// R = weight rows per wave.  EPI_QKV_ROPE: R == 2 (one interleaved RoPE pair).
// EPI_SWIGLU: R == 2*P, rows [0,P) are gate rows i..i+P-1 and [P,2P) the matching up rows.
// MSPLIT (round 5): blockIdx.y takes the token rows [y * m_chunk, (y + 1) * m_chunk) -- for the two load-time table builds, whose 65,632 /
// 2,051 rows every block used to walk alone, 4 at a time (96 ms of GPU time per csm_create).  Same lane / k mapping and the same
// per-row arithmetic: the same bits.  Without it the constants below fold and the decode-step instantiations are unchanged.
template <int MT, int KITERS, int R, int PRO, int EPI, int HD, int WT = 0, bool MSPLIT = false>
__global__ __launch_bounds__(256) void k_gemv(const GemvArgs a) {
    constexpr int K = KITERS * 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* xs = reinterpret_cast<bf16_t*>(smem);
    float* red = reinterpret_cast<float*>(smem + (size_t)MT * K * 2);
#ifdef GEMV_PF_HOOKS
    if (a.pf[0].base != nullptr && (int)blockIdx.x >= a.work_blocks) { gemv_prefetch_blocks(a); return; }
    if (a.progress != nullptr && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_fetch_add(a.progress, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int unit = blockIdx.x * 4 + wave;              // one unit = R weight rows
    const int m_begin = MSPLIT ? (int)blockIdx.y * a.m_chunk : 0;
    const int m_end = MSPLIT ? (m_begin + a.m_chunk < a.M ? m_begin + a.m_chunk : a.M) : a.M;

    // ---- resolve this wave's R weight rows (nullptr = past the end) ------------------------
    constexpr int WS = WT == 1 ? 1 : 2;                   // bytes per weight
    const char* wrow[R];
    float wsc[R];                                         // WT == 1: the row's scale
    int orow[R];                                          // output row index
#pragma unroll
    for (int r = 0; r < R; ++r) wsc[r] = 1.0f;
    if (EPI == EPI_QKV_ROPE) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = unit * R + r;
            orow[r] = row;
            if (row < a.nq) { wrow[r] = (const char*)a.w0 + (long)row * K * WS; if (WT) wsc[r] = a.s0[row]; }
            else if (row < a.nq + a.nkv) { wrow[r] = (const char*)a.w1 + (long)(row - a.nq) * K * WS; if (WT) wsc[r] = a.s1[row - a.nq]; }
            else if (row < a.N) { wrow[r] = (const char*)a.w2 + (long)(row - a.nq - a.nkv) * K * WS; if (WT) wsc[r] = a.s2[row - a.nq - a.nkv]; }
            else wrow[r] = nullptr;
        }
    } else if (EPI == EPI_SWIGLU) {
        constexpr int P = R / 2;
#pragma unroll
        for (int r = 0; r < P; ++r) {
            int row = unit * P + r;
            orow[r] = orow[r + P] = row;
            wrow[r] = row < a.N ? (const char*)a.w0 + (long)row * K * WS : nullptr;
            wrow[r + P] = row < a.N ? (const char*)a.w1 + (long)row * K * WS : nullptr;
            if (WT && row < a.N) { wsc[r] = a.s0[row]; wsc[r + P] = a.s1[row]; }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int row = unit * R + r;
            orow[r] = row;
            wrow[r] = row < a.N ? (const char*)a.w0 + (long)row * K * WS : nullptr;
            if (WT && row < a.N) wsc[r] = a.s0[row];
        }
    }

    // ---- issue every weight load of this wave up front; they stay in VGPRs ----------------
    // WT == 1: a lane's 16 bytes hold 16 weights, so one wave load covers 1024 k (half of them masked at K = 512)
    constexpr int WI = WT == 1 ? (KITERS + 1) / 2 : KITERS;
    uint4 wv[R][WI];
    if constexpr (WT == 1) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < WI; ++i)
                wv[r][i] = (wrow[r] && (i * 64 + lane) * 16 < K) ? ldg16<true>(reinterpret_cast<const uint4*>(wrow[r]) + i * 64 + lane)
                                                                  : make_uint4(0, 0, 0, 0);
    } else {
        if (a.nt) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < WI; ++i)
                    wv[r][i] = wrow[r] ? ldg16<true>(reinterpret_cast<const uint4*>(wrow[r]) + i * 64 + lane)
                                       : make_uint4(0, 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int i = 0; i < WI; ++i)
                    wv[r][i] = wrow[r] ? ldg16<false>(reinterpret_cast<const uint4*>(wrow[r]) + i * 64 + lane)
                                       : make_uint4(0, 0, 0, 0);
        }
    }

    // ---- prefetch the epilogue's operands for the first M tile (lane m finishes row m): the
    //      residual values / the RoPE table entry would otherwise be dependent loads at the very end
    float rpre[R];
    uint32_t cspre = 0;
    int ppre = 0;
    if constexpr (EPI == EPI_RESID) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            rpre[r] = (lane < MT && m_begin + lane < m_end && orow[r] < a.N) ? bf2f(a.resid[(long)(m_begin + lane) * a.ldo + orow[r]]) : 0.f;
    }
    if constexpr (EPI == EPI_QKV_ROPE) {
        if (lane < MT && m_begin + lane < m_end) {
            ppre = row_pos(a, m_begin + lane);
            if (orow[0] < a.nq + a.nkv) {
                const int e = (orow[0] < a.nq ? orow[0] : orow[0] - a.nq) % HD;
                cspre = reinterpret_cast<const uint32_t*>(a.rope)[(long)ppre * (HD / 2) + e / 2];
            }
        }
    }

    for (int m0 = m_begin; m0 < m_end; m0 += MT) {
        if (m0 > m_begin) __syncthreads();
        if constexpr (PRO == PRO_ATTN) stage_attn<MT, KITERS>(xs, red + 16, a, m0);
        else if constexpr (PRO == PRO_COMBINE) stage_combine<MT, KITERS>(xs, a, m0);
        else stage_x<MT, KITERS, PRO == PRO_NORM>(xs, red, a, m0);

        float acc[MT][R];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[m][r] = 0.f;
        if constexpr (WT == 1) {
#pragma unroll
            for (int i = 0; i < WI; ++i) {
                const int c2 = (i * 64 + lane) * 2;          // first of the lane's two 8-element x chunks
                const bool in = c2 * 8 < K;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const uint4 x0 = in ? reinterpret_cast<const uint4*>(xs + m * K)[c2] : make_uint4(0, 0, 0, 0);
                    const uint4 x1 = in ? reinterpret_cast<const uint4*>(xs + m * K)[c2 + 1] : make_uint4(0, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[m][r] = dot16_fp8(wv[r][i], x0, x1, acc[m][r]);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < KITERS; ++i) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const uint4 xv = reinterpret_cast<const uint4*>(xs + m * K)[i * 64 + lane];
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[m][r] = dot8(wv[r][i], xv, acc[m][r]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[m][r] = wave_sum(acc[m][r]) * wsc[r];     // power-of-two scale: exact

        // ---- epilogue: lane m finishes token row m0+m -------------------------------------
        {
#pragma clang fp contract(off)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (lane != m || m0 + m >= m_end) continue;
            const long mrow = m0 + m;
            if constexpr (EPI == EPI_STORE || EPI == EPI_RESID) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (orow[r] >= a.N) continue;
                    float y = round_bf(acc[m][r]);
                    if (EPI == EPI_RESID) y = y + (m0 == m_begin ? rpre[r] : bf2f(a.resid[mrow * a.ldo + orow[r]]));
                    a.out[mrow * a.ldo + orow[r]] = f2bf(y);
                }
            } else if constexpr (EPI == EPI_SWIGLU) {
                constexpr int P = R / 2;
#pragma unroll
                for (int r = 0; r < P; ++r) {
                    if (orow[r] >= a.N) continue;
                    const float g = round_bf(acc[m][r]);
                    const float u = round_bf(acc[m][r + P]);
                    const float s = round_bf(g / (1.0f + __expf(-g)));      // F.silu in bf16
                    a.out[mrow * a.ldo + orow[r]] = f2bf(s * u);
                }
            } else {   // EPI_QKV_ROPE: rows (2i, 2i+1) of one head
                const int row = orow[0];
                if (row >= a.N) continue;
                float v0 = round_bf(acc[m][0]), v1 = round_bf(acc[m][1]);
                const int p = m0 == m_begin ? ppre : row_pos(a, mrow);    // clamped: the host guards length
                const int b = (int)(mrow / a.rows_per_seq);
                if (row < a.nq + a.nkv) {                 // q or k: interleaved Llama3-scaled RoPE
                    const int e = (row < a.nq ? row : row - a.nq) % HD;
                    const uint32_t cs = m0 == m_begin ? cspre : reinterpret_cast<const uint32_t*>(a.rope)[(long)p * (HD / 2) + e / 2];
                    const float c = lo2f(cs), s = hi2f(cs);
                    const float o0 = v0 * c - v1 * s;
                    const float o1 = v1 * c + v0 * s;
                    v0 = o0; v1 = o1;
                }
                const uint32_t packed = pack_bf(v0, v1);
                if (row < a.nq) {
                    *reinterpret_cast<uint32_t*>(a.out + mrow * a.ldo + row) = packed;
                } else {
                    const bool isk = row < a.nq + a.nkv;
                    const int rk = row - a.nq - (isk ? 0 : a.nkv);
                    const int kvh = rk / HD, e = rk % HD;
                    bf16_t* dst = (isk ? a.kcache : a.vcache) + (((long)b * a.kv_heads + kvh) * a.smax + p) * HD + e;
                    *reinterpret_cast<uint32_t*>(dst) = packed;
                }
            }
        }
        }
    }
}
