// KV-cached GQA attention for token rows with per-row absolute positions (decode and
// chunked prefill): softmax(q k^T / sqrt(hd)) v over keys [0, pos] of the row's sequence.
//
// Replaces torchtune's SDPA over the full 2048-slot cache with a boolean mask row
// (sesameai/models.py:154,172): attention is bounded by position, so
// only (pos+1) keys of the 8 (resp. 2) real KV heads are read.
//
// grid = (M rows, KV heads, nsplit key ranges); block = 4 waves = the 4 query heads of the GQA
// group, so a K/V tile fetched by one wave is an L1/L2 hit for its three siblings.  Inside a
// wave a key row (hd bf16) is spread over hd/8 lanes with 16-byte loads (8 resp. 4 keys per
// wave-instruction); the q.k dot is finished with 3-4 xor-shuffles, each key slot keeps its
// own online-softmax state and the slots are merged by shuffles at the end.
#pragma once
#include "common.cuh"

struct AttnArgs {
    const bf16_t* q;        // [M][H*HD]
    const bf16_t* kcache;   // [B][KV][smax][HD]
    const bf16_t* vcache;
    const int* pos;         // [M]
    int M, rows_per_seq, H, KV, smax, nsplit;
    float scale;
    bf16_t* out;            // [M][H*HD]                 (nsplit == 1)
    float* part;            // [M][H][nsplit][ATTN_PS(HD)] (nsplit > 1): o[HD], m, l, padding to whole 128-byte lines
    int out_packed;         // out in matrix-core operand order (common.cuh xp_off), K = H*HD
    int* ctr;               // nsplit > 1: [M][KV] arrival counters (zero between launches) -> the LAST key-range block of a
                            // (row, KV head) merges the partials itself (no k_attn_combine launch); nullptr: partials only
};

#define ATTN_MAX_SPLIT 8            // key ranges per (row, KV head) the in-kernel merge handles (csm_engine.hip BB_NSPLIT_MAX)
__device__ __forceinline__ void attn_st16_sc1(float* p, const float4& v) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    f32x4v t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void attn_st4_sc1(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// U = key rows per lane in flight per batch (hd 128 / depth decoder: 8, so its <= 32 keys are ONE round trip)
template <int HD, int U = 4>
__global__ __launch_bounds__(256) void k_attn(const AttnArgs a) {
    constexpr int LPK = HD / 8;          // lanes per key
    constexpr int KPI = 64 / LPK;        // keys per wave-iteration
    const int m = blockIdx.x, kvh = blockIdx.y, sp = blockIdx.z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int G = a.H / a.KV;
    const int slot = lane / LPK, e = lane % LPK;
    const int p = min(max(a.pos[m], 0), a.smax - 1);
    const int b = m / a.rows_per_seq;
    const int nkeys = p + 1;
    const int chunk = (nkeys + a.nsplit - 1) / a.nsplit;
    const int j0 = sp * chunk, j1 = min(nkeys, j0 + chunk);
    const bf16_t* kb = a.kcache + ((long)b * a.KV + kvh) * a.smax * HD;
    const bf16_t* vb = a.vcache + ((long)b * a.KV + kvh) * a.smax * HD;

    for (int g = wave; g < G; g += 4) {
        const int h = kvh * G + g;
        const uint4 qv = *reinterpret_cast<const uint4*>(a.q + ((long)m * a.H + h) * HD + e * 8);
        float mx = -INFINITY, l = 0.f, o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;

        // U key rows per lane in flight: all loads of a batch are issued before the first is consumed
        for (int jb = j0 + slot; jb < j1; jb += U * KPI) {
            uint4 kv[U], vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = min(jb + u * KPI, j1 - 1);
                kv[u] = *reinterpret_cast<const uint4*>(kb + (long)j * HD + e * 8);
                vv[u] = *reinterpret_cast<const uint4*>(vb + (long)j * HD + e * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float s = dot8(qv, kv[u], 0.f);
#pragma unroll
                for (int off = 1; off < LPK; off <<= 1) s += __shfl_xor(s, off, WAVE);
                if (jb + u * KPI >= j1) continue;            // uniform across the LPK lanes of a key
                s *= a.scale;
                const float mn = fmaxf(mx, s);
                const float corr = __expf(mx - mn);          // mx = -inf on the first key -> 0
                const float pw = __expf(s - mn);
                l = l * corr + pw;
                o[0] = o[0] * corr + pw * lo2f(vv[u].x); o[1] = o[1] * corr + pw * hi2f(vv[u].x);
                o[2] = o[2] * corr + pw * lo2f(vv[u].y); o[3] = o[3] * corr + pw * hi2f(vv[u].y);
                o[4] = o[4] * corr + pw * lo2f(vv[u].z); o[5] = o[5] * corr + pw * hi2f(vv[u].z);
                o[6] = o[6] * corr + pw * lo2f(vv[u].w); o[7] = o[7] * corr + pw * hi2f(vv[u].w);
                mx = mn;
            }
        }
        // merge the KPI key slots (lanes that share e)
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) {
            const float mo = __shfl_xor(mx, off, WAVE);
            const float lo = __shfl_xor(l, off, WAVE);
            const float mn = fmaxf(mx, mo);
            const float c0 = (mx == -INFINITY) ? 0.f : __expf(mx - mn);
            const float c1 = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
            l = l * c0 + lo * c1;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float oo = __shfl_xor(o[i], off, WAVE);
                o[i] = o[i] * c0 + oo * c1;
            }
            mx = mn;
        }
        if (slot == 0) {
            if (a.nsplit == 1) {
                const float inv = 1.0f / l;
                uint4 r;
                r.x = pack_bf(o[0] * inv, o[1] * inv); r.y = pack_bf(o[2] * inv, o[3] * inv);
                r.z = pack_bf(o[4] * inv, o[5] * inv); r.w = pack_bf(o[6] * inv, o[7] * inv);
                bf16_t* dst = a.out_packed ? a.out + xp_off(m, h * HD + e * 8, (long)a.H * HD) : a.out + ((long)m * a.H + h) * HD + e * 8;
                *reinterpret_cast<uint4*>(dst) = r;
            } else {
                float* dst = a.part + (((long)m * a.H + h) * a.nsplit + sp) * ATTN_PS(HD);
                if (a.ctr != nullptr) {           // merged in this launch by a block that may sit on another XCD: write through (sc1)
                    attn_st16_sc1(dst + e * 8, make_float4(o[0], o[1], o[2], o[3]));
                    attn_st16_sc1(dst + e * 8 + 4, make_float4(o[4], o[5], o[6], o[7]));
                    if (e == 0) { attn_st4_sc1(dst + HD, mx); attn_st4_sc1(dst + HD + 1, l); }
                } else {
                    *reinterpret_cast<float4*>(dst + e * 8) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4*>(dst + e * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
                    if (e == 0) { dst[HD] = mx; dst[HD + 1] = l; }
                }
            }
        }
    }
    if (a.nsplit > 1 && a.ctr != nullptr) {
        // The last of the nsplit blocks of this (row, KV head) to get here merges: k_attn_combine's arithmetic in k_attn_combine's
        // order (same bits), one launch less per layer.  No block waits for another (the merge is done by whoever comes last), so
        // nothing depends on the blocks being resident together.  The partials travel like the persistent launches' exchanges:
        // write-through stores, acknowledged (vmcnt 0) before the arrival is counted, read back with L2-bypassing loads -- agent-scope
        // fences instead cost a whole-L2 write-back per block (measured: +58 us per layer at 32 rows).
        __shared__ int s_ticket;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(a.ctr + m * a.KV + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_ticket != a.nsplit - 1) return;
        if (threadIdx.x == 0) a.ctr[m * a.KV + kvh] = 0;                 // for the next launch (ordered by the kernel boundary)
        for (int idx = threadIdx.x; idx < G * HD; idx += 256) {
            const int h = kvh * G + idx / HD, t = idx % HD;
            const float* src = a.part + ((long)m * a.H + h) * a.nsplit * ATTN_PS(HD);
            float ms[ATTN_MAX_SPLIT], ls[ATTN_MAX_SPLIT], vs[ATTN_MAX_SPLIT];
#pragma unroll
            for (int s = 0; s < ATTN_MAX_SPLIT; ++s) {                   // every load in flight at once (splits past nsplit re-read the last one)
                const float* ps = src + min(s, a.nsplit - 1) * ATTN_PS(HD);
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(ms[s]) : "v"(ps + HD) : "memory");
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(ls[s]) : "v"(ps + HD + 1) : "memory");
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(vs[s]) : "v"(ps + t) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < ATTN_MAX_SPLIT; ++s) asm volatile("" : "+v"(ms[s]), "+v"(ls[s]), "+v"(vs[s]));
            float mxx = -INFINITY;
#pragma unroll
            for (int s = 0; s < ATTN_MAX_SPLIT; ++s) if (s < a.nsplit) mxx = fmaxf(mxx, ms[s]);
            float num = 0.f, den = 0.f;
#pragma unroll
            for (int s = 0; s < ATTN_MAX_SPLIT; ++s) {
                if (s < a.nsplit) {
                    const float w = (ms[s] == -INFINITY) ? 0.f : __expf(ms[s] - mxx);
                    num += w * vs[s];
                    den += w * ls[s];
                }
            }
            a.out[a.out_packed ? xp_off(m, h * HD + t, (long)a.H * HD) : ((long)m * a.H + h) * HD + t] = f2bf(num / den);
        }
    }
}

// merges the nsplit partial softmax states of one (row, head): grid (M, H), block HD threads
template <int HD>
__global__ void k_attn_combine(const float* part, int nsplit, bf16_t* out, int H, int out_packed) {
    const int m = blockIdx.x, h = blockIdx.y, t = threadIdx.x;
    const float* src = part + ((long)m * H + h) * nsplit * ATTN_PS(HD);
    float mx = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, src[s * ATTN_PS(HD) + HD]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = src[s * ATTN_PS(HD) + HD];
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - mx);
        num += w * src[s * ATTN_PS(HD) + t];
        den += w * src[s * ATTN_PS(HD) + HD + 1];
    }
    out[out_packed ? xp_off(m, h * HD + t, (long)H * HD) : ((long)m * H + h) * HD + t] = f2bf(num / den);
}
