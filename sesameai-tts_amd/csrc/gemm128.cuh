// Long prompts / batched prefill (M >= 256 token rows): LDS-tiled bf16 GEMM on the matrix cores.
//
//   out[M][N] = x[M][K] . W[N][K]^T       W in its ORIGINAL row-major layout (no packed copy needed)
//
// k_mm32 (mm.cuh) gives every 32 x 32 output tile its own block, which is right for a handful of row
// tiles but re-reads x N/32 times and W M/32 times; at thousands of rows that traffic (and the
// 64-cache-line x gathers feeding the MFMAs) is what bounds it.  Here a block owns a 128 x 128 tile
// (SwiGLU: 128 rows x 64 gate + 64 up columns), stages 64-deep K slices of both operands through LDS
// with coalesced 128-byte row reads (8 lanes per row), and its four waves (2 x 2) each run 2 x 2
// v_mfma_f32_32x32x16_bf16 tiles from conflict-free ds_read_b128 fragments (segment s of row t is stored at
// s ^ ((t >> 1) & 7): the 16 lanes of a read phase then cover all 64 banks once).
// Global loads of slice c+1 are in flight while slice c feeds the MFMAs (LDS double buffer, one
// barrier per slice).
//
// BIT-IDENTICAL to k_mm32's prompt mode by construction: the same MFMA instruction with the same
// operand-to-k mapping inside each 64-chunk (lane half h, step q <- k = 32h + 8q .. +8), chained in
// the same order, and the same association of the four K-quarter partial sums
// (((0 + p0) + p1) + p2) + p3 -- k_mm32 spreads the quarters over its four waves and adds them through
// LDS, this kernel runs them back to back and folds each into `tot`.  A prompt row therefore has the
// same bits whether it was prefilled cold with 1,500 others or alone after a prefix-KV hit
// (tests/test_ops_gpu.py::test_gemm128_equals_mm32_bitwise).
#pragma once
#include "mm.cuh"

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8_t as_bf16x8(const u32x4_t& v) { return __builtin_bit_cast(bf16x8_t, v); }

#define G128_LD 64                        // LDS row = one 64-deep K slice (128 B); 16-byte segments XOR-swizzled
#define G128_SMEM (2 * 256 * G128_LD * 2) // two buffers of (128 A rows + 128 B rows) = 64 KB
#define G64_SMEM (2 * 192 * G128_LD * 2)  // MI = 1: two buffers of (64 A rows + 128 B rows) = 48 KB
#define G256_SMEM (3 * 384 * G128_LD * 2) // MI = 4: THREE buffers of (256 A rows + 128 B rows) = 144 KB, filled by LDS-DMA

// MI = 32-row tiles per wave along M: 1 -> a 64 x 128 block (round 4: for launches whose 128-row tiling leaves one block per CU --
// q|k|v at ~1,300 rows is 264 blocks of 32 slices, each slice one exposed memory round trip; 64-row blocks put two on every CU),
// 2 -> the 128 x 128 block (two blocks per CU), 4 -> a 256 x 128 block (round 3; one block per CU;
// NOT used by default: measured slower, see csm_engine.hip G256_MIN_ROWS).  The bigger block's bytes per flop fall from 1/64 to 1/85 --
// these kernels need 64 KB per CU and slice pair from the L2s, whose ~70 GB/s per CU (MI355X_MICROARCH.md) caps them near 46 % of the
// matrix cores' peak -- but its accumulators take the register file: one block per CU, one wave per SIMD, and that costs more than the
// bytes save.  Same per-tile MFMA chains and the same K-quarter fold in both: same bits.
template <int EPI, int HD, int DBG = 0, int MI = 2>
__global__ __launch_bounds__(256, MI <= 2 ? 2 : 1) void k_gemm128(const GemvArgs a, const int K, const int mt8, const long ldw) {
    constexpr int BM = 64 * MI;                             // rows per block
    constexpr int NA = MI * 2;                              // 16-byte A pieces per thread and slice (rows row0 + 32 i)
    constexpr int LROWS = BM + 128;                         // LDS rows per buffer
    extern __shared__ __align__(16) unsigned char g128_smem[];
    bf16_t* lds = reinterpret_cast<bf16_t*>(g128_smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (block L runs on XCD L % 8).  mt8 > 0 (thousands of rows): XCD c owns the row tiles
    // {c, c+8, ...} (mt8 of them) and walks them for one column tile after the other -- its share of x stays in its
    // L2 and each W tile is fetched once per XCD.  mt8 < 0 (-mt8 = row tiles, a few hundred to ~4,000 rows): XCD c
    // owns the column tiles {c, c+8, ...} and runs all row tiles of one column tile back to back -- every XCD is
    // busy whatever the row-tile count, the W tile stays in L2 and x streams from the Infinity Cache.
    // EPI_SLAB (residual projections when the tiles alone would not fill the chip): the grid is 4x larger and block
    // (tile, kq) runs only K quarter kq, writing its fp32 partial to slab[kq]; k_resid_norm then adds the four slabs
    // in quarter order -- the very association the one-block version uses, so the bits are the same.
    int L = blockIdx.x, kq = 0;
    if (EPI == EPI_SLAB) { kq = L & 3; L >>= 2; }
    const int xcd = L & 7, j = L >> 3;
    int mt, nt;
    if (mt8 > 0) { mt = (j % mt8) * 8 + xcd; nt = j / mt8; }
    else { mt = j % (-mt8); nt = (j / (-mt8)) * 8 + xcd; }
    // (round 4, measured and removed: XCD c owning ROW HALF c & 1 x COLUMN QUARTER c >> 1 -- 0.16 instead of 0.77 GB per gate/up launch over
    //  the fabric at 1,334 rows: gate/up 143.8 against 143.6 us, q|k|v 72.8 against 48.4; these launches are not bound by fabric bytes)
    if (mt * BM >= a.M) return;
    constexpr int NOUT = EPI == EPI_SWIGLU ? 64 : 128;      // output columns per block
    const int m0 = mt * BM, n0 = nt * NOUT;
    if (n0 >= a.N) return;

    // this thread's four 16-byte pieces of each operand slice: rows (tid >> 3) + 32 i, segment tid & 7
    const int seg = tid & 7, row0 = tid >> 3;
    const bf16_t* pa[NA];
    const bf16_t* pb[4];
#pragma unroll
    for (int i = 0; i < NA; ++i) pa[i] = a.x + (long)min(m0 + row0 + 32 * i, a.M - 1) * a.x_row_stride + a.x_row_offset + seg * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tr = row0 + 32 * i;
        const bf16_t* wsrc;
        int n;
        if (EPI == EPI_SWIGLU) {                            // B rows: [wave column][gate 32 | up 32]
            n = min(n0 + (tr >> 6) * 32 + (tr & 31), a.N - 1);
            wsrc = ((tr >> 5) & 1) ? a.w1 : a.w0;
        } else if (EPI == EPI_QKV_ROPE) {
            n = min(n0 + tr, a.N - 1);
            if (n < a.nq) wsrc = a.w0;
            else if (n < a.nq + a.nkv) { wsrc = a.w1; n -= a.nq; }
            else { wsrc = a.w2; n -= a.nq + a.nkv; }
        } else {
            n = min(n0 + tr, a.N - 1);
            wsrc = a.w0;
        }
        pb[i] = wsrc + (long)n * ldw + seg * 8;
    }
    // TWO K slices in flight in registers (round 3): slices s + 1 and s + 2 are requested while slice s feeds the MFMAs.  With one slice in flight a block paid one memory round trip per 64-deep
    // slice (32 slices x ~2 us at K = 2048: q|k|v 67 us for 17 GFLOP) -- latency, neither LDS nor matrix-core time.  The loop is
    // unrolled by two so that the register set of a slice is a compile-time choice (a runtime index would put them in scratch).
    u32x4_t ra[2][NA], rb[2][4];          // (MI = 2 only) compiler vector type: HIP's uint4 struct kept these in scratch memory
    // (asm loads + hand-written waits: with plain loads hipcc's wait-count pass put vmcnt(0) at the loop header -- it waited for the
    //  OLDER slice before requesting the next one, i.e. one slice in flight again.  Requests return in order, so "all but the 8
    //  youngest" is exactly the older register set.  No scratch in these kernels (tools/kres.sh): an asm-loaded register that is
    //  spilled before its wait would be saved with stale contents.)
#define G128_GLOAD(set, kc)                                                              \
    _Pragma("unroll") for (int i = 0; i < NA; ++i)                                       \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ra[set][i]) : "v"(pa[i] + (kc)) : "memory"); \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rb[set][i]) : "v"(pb[i] + (kc)) : "memory");
    // (NA + 4 requests per slice: "all but the NA + 4 youngest" have arrived)
#define G128_ARRIVED(set, younger)                                                       \
    {                                                                                    \
        if (!(younger)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 \
        else if (NA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");               \
        else if (NA == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               \
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                           \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(ra[set][i])); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(rb[set][i])); \
    }
    // rows row0 + 32 i share (row >> 1) & 7, so one swizzled segment serves all four pieces
    const int wseg = seg ^ ((row0 >> 1) & 7);
    bf16_t* const wbase = lds + row0 * G128_LD + wseg * 8;
#define G128_LWRITE(set, buf)                                                            \
    _Pragma("unroll") for (int i = 0; i < NA; ++i)                                       \
        *reinterpret_cast<u32x4_t*>(wbase + (buf) * (LROWS * G128_LD) + (32 * i) * G128_LD) = ra[set][i];      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                        \
        *reinterpret_cast<u32x4_t*>(wbase + (buf) * (LROWS * G128_LD) + (BM + 32 * i) * G128_LD) = rb[set][i];
    const int sw = (r >> 1) & 7;              // fragment rows are wm*32*MI + mi*32 + r: (row >> 1) & 7 == (r >> 1) & 7

    f32x16_t tot[MI][2], acc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot[mi][ni][i] = 0.f; acc[mi][ni][i] = 0.f; }

    // G128_NOFOLD (an A/B build only, `make nofold`): ONE K-ascending accumulation chain, no K-quarter fold -- NOT the
    // bits of k_mm32 / k_mmt / k_mmq any more (parity tests off), 64 instead of 128 accumulator registers at MI = 2: prices the constraint.
#ifdef G128_NOFOLD
#define G128_FOLD ++in_quarter;
#define G128_RESULT(mi_, ni_, i_) acc[mi_][ni_][i_]
#else
#define G128_FOLD                                                                        \
        if (++in_quarter == per_quarter) {                 /* end of a K quarter: fold the partial, restart the chain */ \
            in_quarter = 0;                                                              \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                            \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                         \
                    _Pragma("unroll") for (int i = 0; i < 16; ++i) { tot[mi][ni][i] += acc[mi][ni][i]; acc[mi][ni][i] = 0.f; } \
        }
#define G128_RESULT(mi_, ni_, i_) tot[mi_][ni_][i_]
#endif
    const int per_quarter = K / 256;
    const int c_first = EPI == EPI_SLAB ? kq * per_quarter : 0, ns = EPI == EPI_SLAB ? per_quarter : K / 64;   // slices c_first .. c_first + ns - 1
    int in_quarter = 0;
    // one slice from LDS buffer `buf`: the same MFMA chain per K quarter as before (k ascending; fold at the quarter's end)
#define G128_MMA(buf)                                                                    \
    {                                                                                    \
        const bf16_t* A = lds + (buf) * (LROWS * G128_LD) + (wm * 32 * MI + r) * G128_LD; \
        const bf16_t* B = lds + (buf) * (LROWS * G128_LD) + (BM + wn * 64 + r) * G128_LD; \
        _Pragma("unroll") for (int q = 0; q < (DBG == 2 ? 0 : 4); ++q) {                 \
            const int so = ((4 * h + q) ^ sw) * 8;                                       \
            const u32x4_t b0 = *reinterpret_cast<const u32x4_t*>(B + so);                \
            const u32x4_t b1 = *reinterpret_cast<const u32x4_t*>(B + 32 * G128_LD + so); \
            _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                          \
                const u32x4_t am = *reinterpret_cast<const u32x4_t*>(A + mi * 32 * G128_LD + so); \
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(am), as_bf16x8(b0), acc[mi][0], 0, 0, 0); \
                acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(am), as_bf16x8(b1), acc[mi][1], 0, 0, 0); \
            }                                                                            \
        }                                                                                \
        G128_FOLD                                                                        \
    }
    // (slices past the last one re-load the last one: no control flow around the register staging)
#define G128_KOFF(s_) ((c_first + min((s_), ns - 1)) * 64)
    if constexpr (MI <= 2) {
        // Every asm-loaded register set is requested AND consumed inside one loop iteration (two slices per iteration): no such value is
        // live across the loop's back edge or its entry, where the register allocator may insert copies -- a copy of a register whose
        // load is still in flight copies stale bits (seen in a first version of a 4-deep ring for k_attn_flash: v_mov at the back edge).
        G128_GLOAD(0, G128_KOFF(0))
        G128_ARRIVED(0, false)
        G128_LWRITE(0, 0)
        __syncthreads();
        // (sched_barrier: the requests leave before the MFMAs and the LDS writes come after them)
        for (int s = 0; s < ns; s += 2) {
            if (DBG != 1) { G128_GLOAD(1, G128_KOFF(s + 1)) G128_GLOAD(0, G128_KOFF(s + 2)) }   // slices s + 1 and s + 2
            __builtin_amdgcn_sched_barrier(0);
            G128_MMA(0)
            __builtin_amdgcn_sched_barrier(0);
            G128_ARRIVED(1, DBG != 1)                         // slice s + 1 (the older eight requests)
            G128_LWRITE(1, 1)
            __syncthreads();
            if (s + 1 < ns) { G128_MMA(1) }
            __builtin_amdgcn_sched_barrier(0);
            G128_ARRIVED(0, false)                            // slice s + 2
            G128_LWRITE(0, 0)
            __syncthreads();
        }
    } else {
        // MI = 4: the accumulators take the register file, so the operands go global -> LDS directly (global_load_lds_dwordx4: lane l's
        // 16 bytes land at base + 16 l, no staging registers) into THREE buffers: slices s + 1 and s + 2 are in flight while slice s
        // feeds the MFMAs.  A wave instruction fills one 8-row group (1 KB); the conflict-free image (segment g of row t at position
        // g ^ ((t >> 1) & 7)) is produced by letting lane (row, position p) FETCH segment p ^ ((t >> 1) & 7) of its row.  Wave w owns
        // the A groups w, w + 4, .. (8 of 32) and the B groups w, w + 4, .. (4 of 16): 12 requests per wave and slice, so "all but
        // the 12 youngest" (vmcnt) are the older slice; the barrier then makes every wave's groups visible to all.
        typedef __attribute__((address_space(3))) unsigned char g128_lb;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(g128_lb*)g128_smem;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int grow = lane >> 3, pos = lane & 7;
        const bf16_t* qa[8];
        const bf16_t* qb[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = (wv + 4 * i) * 8 + grow;
            qa[i] = a.x + (long)min(m0 + t, a.M - 1) * a.x_row_stride + a.x_row_offset + (pos ^ ((t >> 1) & 7)) * 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tr = (wv + 4 * i) * 8 + grow;
            const bf16_t* wsrc;
            int n;
            if (EPI == EPI_SWIGLU) { n = min(n0 + (tr >> 6) * 32 + (tr & 31), a.N - 1); wsrc = ((tr >> 5) & 1) ? a.w1 : a.w0; }
            else if (EPI == EPI_QKV_ROPE) {
                n = min(n0 + tr, a.N - 1);
                if (n < a.nq) wsrc = a.w0;
                else if (n < a.nq + a.nkv) { wsrc = a.w1; n -= a.nq; }
                else { wsrc = a.w2; n -= a.nq + a.nkv; }
            } else { n = min(n0 + tr, a.N - 1); wsrc = a.w0; }
            qb[i] = wsrc + (long)n * ldw + (pos ^ ((tr >> 1) & 7)) * 8;
        }
        // (m0 carries the LDS base of an LDS-DMA request; it is a reserved register, so each request saves and restores it instead of
        //  naming it as clobbered -- which hipcc warns "may lead to undefined behaviour")
        uint32_t m0_keep;
#define G256_DMA(buf, kc)                                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                    \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(qa[i] + (kc)), "s"(lds0 + (uint32_t)((buf) * (LROWS * 128) + (wv + 4 * i) * 1024)) : "memory"); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                    \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(m0_keep) : "v"(qb[i] + (kc)), "s"(lds0 + (uint32_t)((buf) * (LROWS * 128) + BM * 128 + (wv + 4 * i) * 1024)) : "memory");
        G256_DMA(0, G128_KOFF(0))
        G256_DMA(1, G128_KOFF(1))
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __syncthreads();
        int b0 = 0, b2 = 2;                                    // buffer of slice s / of slice s + 2
        for (int s = 0; s < ns; ++s) {
            if (DBG != 1) { G256_DMA(b2, G128_KOFF(s + 2)) }
            __builtin_amdgcn_sched_barrier(0);
            G128_MMA(b0)
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // slice s + 1 has landed (this wave's groups; the barrier covers the others')
            __syncthreads();
            b0 = b0 == 2 ? 0 : b0 + 1; b2 = b2 == 2 ? 0 : b2 + 1;
        }
#undef G256_DMA
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the re-loaded last slices: nothing may still be landing in registers the epilogue reuses)
    // epilogue: acc register i of lane (r, h) is row 8 (i / 4) + 4 h + (i % 4), column r of its 32 x 32 tile
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int ni = 0; ni < (EPI == EPI_SWIGLU ? 1 : 2); ++ni) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m0 + wm * 32 * MI + mi * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                if (EPI == EPI_SLAB) {
                    const int n = n0 + wn * 64 + ni * 32 + r;
                    if (m < a.M && n < a.N) a.slab[((long)kq * a.M + m) * a.N + n] = G128_RESULT(mi, ni, i);
                    continue;
                }
                if (EPI == EPI_SWIGLU) mm_finish<EPI, HD>(a, m, n0 + wn * 32 + r, G128_RESULT(mi, 0, i), G128_RESULT(mi, 1, i));
                else mm_finish<EPI, HD>(a, m, n0 + wn * 64 + ni * 32 + r, G128_RESULT(mi, ni, i), 0.f);
            }
        }
    }
}
#undef G128_GLOAD
#undef G128_ARRIVED
#undef G128_LWRITE
#undef G128_MMA
#undef G128_KOFF
#undef G128_FOLD
#undef G128_RESULT
