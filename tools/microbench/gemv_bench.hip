// Diagnostic: time the production k_gemv instantiations as a dependent chain in a hipGraph,
// with depth-decoder / backbone shapes, against the floors measured by chain.hip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../sesameai-tts_amd/csrc/gemv.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KITERS, int R, int PRO, int EPI, int HD>
static void launch(const GemvArgs& a, int units, hipStream_t st) {
    const size_t smem = (size_t)1 * KITERS * 512 * 2 + 64;
    hipLaunchKernelGGL((k_gemv<1, KITERS, R, PRO, EPI, HD>), dim3((units + 3) / 4), dim3(256), smem, st, a);
}

int main(int argc, char** argv) {
    const int N = 124;                      // kernels per graph (4 layers x 31 steps)
    const size_t WB = 600u << 20;           // weight arena > Infinity Cache
    bf16_t *w, *x, *y, *scale, *rope, *kc, *vc; int* pos;
    CK(hipMalloc(&w, WB)); CK(hipMemset(w, 0, WB));
    CK(hipMalloc(&x, 1 << 20)); CK(hipMemset(x, 0, 1 << 20));
    CK(hipMalloc(&y, 1 << 20)); CK(hipMemset(y, 0, 1 << 20));
    CK(hipMalloc(&scale, 1 << 16)); CK(hipMemset(scale, 0, 1 << 16));
    CK(hipMalloc(&rope, 1 << 20)); CK(hipMemset(rope, 0, 1 << 20));
    CK(hipMalloc(&kc, 1 << 20)); CK(hipMalloc(&vc, 1 << 20)); CK(hipMalloc(&pos, 4096)); CK(hipMemset(pos, 0, 4096));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int kind; size_t wbytes; int layers; };
    // layers = how many distinct weight sets the chain cycles through (4 = decoder reuse across steps -> cache resident)
    std::vector<Case> cases = {
        {"dec qkv   (norm+rope)  3.1MB x4 sets", 3, 3145728, 4}, {"dec oproj (resid)     2.1MB x4 sets", 1, 2097152, 4},
        {"dec gateup(swiglu)   33.5MB x4 sets", 4, 33554432, 4}, {"dec down  (resid)    16.8MB x4 sets", 11, 16777216, 4},
        {"dec gateup(swiglu)   33.5MB x16 sets(HBM)", 4, 33554432, 16}, {"dec down (resid) 16.8MB x32 sets(HBM)", 11, 16777216, 32},
        {"dec qkv  3.1MB x124 sets (HBM)", 3, 3145728, 124}, {"head 2051x1024 4.2MB x31 sets", 2, 4200448, 31},
        {"proj 1024x2048 plain 4.2MB x1", 0, 4194304, 1},
        {"dec gateup R=2 (1 pair/wave)  x4 sets", 42, 33554432, 4}, {"dec gateup R=8 (4 pairs/wave) x4 sets", 48, 33554432, 4},
        {"dec down R=2 (2 rows/wave)    x4 sets", 112, 16777216, 4},
    };
    // ---- mixed chain: the real depth-decoder step: proj, 4 x (qkv, oproj, gateup, down), head; 31 steps ----
    // weights: 4 layer sets (58 MB each) + proj 4.2 MB + 31 heads x 4.2 MB, like the real frame
    for (int variant = 0; variant < 6; ++variant) {
        const int gu_nt = (variant == 1 || variant == 2 || variant == 5), dn_nt = (variant == 2), head_nt = (variant != 3 && variant != 5) ? 1 : 0;
        const int attn_nt = 0;
        hipGraph_t g; hipGraphExec_t ge;
        const int steps = 31, layers = 4;
        const size_t head0 = (size_t)(4 * 58 + 8) << 20;      // byte offset of the heads region
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int i = 0;
        for (int sidx = 0; sidx < steps; ++sidx) {
            {   // projection 1024x2048 (default policy)
                GemvArgs a; memset(&a, 0, sizeof a);
                a.x = (i & 1) ? y : x; a.M = 1; bf16_t* out = (i & 1) ? x : y; ++i;
                a.x_row_stride = 2048; a.w0 = w + ((size_t)(4 * 58) << 20) / 2; a.N = 1024; a.out = out; a.ldo = 1024;
                launch<4, 2, PRO_PLAIN, EPI_STORE, 64>(a, 512, st);
            }
            for (int l = 0; l < layers; ++l) {
                const bf16_t* wl = w + (size_t)l * (58u << 20) / 2;
                for (int op = 0; op < 4; ++op, ++i) {
                    GemvArgs a; memset(&a, 0, sizeof a);
                    a.x = (i & 1) ? y : x; a.M = 1; a.norm_scale = scale; a.eps = 1e-5f;
                    bf16_t* out = (i & 1) ? x : y;
                    if (op == 0) { a.nt = attn_nt; a.x_row_stride = 1024; a.w0 = wl; a.w1 = wl + 1024 * 1024; a.w2 = wl + 1280 * 1024; a.N = 1536; a.out = out; a.ldo = 1024;
                        a.nq = 1024; a.nkv = 256; a.smax = 32; a.rows_per_seq = 1; a.kv_heads = 2; a.pos_base = 5; a.rope = rope; a.kcache = kc; a.vcache = vc;
                        launch<2, 2, PRO_NORM, EPI_QKV_ROPE, 128>(a, 768, st); }
                    else if (op == 1) { a.nt = attn_nt; a.x_row_stride = 1024; a.w0 = wl + (2u << 20); a.N = 1024; a.out = out; a.ldo = 1024; a.resid = out; launch<2, 2, PRO_PLAIN, EPI_RESID, 64>(a, 512, st); }
                    else if (op == 2) { a.nt = gu_nt; a.x_row_stride = 1024; a.w0 = wl + (3u << 20); a.w1 = a.w0 + 8192 * 1024; a.N = 8192; a.out = out; a.ldo = 8192; launch<2, 4, PRO_NORM, EPI_SWIGLU, 64>(a, 4096, st); }
                    else { a.nt = dn_nt; a.x_row_stride = 8192; a.w0 = wl + (20u << 20); a.N = 1024; a.out = out; a.ldo = 1024; a.resid = out; launch<16, 1, PRO_PLAIN, EPI_RESID, 64>(a, 1024, st); }
                }
            }
            {   // head 2051x1024, a different one every step
                GemvArgs a; memset(&a, 0, sizeof a);
                a.x = (i & 1) ? y : x; a.M = 1; a.norm_scale = scale; a.eps = 1e-5f; bf16_t* out = (i & 1) ? x : y; ++i;
                a.nt = head_nt; a.x_row_stride = 1024; a.w0 = w + (head0 + (size_t)sidx * (5u << 20)) / 2; a.N = 2051; a.out = out; a.ldo = 2560;
                launch<2, 2, PRO_NORM, EPI_STORE, 64>(a, 1026, st);
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("decoder-step chain: gateup nt=%d down nt=%d head nt=%d : %7.2f us/step (18 kernels)\n", gu_nt, dn_nt, head_nt, ms * 1e3 / (reps * steps));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    for (auto& c : cases) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            GemvArgs a; memset(&a, 0, sizeof a);
            const bf16_t* wl = w + (size_t)(i % c.layers) * (c.wbytes / 2);
            a.x = (i & 1) ? y : x; a.M = 1; a.norm_scale = scale; a.eps = 1e-5f; a.nt = 0;
            bf16_t* out = (i & 1) ? x : y;
            switch (c.kind) {
                case 3: a.x_row_stride = 1024; a.w0 = wl; a.w1 = wl + 1024 * 1024; a.w2 = wl + 1280 * 1024; a.N = 1536; a.out = out; a.ldo = 1024;
                        a.nq = 1024; a.nkv = 256; a.smax = 32; a.rows_per_seq = 1; a.kv_heads = 2; a.pos = nullptr; a.pos_base = 5; a.rope = rope; a.kcache = kc; a.vcache = vc;
                        launch<2, 2, PRO_NORM, EPI_QKV_ROPE, 128>(a, 768, st); break;
                case 1: a.x_row_stride = 1024; a.w0 = wl; a.N = 1024; a.out = out; a.ldo = 1024; a.resid = out; launch<2, 2, PRO_PLAIN, EPI_RESID, 64>(a, 512, st); break;
                case 4: a.x_row_stride = 1024; a.w0 = wl; a.w1 = wl + 8192 * 1024; a.N = 8192; a.out = out; a.ldo = 8192; launch<2, 4, PRO_NORM, EPI_SWIGLU, 64>(a, 4096, st); break;
                case 11: a.x_row_stride = 8192; a.w0 = wl; a.N = 1024; a.out = out; a.ldo = 1024; a.resid = out; launch<16, 1, PRO_PLAIN, EPI_RESID, 64>(a, 1024, st); break;
                case 2: a.x_row_stride = 1024; a.w0 = wl; a.N = 2051; a.out = out; a.ldo = 2560; launch<2, 2, PRO_NORM, EPI_STORE, 64>(a, 1026, st); break;
                case 42: a.x_row_stride = 1024; a.w0 = wl; a.w1 = wl + 8192 * 1024; a.N = 8192; a.out = out; a.ldo = 8192; launch<2, 2, PRO_NORM, EPI_SWIGLU, 64>(a, 8192, st); break;
                case 48: a.x_row_stride = 1024; a.w0 = wl; a.w1 = wl + 8192 * 1024; a.N = 8192; a.out = out; a.ldo = 8192; launch<2, 8, PRO_NORM, EPI_SWIGLU, 64>(a, 2048, st); break;
                case 112: a.x_row_stride = 8192; a.w0 = wl; a.N = 1024; a.out = out; a.ldo = 1024; a.resid = out; launch<16, 2, PRO_PLAIN, EPI_RESID, 64>(a, 512, st); break;
                case 0: a.x_row_stride = 2048; a.w0 = wl; a.N = 1024; a.out = out; a.ldo = 1024; launch<4, 2, PRO_PLAIN, EPI_STORE, 64>(a, 512, st); break;
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (reps * N);
        printf("%-44s %7.2f us/kernel  %6.2f TB/s\n", c.name, us, c.wbytes / us * 1e-6);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
