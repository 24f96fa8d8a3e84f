// Diagnostic: the wide-M (M = 32) matrix-core kernels of csrc/mm.cuh as dependent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../sesameai-tts_amd/csrc/mm.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 60;
    const size_t WB = 700u << 20;
    bf16_t *w, *x, *y; 
    CK(hipMalloc(&w, WB)); CK(hipMemset(w, 0, WB));
    CK(hipMalloc(&x, 4 << 20)); CK(hipMemset(x, 0, 4 << 20));
    CK(hipMalloc(&y, 4 << 20)); CK(hipMemset(y, 0, 4 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int kind; int K, Nn; size_t wbytes; int kg = 1; };
    float* slab; CK(hipMalloc(&slab, 32u * 32 * 2048 * 4));
    std::vector<Case> cases = {
        {"swiglu  K1024 N8192 (33.5MB)", 4, 1024, 8192, 33554432}, 
        {"resid   K8192 N1024 NW4", 1, 8192, 1024, 16777216}, {"resid   K1024 N1024 (2MB)", 1, 1024, 1024, 2097152},
        {"store   K1024 N2051 (4.2MB)", 0, 1024, 2051, 4259840}, {"swiglu  K2048 N8192 (67MB)", 4, 2048, 8192, 67108864},
        {"swiglu  K1024 N8192 x PACKED", 6, 1024, 8192, 33554432}, {"slab    K1024 N1024 kg4 x PACKED", 7, 1024, 1024, 2097152, 4},
        {"slab    K8192 N1024 kg8 x PACKED", 7, 8192, 1024, 16777216, 8},
        {"slab    K1024 N1024 kg1", 5, 1024, 1024, 2097152, 1}, {"slab    K1024 N1024 kg2", 5, 1024, 1024, 2097152, 2},
        {"slab    K1024 N1024 kg4", 5, 1024, 1024, 2097152, 4},
        {"slab    K8192 N1024 kg8", 5, 8192, 1024, 16777216, 8}, {"slab    K8192 N1024 kg16", 5, 8192, 1024, 16777216, 16},
        {"slab    K8192 N1024 kg32", 5, 8192, 1024, 16777216, 32}, {"slab    K8192 N2048 kg16", 5, 8192, 2048, 33554432, 16},
        {"slab    K2048 N2048 kg2", 5, 2048, 2048, 8388608, 2}, {"slab    K2048 N2048 kg4", 5, 2048, 2048, 8388608, 4},
        {"slab    K2048 N2048 kg8", 5, 2048, 2048, 8388608, 8},
        {"slab    K8192 N2048 kg8", 5, 8192, 2048, 33554432, 8},
    };
    for (auto& c : cases) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            GemvArgs a; memset(&a, 0, sizeof a);
            const bf16_t* wl = w + (size_t)(i % 8) * (c.wbytes / 2);
            a.x = (i & 1) ? y : x; a.x_row_stride = c.K; a.M = 32; a.w0 = wl; a.w1 = wl + c.wbytes / 4; a.N = c.Nn;
            a.out = (i & 1) ? x : y; a.ldo = c.kind == 4 ? c.Nn : (c.Nn + 31) / 32 * 32; a.resid = a.out;
            dim3 grid((c.Nn + 31) / 32, 1, c.kg);
            a.slab = slab;
            if (c.kind == 5) { hipLaunchKernelGGL((k_mm32<EPI_SLAB, 64, 4>), grid, dim3(256), 0, st, a, c.K, 0, c.kg); continue; }
            if (c.kind == 6) { hipLaunchKernelGGL((k_mm32<EPI_SWIGLU, 64, 4, 0, true>), grid, dim3(256), 0, st, a, c.K, 0, 1); continue; }
            if (c.kind == 7) { hipLaunchKernelGGL((k_mm32<EPI_SLAB, 64, 4, 0, true>), grid, dim3(256), 0, st, a, c.K, 0, c.kg); continue; }
            if (c.kind == 4) hipLaunchKernelGGL((k_mm32<EPI_SWIGLU, 64, 4>), grid, dim3(256), 0, st, a, c.K, 0, 1);
            else if (c.kind == 1) hipLaunchKernelGGL((k_mm32<EPI_RESID, 64, 4>), grid, dim3(256), 0, st, a, c.K, 0, 1);
            else hipLaunchKernelGGL((k_mm32<EPI_STORE, 64, 4>), grid, dim3(256), 0, st, a, c.K, 0, 1);
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
        printf("%-40s %7.2f us/kernel  %6.2f TB/s\n", c.name, us, c.wbytes / us * 1e-6);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
