// Diagnostic: the 128 x 128 LDS-tiled prompt GEMM (csrc/gemm128.cuh) vs the 32 x 32 kernel at prefill sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../sesameai-tts_amd/csrc/gemm128.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int EPI, int DBG = 0> static int run(const char* name, int M, int K, int N, bf16_t* x, bf16_t* w, bf16_t* y, hipStream_t st, int pad = 0) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm128<EPI, 64, DBG>), hipFuncAttributeMaxDynamicSharedMemorySize, G128_SMEM));
    GemvArgs a; memset(&a, 0, sizeof a);
    a.x = x; a.x_row_stride = K + pad; a.M = M; a.w0 = w; a.w1 = w + (size_t)N * (K + pad); a.N = N; a.out = y; a.ldo = N; a.resid = y;
    const int nout = EPI == EPI_SWIGLU ? 64 : 128, mt8 = ((M + 127) / 128 + 7) / 8, nt = (N + nout - 1) / nout;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_gemm128<EPI, 64, DBG>), dim3(8 * nt * mt8), dim3(256), G128_SMEM, st, a, K, mt8, (long)(K + pad));
    CK(hipGetLastError());
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_gemm128<EPI, 64, DBG>), dim3(8 * nt * mt8), dim3(256), G128_SMEM, st, a, K, mt8, (long)(K + pad));
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, fl = 2.0 * M * K * (double)N * (EPI == EPI_SWIGLU ? 2 : 1);
    printf("pad %3d %-34s M=%5d K=%5d N=%5d  %8.1f us  %7.1f TFLOP/s\n", pad, name, M, K, N, us, fl / us * 1e-6);
    return 0;
}
int main() {
    bf16_t *w, *x, *y;
    CK(hipMalloc(&w, 512u << 20)); CK(hipMemset(w, 0, 512u << 20));
    CK(hipMalloc(&x, 256u << 20)); CK(hipMemset(x, 0, 256u << 20));
    CK(hipMalloc(&y, 256u << 20)); CK(hipMemset(y, 0, 256u << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (run<EPI_STORE, 1>("store, no global loads in loop", 6080, 2048, 3072, x, w, y, st)) return 1;
    if (run<EPI_STORE, 2>("store, no ds_read/MFMA", 6080, 2048, 3072, x, w, y, st)) return 1;
    if (run<EPI_STORE, 0>("store, full", 6080, 2048, 3072, x, w, y, st)) return 1;
    if (run<EPI_STORE, 0>("store, 1 row tile per XCD (M=1024)", 1024, 2048, 3072, x, w, y, st)) return 1;
    if (run<EPI_STORE, 0>("store, K=256 (epilogue-dominated)", 6080, 256, 3072, x, w, y, st)) return 1;
    for (int M : {1334, 6080}) {
        if (run<EPI_STORE>("store (qkv-like)", M, 2048, 3072, x, w, y, st)) return 1;
        if (run<EPI_RESID>("resid o-proj", M, 2048, 2048, x, w, y, st)) return 1;
        if (run<EPI_SWIGLU>("swiglu gate/up", M, 2048, 8192, x, w, y, st)) return 1;
        if (run<EPI_RESID>("resid down", M, 8192, 2048, x, w, y, st)) return 1;
    }
    return 0;
}
