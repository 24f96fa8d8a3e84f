// correctness probe for the DPP wave reductions in csrc/common.cuh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "../../sesameai-tts_amd/csrc/common.cuh"
__global__ void k(const float* in, float* out_sum, float* out_max, float* ref_sum, int n) {
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        float v = in[t * 64 + threadIdx.x];
        float s = wave_sum(v), m = wave_max(v);
        float r = v;
        for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
        if (threadIdx.x == (t % 64)) { out_sum[t] = s; out_max[t] = m; ref_sum[t] = r; }
    }
}
int main() {
    const int n = 4096;
    float* h = (float*)malloc(n * 64 * 4);
    srand(1);
    for (int i = 0; i < n * 64; ++i) h[i] = (i / 64 < 64) ? ((i % 64) == (i / 64) ? 1.f : 0.f) : (float)rand() / RAND_MAX - 0.5f;
    float *d, *s, *m, *r; hipMalloc(&d, n * 64 * 4); hipMalloc(&s, n * 4); hipMalloc(&m, n * 4); hipMalloc(&r, n * 4);
    hipMemcpy(d, h, n * 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(64), dim3(64), 0, 0, d, s, m, r, n);
    float *hs = (float*)malloc(n * 4), *hm = (float*)malloc(n * 4), *hr = (float*)malloc(n * 4);
    hipMemcpy(hs, s, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hm, m, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hr, r, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < n; ++t) {
        double es = 0; float em = -1e30f;
        for (int l = 0; l < 64; ++l) { es += h[t * 64 + l]; em = fmaxf(em, h[t * 64 + l]); }
        if (fabs(hs[t] - es) > 1e-5 || hm[t] != em) { if (bad < 10) printf("t=%d sum %g (exact %g, shfl %g) max %g (exact %g)\n", t, hs[t], es, hr[t], hm[t], em); ++bad; }
    }
    printf("%d bad of %d\n", bad, n);
    return 0;
}
