// Diagnostic: what does one DEPENDENT small kernel cost inside a hipGraph on MI355X?
// Variants of a "read the 2-4 KB vector the previous kernel wrote, stream W bytes of weights,
// write the next vector" kernel, chained N times in one graph.   hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big { char pad[200]; };   // kernarg size probe

__global__ void k_empty() {}
__global__ void k_empty_args(Big b, const float* x, float* y) { if (b.pad[0] == 77) y[0] = x[0]; }

// mode 0: x -> registers -> y (one dependent round trip)   mode 1: + LDS stage + barrier
// mode 2: + wave shuffle reduction   wbytes: weight bytes streamed per block (16 B/lane loads)
template <int MODE>
__global__ __launch_bounds__(256) void k_step(const uint4* __restrict__ w, long wchunks_per_block, const float* x, float* y, int n) {
    __shared__ float xs[2048];
    const int tid = threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4* wp = w + (long)blockIdx.x * wchunks_per_block;
    for (long c = tid; c < wchunks_per_block; c += 256) { uint4 v = wp[c]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    float s = 0.f;
    for (int i = tid; i < n; i += 256) { float v = x[i]; if (MODE >= 1) xs[i] = v; s += v; }
    if (MODE >= 1) { __syncthreads(); s = 0.f; for (int i = tid; i < n; i += 256) s += xs[(i + 1) % n]; }
    if (MODE >= 2) { for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64); }
    const int row = blockIdx.x * 4 + (tid >> 6);
    if ((tid & 63) == 0 && row < n) y[row] = s * 1e-3f + (float)(acc.x & 1);
}

int main() {
    const int N = 200, n = 1024;
    float *a, *b; uint4* w;
    CK(hipMalloc(&a, 8192)); CK(hipMalloc(&b, 8192)); CK(hipMalloc(&w, 64 << 20));
    CK(hipMemset(a, 0, 8192)); CK(hipMemset(b, 0, 8192)); CK(hipMemset(w, 0, 64 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Case { const char* name; int kind; int blocks; long wbytes; };
    std::vector<Case> cases = {
        {"empty 1 block", 0, 1, 0}, {"empty 256 blocks", 0, 256, 0}, {"empty+200B kernargs 256 blocks", 1, 256, 0},
        {"x->y regs, 256 blk", 10, 256, 0}, {"x->LDS->y, 256 blk", 11, 256, 0}, {"x->LDS->shfl->y, 256 blk", 12, 256, 0},
        {"x->LDS->shfl->y, 128 blk", 12, 128, 0}, {"x->LDS->shfl->y, 64 blk", 12, 64, 0}, {"x->LDS->shfl->y, 32 blk", 12, 32, 0},
        {"same + 2 MB weights, 256 blk", 12, 256, 2 << 20}, {"same + 4 MB weights, 256 blk", 12, 256, 4 << 20},
        {"same + 16 MB weights, 256 blk", 12, 256, 16 << 20}, {"same + 32 MB weights, 256 blk", 12, 256, 32 << 20},
        {"same + 32 MB weights, 512 blk", 12, 512, 32 << 20}, {"same + 32 MB weights, 1024 blk", 12, 1024, 32 << 20},
        {"same + 32 MB weights, 2048 blk", 12, 2048, 32 << 20},
    };
    for (auto& c : cases) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            float* x = (i & 1) ? b : a; float* y = (i & 1) ? a : b;
            long wc = c.wbytes / 16 / c.blocks;
            if (c.kind == 0) hipLaunchKernelGGL(k_empty, dim3(c.blocks), dim3(256), 0, st);
            else if (c.kind == 1) hipLaunchKernelGGL(k_empty_args, dim3(c.blocks), dim3(256), 0, st, Big(), x, y);
            else if (c.kind == 10) hipLaunchKernelGGL(k_step<0>, dim3(c.blocks), dim3(256), 0, st, w, wc, x, y, n);
            else if (c.kind == 11) hipLaunchKernelGGL(k_step<1>, dim3(c.blocks), dim3(256), 0, st, w, wc, x, y, n);
            else hipLaunchKernelGGL(k_step<2>, dim3(c.blocks), dim3(256), 0, st, w, wc, x, y, n);
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
        printf("%-42s %7.2f us/kernel\n", c.name, ms * 1e3 / (reps * N));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
