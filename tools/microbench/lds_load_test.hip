// Probe: global_load_lds_dwordx4 (gfx950) -- does lane l's 16 bytes land at lds_base + 16 l ?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4* g, uint4* out) {
    __shared__ __attribute__((aligned(16))) uint4 buf[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + wave * 64 + lane),
                                     (void __attribute__((address_space(3)))*)(buf + wave * 64), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[threadIdx.x] = buf[threadIdx.x];
}
int main() {
    std::vector<uint4> h(256);
    for (int i = 0; i < 256; ++i) h[i] = make_uint4(i, 1000 + i, 2000 + i, 3000 + i);
    uint4 *g, *o;
    hipMalloc(&g, 4096); hipMalloc(&o, 4096);
    hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, g, o);
    std::vector<uint4> r(256);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i].x != h[i].x || r[i].y != h[i].y || r[i].z != h[i].z || r[i].w != h[i].w;
    printf("global_load_lds_dwordx4: %d of 256 pieces misplaced (r[1] = %u %u %u %u, r[65] = %u)\n", bad, r[1].x, r[1].y, r[1].z, r[1].w, r[65].x);
    return 0;
}
