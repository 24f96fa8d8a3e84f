// The B = 1 persistent depth-decoder launch (dec_persist.cuh) in a code object of its own.  Its 107 KB of instructions stream through
// the 64 KB instruction caches thirty times per launch; from the engine's 2.6 MB code object the identical instructions ran 0.7 % slower
// (2090 / 2102 / 2105 against 2074 / 2084 / 2093 us per launch, alternating on one box, round 3) -- see DESIGN.md.
#include "dec_persist.cuh"

hipError_t csm_launch_dec_persist(const DecPersistArgs& p, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_persist, dim3(DP_NB), dim3(512), DP_LDS_BYTES, st, p);
    return hipGetLastError();
}
const void* csm_dec_persist_kernel() { return reinterpret_cast<const void*>(&k_dec_persist); }
