// The first depth-decoder step of a batch-1 frame as one launch (dec_first.cuh), in a code object of its own like k_dec_persist's.
#define CSM_DEC_PERSIST_ELSEWHERE    /* k_dec_persist: csm_dec_persist.hip */
#define CSM_DEC_FIRST_HERE
#include "dec_first.cuh"

hipError_t csm_launch_dec_first(const DecFirstArgs& p, hipStream_t st) {
    hipLaunchKernelGGL(k_dec_first, dim3(DP_NB), dim3(512), DF_LDS_BYTES, st, p);
    return hipGetLastError();
}
const void* csm_dec_first_kernel() { return reinterpret_cast<const void*>(&k_dec_first); }
