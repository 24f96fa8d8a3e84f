// Diagnostic: k_sample (csrc/sampler.cuh) chained in a hipGraph: per-launch time by phase.
// -DSB_STOP=n builds a variant that leaves the kernel after phase n (1 per-thread maxima, 2 compaction, 3 wave-0
// selection + softmax + race) -- the difference between variants is the phase's share.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../sesameai-tts_amd/csrc/sampler.cuh"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 100, V = 2051, ldl = 2560, d = 1024, B = 1;
    std::vector<unsigned short> hl((size_t)N * ldl);
    unsigned s = 12345;
    for (auto& v : hl) { s = s * 1664525u + 1013904223u; float f = ((int)(s >> 9) % 2000 - 1000) * 0.003f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    bf16_t *lg, *emb, *out; int* frame; uint64_t* rng;
    CK(hipMalloc(&lg, hl.size() * 2)); CK(hipMemcpy(lg, hl.data(), hl.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&emb, (size_t)32 * V * d * 2)); CK(hipMemset(emb, 0, (size_t)32 * V * d * 2));
    CK(hipMalloc(&out, 8192)); CK(hipMalloc(&frame, 4096)); CK(hipMalloc(&rng, 16)); CK(hipMemset(rng, 0, 16));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int topk : {50, 1}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) {
            SampleArgs a; memset(&a, 0, sizeof a);
            a.logits = lg + (size_t)i * ldl; a.ldl = ldl; a.V = V; a.temperature = 0.9f; a.topk = topk; a.rng = rng;
            a.codebook = i % 31; a.ncb = 32; a.frame = frame; a.audio_emb = emb; a.audio_vocab = V; a.d = d; a.emb_out = out; a.emb_stride = d;
            CK(launch_sample(a, B, st));
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_sample topk=%d stop=%d: %.2f us/launch\n", topk,
#ifdef SB_STOP
               SB_STOP,
#else
               0,
#endif
               ms * 1e3 / (10 * N));
    }
    return 0;
}
