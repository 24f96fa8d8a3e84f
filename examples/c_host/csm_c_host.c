/*
 * A host for libcsm_hip.so in plain C: no Python, no torch -- only the HIP runtime and include/csm_hip.h.
 *
 *     csm_c_host <model.blob> <n_frames>
 *
 * Reads a model blob (config, a prompt, every tensor of CsmWeights in declaration order; written by
 * tests/test_c_host_gpu.py from the same tensors the Python shim hands to csm_create), uploads it, and drives the reference's frame
 * loop through the C ABI exactly as INTEGRATION.md section 3 lays it out: csm_create -> csm_reset -> csm_seed -> csm_prefill (the
 * prompt) -> csm_depth (frame 0, committed) -> csm_frame_step x n (graph replays) -> csm_read_frames.  Prints the frames, one per
 * line.  The test runs it beside the Python host on the same blob: the frames must be identical.
 *
 * Build (examples/c_host/Makefile):  gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include csm_c_host.c \
 *                                        -L../../sesameai-tts_amd/lib -lcsm_hip -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csm_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); exit(1); } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define CSM(h, x) do { int rc_ = (x); if (rc_ != CSM_OK) DIE("%s = %d: %s", #x, rc_, csm_last_error(h)); } while (0)

static void must_read(void* dst, size_t n, FILE* f) {
    if (fread(dst, 1, n, f) != n) DIE("model blob is truncated");
}

/* next tensor of the blob: int64 byte count + data -> device memory */
static const void* next_tensor(FILE* f) {
    int64_t n;
    must_read(&n, sizeof n, f);
    void* host = malloc((size_t)n);
    if (!host) DIE("out of host memory");
    must_read(host, (size_t)n, f);
    void* dev = NULL;
    HIP(hipMalloc(&dev, (size_t)n));
    HIP(hipMemcpy(dev, host, (size_t)n, hipMemcpyHostToDevice));
    free(host);
    return dev;
}

static void next_layer(FILE* f, CsmLayerWeights* L) {
    L->wq = next_tensor(f); L->wk = next_tensor(f); L->wv = next_tensor(f); L->wo = next_tensor(f);
    L->w1 = next_tensor(f); L->w2 = next_tensor(f); L->w3 = next_tensor(f);
    L->sa_norm = next_tensor(f); L->mlp_norm = next_tensor(f);
}

int main(int argc, char** argv) {
    if (argc < 3) DIE("usage: %s <model.blob> <n_frames>", argv[0]);
    const int n_frames = atoi(argv[2]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) DIE("cannot open %s", argv[1]);
    char magic[4];
    must_read(magic, 4, f);
    if (memcmp(magic, "CSMB", 4) != 0) DIE("not a model blob");
    CsmConfig cfg;
    must_read(&cfg, sizeof cfg, f);
    int32_t S;
    must_read(&S, sizeof S, f);
    const int row = cfg.n_codebooks + 1;
    int32_t* tokens = (int32_t*)malloc((size_t)S * row * sizeof(int32_t));
    uint8_t* mask = (uint8_t*)malloc((size_t)S * row);
    int32_t* pos = (int32_t*)malloc((size_t)S * sizeof(int32_t));
    must_read(tokens, (size_t)S * row * sizeof(int32_t), f);
    must_read(mask, (size_t)S * row, f);
    for (int i = 0; i < S; ++i) pos[i] = i;

    static CsmWeights w;                                  /* zero-initialised: no fp8 stream */
    w.text_emb = next_tensor(f); w.audio_emb = next_tensor(f);
    for (int l = 0; l < cfg.backbone.n_layers; ++l) next_layer(f, &w.bb[l]);
    w.bb_norm = next_tensor(f);
    for (int l = 0; l < cfg.decoder.n_layers; ++l) next_layer(f, &w.dec[l]);
    w.dec_norm = next_tensor(f);
    w.projection = next_tensor(f); w.c0_head = next_tensor(f); w.audio_head_t = next_tensor(f);
    w.bb_rope = next_tensor(f); w.dec_rope = next_tensor(f);
    fclose(f);

    hipStream_t st;
    HIP(hipStreamCreate(&st));
    csm_handle h = NULL;
    CSM(NULL, csm_create(&cfg, &w, 1, S > 64 ? S : 64, n_frames + 8, &h));
    CSM(h, csm_reset(h, st));
    CSM(h, csm_seed(h, 7, st));
    int32_t *d_tok, *d_pos; uint8_t* d_msk;
    HIP(hipMalloc((void**)&d_tok, (size_t)S * row * sizeof(int32_t)));
    HIP(hipMalloc((void**)&d_msk, (size_t)S * row));
    HIP(hipMalloc((void**)&d_pos, (size_t)S * sizeof(int32_t)));
    HIP(hipMemcpyAsync(d_tok, tokens, (size_t)S * row * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_msk, mask, (size_t)S * row, hipMemcpyHostToDevice, st));
    HIP(hipMemcpyAsync(d_pos, pos, (size_t)S * sizeof(int32_t), hipMemcpyHostToDevice, st));
    CSM(h, csm_prefill(h, d_tok, d_msk, d_pos, 1, S, 1, st));                      /* the prompt */
    CSM(h, csm_depth(h, 1, 0.9f, 50, NULL, NULL, NULL, NULL, 1, st));             /* frame 0 */
    for (int i = 1; i < n_frames; ++i) CSM(h, csm_frame_step(h, 1, 0.9f, 50, 1, st));
    int32_t* frames = (int32_t*)malloc((size_t)n_frames * cfg.n_codebooks * sizeof(int32_t));
    int32_t eos_at = -2;
    CSM(h, csm_read_frames(h, 1, 0, n_frames, frames, &eos_at, st));
    if (csm_num_frames(h) != n_frames) DIE("csm_num_frames = %d, expected %d", csm_num_frames(h), n_frames);
    for (int i = 0; i < n_frames; ++i) {
        for (int c = 0; c < cfg.n_codebooks; ++c) printf(c ? " %d" : "%d", frames[i * cfg.n_codebooks + c]);
        putchar('\n');
    }
    printf("eos_at %d\n", eos_at);
    csm_destroy(h);
    return 0;
}
