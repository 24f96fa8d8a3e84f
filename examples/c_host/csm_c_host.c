/*
 * A host for libcsm_hip.so in plain C: no Python, no torch -- only the HIP runtime, RCCL and include/csm_hip.h.
 *
 *     csm_c_host <model.blob> <n_frames> [temperature [topk]]        (defaults 0.9 / 50; "1.0 1" = greedy)
 *
 * Reads a model blob (config, a prompt, every tensor of CsmWeights in declaration order; written by
 * tests/test_c_host_gpu.py from the same tensors the Python shim hands to csm_create) and drives the reference's frame loop
 * through the C ABI exactly as INTEGRATION.md section 3 lays it out:
 *
 *   weights packed into ONE device blob on GPU 0 -> csm_broadcast_weights on a communicator THIS host owns (ncclCommInitAll over
 *   $CSM_C_HOST_GPUS GPUs, default 1: the replica layout of SURVEY.md 8e -- one broadcast at start-up, no per-step collective) ->
 *   per GPU: csm_create (CsmWeights pointing into that GPU's blob) -> csm_reset -> csm_seed -> csm_prefill (the prompt) -> csm_depth
 *   (frame 0, committed) -> csm_frame_step x n (graph replays) -> csm_read_frames.
 *
 * Prints GPU 0's frames, one per line (every replica must have produced the same ones: same weights, same seed).  The test runs it
 * beside the Python host on the same blob: the frames must be identical.
 *
 * Build (examples/c_host/Makefile):  gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I../../include csm_c_host.c \
 *                                        -L../../sesameai-tts_amd/lib -lcsm_hip -L/opt/rocm/lib -lamdhip64 -lrccl
 */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "csm_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); exit(1); } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) DIE("%s: %s", #x, ncclGetErrorString(r_)); } while (0)
#define CSM(h, x) do { int rc_ = (x); if (rc_ != CSM_OK) DIE("%s = %d: %s", #x, rc_, csm_last_error(h)); } while (0)
#define MAX_GPUS 8
#define MAX_TENSORS 1024

static void must_read(void* dst, size_t n, FILE* f) {
    if (fread(dst, 1, n, f) != n) DIE("model blob is truncated");
}

/* the tensors of the file, in order: where each starts inside the packed device blob (256-byte aligned) */
static size_t t_off[MAX_TENSORS];
static int n_tensors, next_t;

/* pass 1 (image == NULL): sizes -> offsets; pass 2: the bytes into the host image */
static size_t scan_tensors(FILE* f, unsigned char* image) {
    size_t off = 0;
    int64_t n;
    n_tensors = 0;
    while (fread(&n, sizeof n, 1, f) == 1) {
        if (n_tensors == MAX_TENSORS) DIE("too many tensors in the blob");
        t_off[n_tensors++] = off;
        if (image) must_read(image + off, (size_t)n, f);
        else if (fseek(f, (long)n, SEEK_CUR) != 0) DIE("model blob is truncated");
        off += ((size_t)n + 255) / 256 * 256;
    }
    return off;
}

static const void* next_tensor(const unsigned char* dev_blob) {
    if (next_t >= n_tensors) DIE("model blob holds fewer tensors than CsmWeights names");
    return dev_blob + t_off[next_t++];
}

static void next_layer(const unsigned char* b, CsmLayerWeights* L) {
    L->wq = next_tensor(b); L->wk = next_tensor(b); L->wv = next_tensor(b); L->wo = next_tensor(b);
    L->w1 = next_tensor(b); L->w2 = next_tensor(b); L->w3 = next_tensor(b);
    L->sa_norm = next_tensor(b); L->mlp_norm = next_tensor(b);
}

int main(int argc, char** argv) {
    if (argc < 3) DIE("usage: %s <model.blob> <n_frames> [temperature [topk]]", argv[0]);
    const int n_frames = atoi(argv[2]);
    const float temperature = argc > 3 ? (float)atof(argv[3]) : 0.9f;
    const int topk = argc > 4 ? atoi(argv[4]) : 50;
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

    /* ---- the weights as ONE packed blob: on the host, then on GPU 0 ---- */
    const long start = ftell(f);
    const size_t bytes = scan_tensors(f, NULL);
    unsigned char* image = (unsigned char*)malloc(bytes);
    if (!image) DIE("out of host memory");
    fseek(f, start, SEEK_SET);
    scan_tensors(f, image);
    fclose(f);

    int n_gpus = getenv("CSM_C_HOST_GPUS") ? atoi(getenv("CSM_C_HOST_GPUS")) : 1, visible = 0;
    HIP(hipGetDeviceCount(&visible));
    if (n_gpus < 1 || n_gpus > visible || n_gpus > MAX_GPUS) DIE("CSM_C_HOST_GPUS=%d but %d GPU(s) are visible", n_gpus, visible);
    unsigned char* blob[MAX_GPUS];
    hipStream_t st[MAX_GPUS];
    int devs[MAX_GPUS];
    for (int g = 0; g < n_gpus; ++g) {
        devs[g] = g;
        HIP(hipSetDevice(g));
        HIP(hipMalloc((void**)&blob[g], bytes));
        HIP(hipStreamCreate(&st[g]));
    }
    HIP(hipSetDevice(0));
    HIP(hipMemcpy(blob[0], image, bytes, hipMemcpyHostToDevice));      /* only the root holds the checkpoint */
    free(image);

    /* ---- one broadcast on a communicator this host owns: the only collective of the replica layout ---- */
    ncclComm_t comm[MAX_GPUS];
    NCCL(ncclCommInitAll(comm, n_gpus, devs));
    NCCL(ncclGroupStart());
    for (int g = 0; g < n_gpus; ++g) {
        HIP(hipSetDevice(g));
        CSM(NULL, csm_broadcast_weights(blob[g], bytes, comm[g], 0, st[g]));
    }
    NCCL(ncclGroupEnd());
    for (int g = 0; g < n_gpus; ++g) { HIP(hipSetDevice(g)); HIP(hipStreamSynchronize(st[g])); }

    /* ---- one replica per GPU: the reference's frame loop ---- */
    int32_t* frames0 = NULL;
    int32_t eos0 = -2;
    for (int g = 0; g < n_gpus; ++g) {
        HIP(hipSetDevice(g));
        static CsmWeights w;
        memset(&w, 0, sizeof w);                           /* all-zero: no fp8 stream */
        next_t = 0;
        w.text_emb = next_tensor(blob[g]); w.audio_emb = next_tensor(blob[g]);
        for (int l = 0; l < cfg.backbone.n_layers; ++l) next_layer(blob[g], &w.bb[l]);
        w.bb_norm = next_tensor(blob[g]);
        for (int l = 0; l < cfg.decoder.n_layers; ++l) next_layer(blob[g], &w.dec[l]);
        w.dec_norm = next_tensor(blob[g]);
        w.projection = next_tensor(blob[g]); w.c0_head = next_tensor(blob[g]); w.audio_head_t = next_tensor(blob[g]);
        w.bb_rope = next_tensor(blob[g]); w.dec_rope = next_tensor(blob[g]);

        csm_handle h = NULL;
        CSM(NULL, csm_create(&cfg, &w, 1, S > 64 ? S : 64, n_frames + 8, &h));
        CSM(h, csm_reset(h, st[g]));
        CSM(h, csm_seed(h, 7, st[g]));
        int32_t *d_tok, *d_pos; uint8_t* d_msk;
        HIP(hipMalloc((void**)&d_tok, (size_t)S * row * sizeof(int32_t)));
        HIP(hipMalloc((void**)&d_msk, (size_t)S * row));
        HIP(hipMalloc((void**)&d_pos, (size_t)S * sizeof(int32_t)));
        HIP(hipMemcpyAsync(d_tok, tokens, (size_t)S * row * sizeof(int32_t), hipMemcpyHostToDevice, st[g]));
        HIP(hipMemcpyAsync(d_msk, mask, (size_t)S * row, hipMemcpyHostToDevice, st[g]));
        HIP(hipMemcpyAsync(d_pos, pos, (size_t)S * sizeof(int32_t), hipMemcpyHostToDevice, st[g]));
        CSM(h, csm_prefill(h, d_tok, d_msk, d_pos, 1, S, 1, st[g]));                          /* the prompt */
        CSM(h, csm_depth(h, 1, temperature, topk, NULL, NULL, NULL, NULL, 1, st[g]));        /* frame 0 */
        for (int i = 1; i < n_frames; ++i) CSM(h, csm_frame_step(h, 1, temperature, topk, 1, st[g]));
        int32_t* frames = (int32_t*)malloc((size_t)n_frames * cfg.n_codebooks * sizeof(int32_t));
        int32_t eos_at = -2;
        CSM(h, csm_read_frames(h, 1, 0, n_frames, frames, &eos_at, st[g]));
        if (csm_num_frames(h) != n_frames) DIE("csm_num_frames = %d, expected %d", csm_num_frames(h), n_frames);
        csm_destroy(h);
        if (g == 0) { frames0 = frames; eos0 = eos_at; }
        else if (eos_at != eos0 || memcmp(frames, frames0, (size_t)n_frames * cfg.n_codebooks * sizeof(int32_t)) != 0)
            DIE("the replica on GPU %d produced other frames than GPU 0's", g);
    }
    for (int i = 0; i < n_frames; ++i) {
        for (int c = 0; c < cfg.n_codebooks; ++c) printf(c ? " %d" : "%d", frames0[i * cfg.n_codebooks + c]);
        putchar('\n');
    }
    printf("eos_at %d\n", eos0);
    printf("replicas %d broadcast_bytes %zu\n", n_gpus, bytes);
    for (int g = 0; g < n_gpus; ++g) NCCL(ncclCommDestroy(comm[g]));
    return 0;
}
