/*
 * The codec half of the boundary from plain C: libcsm_hip.so's include/mimi_hip.h with gcc, the HIP runtime and nothing else.
 *
 *     mimi_c_host <codec.blob> <out.pcm>
 *
 * The blob (tests/test_c_host_gpu.py) holds MimiConfig, an image of the MimiWeights the Python shim hands to mimi_create -- its
 * pointers are the shim's device addresses -- the tensors behind those addresses, and codes [32][T].  Every pointer field is looked
 * up in the tensor table and replaced by this process's copy (decode side only), then: mimi_create -> mimi_decode (stateless, what
 * generator.py:299 does) twice -- the launch chain, then the chunk's middle from the hipGraph -- and the PCM is written to <out.pcm>.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mimi_hip.h"

#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); exit(1); } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) DIE("%s: %s", #x, hipGetErrorString(e_)); } while (0)
#define MIMI(h, x) do { int rc_ = (x); if (rc_ != 0) DIE("%s = %d: %s", #x, rc_, mimi_last_error(h)); } while (0)

static void must_read(void* dst, size_t n, FILE* f) {
    if (fread(dst, 1, n, f) != n) DIE("codec blob is truncated");
}

static int n_tensors;
static uint64_t* old_addr;
static const float** new_addr;

static void reloc(const float** p) {
    if (!*p) return;
    for (int i = 0; i < n_tensors; ++i)
        if (old_addr[i] == (uint64_t)(uintptr_t)*p) { *p = new_addr[i]; return; }
    DIE("a weight pointer of the blob has no tensor behind it");
}
static void reloc_conv(MimiConv* c) { reloc(&c->w); reloc(&c->bias); }

int main(int argc, char** argv) {
    if (argc < 3) DIE("usage: %s <codec.blob> <out.pcm>", argv[0]);
    FILE* f = fopen(argv[1], "rb");
    if (!f) DIE("cannot open %s", argv[1]);
    char magic[4];
    must_read(magic, 4, f);
    if (memcmp(magic, "MIMB", 4) != 0) DIE("not a codec blob");
    static MimiConfig cfg;
    static MimiWeights w;
    int32_t sizes[2];
    must_read(sizes, sizeof sizes, f);
    if (sizes[0] != (int32_t)sizeof cfg || sizes[1] != (int32_t)sizeof w) DIE("struct sizes differ: blob %d / %d, header %zu / %zu", sizes[0], sizes[1], sizeof cfg, sizeof w);
    must_read(&cfg, sizeof cfg, f);
    must_read(&w, sizeof w, f);
    int32_t nt;
    must_read(&nt, sizeof nt, f);
    n_tensors = nt;
    old_addr = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)nt);
    new_addr = (const float**)malloc(sizeof(float*) * (size_t)nt);
    for (int i = 0; i < nt; ++i) {
        int64_t bytes;
        must_read(&old_addr[i], sizeof(uint64_t), f);
        must_read(&bytes, sizeof bytes, f);
        void* host = malloc((size_t)bytes);
        must_read(host, (size_t)bytes, f);
        void* dev = NULL;
        HIP(hipMalloc(&dev, (size_t)bytes));
        HIP(hipMemcpy(dev, host, (size_t)bytes, hipMemcpyHostToDevice));
        free(host);
        new_addr[i] = (const float*)dev;
    }
    int32_t T;
    must_read(&T, sizeof T, f);
    int32_t* codes = (int32_t*)malloc(sizeof(int32_t) * (size_t)cfg.n_codebooks * (size_t)T);
    must_read(codes, sizeof(int32_t) * (size_t)cfg.n_codebooks * (size_t)T, f);
    fclose(f);

    reloc(&w.codebooks); reloc(&w.proj_first); reloc(&w.proj_rest); reloc(&w.rope_freqs); reloc(&w.upsample);
    for (int l = 0; l < cfg.tr_layers; ++l) {
        MimiTrLayer* L = &w.tr[l];
        reloc(&L->ln1_w); reloc(&L->ln1_b); reloc(&L->in_proj); reloc(&L->out_proj); reloc(&L->ls1);
        reloc(&L->ln2_w); reloc(&L->ln2_b); reloc(&L->lin1); reloc(&L->lin2); reloc(&L->ls2);
    }
    reloc_conv(&w.conv_in);
    for (int j = 0; j < cfg.n_stages; ++j) { reloc_conv(&w.up[j]); reloc_conv(&w.res1[j]); reloc_conv(&w.res2[j]); }
    reloc_conv(&w.conv_out);
    w.has_encoder = 0;                                    /* decode side only */

    long hop = 2;
    for (int j = 0; j < cfg.n_stages; ++j) hop *= cfg.ratios[j];
    hipStream_t st;
    HIP(hipStreamCreate(&st));
    mimi_handle h = NULL;
    MIMI(NULL, mimi_create(&cfg, &w, T, 0, &h));
    int32_t* d_codes; float* d_pcm;
    HIP(hipMalloc((void**)&d_codes, sizeof(int32_t) * (size_t)cfg.n_codebooks * (size_t)T));
    HIP(hipMalloc((void**)&d_pcm, sizeof(float) * (size_t)(hop * T)));
    HIP(hipMemcpyAsync(d_codes, codes, sizeof(int32_t) * (size_t)cfg.n_codebooks * (size_t)T, hipMemcpyHostToDevice, st));
    float* pcm = (float*)malloc(sizeof(float) * (size_t)(hop * T));
    float* first = (float*)malloc(sizeof(float) * (size_t)(hop * T));
    for (int rep = 0; rep < 3; ++rep) {                  /* launch chain, capture, replay */
        MIMI(h, mimi_decode(h, d_codes, 1, T, (long)cfg.n_codebooks * T, T, d_pcm, 0, st));
        HIP(hipMemcpyAsync(pcm, d_pcm, sizeof(float) * (size_t)(hop * T), hipMemcpyDeviceToHost, st));
        HIP(hipStreamSynchronize(st));
        if (rep == 0) memcpy(first, pcm, sizeof(float) * (size_t)(hop * T));
        else if (memcmp(first, pcm, sizeof(float) * (size_t)(hop * T)) != 0) DIE("decode %d differs from the first", rep);
    }
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(pcm, sizeof(float), (size_t)(hop * T), o) != (size_t)(hop * T)) DIE("cannot write %s", argv[2]);
    fclose(o);
    printf("decoded %d frames -> %ld samples\n", T, hop * T);
    mimi_destroy(h);
    return 0;
}
