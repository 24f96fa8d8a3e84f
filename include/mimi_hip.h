/*
 * libcsm_hip.so -- Mimi codec DECODE path (RVQ lookup + upsample + 8-layer transformer +
 * SEANet conv decoder), fp32, gfx950.
 *
 * Replaces `self._audio_tokenizer.decode(codes)` of the reference, i.e. moshi 0.2.2
 * `MimiModel.decode` as called at sesameai/generator.py:116 (stateless 10-frame stream
 * chunks), sesameai/generator.py:299 and tts_service.py:245 (whole utterance).
 * All pointers are device pointers unless stated; conv/linear weights are fp32 and already
 * re-laid-out by the loader (sesameai-tts_amd/sesameai/mimi.py) into the tap-major form the
 * kernels stream:  w[phase][tap][c_out][c_in].
 */
#ifndef MIMI_HIP_H
#define MIMI_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MIMI_MAX_TR_LAYERS 16
#define MIMI_MAX_STAGES     8

typedef struct MimiConfig {
    int32_t hidden;          /* 512  */
    int32_t codebook_size;   /* 2048 */
    int32_t codebook_dim;    /* 256  */
    int32_t n_codebooks;     /* 32   */
    int32_t n_semantic;      /* 1    */
    int32_t tr_layers, tr_heads, tr_ffn, tr_context;   /* 8, 8, 2048, 250 */
    float   rope_theta;      /* 10000 */
    float   norm_eps;        /* 1e-5  */
    int32_t n_stages;        /* 4 SEANet upsampling stages */
    int32_t ratios[MIMI_MAX_STAGES];   /* 8,6,5,4 */
    int32_t n_filters;       /* 64 */
    int32_t kernel, last_kernel, res_kernel;   /* 7, 3, 3 */
} MimiConfig;

/* one (possibly transposed) causal convolution in tap-major layout */
typedef struct MimiConv {
    const float* w;          /* [phases][taps][c_out][c_in]                            */
    const float* bias;       /* [c_out] or NULL                                        */
    int32_t c_in, c_out, taps, phases;
} MimiConv;

typedef struct MimiTrLayer {
    const float *ln1_w, *ln1_b, *in_proj /*[3d][d]*/, *out_proj /*[d][d]*/, *ls1 /*[d]*/;
    const float *ln2_w, *ln2_b, *lin1 /*[ffn][d]*/, *lin2 /*[d][ffn]*/, *ls2 /*[d]*/;
} MimiTrLayer;

typedef struct MimiWeights {
    const float* codebooks;  /* [n_codebooks][codebook_size][codebook_dim] = embedding_sum / clamp(usage,1e-5) */
    const float* proj_first; /* [codebook_dim][hidden]  rvq_first.output_proj, K-major    */
    const float* proj_rest;  /* [codebook_dim][hidden]  rvq_rest.output_proj,  K-major    */
    const float* rope_freqs; /* [head_dim/2] fp32: theta^(-2i/hd), built on the host       */
    const float* upsample;   /* [2 phases][2 taps][hidden] depthwise ConvTranspose1d k4 s2 */
    MimiTrLayer tr[MIMI_MAX_TR_LAYERS];
    MimiConv conv_in;
    MimiConv up[MIMI_MAX_STAGES];     /* ConvTranspose1d k=2r s=r: phases=r, taps=2     */
    MimiConv res1[MIMI_MAX_STAGES];   /* Conv1d k3  C -> C/2                             */
    MimiConv res2[MIMI_MAX_STAGES];   /* Conv1d k1  C/2 -> C                             */
    MimiConv conv_out;                /* Conv1d k3  n_filters -> 1                       */
    /* ---- ENCODE side (voice prompts, sesameai/generator.py:86); has_encoder == 0: absent ---- */
    int32_t has_encoder;
    const float* enc_conv_in_w;       /* [taps][n_filters]  (c_in = 1)                   */
    const float* enc_conv_in_b;       /* [n_filters]                                     */
    MimiConv enc_res1[MIMI_MAX_STAGES];   /* Conv1d k3  C -> C/2   (stage j works at C = n_filters << j) */
    MimiConv enc_res2[MIMI_MAX_STAGES];   /* Conv1d k1  C/2 -> C                         */
    MimiConv enc_down[MIMI_MAX_STAGES];   /* Conv1d k=2r stride r, C -> 2C, r = ratios reversed; taps = 2r */
    MimiConv enc_conv_out;            /* Conv1d k3  (n_filters << n_stages) -> hidden    */
    MimiTrLayer enc_tr[MIMI_MAX_TR_LAYERS];
    const float* downsample;          /* [4 taps][hidden][hidden] Conv1d k4 s2, replicate padding, no bias */
    const float* in_proj_first;       /* [codebook_dim][hidden]  rvq_first.input_proj    */
    const float* in_proj_rest;        /* [codebook_dim][hidden]  rvq_rest.input_proj     */
    const float* codebook_sqnorm;     /* [n_codebooks][codebook_size]  |e|^2             */
} MimiWeights;

typedef struct MimiDecoder* mimi_handle;

/* max_frames = longest code sequence one call (or one stream) will carry.
 * A handle's workspaces (activations, K-split partial tiles and their arrival tickets, the stream's history) are shared by all of its
 * calls: drive ONE HIP stream per handle at a time (a side-stream chunk decode and a whole-clip decode need two handles).            */
int  mimi_create(const MimiConfig* cfg, const MimiWeights* w, int max_frames, int reserved, mimi_handle* out);
void mimi_destroy(mimi_handle h);
const char* mimi_last_error(mimi_handle h);

/* codes: int32, element (b,k,t) at codes[b*stride_b + k*stride_k + t*stride_t] -- lets the
 * caller pass either a (B,32,T) tensor or the frame history [T][B][32] of csm_frames_dev()
 * without a transpose.  pcm: [B][hop*T] fp32.  stateful == 0: stateless decode of exactly
 * these T frames (what the reference does, whole utterance or per 10-frame chunk);
 * stateful != 0: continue the stream of the previous stateful call (B must be 1): conv left
 * contexts and the transformer KV window carry over, so chunked output == whole decode.
 * Codes >= codebook_size (CSM's vocab is 2051 > 2048) are clamped to codebook_size-1.
 * A stateless decode of T <= 32 frames (a streaming chunk) replays everything between the code
 * lookup and the output convolution from a hipGraph the handle captures at the second decode
 * of that T (kernel nodes only; MIMI_GRAPH_MAX_T=0 turns it off): same kernels, a quarter of
 * the host time per chunk.  One handle serves one thread at a time.                          */
int mimi_decode(mimi_handle h, const int32_t* codes, int B, int T, long stride_b, long stride_k,
                void* pcm, int stateful, void* stream);
/* stride_t is 1 for a (B,32,T) tensor; use mimi_decode_strided for other layouts. */
int mimi_decode_strided(mimi_handle h, const int32_t* codes, int B, int T, long stride_b, long stride_k,
                        long stride_t, void* pcm, int stateful, void* stream);
int mimi_reset_stream(mimi_handle h, void* stream);

/* MimiModel.encode (sesameai/generator.py:86): wav [B] rows of n_samples fp32 @ 24 kHz (row b at
 * wav + b*stride_b) -> codes [B][n_codebooks][T] int32, T = ceil(n_samples / hop).  Residual
 * vector quantisation = nearest centroid per level (first index on ties).  Uses the decoder's
 * work buffers: it ends any stateful decode stream.  n_samples <= hop * max_frames.            */
int mimi_encode(mimi_handle h, const float* wav, long n_samples, long stride_b, int B, int32_t* codes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
