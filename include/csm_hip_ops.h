/*
 * Op-level entry points of libcsm_hip.so: thin launchers over the SAME kernels csm_prefill /
 * csm_depth / csm_frame_step sequence, exported so that the parity tests (tests/test_ops_gpu.py)
 * can check each hot-path op against the oracle in isolation.  Not part of the drop-in surface.
 * All pointers are device pointers; bf16 tensors are raw 16-bit words.
 */
#ifndef CSM_HIP_OPS_H
#define CSM_HIP_OPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* kind: 0 Linear; 1 Linear + residual; 2 RMSNorm + Linear (heads); 3 RMSNorm + fused q/k/v +
 * interleaved RoPE + KV append; 4 RMSNorm + gate/up + SiLU*up.  K = 512*{1,2,4,16}.
 * kind + 10 (10, 11, 13, 14): the wide-M matrix-core kernels (M >= 3 rows, prompt prefill and
 * batched decode) of kinds 0, 1, 3, 4; x must already be normalised, K % 256 == 0.
 * kind + 20 (20, 21, 23, 24): the same through the 128 x 128 LDS-tiled kernel that long prompts (M >= 512 rows)
 * take; bit-identical to kind + 10.
 * Replaces torchtune's RMSNorm / nn.Linear / Llama3ScaledRoPE / KVCache.update / FeedForward
 * as called from sesameai/models.py:158,173.                          */
int csm_op_gemv(int kind, int M, int K, int N, const void* x, long x_row_stride, long x_row_offset,
                const void* norm_scale, float eps, const void* w0, const void* w1, const void* w2,
                const void* resid, void* out, long ldo, void* normed_out, long normed_stride, int nt,
                int head_dim, int nq, int nkv, int kv_heads, int smax, int rows_per_seq, const int32_t* pos,
                const void* rope, void* kcache, void* vcache, void* stream);

/* GQA attention of M rows over keys [0,pos[m]] of their sequence's cache (SDPA at
 * sesameai/models.py:154,158,172-173).  part: fp32 scratch [M][H][nsplit][ceil((head_dim+2)/32)*32] (partial rows are whole 128-byte lines) if nsplit>1. */
int csm_op_attn(int M, int rows_per_seq, int H, int KV, int head_dim, int smax, int nsplit, const void* q,
                const void* kcache, const void* vcache, const int32_t* pos, void* out, float* part, void* stream);

/* Depth-decoder layer step "attention + output projection + residual" as ONE kernel (head_dim
 * 128, <= 32 cached keys): out[m] = resid[m] + Wo . SDPA(q[m], K[0..pos[m]], V[0..pos[m]]).
 * wo [N][H*128], resid/out [M][N] (may alias).                                               */
int csm_op_attn_oproj(int M, int rows_per_seq, int H, int KV, int smax, const void* q, const void* kcache,
                      const void* vcache, const int32_t* pos, const void* wo, int N, const void* resid, void* out,
                      void* stream);

/* masked 33-slot embedding sum (sesameai/models.py:155-157,193-203). */
int csm_op_embed_sum(int M, int ncb, int d, int audio_vocab, int text_vocab, const int32_t* tokens,
                     const uint8_t* mask, const void* text_emb, const void* audio_emb, void* h, void* stream);

/* sample_topk (sesameai/models.py:72-87): logits [B][ldl] bf16 (ldl = 512*ceil(V/512)),
 * noise optional [B][V] bf16 Exp(1); writes frame[b*ncb + codebook].                        */
int csm_op_sample(int B, int V, int ldl, const void* logits, float temperature, int topk, const void* noise,
                  const uint64_t* rng, int codebook, int ncb, int32_t* frame, void* stream);

/* Debug timeline of the persistent depth-decoder launch (csrc/dec_persist.cuh): the first call (host == NULL) turns it
 * on, later calls copy [32 steps][32] s_memrealtime ticks (100 MHz) of workgroup 100's gather wave to `host`.
 * Per step s: words l*4 + {0: x of q|k|v ready, 1: q/k/v in LDS, 2: x of the MLP ready, 3: layer-l rows published},
 * 16: x of the head ready, 17: logits in LDS, 18: code sampled, 19: next step's table rows in LDS, 20-23: head / sampler waves, 24-27: poll passes of the logits / head-x / q|k|v / partials sweeps.   CSM_E_STATE when the
 * handle does not run the persistent launch, or when the library is the product build: the stamps exist only in
 * libcsm_hip_timeline.so (make -C sesameai-tts_amd/csrc timeline, loaded with CSM_HIP_TIMELINE=1).                      */
int csm_debug_persist_stamps(csm_handle h, uint64_t* host, int n_words);

/* Which optional all-CU launches the handle runs (so a test can assert the path it means to cover): bit 0 persistent depth decoder
 * (B = 1), bit 1 batched persistent depth decoder (B = 2..32), bit 2 backbone attention block, bit 3 one-launch backbone layer
 * (bf16 stream), bit 4 one-launch backbone layer (e4m3 stream, fp8 mode), bit 5 the first depth-decoder step (codebook 1, both
 * positions) as one launch (csrc/dec_first.cuh, B = 1).                                                                           */
int csm_debug_fast_paths(csm_handle h);

/* How many frame-step graphs the handle has captured + instantiated since csm_create.  csm_frame_step keeps up to 4 captured steps in an
 * LRU keyed on (batch, top-k, temperature): revisiting a key replays, it does not capture again (tests/test_frame_gpu.py).              */
int csm_debug_graph_captures(csm_handle h);

/* Measurement hook of bench.py (roofline.dominant_kernels): times, on the handle's current state after at least one frame
 * step, `reps` back-to-back launches of (a) the persistent depth-decoder launch for batch B -- csrc/dec_persist.cuh at B = 1,
 * csrc/dec_persist_m.cuh at B = 2..32 -- and (b) a batch-1 backbone decode step (16 one-launch layers, csrc/bb_block.cuh),
 * each between HIP events on `stream`.  out[0] = avg us per decoder launch (NaN if the launch chain is in charge),
 * out[1] = weight bytes it streams per launch, out[2] = avg us per backbone layer launch (NaN unless B == 1 and the
 * one-launch layer is active), out[3] = weight bytes of one backbone layer, out[4] = avg us of the first decoder step as one
 * launch (csrc/dec_first.cuh; NaN unless B == 1 and it is in charge), out[5] = the weight bytes it streams.  Clobbers the current
 * frame's codes.                                                                                                              */
int csm_debug_time_kernels(csm_handle h, int B, int reps, float temperature, int topk, double* out /*[6] host*/, void* stream);

#ifdef __cplusplus
}
#endif
#endif
