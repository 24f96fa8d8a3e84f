/*
 * libcsm_hip.so -- C ABI of the MI355X-native CSM speech-generation hot path.
 *
 * The reference (zenoran/sesameai-tts) has no FFI: its boundary for this path is the Python
 * surface `Model.generate_frame` / `Model.setup_caches` / `Model.reset_caches`
 * (sesameai/models.py:120-188) and `mimi.decode` (sesameai/generator.py:299).  This header is
 * the C ABI the build puts UNDER that surface; each entry point cites the reference
 * interface it replaces.  The Python shim (sesameai-tts_amd/sesameai/_abi.py) binds exactly
 * these symbols with ctypes; tests/test_abi.py checks that every one is exported.
 *
 * Conventions: plain pointers and sizes only (no torch types).  Every pointer marked
 * `dev` is a device pointer owned by the caller and must stay valid for the lifetime of the
 * handle (weights) or of the call (inputs/outputs).  `stream` is a hipStream_t passed as
 * void*.  All functions return 0 on success or a negative CSM_E_* code; the message is
 * available from csm_last_error().  A handle is not thread-safe: one handle per GPU
 * (the reference model is single-threaded too: one KV-cache set, tts_service.py:191).
 */
#ifndef CSM_HIP_H
#define CSM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSM_OK            0
#define CSM_E_INVALID    -1   /* bad argument / unsupported shape                       */
#define CSM_E_HIP        -2   /* a HIP runtime call failed                              */
#define CSM_E_STATE      -3   /* call order violated (e.g. frame step before prefill)   */
#define CSM_E_TOO_LONG   -4   /* position would exceed max_seq (generator.py:276-281)   */

#define CSM_MAX_LAYERS   32

/* One Llama stack = torchtune llama3_2(...) with tok_embeddings/output stripped
 * (sesameai/models.py:10-52). */
typedef struct CsmLlamaDims {
    int32_t n_layers, n_heads, n_kv_heads, dim, ffn, max_seq;
    float   norm_eps;
} CsmLlamaDims;

/* ModelArgs (sesameai/models.py:90-96) + the two flavours. */
typedef struct CsmConfig {
    CsmLlamaDims backbone;      /* llama-1B   : 16L 32H/8KV d2048 ffn8192 */
    CsmLlamaDims decoder;       /* llama-100M :  4L  8H/2KV d1024 ffn8192 */
    int32_t text_vocab;         /* 128256 */
    int32_t audio_vocab;        /* 2051   */
    int32_t n_codebooks;        /* 32     */
} CsmConfig;

/* bf16 tensors, row-major, nn.Linear layout [out][in]. */
typedef struct CsmLayerWeights {
    const void *wq, *wk, *wv, *wo;      /* [H*hd][d] [KV*hd][d] [KV*hd][d] [d][H*hd] */
    const void *w1, *w2, *w3;           /* gate [ffn][d], down [d][ffn], up [ffn][d] */
    const void *sa_norm, *mlp_norm;     /* [d] */
} CsmLayerWeights;

typedef struct CsmWeights {
    const void *text_emb;               /* [text_vocab][d_bb]                         */
    const void *audio_emb;              /* [n_codebooks*audio_vocab][d_bb]            */
    CsmLayerWeights bb[CSM_MAX_LAYERS];
    const void *bb_norm;                /* [d_bb]                                     */
    CsmLayerWeights dec[CSM_MAX_LAYERS];
    const void *dec_norm;               /* [d_dec]                                    */
    const void *projection;             /* [d_dec][d_bb]                              */
    const void *c0_head;                /* [audio_vocab][d_bb]                        */
    const void *audio_head_t;           /* [n_codebooks-1][audio_vocab][d_dec]: the reference's
                                           K-major audio_head (models.py:118,176) transposed once
                                           at load time so each logit is a contiguous row      */
    const void *bb_rope;                /* [max_seq][hd/2][2] bf16 (cos,sin), Llama3ScaledRoPE
                                           cache after model.to(bf16) (generator.py:343)       */
    const void *dec_rope;               /* same for the decoder's head_dim                     */
    /* ---- optional OCP-e4m3 weight stream for the decode step (BASELINE config 5) -------------
     * fp8 != 0: bb8/dec8 hold wq..w3 as e4m3 bytes [out][in] and bb8s/dec8s the per-output-row
     * fp32 scales (powers of two, so e4m3*scale is exactly the bf16 value in bb/dec above, which
     * the prefill / batched path keeps using).  Heads likewise.                                */
    int32_t fp8;
    CsmLayerWeights bb8[CSM_MAX_LAYERS], bb8s[CSM_MAX_LAYERS];
    CsmLayerWeights dec8[CSM_MAX_LAYERS], dec8s[CSM_MAX_LAYERS];
    const void *c0_head8, *c0_head8s;   /* [audio_vocab][d_bb] e4m3, [audio_vocab] f32          */
    const void *audio_head8, *audio_head8s;   /* [n_codebooks-1][audio_vocab][d_dec] e4m3, [..][audio_vocab] f32 */
} CsmWeights;

typedef struct CsmModel* csm_handle;

/* Model(config) + setup_caches(max_batch) (models.py:107-130): allocates KV caches
 * [L][B][KV][max_seq][hd] (backbone) / [L][B][KV][n_codebooks][hd] (decoder) and workspaces.
 * max_rows = largest B*S one csm_prefill call will carry.                                   */
int  csm_create(const CsmConfig* cfg, const CsmWeights* w /*dev ptrs*/, int max_batch, int max_rows,
                int max_frames, csm_handle* out);
void csm_destroy(csm_handle h);
const char* csm_last_error(csm_handle h);   /* h may be NULL: last create() error */

/* Model.reset_caches() (models.py:186-188): rewinds positions and the frame history.  The
 * caches need no zeroing: attention is bounded by position, never by a 2048-wide mask.     */
int csm_reset(csm_handle h, void* stream);

/* Seeds the on-device Philox sampler (the reference uses the global torch RNG, models.py:73).  Frame steps draw at
 * (seed, frame counter, sequence, codebook); the frame 0 of a slot refill (csm_prefill_slot) draws from a second key domain
 * (seed ^ salt, refill counter), so neither two refills nor a refill and a frame step ever share a noise stream.            */
int csm_seed(csm_handle h, uint64_t seed, void* stream);

/* The backbone half of Model.generate_frame (models.py:153-160) on B sequences x S rows:
 * masked embedding sum -> 16 layers with KV append at pos -> final norm of each sequence's
 * last row (kept in the handle as last_h).  tokens [B][S][33] i32, mask [B][S][33] u8,
 * pos [B][S] i32, all dev.  After it the internal position of sequence b is pos[b][S-1]+1.
 * prompt_mode != 0: always take the matrix-core (wide-M) kernels, so that a prompt row's result
 * does not depend on how many rows share the call (needed for bit-identical prefix-KV reuse);
 * 0: fewer than 16 rows run on the weight-stationary GEMV kernels of the decode step.       */
int csm_prefill(csm_handle h, const int32_t* tokens, const uint8_t* mask, const int32_t* pos,
                int B, int S, int prompt_mode, void* stream);

/* The depth half of Model.generate_frame (models.py:160-184): c0 head + sample, then 31
 * decoder steps.  Consumes last_h; writes the frame into out_frame [B][32] i32 (dev, may be
 * NULL) and into the handle's history.  topk==1 selects the deterministic lowest-index argmax.
 * forced [B][32] i32 (dev, may be NULL): teacher forcing -- fed back instead of the samples.
 * logits_out (dev, may be NULL): [32][B][audio_vocab] bf16 capture for tests.
 * noise (dev, may be NULL): [32][B][audio_vocab] bf16 Exp(1) draws replacing the Philox RNG.
 * commit != 0: append the frame to the history and stage it (or `forced`, if given) as the
 * input of the next csm_frame_step -- the first generated frame after a prefill.             */
int csm_depth(csm_handle h, int B, float temperature, int topk, const int32_t* forced,
              int32_t* out_frame, void* logits_out, const void* noise, int commit, void* stream);

/* One whole generated frame for the continuing loop (generator.py:283-294): the previous
 * frame (kept on device) is embedded with mask [1 x32, 0] at the internal position, one
 * backbone step, csm_depth, position += 1, frame appended to the history, EOS flag
 * (all 32 codes == 0, generator.py:285) accumulated per sequence.  No host sync; the whole
 * step is captured once into a hipGraph and replayed.  Up to 4 captured steps are kept per handle, keyed on
 * (B, topk, temperature) and replaced least-recently-used first: callers that alternate sampling parameters or batch
 * sizes (tts_service.py:175 uses 0.9/50, :266 0.8/40) replay, they do not capture again.
 * Batch 1 on the CSM-1B shapes: codebooks 1..31 and every backbone layer run as launches of 256 workgroups
 * that must all be resident at once (csrc/dec_first.cuh, csrc/dec_persist.cuh, csrc/bb_block.cuh).  Drive ONE frame loop per GPU (batch, or
 * one process per GPU); two loops sharing a GPU can starve each other, which ends -- after a bounded 50 ms spin, never a
 * hang -- in CSM_E_HIP from csm_read_frames; csm_reset makes the handle usable again.  CSM_PERSIST=0 / CSM_DEC_FIRST=0 / CSM_BB_BLOCK=0
 * select the plain launch chain.                                                              */
int csm_frame_step(csm_handle h, int B, float temperature, int topk, int use_graph, void* stream);
/* Copies the most recent frame [B][32] i32 to out_frame (dev) on the stream.                 */
int csm_copy_frame(csm_handle h, int B, int32_t* out_frame, void* stream);

/* Replace the "previous frame" inputs of the next csm_frame_step with caller data
 * (Model.generate_frame called directly with S==1, tts_service.py:225):
 * tokens [B][33] i32, mask [B][33] u8, pos [B] i32, dev.                                     */
int csm_set_step_inputs(csm_handle h, const int32_t* tokens, const uint8_t* mask, const int32_t* pos,
                        int B, void* stream);

/* Model.generate_frame for every frame AFTER the prompt, with the reference's own tensors (models.py:132-139 as called from
 * tts_service.py:224-241 and generator.py:283-294): tokens [B][1][33] int64, mask [B][1][33] bool (one byte each), pos [B][1]
 * int64, all dev and contiguous -> out_frame [B][32] int32 (dev).  = csm_set_step_inputs + csm_frame_step (hipGraph replay) +
 * csm_copy_frame in ONE call: the staging kernel reads the reference's dtypes (no conversion kernels on the host side), the
 * frame carries -1 if an all-CU launch gave up, and the handle's device is made current inside the call.  No host sync.       */
int csm_generate_frame_s1(csm_handle h, const int64_t* tokens, const uint8_t* mask, const int64_t* pos, int B,
                          float temperature, int topk, int32_t* out_frame, void* stream);

/* Per-slot reset and refill of a live batch (SURVEY.md 8b `csm_reset(handle, batch_slots, n)`; the reference is batch-1 and resets
 * its one cache set per utterance, generator.py:255).  Utterances of a batch end at different frames (generator.py:285): a finished
 * one is retired and its slot given to the next prompt while the other slots keep generating.
 * csm_reset_slots: position 0 and "no EOS yet" for the listed slots (host array); their K/V need no clearing.
 * csm_prefill_slot: the prompt rows tokens [S][33] / mask [S][33] / pos [S] (dev) run through the backbone into slot `slot`'s
 * caches, the depth pass produces the new utterance's frame 0 (written to out_frame [32] dev if given, and into the history at the
 * NEWEST global frame index, replacing that slot's entry there), and the frame is staged as the slot's input of the next
 * csm_frame_step.  The other slots' state is untouched: their frames are bit-identical to an undisturbed run, and
 * csm_copy_frame keeps returning every slot's own newest frame (the refilled slot's: its frame 0).  Call between
 * frame steps, on the stream that runs them.  A batch may also be FILLED slot by slot this way after csm_reset (prompts of
 * different lengths): the first call opens global frame 0.                                                                       */
int csm_reset_slots(csm_handle h, const int32_t* slots /*host*/, int n, void* stream);
int csm_prefill_slot(csm_handle h, int slot, const int32_t* tokens, const uint8_t* mask, const int32_t* pos, int S,
                     int prompt_mode, float temperature, int topk, int32_t* out_frame, void* stream);

/* Refill BESIDE the frame loop (round 4): csm_prefill_slot makes the other slots wait for the new prompt's whole prefill and a
 * batch-1 depth pass (4 ms at 190 rows, > 8 ms at 1,334).  Here the prompt runs a few backbone layers per call between frame
 * steps, in buffers of its own, and the new utterance's frame 0 is sampled BY THE NEXT FRAME STEP, in the batch, from the prompt's
 * last row (an inject node in the frame-step graph; the slot's position stays for that one step and its EOS word restarts).
 *   csm_refill_begin:   embeds the prompt rows (tokens [S][33] / mask [S][33] / pos [S], dev; pos must stay valid until the refill
 *                       completes) and parks the slot: until completion its rows of the frame steps are placeholders whose frames
 *                       the caller ignores and its position is HELD at S (however many steps the prompt takes, it never nears
 *                       max_seq).  One refill at a time per handle (CSM_E_STATE otherwise).  Needs csm_refill_supported(h, max_batch).
 *   csm_refill_advance: runs up to max_layers more backbone layers of the pending prompt.  Returns 1 when the prompt is complete --
 *                       the NEXT csm_frame_step then yields the utterance's frame 0 in that slot's row, at that step's global frame
 *                       index -- 0 while layers remain, < 0 on error.  Call both on the stream that runs the frame steps.
 *   csm_refill_supported: 1 when frame steps of B rows carry the inject node (the matrix-core decode path: B >= 3 by default,
 *                       CSM_WIDE / CSM_WIDE_MIN move it), else 0 -- the ONE predicate begin, the frame step and the host share.
 *                       While a refill is parked or waits to be sampled, csm_frame_step with a B for which this is 0 is refused
 *                       (CSM_E_STATE) instead of stepping the slot like a generating one; csm_reset_slots / csm_prefill_slot on
 *                       the slot whose prompt is still running are refused too, and drop a completed-but-unsampled refill.
 * The other slots' frames are bit-identical to an undisturbed run (their rows never see the refill).                          */
int csm_refill_supported(csm_handle h, int B);
int csm_refill_begin(csm_handle h, int slot, const int32_t* tokens, const uint8_t* mask, const int32_t* pos, int S, void* stream);
int csm_refill_advance(csm_handle h, int max_layers, void* stream);

/* Start-up weight broadcast (SURVEY.md 8b's csm_broadcast_weights; 8e: "one RCCL ncclBroadcast of the packed weight blob at start-up
 * over xGMI ... no per-step collective").  The reference has no counterpart (it is single-GPU: every process downloads its own
 * checkpoint, sesameai/generator.py:330-346); in the replica layout rank `root` holds the weights and every other
 * rank's blob -- `bytes` bytes of device memory at the same layout -- is filled by ONE in-place ncclBroadcast on `stream`.
 * The communicator is the CALLER's (an ncclComm_t made with ncclCommInitRank / ncclCommInitAll, passed as void*): this library does
 * not link RCCL and never creates a communicator -- it resolves ncclBroadcast from the RCCL instance already loaded in the process
 * (global scope, else the loaded librccl.so.1 by soname), i.e. the one the communicator belongs to, so a process never holds two
 * RCCL instances / bootstraps / sets of xGMI rings.  CSM_E_STATE when no RCCL is loaded, CSM_E_HIP when the collective fails
 * (csm_last_error(NULL) has the text).  Python hosts that use torch.distributed keep using dist.broadcast on the same blob
 * (sesameai/parallel.py: torch owns that communicator and does not hand out the ncclComm_t); plain-C hosts call this
 * (examples/c_host/csm_c_host.c).  The CsmWeights pointers handed to csm_create then point into the blob.      */
int csm_broadcast_weights(void* dev_blob, size_t bytes, void* rccl_comm, int root, void* stream);

/* What this handle runs, as one line of text: weight stream, the kernels of a batch-1 / batched backbone step, of the depth decoder and of
 * prompts, the frame-graph cache, and every CSM_* / MIMI_* switch set in the environment.  Writes at most n - 1 characters + NUL into buf (may
 * be NULL) and returns the length the whole text needs.  The reference has no counterpart (it has one eager path); bench.py records it with
 * every number it prints (config.paths).                                                                                              */
int csm_describe(csm_handle h, char* buf, int n);
/* Lists on stderr, once per process, the environment names under CSM_ / MIMI_ that no switch reads (csm_create calls it).           */
void csm_warn_unknown_switches(void);

/* History readback: frames [first, first + n) as [n][B][32] i32 into host memory (synchronises the stream);
 * eos_at[b] = global index of the first all-zero frame of sequence b, or -1.  The history is a RING of max_frames frames
 * (global frame g lives in row g % max_frames), so a frame loop may run for any number of frames as long as every frame is
 * read before max_frames newer ones exist; a range that has been overwritten is CSM_E_INVALID.                                */
int csm_num_frames(csm_handle h);
int csm_read_frames(csm_handle h, int B, int first, int n, int32_t* host_frames, int32_t* host_eos_at,
                    void* stream);
/* Device pointer to the history ring [max_frames][B_stride=max_batch][32] i32 (row = global frame % max_frames).           */
const int32_t* csm_frames_dev(csm_handle h);
/* Device pointer to last_h [max_batch][d_bb] bf16 (tests). */
const void* csm_last_h_dev(csm_handle h);

/* Bytes the dominant kernels stream per generated frame (unique weights + KV at mean
 * position p_mean); used by bench.py's roofline.                              */
double csm_bytes_per_frame(csm_handle h, int B, double p_mean);

#ifdef __cplusplus
}
#endif
#endif /* CSM_HIP_H */
