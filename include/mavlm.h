/* mavlm.h - C ABI of the MI355X-native recurrent memory-token + Memory-Fuser path.
 *
 * Drop-in boundary for the hot path that 1023604540/Memory-Augmented-VLM layers on LLaVA-OneVision.  The
 * reference has no FFI of its own (pure Python on ATen); each entry point below names the reference interface
 * it replaces (file:line relative to the reference root).  Plain pointers and sizes only - no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked "host"; tensors are row-major and contiguous unless a
 *     leading dimension (ld*, in elements) is passed; 16-bit tensors are bf16 (dtype 0) or fp16 (dtype 1);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises or
 *     allocates, so every call can be captured into a hipGraph;
 *   - return value: 0 = success, >0 = hipError_t from the runtime, <0 = MAVLM_E_* argument error.
 *     Nothing throws.  The handle is not re-entrant (the reference module is not either: it owns the
 *     mutable memory_cache list, MemoryController.py:85-87).
 */
#ifndef MAVLM_H_
#define MAVLM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAVLM_ABI_VERSION 3
#define MAVLM_MAX_DEPTH 8
#define MAVLM_MAX_BATCH 64

#define MAVLM_E_ARG (-1)        /* null pointer / bad size */
#define MAVLM_E_SHAPE (-2)      /* shape not supported by the kernels */
#define MAVLM_E_STATE (-3)      /* weights / buffers not bound, or reset missing */

typedef struct mavlm_ctx mavlm_ctx;

/* Hyper-parameters: `Config` (llava/model/memory_module/MemoryController.py:7-18) as overridden at
 * llava/model/llava_arch.py:117-129, plus the FIFO cap (MemoryController.py:153-154). */
typedef struct mavlm_config {
  int32_t hidden;           /* D, mm_hidden_size; multiple of 128; head_dim = D/heads <= 128 (D = 896, 1024 ...) or 448 (D = 3584) */
  int32_t heads;            /* H, mm_num_attention_heads (8) */
  int32_t patches;          /* P, patch_size (196) */
  int32_t mem_tokens;       /* M, num_memory_tokens (8) */
  int32_t depth;            /* L, depth (2) */
  int32_t inter;            /* I, mm_intermediate_size (4D) */
  int32_t cache_cap;        /* FIFO length (10) */
  int32_t max_chunk_frames; /* largest F passed to mavlm_step (32) */
  int32_t dtype;            /* 0 = bf16, 1 = fp16 */
  float eps;                /* mm_layer_norm_eps (1e-12) */
  int32_t batch;            /* B: independent videos stepped together over ONE set of weights (0 / 1 = a single video, the
                             * reference's limit, llava_arch.py:436).  The memory rows of the B videos are stacked into every
                             * weight-shared GEMM / LayerNorm launch ([B*M*P, D]); the attention serves B*heads (video, head)
                             * pairs, each video over its own keys (wide heads, 448: one attention launch per video).  Buffers: see mavlm_buffers,
                             * mavlm_weights.mem0; protocol: mavlm_step_batch / mavlm_fuse_emit_batch. */
  int32_t q_token0;         /* Row shard of ONE video over the ranks of a process group (SURVEY.md section 8e option 2): this */
  int32_t q_tokens;         /* context computes the memory tokens [q_token0, q_token0 + q_tokens) of every step - q projection,
                             * attention rows, dense, LayerNorm and MLP are row-independent - and writes them into its rows
                             * of the FIFO slot; the host all-gathers the other ranks' rows into the slot before the next
                             * mavlm_step (the evolution reads ALL rows of every cached memory as keys).  mem0 = the owned
                             * rows [q_tokens*P, D]; frame scores = this shard's partial sums (all-reduce them).
                             * q_tokens = 0: all tokens (no shard).  Not combined with batch > 1. */
  int32_t fused_ln;         /* Residual blocks (dense + residual + LayerNorm) as ONE kernel: MAVLM_LN_AUTO (0) = wherever the
                             * shape takes it (the value of the process-wide hook mavlm_set_fused_layernorm at mavlm_create is
                             * snapshotted into the context: a context's schedule never changes under it), MAVLM_LN_NEVER (1) =
                             * this context always runs GEMM + row LayerNorm kernel.  See MAVLM_LN_MAX_STREAMS. */
} mavlm_config;
#define MAVLM_LN_AUTO 0
#define MAVLM_LN_NEVER 1
/* Forward progress of the fused Residual kernel.  Its workgroups WAIT (bounded spin) for the row statistics of the other
 * N / 256 workgroups of their row block, in an ordinary launch.  Progress rests on one property of the hardware dispatcher:
 * the workgroups of a launch are handed out in id order, so that at any moment at most ONE row block per XCD is incomplete
 * and at most N / 256 - 1 <= 3 waiting workgroups per launch hold a CU of an XCD (32 CUs).  With n launches running
 * concurrently on n streams that is 3 n CUs: the library's contract is n <= MAVLM_LN_MAX_STREAMS (8: 24 of 32 CUs; never
 * approached by a waiter-only XCD).  A host that runs more streams creates their contexts with fused_ln = MAVLM_LN_NEVER
 * (the Python MemoryPathPool does).  If the property ever failed, the bounded spin turns the stall into a wrong result of
 * that launch plus the timeout word (mavlm_ln_status_async) - never a hang. */
#define MAVLM_LN_MAX_STREAMS 8

/* One `Attention` block (MemoryController.py:31-57) minus its K/V projections.  Weights [out,in] 16-bit
 * (nn.Linear layout), biases and LayerNorm affine parameters fp32.
 * Head padding: with Dp = heads*128, every projection that produces or consumes per-head columns is packed with each
 * head's head_dim rows/columns followed by (128 - head_dim) zeros: wq [Dp,D], bq [Dp], wo [D,Dp] (and the K/V
 * packs below).  For head_dim = 128 this is the plain nn.Linear weight. */
typedef struct mavlm_attn_weights {
  const void* wq;  const float* bq;     /* q_proj            [D,D],[D] */
  const void* wo;  const float* bo;     /* residual.dense    [D,D],[D] */
  const float* ln_g; const float* ln_b; /* residual.layernorm [D]      */
} mavlm_attn_weights;

/* Borrowed device pointers into the packed parameter storage (packed once by the host module from the
 * reference state-dict names listed in SURVEY.md §8b). */
typedef struct mavlm_weights {
  const void* mem0;                     /* [B*M*P, D] 16-bit: initial_memory + memory_pos_embed (MemoryController.py:123), repeated for each of the B = max(batch,1) videos */
  const void* w_kv_seg; const float* b_kv_seg; /* [2*L*Dp, D]: rows K_0,V_0,K_1,V_1,... of layers[l].memory_segment_fusion_attention.{k,v}_proj (head-padded) */
  mavlm_attn_weights layer_attn[MAVLM_MAX_DEPTH];
  const void* w_up[MAVLM_MAX_DEPTH];   const float* b_up[MAVLM_MAX_DEPTH];   /* layers[l].mlp.0          [I,D] */
  const void* w_down[MAVLM_MAX_DEPTH]; const float* b_down[MAVLM_MAX_DEPTH]; /* layers[l].residual.dense [D,I] */
  const float* ln2_g[MAVLM_MAX_DEPTH]; const float* ln2_b[MAVLM_MAX_DEPTH];  /* layers[l].residual.layernorm */
  mavlm_attn_weights evo;               /* memory_update_attention */
  const void* w_kv_evo; const float* b_kv_evo; /* [2Dp, D]: rows K,V of memory_update_attention.{k,v}_proj (head-padded) */
  const void* w_f1; const float* b_f1;  /* memory_fuser.0 [I,D]  (llava_arch.py:132-136) */
  const void* w_f2; const float* b_f2_type0; /* memory_fuser.2 [D,I]; bias + token_type_embedding[0] (llava_arch.py:548-553) */
  const void* type1;                    /* [D] 16-bit token_type_embedding[1] (llava_arch.py:554) */
} mavlm_weights;

/* Caller-allocated state and scratch (the module owns them as torch tensors). */
typedef struct mavlm_buffers {
  void* mem_ring;       /* [cache_cap, B, M*P, D]  16-bit  memory_cache entries, slot = step % cache_cap (B = max(batch,1): a slot
                         * holds the memory of every video of the row batch - one contiguous [B*M*P, D] GEMM operand) */
  void* evo_kv_ring;    /* [B, cache_cap, M*P, 2Dp] 16-bit K|V projections of each cached memory (projected once); the keys
                         * of ONE video are contiguous over its slots */
  void* workspace;      /* mavlm_workspace_bytes() bytes, 256-B aligned, ZERO-FILLED when bound (it holds the launch counter and
                         * the exchange granules of the fused dense + residual + LayerNorm epilogue, mavlm_linear_ln) */
  size_t workspace_bytes;
} mavlm_buffers;

/* --- lifecycle -------------------------------------------------------------------------------------- */
int mavlm_abi_version(void);
/* replaces TransformerProjector.__init__ (MemoryController.py:74-87) - host-side handle only */
int mavlm_create(const mavlm_config* cfg, mavlm_ctx** out);
void mavlm_destroy(mavlm_ctx* ctx);
size_t mavlm_workspace_bytes(const mavlm_config* cfg);
/* byte offsets of the 10 workspace regions {kv_seg, q, ctx, a, h, pre(fp32), mA, mB, lse2, colsum_part}: lets the
 * parity tests read the intermediates of the last sub-layer after a step (stage-wise checks) */
int mavlm_workspace_layout(const mavlm_config* cfg, size_t* offsets, int32_t n);
int mavlm_bind_weights(mavlm_ctx* ctx, const mavlm_weights* w);
/* (the exchange scratch of the fused Residual kernel inside the workspace is zero-filled by the library itself: an
 * asynchronous memset on the stream of the first step after the bind - which therefore must not be inside a hipGraph
 * capture, MAVLM_E_STATE -; mavlm_bind_weights / mavlm_bind_buffers / mavlm_reset discard a mavlm_project_chunk prefetch) */
int mavlm_bind_buffers(mavlm_ctx* ctx, const mavlm_buffers* b);
/* Health of the fused Residual kernel's exchange in this context's workspace, without a synchronisation: enqueues a 16-byte
 * device-to-host copy of the control words {arrivals, launch counter, timeout flag, -} into `host16` (pinned host memory the
 * caller owns) on `stream`; the caller reads host16[2] once an event recorded behind the copy has completed.  Non-zero =
 * some launch since the last clear gave up waiting for a partner (its output is wrong).  clear != 0 also enqueues a reset of
 * the flag behind the copy.  Returns 1 when the configuration never takes the fused form (nothing enqueued).  The Python
 * engine posts one probe per video (at `memory_cache = []`) and raises MavlmError at the next one. */
int mavlm_ln_status_async(mavlm_ctx* ctx, void* host16, int32_t clear, void* stream);

/* --- per-video protocol ------------------------------------------------------------------------------ */
/* replaces `recurrent_model.memory_cache = []` (llava_arch.py:532) */
int mavlm_reset(mavlm_ctx* ctx);
/* number of valid cache entries = min(steps, cache_cap); slot of the newest entry; steps since reset */
int mavlm_cache_len(const mavlm_ctx* ctx);
int mavlm_newest_slot(const mavlm_ctx* ctx);
int mavlm_steps(const mavlm_ctx* ctx);

/* replaces TemporalPositionalEncoding.forward (position_encoding.py:38-69): out[t] = x[t] + table[idx[t]].
 * Index range checking (ValueError, :73-76) is done by the host wrapper on the host copy of idx. */
int mavlm_pe_add(const void* x, const int64_t* idx, const void* table, void* out, int32_t T, int32_t P, int32_t D,
                 int32_t dtype, void* stream);

/* replaces TransformerProjector.forward (MemoryController.py:118-158) for one chunk seg[F,P,D]:
 * memory evolution over the FIFO (t>0, :89-115), L formation layers (:59-72,132-133), append to the ring.
 * frame_scores: null, or [F] (fp32 if scores_f32 else 16-bit) = probs.sum(heads).sum(queries).view(F,P).mean(1)
 * of the last layer (:135-139). */
int mavlm_step(mavlm_ctx* ctx, const void* seg, int32_t F, void* frame_scores, int32_t scores_f32, void* stream);
/* the same step for the B videos of a row-batched context (config.batch = B): segs = HOST array of B device pointers, the
 * chunk [F,P,D] of each video (all videos step with the same F - group videos by shape); frame_scores: null or [B, F].
 * The reference runs one video per forward (llava_arch.py:436, batch 1 per GPU); stacking the memory rows of several is
 * exact for every row-wise operator, the attention keeps each video on its own keys.  Not bit-identical to B single-video
 * contexts where the attention schedule differs (fp32 summation order of the cut units); same rounding points. */
int mavlm_step_batch(mavlm_ctx* ctx, const void* const* segs, int32_t F, void* frame_scores, int32_t scores_f32, void* stream);
int mavlm_batch(const mavlm_ctx* ctx);      /* B = max(config.batch, 1) */
/* K/V projection of a chunk ahead of its mavlm_step (the one GEMM of a step that does not read the memory): lets a host
 * overlap it with an exchange the step has to wait for - the all-gather of the previous memory's rows in the row-sharded
 * mode.  The next mavlm_step with the same (seg, F) skips the projection (and waits for it by an event when it runs on another
 * stream); a step with another chunk, mavlm_reset, mavlm_bind_weights and mavlm_bind_buffers discard it.  Single videos only
 * (MAVLM_E_STATE for a row batch); not inside a graph capture. */
int mavlm_project_chunk(mavlm_ctx* ctx, const void* seg, int32_t F, void* stream);
/* The same for the step AFTER the next one, on ANOTHER stream: call it before mavlm_step(chunk t) with chunk t + 1 and a side
 * stream, and the projection of chunk t + 1 - the largest GEMM of a step at few memory tokens - runs beside step t's small-grid
 * kernels (round 4).  The library keeps two chunk K/V buffers (single videos) and orders the streams with events: the projection
 * waits for the last step that read its buffer, the consuming step waits for the projection; nothing to synchronise for the host.
 * Not inside a graph capture (MAVLM_E_STATE).  `seg` must stay valid until the consuming step has been enqueued. */
int mavlm_project_chunk_ahead(mavlm_ctx* ctx, const void* seg, int32_t F, void* stream);
/* number of mavlm_step calls of this context that found (and used) a projection left by mavlm_project_chunk[_ahead] */
int mavlm_prefetch_hits(const mavlm_ctx* ctx);

/* replaces memory_fuser(cat(memory_cache)) + token_type add + fine-frame gather/add + prompt/newline concat
 * (llava_arch.py:513-524,545-554,620-629,708-731).  Writes
 *   out = [mem_prompt(10) ; fused memory (n*M*P rows, oldest first) ; newline ; frame_prompt(9) ; fine (n_fine*P) ; newline]
 * x_pe: [T,P,D] PE-added frames; fine_idx: [n_fine] int64 frame indices into x_pe.  Returns rows written via *rows (host).
 * Every device pointer 16-B aligned (the literal rows are copied by one kernel with 16-B accesses, not by memcpy nodes). */
int mavlm_fuse_emit(mavlm_ctx* ctx, const void* x_pe, const int64_t* fine_idx, int32_t n_fine, const void* mem_prompt,
                    int32_t n_mem_prompt, const void* frame_prompt, int32_t n_frame_prompt, const void* newline,
                    int32_t with_frames, void* out, int64_t out_capacity_rows, int64_t* rows, void* stream);
/* the same for a row-batched context: x_pe = HOST array of B device pointers (each video's PE-added frames; fine_idx is
 * shared: the videos have the same length), out = [B, rows_per_video, D] - video b's block starts b * rows_per_video rows
 * into it; *rows = rows written per video. */
int mavlm_fuse_emit_batch(mavlm_ctx* ctx, const void* const* x_pe, const int64_t* fine_idx, int32_t n_fine,
                          const void* mem_prompt, int32_t n_mem_prompt, const void* frame_prompt, int32_t n_frame_prompt,
                          const void* newline, int32_t with_frames, void* out, int64_t rows_per_video, int64_t* rows,
                          void* stream);

/* --- operator-level entry points (used by the parity tests; same kernels the step uses) --------------- */
/* C = epi(A[M,K] . W[N,K]^T + bias); epilogue: 0 bias, 1 bias+ReLU, 2 bias+GELU(erf),
 * 3 bias+residual -> fp32 C, 4 bias -> fp32 C.   nn.Linear call sites: MemoryController.py:23,37-39,63-67; llava_arch.py:132-136 */
int mavlm_linear(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const void* res, int32_t ldr,
                 void* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t dtype, void* stream);
/* The attention forward + the frame scores of MemoryController.py:135-139 in the same pass: the S keys are S / patches
 * frames of `patches` consecutive keys (patches % 4 == 0, patches >= 64, <= 64 frames), frame_scores[f] (fp32) = mean over
 * the frame's keys of the column sums (over heads and queries) of the normalised probabilities.  Runs the schedule
 * mavlm_attention_ws runs for the shape (plain grid or levelled stream-K: O and lse2 are bit-identical to its results);
 * the small grids that mavlm_attention_ws splits over the keys run the plain grid here (= mavlm_attention).  ws:
 * mavlm_attention_frames_ws_floats(...) floats of scratch (0 = shape not supported).  What mavlm_step runs for the last
 * formation layer (mavlm_frame_scores_fused). */
int64_t mavlm_attention_frames_ws_floats(int32_t R, int32_t S, int32_t H, int32_t patches);
int mavlm_attention_frames(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                           int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, float scale, int32_t patches,
                           float* ws, int64_t ws_floats, float* frame_scores, int32_t dtype, void* stream);
/* The whole Residual block (MemoryController.py:20-29) in ONE kernel: out = LayerNorm(A . W^T + bias + res) * gamma + beta,
 * 16-bit out [M, ldo]; res 16-bit [M, ldr]; N % 256 == 0, N <= 1024 (wider rows - up to 4096 columns - only under the test
 * hook mavlm_set_fused_layernorm(2): correct but slower than the two-kernel form), and a grid that fills the chip
 * (mavlm_linear_ln_ws_bytes > 0; 0 = this shape takes mavlm_linear(epilogue 4) + mavlm_layernorm).  The N / 256 workgroups
 * of a 224/256-row block exchange their per-row (mean, centred sum of squares) through `ws` and merge them in a fixed order;
 * the fp32 dense output never goes through HBM.  pre_out: null, or [M, N] fp32 = A . W^T + bias (what the backward needs).
 * ws: mavlm_linear_ln_ws_bytes(M,N,K) bytes, 16-B aligned, zero-filled ONCE by the caller before its first use and then
 * left to the launches of ONE stream (it carries a launch counter: nothing is re-zeroed per call, hipGraph replays are
 * fine; two streams need two scratches - the Python operator keeps one per (device, stream)).
 * mavlm_step uses the same kernel wherever it applies (mavlm_set_fused_layernorm(0): the two-kernel form). */
int64_t mavlm_linear_ln_ws_bytes(int32_t M, int32_t N, int32_t K);
int mavlm_linear_ln(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const void* res, int32_t ldr,
                    const float* gamma, const float* beta, float eps, void* out, int32_t ldo, float* pre_out, int32_t M,
                    int32_t N, int32_t K, void* ws, int64_t ws_bytes, int32_t dtype, void* stream);
/* tuning hook: 1 (default) = the Residual blocks whose GEMM fills the chip run as one kernel (mavlm_linear_ln), 0 = GEMM with
 * fp32 epilogue + row LayerNorm kernel everywhere, 2 = as 1 and also for rows of up to 4096 columns (tests).  Same fp32 inputs to the normalisation; the row statistics are added in
 * a different order (part of the result, like the attention schedule). */
int mavlm_set_fused_layernorm(int32_t on);
/* byte offset, inside the workspace of a context with this config, of the 4 control words {arrivals, launch counter, timeout
 * flag, -} of the fused epilogue's exchange, or -1 when the config never takes the fused form.  The timeout flag is set when a
 * workgroup gave up waiting for a partner's statistics (bounded spin; the output of that launch is then wrong): 0 after any
 * correct run - the tests and bench.py read it. */
int64_t mavlm_workspace_ln_ctl_offset(const mavlm_config* cfg);
/* the same for an existing context (which snapshotted the hook mavlm_set_fused_layernorm when it was created) */
int64_t mavlm_ln_ctl_offset(const mavlm_ctx* ctx);
/* ctx[R,H*128] = softmax(Q K^T / sqrt(128)) V per head; lse2 [H,R] fp32 optional.  MemoryController.py:51-54 */
int mavlm_attention(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                    int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, float scale, int32_t dtype, void* stream);
/* As mavlm_linear, with the split-K path for GEMMs with few output tiles and a long contraction (<= 128 tiles of 128^2,
 * K >= 2048, contiguous C: e.g. the 4D -> D projections at 8 memory tokens): the contraction is split over up to 4
 * workgroup planes (fp32 partials in `ws`, mavlm_linear_ws_floats(...) floats; 0 = this shape does not split) and
 * bias + epilogue are applied once in the reduce.  Pure function of the shape; mavlm_step uses the same plan. */
int64_t mavlm_linear_ws_floats(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t ldc);
int mavlm_linear_ws(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const void* res, int32_t ldr,
                    void* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue, float* ws, int64_t ws_floats,
                    int32_t dtype, void* stream);
/* As mavlm_attention, with the two scheduled forms that need a workspace `ws` of mavlm_attention_ws_floats(R,S,H) floats
 * (0 = this shape runs the plain grid):
 *   - split-KV for grids too small to fill the chip (ceil(R/128)*H < 320 workgroups, e.g. the reference's default 8
 *     memory tokens): the keys are split over up to 8 workgroup planes;
 *   - the levelled stream-K schedule for grids with more units than resident workgroups and >= 64 key tiles (DESIGN.md
 *     section 4.2): whole units first, then the remainder in binary levels whose units are cut into 2^k key ranges.
 * Both write normalised fp32 partials + their log-sum-exp into `ws`, a second kernel merges them.  The plan is a pure
 * function of (R,S,H) - mavlm_step uses the same one, so both produce the same bits; mavlm_attention_plan reports it:
 * info[0] = waves per workgroup (4 | 8), info[1] = stream-K workgroups (0 = not scheduled), info[2] = levels,
 * info[3] = split-KV planes (1 = none). */
int64_t mavlm_attention_ws_floats(int32_t R, int32_t S, int32_t H);
int mavlm_attention_plan(int32_t R, int32_t S, int32_t H, int32_t info[4]);
/* Which unit (head-major index h * ceil(R / QB) + query block) the stream-K schedule runs at a position: level < 0: the unit
 * of whole round `a` on virtual workgroup `b`; level >= 0: the b-th unit of that level (cut into 2^k key ranges).  The
 * units are dealt XCD-major (every XCD walks a contiguous range of the head-major order: one head's K / V in its L2 at a
 * time); which units are cut is part of a result's rounding, so the oracle mirrors this map.  -1 = no plan / out of range. */
int mavlm_attention_plan_unit(int32_t R, int32_t S, int32_t H, int32_t level, int32_t a, int32_t b);
/* tuning / test hook: 1 (default) = XCD-affine unit order of the stream-K schedule, 0 = position p runs unit p (the order of
 * rounds 1-3).  Part of a result's rounding (which units are cut): the oracle mirrors it (STREAMK_AFFINE). */
int mavlm_set_attention_unit_order(int32_t affine);
int mavlm_attention_ws(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                       int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, float scale, float* ws,
                       int64_t ws_floats, int32_t dtype, void* stream);
/* same for wide heads: head h occupies columns [h*head_dim, (h+1)*head_dim); head_dim 448 (LLaVA-OneVision-7B: hidden
 * 3584 / 8 heads, llava_arch.py:117-122), 128, or 256 / 224 (the 4-head TransformerEncoder of the inactive
 * MemoryFuser variant, memory_module/MemoryFuser.py:12-19; forward only, no column-sum pass) */
int mavlm_attention_hd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                       int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, int32_t head_dim, float scale,
                       int32_t dtype, void* stream);
/* mavlm_attention_hd with the split-KV path for small grids (ceil(R/128)*H < 200 workgroups, >= 32 key tiles of 32: e.g.
 * 8 memory tokens at the OneVision-7B width: 104 units, 2 splits); ws = mavlm_attention_hd_ws_floats(...) floats (0 = no split).  Same
 * scheme and merge kernel as mavlm_attention_ws; mavlm_step uses the same plan. */
int64_t mavlm_attention_hd_ws_floats(int32_t R, int32_t S, int32_t H, int32_t head_dim);
/* schedule of the wide-head forward for this shape (H = heads of ALL videos of a row batch): info[0] = workgroups of the
 * levelled stream-K plan of the 32-query-wave kernel (head_dim 448; 0 = plain grid), info[1] = whole units per workgroup,
 * info[2] = cut levels, info[3] = key splits of the small-grid form (1 = none).  Part of the rounding plan (the oracle mirrors
 * it: oracle/memory_path.py streamk_plan_wide / split_plan_wide). */
int mavlm_attention_hd_plan_info(int32_t R, int32_t S, int32_t H, int32_t head_dim, int32_t* info);
int mavlm_attention_hd_ws(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                          int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, int32_t head_dim, float scale, float* ws,
                          int64_t ws_floats, int32_t dtype, void* stream);
int mavlm_attention_colsum_hd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const float* lse2, float* part,
                              int32_t R, int32_t S, int32_t H, int32_t head_dim, float scale, int32_t dtype, void* stream);
/* part[H,S] fp32 = column sums over queries of the normalised probabilities.  MemoryController.py:135
 * `part` is also the pass's scratch: it must hold mavlm_attention_colsum_floats(R,S,H) floats (>= H*S: the balanced
 * schedule writes one plane of [H,S] per piece of a (key block, head) unit and adds the planes in order - deterministic,
 * no atomics); the result is its first H*S floats.  part_floats = the floats `part` holds: MAVLM_E_ARG when too few. */
int64_t mavlm_attention_colsum_floats(int32_t R, int32_t S, int32_t H);
/* the schedule the pass runs (pure function of the shape): info[0] = workgroups, info[1] = planes it writes */
int mavlm_attention_colsum_plan(int32_t R, int32_t S, int32_t H, int32_t info[2]);
int mavlm_attention_colsum(const void* Q, int32_t ldq, const void* K, int32_t ldk, const float* lse2, float* part,
                           int64_t part_floats, int32_t R, int32_t S, int32_t H, float scale, int32_t dtype, void* stream);
/* out = LayerNorm(x fp32 [rows,D] + res) * gamma + beta -> 16-bit; res: 16-bit [rows, ldr] residual or null.
 * MemoryController.py:24,26-28 */
int mavlm_layernorm(const float* x, const void* res, int32_t ldr, const float* gamma, const float* beta, void* out,
                    int32_t rows, int32_t D, float eps, int32_t dtype, void* stream);
/* out[t,p,:] = x[src[t],p,:] + table[idx[t],:] (src/idx may be null).  position_encoding.py:64; llava_arch.py:524,554 */
int mavlm_row_add(const void* x, const int64_t* src, const void* table, const int64_t* idx, void* out, int32_t T,
                  int32_t P, int32_t D, int32_t dtype, void* stream);

/* step before the path: get_2dPool bilinear branch, llava_arch.py:277-297 -
 * x [F, side*side, D] -> out [F, ceil(side/stride)^2, D] (F.interpolate bilinear, align_corners=False), optionally fused
 * with the temporal PE add (pe_table/idx non-null; position_encoding.py:58,64).  One rounding per reference op. */
int mavlm_pool_bilinear(const void* x, void* out, const void* pe_table, const int64_t* idx, int32_t F, int32_t side,
                        int32_t stride, int32_t D, int32_t dtype, void* stream);

/* ---- backward pass of the path.  The reference differentiates the same modules with torch
 * autograd (training scripts: train.py:1708-1724); these are the device ops an autograd.Function calls.
 *
 * Attention backward, head_dim 128 (MemoryController.py:48-54): recomputes P = exp2(S*c - lse2) per tile.
 * O = forward output, dO its gradient, lse2 [H,R] from mavlm_attention, delta [H,R] fp32 scratch.
 * Any of dQ / dK / dV may be null (not computed: the frame features carry no gradient, llava_arch.py:302). */
int mavlm_attention_bwd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, const void* O,
                        int32_t ldo, const void* dO, int32_t lddo, const float* lse2, float* delta, void* dQ,
                        int32_t lddq, void* dK, int32_t lddk, void* dV, int32_t lddv, int32_t R, int32_t S, int32_t H,
                        float scale, int32_t dtype, void* stream);
/* The same for head_dim 448 (LLaVA-OneVision-7B: hidden 3584 / 8 heads, llava_arch.py:117-122; finetune_long.sh:33): all
 * operands [rows, H*448]; other head sizes -> MAVLM_E_SHAPE.  Flash style as well (rounds 2-4 composed this width's backward
 * from GEMMs over one head's materialised [R,S] scores). */
int mavlm_attention_bwd_hd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, const void* O,
                           int32_t ldo, const void* dO, int32_t lddo, const float* lse2, float* delta, void* dQ,
                           int32_t lddq, void* dK, int32_t lddk, void* dV, int32_t lddv, int32_t R, int32_t S, int32_t H,
                           int32_t head_dim, float scale, int32_t dtype, void* stream);
/* C[M,N] 16-bit = A[M,K] . W[N,K]^T, contraction split over `splits` workgroup planes; ws = splits*M*N fp32;
 * zero_bias = N fp32 zeros.  For dW = dY^T X, whose contraction runs over all rows of the activations. */
int mavlm_linear_splitk(const void* A, int32_t lda, const void* W, int32_t ldw, void* C, int32_t M, int32_t N, int32_t K,
                        int32_t splits, float* ws, const float* zero_bias, int32_t dtype, void* stream);
/* LayerNorm(x + res) backward: dz (16-bit [rows,D]) is the gradient of BOTH x and res; dgamma/dbeta fp32 [D];
 * ws = mavlm_layernorm_bwd_ws_floats(D) fp32 scratch.  MemoryController.py:24,26-28 */
int64_t mavlm_layernorm_bwd_ws_floats(int32_t D);
int mavlm_layernorm_bwd(const void* dy, const float* x, const void* res, int32_t ldr, const float* gamma, void* dz,
                        float* dgamma, float* dbeta, float* ws, int32_t rows, int32_t D, float eps, int32_t dtype,
                        void* stream);
/* out[cols, ldo] = in[rows, cols]^T for 16-bit elements; columns rows..roundup(rows,64)-1 of out are zero-filled. */
int mavlm_transpose(const void* in, int32_t ldi, int32_t rows, int32_t cols, void* out, int32_t ldo, void* stream);
/* out[r] fp32 = sum_c in[r, c] (bias gradient from dY^T). */
int mavlm_rowsum(const void* in, int32_t ld, int32_t rows, int32_t cols, float* out, int32_t dtype, void* stream);
/* elementwise over n (multiple of 8) 16-bit values: kind 0 out = gelu(x); 1 out = dy*gelu'(x) (x = pre-activation);
 * 2 out = x > 0 ? dy : 0 (x = ReLU output).  llava_arch.py:134; MemoryController.py:64 */
int mavlm_act(int32_t kind, const void* x, const void* dy, void* out, int64_t n, int32_t dtype, void* stream);

/* Wide heads (head_dim 448: LLaVA-OneVision-7B, the shape scripts/train/finetune_long.sh trains): the flash-style
 * backward does not fit the register file, so that case materialises ONE head's scores at a time and runs every
 * product through mavlm_linear / mavlm_linear_splitk; these are the element-wise steps in between.
 *   P[r,s]  = exp2(S[r,s]*scale*log2e - lse2[r]) for s < valid, else 0     (S fp32 [R, lds], P 16-bit [R, ldp])
 *   dS[r,s] = exp2(S[r,s]*scale*log2e - lse2[r]) * (dP[r,s] - delta[r]) * scale for s < valid, else 0
 *   out[h,r] = sum_d a[r, h*head_dim+d] * b[r, h*head_dim+d]                (delta = rowdot(dO, O)) */
int mavlm_attention_probs(const float* S, int32_t lds, const float* lse2, void* P, int32_t ldp, int32_t R, int32_t cols,
                          int32_t valid, float scale, int32_t dtype, void* stream);
int mavlm_attention_dscores(const float* S, int32_t lds, const float* dP, int32_t lddp, const float* lse2, const float* delta,
                            void* dS, int32_t ldds, int32_t R, int32_t cols, int32_t valid, float scale, int32_t dtype,
                            void* stream);
int mavlm_rowdot_heads(const void* a, int32_t lda, const void* b, int32_t ldb, float* out, int32_t R, int32_t H,
                       int32_t head_dim, int32_t dtype, void* stream);

/* ---- inactive variants of the reference (dead code there, forward only here) ----------------
 * v[f,:] = mean over the P patch rows of x[f,:,:] (16-bit and/or fp32 output).  bigru.py:50; segment.py:268 */
int mavlm_frame_mean(const void* x, void* out16, float* out32, int32_t F, int32_t P, int32_t D, int32_t dtype,
                     void* stream);
/* out[i] = cosine_similarity(v[i], v[i+1], eps) for i < n-1, v fp32 [n, D] (torch semantics:
 * x.y / (max(|x|,eps) max(|y|,eps))).  segment.py:33 */
int mavlm_adjacent_cosine(const float* v, float* out, int32_t n, int32_t D, float eps, void* stream);
/* recurrent half of nn.GRU over F steps, ndir directions (1 or 2; direction 1 runs backwards): xg fp32 [F, ndir*3H] =
 * W_ih x + b_ih (gate order r,z,n), whh 16-bit [ndir,3H,H], bhh fp32 [ndir,3H], out 16-bit [F, ndir*H]; H <= 512.
 * bigru.py:26-32,68 */
int mavlm_gru_sequence(const float* xg, const void* whh, const float* bhh, void* out, int32_t F, int32_t H, int32_t ndir,
                       int32_t dtype, void* stream);

/* tuning hook: 1 = attention backward computes dK and dV in ONE kernel (7 instead of 8 recompute products, 512-register
 * waves at one workgroup per CU); 0 = separate dK / dV kernels.  Same rounding points. */
int mavlm_set_attention_bwd_fused(int32_t on);
/* tuning hook: force the GEMM kernel (128 = 128^2 tile, 129 = 128x256 tile with two workgroups per CU, 256 = 256^2
 * non-persistent, 257 = 256^2 persistent;
 * 0 = automatic choice by grid size and epilogue).  Results are identical
 * up to fp32 summation order. */
int mavlm_set_gemm_tile(int32_t tile);
/* tuning hook: height of the 256-column GEMM workgroup tile - 256, 224 (= 7 x 32: divides M_tokens x 196 rows when
 * M_tokens % 8 == 0), or 0 = automatic (fewer row-rounds on 256 CUs).  Results are bit-identical. */
int mavlm_set_gemm_rows(int32_t rows);
/* tuning hook: tile order of the persistent GEMM for N >= 2048 (1 = 8 x 4 tile blocks per XCD, default; 0 = consecutive
 * tiles).  Speed only. */
int mavlm_set_gemm_order(int32_t order);
/* tuning hook: attention forward kernel - 2 = register-staged (attention.hip), 3 = software-pipelined LDS-DMA
 * (attention3.hip), 0 = default (3).  Same rounding points; results equal up to fp32 summation order. */
int mavlm_set_attention_impl(int32_t impl);
/* (all tuning hooks below that are "part of the result" are process-wide: a context created before a change keeps working -
 * its workspace covers either stream-K workgroup shape - and mavlm_step returns MAVLM_E_STATE if a plan would not fit it)
 * tuning / test hook: the head_dim-128 forward switches to its stream-K schedule (persistent workgroups over equal ranges of
 * the global key-tile sequence) when there are more units than workgroup slots AND at least this many 64-key
 * tiles per unit (default 64).  The schedule is part of the result (fp32 summation order): set it before sizing workspaces. */
int mavlm_set_attention_streamk_min_tiles(int32_t tiles);
/* tuning hook: waves per workgroup of the stream-K schedule - 8 (256 queries per unit, 256 workgroups: K / V staged once per
 * eight waves), 4 (128 queries, 512 workgroups), 0 = automatic (8 where its plan applies).  Part of the result, as above. */
int mavlm_set_attention_streamk_waves(int32_t waves);
/* workgroups of the column-sum pass's balanced schedule: 0 = automatic (512), or 64 .. 1024.  Changes the fp32 summation
 * order of the column sums (pieces per unit), nothing else. */
int mavlm_set_attention_colsum_wgs(int32_t wgs);
/* the head_dim-448 forward: 0 = automatic (2), 2 = 32-query waves on 32x32x16 MFMAs, software-pipelined tile loop (half the
 * LDS bytes per flop of the 16-query form), 1 = the 16-query waves of rounds 1-2.  Same rounding points; a 32-query wave takes
 * its deferred rescales as one group and adds the fp32 row sums in a different order (results agree to the tolerance of the
 * oracle tests, not bit for bit). */
int mavlm_set_attention_wide_groups(int32_t groups);
/* how mavlm_step obtains the frame scores of the last formation layer (patches % 4 == 0, <= 64 frames per chunk): 1 (default) =
 * fused into that layer's attention forward - heads of <= 128 columns: every query row carries the probability mass of the
 * current frame next to its row sum; heads of 448 (round 4): the forward writes one log-mass entry per query row and 32-key tile,
 * whatever its schedule, and a small kernel adds them per frame once the row's log-sum-exp is final (up to 1 GiB of entries in the
 * workspace, else mode 0) -; 0 = the separate column-sum pass over Q, K and lse2.  Heads of <= 128 columns on the small grids
 * that split their keys (one video with the checkpoint's 8 memory tokens), where the per-row frame masses are not available (one
 * writer per entry), take the tile-entry form as well (64-key tiles; round 4); 2 = diagnostics: the tile-entry form wherever it is
 * supported.  Same values up to fp32 summation order; the memory never depends on the mode. */
int mavlm_set_frame_score_mode(int32_t mode);
/* dense + bias + residual + LayerNorm where the GEMM splits its contraction (small grids, K >= 2048 - the 4D -> D projection at few
 * memory tokens): 1 (default) = the fp32 planes go straight into ONE reduce + LayerNorm kernel (round 4), 0 = reduction pass, fp32
 * dense output, LayerNorm kernel.  Same arithmetic in the same order: same bits. */
int mavlm_set_splitk_layernorm(int32_t on);
/* K ranges of the fp32 dense output (Residual block) on small grids with 1024 <= K < 2048 (round 4; default 2, 1 = no split).  Part of
 * the rounding plan of those shapes; set it BEFORE mavlm_create (the workspace is sized for the plan). */
int mavlm_set_gemm_short_splits(int32_t splits);
/* 1 if mavlm_step (single video) takes the fused form for a last-layer attention of R memory rows over S = F * patches keys
 * with heads of <= 128 columns in its per-(row, frame) form - patches % 4 == 0, patches >= 64, <= 64 frames, and not one of the
 * small grids that split their keys (mavlm_attention_ws_floats), which take the tile-entry form (mavlm_set_frame_score_mode).  The fused launch runs the SAME schedule as the
 * plain forward of that shape (mavlm_attention_ws): the context and the memory do not depend on whether scores are asked for. */
int mavlm_frame_scores_fused(int32_t R, int32_t S, int32_t H, int32_t patches);

/* --- measurement hooks (bench.py only; new - the reference has no profiling) ------------- */
/* When enabled, every kernel launch is bracketed by HIP events on its own stream.  Kinds: 0 GEMM, 1 attention
 * forward, 2 attention column-sum, 3 LayerNorm (forward and backward), 4 row-add, 5 misc, 6 attention backward,
 * 7 split-K GEMM, 8 transpose, 9 attention merge (split-KV / stream-K partials), 10 attention forward carrying the frame
 * masses (last formation layer), 11 GEMM with the fused residual + LayerNorm epilogue (mavlm_linear_ln: the Residual blocks;
 * its time contains what kinds 0 + 3 spend on the two-kernel form).  Not re-entrant, not graph-capturable. */
int mavlm_prof_enable(int32_t on);
/* host arrays of length nkinds >= 12: total milliseconds, launches, algorithmic flops, algorithmic bytes per kind */
int mavlm_prof_read(double* ms, int64_t* launches, double* flops, double* bytes, int32_t nkinds);

#ifdef __cplusplus
}
#endif
#endif /* MAVLM_H_ */
