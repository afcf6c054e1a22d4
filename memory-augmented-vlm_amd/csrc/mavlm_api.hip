// C-ABI layer (include/mavlm.h): host-side handle, workspace carving and the per-chunk launch sequence.
// No allocation, no synchronisation: every entry point only enqueues kernels on the caller's stream.
#include "../../include/mavlm.h"

#include <math.h>

#include <new>

#include "mavlm_kernels.h"

struct mavlm_ctx {
  mavlm_config cfg;
  mavlm_weights w;
  mavlm_buffers b;
  bool has_w = false, has_b = false;
  int steps = 0;
  // workspace carve (byte offsets)
  size_t o_kv, o_q, o_ctx, o_a, o_h, o_pre, o_mA, o_mB, o_lse, o_part, o_split, o_gsplit, gsplit_floats, total;
  size_t o_fscr = 0, o_fout = 0;   // frame-score variant of the last layer's forward (0 = not available for this config)
  bool ftiles_ok = false;          // ... and the carve covers its tile-entry form (attention3.hip FR = 2)
  size_t split_floats = 0;         // floats carved at o_split (attention partials: split-KV / stream-K)
  size_t o_lnx = 0, lnx_bytes = 0; // scratch of the fused dense + residual + LayerNorm GEMM epilogue (0 = not used for this config)
  // K/V projections of chunks ahead of their step (mavlm_project_chunk: the next step; mavlm_project_chunk_ahead: the one after it,
  // on another stream): two chunk K/V buffers (o_kv, o_kv2; single videos), a slot per buffer
  struct pre_slot {
    const void* seg = nullptr;     // the chunk whose K/V the buffer holds / will hold (0 = none)
    int F = 0, step = -1;          // ... for the step with this index
    void* stream = nullptr;        // the stream the projection was enqueued on
    bool pending = false;          // a projection into this buffer may still be in flight on `stream` (ev_pre not waited for yet)
  } pre[2];
  size_t o_kv2 = 0;                // second chunk K/V buffer (0 = none: row batches)
  hipEvent_t ev_pre[2] = {nullptr, nullptr};    // recorded behind a projection into buffer b
  hipEvent_t ev_done[2] = {nullptr, nullptr};   // recorded behind the last step that read buffer b
  bool done_rec[2] = {false, false};
  int pre_hits = 0;                // steps that reused a projection (mavlm_prefetch_hits: tests)
  ~mavlm_ctx() {
    for (int i = 0; i < 2; ++i) {
      if (ev_pre[i]) (void)hipEventDestroy(ev_pre[i]);
      if (ev_done[i]) (void)hipEventDestroy(ev_done[i]);
    }
  }
  int fuse_mems = 1;   // cached memories the Memory-Fuser MLP takes per GEMM launch (mavlm_fuse_emit)
  int fused_ln = 1;    // snapshot of the process-wide hook at mavlm_create (0 = two-kernel form, 1 / 2 = fused where supported)
  int ln_wide = 0;     // ... and of its "rows of up to 4096 columns" test mode
  bool lnx_clean = false;   // the exchange scratch of this workspace has been zero-filled since the last mavlm_bind_buffers
};

int g_mavlm_fused_ln = 1;   // process-wide hook (mavlm_set_fused_layernorm); contexts snapshot it at mavlm_create
int g_mavlm_splitk_ln = 1;  // dense + LayerNorm over a split contraction: planes -> reduce + LayerNorm in one kernel (mavlm_set_splitk_layernorm)

namespace {

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

bool cfg_ok(const mavlm_config* c) {
  if (!c) return false;
  if (c->hidden <= 0 || c->heads <= 0 || c->patches <= 0 || c->mem_tokens <= 0 || c->depth <= 0 ||
      c->depth > MAVLM_MAX_DEPTH || c->inter <= 0 || c->cache_cap <= 0 || c->max_chunk_frames <= 0)
    return false;
  if (c->dtype != 0 && c->dtype != 1) return false;
  if (c->batch < 0 || c->batch > MAVLM_MAX_BATCH) return false;
  if (c->q_tokens < 0 || c->q_token0 < 0 || c->q_token0 + c->q_tokens > c->mem_tokens || (c->q_tokens == 0 && c->q_token0 != 0))
    return false;
  if (c->q_tokens > 0 && c->batch > 1) return false;           // a row shard of ONE video, or a row batch of whole videos
  if (c->fused_ln != MAVLM_LN_AUTO && c->fused_ln != MAVLM_LN_NEVER) return false;
  return true;
}
inline int nbatch(const mavlm_config& c) { return c.batch > 1 ? c.batch : 1; }
// row shard (mavlm_config.q_tokens): memory tokens this context computes per step, and the first of them
inline int q_tokens(const mavlm_config& c) { return c.q_tokens > 0 ? c.q_tokens : c.mem_tokens; }
inline int q_row0(const mavlm_config& c) { return c.q_tokens > 0 ? c.q_token0 * c.patches : 0; }

// shapes the gfx950 kernels implement (DESIGN.md "Supported shapes"): head_dim <= 128 (heads are zero-padded to
// 128 columns in the Q/K/V/ctx buffers and in the packed weights), D and I multiples of 128
bool shape_ok(const mavlm_config* c) {
  if (c->hidden % 128 != 0 || c->hidden % c->heads != 0 || c->inter % 128 != 0) return false;
  const int hd = c->hidden / c->heads;
  return hd <= 128 || hd == 448;      // 448: wide-head kernels (attention_hd.hip), no padding
}
inline bool wide_heads(const mavlm_config& c) { return c.hidden / c.heads > 128; }
inline int padded_width(const mavlm_config& c) { return wide_heads(c) ? c.hidden : c.heads * 128; }
inline float attn_scale(const mavlm_config& c) { return 1.0f / sqrtf((float)(c.hidden / c.heads)); }

// Memory-Fuser batching: the fuser MLP is row-independent, and the FIFO ring is one contiguous [cap, R, D] buffer, so
// cached memories that sit in consecutive slots go through ONE pair of GEMMs.  At the reference's 8 memory tokens
// (R = 1568) a launch per memory fills 7 of 256 CUs' worth of 256-row tiles; up to FUSE_ROWS rows per launch fill the chip.
constexpr int FUSE_ROWS = 32768;
inline int fuse_mems_per_launch(const mavlm_config& c) {
  const int R = c.mem_tokens * c.patches * nbatch(c);       // rows of one FIFO slot (all videos of the row batch)
  int k = FUSE_ROWS / R;
  if (k < 1) k = 1;
  return k > c.cache_cap ? c.cache_cap : k;
}

void carve(mavlm_ctx* x) {
  const mavlm_config& c = x->cfg;
  const size_t B = (size_t)nbatch(c);
  const size_t Rf = (size_t)c.mem_tokens * c.patches;          // memory rows of ONE video (row buffers: sized for all of them)
  const size_t R1 = (size_t)q_tokens(c) * c.patches;           // query rows of one video this context computes (row shard)
  const size_t R = Rf * B;                                     // rows of every row-wise operator (all videos stacked)
  const size_t S = (size_t)c.max_chunk_frames * c.patches, D = c.hidden, I = c.inter, L = c.depth, H = c.heads,
               Dp = (size_t)padded_width(c);
  size_t o = 0;
  x->o_kv = o;   o += al(B * S * 2 * L * Dp * 2);
  x->o_kv2 = 0;
  if (B == 1) { x->o_kv2 = o; o += al(S * 2 * L * Dp * 2); }      // landing zone of mavlm_project_chunk_ahead
  x->o_q = o;    o += al(R * Dp * 2);
  x->o_ctx = o;  o += al(R * Dp * 2);
  x->o_a = o;    o += al(R * D * 2);
  x->fuse_mems = fuse_mems_per_launch(c);
  x->o_h = o;    o += al((size_t)x->fuse_mems * R * I * 2);     // MLP hidden of a step (R rows) / of a fuser batch
  x->o_pre = o;  o += al(R * D * 4);
  x->o_mA = o;   o += al(R * D * 2);
  x->o_mB = o;   o += al(R * D * 2);
  x->o_lse = o;  o += al(B * H * R1 * 4);
  {   // column-sum planes (attention3.hip: balanced schedule): the plane count depends on the chunk's key count
    size_t fl = H * S;
    if (!wide_heads(c))
      for (int f = 1; f <= c.max_chunk_frames; ++f) {
        const size_t v = mavlm_colsum_part_floats((int)R1, f * c.patches, (int)H);
        if (v > fl) fl = v;
      }
    x->o_part = o; o += al(fl * 4);
  }
  // attention partials.  Single videos: split-KV of the small grids (mavlm_attention_splits), worst case over the key
  // count, or the levelled stream-K schedule (more units than workgroup slots; independent of the key count).  The
  // stream-K size is the maximum over both workgroup shapes (4 / 8 waves), so that mavlm_set_attention_streamk_waves
  // may change after the context exists; attn_block checks the carved size anyway.
  x->o_split = o;
  {
    const size_t items = ((R1 + 127) / 128) * H;
    size_t fl = 0;
    if (B == 1) {
      size_t cap;
      if (!wide_heads(c)) cap = items < 320 ? (512 / items > 8 ? 8 : 512 / items) : 0;      // mavlm_attention_splits
      else cap = (size_t)mavlm_attention_hd_splits((int)R1, 1 << 20, (int)H, nullptr);       // (its maximum over the key count)
      fl = cap >= 2 ? cap * (R1 * Dp + H * R1) : 0;
    }
    const int s_long = 1 << 20;          // (the stream-K schedules only depend on "enough key tiles")
    if (!wide_heads(c)) {
      const size_t sk = mavlm_attention_split_ws_floats_max((int)R1, s_long, (int)(H * B));
      if (sk > fl) fl = sk;
    } else if (mavlm_attention_hd_streamk((int)R1, s_long, (int)(H * B), c.hidden / c.heads, nullptr) > 0) {
      const size_t sk = mavlm_attention_hd_split_ws_floats((int)R1, s_long, (int)(H * B), c.hidden / c.heads);
      if (sk > fl) fl = sk;
    }
    x->split_floats = fl;
    if (fl) o += al(fl * 4);
  }
  // split-K planes of the GEMMs with few output tiles and a long contraction (mavlm_gemm_splits): the I -> D
  // projections (MLP down, fuser second layer) at small R
  x->o_gsplit = o;
  {
    x->gsplit_floats = 0;
    const int rows_[2] = {(int)R, (int)(R1 * B)};              // every row of the memory / the rows of a row shard
    for (int r : rows_) {
      const int sk = (int)S, d = (int)D, dp = (int)Dp, in = (int)I, l2 = (int)(2 * L * Dp);
      const size_t need[] = {mavlm_gemm_split_ws_floats(r, dp, d, MAVLM_EPI_BIAS, dp),      // q projection
                             mavlm_gemm_split_ws_floats(r, d, dp, MAVLM_EPI_F32, d),        // attention out dense
                             mavlm_gemm_split_ws_floats(r, in, d, MAVLM_EPI_RELU, in),      // MLP up / fuser first
                             mavlm_gemm_split_ws_floats(r, d, in, MAVLM_EPI_F32, d),        // MLP down / fuser second
                             mavlm_gemm_split_ws_floats(sk, l2, d, MAVLM_EPI_BIAS, l2),     // chunk K/V
                             mavlm_gemm_split_ws_floats(r, 2 * dp, d, MAVLM_EPI_BIAS, 2 * dp)};   // evolution K/V
      for (size_t n : need) x->gsplit_floats = n > x->gsplit_floats ? n : x->gsplit_floats;
    }
  }
  o += al(x->gsplit_floats * 4);
  // scratch of the frame-score variant of the last formation layer's forward (attention3.hip): (a, m) per (head, memory
  // row, frame) and the partial frame sums per (unit, 32-query group)
  x->o_fscr = x->o_fout = 0;
  x->ftiles_ok = false;
  {
    const int fc = c.max_chunk_frames < 64 ? c.max_chunk_frames : 64;      // chunks of more frames take the column-sum pass
    const bool sup1 = !wide_heads(c) && mavlm_attention_frames_supported((int)R1, fc * c.patches, (int)(H * B), c.patches);
    const bool sup2 = !wide_heads(c) && mavlm_attention_frame_tiles_supported((int)R1, fc * c.patches, (int)(H * B), c.patches) &&
                      mavlm_attention_frame_tiles_scr_floats((int)R1, fc * c.patches, (int)(H * B)) * 4 <= ((size_t)1 << 30);
    if (sup1 || sup2) {
      // the per-(row, frame) form and / or the per-(row, 64-key tile) form (the small grids that split their keys): the larger of both
      size_t scr = sup1 ? mavlm_attention_frames_scr_floats((int)R1, fc * c.patches, (int)(H * B), c.patches) : 0;
      size_t out = sup1 ? mavlm_attention_frames_out_floats((int)R1, fc * c.patches, (int)(H * B), c.patches) : 0;
      if (sup2) {
        const size_t s2 = mavlm_attention_frame_tiles_scr_floats((int)R1, fc * c.patches, (int)(H * B));
        const size_t o2 = mavlm_attention_frame_tiles_out_floats((int)R1, c.patches, (int)(H * B), c.patches) * 64;
        scr = s2 > scr ? s2 : scr;
        out = o2 > out ? o2 : out;
      }
      x->ftiles_ok = sup2;
      x->o_fscr = o; o += al(scr * 4);
      x->o_fout = o; o += al(out * 4);
    } else if (wide_heads(c) && mavlm_attention_hd_frames_supported((int)R1, fc * c.patches, (int)(H * B), c.hidden / c.heads, c.patches) &&
               mavlm_attention_hd_frames_scr_floats((int)R1, fc * c.patches, (int)(H * B)) * 4 <= ((size_t)1 << 30)) {
      // head_dim 448 (attention_hd.hip): one 8-byte entry per (head, memory row, 32-key tile) - up to 1 GiB, else the column-sum pass
      x->o_fscr = o; o += al(mavlm_attention_hd_frames_scr_floats((int)R1, fc * c.patches, (int)(H * B)) * 4);
      x->o_fout = o; o += al(mavlm_attention_hd_frames_out_floats((int)R1, c.patches, (int)(H * B), c.patches) * 64 * 4);
    }
  }
  // scratch of the fused dense + residual + LayerNorm epilogue (gemm256.hip EPI_LN): {epoch, value} granules of the row
  // blocks + control words.  MUST be zero when the workspace is bound (mavlm_buffers.workspace).
  x->o_lnx = 0;
  x->lnx_bytes = 0;
  // the process-wide hook is SNAPSHOTTED here (carve runs once per context, at mavlm_create): a live context keeps its form
  x->fused_ln = c.fused_ln == MAVLM_LN_NEVER ? 0 : g_mavlm_fused_ln;
  x->ln_wide = g_mavlm_gemm_ln_wide;
  if (x->fused_ln) {
    const int rows_[2] = {(int)R, (int)(R1 * B)};
    for (int r : rows_)
      if (mavlm_gemm_ln_supported(r, (int)D, (int)Dp, x->ln_wide) || mavlm_gemm_ln_supported(r, (int)D, (int)I, x->ln_wide)) {
        const size_t b = mavlm_gemm_ln_ws_bytes(r, (int)D);
        if (b > x->lnx_bytes) x->lnx_bytes = b;
      }
    if (x->lnx_bytes) { x->o_lnx = o; o += al(x->lnx_bytes); }
  }
  x->total = o;
}

inline char* ws(mavlm_ctx* x, size_t off) { return (char*)x->b.workspace + off; }

#define MAVLM_TRY(expr)                       \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

hipError_t gemm(int dtype, hipStream_t s, const void* A, int lda, const void* W, int ldw, const float* bias, void* C, int ldc,
                int M, int N, int K, int epi, const void* res = nullptr, int ldr = 0, float* split_ws = nullptr,
                size_t split_floats = 0) {
  mavlm_gemm_args g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.res = res; g.ldr = ldr; g.C = C; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.epilogue = epi;
  // split-K only when the caller's workspace covers this shape's plan (the plan is part of the result)
  if (split_ws != nullptr && mavlm_gemm_split_ws_floats(M, N, K, epi, ldc) <= split_floats) g.splitk_ws = split_ws;
  return mavlm_launch_gemm(g, dtype, s);
}

// GEMM of the fused step: the split-K workspace of the context rides along, so that every launch takes the same plan
// as the stand-alone operator (mavlm_linear_ws) would for its shape
inline hipError_t gemm_x(mavlm_ctx* x, hipStream_t s, const void* A, int lda, const void* W, int ldw, const float* bias, void* C,
                         int ldc, int M, int N, int K, int epi) {
  return gemm(x->cfg.dtype, s, A, lda, W, ldw, bias, C, ldc, M, N, K, epi, nullptr, 0,
              x->gsplit_floats ? (float*)ws(x, x->o_gsplit) : nullptr, x->gsplit_floats);
}

// Residual block (MemoryController.py:26-29): out = LayerNorm(A . W^T + bias + res) * gamma + beta.  One kernel where the
// 256-column-tile GEMM fills the chip (EPI_LN: the fp32 dense output never goes through HBM), else the GEMM with its
// fp32 epilogue + the row LayerNorm kernel.  Same fp32 values into the normalisation either way; the two forms add the
// row statistics in different orders (a pure function of the shape which one runs: mavlm_linear_ln_fused).
int dense_ln(mavlm_ctx* x, hipStream_t s, const void* A, int lda, const void* W, int ldw, const float* bias, const void* res,
             const float* gamma, const float* beta, void* out, int rows, int N, int K) {
  const mavlm_config& c = x->cfg;
  if (x->fused_ln && x->lnx_bytes && mavlm_gemm_ln_supported(rows, N, K, x->ln_wide) && mavlm_gemm_ln_ws_bytes(rows, N) <= x->lnx_bytes) {
    if (!x->lnx_clean) {
      // the library owns the "zero before the first launch" invariant of the exchange scratch: one asynchronous memset on
      // the stream of the first step after mavlm_bind_buffers.  Not inside a capture: a replayed memset would reset the
      // launch counter under the granules of the previous replay.
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return MAVLM_E_STATE;
      MAVLM_TRY(hipMemsetAsync(ws(x, x->o_lnx), 0, x->lnx_bytes, s));
      x->lnx_clean = true;
    }
    mavlm_gemm_args g;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.res = res; g.ldr = N; g.C = out; g.ldc = N;
    g.M = rows; g.N = N; g.K = K; g.epilogue = MAVLM_EPI_LN;
    g.ln.gamma = gamma; g.ln.beta = beta; g.ln.eps = c.eps; g.ln.wide = x->ln_wide;
    // (control words FIRST: their place must not depend on the shape - the launch counter is what keeps epochs unique)
    g.ln.ctl = (unsigned*)ws(x, x->o_lnx);
    g.ln.gran = (unsigned long long*)(ws(x, x->o_lnx) + 64);
    MAVLM_TRY(mavlm_launch_gemm(g, c.dtype, s));
    return 0;
  }
  // small grids whose contraction is split (the 4D -> D projection at few memory tokens): the fp32 planes go straight into a
  // reduce + bias + residual + LayerNorm kernel - no reduction pass, no fp32 round trip of the dense output (same bits)
  if (g_mavlm_splitk_ln && x->gsplit_floats && (N & 7) == 0 && N <= 4096 &&
      mavlm_gemm_split_ws_floats(rows, N, K, MAVLM_EPI_F32, N) != 0 &&
      mavlm_gemm_split_ws_floats(rows, N, K, MAVLM_EPI_F32, N) <= x->gsplit_floats) {
    mavlm_gemm_args g;
    int planes = 0;
    g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.res = nullptr; g.ldr = 0; g.C = ws(x, x->o_pre); g.ldc = N;
    g.M = rows; g.N = N; g.K = K; g.epilogue = MAVLM_EPI_F32;
    g.splitk_ws = (float*)ws(x, x->o_gsplit); g.planes_only = 1; g.planes_out = &planes;
    MAVLM_TRY(mavlm_launch_gemm(g, c.dtype, s));
    if (planes > 0) {
      MAVLM_TRY(mavlm_launch_layernorm_planes((const float*)ws(x, x->o_gsplit), planes, bias, res, N, gamma, beta, out, rows, N,
                                              c.eps, c.dtype, s));
      return 0;
    }
    // (the launch did not split after all - a tuning hook: it wrote the dense output to `pre`)
    MAVLM_TRY(mavlm_launch_layernorm((const float*)ws(x, x->o_pre), res, N, gamma, beta, out, rows, N, c.eps, c.dtype, s));
    return 0;
  }
  MAVLM_TRY(gemm_x(x, s, A, lda, W, ldw, bias, ws(x, x->o_pre), N, rows, N, K, MAVLM_EPI_F32));
  MAVLM_TRY(mavlm_launch_layernorm((const float*)ws(x, x->o_pre), res, N, gamma, beta, out, rows, N, c.eps, c.dtype, s));
  return 0;
}

// One `Attention` block given projected K/V:  out = LN(dense(attn(q_proj(xq), K, V)) + xq).  All videos of the row batch
// at once: xq / out are [B R, D]; video b's S keys start kv_bs elements after video b-1's.
// frame_rows (out): rows of the partial frame sums per video when `frames` (argument of the finish kernel)
int attn_block(mavlm_ctx* x, hipStream_t s, const mavlm_attn_weights& aw, const void* xq, const void* K, int ldk,
               const void* V, int ldv, long long kv_bs, int S, void* out, float* lse2, int frames = 0,
               int* frame_rows = nullptr) {
  const mavlm_config& c = x->cfg;
  const int B = nbatch(c), R1 = q_tokens(c) * c.patches, R = R1 * B, D = c.hidden, H = c.heads, dt = c.dtype,
            Dp = padded_width(c);
  MAVLM_TRY(gemm_x(x, s, xq, D, aw.wq, D, aw.bq, ws(x, x->o_q), Dp, R, Dp, D, MAVLM_EPI_BIAS));
  mavlm_attn_args a;
  a.Q = ws(x, x->o_q); a.ldq = Dp; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = ws(x, x->o_ctx); a.ldo = Dp;
  a.lse2 = lse2; a.R = R1; a.S = S; a.H = H * B; a.nb = B; a.kv_bstride = kv_bs; a.scale = attn_scale(c);
  if (wide_heads(c)) {
    // wide heads (448: the OneVision-7B width): the row batch is the grid's z dimension, or - when its units exceed the 256
    // workgroups of one-per-CU - the levelled stream-K plan over the (video, head) pairs; a single video with a small grid takes
    // the kernel's split-KV form.  The schedule is part of the result: the plan is taken only when the carved workspace covers
    // it under the CURRENT tuning hooks.
    const int hd = c.hidden / c.heads;
    const bool sk = mavlm_attention_hd_streamk(R1, S, H * B, hd, nullptr) > 0;
    if (sk) a.split_ws = mavlm_attention_hd_split_ws_floats(R1, S, H * B, hd) <= x->split_floats ? (float*)ws(x, x->o_split) : nullptr;
    else a.split_ws = (B == 1 && x->split_floats) ? (float*)ws(x, x->o_split) : nullptr;
    if (frames) {                             // forward + per-(row, tile) probability masses in one pass (no column-sum pass)
      a.frame_scr = (float*)ws(x, x->o_fscr);
      a.frame_out = (float*)ws(x, x->o_fout);
      a.frame_keys = c.patches;
      if (frame_rows) *frame_rows = mavlm_attention_hd_frames_rows_per_video(R1, H);
      MAVLM_TRY(mavlm_launch_attention_hd_frames(a, hd, dt, s));
    } else {
      MAVLM_TRY(mavlm_launch_attention_hd(a, hd, dt, s));
    }
  } else {
    // the schedule is part of the result: take it only when the carved workspace covers this shape's plan under the
    // CURRENT tuning hooks (they may have changed since mavlm_create) - never write past the carve
    // (a row batch never takes the split-KV form of the small grids: attention3.hip)
    const size_t need = (B == 1 || mavlm_attention_streamk_wgs(R1, S, H * B) > 0) ? mavlm_attention_split_ws_floats(R1, S, H * B) : 0;
    if (need > x->split_floats) return MAVLM_E_STATE;
    a.split_ws = x->split_floats ? (float*)ws(x, x->o_split) : nullptr;
    if (frames == 2) {                      // forward + per-(row, tile) masses: any schedule, incl. the split-KV small grids
      a.frame_scr = (float*)ws(x, x->o_fscr);
      a.frame_out = (float*)ws(x, x->o_fout);
      a.frame_keys = c.patches;
      if (frame_rows) *frame_rows = H * ((R1 + 63) / 64);
      MAVLM_TRY(mavlm_launch_attention3_frame_tiles(a, dt, s));
    } else if (frames) {                    // forward + per-frame probability mass in one pass (no column-sum pass)
      a.frame_scr = (float*)ws(x, x->o_fscr);
      a.frame_out = (float*)ws(x, x->o_fout);
      a.frame_keys = c.patches;
      if (frame_rows) *frame_rows = mavlm_attention_frames_rows_per_video(a);
      MAVLM_TRY(mavlm_launch_attention3_frames(a, dt, s));
    } else {
      MAVLM_TRY(mavlm_launch_attention(a, dt, s));
    }
  }
  // Residual: dense + bias + residual in fp32, LayerNorm (MemoryController.py:26-29)
  return dense_ln(x, s, ws(x, x->o_ctx), Dp, aw.wo, Dp, aw.bo, xq, aw.ln_g, aw.ln_b, out, R, D, Dp);
}

// does the step take the fused frame scores for a last-layer attention over S keys?  (single videos: not the small grids
// that split their keys - mavlm_frame_scores_fused; a row batch never splits)
// 0 = no (column-sum pass), 1 = per-(row, frame) masses (attention3.hip FR = 1) or the wide heads' tile entries, 2 = the tile-entry
// form of the 128-wide kernel (FR = 2: the small grids that split their keys; everywhere under mavlm_set_frame_score_mode(2))
int step_frames_fused(const mavlm_ctx* x, int S) {
  const mavlm_config& c = x->cfg;
  const int B = nbatch(c), R1 = q_tokens(c) * c.patches;
  if (x->o_fscr == 0 || g_mavlm_frame_score_mode == 0) return 0;
  const bool fits = S <= (c.max_chunk_frames < 64 ? c.max_chunk_frames : 64) * c.patches;
  if (wide_heads(c))       // (any schedule: a (row, tile) entry has one writer under all of them)
    return fits && mavlm_attention_hd_frames_supported(R1, S, c.heads * B, c.hidden / c.heads, c.patches) ? 1 : 0;
  if (g_mavlm_attn_impl == 2) return 0;
  const bool tiles = x->ftiles_ok && fits && mavlm_attention_frame_tiles_supported(R1, S, c.heads * B, c.patches);
  if (g_mavlm_frame_score_mode == 2) return tiles ? 2 : 0;
  if (B == 1) return mavlm_frame_scores_fused(R1, S, c.heads, c.patches) != 0 ? 1 : (tiles ? 2 : 0);
  return mavlm_attention_frames_supported(R1, S, c.heads * B, c.patches) ? 1 : 0;
}

int step_impl(mavlm_ctx* x, const void* const* segs, int32_t F, void* frame_scores, int32_t scores_f32, hipStream_t s) {
  const mavlm_config& c = x->cfg;
  // R1: memory rows of one video (keys of the evolution, ring strides); Rq1: the rows of them this context computes (row
  // shard: q_tokens memory tokens from row r0 on; else all); R: rows of the row-wise operators
  const int B = nbatch(c), R1 = c.mem_tokens * c.patches, Rq1 = q_tokens(c) * c.patches, r0 = q_row0(c), R = Rq1 * B,
            D = c.hidden, I = c.inter, L = c.depth, H = c.heads, dt = c.dtype;
  const int Dp = padded_width(c);
  const int S = F * c.patches;
  const size_t slot_bytes = (size_t)R1 * B * D * 2;           // one FIFO slot: the newest memory of every video, [B, R1, D]
  const int cap = c.cache_cap;
  const int n = x->steps < cap ? x->steps : cap;
  // evolution K/V ring: [B][cap][R1][2 Dp] - the keys of ONE video are contiguous over its slots
  const long long evo_bs = (long long)cap * R1 * 2 * Dp;

  const void* cur = x->w.mem0;
  if (x->steps > 0) {
    // ---- memory evolution (MemoryController.py:89-97): q = newest memory, kv = every cached memory.
    const int newest = (x->steps - 1) % cap;
    const char* mem_new = (const char*)x->b.mem_ring + (size_t)newest * slot_bytes;     // ALL rows (row shard: gathered)
    char* kv_new = (char*)x->b.evo_kv_ring + (size_t)newest * R1 * 2 * Dp * 2;
    // K/V of a cached memory are row-independent -> project each memory once, when it becomes the newest.  Row batch: ONE
    // GEMM over the stacked rows, block b of R1 rows lands in video b's ring
    {
      mavlm_gemm_args g;
      g.A = mem_new; g.lda = D; g.W = x->w.w_kv_evo; g.ldw = D; g.bias = x->w.b_kv_evo; g.res = nullptr; g.ldr = 0;
      g.C = kv_new; g.ldc = 2 * Dp; g.M = R1 * B; g.N = 2 * Dp; g.K = D; g.epilogue = MAVLM_EPI_BIAS;
      if (B > 1) { g.c_rpb = R1; g.c_nb = B; g.c_bstride = evo_bs; }
      else if (x->gsplit_floats && mavlm_gemm_split_ws_floats(R1, 2 * Dp, D, MAVLM_EPI_BIAS, 2 * Dp) <= x->gsplit_floats)
        g.splitk_ws = (float*)ws(x, x->o_gsplit);
      MAVLM_TRY(mavlm_launch_gemm(g, dt, s));
    }
    const char* kv = (const char*)x->b.evo_kv_ring;
    int rc = attn_block(x, s, x->w.evo, mem_new + (size_t)r0 * D * 2, kv, 2 * Dp, kv + (size_t)Dp * 2, 2 * Dp, evo_bs, n * R1,
                        ws(x, x->o_mA), nullptr);
    if (rc) return rc;
    cur = ws(x, x->o_mA);
  }

  // ---- memory formation (MemoryController.py:132-133): K/V of the chunk for all L layers in one GEMM per video (the shape
  // does not depend on the memory rows: nothing to gain from stacking, and the videos' frames stay where they are)
  const int ldkv = 2 * L * Dp;
  // Which chunk K/V buffer this step reads: the one a projection made ahead for exactly this step and chunk sits in (the step
  // then waits for it, nothing else); otherwise buffer 0 - or buffer 1 while a projection for a LATER step owns buffer 0.  A
  // projection made for this step but another chunk is discarded (its buffer is waited for before anything overwrites it).
  int rb = 0;
  bool pre = false;
  if (B == 1) {
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cst);
    const bool capturing = cst != hipStreamCaptureStatusNone;
    for (int b = 0; b < 2; ++b)
      if (x->pre[b].seg && x->pre[b].step == x->steps) {
        if (!pre && !capturing && x->pre[b].seg == segs[0] && x->pre[b].F == F) { pre = true; rb = b; }
        x->pre[b].seg = nullptr;
      }
    if (!pre) rb = (x->pre[0].seg && x->pre[0].step > x->steps && x->o_kv2) ? 1 : 0;
    if (x->pre[rb].pending && !capturing) {
      if (x->pre[rb].stream != (void*)s || !pre) MAVLM_TRY(hipStreamWaitEvent(s, x->ev_pre[rb], 0));
      x->pre[rb].pending = false;
    }
  }
  x->pre_hits += pre ? 1 : 0;       // a projection made ahead for exactly this chunk
  char* kvs = ws(x, rb ? x->o_kv2 : x->o_kv);
  for (int b = 0; b < B && !pre; ++b)
    MAVLM_TRY(gemm_x(x, s, segs[b], D, x->w.w_kv_seg, D, x->w.b_kv_seg, kvs + (size_t)b * S * ldkv * 2, ldkv, S, ldkv, D,
                     MAVLM_EPI_BIAS));
  for (int l = 0; l < L; ++l) {
    const bool last = l == L - 1;
    const bool want_scores = last && frame_scores != nullptr;
    const char* Kl = kvs + (size_t)(2 * l) * Dp * 2;
    const char* Vl = Kl + (size_t)Dp * 2;
    // frame scores: fused into this layer's forward (default), or the column-sum pass over its Q / K / lse2
    const int fused_scores = want_scores ? step_frames_fused(x, S) : 0;
    float* lse = want_scores ? (float*)ws(x, x->o_lse) : nullptr;      // (the fused form does not need it; kept for inspection)
    int frows = 0;
    int rc = attn_block(x, s, x->w.layer_attn[l], cur, Kl, ldkv, Vl, ldkv, (long long)S * ldkv, S, ws(x, x->o_a), lse,
                        fused_scores, &frows);
    if (rc) return rc;
    if (fused_scores) {
      MAVLM_TRY(mavlm_launch_frame_finish((const float*)ws(x, x->o_fout), frows, B, F, c.patches, frame_scores, scores_f32, dt, s));
    } else if (want_scores) {
      for (int b = 0; b < B; ++b) {          // (one pass per video: the column-sum kernels know nothing of the row batch)
        mavlm_colsum_args ca;
        ca.Q = ws(x, x->o_q) + (size_t)b * Rq1 * Dp * 2; ca.ldq = Dp; ca.K = Kl + (size_t)b * S * ldkv * 2; ca.ldk = ldkv;
        ca.lse2 = lse + (size_t)b * H * Rq1; ca.part = (float*)ws(x, x->o_part);
        ca.R = Rq1; ca.S = S; ca.H = H; ca.scale = attn_scale(c);
        int planes = 1;
        if (wide_heads(c)) {
          MAVLM_TRY(mavlm_launch_colsum_hd(ca, c.hidden / c.heads, dt, s));
        } else {
          ca.keep_planes = 1;                   // frame_scores_kernel adds the planes (same order as the reduce kernel)
          planes = mavlm_colsum_planes(Rq1, S, H);
          MAVLM_TRY(mavlm_launch_colsum(ca, dt, s));
        }
        void* fs = scores_f32 ? (void*)((float*)frame_scores + (size_t)b * F) : (void*)((uint16_t*)frame_scores + (size_t)b * F);
        MAVLM_TRY(mavlm_launch_frame_scores(ca.part, planes, H, S, F, c.patches, fs, scores_f32, dt, s));
      }
    }
    // MLP: Linear(D,I)+ReLU -> Residual(I->D)  (MemoryController.py:63-67,71)
    MAVLM_TRY(gemm_x(x, s, ws(x, x->o_a), D, x->w.w_up[l], D, x->w.b_up[l], ws(x, x->o_h), I, R, I, D, MAVLM_EPI_RELU));
    // (row shard: this context's rows of the slot; the host all-gathers the other ranks' rows into it before the next step)
    void* dst = last ? (void*)((char*)x->b.mem_ring + (size_t)(x->steps % cap) * slot_bytes + (size_t)r0 * D * 2)
                     : (void*)ws(x, (l & 1) ? x->o_mA : x->o_mB);
    {
      int rc2 = dense_ln(x, s, ws(x, x->o_h), I, x->w.w_down[l], I, x->w.b_down[l], ws(x, x->o_a), x->w.ln2_g[l], x->w.ln2_b[l],
                         dst, R, D, I);
      if (rc2) return rc2;
    }
    cur = dst;
  }
  if (B == 1 && x->o_kv2) {          // the chunk K/V buffer this step read is free once the stream gets here
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cst);
    if (cst == hipStreamCaptureStatusNone && x->ev_done[rb]) {
      MAVLM_TRY(hipEventRecord(x->ev_done[rb], s));
      x->done_rec[rb] = true;
    }
  }
  x->steps += 1;   // append; the slot written above evicts the oldest entry once the ring is full (:152-154)
  return 0;
}

}  // namespace

extern "C" {

int mavlm_abi_version(void) { return MAVLM_ABI_VERSION; }

int mavlm_set_attention_impl(int32_t impl) {
  if (impl != 0 && impl != 2 && impl != 3) return MAVLM_E_ARG;
  g_mavlm_attn_impl = impl;
  return 0;
}

extern int g_mavlm_attn_bwd_fused;
int mavlm_set_attention_bwd_fused(int32_t on) {
  g_mavlm_attn_bwd_fused = on ? 1 : 0;
  return 0;
}

int mavlm_set_gemm_tile(int32_t tile) {
  if (tile != 0 && tile != 128 && tile != 129 && tile != 256 && tile != 257) return MAVLM_E_ARG;
  g_mavlm_gemm_tile = tile;
  return 0;
}

extern int g_mavlm_attn_sk_waves;      // (g_mavlm_attn_sk_min_tiles: mavlm_kernels.h)
extern int g_mavlm_attn_unit_order;
int mavlm_set_attention_unit_order(int32_t affine) {
  if (affine != 0 && affine != 1) return MAVLM_E_ARG;
  g_mavlm_attn_unit_order = affine;
  return 0;
}
int mavlm_set_attention_streamk_min_tiles(int32_t tiles) {
  if (tiles < 1) return MAVLM_E_ARG;
  g_mavlm_attn_sk_min_tiles = tiles;
  return 0;
}
int mavlm_set_attention_streamk_waves(int32_t waves) {
  if (waves != 0 && waves != 4 && waves != 8) return MAVLM_E_ARG;
  g_mavlm_attn_sk_waves = waves;
  return 0;
}

int mavlm_set_attention_colsum_wgs(int32_t wgs) {
  if (wgs != 0 && (wgs < 64 || wgs > 1024)) return MAVLM_E_ARG;
  g_mavlm_colsum_wgs = wgs;
  return 0;
}

int mavlm_frame_scores_fused(int32_t R, int32_t S, int32_t H, int32_t patches) {
  // not for the small grids that split their keys (few query blocks x heads, e.g. 8 memory tokens): the frame masses ride on
  // the never-split grid, which would leave most of the chip idle there - those shapes keep the column-sum pass
  return (g_mavlm_frame_score_mode == 1 && g_mavlm_attn_impl != 2 && R > 0 && H > 0 &&
          mavlm_attention_frames_supported(R, S, H, patches) && mavlm_attention_splits(R, S, H, nullptr) <= 1) ? 1 : 0;
}

int mavlm_set_attention_wide_groups(int32_t groups) {
  if (groups < 0 || groups > 2) return MAVLM_E_ARG;
  g_mavlm_attn_hd_qg = groups;
  return 0;
}

int mavlm_set_frame_score_mode(int32_t mode) {
  if (mode < 0 || mode > 2) return MAVLM_E_ARG;       // (2: diagnostics - the tile-entry form wherever it is supported)
  g_mavlm_frame_score_mode = mode;
  return 0;
}

extern int g_mavlm_gemm_f32_short_splits;
int mavlm_set_gemm_short_splits(int32_t splits) {
  if (splits < 1 || splits > 8) return MAVLM_E_ARG;
  g_mavlm_gemm_f32_short_splits = splits;
  return 0;
}

int mavlm_set_splitk_layernorm(int32_t on) {
  if (on != 0 && on != 1) return MAVLM_E_ARG;
  g_mavlm_splitk_ln = on;
  return 0;
}

int mavlm_set_fused_layernorm(int32_t on) {
  if (on < 0 || on > 2) return MAVLM_E_ARG;
  g_mavlm_fused_ln = on ? 1 : 0;
  g_mavlm_gemm_ln_wide = on == 2 ? 1 : 0;
  return 0;
}

int64_t mavlm_linear_ln_ws_bytes(int32_t M, int32_t N, int32_t K) {
  return (g_mavlm_fused_ln && mavlm_gemm_ln_supported(M, N, K, -1)) ? (int64_t)mavlm_gemm_ln_ws_bytes(M, N) : 0;
}

int mavlm_linear_ln(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const void* res, int32_t ldr,
                    const float* gamma, const float* beta, float eps, void* out, int32_t ldo, float* pre_out, int32_t M,
                    int32_t N, int32_t K, void* ws_, int64_t ws_bytes, int32_t dtype, void* stream) {
  if (!A || !W || !bias || !res || !gamma || !beta || !out || M < 0) return MAVLM_E_ARG;
  const int64_t need = mavlm_linear_ln_ws_bytes(M, N, K);
  if (need == 0) return MAVLM_E_SHAPE;
  if (!ws_ || ws_bytes < need || ((uintptr_t)ws_ & 15)) return MAVLM_E_ARG;
  mavlm_gemm_args g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = bias; g.res = res; g.ldr = ldr; g.C = out; g.ldc = ldo;
  g.M = M; g.N = N; g.K = K; g.epilogue = MAVLM_EPI_LN;
  g.ln.gamma = gamma; g.ln.beta = beta; g.ln.eps = eps; g.ln.pre_out = pre_out;
  g.ln.ctl = (unsigned*)ws_;                               // control words first: a fixed place for every shape
  g.ln.gran = (unsigned long long*)((char*)ws_ + 64);
  hipError_t e = mavlm_launch_gemm(g, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

extern int g_mavlm_gemm_order;
int mavlm_set_gemm_order(int32_t order) {
  if (order != 0 && order != 1) return MAVLM_E_ARG;
  g_mavlm_gemm_order = order;
  return 0;
}

int mavlm_set_gemm_rows(int32_t rows) {
  if (rows != 0 && rows != 224 && rows != 256) return MAVLM_E_ARG;
  g_mavlm_gemm_rows = rows;
  return 0;
}

size_t mavlm_workspace_bytes(const mavlm_config* cfg) {
  if (!cfg_ok(cfg)) return 0;
  mavlm_ctx t;
  t.cfg = *cfg;
  carve(&t);
  return t.total;
}

int64_t mavlm_workspace_ln_ctl_offset(const mavlm_config* cfg) {
  if (!cfg_ok(cfg)) return -1;
  mavlm_ctx t;
  t.cfg = *cfg;
  carve(&t);
  return t.lnx_bytes ? (int64_t)t.o_lnx : -1;
}

int64_t mavlm_ln_ctl_offset(const mavlm_ctx* x) { return x ? (x->lnx_bytes ? (int64_t)x->o_lnx : -1) : MAVLM_E_ARG; }

int mavlm_workspace_layout(const mavlm_config* cfg, size_t* offsets, int32_t n) {
  if (!cfg_ok(cfg) || !offsets || n < 10) return MAVLM_E_ARG;
  mavlm_ctx t;
  t.cfg = *cfg;
  carve(&t);
  const size_t o[10] = {t.o_kv, t.o_q, t.o_ctx, t.o_a, t.o_h, t.o_pre, t.o_mA, t.o_mB, t.o_lse, t.o_part};
  for (int i = 0; i < 10; ++i) offsets[i] = o[i];
  return 0;
}

int mavlm_create(const mavlm_config* cfg, mavlm_ctx** out) {
  if (!out || !cfg_ok(cfg)) return MAVLM_E_ARG;
  if (!shape_ok(cfg)) return MAVLM_E_SHAPE;
  mavlm_ctx* x = new (std::nothrow) mavlm_ctx();
  if (!x) return MAVLM_E_ARG;
  x->cfg = *cfg;
  carve(x);
  *out = x;
  return 0;
}

void mavlm_destroy(mavlm_ctx* ctx) { delete ctx; }

int mavlm_bind_weights(mavlm_ctx* x, const mavlm_weights* w) {
  if (!x || !w) return MAVLM_E_ARG;
  if (!w->mem0 || !w->w_kv_seg || !w->b_kv_seg || !w->w_kv_evo || !w->b_kv_evo) return MAVLM_E_ARG;
  // the fuser / token-type weights live outside TransformerProjector (llava_arch.py:132-150): optional here,
  // required by mavlm_fuse_emit
  const mavlm_attn_weights* e = &w->evo;
  if (!e->wq || !e->bq || !e->wo || !e->bo || !e->ln_g || !e->ln_b) return MAVLM_E_ARG;
  for (int l = 0; l < x->cfg.depth; ++l) {
    const mavlm_attn_weights* a = &w->layer_attn[l];
    if (!a->wq || !a->bq || !a->wo || !a->bo || !a->ln_g || !a->ln_b || !w->w_up[l] || !w->b_up[l] || !w->w_down[l] ||
        !w->b_down[l] || !w->ln2_g[l] || !w->ln2_b[l])
      return MAVLM_E_ARG;
  }
  x->w = *w;
  x->pre[0].seg = x->pre[1].seg = nullptr;      // K/V projected with the previous weights are not this context's any more
  x->has_w = true;
  return 0;
}

int mavlm_bind_buffers(mavlm_ctx* x, const mavlm_buffers* b) {
  if (!x || !b || !b->mem_ring || !b->evo_kv_ring || !b->workspace) return MAVLM_E_ARG;
  if (b->workspace_bytes < x->total || ((uintptr_t)b->workspace & 255)) return MAVLM_E_ARG;
  x->b = *b;
  x->has_b = true;
  x->lnx_clean = false;            // a new workspace: its exchange scratch is zero-filled at the first step
  x->pre[0] = x->pre[1] = mavlm_ctx::pre_slot();     // (a prefetched projection lived in the old workspace)
  x->done_rec[0] = x->done_rec[1] = false;
  return 0;
}

int mavlm_ln_status_async(mavlm_ctx* x, void* host16, int32_t clear, void* stream) {
  if (!x || !host16) return MAVLM_E_ARG;
  if (!x->has_b) return MAVLM_E_STATE;
  if (!x->lnx_bytes || !x->lnx_clean) return 1;       // never fused / nothing launched yet: nothing to report
  hipStream_t s = (hipStream_t)stream;
  MAVLM_TRY(hipMemcpyAsync(host16, ws(x, x->o_lnx), 16, hipMemcpyDeviceToHost, s));
  if (clear) MAVLM_TRY(hipMemsetAsync(ws(x, x->o_lnx) + 8, 0, 4, s));
  return 0;
}

int mavlm_reset(mavlm_ctx* x) {
  if (!x) return MAVLM_E_ARG;
  x->steps = 0;
  x->pre[0].seg = x->pre[1].seg = nullptr;      // (a projection still in flight keeps its `pending` mark: the next writer waits)
  return 0;
}

int mavlm_prefetch_hits(const mavlm_ctx* x) { return x ? x->pre_hits : MAVLM_E_ARG; }
int mavlm_cache_len(const mavlm_ctx* x) { return x ? (x->steps < x->cfg.cache_cap ? x->steps : x->cfg.cache_cap) : MAVLM_E_ARG; }
int mavlm_newest_slot(const mavlm_ctx* x) { return x ? (x->steps ? (x->steps - 1) % x->cfg.cache_cap : -1) : MAVLM_E_ARG; }
int mavlm_steps(const mavlm_ctx* x) { return x ? x->steps : MAVLM_E_ARG; }

int mavlm_pe_add(const void* xin, const int64_t* idx, const void* table, void* out, int32_t T, int32_t P, int32_t D,
                 int32_t dtype, void* stream) {
  if (!xin || !idx || !table || !out || T < 0) return MAVLM_E_ARG;
  return (int)mavlm_launch_row_add(xin, nullptr, table, idx, out, T, P, D, dtype, (hipStream_t)stream);
}

int mavlm_step(mavlm_ctx* x, const void* seg, int32_t F, void* frame_scores, int32_t scores_f32, void* stream) {
  if (!x || !seg) return MAVLM_E_ARG;
  if (!x->has_w || !x->has_b) return MAVLM_E_STATE;
  if (nbatch(x->cfg) != 1) return MAVLM_E_STATE;             // a row-batched context steps through mavlm_step_batch
  if (F <= 0 || F > x->cfg.max_chunk_frames) return MAVLM_E_SHAPE;
  return step_impl(x, &seg, F, frame_scores, scores_f32, (hipStream_t)stream);
}

int mavlm_step_batch(mavlm_ctx* x, const void* const* segs, int32_t F, void* frame_scores, int32_t scores_f32, void* stream) {
  if (!x || !segs) return MAVLM_E_ARG;
  if (!x->has_w || !x->has_b) return MAVLM_E_STATE;
  if (F <= 0 || F > x->cfg.max_chunk_frames) return MAVLM_E_SHAPE;
  for (int b = 0; b < nbatch(x->cfg); ++b)
    if (!segs[b]) return MAVLM_E_ARG;
  return step_impl(x, segs, F, frame_scores, scores_f32, (hipStream_t)stream);
}

int mavlm_batch(const mavlm_ctx* x) { return x ? nbatch(x->cfg) : MAVLM_E_ARG; }

namespace {
// K/V projection of chunk `seg` for the step with index `for_step` (>= x->steps), enqueued on `stream`
int project_for(mavlm_ctx* x, const void* seg, int32_t F, void* stream, int for_step) {
  if (!x || !seg) return MAVLM_E_ARG;
  if (!x->has_w || !x->has_b || nbatch(x->cfg) != 1) return MAVLM_E_STATE;
  const mavlm_config& c = x->cfg;
  if (F <= 0 || F > c.max_chunk_frames) return MAVLM_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cst);
  if (cst != hipStreamCaptureStatusNone) return MAVLM_E_STATE;         // (events across streams: not inside a graph capture)
  for (int i = 0; i < 2; ++i) {
    if (!x->ev_pre[i]) MAVLM_TRY(hipEventCreateWithFlags(&x->ev_pre[i], hipEventDisableTiming));
    if (!x->ev_done[i]) MAVLM_TRY(hipEventCreateWithFlags(&x->ev_done[i], hipEventDisableTiming));
  }
  // The buffer.  For the upcoming step (index n = x->steps): buffer 0, or 1 while buffer 0 holds a projection for a later step.
  // For the step after it: buffer 1 - the upcoming step reads (or projects inline into) buffer 0 - unless buffer 1 holds the
  // projection the upcoming step is going to read, then 0.
  int tb = 0;
  if (x->o_kv2) {
    const int n = x->steps;
    if (for_step == n) tb = (x->pre[0].seg && x->pre[0].step > n) ? 1 : 0;
    else tb = (x->pre[1].seg && x->pre[1].step == n) ? 0 : 1;
  }
  if (x->pre[tb].pending && x->pre[tb].stream != stream) MAVLM_TRY(hipStreamWaitEvent(s, x->ev_pre[tb], 0));   // an older projection
  if (x->done_rec[tb]) MAVLM_TRY(hipStreamWaitEvent(s, x->ev_done[tb], 0));     // the last step that read this buffer has finished
  const int D = c.hidden, Dp = padded_width(c), ldkv = 2 * c.depth * Dp, S = F * c.patches;
  MAVLM_TRY(gemm_x(x, s, seg, D, x->w.w_kv_seg, D, x->w.b_kv_seg, ws(x, tb ? x->o_kv2 : x->o_kv), ldkv, S, ldkv, D, MAVLM_EPI_BIAS));
  MAVLM_TRY(hipEventRecord(x->ev_pre[tb], s));
  x->pre[tb].seg = seg;
  x->pre[tb].F = F;
  x->pre[tb].step = for_step;
  x->pre[tb].stream = stream;
  x->pre[tb].pending = true;
  return 0;
}
}  // namespace

int mavlm_project_chunk(mavlm_ctx* x, const void* seg, int32_t F, void* stream) { return project_for(x, seg, F, stream, x ? x->steps : 0); }
int mavlm_project_chunk_ahead(mavlm_ctx* x, const void* seg, int32_t F, void* stream) {
  if (x && !x->o_kv2) return MAVLM_E_STATE;
  return project_for(x, seg, F, stream, x ? x->steps + 1 : 0);
}

namespace {
int fuse_emit_impl(mavlm_ctx* x, const void* const* x_pe, const int64_t* fine_idx, int32_t n_fine, const void* mem_prompt,
                   int32_t n_mem_prompt, const void* frame_prompt, int32_t n_frame_prompt, const void* newline,
                   int32_t with_frames, void* out, int64_t cap_rows, int64_t* rows, hipStream_t s) {
  if (!x->w.w_f1 || !x->w.b_f1 || !x->w.w_f2 || !x->w.b_f2_type0 || (with_frames && !x->w.type1)) return MAVLM_E_STATE;
  if (n_mem_prompt < 0 || n_frame_prompt < 0 || n_fine < 0 || (n_mem_prompt && !mem_prompt) ||
      (with_frames && ((n_frame_prompt && !frame_prompt) || (n_fine && (!x_pe || !fine_idx)))))
    return MAVLM_E_ARG;
  const mavlm_config& c = x->cfg;
  const int B = nbatch(c), R1 = c.mem_tokens * c.patches, R = R1 * B, D = c.hidden, I = c.inter, dt = c.dtype, cap = c.cache_cap;
  const int n = x->steps < cap ? x->steps : cap;
  if (n == 0) return MAVLM_E_STATE;
  int64_t need = (int64_t)n_mem_prompt + (int64_t)n * R1 + 1;
  if (with_frames) need += (int64_t)n_frame_prompt + (int64_t)n_fine * c.patches + 1;
  if (cap_rows < need) return MAVLM_E_ARG;
  if (with_frames && n_fine)
    for (int b = 0; b < B; ++b)
      if (!x_pe[b]) return MAVLM_E_ARG;
  const size_t rowb = (size_t)D * 2;
  // (video b's token block starts b * cap_rows rows into `out`)
  char* o = (char*)out;
  int64_t row = 0;
  {
    // the literal rows (memory prompt, newline, frame prompt, newline) of all B blocks: one launch, rows known up front
    const int64_t r_nl1 = (int64_t)n_mem_prompt + (int64_t)n * R1, r_fp = r_nl1 + 1;
    const int64_t r_nl2 = r_fp + n_frame_prompt + (int64_t)n_fine * c.patches;
    const void* src[4] = {mem_prompt, newline, frame_prompt, newline};
    const int cnt[4] = {n_mem_prompt, 1, with_frames ? n_frame_prompt : 0, with_frames ? 1 : 0};
    const long long dst[4] = {0, r_nl1, r_fp, r_nl2};
    if ((((uintptr_t)mem_prompt | (uintptr_t)newline | (uintptr_t)frame_prompt | (uintptr_t)out) & 15)) return MAVLM_E_ARG;
    MAVLM_TRY(mavlm_launch_copy_rows(src, cnt, dst, 4, out, (long long)cap_rows * D, B, D, s));
  }
  row += n_mem_prompt;
  const int oldest = x->steps <= cap ? 0 : x->steps % cap;
  // torch.cat(memory_cache) order = oldest first (llava_arch.py:545) = ring slots oldest..cap-1, then 0..oldest-1: at most
  // two contiguous slot ranges, each fused in batches of up to fuse_mems memories per GEMM pair (llava_arch.py:546).  A
  // slot holds the memories of all B videos ([B, R1, D]): the second GEMM writes block b of a slot into video b's tokens.
  for (int done = 0; done < n;) {
    const int slot = (oldest + done) % cap;
    int run = n - done < cap - slot ? n - done : cap - slot;
    if (run > x->fuse_mems) run = x->fuse_mems;
    const int rows_ = run * R;
    const char* mem = (const char*)x->b.mem_ring + (size_t)slot * R * rowb;
    MAVLM_TRY(gemm_x(x, s, mem, D, x->w.w_f1, D, x->w.b_f1, ws(x, x->o_h), I, rows_, I, D, MAVLM_EPI_GELU));
    if (B == 1) {
      MAVLM_TRY(gemm_x(x, s, ws(x, x->o_h), I, x->w.w_f2, I, x->w.b_f2_type0, o + (size_t)row * rowb, D, rows_, D, I,
                       MAVLM_EPI_BIAS));
    } else {
      mavlm_gemm_args g;
      g.A = ws(x, x->o_h); g.lda = I; g.W = x->w.w_f2; g.ldw = I; g.bias = x->w.b_f2_type0; g.res = nullptr; g.ldr = 0;
      g.C = o + (size_t)row * rowb; g.ldc = D; g.M = rows_; g.N = D; g.K = I; g.epilogue = MAVLM_EPI_BIAS;
      g.c_rpb = R1; g.c_nb = B; g.c_bstride = (long long)cap_rows * D;
      MAVLM_TRY(mavlm_launch_gemm(g, dt, s));
    }
    row += (int64_t)run * R1;
    done += run;
  }
  row += 1;
  if (with_frames) {
    row += n_frame_prompt;
    if (n_fine)
      MAVLM_TRY(mavlm_launch_row_add_batch(x_pe, fine_idx, x->w.type1, o + (size_t)row * rowb, (long long)cap_rows * D, B, n_fine,
                                           c.patches, D, dt, s));
    row += (int64_t)n_fine * c.patches;
    row += 1;
  }
  *rows = row;
  return 0;
}
}  // namespace

int mavlm_fuse_emit(mavlm_ctx* x, const void* x_pe, const int64_t* fine_idx, int32_t n_fine, const void* mem_prompt,
                    int32_t n_mem_prompt, const void* frame_prompt, int32_t n_frame_prompt, const void* newline,
                    int32_t with_frames, void* out, int64_t cap_rows, int64_t* rows, void* stream) {
  if (!x || !out || !newline || !rows) return MAVLM_E_ARG;
  if (!x->has_w || !x->has_b) return MAVLM_E_STATE;
  if (nbatch(x->cfg) != 1) return MAVLM_E_STATE;
  return fuse_emit_impl(x, x_pe ? &x_pe : nullptr, fine_idx, n_fine, mem_prompt, n_mem_prompt, frame_prompt, n_frame_prompt, newline,
                        with_frames, out, cap_rows, rows, (hipStream_t)stream);
}

int mavlm_fuse_emit_batch(mavlm_ctx* x, const void* const* x_pe, const int64_t* fine_idx, int32_t n_fine, const void* mem_prompt,
                          int32_t n_mem_prompt, const void* frame_prompt, int32_t n_frame_prompt, const void* newline,
                          int32_t with_frames, void* out, int64_t rows_per_video, int64_t* rows, void* stream) {
  if (!x || !out || !newline || !rows) return MAVLM_E_ARG;
  if (!x->has_w || !x->has_b) return MAVLM_E_STATE;
  return fuse_emit_impl(x, x_pe, fine_idx, n_fine, mem_prompt, n_mem_prompt, frame_prompt, n_frame_prompt, newline, with_frames,
                        out, rows_per_video, rows, (hipStream_t)stream);
}

int mavlm_linear(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const void* res, int32_t ldr,
                 void* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t dtype, void* stream) {
  if (!A || !W || !bias || !C || M < 0) return MAVLM_E_ARG;
  if (N % 128 || K % 64 || N <= 0 || K <= 0) return MAVLM_E_SHAPE;
  hipError_t e = gemm(dtype, (hipStream_t)stream, A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, res, ldr);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int64_t mavlm_linear_ws_floats(int32_t M, int32_t N, int32_t K, int32_t epilogue, int32_t ldc) {
  return (int64_t)mavlm_gemm_split_ws_floats(M, N, K, epilogue, ldc);
}

int mavlm_linear_ws(const void* A, int32_t lda, const void* W, int32_t ldw, const float* bias, const void* res, int32_t ldr,
                    void* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t epilogue, float* ws_, int64_t ws_floats,
                    int32_t dtype, void* stream) {
  if (!A || !W || !bias || !C || M < 0) return MAVLM_E_ARG;
  if (N % 128 || K % 64 || N <= 0 || K <= 0) return MAVLM_E_SHAPE;
  const int64_t need = (int64_t)mavlm_gemm_split_ws_floats(M, N, K, epilogue, ldc);
  if (need > 0 && (!ws_ || ws_floats < need)) return MAVLM_E_ARG;      // the plan is part of the result
  hipError_t e = gemm(dtype, (hipStream_t)stream, A, lda, W, ldw, bias, C, ldc, M, N, K, epilogue, res, ldr,
                      need > 0 ? ws_ : nullptr, need > 0 ? (size_t)ws_floats : 0);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_attention(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                    int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, float scale, int32_t dtype, void* stream) {
  mavlm_attn_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.lse2 = lse2;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  hipError_t e = mavlm_launch_attention(a, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int64_t mavlm_attention_frames_ws_floats(int32_t R, int32_t S, int32_t H, int32_t patches) {
  if (R <= 0 || H <= 0 || !mavlm_attention_frames_supported(R, S, H, patches)) return 0;
  // [entries | partial frame sums | partials of the stream-K schedule, when this shape runs it]
  const size_t sk = mavlm_attention_streamk_wgs(R, S, H) > 0 ? mavlm_attention_split_ws_floats(R, S, H) : 0;
  return (int64_t)(mavlm_attention_frames_scr_floats(R, S, H, patches) + mavlm_attention_frames_out_floats(R, S, H, patches) + sk);
}

int mavlm_attention_frames(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                           int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, float scale, int32_t patches,
                           float* ws_, int64_t ws_floats, float* frame_scores, int32_t dtype, void* stream) {
  const int64_t need = mavlm_attention_frames_ws_floats(R, S, H, patches);
  if (!Q || !K || !V || !O || !frame_scores || need == 0 || !ws_ || ws_floats < need) return MAVLM_E_ARG;
  mavlm_attn_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.lse2 = lse2;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  a.frame_scr = ws_;
  a.frame_out = ws_ + mavlm_attention_frames_scr_floats(R, S, H, patches);
  a.frame_keys = patches;
  if (mavlm_attention_streamk_wgs(R, S, H) > 0)
    a.split_ws = a.frame_out + mavlm_attention_frames_out_floats(R, S, H, patches);
  const int frows = mavlm_attention_frames_rows_per_video(a);
  hipError_t e = mavlm_launch_attention3_frames(a, dtype, (hipStream_t)stream);
  if (e == hipSuccess)
    e = mavlm_launch_frame_finish(a.frame_out, frows, 1, S / patches, patches, frame_scores, 1, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int64_t mavlm_attention_ws_floats(int32_t R, int32_t S, int32_t H) {
  return (R > 0 && S > 0 && H > 0) ? (int64_t)mavlm_attention_split_ws_floats(R, S, H) : 0;
}

int mavlm_attention_plan(int32_t R, int32_t S, int32_t H, int32_t info[4]) {
  if (!info || R <= 0 || S <= 0 || H <= 0) return MAVLM_E_ARG;
  int v[4];
  mavlm_attention_plan_info(R, S, H, v);
  for (int i = 0; i < 4; ++i) info[i] = v[i];
  return 0;
}

int mavlm_attention_plan_unit(int32_t R, int32_t S, int32_t H, int32_t level, int32_t a, int32_t b) {
  if (R <= 0 || S <= 0 || H <= 0) return MAVLM_E_ARG;
  return mavlm_attention_plan_unit_(R, S, H, level, a, b);
}

int mavlm_attention_ws(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                       int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, float scale, float* ws,
                       int64_t ws_floats, int32_t dtype, void* stream) {
  mavlm_attn_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.lse2 = lse2;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  if (R > 0 && S > 0 && H > 0) {
    const int64_t need = (int64_t)mavlm_attention_split_ws_floats(R, S, H);
    if (need > 0 && (!ws || ws_floats < need)) return MAVLM_E_ARG;     // the plan is part of the result: no silent change
    a.split_ws = need > 0 ? ws : nullptr;
  }
  hipError_t e = mavlm_launch_attention(a, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_attention_hd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                       int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, int32_t head_dim, float scale,
                       int32_t dtype, void* stream) {
  if (!Q || !K || !V || !O || R <= 0 || S <= 0 || H <= 0 || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) ||
      ldq < H * head_dim || ldk < head_dim || ldv < head_dim)
    return MAVLM_E_ARG;
  if (head_dim != 448 && head_dim != 128 && head_dim != 256 && head_dim != 224) return MAVLM_E_SHAPE;
  mavlm_attn_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.lse2 = lse2;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  return (int)mavlm_launch_attention_hd(a, head_dim, dtype, (hipStream_t)stream);
}

int mavlm_attention_hd_plan_info(int32_t R, int32_t S, int32_t H, int32_t head_dim, int32_t* info) {
  if (R <= 0 || S <= 0 || H <= 0 || !info) return MAVLM_E_ARG;
  int v[3] = {0, 0, 0};
  mavlm_attention_hd_streamk(R, S, H, head_dim, v);
  info[0] = v[0]; info[1] = v[1]; info[2] = v[2];
  info[3] = v[0] > 0 ? 1 : mavlm_attention_hd_splits(R, S, H, nullptr);
  return 0;
}

int64_t mavlm_attention_hd_ws_floats(int32_t R, int32_t S, int32_t H, int32_t head_dim) {
  return (R > 0 && S > 0 && H > 0 && head_dim > 0) ? (int64_t)mavlm_attention_hd_split_ws_floats(R, S, H, head_dim) : 0;
}

int mavlm_attention_hd_ws(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, void* O,
                          int32_t ldo, float* lse2, int32_t R, int32_t S, int32_t H, int32_t head_dim, float scale, float* ws_,
                          int64_t ws_floats, int32_t dtype, void* stream) {
  if (!Q || !K || !V || !O || R <= 0 || S <= 0 || H <= 0 || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3) ||
      ldq < H * head_dim || ldk < head_dim || ldv < head_dim)
    return MAVLM_E_ARG;
  if (head_dim != 448 && head_dim != 128 && head_dim != 256 && head_dim != 224) return MAVLM_E_SHAPE;
  const int64_t need = (int64_t)mavlm_attention_hd_split_ws_floats(R, S, H, head_dim);
  if (need > 0 && (!ws_ || ws_floats < need)) return MAVLM_E_ARG;        // the plan is part of the result
  mavlm_attn_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.lse2 = lse2;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  a.split_ws = need > 0 ? ws_ : nullptr;
  return (int)mavlm_launch_attention_hd(a, head_dim, dtype, (hipStream_t)stream);
}

int mavlm_attention_colsum_hd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const float* lse2, float* part,
                              int32_t R, int32_t S, int32_t H, int32_t head_dim, float scale, int32_t dtype, void* stream) {
  if (!Q || !K || !lse2 || !part || R <= 0 || S <= 0 || H <= 0 || (ldq & 7) || (ldk & 7)) return MAVLM_E_ARG;
  if (head_dim != 448 && head_dim != 128) return MAVLM_E_SHAPE;
  mavlm_colsum_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.lse2 = lse2; a.part = part; a.R = R; a.S = S; a.H = H; a.scale = scale;
  return (int)mavlm_launch_colsum_hd(a, head_dim, dtype, (hipStream_t)stream);
}

int64_t mavlm_attention_colsum_floats(int32_t R, int32_t S, int32_t H) {
  return (R > 0 && S > 0 && H > 0) ? (int64_t)mavlm_colsum_part_floats(R, S, H) : 0;
}

int mavlm_attention_colsum_plan(int32_t R, int32_t S, int32_t H, int32_t info[2]) {
  if (!info || R <= 0 || S <= 0 || H <= 0) return MAVLM_E_ARG;
  int planes = 1;
  info[0] = mavlm_colsum_plan(R, S, H, &planes);
  info[1] = mavlm_colsum_planes(R, S, H);
  return 0;
}

int mavlm_attention_colsum(const void* Q, int32_t ldq, const void* K, int32_t ldk, const float* lse2, float* part,
                           int64_t part_floats, int32_t R, int32_t S, int32_t H, float scale, int32_t dtype, void* stream) {
  if (R <= 0 || S <= 0 || H <= 0 || !part || part_floats < (int64_t)mavlm_colsum_part_floats(R, S, H)) return MAVLM_E_ARG;
  mavlm_colsum_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.lse2 = lse2; a.part = part; a.R = R; a.S = S; a.H = H; a.scale = scale;
  hipError_t e = mavlm_launch_colsum(a, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_layernorm(const float* xin, const void* res, int32_t ldr, const float* gamma, const float* beta, void* out,
                    int32_t rows, int32_t D, float eps, int32_t dtype, void* stream) {
  hipError_t e = mavlm_launch_layernorm(xin, res, ldr, gamma, beta, out, rows, D, eps, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_pool_bilinear(const void* xin, void* out, const void* pe_table, const int64_t* idx, int32_t F, int32_t side,
                        int32_t stride, int32_t D, int32_t dtype, void* stream) {
  hipError_t e = mavlm_launch_pool_bilinear(xin, out, pe_table, idx, F, side, stride, D, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_row_add(const void* xin, const int64_t* src, const void* table, const int64_t* idx, void* out, int32_t T,
                  int32_t P, int32_t D, int32_t dtype, void* stream) {
  hipError_t e = mavlm_launch_row_add(xin, src, table, idx, out, T, P, D, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

// ---- backward pass ------------------------------------------------------------------------
int mavlm_attention_bwd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, const void* O,
                        int32_t ldo, const void* dO, int32_t lddo, const float* lse2, float* delta, void* dQ,
                        int32_t lddq, void* dK, int32_t lddk, void* dV, int32_t lddv, int32_t R, int32_t S, int32_t H,
                        float scale, int32_t dtype, void* stream) {
  if (!Q || !K || !V || !O || !dO || !lse2 || !delta || R <= 0 || S <= 0 || H <= 0) return MAVLM_E_ARG;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3))
    return MAVLM_E_ARG;
  const int w = H * 128;
  if (ldq < w || ldk < w || ldv < w || ldo < w || lddo < w || (dQ && lddq < w) || (dK && lddk < w) || (dV && lddv < w))
    return MAVLM_E_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO) & 15) return MAVLM_E_ARG;
  mavlm_attn_bwd_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.dO = dO; a.lddo = lddo;
  a.lse2 = lse2; a.delta = delta; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  const double units = (dQ ? 3.0 : 0.0) + (dK ? 3.0 : 0.0) + (dV ? 2.0 : 0.0);
  mavlm_prof_scope prof(MAVLM_K_ATTN_BWD, units * 2.0 * R * (double)S * H * 128.0, 2.0 * 128.0 * H * (4.0 * R + 4.0 * S),
                        (hipStream_t)stream);
  hipError_t e = mavlm_launch_attention_bwd(a, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_attention_bwd_hd(const void* Q, int32_t ldq, const void* K, int32_t ldk, const void* V, int32_t ldv, const void* O,
                           int32_t ldo, const void* dO, int32_t lddo, const float* lse2, float* delta, void* dQ,
                           int32_t lddq, void* dK, int32_t lddk, void* dV, int32_t lddv, int32_t R, int32_t S, int32_t H,
                           int32_t head_dim, float scale, int32_t dtype, void* stream) {
  if (!Q || !K || !V || !O || !dO || !lse2 || !delta || R <= 0 || S <= 0 || H <= 0) return MAVLM_E_ARG;
  if (head_dim != 448) return MAVLM_E_SHAPE;
  if ((ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 7) || (lddo & 7) || (lddq & 3) || (lddk & 3) || (lddv & 3))
    return MAVLM_E_ARG;
  const int w = H * head_dim;
  if (ldq < w || ldk < w || ldv < w || ldo < w || lddo < w || (dQ && lddq < w) || (dK && lddk < w) || (dV && lddv < w))
    return MAVLM_E_ARG;
  if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O | (uintptr_t)dO) & 15) return MAVLM_E_ARG;
  if (((uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV) & 7) return MAVLM_E_ARG;
  mavlm_attn_bwd_args a;
  a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo; a.dO = dO; a.lddo = lddo;
  a.lse2 = lse2; a.delta = delta; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
  a.R = R; a.S = S; a.H = H; a.scale = scale;
  // (algorithmic flops: 3 products for dQ, 3 for dK, 2 for dV; the kernels run 5 / 5 / 2 - attention_bwd_hd.hip)
  const double units = (dQ ? 3.0 : 0.0) + (dK ? 3.0 : 0.0) + (dV ? 2.0 : 0.0);
  mavlm_prof_scope prof(MAVLM_K_ATTN_BWD, units * 2.0 * R * (double)S * H * head_dim, 2.0 * head_dim * H * (4.0 * R + 4.0 * S),
                        (hipStream_t)stream);
  hipError_t e = mavlm_launch_attention_bwd_hd(a, head_dim, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_linear_splitk(const void* A, int32_t lda, const void* W, int32_t ldw, void* C, int32_t M, int32_t N, int32_t K,
                        int32_t splits, float* ws, const float* zero_bias, int32_t dtype, void* stream) {
  if (!A || !W || !C || !ws || !zero_bias || M < 0 || splits < 1) return MAVLM_E_ARG;
  if (N % 128 || K % 64 || N <= 0 || K <= 0) return MAVLM_E_SHAPE;
  mavlm_gemm_args g;
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.bias = zero_bias; g.res = nullptr; g.ldr = 0; g.C = C; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.epilogue = MAVLM_EPI_F32;
  hipError_t e = mavlm_launch_gemm_splitk(g, splits, ws, zero_bias, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int64_t mavlm_layernorm_bwd_ws_floats(int32_t D) { return (int64_t)mavlm_layernorm_bwd_partial_floats(D); }

int mavlm_layernorm_bwd(const void* dy, const float* xin, const void* res, int32_t ldr, const float* gamma, void* dz,
                        float* dgamma, float* dbeta, float* ws, int32_t rows, int32_t D, float eps, int32_t dtype,
                        void* stream) {
  if (!dy || !xin || !gamma || !dz || !dgamma || !dbeta || !ws) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_layernorm_bwd(dy, xin, res, ldr, gamma, dz, dgamma, dbeta, ws, rows, D, eps, dtype,
                                            (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_transpose(const void* in, int32_t ldi, int32_t rows, int32_t cols, void* out, int32_t ldo, void* stream) {
  if (!in || !out) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_transpose(in, ldi, rows, cols, out, ldo, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_rowsum(const void* in, int32_t ld, int32_t rows, int32_t cols, float* out, int32_t dtype, void* stream) {
  if (!in || !out) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_rowsum(in, ld, rows, cols, out, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_act(int32_t kind, const void* xin, const void* dy, void* out, int64_t n, int32_t dtype, void* stream) {
  if (!xin || !out || n < 0) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_act(kind, xin, dy, out, (size_t)n, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

// wide-head attention backward: element-wise pieces between the per-head GEMMs (backward.hip)
int mavlm_attention_probs(const float* S, int32_t lds_, const float* lse2, void* P, int32_t ldp, int32_t R, int32_t cols,
                          int32_t valid, float scale, int32_t dtype, void* stream) {
  if (!S || !lse2 || !P) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_attn_probs(S, lds_, lse2, P, ldp, R, cols, valid, scale * 1.44269504088896340736f, dtype,
                                         (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_attention_dscores(const float* S, int32_t lds_, const float* dP, int32_t lddp, const float* lse2, const float* delta,
                            void* dS, int32_t ldds, int32_t R, int32_t cols, int32_t valid, float scale, int32_t dtype,
                            void* stream) {
  if (!S || !dP || !lse2 || !delta || !dS) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_attn_dscores(S, lds_, dP, lddp, lse2, delta, dS, ldds, R, cols, valid,
                                           scale * 1.44269504088896340736f, scale, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_rowdot_heads(const void* a, int32_t lda, const void* b, int32_t ldb, float* out, int32_t R, int32_t H, int32_t head_dim,
                       int32_t dtype, void* stream) {
  if (!a || !b || !out) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_rowdot(a, lda, b, ldb, out, R, H, head_dim, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

// ---- inactive variants of the reference -------------------------------------------------------
int mavlm_frame_mean(const void* xin, void* out16, float* out32, int32_t F, int32_t P, int32_t D, int32_t dtype,
                     void* stream) {
  if (!xin || (!out16 && !out32)) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_frame_mean(xin, out16, out32, F, P, D, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_adjacent_cosine(const float* v, float* out, int32_t n, int32_t D, float eps, void* stream) {
  if (!v || !out) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_adjacent_cosine(v, out, n, D, eps, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_ARG : (int)e;
}

int mavlm_gru_sequence(const float* xg, const void* whh, const float* bhh, void* out, int32_t F, int32_t H, int32_t ndir,
                       int32_t dtype, void* stream) {
  if (!xg || !whh || !bhh || !out) return MAVLM_E_ARG;
  hipError_t e = mavlm_launch_gru_seq(xg, whh, bhh, out, F, H, ndir, dtype, (hipStream_t)stream);
  return e == hipErrorInvalidValue ? MAVLM_E_SHAPE : (int)e;
}

}  // extern "C"
