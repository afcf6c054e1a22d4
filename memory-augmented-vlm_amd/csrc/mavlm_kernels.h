// Internal launch interface between the C-ABI layer (mavlm_api.hip) and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

enum { MAVLM_EPI_BIAS = 0, MAVLM_EPI_RELU = 1, MAVLM_EPI_GELU = 2, MAVLM_EPI_RES_F32 = 3, MAVLM_EPI_F32 = 4, MAVLM_EPI_LN = 5 };

// EPI_LN (gemm256.hip): C = LayerNorm(A.W^T + bias + res) * gamma + beta in 16 bits - the Residual block of the reference in
// one kernel (the fp32 dense output never goes through HBM).  gran / ctl: mavlm_gemm_ln_ws_bytes() of scratch, see there.
struct mavlm_ln_epilogue {
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float eps = 0.f;
  unsigned long long* gran = nullptr;   // [row blocks][N/256][256][2] {epoch, value} granules
  unsigned* ctl = nullptr;              // {arrivals of this launch, launch counter, timeout flag, -}
  float* pre_out = nullptr;             // optional [M, N] fp32: dense + bias (before the residual), for the training path
  int wide = -1;                        // mavlm_gemm_ln_supported's `wide` of the caller (a context's snapshot; -1 = the hook)
};

struct mavlm_gemm_args {
  const void* A; int lda;        // [M,K] 16-bit, row stride lda elements
  const void* W; int ldw;        // [N,K] 16-bit (nn.Linear weight layout)
  const float* bias;             // [N] fp32
  const void* res; int ldr;      // [M,N] 16-bit residual (EPI_RES_F32 only)
  void* C; int ldc;              // [M,N] 16-bit, or fp32 for EPI_RES_F32
  int M, N, K;
  int epilogue;
  float* splitk_ws = nullptr;    // mavlm_gemm_split_ws_floats(M,N,K) floats, or null = never split the contraction
  int planes_only = 0;           // split-K: leave the fp32 planes in splitk_ws, no reduction (the caller reduces: dense + LayerNorm);
                                 // *planes_out (if set) receives the number of planes, 0 when the shape takes no split
  int* planes_out = nullptr;
  // Row-batched output (256-column-tile kernels only; 0 = plain [M, ldc] output): the M rows are blocks of c_rpb rows; block
  // q goes to batch element b = q % c_nb as its (q / c_nb)-th block, i.e. row m lands at
  // C + b * c_bstride + ((q / c_nb) * c_rpb + m % c_rpb) * ldc (elements).  Used where the stacked rows of several videos
  // (mavlm_config::batch) are written to per-video buffers: the evolution K/V ring, the fused-token blocks.
  int c_rpb = 0, c_nb = 1;
  long long c_bstride = 0;
  mavlm_ln_epilogue ln;          // EPI_LN only
};
// wide: 1 = rows of up to 4096 columns (test mode), 0 = up to 1024, -1 = the process-wide hook g_mavlm_gemm_ln_wide
bool mavlm_gemm_ln_supported(int M, int N, int K, int wide = -1);
extern int g_mavlm_gemm_ln_wide;
size_t mavlm_gemm_ln_ws_bytes(int M, int N);
hipError_t mavlm_launch_gemm(const mavlm_gemm_args& g, int dtype, hipStream_t s);
// split-K plan for GEMMs with few output tiles and a long contraction (small M, K >= 2048): 1 = none.  Pure function of
// the shape: the fused step and the stand-alone operator take the same path.
int mavlm_gemm_splits(int M, int N, int K, int epilogue, int ldc);
size_t mavlm_gemm_split_ws_floats(int M, int N, int K, int epilogue, int ldc);
// 256x256x64 8-wave kernel (gemm256.hip); mavlm_launch_gemm picks it when the grid fills the chip
bool mavlm_gemm256_supported(const mavlm_gemm_args& g);
hipError_t mavlm_launch_gemm256(const mavlm_gemm_args& g, int dtype, hipStream_t s);
hipError_t mavlm_launch_gemm256_splitk(const mavlm_gemm_args& g, int splits, int ksplit, int dtype, hipStream_t s);
// persistent 256x256x64 kernel (gemm256p.hip): no residual epilogue, K >= 128
bool mavlm_gemm256p_supported(const mavlm_gemm_args& g);
hipError_t mavlm_launch_gemm256p(const mavlm_gemm_args& g, int dtype, hipStream_t s);
// 128x256x64 4-wave kernel, two workgroups per CU (gemm128.hip): all epilogues incl. EPI_LN and row-batched outputs
bool mavlm_gemm128_supported(const mavlm_gemm_args& g);
hipError_t mavlm_launch_gemm128(const mavlm_gemm_args& g, int dtype, hipStream_t s);
// split-K form for the long contractions of the backward (gemm.hip); ws = [splits][M][N] fp32, ldc must equal N
hipError_t mavlm_launch_gemm_splitk(const mavlm_gemm_args& g, int splits, float* ws, const float* zero_bias, int dtype,
                                    hipStream_t s);
extern int g_mavlm_gemm_tile;   // 0 = auto, 128 / 256 = forced non-persistent, 257 = forced persistent, 129 = forced 128x256 two-per-CU (tuning hook)
// height of the 256-column workgroup tile for an M x N output: 256 or 224 rows (gemm256.hip); same results either way
int mavlm_gemm_tile_rows(int M, int N);
extern int g_mavlm_gemm_rows;   // 0 = auto, 224 / 256 = forced (tuning hook)

struct mavlm_attn_args {
  const void* Q; int ldq;        // [R, >=H*128] 16-bit; head h at column h*128
  const void* K; int ldk;        // [S, ...]
  const void* V; int ldv;        // [S, ...]
  void* O; int ldo;              // [R, H*128] 16-bit
  float* lse2;                   // [H, R] fp32 (log2-domain log-sum-exp) or null
  int R, S, H;
  float scale;                   // 1/sqrt(head_dim)
  float* split_ws = nullptr;     // mavlm_attention_split_ws_floats(R,S,H) floats, or null = never split the keys
  // frame-score variant (mavlm_launch_attention3_frames): the keys are S / frame_keys frames of frame_keys patches
  float* frame_scr = nullptr;    // mavlm_attention_frames_scr_floats(...) floats of scratch
  float* frame_out = nullptr;    // mavlm_attention_frames_out_floats(...) floats: per-wave partial frame sums
  int frame_keys = 0;
  // row batch (attention3.hip): the launch serves nb independent videos.  H counts ALL heads (nb x heads per video); Q / O
  // hold the nb x R query rows of the videos one after the other, video b's keys start kv_bstride elements after video
  // b-1's (K and V alike); lse2 is [H, R].  nb = 1: a single video.
  int nb = 1;
  long long kv_bstride = 0;
};
hipError_t mavlm_launch_attention(const mavlm_attn_args& a, int dtype, hipStream_t s);
// the same at head_dim 448 (attention_hd.hip): frame_scr = mavlm_attention_hd_frames_scr_floats floats (one 8-byte entry per query row
// and 32-key tile), frame_out = mavlm_attention_hd_frames_out_floats floats; a.lse2 required
bool mavlm_attention_hd_frames_supported(int R, int S, int H, int head_dim, int frame_keys);
size_t mavlm_attention_hd_frames_scr_floats(int R, int S, int H);
size_t mavlm_attention_hd_frames_out_floats(int R, int S, int H, int frame_keys);
int mavlm_attention_hd_frames_rows_per_video(int R, int Hv);
hipError_t mavlm_launch_attention_hd_frames(const mavlm_attn_args& a, int head_dim, int dtype, hipStream_t s);
hipError_t mavlm_launch_frame_tiles(const float* fent, const float* lse2, float* fout, int R, int H, int Hv, int nt_all,
                                    int tile_keys, int frame_keys, int FN, hipStream_t s);
// head_dim <= 128, the tile-entry form (attention3.hip FR = 2; 64-key tiles): for the small grids that split their keys, where the
// per-(row, frame) form above is not available.  frame_scr = H*R*ceil(S/64)*2 floats, frame_out = H*ceil(R/64)*(S/frame_keys)
// floats (rows per video for the finish kernel: Hv*ceil(R/64)); a.lse2 required
bool mavlm_attention_frame_tiles_supported(int R, int S, int H, int frame_keys);
size_t mavlm_attention_frame_tiles_scr_floats(int R, int S, int H);
size_t mavlm_attention_frame_tiles_out_floats(int R, int S, int H, int frame_keys);
hipError_t mavlm_launch_attention3_frame_tiles(const mavlm_attn_args& a, int dtype, hipStream_t s);
// forward + per-frame probability mass in one pass (attention3.hip; head_dim 128, frame_keys % 4 == 0, <= 64 frames)
bool mavlm_attention_frames_supported(int R, int S, int H, int frame_keys);
size_t mavlm_attention_frames_scr_floats(int R, int S, int H, int frame_keys);
size_t mavlm_attention_frames_out_floats(int R, int S, int H, int frame_keys);
hipError_t mavlm_launch_attention3_frames(const mavlm_attn_args& a, int dtype, hipStream_t s);
// scores[b][f] = (1 / P) sum of video b's `rows` partial frame sums (mavlm_attention_frames_rows_per_video)
hipError_t mavlm_launch_frame_finish(const float* fout, int rows, int nb, int F, int P, void* out, int out_f32, int dtype,
                                     hipStream_t s);
int mavlm_attention_frames_rows_per_video(const mavlm_attn_args& a);
extern int g_mavlm_frame_score_mode;   // 1 (default) = fused into the last layer's forward, 0 = column-sum pass
// split-KV plan for grids too small to fill the chip (attention3.hip): number of key splits (1 = none)
int mavlm_attention_splits(int R, int S, int H, int* tiles_per_split);
// stream-K schedule of the head_dim-128 forward (more units than workgroup slots): persistent workgroups, 0 = not used
int mavlm_attention_streamk_wgs(int R, int S, int H);
void mavlm_attention_plan_info(int R, int S, int H, int info[4]);
int mavlm_attention_plan_unit_(int R, int S, int H, int lv, int a, int b);   // unit at a schedule position (attention3.hip)
size_t mavlm_attention_split_ws_floats(int R, int S, int H);
size_t mavlm_attention_split_ws_floats_max(int R, int S, int H);   // ... over both stream-K workgroup shapes (4 / 8 waves)
// the same for the wide-head kernel (attention_hd.hip) and the merge kernel both use (attention3.hip)
int mavlm_attention_hd_splits(int R, int S, int H, int* tiles_per_split);
size_t mavlm_attention_hd_split_ws_floats(int R, int S, int H, int head_dim);
int mavlm_attention_hd_streamk(int R, int S, int H, int head_dim, int info[3]);
extern int g_mavlm_attn_sk_min_tiles;   // attention3.hip: key tiles (of 64) a unit needs before the stream-K schedules apply
hipError_t mavlm_launch_attention_combine(const float* opart, const float* lpart, void* O, int ldo, float* lse2, int R, int H,
                                          int hd, int ns, int dtype, hipStream_t s);
// software-pipelined LDS-DMA variant (attention3.hip); mavlm_launch_attention dispatches to it
hipError_t mavlm_launch_attention3(const mavlm_attn_args& a, int dtype, hipStream_t s);
extern int g_mavlm_attn_hd_qg;  // head_dim-448 forward: 0 = auto (2 query groups per wave), 1 / 2 = forced (tuning hook)
extern int g_mavlm_attn_impl;   // 0 = auto, 2 = register-staged kernel, 3 = pipelined kernel (tuning hook)

// backward of the head_dim-128 attention (attention_bwd.hip); any of dQ / dK / dV may be null (skipped)
struct mavlm_attn_bwd_args {
  const void* Q; int ldq;
  const void* K; int ldk;
  const void* V; int ldv;
  const void* O; int ldo;        // forward output [R, H*128]
  const void* dO; int lddo;      // its gradient
  const float* lse2;             // [H, R] from the forward
  float* delta;                  // [H, R] fp32 scratch (written here)
  void* dQ; int lddq;
  void* dK; int lddk;
  void* dV; int lddv;
  int R, S, H;
  float scale;
};
hipError_t mavlm_launch_attention_bwd(const mavlm_attn_bwd_args& a, int dtype, hipStream_t s);
// the same for head_dim 448 (attention_bwd_hd.hip); O / dO / dQ / dK / dV are [rows, H*448]
hipError_t mavlm_launch_attention_bwd_hd(const mavlm_attn_bwd_args& a, int head_dim, int dtype, hipStream_t s);

// backward.hip
size_t mavlm_layernorm_bwd_partial_floats(int D);
hipError_t mavlm_launch_layernorm_bwd(const void* dy, const float* x, const void* res, int ldr, const float* gamma,
                                      void* dz, float* dgamma, float* dbeta, float* part, int rows, int D, float eps,
                                      int dtype, hipStream_t s);
hipError_t mavlm_launch_transpose(const void* in, int ldi, int rows, int cols, void* out, int ldo, hipStream_t s);
hipError_t mavlm_launch_rowsum(const void* in, int ld, int rows, int cols, float* out, int dtype, hipStream_t s);
hipError_t mavlm_launch_act(int kind, const void* x, const void* dy, void* out, size_t n, int dtype, hipStream_t s);
hipError_t mavlm_launch_splitk_reduce(const float* part, int splits, size_t n, void* out, int dtype, hipStream_t s,
                                      const float* bias = nullptr, int N = 0, int epilogue = MAVLM_EPI_BIAS);

// variants.hip (inactive variants of the reference)
hipError_t mavlm_launch_frame_mean(const void* x, void* out16, float* out32, int F, int P, int D, int dtype, hipStream_t s);
hipError_t mavlm_launch_adjacent_cosine(const float* v, float* out, int n, int D, float eps, hipStream_t s);
hipError_t mavlm_launch_gru_seq(const float* xg, const void* whh, const float* bhh, void* out, int F, int H, int ndir,
                                int dtype, hipStream_t s);

// element-wise pieces of the wide-head attention backward (backward.hip)
hipError_t mavlm_launch_attn_probs(const float* S, int lds_, const float* lse2, void* P, int ldp, int R, int cols, int valid,
                                   float c, int dtype, hipStream_t s);
hipError_t mavlm_launch_attn_dscores(const float* S, int lds_, const float* dP, int lddp, const float* lse2, const float* delta,
                                     void* dS, int ldds, int R, int cols, int valid, float c, float scale, int dtype,
                                     hipStream_t s);
hipError_t mavlm_launch_rowdot(const void* a, int lda, const void* b, int ldb, float* out, int R, int H, int hd, int dtype,
                               hipStream_t s);

// column sums of the normalised probabilities: part[h][k] = sum_q exp2(s*c - lse2[h][q])
struct mavlm_colsum_args {
  const void* Q; int ldq;
  const void* K; int ldk;
  const float* lse2;             // [H, R]
  float* part;                   // mavlm_colsum_part_floats(R,S,H) floats; the result is [H, S] at its start, or - with
                                 // keep_planes - mavlm_colsum_planes(R,S,H) planes of [H, S] the consumer adds in order
  int R, S, H;
  float scale;
  int keep_planes = 0;
};
// column-sum pass of the head_dim-128 kernels: workgroups of the balanced schedule / planes of `part` (attention3.hip)
int mavlm_colsum_plan(int R, int S, int H, int* planes);
int mavlm_colsum_planes(int R, int S, int H);         // planes the selected kernel writes (1 for the register-staged one)
size_t mavlm_colsum_part_floats(int R, int S, int H);
extern int g_mavlm_colsum_wgs;
hipError_t mavlm_launch_colsum(const mavlm_colsum_args& a, int dtype, hipStream_t s);
hipError_t mavlm_launch_colsum3(const mavlm_colsum_args& a, int dtype, hipStream_t s);   // pipelined (attention3.hip)
// wide heads (attention_hd.hip): head_dim 448 (OV-7B), also 128 for cross-checks; columns of head h start at h*head_dim
hipError_t mavlm_launch_attention_hd(const mavlm_attn_args& a, int head_dim, int dtype, hipStream_t s);
hipError_t mavlm_launch_colsum_hd(const mavlm_colsum_args& a, int head_dim, int dtype, hipStream_t s);

// frame_scores[f] = (1/P) * sum_{p<P} sum_h part[h][f*P+p]     (MemoryController.py:135-139)
hipError_t mavlm_launch_frame_scores(const float* part, int planes, int H, int S, int F, int P, void* out, int out_f32,
                                     int dtype, hipStream_t s);

// out[r,:] = LayerNorm(x[r,:] + res[r,:]) * gamma + beta   (x fp32 [rows, D]; res 16-bit [rows, ldr] or null;
// biased variance; rsqrt(var+eps))
hipError_t mavlm_launch_layernorm(const float* x, const void* res, int ldr, const float* gamma, const float* beta,
                                  void* out, int rows, int D, float eps, int dtype, hipStream_t s);
// the same with x given as `splits` fp32 planes of a split-K GEMM ([splits][rows][D], summed in order) + bias [D]: the reduction pass
// and the LayerNorm in one kernel (same arithmetic as mavlm_launch_splitk_reduce followed by mavlm_launch_layernorm: same bits).
// D % 8 == 0, D <= 4096, ldr % 8 == 0
hipError_t mavlm_launch_layernorm_planes(const float* planes, int splits, const float* bias, const void* res, int ldr,
                                         const float* gamma, const float* beta, void* out, int rows, int D, float eps, int dtype,
                                         hipStream_t s);

// out[t,p,:] = x[src[t],p,:] + table[idx[t],:]   (src null = identity, idx null = row 0)
hipError_t mavlm_launch_row_add(const void* x, const int64_t* src, const void* table, const int64_t* idx, void* out,
                                int T, int P, int D, int dtype, hipStream_t s);

// row_add for B videos in one launch: out[b*vstride + (t*P + p)*D ..] = x[b][src[t], p, :] + table_row[:]  (x = HOST array of B
// device pointers; vstride in elements)
hipError_t mavlm_launch_row_add_batch(const void* const* x, const int64_t* src, const void* table_row, void* out,
                                      long long vstride, int B, int T, int P, int D, int dtype, hipStream_t s);

// `runs` (<= 4) literal row runs: out[b*vstride + (dst[i]+r)*D ..] = src[i][r*D ..] for r < n[i], every video b < B
// (vstride in elements); one launch for the prompt / newline rows of mavlm_fuse_emit
hipError_t mavlm_launch_copy_rows(const void* const* src, const int* n, const long long* dst, int runs, void* out,
                                  long long vstride, int B, int D, hipStream_t s);

// out[f, oy*os+ox, :] = bilinear(x[f, side x side, :]) (+ table[idx[f], :] when table != null); os = ceil(side/stride)
hipError_t mavlm_launch_pool_bilinear(const void* x, void* out, const void* table, const int64_t* idx, int F, int side,
                                      int stride, int D, int dtype, hipStream_t s);

// ---- optional per-kernel HIP-event profiling (bench.py roofline line); off by default, zero cost when off.
enum { MAVLM_K_GEMM = 0, MAVLM_K_ATTN = 1, MAVLM_K_COLSUM = 2, MAVLM_K_LN = 3, MAVLM_K_ROWADD = 4, MAVLM_K_MISC = 5,
       MAVLM_K_ATTN_BWD = 6, MAVLM_K_GEMM_SPLITK = 7, MAVLM_K_TRANSPOSE = 8, MAVLM_K_ATTN_MERGE = 9, MAVLM_K_ATTN_FRAMES = 10, MAVLM_K_GEMM_LN = 11, MAVLM_K_COUNT = 12 };
struct mavlm_prof_scope {
  int slot;
  hipStream_t s;
  mavlm_prof_scope(int kind, double flops, double bytes, hipStream_t stream);
  ~mavlm_prof_scope();
};
