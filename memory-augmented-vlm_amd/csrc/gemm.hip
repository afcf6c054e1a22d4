// C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias)  -- nn.Linear layout (weight [out,in]), bf16/fp16 operands,
// fp32 accumulation on the gfx950 matrix cores (v_mfma_f32_16x16x32_{bf16,f16}).
//
// Replaces the ATen addmm call sites of the reference memory path (nn.Linear q/k/v/dense/mlp/fuser:
// llava/model/memory_module/MemoryController.py:23,37-39,48-50,63-67; llava/model/llava_arch.py:132-136).
//
// Structure (v1): 128x128x64 block tile, 4 waves (2x2) each owning a 64x64 sub-tile = 4x4 MFMA tiles,
// operands staged HBM -> LDS with 16-byte global_load_lds into an XOR-swizzled image (the swizzle is applied
// to the per-lane SOURCE address, the LDS destination stays lane-linear), two LDS stages, one barrier per
// K-tile.  The MFMA is issued as D' = W_tile . A_tile^T so every lane ends up with 4 CONSECUTIVE output
// columns of one row -> 8-byte (16-bit out) / 16-byte (fp32 out) epilogue stores.
//
// Ragged M is allowed (row loads are clamped to M-1, stores masked); N % 128 == 0 and K % 64 == 0 required
// (checked on the host in mavlm_api.hip).
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;        // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // A then B
constexpr int GEMM_LDS = 2 * STAGE_BYTES;      // 64 KiB

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const uint16_t* __restrict__ A, int lda,
                                                         const uint16_t* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias,
                                                         const uint16_t* __restrict__ res, int ldr,
                                                         void* __restrict__ Cout, int ldc, int M, int N, int K,
                                                         int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // split-K (backward dW = dY^T X, contraction over tens of thousands of rows with only (N/128)^2 output tiles):
  // blockIdx.y owns K range [y*ksplit, (y+1)*ksplit) and its own fp32 partial plane (EPI_F32 only)
  if (ksplit > 0) {
    const int kz = blockIdx.y * ksplit;
    A += kz;
    W += kz;
    K = (K - kz < ksplit) ? K - kz : ksplit;
    Cout = (float*)Cout + (size_t)blockIdx.y * M * ldc;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int ntn = N / BN;
  const int ntm = (M + BM - 1) / BM;
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (wg / ntn) * BM;
  const int n0 = (wg % ntn) * BN;

  // ---- staging: wave w issues instructions inst = 4w..4w+3 per operand; one instruction = 8 rows x 128 B.
  const int srow = lane >> 3;                 // row inside the 8-row group
  const int sp = lane & 7;                    // physical 16-B chunk in the row
  const uint16_t* gA[4];
  const uint16_t* gB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = (wave * 4 + j) * 8 + srow;
    const int c = sp ^ ((row >> 1) & 7);      // logical chunk that lives at physical chunk sp
    int ar = m0 + row;
    ar = ar < M ? ar : M - 1;
    gA[j] = A + (size_t)ar * lda + c * 8;
    gB[j] = W + (size_t)(n0 + row) * ldw + c * 8;
  }

  // ---- fragment read offsets (bytes inside an operand tile)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (lane >> 1) & 7;             // ((row>>1)&7) for row = 16*i + fr
  const int offA = (wm * 64 + fr) * 128;
  const int offB = (wn * 64 + fr) * 128;
  const int ck0 = ((fq ^ sw) << 4);           // k-step 0 : chunk fq
  const int ck1 = (((4 + fq) ^ sw) << 4);     // k-step 1 : chunk 4+fq

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int buf, int kt) {
    char* sA = smem + buf * STAGE_BYTES + wave * 4096;
    char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      __builtin_amdgcn_global_load_lds((const MAVLM_GLOBAL void*)(gA[j] + kt * BK), (MAVLM_LDS void*)(sA + j * 1024), 16,
                                       0, 0);
      __builtin_amdgcn_global_load_lds((const MAVLM_GLOBAL void*)(gB[j] + kt * BK), (MAVLM_LDS void*)(sB + j * 1024), 16,
                                       0, 0);
    }
  };

  const int nk = K / BK;
  stage(0, 0);
  __syncthreads();   // drains vmcnt(0): tile 0 resident

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* sA = smem + cur * STAGE_BYTES;
    const char* sB = sA + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ck = ks ? ck1 : ck0;
      typename T::vec8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *(const typename T::vec8*)(sA + offA + i * 2048 + ck);
        b[i] = *(const typename T::vec8*)(sB + offB + i * 2048 + ck);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = T::mfma16(b[j], a[i], acc[i][j]);   // D'[n][m]
    }
    __syncthreads();   // next tile landed (vmcnt(0)) and everyone is done reading buf[cur]
  }

  // ---- epilogue: lane holds C[m][n..n+3], m = m0+wm*64+16i+fr, n = n0+wn*64+16j+4fq.  Every load (4 bias vectors, the
  // 16 residual vectors of the RES epilogue) is issued BEFORE the first store: with the bias load inside the store loop
  // hipcc emitted load / s_waitcnt vmcnt(0) / store sixteen times over - and vmcnt(0) also waits for the previous store,
  // i.e. sixteen serial round trips to memory per tile (~10 us on a 15-30 us small-grid GEMM).
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bv[j] = ksplit > 0 ? f32x4{0.f, 0.f, 0.f, 0.f}                     // split-K planes: bias in the reduce
                       : *(const f32x4*)(bias + n0 + wn * 64 + j * 16 + fq * 4);
  u16x4 rv[4][4];
  if (EPI == MAVLM_EPI_RES_F32) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = m0 + wm * 64 + i * 16 + fr;
      m = m < M ? m : M - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) rv[i][j] = *(const u16x4*)(res + (size_t)m * ldr + n0 + wn * 64 + j * 16 + fq * 4);
    }
  }
  // Stores are BUFFER stores through a descriptor that ends after row M-1: the rows of a ragged last tile are dropped
  // by the hardware, so the store loop has no divergent branch (at every branch join hipcc waited vmcnt(0) for the stores
  // before it).  (32-bit offsets: the launcher checks M * ldc * element size < 4 GiB.)
  constexpr int ESZ = (EPI == MAVLM_EPI_RES_F32 || EPI == MAVLM_EPI_F32) ? 4 : 2;
  __amdgpu_buffer_rsrc_t crs;
  {
    const uintptr_t a = (uintptr_t)Cout;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane(((uint32_t)(M - 1) * (uint32_t)ldc + (uint32_t)N) * (uint32_t)ESZ);
    crs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fq * 4;
      const int off = (m * ldc + n) * ESZ;
      float v0 = acc[i][j][0] + bv[j][0], v1 = acc[i][j][1] + bv[j][1], v2 = acc[i][j][2] + bv[j][2],
            v3 = acc[i][j][3] + bv[j][3];
      if (EPI == MAVLM_EPI_RELU) {
        v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
      } else if (EPI == MAVLM_EPI_GELU) {
        v0 = gelu_erf_fast(v0); v1 = gelu_erf_fast(v1); v2 = gelu_erf_fast(v2); v3 = gelu_erf_fast(v3);
      }
      if (EPI == MAVLM_EPI_RES_F32) {
        f32x4 o;
        o[0] = v0 + T::to_f32(rv[i][j][0]); o[1] = v1 + T::to_f32(rv[i][j][1]);
        o[2] = v2 + T::to_f32(rv[i][j][2]); o[3] = v3 + T::to_f32(rv[i][j][3]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), crs, off, 0, 0);
      } else if (EPI == MAVLM_EPI_F32) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4{v0, v1, v2, v3})), crs, off, 0, 0);
      } else {
        __builtin_amdgcn_raw_buffer_store_b64(pack4<T>(v0, v1, v2, v3), crs, off, 0, 0);
      }
    }
  }
}

template <typename T, int EPI>
hipError_t launch(const mavlm_gemm_args& g, hipStream_t s) {
  if ((double)g.M * g.ldc * 4.0 >= 4294967296.0) return hipErrorInvalidValue;     // 32-bit offsets of the buffer stores
  auto kern = gemm_tn_kernel<T, EPI>;
  static mavlm_per_device_once once;   // per instantiation
  {
    hipError_t e = once.dyn_lds((const void*)kern, GEMM_LDS);
    if (e != hipSuccess) return e;
  }
  const int ntm = (g.M + BM - 1) / BM, ntn = g.N / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(256), GEMM_LDS, s, (const uint16_t*)g.A, g.lda, (const uint16_t*)g.W,
                     g.ldw, g.bias, (const uint16_t*)g.res, g.ldr, g.C, g.ldc, g.M, g.N, g.K, 0);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_splitk(const mavlm_gemm_args& g, int splits, int ksplit, hipStream_t s) {
  if ((double)g.M * g.ldc * 4.0 >= 4294967296.0) return hipErrorInvalidValue;     // (per plane)
  auto kern = gemm_tn_kernel<T, MAVLM_EPI_F32>;
  static mavlm_per_device_once once;
  {
    hipError_t e = once.dyn_lds((const void*)kern, GEMM_LDS);
    if (e != hipSuccess) return e;
  }
  const int ntm = (g.M + BM - 1) / BM, ntn = g.N / BN;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, splits), dim3(256), GEMM_LDS, s, (const uint16_t*)g.A, g.lda,
                     (const uint16_t*)g.W, g.ldw, g.bias, (const uint16_t*)nullptr, 0, g.C, g.ldc, g.M, g.N, g.K, ksplit);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_epi(const mavlm_gemm_args& g, hipStream_t s) {
  switch (g.epilogue) {
    case MAVLM_EPI_BIAS: return launch<T, MAVLM_EPI_BIAS>(g, s);
    case MAVLM_EPI_RELU: return launch<T, MAVLM_EPI_RELU>(g, s);
    case MAVLM_EPI_GELU: return launch<T, MAVLM_EPI_GELU>(g, s);
    case MAVLM_EPI_RES_F32: return launch<T, MAVLM_EPI_RES_F32>(g, s);
    case MAVLM_EPI_F32: return launch<T, MAVLM_EPI_F32>(g, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

int g_mavlm_gemm_tile = 0;

// C16[M,N] = A[M,K] . W[N,K]^T with the contraction split over `splits` workgroup planes (fp32 partials in `ws`,
// [splits][M][N]) and a deterministic reduction.  `zero_bias`: N fp32 zeros (the kernel's epilogue adds a bias).
hipError_t mavlm_launch_gemm_splitk(const mavlm_gemm_args& g0, int splits, float* ws, const float* zero_bias, int dtype,
                                    hipStream_t s) {
  if (g0.M <= 0) return hipSuccess;
  if (g0.N % BN != 0 || g0.K % BK != 0 || g0.K <= 0 || (g0.lda & 7) || (g0.ldw & 7) || (g0.ldc & 3) || splits < 1 ||
      g0.ldc != g0.N)
    return hipErrorInvalidValue;
  const int nk = g0.K / BK;
  if (splits > nk) splits = nk;
  const int ksplit = ((nk + splits - 1) / splits) * BK;
  splits = (g0.K + ksplit - 1) / ksplit;
  mavlm_prof_scope prof(MAVLM_K_GEMM_SPLITK, 2.0 * g0.M * (double)g0.N * g0.K,
                        2.0 * ((double)g0.M * g0.K + (double)g0.N * g0.K) + (4.0 * splits + 2.0) * g0.M * (double)g0.N, s);
  mavlm_gemm_args g = g0;
  void* out16 = g0.C;
  g.C = ws;
  g.bias = zero_bias;
  g.epilogue = MAVLM_EPI_F32;
  hipError_t e = dtype == MAVLM_F16 ? launch_splitk<F16>(g, splits, ksplit, s) : launch_splitk<BF16>(g, splits, ksplit, s);
  if (e != hipSuccess) return e;
  return mavlm_launch_splitk_reduce(ws, splits, (size_t)g0.M * g0.N, out16, dtype, s);
}

// Tile choice: the 256^2 kernel runs one workgroup per CU, so it wants >= ~1 full wave of 256 tiles; below that the
// 128^2 kernel (two workgroups per CU, 4x the tiles) fills the chip better.
static bool use_256(const mavlm_gemm_args& g) {
  if (!mavlm_gemm256_supported(g)) return false;
  // the 256^2 kernels store 16 bytes per lane (fp32: always; 16-bit: widened pairs): C rows must be 16-byte aligned
  const bool f32out = g.epilogue == MAVLM_EPI_RES_F32 || g.epilogue == MAVLM_EPI_F32;
  if (((uintptr_t)g.C & 15) || (f32out ? (g.ldc & 3) : (g.ldc & 7))) return false;
  if (g_mavlm_gemm_tile == 256 || g_mavlm_gemm_tile == 257 || g_mavlm_gemm_tile == 129) return true;
  if (g_mavlm_gemm_tile == 128) return false;
  const long tiles = (long)((g.M + 255) / 256) * (g.N / 256);
  return tiles >= 192;
}


// 128x256 tiles, two workgroups per CU (gemm128.hip) instead of the 256-row kernels.  Pure speed choice: every output
// element sums its K products in the same order in all three kernels (bit-identical results).
static bool use_128x256(const mavlm_gemm_args& g) {
  if (!mavlm_gemm128_supported(g)) return false;
  const bool f32out = g.epilogue == MAVLM_EPI_RES_F32 || g.epilogue == MAVLM_EPI_F32;
  if (((uintptr_t)g.C & 15) || (f32out ? (g.ldc & 3) : (g.ldc & 7))) return false;     // 16-byte stores per lane
  if (g_mavlm_gemm_tile == 129) return true;
  if (g_mavlm_gemm_tile != 0) return false;
  // Automatic choice (measured, tools/diag_gemm128_small.py): the mid-size grids - too few 256-row tiles to fill the chip
  // (< 192: the 128^2 kernel's territory) but at least ~190 tiles of 128 x 256, i.e. most CUs busy with one 4-wave
  // workgroup each - run 5-12 % faster here than on the 128^2 kernel (half the B traffic per flop); e.g. the MLP-up
  // projection of a single video at 8 memory tokens (1568 x 4096 x 1024).  Everywhere else the 256-row kernels win: two
  // 128 x 256 workgroups per CU stage 1.5 x the operand bytes of one 256 x 256 workgroup, and L2 -> LDS staging
  // (~12.8 TB/s chip-wide under MFMA load) is what bounds these loops.
  const long t256 = (long)((g.M + 255) / 256) * (g.N / 256), t128 = (long)((g.M + 127) / 128) * (g.N / 256);
  return t256 < 192 && t128 >= 192;
}

int g_mavlm_gemm_f32_short_splits = 2;    // (tuning hook: mavlm_set_gemm_short_splits; 1 = off; measured 1.0575 / 1.0401 / 1.0510 ms per M = 8 video for 1 / 2 / 4)
int mavlm_gemm_splits(int M, int N, int K, int epilogue, int ldc) {
  if (M <= 0 || N % BN || K % BK || ldc != N || epilogue == MAVLM_EPI_RES_F32) return 1;
  const long tiles = (long)((M + BM - 1) / BM) * (N / BN);
  // at most one 128^2 tile per two CUs and a long contraction.  (Measured: extending this to grids of up to 448 tiles,
  // e.g. the 364-tile GEMMs of R = 1568 at the OneVision-7B width, LOSES 4 % end to end - the partial planes cost more
  // than the idle CUs.)
  // Round 4: the fp32 dense output of a Residual block (EPI_F32) on a small grid with a SHORT contraction (the attention output
  // projection at few memory tokens: 1568 x 1024 x 1024 = 104 tiles x 16 K-tiles, 21 us of per-tile latency): its planes are
  // reduced inside the LayerNorm kernel for free (mavlm_launch_layernorm_planes), so the split pays from K = 1024.
  if (epilogue == MAVLM_EPI_F32 && tiles <= 128 && K >= 1024 && K < 2048 && g_mavlm_gemm_f32_short_splits > 1)
    return g_mavlm_gemm_f32_short_splits;
  if (tiles > 128 || K < 2048) {
    // Round 4: a LONG contraction over a grid that fills 1/4 - 1/2 of the chip with 256-row tiles (the 4D -> D projection of
    // a single video at the OneVision-7B width: 1568 x 3584 x 14336 = 98 tiles x 224 K-tiles): two (up to four) K ranges on
    // the 256-row kernel, 196+ workgroups of 112 K-tiles each + one reduce pass, instead of 364 tiles of 128^2 walking all
    // 224 K-tiles (measured 193 -> ~150 us).  Only from K = 8192: below, the reduce costs what the split saves.
    const long t256 = (long)((M + 255) / 256) * (N / 256);
    if (K >= 8192 && N % 256 == 0 && t256 >= 64 && t256 <= 128) {
      const int s2 = (int)(256 / t256);
      return s2 > 4 ? 4 : (s2 < 2 ? 1 : s2);
    }
    return 1;
  }
  int splits = K / 1024;
  if (splits > 4) splits = 4;
  return splits < 2 ? 1 : splits;
}
// which kernel runs the split planes of mavlm_gemm_splits: the 256-row one for the round-4 rule, else the 128^2 one
static bool splits_on_256(int M, int N, int K) {
  const long tiles = (long)((M + BM - 1) / BM) * (N / BN);
  (void)K;
  return tiles > 128;
}

size_t mavlm_gemm_split_ws_floats(int M, int N, int K, int epilogue, int ldc) {
  const int sp = mavlm_gemm_splits(M, N, K, epilogue, ldc);
  return sp > 1 ? (size_t)sp * M * N : 0;
}

hipError_t mavlm_launch_gemm(const mavlm_gemm_args& g, int dtype, hipStream_t s) {
  if (g.planes_out) *g.planes_out = 0;
  if (g.M <= 0) return hipSuccess;
  if (g.epilogue == MAVLM_EPI_LN) {
    // dense + residual + LayerNorm in one kernel: the non-persistent 256-column-tile kernel only (its row-block exchange
    // needs the N / 256 workgroups of a row block in flight together)
    if (!mavlm_gemm_ln_supported(g.M, g.N, g.K, g.ln.wide) || !mavlm_gemm256_supported(g) || g.c_rpb > 0 || !g.res || (g.ldr & 3) ||
        !g.ln.gamma || !g.ln.beta || !g.ln.gran || !g.ln.ctl || ((uintptr_t)g.C & 15) || (g.ldc & 7) || (g.lda & 7) || (g.ldw & 7))
      return hipErrorInvalidValue;
    mavlm_prof_scope prof(MAVLM_K_GEMM_LN, 2.0 * g.M * (double)g.N * g.K,
                          2.0 * ((double)g.M * g.K + (double)g.N * g.K) + 4.0 * g.M * (double)g.N, s);
    if (use_128x256(g)) return mavlm_launch_gemm128(g, dtype, s);
    return mavlm_launch_gemm256(g, dtype, s);
  }
  if (g.c_rpb > 0) {
    // row-batched output: only the 256-column-tile kernels scatter their rows (the stacked rows of several videos fill
    // the chip - there is no small-grid case to serve)
    if (g.M % g.c_rpb != 0 || g.c_nb < 1 || (g.M / g.c_rpb) % g.c_nb != 0 || (g.c_bstride & 7) || g.epilogue == MAVLM_EPI_RES_F32 ||
        !mavlm_gemm256_supported(g) || ((uintptr_t)g.C & 15) || (g.ldc & 7) || (g.lda & 7) || (g.ldw & 7))
      return hipErrorInvalidValue;
    const double osz = g.epilogue == MAVLM_EPI_F32 ? 4.0 : 2.0;
    mavlm_prof_scope prof(MAVLM_K_GEMM, 2.0 * g.M * (double)g.N * g.K,
                          2.0 * ((double)g.M * g.K + (double)g.N * g.K) + osz * g.M * (double)g.N, s);
    if (use_128x256(g)) return mavlm_launch_gemm128(g, dtype, s);
    const int rows = mavlm_gemm_tile_rows(g.M, g.N);
    const long tiles = (long)((g.M + rows - 1) / rows) * (g.N / 256);
    if (tiles > 256 && mavlm_gemm256p_supported(g)) return mavlm_launch_gemm256p(g, dtype, s);
    return mavlm_launch_gemm256(g, dtype, s);
  }
  if (g.splitk_ws != nullptr && g_mavlm_gemm_tile == 0) {
    // few output tiles, long contraction (e.g. the 4D -> D projections at 8 memory tokens): one workgroup per CU would
    // walk all K-tiles alone; split the contraction over blockIdx.y instead and apply bias + epilogue in the reduce
    const int sp = mavlm_gemm_splits(g.M, g.N, g.K, g.epilogue, g.ldc);
    if (sp > 1) {
      if ((g.lda & 7) || (g.ldw & 7)) return hipErrorInvalidValue;
      const int nk = g.K / BK;
      const int ksplit = ((nk + sp - 1) / sp) * BK;
      const int splits = (g.K + ksplit - 1) / ksplit;
      mavlm_prof_scope prof(MAVLM_K_GEMM, 2.0 * g.M * (double)g.N * g.K,
                            2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (8.0 * splits + 2.0) * g.M * (double)g.N, s);
      mavlm_gemm_args p = g;
      p.C = g.splitk_ws;
      p.epilogue = MAVLM_EPI_F32;
      hipError_t e = splits_on_256(g.M, g.N, g.K) ? mavlm_launch_gemm256_splitk(p, splits, ksplit, dtype, s)
                     : (dtype == MAVLM_F16 ? launch_splitk<F16>(p, splits, ksplit, s) : launch_splitk<BF16>(p, splits, ksplit, s));
      if (e != hipSuccess) return e;
      if (g.planes_out) *g.planes_out = splits;
      if (g.planes_only) return hipSuccess;                 // (the caller reduces: dense_ln's reduce + LayerNorm kernel)
      // the planes are pure partial products (the kernel skips its bias when ksplit > 0); bias + epilogue once, here
      return mavlm_launch_splitk_reduce(g.splitk_ws, splits, (size_t)g.M * g.N, g.C, dtype, s, g.bias, g.N, g.epilogue);
    }
  }
  if (g.N % BN != 0 || g.K % BK != 0 || g.K <= 0 || (g.lda & 7) || (g.ldw & 7) || (g.ldc & 3)) return hipErrorInvalidValue;
  if (g.epilogue == MAVLM_EPI_RES_F32 && (g.res == nullptr || (g.ldr & 3))) return hipErrorInvalidValue;
  const double osz = g.epilogue == MAVLM_EPI_RES_F32 ? 6.0 : (g.epilogue == MAVLM_EPI_F32 ? 4.0 : 2.0);
  mavlm_prof_scope prof(MAVLM_K_GEMM, 2.0 * g.M * (double)g.N * g.K,
                        2.0 * ((double)g.M * g.K + (double)g.N * g.K) + osz * g.M * (double)g.N, s);
  if (use_128x256(g)) return mavlm_launch_gemm128(g, dtype, s);
  if (use_256(g)) {
    // persistent kernel when workgroups get more than one tile each (its pipeline never drains between tiles);
    // with at most one tile per CU the plain kernel is the same work with less code in flight
    const int rows = mavlm_gemm_tile_rows(g.M, g.N);
    const long tiles = (long)((g.M + rows - 1) / rows) * (g.N / 256);
    const bool want_p = g_mavlm_gemm_tile == 257 || (g_mavlm_gemm_tile == 0 && tiles > 256);
    if (want_p && mavlm_gemm256p_supported(g)) return mavlm_launch_gemm256p(g, dtype, s);
    return mavlm_launch_gemm256(g, dtype, s);
  }
  return dtype == MAVLM_F16 ? launch_epi<F16>(g, s) : launch_epi<BF16>(g, s);
}
