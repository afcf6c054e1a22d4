// PERSISTENT variant of gemm256_kernel (gemm256.hip; read that header first - tile geometry, LDS image, 4-phase
// K-tile, ping-pong skew and the hazard analysis are identical).  One workgroup per CU walks tiles
// blockIdx.x, blockIdx.x + gridDim.x, ...  and the K pipeline NEVER DRAINS between tiles: K-tiles are numbered
// v = it*nk + kt across the tile sequence, stage = v & 1, and the half-tile DMAs issued during the last two K-tiles
// of a tile already fetch the first two K-tiles of the next one.  The epilogue of tile `it` therefore runs while
// those DMAs are in flight (the non-persistent kernel pays ~8 us per tile for prologue + epilogue + relaunch at
// K = 1024, a quarter of its tile time).
//   * The epilogue must not contain VMEM loads: vmcnt retires in order, so waiting for a bias or residual load
//     would drain the DMA queue.  The tile's 256 bias values arrive by a 1 KiB LDS-DMA (issued by wave 0 as the
//     OLDEST operation of the tile's first K-tile, so the counted vmcnt(6) waits are unchanged) and are read from
//     LDS; the residual add of the LN-fused epilogue is done by the LayerNorm kernel instead (EPI_F32 here).
//   * Epilogue stores count in vmcnt too, but they are older than the three half-tiles that stay in flight.
// Same contract as gemm_tn_kernel (gemm.hip): C[M,N] = epi(A[M,K] . W[N,K]^T + bias), nn.Linear layout.
//
// Structure (after cdna_hip_programming.md "The 256^2 8-phase template", re-derived for this layout):
//   * 8 waves (2 along M x 4 along N), wave tile 128x64 = 8x4 MFMA 16x16x32 tiles (128 accumulator VGPRs), one
//     workgroup per CU, 128 KiB LDS = 2 stages x {A0,A1,B0,B1} half-tiles of 128 rows x 64 k (16 KiB each).
//   * Operands arrive by LDS-DMA (16-byte global_load_lds), XOR-swizzled through the per-lane SOURCE address;
//     a half-tile is 2 DMA instructions per wave.
//   * A K-tile is consumed in 4 phases of 16 MFMAs (one 64x32 quadrant of the wave tile x K=64); reads 12 / 4 / 8 / 0 and
//     DMAs 0 / 1 / 1 / 2 half-tiles per phase, one counted vmcnt(6) per K-tile: the table in gemm256.hip.
//   * Raw s_barrier (a __syncthreads() would drain the DMA queue), MFMA clusters bracketed by s_setprio.
//   * Ping-pong: every phase is two barrier segments, L (fragment reads + DMA issue + the counted wait) and M (16
//     MFMAs).  Waves 4-7 run ONE SEGMENT BEHIND waves 0-3 (they execute one extra s_barrier before their first
//     segment, waves 0-3 one extra after their last, so both groups execute the same number of barriers).  Each
//     SIMD hosts one wave of either group, so while one group's M segment owns the matrix pipe the other group
//     issues its LDS reads and DMAs.  Without the skew all eight waves are in lockstep: both waves of a SIMD do
//     their MFMAs back to back and then both sit in the L segment with the pipe idle (measured: 3.8k clk per
//     K-tile against 2.05k of MFMA).
//
// Hazards (LDS-DMA is ordered by nothing but the issuing wave's vmcnt + a barrier), with g = global barrier index,
// the leading group executing segment g and the trailing group segment g-1, K-tile kt = segments 8kt+1 .. 8kt+8
// (L1 M1 L2 M2 L3 M3 L4 M4), stage s = kt & 1:
//   (reads / WAR / RAW per barrier index: as listed in gemm256.hip; the bias DMA of a tile is the oldest VMEM operation of
//   its first K-tile's phase 2, older than the six DMAs the counted wait leaves in flight.)
#include "mavlm_common.h"
#include "mavlm_kernels.h"

extern int g_mavlm_gemm_order;

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int HALF = 128 * BK2 * 2;          // 16 KiB half-tile
constexpr int STAGE2 = 4 * HALF;             // A0 A1 B0 B1
constexpr int GEMM256_LDS = 2 * STAGE2;      // 128 KiB
constexpr int BIAS_OFF = GEMM256_LDS;        // 2 x 1 KiB bias ring behind the stages
constexpr int GEMM256P_LDS = GEMM256_LDS + 2048;

// raw workgroup barrier fenced against compiler motion of memory operations (s_barrier itself is IntrNoMem)
#define MAVLM_BAR()                          \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  } while (0)
// all LDS reads of this phase retired before its MFMAs (WAR rule above); sched_barrier: hipcc may hoist a
// register-only MFMA above an inline-asm wait (cdna_hip_programming.md rule 18)
#define MAVLM_LGKM0()                                          \
  do {                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)

__device__ __forceinline__ float gelu_erf2p(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// MT1: 4 -> 256-row tiles, 3 -> 224-row tiles (see gemm256_kernel)
template <typename T, int EPI, int MT1>
__global__ __launch_bounds__(512, 2) void gemm256p_kernel(const uint16_t* __restrict__ A, int lda,
                                                          const uint16_t* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, void* __restrict__ Cout,
                                                          int ldc, int M, int N, int K, int total_tiles, int c_rpb,
                                                          int c_nb, long long c_bs, int g_order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  constexpr int MHALF = 64 + 16 * MT1;
  constexpr int BMT = 2 * MHALF;

  const int ntn = N / BN2;
  const int G = gridDim.x;
  const int ntl = (total_tiles - (int)blockIdx.x + G - 1) / G;      // tiles of this workgroup (>= 1)
  const int nk = K / BK2;                                           // >= 2 (host check)
  const int VT = ntl * nk;                                          // virtual K-tiles of this workgroup

  // ---- LDS-DMA sources: wave w stages 8-row groups g = 2w, 2w+1 of every half-tile.  32-bit element offsets of
  // the tile being computed (cur) and of the next one (nxt): the DMAs run up to two K-tiles ahead.
  const int srow = lane >> 3, sp = lane & 7;
  int offA[2][2][2], offB[2][2][2];                                 // [cur/nxt][half][inst]
  int m0c = 0, n0c = 0;
  auto tile_origin = [&](int it, int& m0, int& n0) {
    int lin = (int)blockIdx.x + it * G;
    // XCD-aware order inside every full window of G tiles (blocks b, b+8 share an XCD): the 32 tiles an XCD works on at a
    // time should share operands through its L2.  N = 1024 (4 column tiles): 32 consecutive tiles = 8 row blocks x 4 column
    // tiles.  N = 4096 (round 4): a window of 256 tiles is 16 row blocks x 16 column tiles; an XCD takes an
    // 8 x 4 block of it instead of 2 x 16 consecutive tiles - per K-tile its 32 workgroups stage 8 A + 4 B
    // half-operands from L2 instead of 2 + 16 (measured +0.7 ... +1.8 %; the 4 x 8 form of N = 2048 measured -1 %: not taken).
    // Speed only: any bijection of the window is correct.
    if ((it + 1) * G <= total_tiles && (G & 7) == 0) {
      if (g_order == 1 && G == 256 && ntn == 16) {
        const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;       // XCD label, index inside the XCD (0..31)
        const int xc = x % (ntn >> 2), xr = x / (ntn >> 2);
        lin = it * G + (xr * 8 + (j >> 2)) * ntn + xc * 4 + (j & 3);
      } else {
        lin = it * G + xcd_remap((int)blockIdx.x, G);
      }
    }
    m0 = (lin / ntn) * BMT;
    n0 = (lin % ntn) * BN2;
  };
  auto set_offsets = [&](int which, int m0, int n0) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + srow;
        const int c = sp ^ ((row >> 1) & 7);
        // byte offsets from the operand's base (32-bit: host check); rows past M lie behind the A descriptor's end and
        // read as zeros (never stored)
        offA[which][h][j] = (int)(((unsigned)(m0 + h * MHALF + row) * (unsigned)lda + (unsigned)(c * 8)) * 2u);
        offB[which][h][j] = (int)(((unsigned)(n0 + h * 128 + row) * (unsigned)ldw + (unsigned)(c * 8)) * 2u);
      }
  };
  // Buffer loads (see gemm256_kernel): one descriptor per operand, 32-bit lane offsets, the K-tile as scalar offset
  auto whole_rsrc = [&](const uint16_t* base, int rows, int ld) {
    const uintptr_t a = (uintptr_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(rows - 1) * (uint32_t)ld * 2u + (uint32_t)K * 2u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t rsA = whole_rsrc(A, M, lda), rsB = whole_rsrc(W, N, ldw);
  const unsigned lds_wave = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem + wave * 2048;
  // half-tile ids: 0 = A0, 1 = A1, 2 = B0, 3 = B1 ; u = virtual K-tile ; it = tile being computed
  auto dma = [&](int half_id, int u, int it) {
    const int which = u >= (it + 1) * nk;                           // wave-uniform: next tile?
    const int kt = u - (it + which) * nk;
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));            // M0 = base + constant: recomputed, not hoisted into scarce SGPRs
    const unsigned dst = base + (u & 1) * STAGE2 + half_id * HALF;
    const int h = half_id & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int oa = which ? offA[1][h][j] : offA[0][h][j];     // (static indices: runtime-indexed arrays go to scratch)
      const int ob = which ? offB[1][h][j] : offB[0][h][j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(half_id < 2 ? rsA : rsB, (MAVLM_LDS void*)(uintptr_t)(dst + j * 1024), 16,
                                               half_id < 2 ? oa : ob, kt * BK2 * 2, 0, 0);
    }
  };
  auto dma_bias = [&](int it, int n0) {                             // 256 floats = 64 lanes x 16 B, wave 0 only
    if (wave == 0)
      __builtin_amdgcn_global_load_lds((const MAVLM_GLOBAL void*)(bias + n0 + lane * 4),
                                       (MAVLM_LDS void*)(smem + BIAS_OFF + (it & 1) * 1024), 16, 0, 0);
  };

  // ---- fragment read offsets
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (lane >> 1) & 7;
  const int ck0 = (fq ^ sw) << 4, ck1 = ((4 + fq) ^ sw) << 4;
  const int offA_f = wm * HALF + fr * 128;                               // + mh*8192 + mt*2048
  const int offB_f = 2 * HALF + (wn >> 1) * HALF + ((wn & 1) * 64 + fr) * 128;   // + nh*4096 + nt*2048

  f32x4 acc[4 + MT1][4];
#pragma unroll
  for (int i = 0; i < 4 + MT1; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  typename T::vec8 af[4][2];      // [m-tile of the current 64-row slice][k-step]
  typename T::vec8 bf[4][2];      // [n-tile of the wave's 64 columns][k-step]

  const bool trailing = wm == 1;     // waves 4-7 (wave-uniform: wm comes from a readfirstlane)

  // ---- prologue: bias(0), K-tile 0 completely, K-tile 1 minus its last half-tile (order B0,B1,A0,A1)
  tile_origin(0, m0c, n0c);
  set_offsets(0, m0c, n0c);
  {
    int m0n = 0, n0n = 0;
    if (ntl > 1) tile_origin(1, m0n, n0n);
    set_offsets(1, m0n, n0n);
  }
  dma_bias(0, n0c);
  dma(2, 0, 0); dma(3, 0, 0); dma(0, 0, 0); dma(1, 0, 0);
  dma(2, 1, 0); dma(3, 1, 0); dma(0, 1, 0);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  MAVLM_BAR();

  auto read_a = [&](const char* st, int mh) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mh == 1 && mt >= MT1) continue;
      af[mt][0] = *(const typename T::vec8*)(st + offA_f + mh * 8192 + mt * 2048 + ck0);
      af[mt][1] = *(const typename T::vec8*)(st + offA_f + mh * 8192 + mt * 2048 + ck1);
    }
  };
  auto read_b_half = [&](const char* st, int nh) {
#pragma unroll
    for (int nt = 2 * nh; nt < 2 * nh + 2; ++nt) {
      bf[nt][0] = *(const typename T::vec8*)(st + offB_f + nt * 2048 + ck0);
      bf[nt][1] = *(const typename T::vec8*)(st + offB_f + nt * 2048 + ck1);
    }
  };
#define MAVLM_QUADRANT(MH, NH)                                                              \
  {                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
    _Pragma("unroll") for (int mt = 0; mt < (MH ? MT1 : 4); ++mt)                           \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
      acc[MH * 4 + mt][NH * 2 + nt] = T::mfma16(bf[NH * 2 + nt][ks], af[mt][ks], acc[MH * 4 + mt][NH * 2 + nt]); \
    __builtin_amdgcn_s_setprio(0);                                                          \
  }

  if (trailing) MAVLM_BAR();           // ping-pong skew: pairs with the leading group's first in-loop barrier

  int v = 0;
  for (int it = 0; it < ntl; ++it) {
    for (int kt = 0; kt < nk; ++kt, ++v) {
      const char* st = smem + (v & 1) * STAGE2;
      // -------- phase 1
      read_a(st, 0);
      read_b_half(st, 0);
      MAVLM_BAR();
      MAVLM_LGKM0();
      MAVLM_QUADRANT(0, 0)
      MAVLM_BAR();
      // -------- phase 2
      read_b_half(st, 1);
      if (kt == 0 && it > 0) dma_bias(it, n0c);              // oldest VMEM operation of this K-tile
      if (v + 1 < VT) dma(1, v + 1, it);                     // A1 of the next virtual K-tile
      MAVLM_BAR();
      MAVLM_LGKM0();
      MAVLM_QUADRANT(0, 1)
      MAVLM_BAR();
      // -------- phase 3   (B half-tiles of this stage are dead: everything is in registers)
      read_a(st, 1);
      if (v + 2 < VT) dma(2, v + 2, it);
      MAVLM_BAR();
      MAVLM_LGKM0();
      MAVLM_QUADRANT(1, 1)
      MAVLM_BAR();
      // -------- phase 4   (A half-tiles dead)
      if (v + 2 < VT) {
        dma(3, v + 2, it);
        dma(0, v + 2, it);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K-tile v+1 landed; 3 half-tiles of v+2 stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      MAVLM_BAR();
      MAVLM_QUADRANT(1, 0)
      MAVLM_BAR();
    }

    // ---- epilogue of tile `it` (no barriers, no VMEM loads): the next tile's first K-tiles are already in flight.
    // lane holds C[m][n..n+3], m = m0 + wm*128 + 16 i + fr, n = n0 + wn*64 + 16 j + 4 fq
    const float* bl = (const float*)(smem + BIAS_OFF + (it & 1) * 1024) + wn * 64 + fq * 4;
    f32x4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(bl + j * 16);
      auto act4 = [&](f32x4 v) -> f32x4 {
      if (EPI == MAVLM_EPI_RELU) return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
      if (EPI == MAVLM_EPI_GELU) return gelu_erf_fast4(v);      // packed fp32 math (mavlm_common.h)
      return v;
    };
    // element offset of output row m; row-batched outputs: see gemm256_kernel / mavlm_gemm_args::c_rpb
    auto crow = [&](int m) -> size_t {
      if (c_rpb <= 0) return (size_t)m * ldc;
      const int q = m / c_rpb, r = m - q * c_rpb;
      return (size_t)(q % c_nb) * (size_t)c_bs + ((size_t)(q / c_nb) * c_rpb + r) * ldc;
    };
#pragma unroll
    for (int i = 0; i < 4 + MT1; ++i) {
      const int m = m0c + wm * MHALF + i * 16 + fr;
      const size_t co = crow(m < M ? m : M - 1);
      if (EPI == MAVLM_EPI_F32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0c + wn * 64 + j * 16 + fq * 4;
          const f32x4 o = acc[i][j] + bv[j];
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (m < M) *(f32x4*)((float*)Cout + co + n) = o;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j += 2) {                   // 16 contiguous bytes per lane (widen_pair, mavlm_common.h)
          const f32x4 x = acc[i][j] + bv[j], y = acc[i][j + 1] + bv[j + 1];
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          acc[i][j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
          const f32x4 xa = act4(x), ya = act4(y);
          const u32x4 w = widen_pair(pack4<T>(xa[0], xa[1], xa[2], xa[3]), pack4<T>(ya[0], ya[1], ya[2], ya[3]));
          const int n = n0c + wn * 64 + 16 * (j + (fq & 1)) + 8 * (fq >> 1);
          if (m < M) *(u32x4*)((uint16_t*)Cout + co + n) = w;
        }
      }
    }
    // ---- next tile becomes current; offsets of the one after it
    if (it + 1 < ntl) {
      tile_origin(it + 1, m0c, n0c);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) { offA[0][h][j] = offA[1][h][j]; offB[0][h][j] = offB[1][h][j]; }
      if (it + 2 < ntl) {
        int m0n, n0n;
        tile_origin(it + 2, m0n, n0n);
        set_offsets(1, m0n, n0n);
      }
    }
  }
  if (!trailing) MAVLM_BAR();          // matches the trailing group's last barrier
#undef MAVLM_QUADRANT
}

template <typename T, int EPI, int MT1>
hipError_t launch256ph(const mavlm_gemm_args& g, hipStream_t s) {
  auto kern = gemm256p_kernel<T, EPI, MT1>;
  constexpr int BMT = 2 * (64 + 16 * MT1);
  static mavlm_per_device_once once;
  static int cus = 0;          // MI355X: 256 on every device of a node
  {
    hipError_t e = once.dyn_lds((const void*)kern, GEMM256P_LDS);
    if (e != hipSuccess) return e;
    if (cus <= 0) {
      int dev = 0;
      e = hipGetDevice(&dev);
      if (e != hipSuccess) return e;
      e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (e != hipSuccess || cus <= 0) return hipErrorInvalidValue;
    }
  }
  const int tiles = ((g.M + BMT - 1) / BMT) * (g.N / BN2);
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), GEMM256P_LDS, s, (const uint16_t*)g.A, g.lda, (const uint16_t*)g.W,
                     g.ldw, g.bias, g.C, g.ldc, g.M, g.N, g.K, tiles, g.c_rpb, g.c_nb > 0 ? g.c_nb : 1, (long long)g.c_bstride,
                     g_mavlm_gemm_order);
  return hipGetLastError();
}

template <typename T, int EPI>
hipError_t launch256p(const mavlm_gemm_args& g, hipStream_t s) {
  return mavlm_gemm_tile_rows(g.M, g.N) == 224 ? launch256ph<T, EPI, 3>(g, s) : launch256ph<T, EPI, 4>(g, s);
}

}  // namespace

int g_mavlm_gemm_order = 1;     // tuning hook: 1 = 8 x 4 tile blocks per XCD for N >= 2048 (persistent kernel), 0 = consecutive tiles

// persistent kernel: bias / ReLU / GELU / fp32-out epilogues (no residual: see the header), K >= 128,
// operands addressable with 32-bit element offsets
bool mavlm_gemm256p_supported(const mavlm_gemm_args& g) {
  if (g.N % BN2 != 0 || g.K % BK2 != 0 || g.K < 2 * BK2 || g.M < 1) return false;
  if (g.epilogue == MAVLM_EPI_RES_F32) return false;
  if ((double)(g.M + 256) * g.lda * 2.0 >= 4.0e9 || (double)g.N * g.ldw * 2.0 >= 4.0e9) return false;   // 32-bit byte offsets
  return true;
}

hipError_t mavlm_launch_gemm256p(const mavlm_gemm_args& g, int dtype, hipStream_t s) {
  const bool h = dtype == MAVLM_F16;
  switch (g.epilogue) {
    case MAVLM_EPI_BIAS: return h ? launch256p<F16, MAVLM_EPI_BIAS>(g, s) : launch256p<BF16, MAVLM_EPI_BIAS>(g, s);
    case MAVLM_EPI_RELU: return h ? launch256p<F16, MAVLM_EPI_RELU>(g, s) : launch256p<BF16, MAVLM_EPI_RELU>(g, s);
    case MAVLM_EPI_GELU: return h ? launch256p<F16, MAVLM_EPI_GELU>(g, s) : launch256p<BF16, MAVLM_EPI_GELU>(g, s);
    case MAVLM_EPI_F32: return h ? launch256p<F16, MAVLM_EPI_F32>(g, s) : launch256p<BF16, MAVLM_EPI_F32>(g, s);
  }
  return hipErrorInvalidValue;
}
