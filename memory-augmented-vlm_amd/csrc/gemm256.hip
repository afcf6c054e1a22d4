// 256x256x64 "4-phase per K-tile" MFMA GEMM for the large projections of the memory path.
// Same contract as gemm_tn_kernel (gemm.hip): C[M,N] = epi(A[M,K] . W[N,K]^T + bias), nn.Linear layout.
//
// Structure (after cdna_hip_programming.md "The 256^2 8-phase template", re-derived for this layout):
//   * 8 waves (2 along M x 4 along N), wave tile 128x64 = 8x4 MFMA 16x16x32 tiles (128 accumulator VGPRs), one
//     workgroup per CU, 128 KiB LDS = 2 stages x {A0,A1,B0,B1} half-tiles of 128 rows x 64 k (16 KiB each).
//   * Operands arrive by LDS-DMA (16-byte global_load_lds), XOR-swizzled through the per-lane SOURCE address;
//     a half-tile is 2 DMA instructions per wave.
//   * A K-tile is consumed in 4 phases of 16 MFMAs (one 64x32 quadrant of the wave tile x K=64), every phase an L
//     segment (fragment reads + DMA issue) and an M segment (the MFMAs):
//        phase 1: read A(rows 0-63) + B(columns 0-31)   (12 ds_read_b128), quadrant (0,0)
//        phase 2: read B(columns 32-63) (4),  DMA A1 of K-tile kt+1 (other stage),  quadrant (0,1)
//        phase 3: read A(rows 64-127) (8),    DMA B0 of K-tile kt+2 (this stage),   quadrant (1,1)
//        phase 4: no reads,                   DMA B1 + A0 of K-tile kt+2, the counted wait,  quadrant (1,0)
//     so the B half-tiles of the stage are dead after phase 2 and the A half-tiles after phase 3.  (Round 1 read all of
//     B in phase 1 and issued one half-tile per phase: L segments of 16 / 0 / 8 / 0 reads + 1 DMA each, the first far
//     longer than the 256-clock M segment of the other wave group it should hide behind; 12 / 4 / 8 / 0 reads with
//     0 / 1 / 1 / 2 DMAs measured 2-4 % faster on every shape, tools/diag_gemm_ablate.sh.)  Loads stay in flight across
//     barriers; the only vmcnt wait is a COUNTED one per K-tile (vmcnt(6): the three youngest half-tiles stay in
//     flight), never 0 inside the loop.
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
//   reads of stage s: L1 (A rows 0-63, B columns 0-31), L2 (B columns 32-63), L3 (A rows 64-127): g = 8kt+1, +3, +5
//        (leading), 8kt+2, +4, +6 (trailing); every read is retired by the lgkmcnt(0) at the top of the following M
//        segment.
//   WAR  B half-tiles of s are dead from g = 8kt+5, A0 (read by the leading group only) from g = 8kt+7, A1 (trailing
//        group only) from g = 8kt+8.  Re-staging (K-tile kt+2) is issued by the leading group in L3 (g = 8kt+5: B0), L4
//        (8kt+7: B1, A0) and the next K-tile's L2 (8kt+11: A1).  (A DMA issued at barrier index g lands hundreds of
//        clocks later; the trailing group's last reads of a half-tile retire at the top of the same index.)
//   RAW  K-tile kt+1 is first read at g = 8kt+9.  Its last half-tile (A1) is issued in L2 of kt; both groups wait
//        vmcnt(6) in L4 (g = 8kt+7 / 8kt+8) - the six youngest DMAs are B0, B1, A0 of kt+2 - and pass a barrier before
//        g = 8kt+9.
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

constexpr int BM2 = 256, BN2 = 256, BK2 = 64;
constexpr int HALF = 128 * BK2 * 2;          // 16 KiB half-tile
constexpr int STAGE2 = 4 * HALF;             // A0 A1 B0 B1
constexpr int GEMM256_LDS = 2 * STAGE2;      // 128 KiB

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


__device__ __forceinline__ float gelu_erf2(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// MT1 = 16-row MFMA tiles in the SECOND 64-row slice of a wave's rows: 4 -> 256-row workgroup tile (wave tile 128x64),
// 3 -> 224-row tile (wave tile 112x64; phases 3 and 4 run 12 MFMAs).  224 = 7 * 32 divides R = M_tokens * 196 whenever
// M_tokens % 8 == 0: R = 12 544 is 56 x 224 but 49 x 256, so on 256 CUs a 224-row grid is 224 / 448 / 896 tiles where
// the 256-row grid is 196 / 392 / 784 - the same number of rounds at 7/8 of the work per round (mavlm_gemm_tile_rows).
// The LDS image keeps its 128-row half-tile slots (rows 112-127 of a slot are staged but never read), so the DMA
// instruction count - and with it every counted vmcnt - is the same for both heights.  Each output element sums its
// K products in the same order for both heights: results are bit-identical.
template <typename T, int EPI, int MT1>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const uint16_t* __restrict__ A, int lda,
                                                         const uint16_t* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias,
                                                         const uint16_t* __restrict__ res, int ldr,
                                                         void* __restrict__ Cout, int ldc, int M, int N, int K,
                                                         int c_rpb, int c_nb, long long c_bs, mavlm_ln_epilogue ln,
                                                         int ksplit) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // split-K (round 4; EPI_F32 only): blockIdx.y owns the K range [y ksplit, (y + 1) ksplit) and its own fp32 plane of the
  // output - pure partial products, bias and epilogue are applied once by splitk_reduce_kernel.  For the long contractions
  // over few 256-row tiles (the 4D -> D projection of a single video at the OneVision-7B width: 98 tiles x 224 K-tiles).
  if (ksplit > 0) {
    const int kz = (int)blockIdx.y * ksplit;
    A += kz;
    W += kz;
    Cout = (void*)((float*)Cout + (size_t)blockIdx.y * (size_t)M * ldc);
    K = (K - kz < ksplit) ? K - kz : ksplit;
  }
  constexpr int MHALF = 64 + 16 * MT1;        // rows per wave group (= valid rows of an A half-tile slot)
  constexpr int BMT = 2 * MHALF;              // workgroup tile height

  const int ntn = N / BN2;
  const int ntm = (M + BMT - 1) / BMT;
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (wg / ntn) * BMT;
  const int n0 = (wg % ntn) * BN2;

  // ---- LDS-DMA sources: wave w stages 8-row groups g = 2w, 2w+1 of every half-tile.  Buffer loads
  // (`buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`): one descriptor per operand over the rows of this tile, a
  // loop-invariant 32-bit lane offset per (half, instruction) and the K-tile as the scalar offset - a DMA is `s_mov m0`
  // + the load, no 64-bit vector address arithmetic per piece.  The A descriptor ends after row M-1: rows past M read
  // as zeros (their outputs are never stored).
  const int srow = lane >> 3, sp = lane & 7;
  int oA[2][2], oB[2][2];     // [half][inst] byte offsets from the tile's first row
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = (wave * 2 + j) * 8 + srow;          // row inside the half-tile
      const int c = sp ^ ((row >> 1) & 7);                // logical 16-B chunk stored at physical chunk sp
      oA[h][j] = ((h * MHALF + row) * lda + c * 8) * 2;   // (rows >= MHALF of a slot: staged, never read)
      oB[h][j] = ((h * 128 + row) * ldw + c * 8) * 2;
    }
  auto tile_rsrc = [&](const uint16_t* base, int row0, int rows, int ld) {
    const uintptr_t a = (uintptr_t)(base + (size_t)row0 * ld);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(rows - 1) * (uint32_t)ld * 2u + (uint32_t)K * 2u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const int arows = M - m0 < BMT ? M - m0 : BMT;
  const __amdgpu_buffer_rsrc_t rsA = tile_rsrc(A, m0, arows, lda), rsB = tile_rsrc(W, n0, BN2, ldw);
  const unsigned lds_wave = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem + wave * 2048;
  // half-tile ids: 0 = A0, 1 = A1, 2 = B0, 3 = B1
  auto dma = [&](int stage, int half_id, int kt) {
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));            // M0 = base + constant stays a one-instruction recompute (no SGPR hoisting)
    const unsigned dst = base + stage * STAGE2 + half_id * HALF;
    const int h = half_id & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(half_id < 2 ? rsA : rsB, (MAVLM_LDS void*)(uintptr_t)(dst + j * 1024), 16,
                                               half_id < 2 ? oA[h][j] : oB[h][j], kt * BK2 * 2, 0, 0);
  };

  // ---- fragment read offsets
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (lane >> 1) & 7;
  const int ck0 = (fq ^ sw) << 4, ck1 = ((4 + fq) ^ sw) << 4;
  const int offA = wm * HALF + fr * 128;                               // + mh*8192 + mt*2048
  const int offB = 2 * HALF + (wn >> 1) * HALF + ((wn & 1) * 64 + fr) * 128;   // + nh*4096 + nt*2048

  f32x4 acc[4 + MT1][4];
#pragma unroll
  for (int i = 0; i < 4 + MT1; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  typename T::vec8 af[4][2];      // [m-tile of the current 64-row slice][k-step]
  typename T::vec8 bf[4][2];      // [n-tile of the wave's 64 columns][k-step]

  const int nk = K / BK2;
  const bool trailing = wm == 1;     // waves 4-7 (wave-uniform: wm comes from a readfirstlane)

  // ---- prologue: K-tile 0 completely, K-tile 1 minus its last half-tile (order B0,B1,A0,A1)
  dma(0, 2, 0); dma(0, 3, 0); dma(0, 0, 0); dma(0, 1, 0);
  if (nk > 1) {
    dma(1, 2, 1); dma(1, 3, 1); dma(1, 0, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  MAVLM_BAR();

  auto read_a = [&](const char* st, int mh) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      if (mh == 1 && mt >= MT1) continue;     // (mh is a literal at every call site)
      af[mt][0] = *(const typename T::vec8*)(st + offA + mh * 8192 + mt * 2048 + ck0);
      af[mt][1] = *(const typename T::vec8*)(st + offA + mh * 8192 + mt * 2048 + ck1);
    }
  };
  auto read_b_half = [&](const char* st, int nh) {
#pragma unroll
    for (int nt = 2 * nh; nt < 2 * nh + 2; ++nt) {
      bf[nt][0] = *(const typename T::vec8*)(st + offB + nt * 2048 + ck0);
      bf[nt][1] = *(const typename T::vec8*)(st + offB + nt * 2048 + ck1);
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

  for (int kt = 0; kt < nk; ++kt) {
    const int s = kt & 1;
    const char* st = smem + s * STAGE2;
    // -------- phase 1
    read_a(st, 0);
    read_b_half(st, 0);
    MAVLM_BAR();
    MAVLM_LGKM0();
    MAVLM_QUADRANT(0, 0)
    MAVLM_BAR();
    // -------- phase 2
    read_b_half(st, 1);
    if (kt + 1 < nk) dma(s ^ 1, 1, kt + 1);              // A1 of the next K-tile (needed at its L1)
    MAVLM_BAR();
    MAVLM_LGKM0();
    MAVLM_QUADRANT(0, 1)
    MAVLM_BAR();
    // -------- phase 3   (B half-tiles of this stage are dead)
    read_a(st, 1);
    if (kt + 2 < nk) dma(s, 2, kt + 2);
    MAVLM_BAR();
    MAVLM_LGKM0();
    MAVLM_QUADRANT(1, 1)
    MAVLM_BAR();
    // -------- phase 4   (A half-tiles dead)
    if (kt + 2 < nk) {
      dma(s, 3, kt + 2);
      dma(s, 0, kt + 2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // K-tile kt+1 landed; 3 half-tiles of kt+2 stay in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    MAVLM_BAR();
    MAVLM_QUADRANT(1, 0)
    MAVLM_BAR();
  }
  if (!trailing) MAVLM_BAR();          // matches the trailing group's last barrier
#undef MAVLM_QUADRANT

  if constexpr (EPI == MAVLM_EPI_LN) {
    // ---- fused Residual epilogue (MemoryController.py:26-29): out = LayerNorm(acc + bias + res) * gamma + beta, 16-bit.
    // A row's N columns belong to the ntn workgroups of its row block; each computes the mean and the centred sum of squares
    // of ITS 256 columns (exact fp32, two passes over registers), the workgroups exchange those 2 floats per row through
    // global memory and merge them (Chan's parallel variance: exact, order fixed -> all workgroups get the same bits).
    // Exchange = the "data is the flag" granule form (cdna_hip_programming.md Guideline 16, R2): 8-byte {epoch, value}
    // granules stored and polled with agent-scope relaxed atomics (sc1: no fence, no separate flag).  The epoch comes
    // from a launch counter in memory that the LAST workgroup of a launch advances - nothing is re-zeroed between launches
    // and a hipGraph replay sees fresh epochs.  Spins are bounded (a timeout sets ln.ctl[2] and lets the launch drain).
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MAVLM_BAR();                                            // every wave is out of the K loop: the LDS stages are free
    float* red = (float*)smem;                              // [4 wave columns][256 rows]
    float* peer = (float*)(smem + 4096);                    // [ntn][256 rows][2]
    const int rb = wg / ntn, ct = wg - rb * ntn;
    const unsigned epoch = __hip_atomic_load(ln.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    constexpr int NI = 4 + MT1;
    // 1. v = acc + bias + residual (fp32), in place
    {
      f32x4 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(bias + n0 + wn * 64 + j * 16 + fq * 4);
      u32x2 rr[NI][4];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = m0 + wm * MHALF + i * 16 + fr;
        const int mc = m < M ? m : M - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          rr[i][j] = *(const u32x2*)(res + (size_t)mc * ldr + n0 + wn * 64 + j * 16 + fq * 4);
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = m0 + wm * MHALF + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 o = acc[i][j] + bv[j];
          if (ln.pre_out != nullptr && m < M)               // training: the dense output before the residual (for the backward)
            *(f32x4*)(ln.pre_out + (size_t)m * N + n0 + wn * 64 + j * 16 + fq * 4) = o;
          o[0] += T::to_f32((uint16_t)(rr[i][j][0] & 0xffffu)); o[1] += T::to_f32((uint16_t)(rr[i][j][0] >> 16));
          o[2] += T::to_f32((uint16_t)(rr[i][j][1] & 0xffffu)); o[3] += T::to_f32((uint16_t)(rr[i][j][1] >> 16));
          acc[i][j] = o;
        }
      }
    }
    // 2. mean and centred sum of squares over this workgroup's 256 columns, per row
    auto row_reduce = [&](float (&v)[NI]) {                 // in: per-lane partials; out: the 256-column sums, on every lane
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float s_ = v[i];
        s_ += __shfl_xor(s_, 16);
        s_ += __shfl_xor(s_, 32);
        if (fq == 0) red[wn * 256 + wm * MHALF + i * 16 + fr] = s_;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MAVLM_BAR();
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r_ = wm * MHALF + i * 16 + fr;
        v[i] = ((red[r_] + red[256 + r_]) + red[512 + r_]) + red[768 + r_];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MAVLM_BAR();                                          // (red is reused)
    };
    float mk[NI], qk[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float s_ = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s_ += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
      mk[i] = s_;
    }
    row_reduce(mk);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      mk[i] *= (1.0f / 256.0f);
      float s_ = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d_ = acc[i][j][e] - mk[i];
          s_ += d_ * d_;
        }
      qk[i] = s_;
    }
    row_reduce(qk);
    // 3. publish (mean, M2) of every row of the tile - also of rows past M: the peers poll all of them
    unsigned long long* gran = ln.gran + ((size_t)rb * ntn) * (256 * 2);     // [ntn][256][2] granules of this row block
    if (fq == 0 && wn == 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r_ = wm * MHALF + i * 16 + fr;
        unsigned long long* g_ = gran + ((size_t)ct * 256 + r_) * 2;
        __hip_atomic_store(g_, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, mk[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g_ + 1, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, qk[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        peer[(ct * 256 + r_) * 2] = mk[i];
        peer[(ct * 256 + r_) * 2 + 1] = qk[i];
      }
    }
    // 4. collect the peers' granules into LDS: thread t polls granules t, t + 512, ... of the other ntn - 1 tiles
    {
      const int per = BMT * 2, total = (ntn - 1) * per;
      for (int g0 = tid; g0 < total; g0 += 512) {
        int k_ = g0 / per;
        const int w_ = g0 - k_ * per;                       // row * 2 + {mean, M2}
        k_ += k_ >= ct;                                     // skip our own tile
        const unsigned long long* g_ = gran + (size_t)k_ * 512 + w_;
        unsigned long long x_ = 0;
        unsigned spins = 0;
        for (;;) {
          x_ = __hip_atomic_load(g_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(x_ >> 32) == epoch) break;
          if (++spins > (1u << 22)) {                       // bounded: never hang the chip on a lost partner
            __hip_atomic_store(ln.ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        peer[k_ * 512 + w_] = __builtin_bit_cast(float, (unsigned)x_);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MAVLM_BAR();
    // 5. merge in tile order (every workgroup of the row block adds the same numbers in the same order), normalise, store
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r_ = wm * MHALF + i * 16 + fr;
      float msum = 0.f;
      for (int k_ = 0; k_ < ntn; ++k_) msum += peer[(k_ * 256 + r_) * 2];
      const float mean = msum / (float)ntn;
      float m2 = 0.f;
      for (int k_ = 0; k_ < ntn; ++k_) {
        const float d_ = peer[(k_ * 256 + r_) * 2] - mean;
        m2 += peer[(k_ * 256 + r_) * 2 + 1] + 256.0f * d_ * d_;
      }
      const float rstd = rsqrtf(m2 * inv_n + ln.eps);
      const int m = m0 + wm * MHALF + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const int nx = n0 + wn * 64 + j * 16 + fq * 4, ny = nx + 16;
        const f32x4 gx = *(const f32x4*)(ln.gamma + nx), gy = *(const f32x4*)(ln.gamma + ny);
        const f32x4 bx = *(const f32x4*)(ln.beta + nx), by = *(const f32x4*)(ln.beta + ny);
        const f32x4 x = acc[i][j], y = acc[i][j + 1];
        const u32x4 w = widen_pair(pack4<T>((x[0] - mean) * rstd * gx[0] + bx[0], (x[1] - mean) * rstd * gx[1] + bx[1],
                                            (x[2] - mean) * rstd * gx[2] + bx[2], (x[3] - mean) * rstd * gx[3] + bx[3]),
                                   pack4<T>((y[0] - mean) * rstd * gy[0] + by[0], (y[1] - mean) * rstd * gy[1] + by[1],
                                            (y[2] - mean) * rstd * gy[2] + by[2], (y[3] - mean) * rstd * gy[3] + by[3]));
        const int n = n0 + wn * 64 + 16 * (j + (fq & 1)) + 8 * (fq >> 1);
        if (m < M) *(u32x4*)((uint16_t*)Cout + (size_t)m * ldc + n) = w;
      }
    }
    // 6. the last workgroup of the launch to get here advances the launch counter (the next launch's epoch) and clears
    // the arrival counter; both are read / written again only by later launches of the same stream
    MAVLM_BAR();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(ln.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(ntm * ntn) - 1u) {
        __hip_atomic_store(ln.ctl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ln.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  // ---- epilogue: lane holds C[m][n..n+3], m = m0 + wm*MHALF + 16 i + fr, n = n0 + wn*64 + 16 j + 4 fq.
  // 16-bit outputs: column blocks (j, j+1) are exchanged between lane groups (widen_pair) so that every lane stores
  // 16 contiguous bytes - half the store instructions, 64-byte instead of 32-byte row segments.
  auto act4 = [&](f32x4 v) -> f32x4 {
    if (EPI == MAVLM_EPI_RELU) return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    if (EPI == MAVLM_EPI_GELU) return gelu_erf_fast4(v);      // packed fp32 math (mavlm_common.h)
    return v;
  };
  // element offset of output row m.  Row-batched outputs (mavlm_gemm_args::c_rpb): block q = m / c_rpb of c_rpb rows goes
  // to batch element q % c_nb, as its (q / c_nb)-th block
  auto crow = [&](int m) -> size_t {
    if (c_rpb <= 0) return (size_t)m * ldc;
    const int q = m / c_rpb, r = m - q * c_rpb;
    return (size_t)(q % c_nb) * (size_t)c_bs + ((size_t)(q / c_nb) * c_rpb + r) * ldc;
  };
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bv[j] = ksplit > 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(bias + n0 + wn * 64 + j * 16 + fq * 4);
#pragma unroll
  for (int i = 0; i < 4 + MT1; ++i) {
    const int m = m0 + wm * MHALF + i * 16 + fr;
    const size_t co = crow(m < M ? m : M - 1);
    if (EPI == MAVLM_EPI_RES_F32 || EPI == MAVLM_EPI_F32) {
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fq * 4;
        f32x4 o = acc[i][j] + bv[j];
        if (EPI == MAVLM_EPI_RES_F32) {
          const u16x4 rv = *(const u16x4*)(res + (size_t)m * ldr + n);
          o[0] += T::to_f32(rv[0]); o[1] += T::to_f32(rv[1]); o[2] += T::to_f32(rv[2]); o[3] += T::to_f32(rv[3]);
        }
        *(f32x4*)((float*)Cout + co + n) = o;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j += 2) {                     // (all lanes take part in the swaps; the store is masked)
        const f32x4 x = acc[i][j] + bv[j], y = acc[i][j + 1] + bv[j + 1];
        const f32x4 xa = act4(x), ya = act4(y);
        const u32x4 w = widen_pair(pack4<T>(xa[0], xa[1], xa[2], xa[3]), pack4<T>(ya[0], ya[1], ya[2], ya[3]));
        const int n = n0 + wn * 64 + 16 * (j + (fq & 1)) + 8 * (fq >> 1);
        if (m < M) *(u32x4*)((uint16_t*)Cout + co + n) = w;
      }
    }
  }
}

template <typename T, int EPI, int MT1>
hipError_t launch256h(const mavlm_gemm_args& g, hipStream_t s, int splits = 1, int ksplit = 0) {
  auto kern = gemm256_kernel<T, EPI, MT1>;
  constexpr int BMT = 2 * (64 + 16 * MT1);
  static mavlm_per_device_once once;
  {
    hipError_t e = once.dyn_lds((const void*)kern, GEMM256_LDS);
    if (e != hipSuccess) return e;
  }
  const int ntm = (g.M + BMT - 1) / BMT, ntn = g.N / BN2;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, splits), dim3(512), GEMM256_LDS, s, (const uint16_t*)g.A, g.lda, (const uint16_t*)g.W,
                     g.ldw, g.bias, (const uint16_t*)g.res, g.ldr, g.C, g.ldc, g.M, g.N, g.K, g.c_rpb, g.c_nb > 0 ? g.c_nb : 1,
                     (long long)g.c_bstride, g.ln, ksplit);
  return hipGetLastError();
}

template <typename T, int EPI>
hipError_t launch256(const mavlm_gemm_args& g, hipStream_t s) {
  return mavlm_gemm_tile_rows(g.M, g.N) == 224 ? launch256h<T, EPI, 3>(g, s) : launch256h<T, EPI, 4>(g, s);
}

template <typename T>
hipError_t launch256_epi(const mavlm_gemm_args& g, hipStream_t s) {
  switch (g.epilogue) {
    case MAVLM_EPI_BIAS: return launch256<T, MAVLM_EPI_BIAS>(g, s);
    case MAVLM_EPI_RELU: return launch256<T, MAVLM_EPI_RELU>(g, s);
    case MAVLM_EPI_GELU: return launch256<T, MAVLM_EPI_GELU>(g, s);
    case MAVLM_EPI_RES_F32: return launch256<T, MAVLM_EPI_RES_F32>(g, s);
    case MAVLM_EPI_F32: return launch256<T, MAVLM_EPI_F32>(g, s);
    case MAVLM_EPI_LN: return launch256<T, MAVLM_EPI_LN>(g, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

int g_mavlm_gemm_ln_wide = 0;     // test hook (mavlm_set_fused_layernorm(2)): admit up to 16 partners per row block
// Fused dense + residual + LayerNorm epilogue (EPI_LN): the N / 256 workgroups of a row block exchange 2 floats per row
// through `gran`.  Scratch: 64 bytes of control words (arrivals, launch counter, timeout flag; FIRST, so that their place does
// not depend on the shape: a scratch reused for another shape keeps counting epochs upward) + [ceil(M / rows)][N / 256][256][2]
// 8-byte granules - zero-filled ONCE before the first launch, then owned by the launches of ONE stream.
bool mavlm_gemm_ln_supported(int M, int N, int K, int wide) {
  if (wide < 0) wide = g_mavlm_gemm_ln_wide;
  // N <= 1024: up to 4 partners per row block.  The kernel is correct for up to 16 (N = 3584 is tested through
  // mavlm_set_fused_layernorm(2)), but with 14 partners every thread polls 11 granules one after the other and the row
  // blocks of a grid larger than the chip straddle its rounds (workgroups spin on CUs their partners are waiting for):
  // measured 730 TFLOP/s against 1 290 for the plain GEMM at the OneVision-7B width - the two-kernel form wins there.
  if (M <= 0 || N % BN2 != 0 || N / BN2 > (wide ? 16 : 4) || N / BN2 < 1 || K % BK2 != 0 || K <= 0) return false;
  // (the same "fills the chip" rule as the plain 256-column-tile kernels: below it the 128-tile / split-K kernels + the
  // row LayerNorm kernel are faster)
  return (long)((M + 255) / 256) * (N / BN2) >= 192;
}
size_t mavlm_gemm_ln_ws_bytes(int M, int N) {
  const size_t rbs = (size_t)((M + 223) / 224);             // (224-row tiles give the most row blocks)
  return rbs * (size_t)(N / BN2) * 512 * 8 + 64;
}

// (operand tiles are addressed through 32-bit buffer offsets: 256 rows x leading dimension must stay below 2 GiB)
bool mavlm_gemm256_supported(const mavlm_gemm_args& g) {
  return g.N % BN2 == 0 && g.K % BK2 == 0 && g.M >= 1 && (double)g.lda * 512.0 < 2.0e9 && (double)g.ldw * 512.0 < 2.0e9;
}

// Height of the workgroup tile (256 or 224 rows) for an M x N output on 256 CUs at one workgroup per CU: the height
// whose grid costs fewer row-rounds, rounds x height (ties -> 256).  Pure function of the shape; the result of the GEMM
// does not depend on it.
int g_mavlm_gemm_rows = 0;      // diagnostics: 0 = automatic, 224 / 256 = forced
int mavlm_gemm_tile_rows(int M, int N) {
  if (g_mavlm_gemm_rows == 224 || g_mavlm_gemm_rows == 256) return g_mavlm_gemm_rows;
  const long ntn = N / BN2;
  const long t256 = (long)((M + 255) / 256) * ntn, t224 = (long)((M + 223) / 224) * ntn;
  const long c256 = ((t256 + 255) / 256) * 256, c224 = ((t224 + 255) / 256) * 224;
  return c224 < c256 ? 224 : 256;
}

// split-K form: g.C = [splits][M][N] fp32 planes (ldc == N), g.epilogue ignored (EPI_F32, no bias); 256-row tiles
hipError_t mavlm_launch_gemm256_splitk(const mavlm_gemm_args& g, int splits, int ksplit, int dtype, hipStream_t s) {
  if (splits < 1 || ksplit <= 0 || ksplit % BK2 != 0 || g.ldc != g.N || (double)g.M * g.N * 4.0 * splits >= 1.7e10) return hipErrorInvalidValue;
  return dtype == MAVLM_F16 ? launch256h<F16, MAVLM_EPI_F32, 4>(g, s, splits, ksplit) : launch256h<BF16, MAVLM_EPI_F32, 4>(g, s, splits, ksplit);
}

hipError_t mavlm_launch_gemm256(const mavlm_gemm_args& g, int dtype, hipStream_t s) {
  return dtype == MAVLM_F16 ? launch256_epi<F16>(g, s) : launch256_epi<BF16>(g, s);
}
