// 128x256x64 MFMA GEMM with TWO INDEPENDENT WORKGROUPS PER CU (round 4).
// Same contract as gemm_tn_kernel (gemm.hip): C[M,N] = epi(A[M,K] . W[N,K]^T + bias), nn.Linear layout.
//
// Why it exists, and what it turned out to be good for.  The 256x256 kernels (gemm256.hip, gemm256p.hip)
// run one 8-wave workgroup per CU and whole grids in lock-step: prologues and epilogues (bias / GELU / the fused LayerNorm
// exchange / stores) are exposed and a grid of 1.53 or 1.75 rounds of tiles costs 2.  Here a workgroup is 4 waves on a
// 128x256 tile at 80 KiB of LDS and <= 256 registers: two of them share a CU (one wave of each per SIMD), independent of
// each other, so one's K loop runs under the other's epilogue and the grid is cut twice as fine.  Both effects are real
// (the GELU launch costs +8 % over ReLU here against +18 % on the 256-row kernel) - but two 128x256 tiles stage 1.5 x the
// operand bytes of one 256x256 tile per flop, and L2 -> LDS staging (~12.8 TB/s chip-wide under MFMA load: 48-50 GB/s
// per CU in every GEMM loop measured in this repository, the vendor library's included) is what bounds these loops: this
// kernel tops out at ~1 120-1 165 TFLOP/s in its K loop where the 256-row one reaches ~1 430.  It is therefore the
// automatic choice only for mid-size grids (gemm.hip: use_128x256) and a tuning hook (mavlm_set_gemm_tile(129))
// elsewhere; its results are bit-identical to the other kernels' (same K order per output element).
//
// Structure:
//   * 4 waves side by side along N: wave w owns all 128 rows x columns [64w, 64w+64) = 8x4 MFMA 16x16x32 tiles
//     (128 accumulator registers).  A K-tile (64 deep) is 4 phases of 16 MFMAs in Gray order over (row half, column half):
//     even K-tiles (lo,c0) (lo,c1) (hi,c1) (hi,c0), odd ones (lo,c1) (lo,c0) (hi,c0) (hi,c1) - one operand changes per
//     phase, and the operand a phase needs NEXT always goes into the register set that died in the phase before:
//         P1: read second B half of kt        P2: read A_hi(kt)        P3: read A_lo(kt+1)        P4: read first B half of kt+1
//     every fragment read is issued one phase ahead of its MFMAs (in-wave software pipeline; the wave waits
//     lgkmcnt(0) only AFTER its 16 MFMAs).  96 fragment registers: alo, ahi, two B halves.
//   * The B operand of a wave (its 64 weight rows) is PRIVATE to the wave: it stages it itself (LDS-DMA) into its own
//     3-slot ring of 4 KiB column halves and needs no barrier for it - only its own counted vmcnt.  The A operand (128
//     rows) is shared: {A_lo, A_hi} of two K-tiles = 4 slots of 8 KiB, each wave stages a quarter of every piece.
//   * ONE barrier per K-tile (64 MFMAs per wave), at the top of P2: before it every wave has waited for its parts of
//     A_hi(kt) and A_lo(kt+1) (RAW: both are read behind it) and has retired its reads of A_hi(kt-1) and A_lo(kt) (WAR:
//     their slots are re-staged right behind it with A_hi(kt+1), A_lo(kt+2) - a whole K-tile of lead).  B slots: the
//     wave's own reads are retired (end of the phase that issued them) before its own DMA re-stages the slot.
//   * DMA issue per K-tile and wave: P1 4 (second half of kt+1), P2 2 + 2 (A), P3 4 (first half of kt+2); three counted
//     waits, never vmcnt(0) inside the loop: P1 vmcnt(8), P2 vmcnt(8), P4 vmcnt(12) (the counts of the last two K-tiles
//     follow what is really behind them, see ktile()).  8-12 KiB per wave in flight.
//   * Operand images: rows of 128 B, XOR-swizzled through the per-lane SOURCE address (chunk c of row r is stored at chunk
//     c ^ ((r >> 1) & 7)), conflict-free ds_read_b128 fragments; buffer-load DMA: one descriptor per operand, two lane
//     offsets per operand (odd / even 8-row group), row group + K-tile in the scalar offset.
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

constexpr int BM1 = 128, BN1 = 256, BK1 = 64;
constexpr int ASUB = 64 * BK1 * 2;                 // 8 KiB: 64 rows x 128 B
constexpr int BSUB = 32 * BK1 * 2;                 // 4 KiB: 32 weight rows x 128 B
constexpr int A_RING = 4 * ASUB;                   // 32 KiB: {A_lo, A_hi} of two K-tiles
constexpr int B_RING = 3 * BSUB;                   // 12 KiB per wave
constexpr int GEMM128_LDS = A_RING + 4 * B_RING;   // 80 KiB -> two workgroups per CU (160 KiB)

template <int V>
struct IC1 { static constexpr int value = V; };

#define MAVLM_BAR1()                         \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_barrier();            \
    asm volatile("" ::: "memory");           \
  } while (0)
// lgkmcnt(0) through the BUILTIN (imm 0xC07F: vmcnt 63, expcnt 7, lgkmcnt 0): hipcc's own wait-count pass sees it.  With an
// inline-asm wait the pass still believes the fragment reads of the previous phase outstanding and, across the loop's back
// edge, puts an lgkmcnt(0) in front of the next phase's MFMAs - i.e. waits for the reads that phase has just issued for
// a LATER phase (measured in the first build of this loop: every K-tile stalled on its own prefetch).
#define MAVLM_LGKM0_1()                                        \
  do {                                                         \
    __builtin_amdgcn_s_waitcnt(0xC07F);                        \
    __builtin_amdgcn_sched_barrier(0);                         \
  } while (0)

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(const uint16_t* __restrict__ A, int lda,
                                                         const uint16_t* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias,
                                                         const uint16_t* __restrict__ res, int ldr,
                                                         void* __restrict__ Cout, int ldc, int M, int N, int K,
                                                         int c_rpb, int c_nb, long long c_bs, mavlm_ln_epilogue ln) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int ntn = N / BN1;
  const int ntm = (M + BM1 - 1) / BM1;
  const int wg = xcd_remap(blockIdx.x, ntm * ntn);
  const int m0 = (wg / ntn) * BM1;
  const int n0 = (wg % ntn) * BN1;

  // ---- LDS-DMA sources.  One wave instruction stages 8 rows x 128 B (1 KiB, lane-linear in LDS); lane -> (row srow,
  // physical chunk sp) fetches logical chunk sp ^ ((row_in_subpiece >> 1) & 7).  row_in_subpiece = 8 g + srow, so the
  // swizzle depends on the parity of the row group g only: two lane offsets per operand.
  const int srow = lane >> 3, sp = lane & 7;
  int voA[2], voB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = sp ^ ((4 * j + (srow >> 1)) & 7);
    voA[j] = (srow * lda + c * 8) * 2;
    voB[j] = (srow * ldw + c * 8) * 2;
  }
  auto tile_rsrc = [&](const uint16_t* base, int row0, int rows, int ld) {
    const uintptr_t a = (uintptr_t)(base + (size_t)row0 * ld);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(rows - 1) * (uint32_t)ld * 2u + (uint32_t)K * 2u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const int arows = M - m0 < BM1 ? M - m0 : BM1;
  // the A descriptor ends after row M-1 (rows past M read as zeros, never stored); the B descriptor covers this wave's
  // 64 weight rows
  const __amdgpu_buffer_rsrc_t rsA = tile_rsrc(A, m0, arows, lda), rsB = tile_rsrc(W, n0 + wave * 64, 64, ldw);
  const unsigned lds0 = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem;
  const unsigned ldsA_w = lds0 + wave * 2048;                       // this wave's 2 KiB of every A sub-piece
  const unsigned ldsB_w = lds0 + A_RING + wave * B_RING;            // this wave's B ring
  // A sub-piece `sub` (0 = rows 0-63, 1 = rows 64-127) of K-tile kt -> ring slot; this wave stages row groups 2w, 2w+1
  auto dma_a = [&](int slot, int sub, int kt) {
    unsigned base = ldsA_w;
    asm volatile("" : "+s"(base));
    const unsigned dst = base + slot * ASUB;
    const int so = ((sub * 64 + wave * 16) * lda + kt * BK1) * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (MAVLM_LDS void*)(uintptr_t)(dst + j * 1024), 16, voA[j],
                                               so + j * 16 * lda, 0, 0);
  };
  // B sub-piece q (0 = this wave's columns 0-31, 1 = columns 32-63) of K-tile kt -> slot of the wave's own ring
  auto dma_b = [&](int slot, int q, int kt) {
    unsigned base = ldsB_w;
    asm volatile("" : "+s"(base));
    const unsigned dst = base + slot * BSUB;
    const int so = (q * 32 * ldw + kt * BK1) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (MAVLM_LDS void*)(uintptr_t)(dst + j * 1024), 16, voB[j & 1],
                                               so + j * 16 * ldw, 0, 0);
  };

  // ---- fragment read offsets (16x16x32: lane (fr, fq) holds row fr, k = 8 fq .. 8 fq + 7 of a 32-deep step)
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (lane >> 1) & 7;
  const int ck0 = (fq ^ sw) << 4, ck1 = ((4 + fq) ^ sw) << 4;
  const int offA = fr * 128;                                         // + slot*ASUB + mt*2048 + ck
  const int offB = A_RING + wave * B_RING + fr * 128;                // + slot*BSUB + t*2048 + ck

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  typename T::vec8 alo[4][2], ahi[4][2];      // [m-tile of the 64-row half][k-step]
  typename T::vec8 bf[4][2];                  // [n-tile of the wave's 64 columns][k-step]

  const int nk = K / BK1;

  auto read_a = [&](typename T::vec8 (&af)[4][2], int slot) {
    const char* st = smem + slot * ASUB + offA;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      af[mt][0] = *(const typename T::vec8*)(st + mt * 2048 + ck0);
      af[mt][1] = *(const typename T::vec8*)(st + mt * 2048 + ck1);
    }
  };
  auto read_b = [&](int q, int slot) {
    const char* st = smem + slot * BSUB + offB;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf[2 * q + t][0] = *(const typename T::vec8*)(st + t * 2048 + ck0);
      bf[2 * q + t][1] = *(const typename T::vec8*)(st + t * 2048 + ck1);
    }
  };
#define MAVLM_QUAD(AF, MH, NH)                                                              \
  {                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                        \
    _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                        \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
      acc[MH * 4 + mt][NH * 2 + nt] = T::mfma16(bf[NH * 2 + nt][ks], AF[mt][ks], acc[MH * 4 + mt][NH * 2 + nt]); \
    __builtin_amdgcn_s_setprio(0);                                                          \
  }

  // ---- software pipeline (see the header).  Register sets: alo / ahi = A rows 0-63 / 64-127, bf[0..1] = this wave's
  // columns 0-31, bf[2..3] = columns 32-63 - always; an even K-tile walks (lo,c0) (lo,c1) (hi,c1) (hi,c0), an odd one
  // (lo,c1) (lo,c0) (hi,c0) (hi,c1), so that the operand a phase loads for a later phase is always a dead register set.
  // B pieces in CONSUMPTION order: first(kt) = column half (kt & 1), second(kt) = the other; ring index 2 kt + {0, 1}.
  auto bq_first = [&](int kt) { return kt & 1; };
  // wave-uniform waits with literal counts
#define MAVLM_VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // prologue: first(0) A_lo(0) | second(0) | A_hi(0) A_lo(1) | first(1)   (the issue order of the steady state)
  dma_b(0, 0, 0);                               // idx 0 -> slot 0: columns 0-31 of K-tile 0
  dma_a(0, 0, 0);                               // A_lo(0) -> A slot 0
  dma_b(1, 1, 0);                               // idx 1 -> slot 1: columns 32-63 of K-tile 0
  dma_a(1, 1, 0);                               // A_hi(0) -> A slot 1
  if (nk > 1) {
    dma_a(2, 0, 1);                             // A_lo(1) -> A slot 2
    dma_b(2, 1, 1);                             // idx 2 = first(1) -> slot 2: columns 32-63 of K-tile 1
    MAVLM_VMCNT(12);                            // first(0), A_lo(0) landed (behind them: 4 + 2 + 2 + 4)
  } else {
    MAVLM_VMCNT(6);
  }
  MAVLM_BAR1();
  read_a(alo, 0);
  read_b(0, 0);
  MAVLM_LGKM0_1();

  // one K-tile.  PAR = kt & 1 (compile time): which column half goes first.  sb = ring slot of idx 2 kt (first(kt)).
  auto ktile = [&](auto par, int kt, int sb) {
    constexpr int PAR = decltype(par)::value;
    constexpr int C0 = PAR, C1 = PAR ^ 1;                   // column half of P1 / P4, and of P2 / P3
    const int sb1 = sb == 2 ? 0 : sb + 1;                   // slot of idx 2 kt + 1 (second(kt))
    const int sb2 = sb1 == 2 ? 0 : sb1 + 1;                 // slot of idx 2 kt + 2 (first(kt+1))
    const int sa = 2 * PAR;                                 // A slots of this K-tile: sa (lo), sa + 1 (hi); the other pair: kt +- 1
    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
    // -------- P1: (lo, C0).  second(kt) landed (issued in P1 of kt-1; behind it A_hi(kt) [+ A_lo(kt+1), first(kt+1)])
    if (n1) MAVLM_VMCNT(8);
    else MAVLM_VMCNT(2);
    if (n1) dma_b(sb, C0, kt + 1);                          // second(kt+1) = column half C0 -> slot of first(kt) (its reads retired)
    read_b(C1, sb1);
    if constexpr (C0 == 0) MAVLM_QUAD(alo, 0, 0) else MAVLM_QUAD(alo, 0, 1)
    MAVLM_LGKM0_1();
    // -------- P2: (lo, C1).  A_hi(kt), A_lo(kt+1) landed for every wave; all reads of A_hi(kt-1), A_lo(kt) retired
    if (n1) MAVLM_VMCNT(8);
    else MAVLM_VMCNT(0);
    MAVLM_BAR1();
    if (n1) dma_a(3 - sa, 1, kt + 1);                       // A_hi(kt+1) -> slot of A_hi(kt-1)
    if (n2) dma_a(sa, 0, kt + 2);                           // A_lo(kt+2) -> slot of A_lo(kt)
    read_a(ahi, sa + 1);
    if constexpr (C1 == 0) MAVLM_QUAD(alo, 0, 0) else MAVLM_QUAD(alo, 0, 1)
    MAVLM_LGKM0_1();
    // -------- P3: (hi, C1)
    if (n2) dma_b(sb1, C0, kt + 2);                         // first(kt+2) = column half C0 -> slot of second(kt) (reads retired in P1)
    if (n1) read_a(alo, 2 - sa);                            // A_lo(kt+1)
    if constexpr (C1 == 0) MAVLM_QUAD(ahi, 1, 0) else MAVLM_QUAD(ahi, 1, 1)
    MAVLM_LGKM0_1();
    // -------- P4: (hi, C0).  first(kt+1) landed (issued in P3 of kt-1; behind it: P1 4, P2 2 [+ 2], P3 [4] of this K-tile)
    if (n1) {
      if (n2) MAVLM_VMCNT(12);
      else MAVLM_VMCNT(6);
      read_b(C1, sb2);                                      // first(kt+1) = column half C1 (dead since P3)
    }
    if constexpr (C0 == 0) MAVLM_QUAD(ahi, 1, 0) else MAVLM_QUAD(ahi, 1, 1)
    MAVLM_LGKM0_1();
  };
  {
    int sb = 0;                                             // slot of idx 2 kt; + 2 per K-tile (mod 3)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
      ktile(IC1<0>{}, kt, sb);
      sb = sb == 0 ? 2 : sb - 1;
      ktile(IC1<1>{}, kt + 1, sb);
      sb = sb == 0 ? 2 : sb - 1;
    }
    if (kt < nk) ktile(IC1<0>{}, kt, sb);
  }
#undef MAVLM_VMCNT
#undef MAVLM_QUAD

  const int wn = wave;
  if constexpr (EPI == MAVLM_EPI_LN) {
    // ---- fused Residual epilogue: see gemm256_kernel (same protocol; a row block is 128 rows here, the partners are the
    // N / 256 workgroups of the row block, each with 4 wave columns)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    MAVLM_BAR1();
    float* red = (float*)smem;                              // [4 wave columns][128 rows]
    float* peer = (float*)(smem + 2048);                    // [ntn][128 rows][2]
    const int rb = wg / ntn, ct = wg - rb * ntn;
    const unsigned epoch = __hip_atomic_load(ln.ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    constexpr int NI = 8;
    {
      f32x4 bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(bias + n0 + wn * 64 + j * 16 + fq * 4);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int m = m0 + i * 16 + fr;
        const int mc = m < M ? m : M - 1;
        u32x2 rr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rr[j] = *(const u32x2*)(res + (size_t)mc * ldr + n0 + wn * 64 + j * 16 + fq * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f32x4 o = acc[i][j] + bv[j];
          if (ln.pre_out != nullptr && m < M)
            *(f32x4*)(ln.pre_out + (size_t)m * N + n0 + wn * 64 + j * 16 + fq * 4) = o;
          o[0] += T::to_f32((uint16_t)(rr[j][0] & 0xffffu)); o[1] += T::to_f32((uint16_t)(rr[j][0] >> 16));
          o[2] += T::to_f32((uint16_t)(rr[j][1] & 0xffffu)); o[3] += T::to_f32((uint16_t)(rr[j][1] >> 16));
          acc[i][j] = o;
        }
      }
    }
    auto row_reduce = [&](float (&v)[NI]) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        float s_ = v[i];
        s_ += __shfl_xor(s_, 16);
        s_ += __shfl_xor(s_, 32);
        if (fq == 0) red[wn * 128 + i * 16 + fr] = s_;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MAVLM_BAR1();
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r_ = i * 16 + fr;
        v[i] = ((red[r_] + red[128 + r_]) + red[256 + r_]) + red[384 + r_];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MAVLM_BAR1();
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
    unsigned long long* gran = ln.gran + ((size_t)rb * ntn) * (BM1 * 2);      // [ntn][128][2] granules of this row block
    if (fq == 0 && wn == 0) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int r_ = i * 16 + fr;
        unsigned long long* g_ = gran + ((size_t)ct * BM1 + r_) * 2;
        __hip_atomic_store(g_, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, mk[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(g_ + 1, ((unsigned long long)epoch << 32) | __builtin_bit_cast(unsigned, qk[i]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        peer[(ct * BM1 + r_) * 2] = mk[i];
        peer[(ct * BM1 + r_) * 2 + 1] = qk[i];
      }
    }
    {
      const int per = BM1 * 2, total = (ntn - 1) * per;
      for (int g0 = tid; g0 < total; g0 += 256) {
        int k_ = g0 / per;
        const int w_ = g0 - k_ * per;
        k_ += k_ >= ct;
        const unsigned long long* g_ = gran + (size_t)k_ * per + w_;
        unsigned long long x_ = 0;
        unsigned spins = 0;
        for (;;) {
          x_ = __hip_atomic_load(g_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(x_ >> 32) == epoch) break;
          if (++spins > (1u << 22)) {
            __hip_atomic_store(ln.ctl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        peer[k_ * per + w_] = __builtin_bit_cast(float, (unsigned)x_);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MAVLM_BAR1();
    const float inv_n = 1.0f / (float)N;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r_ = i * 16 + fr;
      float msum = 0.f;
      for (int k_ = 0; k_ < ntn; ++k_) msum += peer[(k_ * BM1 + r_) * 2];
      const float mean = msum / (float)ntn;
      float m2 = 0.f;
      for (int k_ = 0; k_ < ntn; ++k_) {
        const float d_ = peer[(k_ * BM1 + r_) * 2] - mean;
        m2 += peer[(k_ * BM1 + r_) * 2 + 1] + 256.0f * d_ * d_;
      }
      const float rstd = rsqrtf(m2 * inv_n + ln.eps);
      const int m = m0 + i * 16 + fr;
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
    MAVLM_BAR1();
    if (tid == 0) {
      const unsigned old = __hip_atomic_fetch_add(ln.ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (unsigned)(ntm * ntn) - 1u) {
        __hip_atomic_store(ln.ctl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(ln.ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  // ---- epilogue: lane holds C[m][n..n+3], m = m0 + 16 i + fr, n = n0 + wn*64 + 16 j + 4 fq (see gemm256_kernel)
  auto act4 = [&](f32x4 v) -> f32x4 {
    if (EPI == MAVLM_EPI_RELU) return f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    if (EPI == MAVLM_EPI_GELU) return gelu_erf_fast4(v);      // packed fp32 math (mavlm_common.h)
    return v;
  };
  auto crow = [&](int m) -> size_t {
    if (c_rpb <= 0) return (size_t)m * ldc;
    const int q = m / c_rpb, r = m - q * c_rpb;
    return (size_t)(q % c_nb) * (size_t)c_bs + ((size_t)(q / c_nb) * c_rpb + r) * ldc;
  };
  f32x4 bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bv[j] = *(const f32x4*)(bias + n0 + wn * 64 + j * 16 + fq * 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + i * 16 + fr;
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
      for (int j = 0; j < 4; j += 2) {
        const f32x4 x = acc[i][j] + bv[j], y = acc[i][j + 1] + bv[j + 1];
        const f32x4 xa = act4(x), ya = act4(y);
        const u32x4 w = widen_pair(pack4<T>(xa[0], xa[1], xa[2], xa[3]), pack4<T>(ya[0], ya[1], ya[2], ya[3]));
        const int n = n0 + wn * 64 + 16 * (j + (fq & 1)) + 8 * (fq >> 1);
        if (m < M) *(u32x4*)((uint16_t*)Cout + co + n) = w;
      }
    }
  }
}

template <typename T, int EPI>
hipError_t launch128(const mavlm_gemm_args& g, hipStream_t s) {
  auto kern = gemm128_kernel<T, EPI>;
  static mavlm_per_device_once once;
  {
    hipError_t e = once.dyn_lds((const void*)kern, GEMM128_LDS);
    if (e != hipSuccess) return e;
  }
  const int ntm = (g.M + BM1 - 1) / BM1, ntn = g.N / BN1;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(256), GEMM128_LDS, s, (const uint16_t*)g.A, g.lda, (const uint16_t*)g.W,
                     g.ldw, g.bias, (const uint16_t*)g.res, g.ldr, g.C, g.ldc, g.M, g.N, g.K, g.c_rpb, g.c_nb > 0 ? g.c_nb : 1,
                     (long long)g.c_bstride, g.ln);
  return hipGetLastError();
}

template <typename T>
hipError_t launch128_epi(const mavlm_gemm_args& g, hipStream_t s) {
  switch (g.epilogue) {
    case MAVLM_EPI_BIAS: return launch128<T, MAVLM_EPI_BIAS>(g, s);
    case MAVLM_EPI_RELU: return launch128<T, MAVLM_EPI_RELU>(g, s);
    case MAVLM_EPI_GELU: return launch128<T, MAVLM_EPI_GELU>(g, s);
    case MAVLM_EPI_RES_F32: return launch128<T, MAVLM_EPI_RES_F32>(g, s);
    case MAVLM_EPI_F32: return launch128<T, MAVLM_EPI_F32>(g, s);
    case MAVLM_EPI_LN: return launch128<T, MAVLM_EPI_LN>(g, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace

// (operand tiles are addressed through 32-bit buffer offsets: 128 rows x leading dimension must stay below 2 GiB)
bool mavlm_gemm128_supported(const mavlm_gemm_args& g) {
  return g.N % BN1 == 0 && g.K % BK1 == 0 && g.M >= 1 && (double)g.lda * 512.0 < 2.0e9 && (double)g.ldw * 512.0 < 2.0e9;
}

hipError_t mavlm_launch_gemm128(const mavlm_gemm_args& g, int dtype, hipStream_t s) {
  return dtype == MAVLM_F16 ? launch128_epi<F16>(g, s) : launch128_epi<BF16>(g, s);
}
