// Backward of the fused multi-head cross-attention for WIDE heads (head_dim 448: LLaVA-OneVision-7B, hidden 3584 / 8 heads) -
// flash style: the probabilities are recomputed per tile from the saved log-sum-exp, nothing of size [R, S] touches memory.
// (Rounds 2-3, and round 4 until this kernel, trained this width through a GEMM-composed backward: one head's [R, S] scores materialised in fp32, five GEMMs,
// five transposes and two elementwise passes per head - 30 launches per head, 240 per attention.)
//
//   P  = exp2(S*c - lse2[q])                 S = Q.K^T (raw), c = scale*log2(e), lse2 from the forward
//   dV = P^T dO
//   dP = dO V^T,  dS = P o (dP - delta[q]),  delta[q] = sum_d dO[q,d] O[q,d]
//   dQ = scale * dS K,   dK = scale * dS^T Q                                   (MemoryController.py:48-54 under autograd)
//
// Mapping = attn_bwd_kernel's (attention_bwd.hip), on the LDS image of attn_fwd_hd2_kernel (attention_hd.hip): a "stationary"
// operand X sits in registers with its row on the LANE, a "streamed" operand Y goes through LDS in 32-row tiles of four
// 128-column sub-images:
//   T^T[y][x]   = Y . X^T                    A = Y rows (ds_read_b128), B = X fragments (registers); 28 k-steps of 16 columns
//   A^T[d][x]  += Z^T[d][y] . E^T[y][x]      A = Z^T by ds_read_b64_tr_b16, B = the T accumulator converted in place
//     MODE 0 (dQ):        X = Q, X2 = dO | Y = K, Y2 = V  | Z = K          | lse2/delta indexed by the lane (x)
//     MODE 1 (dK):        X = K, X2 = V  | Y = Q, Y2 = dO | Z = Q          | lse2/delta indexed by the streamed row (y), from LDS
//     MODE 2 (dV):        X = K          | Y = Q          | Z = dO         | E = P
//     MODE 3 (dK and dV): MODE 1 plus dV^T += dO^T . P (Z2 = dO, already staged for dP)
// What does not fit at this width is a wave's state: 32 rows of X and X2 are 224 registers, a 32 x 448 fp32 output 224 more.
//   * MODE 2 has one stationary operand: 112 + 224 registers, every wave owns 32 rows of X (128 per workgroup).
//   * MODE 0 / 1 / 3: the four waves are 2 row groups x 2 ROLES.  In the first phase wave (g, 0) holds the rows of X and computes
//     T = Y . X^T, wave (g, 1) holds the same rows of X2 and computes dP = Y2 . X2^T - 28 MFMAs each, one stationary operand
//     (112 registers) each; the two 32 x 32 fp32 tiles are exchanged through LDS (4 KiB per wave, one extra barrier per tile),
//     both waves evaluate dS (and P) on them, and in the second phase each wave accumulates ITS 7 of the 14 output blocks
//     (wave (g, s): blocks 2 db + s, so that both walk the sub-images alike and every LDS offset is an immediate): exactly the
//     3 (dQ) / 4 (dK + dV) matrix products of the math, none computed twice (a first form of this kernel let both waves of a row
//     group compute T and dP in full: 5 products' worth of MFMAs for 3).
// One workgroup per CU (512-register waves), 145 KiB of LDS: {Y, Y2} x 2 stages + the exchange; the next tile's 16 LDS-DMA pieces
// per wave are issued between the MFMAs of the first phase.  No atomics: each output element is owned by one lane, results are
// deterministic.  Rounding points as the 128-wide kernels (and the composed path it replaces): P and dS rounded to 16 bits for
// the second product, fp32 accumulation everywhere.
#include "mavlm_common.h"
#include "mavlm_kernels.h"
#include <type_traits>
#include <utility>

namespace {

template <int V>
struct WIC { static constexpr int value = V; };
template <int... I, typename F>
__device__ __forceinline__ void bw_for_each(std::integer_sequence<int, I...>, F&& f) { (f(WIC<I>{}), ...); }

constexpr int WKT = 32;                                   // streamed rows per tile

__device__ __forceinline__ int wimg_x(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int N>
__device__ __forceinline__ void wait_lgkm() {
  static_assert(N == 0 || N == 1 || N == 2 || N == 3 || N == 4 || N == 6 || N == 8, "add the count to the table");
  if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
}

template <typename T, int MODE, int HD>
__global__ __launch_bounds__(256, 1) void attn_bwd_hd_kernel(const uint16_t* __restrict__ X, int ldx,
                                                             const uint16_t* __restrict__ X2, int ldx2,
                                                             const uint16_t* __restrict__ Y, int ldy,
                                                             const uint16_t* __restrict__ Y2, int ldy2,
                                                             const float* __restrict__ lse2, const float* __restrict__ delta,
                                                             uint16_t* __restrict__ Out, int ldo, int NX, int NY, int R, int H,
                                                             float c, float out_scale, uint16_t* __restrict__ Out2, int ldo2) {
  constexpr int NSUB = (HD + 127) / 128;
  constexpr int SUB = WKT * 256;                            // 8 KiB: [32 rows][256 B]
  constexpr int TILE = NSUB * SUB;                          // 32 KiB
  constexpr int KS = HD / 16;                               // k-steps of the first product
  constexpr bool TWO = (MODE != 2);                         // two first-phase products, one per wave of a row group
  constexpr bool DKV = (MODE == 3);
  constexpr int SL = TWO ? 2 : 1;                           // waves sharing a row group (= output slabs)
  constexpr int DBS = HD / 32 / SL;                         // 32-column output blocks per wave
  constexpr int NA = 2 * DBS;                               // steps of the second phase: (block, 16-row step)
  constexpr int NPW = NSUB * 2;                             // 1 KiB DMA pieces per wave, image and tile
  constexpr int XCH = 4 * TILE;                             // exchange: [row group][role][4][64 lanes][16 B] = 16 KiB
  constexpr int STAT = XCH + 16384;                         // 2 stages x {lse2[64], delta[64]} floats
  constexpr int KPF = 4;                                    // first phase: fragment reads in flight ahead of their MFMA
  constexpr int ZPF = DKV ? 2 : 4;                          // second phase (two / four reads per step)
  static_assert(HD % 64 == 0 && (HD / 32) % SL == 0 && DBS >= 6 && 2 * NPW <= KS && TILE + (NSUB - 1) * SUB + 4096 < 65536, "wide head");
  // LDS: Y stage 0, Y stage 1, Y2 stage 0, Y2 stage 1 (stage and sub-image offsets then fit the 16-bit instruction immediates)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave / SL, slab = wave % SL;                // slab = role in the first phase = block parity in the second
  const int h = blockIdx.x % H;
  const int x0 = (blockIdx.x / H) * (32 * (4 / SL)) + grp * 32;
  const int r = lane & 31, hh = lane >> 5;
  const int nt = (NY + WKT - 1) / WKT;

  // ---- stationary fragments (B operand) of this wave's first-phase product: lane holds X[x0 + r][h*HD + 16 ks + 8 hh + 0..7]
  // (role 1: X2)
  typename T::vec8 xf[KS];
  int xrow = x0 + r;
  xrow = xrow < NX ? xrow : NX - 1;
  {
    const uint16_t* xp = (TWO && slab == 1) ? X2 + (size_t)xrow * ldx2 : X + (size_t)xrow * ldx;
    xp += h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const typename T::vec8*)(xp + 16 * ks);
  }
  float lse_l = 0.f, del_l = 0.f;
  if (MODE == 0) {
    lse_l = lse2[(size_t)h * R + xrow];
    del_l = delta[(size_t)h * R + xrow];
  }

  // ---- LDS-DMA (attn_fwd_hd2_kernel's): piece k (0..NPW-1) of wave w is 1 KiB = rows 4 rg .. 4 rg + 3, rg = w + 4 (k & 1), of
  // sub-image k >> 1; the lane's source chunk (swizzled through the source address) does not depend on k.  One descriptor per
  // streamed operand over this head's columns, ending after row NY-1: rows of a ragged last tile (and a tile past the last one)
  // read as ZEROS, and so do their statistics - a zero row contributes nothing to any product (MODE 0 masks the keys past the
  // end explicitly).
  auto rsrc_of = [&](const void* base, uint32_t bytes) {
    const uintptr_t a = (uintptr_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
  };
  const __amdgpu_buffer_rsrc_t yrs = rsrc_of(Y + h * HD, (uint32_t)(NY - 1) * (uint32_t)ldy * 2u + (uint32_t)HD * 2u);
  const __amdgpu_buffer_rsrc_t y2rs = rsrc_of(Y2 + h * HD, (uint32_t)(NY - 1) * (uint32_t)ldy2 * 2u + (uint32_t)HD * 2u);
  const __amdgpu_buffer_rsrc_t lrs = rsrc_of(lse2 + (size_t)h * R, (uint32_t)NY * 4u);     // MODE 1-3: NY == R (queries)
  const __amdgpu_buffer_rsrc_t drs = rsrc_of(delta + (size_t)h * R, (uint32_t)NY * 4u);
  const int drow = 4 * wave + (lane >> 4);
  const int dch = (lane & 15) ^ (((lane >> 4) << 2) | wave);
  const int yoff = (drow * ldy + dch * 8) * 2, y2off = (drow * ldy2 + dch * 8) * 2;
  const unsigned lds_wave = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem + wave * 1024;
  auto dma_piece = [&](__amdgpu_buffer_rsrc_t rs, int off, int ld, int toff, int slot_off, auto kc) {
    constexpr int k = decltype(kc)::value;
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));            // M0 = base + constant stays a one-instruction recompute
    // (the sub-image's column offset rides in the instruction immediate, which the hardware adds to the LDS address too)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (MAVLM_LDS void*)(uintptr_t)(base + slot_off + k * 4096 - (k >> 1) * 256), 16, off,
                                             toff + (k & 1) * 32 * ld, (k >> 1) * 256, 0);
  };
  auto dma_stat = [&](int t, int stage) {
    if (MODE != 0 && wave < 2) {              // per-row statistics of the streamed queries: wave 0 lse2, wave 1 delta
      unsigned sb = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem;
      asm volatile("" : "+s"(sb));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? lrs : drs,
                                               (MAVLM_LDS void*)(uintptr_t)(sb + STAT + stage * 512 + wave * 256), 4, lane * 4,
                                               t * WKT * 4, 0, 0);
    }
  };

  // ---- fragment read geometry (loop-invariant 32-bit LDS addresses; stage, sub-image and 16-row step are immediates)
  const int xr = wimg_x(r);
  const unsigned sbase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)smem;
  unsigned kad[8];                                            // k-step ks reads kad[ks & 7] + (ks >> 3) SUB + stage TILE
#pragma unroll
  for (int j = 0; j < 8; ++j) kad[j] = sbase + ((TWO && slab == 1) ? 2 * TILE : 0) + 256 * r + 16 * ((2 * j + hh) ^ xr);
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  const int z_rd = 256 * (4 * hh + tq) + 8 * (tp & 1) + 16 * ((tp >> 1) ^ hh);
  // Output blocks of a wave: the blocks 2 db + slab of the head when two waves share a row group, all 14 otherwise.
  // Block db reads zad[db % BPS][jj] + (db / BPS) SUB + 4096 sx + stage TILE (the second output of MODE 3: + 2 TILE, the Y2 image).
  constexpr int BPS = 4 / SL;                                 // blocks of a wave per sub-image
  unsigned zad[BPS][2], zad2[DKV ? BPS : 1][2];
#pragma unroll
  for (int j = 0; j < BPS; ++j)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int db = j * SL + slab;                           // block inside a sub-image
      zad[j][jj] = sbase + (MODE == 2 ? 2 * TILE : 0) + z_rd + 256 * 8 * jj + 16 * (((db ^ tq) << 2) | ((tg1 ^ jj) << 1));
      if constexpr (DKV) zad2[j][jj] = zad[j][jj] + 2 * TILE;
    }
  // exchange slots of this row group: role 0 (T) at xch, role 1 (dP) at xch + 4096; chunk j of a tile at + 1024 j + 16 lane
  char* const xch = smem + XCH + grp * 8192 + lane * 16;

  f32x16 acc[DBS], acc2[DKV ? DBS : 1];
#pragma unroll
  for (int d = 0; d < DBS; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[d][i] = 0.f;
      if constexpr (DKV) acc2[d][i] = 0.f;
    }

  // ---- prologue: tile 0 into stage 0
  bw_for_each(std::make_integer_sequence<int, NPW>{}, [&](auto kc) { dma_piece(yrs, yoff, ldy, 0, 0, kc); });
  bw_for_each(std::make_integer_sequence<int, NPW>{}, [&](auto kc) { dma_piece(y2rs, y2off, ldy2, 0, 2 * TILE, kc); });
  dma_stat(0, 0);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xf[ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  auto tile = [&](auto par, int t) {
    constexpr int P = decltype(par)::value;
    constexpr int N = P ^ 1;
    const float* stat = (const float*)(smem + STAT + P * 512);
    const bool ragged = (t == nt - 1) && (NY & (WKT - 1));
    const int ytile = (t + 1) * WKT * ldy * 2, y2tile = (t + 1) * WKT * ldy2 * 2;     // scalar byte offsets of tile t+1

    // second-phase fragments (step i = (16-row step i / DBS, block i % DBS)): the first ZPF steps' reads are issued before the
    // exchange, so that they travel under its write - barrier - read round trip
    constexpr int NRD = DKV ? 4 : 2;                          // reads per step
    u32x2 zlo[NA], zhi[NA], z2lo[DKV ? NA : 1], z2hi[DKV ? NA : 1];
    auto zrd = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int db = i % DBS, sx = i / DBS;
      constexpr int off = P * TILE + (db / BPS) * SUB + 4096 * sx;
      u32x2 lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(zad[db % BPS][0]), "i"(off));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(zad[db % BPS][1]), "i"(off));
      zlo[i] = lo; zhi[i] = hi;
      if constexpr (DKV) {
        u32x2 lo2, hi2;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo2) : "v"(zad2[db % BPS][0]), "i"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi2) : "v"(zad2[db % BPS][1]), "i"(off));
        z2lo[i] = lo2; z2hi[i] = hi2;
      }
    };

    // ---- first phase: this wave's product (role 0: T^T = Y . X^T, role 1: dP^T = Y2 . X2^T; kad points at its image); the DMA
    // pieces of tile t+1 (other stage - dead since the barrier that ended tile t-1) ride along, one per k-step; a tile past the
    // last one lies behind the descriptors' end.
    f32x16 tt, dp;
    {
      f32x16 mine;
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[i] = 0.f;
      u32x4 yfr[KS];
      auto yrd = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int off = P * TILE + (i >> 3) * SUB;
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(kad[i & 7]), "i"(off));
        yfr[i] = v;
      };
      auto sstep = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i + KPF < KS) yrd(WIC<(i + KPF < KS ? i + KPF : KS - 1)>{});
        constexpr int ahead = (KS - 1 - i) < KPF ? (KS - 1 - i) : KPF;
        wait_lgkm<ahead>();
        __builtin_amdgcn_sched_barrier(0);                      // keep the MFMA below the wait
        mine = T::mfma32(__builtin_bit_cast(typename T::vec8, yfr[i]), xf[i], mine);
        if constexpr (i < NPW) dma_piece(yrs, yoff, ldy, ytile, N * TILE, WIC<(i < NPW ? i : 0)>{});
        else if constexpr (i < 2 * NPW) dma_piece(y2rs, y2off, ldy2, y2tile, (2 + N) * TILE, WIC<(i < 2 * NPW ? i - NPW : 0)>{});
        else if constexpr (i == 2 * NPW) dma_stat(t + 1, N);
        __builtin_amdgcn_sched_barrier(0);
      };
      wait_lgkm<0>();
      bw_for_each(std::make_integer_sequence<int, KPF>{}, yrd);
      __builtin_amdgcn_sched_barrier(0);
      bw_for_each(std::make_integer_sequence<int, KS>{}, sstep);
      bw_for_each(std::make_integer_sequence<int, ZPF>{}, zrd);
      if constexpr (TWO) {
        // exchange with the partner wave of the row group (same lanes hold the same elements of both tiles)
        char* const mp = xch + slab * 4096;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(f32x4*)(mp + 1024 * j) = f32x4{mine[4 * j], mine[4 * j + 1], mine[4 * j + 2], mine[4 * j + 3]};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (no vmcnt wait: the DMAs of tile t+1 stay in flight)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 a = *(const f32x4*)(xch + 1024 * j), b = *(const f32x4*)(xch + 4096 + 1024 * j);
#pragma unroll
          for (int e = 0; e < 4; ++e) { tt[4 * j + e] = a[e]; dp[4 * j + e] = b[e]; }
        }
      } else {
        tt = mine;
      }
    }

    // ---- E = P (dV) or dS = P o (dP - delta); value i of the lane is streamed row (i & 3) + 8 (i >> 2) + 4 hh of the tile.
    // Second phase: A^T[d][x] += Z^T[d][y] . E^T[y][x] over this wave's DBS blocks (MODE 3: and dV^T[d][x] += dO^T[d][y] . P^T[y][x]
    // from the Y2 image), step i = (16-row step sx = i / DBS, block db = i % DBS): the steps of sx = 0 need the values 0-7 only, so
    // the values 8-15 are evaluated between their MFMAs, and the first fragments are on their way while the values 0-7 are.
    typename T::vec8 ef[2], pf[2];
    {
      f32x4 l4[4], d4[4];                                       // statistics of the streamed rows (MODE 1-3)
      auto vstat = [&](int gq) {
        if constexpr (MODE != 0) {
          l4[gq] = *(const f32x4*)(stat + 8 * gq + 4 * hh);
          if constexpr (MODE != 2) d4[gq] = *(const f32x4*)(stat + 64 + 8 * gq + 4 * hh);
        }
      };
      auto vval = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int gq = i >> 2, e = i & 3;
        const float l = (MODE == 0) ? lse_l : l4[gq][e];
        float p = __builtin_amdgcn_exp2f(tt[i] * c - l);
        if (MODE == 0 && ragged && t * WKT + e + 8 * gq + 4 * hh >= NY) p = 0.f;        // key past the end
        // (MODE 1-3: a query past the end has zero Y / Y2 rows and zero statistics - p = 1, every product 0)
        if constexpr (MODE == 2) {
          tt[i] = p;
        } else {
          tt[i] = p * (dp[i] - ((MODE == 0) ? del_l : d4[gq][e]));
          if constexpr (DKV) dp[i] = p;                         // dP is consumed: its registers carry P
        }
      };
      auto vpack = [&](auto sc) {
        constexpr int sx = decltype(sc)::value;
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2<T>(tt[8 * sx + 2 * j], tt[8 * sx + 2 * j + 1]);
        asm volatile("" : "+v"(w));
        ef[sx] = __builtin_bit_cast(typename T::vec8, w);
        if constexpr (DKV) {
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack2<T>(dp[8 * sx + 2 * j], dp[8 * sx + 2 * j + 1]);
          asm volatile("" : "+v"(w));
          pf[sx] = __builtin_bit_cast(typename T::vec8, w);
        }
      };
      vstat(0); vstat(1); vstat(2); vstat(3);
      bw_for_each(std::make_integer_sequence<int, 8>{}, vval);
      vpack(WIC<0>{});
      auto zstep = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int db = i % DBS, sx = i / DBS;
        if constexpr (i + ZPF < NA) zrd(WIC<(i + ZPF < NA ? i + ZPF : NA - 1)>{});
        constexpr int ahead = (NA - 1 - i) < ZPF ? (NA - 1 - i) : ZPF;
        wait_lgkm<ahead * NRD>();
        __builtin_amdgcn_sched_barrier(0);
        u32x4 both;
        both[0] = zlo[i][0]; both[1] = zlo[i][1]; both[2] = zhi[i][0]; both[3] = zhi[i][1];
        acc[db] = T::mfma32(__builtin_bit_cast(typename T::vec8, both), ef[sx], acc[db]);
        if constexpr (DKV) {
          u32x4 b2;
          b2[0] = z2lo[i][0]; b2[1] = z2lo[i][1]; b2[2] = z2hi[i][0]; b2[3] = z2hi[i][1];
          acc2[db] = T::mfma32(__builtin_bit_cast(typename T::vec8, b2), pf[sx], acc2[db]);
        }
        // the values 8-15 ride along the steps of sx = 0: two per step, packed behind the last of them
        if constexpr (i < 4) {
          vval(WIC<8 + 2 * (i < 4 ? i : 0)>{});
          vval(WIC<9 + 2 * (i < 4 ? i : 0)>{});
        }
        if constexpr (i == 4) vpack(WIC<1>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      bw_for_each(std::make_integer_sequence<int, NA>{}, zstep);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's DMAs of tile t+1 have landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  int t = 0;
  for (; t + 1 < nt; t += 2) {
    tile(WIC<0>{}, t);
    tile(WIC<1>{}, t + 1);
  }
  if (t < nt) tile(WIC<0>{}, t);

  // ---- epilogue: Out[x][h*HD + 32 (SL db + slab) + 8 g + 4 hh + 0..3] = A^T * out_scale
  const int x = x0 + r;
  if (x < NX) {
    uint16_t* op = Out + (size_t)x * ldo + h * HD + 32 * slab + 4 * hh;
#pragma unroll
    for (int db = 0; db < DBS; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(u32x2*)(op + 32 * SL * db + 8 * g) = pack4<T>(acc[db][4 * g] * out_scale, acc[db][4 * g + 1] * out_scale,
                                                        acc[db][4 * g + 2] * out_scale, acc[db][4 * g + 3] * out_scale);
    if constexpr (DKV) {
      uint16_t* op2 = Out2 + (size_t)x * ldo2 + h * HD + 32 * slab + 4 * hh;
#pragma unroll
      for (int db = 0; db < DBS; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(u32x2*)(op2 + 32 * SL * db + 8 * g) = pack4<T>(acc2[db][4 * g], acc2[db][4 * g + 1], acc2[db][4 * g + 2],
                                                           acc2[db][4 * g + 3]);
    }
  }
}

constexpr int bwd_hd_lds(int hd) { return 4 * ((hd + 127) / 128) * WKT * 256 + 16384 + 1024; }

template <typename T, int MODE, int HD>
void launch_mode_hd(dim3 grid, hipStream_t s, const void* X, int ldx, const void* X2, int ldx2, const void* Y, int ldy,
                    const void* Y2, int ldy2, const float* lse2, const float* delta, void* out, int ldo, int NX, int NY, int R,
                    int H, float c, float out_scale, void* out2 = nullptr, int ldo2 = 0) {
  hipLaunchKernelGGL((attn_bwd_hd_kernel<T, MODE, HD>), grid, dim3(256), bwd_hd_lds(HD), s, (const uint16_t*)X, ldx,
                     (const uint16_t*)X2, ldx2, (const uint16_t*)Y, ldy, (const uint16_t*)Y2, ldy2, lse2, delta, (uint16_t*)out,
                     ldo, NX, NY, R, H, c, out_scale, (uint16_t*)out2, ldo2);
}

template <typename T, int HD>
hipError_t launch_all_hd(const mavlm_attn_bwd_args& a, int dtype, hipStream_t s) {
  // streamed operands are addressed through 32-bit buffer offsets (one tile past the end included)
  const double lim = 2147483648.0;
  if (((double)a.S + 64) * a.ldk * 2.0 >= lim || ((double)a.S + 64) * a.ldv * 2.0 >= lim ||
      ((double)a.R + 64) * a.ldq * 2.0 >= lim || ((double)a.R + 64) * a.lddo * 2.0 >= lim)
    return hipErrorInvalidValue;
  const float c = a.scale * 1.44269504088896340736f;
  static mavlm_per_device_once once[4];
  {
    hipError_t e = once[0].dyn_lds((const void*)attn_bwd_hd_kernel<T, 0, HD>, bwd_hd_lds(HD));
    if (e == hipSuccess) e = once[1].dyn_lds((const void*)attn_bwd_hd_kernel<T, 1, HD>, bwd_hd_lds(HD));
    if (e == hipSuccess) e = once[2].dyn_lds((const void*)attn_bwd_hd_kernel<T, 2, HD>, bwd_hd_lds(HD));
    if (e == hipSuccess) e = once[3].dyn_lds((const void*)attn_bwd_hd_kernel<T, 3, HD>, bwd_hd_lds(HD));
    if (e != hipSuccess) return e;
  }
  {
    hipError_t e = mavlm_launch_rowdot(a.dO, a.lddo, a.O, a.ldo, a.delta, a.R, a.H, HD, dtype, s);
    if (e != hipSuccess) return e;
  }
  const dim3 gq(((a.R + 63) / 64) * a.H), gk(((a.S + 63) / 64) * a.H), gv(((a.S + 127) / 128) * a.H);
  if (a.dQ != nullptr)
    launch_mode_hd<T, 0, HD>(gq, s, a.Q, a.ldq, a.dO, a.lddo, a.K, a.ldk, a.V, a.ldv, a.lse2, a.delta, a.dQ, a.lddq, a.R, a.S,
                             a.R, a.H, c, a.scale);
  if (a.dK != nullptr && a.dV != nullptr) {
    launch_mode_hd<T, 3, HD>(gk, s, a.K, a.ldk, a.V, a.ldv, a.Q, a.ldq, a.dO, a.lddo, a.lse2, a.delta, a.dK, a.lddk, a.S, a.R,
                             a.R, a.H, c, a.scale, a.dV, a.lddv);
    return hipGetLastError();
  }
  if (a.dK != nullptr)
    launch_mode_hd<T, 1, HD>(gk, s, a.K, a.ldk, a.V, a.ldv, a.Q, a.ldq, a.dO, a.lddo, a.lse2, a.delta, a.dK, a.lddk, a.S, a.R,
                             a.R, a.H, c, a.scale);
  if (a.dV != nullptr)
    launch_mode_hd<T, 2, HD>(gv, s, a.K, a.ldk, nullptr, 0, a.Q, a.ldq, a.dO, a.lddo, a.lse2, a.delta, a.dV, a.lddv, a.S, a.R,
                             a.R, a.H, c, 1.0f);
  return hipGetLastError();
}

}  // namespace

hipError_t mavlm_launch_attention_bwd_hd(const mavlm_attn_bwd_args& a, int head_dim, int dtype, hipStream_t s) {
  if (head_dim != 448) return hipErrorInvalidValue;
  return dtype == MAVLM_F16 ? launch_all_hd<F16, 448>(a, dtype, s) : launch_all_hd<BF16, 448>(a, dtype, s);
}
