// Backward of the fused multi-head cross-attention (head_dim 128) - training parity for the memory path
// (the reference differentiates MemoryController.py:48-54 with autograd, which materialises
// the [H,R,S] probabilities; here they are recomputed per tile from the saved log-sum-exp, flash-attention style).
//
//   P  = exp2(S*c - lse2[q])                 S = Q.K^T (raw), c = scale*log2(e), lse2 from the forward
//   dV = P^T dO
//   dP = dO V^T,  dS = P o (dP - delta[q]),  delta[q] = sum_d dO[q,d] O[q,d]
//   dQ = scale * dS K,   dK = scale * dS^T Q
//
// ONE kernel template, three instances.  The MFMA mapping is the forward's (attention.hip): a "stationary" operand
// X sits in registers with its row index on the LANE, a "streamed" operand Y goes through LDS:
//   T^T[y][x]   = Y . X^T      A = Y rows (ds_read_b128), B = X fragments (registers)
//   A^T[d][x]  += Z^T[d][y] . E^T[y][x]     A = Z^T by ds_read_b64_tr_b16, B = the T accumulator converted in place
//     MODE 0 (dQ):  X = Q, X2 = dO | Y = K, Y2 = V  | Z = K   | lse2/delta indexed by the lane (x)
//     MODE 1 (dK):  X = K, X2 = V  | Y = Q, Y2 = dO | Z = Q   | lse2/delta indexed by the streamed row (y), from LDS
//     MODE 2 (dV):  X = K          | Y = Q          | Z = dO  | E = P
//     MODE 3 (dK and dV fused): MODE 1 plus a second accumulator dV^T += dO^T . P (Z2 = dO, already staged for dP):
//                   one S recompute less (7 products instead of 8); 128 accumulator + 64 stationary registers force one
//                   workgroup per CU (512-register waves)
// No atomics: each output row is owned by one lane, results are deterministic.
// LDS image and its XOR swizzle: as attention.hip (conflict-free for the row reads and the transposed reads).
#include "mavlm_common.h"
#include "mavlm_kernels.h"

extern int g_mavlm_attn_bwd_fused;

namespace {

constexpr int BHD = 128, BKT = 64;
constexpr int BTILE = BKT * BHD * 2;             // 16 KiB
constexpr int BSTAT = 4 * BTILE;                 // 2 stages x {lse2[64], delta[64]} floats
constexpr int BWD_LDS = 4 * BTILE + 2 * 512;

template <int V>
struct BIC { static constexpr int value = V; };

__device__ __forceinline__ int bimg_x(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <typename T, int MODE>
__global__ __launch_bounds__(256, (MODE == 3 ? 1 : 2)) void attn_bwd_kernel(const uint16_t* __restrict__ X, int ldx,
                                                          const uint16_t* __restrict__ X2, int ldx2,
                                                          const uint16_t* __restrict__ Y, int ldy,
                                                          const uint16_t* __restrict__ Y2, int ldy2,
                                                          const float* __restrict__ lse2, const float* __restrict__ delta,
                                                          uint16_t* __restrict__ Out, int ldo, int NX, int NY, int R,
                                                          int H, float c, float out_scale,
                                                          uint16_t* __restrict__ Out2 = nullptr, int ldo2 = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x % H;
  const int x0 = (blockIdx.x / H) * 128 + wave * 32;
  const int r = lane & 31, hh = lane >> 5;
  const int nt = (NY + BKT - 1) / BKT;

  // ---- stationary fragments (B operands): lane holds X[x0+r][h*128 + 16ks + 8hh + 0..7]
  typename T::vec8 xf[8], x2f[8];
  int xrow = x0 + r;
  xrow = xrow < NX ? xrow : NX - 1;
  {
    const uint16_t* xp = X + (size_t)xrow * ldx + h * BHD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) xf[ks] = *(const typename T::vec8*)(xp + 16 * ks);
    if (MODE != 2) {
      const uint16_t* xp2 = X2 + (size_t)xrow * ldx2 + h * BHD + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) x2f[ks] = *(const typename T::vec8*)(xp2 + 16 * ks);
    }
  }
  float lse_l = 0.f, del_l = 0.f;
  if (MODE == 0) {
    lse_l = lse2[(size_t)h * R + xrow];
    del_l = delta[(size_t)h * R + xrow];
  }

  // ---- staging by LDS-DMA (buffer loads, as attn_fwd3_kernel): instruction i (0..3) of wave w writes the 1 KiB block of
  // rows 16 i + 4 w + (lane>>4) of an image; physical chunk lane&15 of a row holds logical chunk (lane&15) ^ x(row),
  // x(row) = ((lane>>4)<<2) | w for every i.  One descriptor per streamed operand over this head's columns, ending after
  // row NY-1: rows of a ragged last tile read as ZEROS, and so do their statistics (lse2 = 0, delta = 0) - a zero row
  // contributes nothing to any product (dO / Q / K rows of zeros; MODE 0 masks the keys past the end explicitly).
  // (Round 1 staged through registers: 32 VGPRs of transit data and 8 ds_write_b128 per thread and tile.)
  const int drow = 4 * wave + (lane >> 4);
  const int dch = (lane & 15) ^ (((lane >> 4) << 2) | wave);
  auto rsrc_of = [&](const void* base, uint32_t bytes) {
    const uintptr_t a = (uintptr_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
  };
  const __amdgpu_buffer_rsrc_t yrs = rsrc_of(Y + h * BHD, (uint32_t)(NY - 1) * (uint32_t)ldy * 2u + (uint32_t)BHD * 2u);
  const __amdgpu_buffer_rsrc_t y2rs = rsrc_of(Y2 + h * BHD, (uint32_t)(NY - 1) * (uint32_t)ldy2 * 2u + (uint32_t)BHD * 2u);
  const __amdgpu_buffer_rsrc_t lrs = rsrc_of(lse2 + (size_t)h * R, (uint32_t)NY * 4u);     // MODE 1-3: NY == R (queries)
  const __amdgpu_buffer_rsrc_t drs = rsrc_of(delta + (size_t)h * R, (uint32_t)NY * 4u);
  int yoff[4], y2off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    yoff[i] = ((drow + 16 * i) * ldy + dch * 8) * 2;
    y2off[i] = ((drow + 16 * i) * ldy2 + dch * 8) * 2;
  }
  const unsigned lds_wave = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem + wave * 1024;
  auto dma_piece = [&](int t, int buf, auto jc) {   // piece j: 0-3 = rows 16 j.. of Y, 4-7 = of Y2
    constexpr int j = decltype(jc)::value;
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));
    const unsigned dst = base + buf * 2 * BTILE;
    if constexpr (j < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, (MAVLM_LDS void*)(uintptr_t)(dst + 4096 * j), 16, yoff[j],
                                               t * BKT * ldy * 2, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(y2rs, (MAVLM_LDS void*)(uintptr_t)(dst + BTILE + 4096 * (j - 4)), 16,
                                               y2off[j - 4], t * BKT * ldy2 * 2, 0, 0);
  };
  auto dma_stat = [&](int t, int buf) {
    if (MODE != 0 && wave < 2) {              // per-row statistics of the streamed queries: wave 0 lse2, wave 1 delta
      unsigned sb = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem;
      asm volatile("" : "+s"(sb));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? lrs : drs,
                                               (MAVLM_LDS void*)(uintptr_t)(sb + BSTAT + buf * 512 + wave * 256), 4, lane * 4,
                                               t * BKT * 4, 0, 0);
    }
  };
  auto dma_tile = [&](int t, int buf) {
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));            // M0 = base + constant stays a one-instruction recompute
    const unsigned dst = base + buf * 2 * BTILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(yrs, (MAVLM_LDS void*)(uintptr_t)(dst + 4096 * i), 16, yoff[i],
                                               t * BKT * ldy * 2, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(y2rs, (MAVLM_LDS void*)(uintptr_t)(dst + BTILE + 4096 * i), 16, y2off[i],
                                               t * BKT * ldy2 * 2, 0, 0);
    }
    if (MODE != 0 && wave < 2) {              // per-row statistics of the streamed queries: wave 0 lse2, wave 1 delta
      unsigned sb = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem;
      asm volatile("" : "+s"(sb));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? lrs : drs,
                                               (MAVLM_LDS void*)(uintptr_t)(sb + BSTAT + buf * 512 + wave * 256), 4, lane * 4,
                                               t * BKT * 4, 0, 0);
    }
  };

  // ---- fragment read geometry
  const int xr = bimg_x(r);
  int yaddr[8];                                               // row read: + 8192 b
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) yaddr[ks] = 256 * r + 16 * ((2 * ks + hh) ^ xr);
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  const int z_rd = 256 * (4 * hh + tq) + 8 * (tp & 1) + 16 * ((tp >> 1) ^ hh);
  int zaddr[4][2];                                            // [db][jj]: + 256*(32b+16s)
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) zaddr[db][jj] = z_rd + 256 * 8 * jj + 16 * (((db ^ tq) << 2) | ((tg1 ^ jj) << 1));

  f32x16 acc[4], acc2[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[d][i] = 0.f; acc2[d][i] = 0.f; }

  dma_tile(0, 0);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    asm volatile("" : "+v"(xf[ks]));
    if (MODE != 2) asm volatile("" : "+v"(x2f[ks]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    // (tile t+1 goes into the other stage - dead since the barrier that ended tile t-1 - piece by piece inside half 0;
    // a tile past the last one lies behind the descriptors' end: zeros into a stage nobody reads again, no branch)
    const char* yb = smem + cur * 2 * BTILE;                   // Y image; Y2 image at + BTILE
    const char* zb = (MODE == 2) ? yb + BTILE : yb;            // image read transposed
    const unsigned zbase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)zb;
    const unsigned z2base = (unsigned)(uintptr_t)(MAVLM_LDS const char*)(yb + BTILE);   // MODE 3: dO image, transposed
    const float* stat = (const float*)(smem + BSTAT + cur * 512);
    const bool ragged = (t == nt - 1) && (NY & (BKT - 1));

    const unsigned ybase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)yb;
    // Per tile, with the two 32-row halves b = 0, 1 of the streamed rows:
    //     S(0)   |   S(1) || V(0)   |   A(0) || V(1)   |   A(1)
    //   S(b): T^T = Y . X^T and (dQ, dK) dP^T = Y2 . X2^T  (8 k-steps of 1-2 MFMAs)
    //   V(b): E = P (dV) or dS = P o (dP - delta): exp2 / fma of the 16 values per lane, two per MFMA step of the phase it
    //         hides behind (round 1 ran S, V, A of a half in sequence: the matrix pipe idled through every V)
    //   A(b): A^T += Z^T . E^T  (8 steps)
    // Fragment reads are inline asm, two steps ahead, with hand-counted lgkmcnt waits (in-order completion: LDS operations
    // hipcc places in between - the statistics of V - only make a wait more conservative); through the builtin hipcc put
    // an s_waitcnt vmcnt(0) in front of the first transposed read, draining the next tile's DMAs mid-tile.  Those DMAs
    // are issued one piece per k-step of S(0), the statistics at the first step of S(1).
    f32x16 tt[2], dp[2];
    typename T::vec8 ef[2][2], pf[2][2];
    f32x4 l4, d4;
    auto vpair = [&](auto vbc, auto jc) {                      // values 2j, 2j+1 of half vb
      constexpr int vb = decltype(vbc)::value, j = decltype(jc)::value;
      constexpr int g = j >> 1, jj = 2 * (j & 1);              // streamed row of value i: 32 vb + (i&3) + 8(i>>2) + 4hh
      if constexpr (MODE != 0 && (j & 1) == 0) {
        l4 = *(const f32x4*)(stat + 32 * vb + 8 * g + 4 * hh);
        if constexpr (MODE == 1 || MODE == 3) d4 = *(const f32x4*)(stat + 64 + 32 * vb + 8 * g + 4 * hh);
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = 2 * j + e;
        const float l = (MODE == 0) ? lse_l : l4[jj + e];
        float x = tt[vb][i];
        asm volatile("" : "+v"(x));
        float p = __builtin_amdgcn_exp2f(x * c - l);
        if (MODE == 0 && ragged && t * BKT + 32 * vb + (jj + e) + 8 * g + 4 * hh >= NY) p = 0.f;   // key past the end
        // (MODE 1-3: a query past the end has zero Y / Y2 rows and zero statistics - p = 1, every product 0)
        float y;
        if (MODE == 2) y = p;
        else y = p * (dp[vb][i] - ((MODE == 0) ? del_l : d4[jj + e]));
        asm volatile("" : "+v"(y));
        tt[vb][i] = y;
        if (MODE == 3) dp[vb][i] = p;                           // dP is consumed: its registers carry P
      }
    };
    auto pack = [&](auto vbc) {
      constexpr int vb = decltype(vbc)::value;
#pragma unroll
      for (int sx = 0; sx < 2; ++sx) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2<T>(tt[vb][8 * sx + 2 * j], tt[vb][8 * sx + 2 * j + 1]);
        ef[vb][sx] = __builtin_bit_cast(typename T::vec8, w);
        if (MODE == 3) {
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack2<T>(dp[vb][8 * sx + 2 * j], dp[vb][8 * sx + 2 * j + 1]);
          pf[vb][sx] = __builtin_bit_cast(typename T::vec8, w);
        }
      }
    };
    auto sphase = [&](auto bc, auto vc) {                      // S(b) [|| V(vb) when vb >= 0]
      constexpr int b = decltype(bc)::value, vb = decltype(vc)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) { tt[b][i] = 0.f; dp[b][i] = 0.f; }
      constexpr int NS = (MODE != 2) ? 2 : 1;                   // reads per step
      u32x4 yfr[8], y2fr[8];
      auto yrd = [&](auto ic) {
        constexpr int ks = decltype(ic)::value;
        const unsigned a = ybase + (unsigned)yaddr[ks];
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(8192 * b));
        yfr[ks] = v;
        if constexpr (MODE != 2) {
          u32x4 v2;
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v2) : "v"(a), "i"(BTILE + 8192 * b));
          y2fr[ks] = v2;
        }
      };
      auto sstep = [&](auto ic) {
        constexpr int ks = decltype(ic)::value;
        if constexpr (ks + 2 < 8) yrd(BIC<(ks + 2 < 8 ? ks + 2 : 7)>{});
        constexpr int ahead = (7 - ks) < 2 ? (7 - ks) : 2;
        if constexpr (ahead * NS == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead * NS == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else if constexpr (ahead * NS == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        tt[b] = T::mfma32(__builtin_bit_cast(typename T::vec8, yfr[ks]), xf[ks], tt[b]);
        if constexpr (MODE != 2) dp[b] = T::mfma32(__builtin_bit_cast(typename T::vec8, y2fr[ks]), x2f[ks], dp[b]);
        if constexpr (b == 0) dma_piece(t + 1, cur ^ 1, BIC<ks>{});
        if constexpr (b == 1 && ks == 0) dma_stat(t + 1, cur ^ 1);
        if constexpr (vb >= 0) vpair(BIC<(vb >= 0 ? vb : 0)>{}, BIC<ks>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      yrd(BIC<0>{});
      yrd(BIC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      sstep(BIC<0>{}); sstep(BIC<1>{}); sstep(BIC<2>{}); sstep(BIC<3>{});
      sstep(BIC<4>{}); sstep(BIC<5>{}); sstep(BIC<6>{}); sstep(BIC<7>{});
    };
    auto aphase = [&](auto bc, auto vc) {                      // A(b) [|| V(vb) when vb >= 0]
      constexpr int b = decltype(bc)::value, vb = decltype(vc)::value;
      const unsigned zbase_b = zbase + 256 * 32 * b, z2base_b = z2base + 256 * 32 * b;
      constexpr int NRD = (MODE == 3) ? 4 : 2;                  // reads per step
      u32x2 zlo[8], zhi[8], z2lo[8], z2hi[8];
      auto zrd = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int db = i >> 1, sx = i & 1;
        const unsigned a0 = zbase_b + zaddr[db][0], a1 = zbase_b + zaddr[db][1];
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "i"(256 * 16 * sx + 0));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "i"(256 * 16 * sx + 0));
        zlo[i] = lo; zhi[i] = hi;
        if constexpr (MODE == 3) {
          const unsigned c0 = z2base_b + zaddr[db][0], c1 = z2base_b + zaddr[db][1];
          u32x2 lo2, hi2;
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo2) : "v"(c0), "i"(256 * 16 * sx + 0));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi2) : "v"(c1), "i"(256 * 16 * sx + 0));
          z2lo[i] = lo2; z2hi[i] = hi2;
        }
      };
      auto zstep = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int db = i >> 1, sx = i & 1;
        if constexpr (i + 2 < 8) zrd(BIC<(i + 2 < 8 ? i + 2 : 7)>{});
        constexpr int ahead = (7 - i) < 2 ? (7 - i) : 2;
        if constexpr (ahead * NRD == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead * NRD == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if constexpr (ahead * NRD == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                      // keep the MFMA below the wait
        u32x4 both;
        both[0] = zlo[i][0]; both[1] = zlo[i][1]; both[2] = zhi[i][0]; both[3] = zhi[i][1];
        acc[db] = T::mfma32(__builtin_bit_cast(typename T::vec8, both), ef[b][sx], acc[db]);
        if constexpr (MODE == 3) {
          u32x4 b2;
          b2[0] = z2lo[i][0]; b2[1] = z2lo[i][1]; b2[2] = z2hi[i][0]; b2[3] = z2hi[i][1];
          acc2[db] = T::mfma32(__builtin_bit_cast(typename T::vec8, b2), pf[b][sx], acc2[db]);
        }
        if constexpr (vb >= 0) vpair(BIC<(vb >= 0 ? vb : 0)>{}, BIC<i>{});
        __builtin_amdgcn_sched_barrier(0);
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // nothing older than the reads below is outstanding
      zrd(BIC<0>{});
      zrd(BIC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      zstep(BIC<0>{}); zstep(BIC<1>{}); zstep(BIC<2>{}); zstep(BIC<3>{});
      zstep(BIC<4>{}); zstep(BIC<5>{}); zstep(BIC<6>{}); zstep(BIC<7>{});
    };
    sphase(BIC<0>{}, BIC<-1>{});
    sphase(BIC<1>{}, BIC<0>{});
    pack(BIC<0>{});
    aphase(BIC<0>{}, BIC<1>{});
    pack(BIC<1>{});
    aphase(BIC<1>{}, BIC<-1>{});

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMAs of tile t+1 have landed
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: Out[x][h*128 + 32db + 8g + 4hh + 0..3] = A^T * out_scale
  const int x = x0 + r;
  if (x < NX) {
    uint16_t* op = Out + (size_t)x * ldo + h * BHD + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(u32x2*)(op + 32 * db + 8 * g) = pack4<T>(acc[db][4 * g] * out_scale, acc[db][4 * g + 1] * out_scale,
                                                   acc[db][4 * g + 2] * out_scale, acc[db][4 * g + 3] * out_scale);
    if (MODE == 3) {
      uint16_t* op2 = Out2 + (size_t)x * ldo2 + h * BHD + 4 * hh;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(u32x2*)(op2 + 32 * db + 8 * g) = pack4<T>(acc2[db][4 * g], acc2[db][4 * g + 1], acc2[db][4 * g + 2],
                                                      acc2[db][4 * g + 3]);
    }
  }
}

// delta[h][q] = sum_d dO[q, h*128+d] * O[q, h*128+d]; one wave per query row, 8 lanes per head.
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const uint16_t* __restrict__ O, int ldo,
                                                         const uint16_t* __restrict__ dO, int lddo,
                                                         float* __restrict__ delta, int R, int H) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= R) return;
  for (int c0 = 0; c0 < H * BHD; c0 += 1024) {
    const int col = c0 + lane * 16;
    float s = 0.f;
    if (col < H * BHD) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u16x8 a = *(const u16x8*)(O + (size_t)q * ldo + col + 8 * k);
        const u16x8 b = *(const u16x8*)(dO + (size_t)q * lddo + col + 8 * k);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += T::to_f32(a[j]) * T::to_f32(b[j]);
      }
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if ((lane & 7) == 0 && col < H * BHD) delta[(size_t)(col / BHD) * R + q] = s;
  }
}

template <typename T, int MODE>
void launch_mode(dim3 grid, hipStream_t s, const void* X, int ldx, const void* X2, int ldx2, const void* Y, int ldy,
                 const void* Y2, int ldy2, const float* lse2, const float* delta, void* out, int ldo, int NX, int NY,
                 int R, int H, float c, float out_scale, void* out2 = nullptr, int ldo2 = 0) {
  hipLaunchKernelGGL((attn_bwd_kernel<T, MODE>), grid, dim3(256), BWD_LDS, s, (const uint16_t*)X, ldx, (const uint16_t*)X2,
                     ldx2, (const uint16_t*)Y, ldy, (const uint16_t*)Y2, ldy2, lse2, delta, (uint16_t*)out, ldo, NX, NY,
                     R, H, c, out_scale, (uint16_t*)out2, ldo2);
}

template <typename T>
hipError_t launch_all(const mavlm_attn_bwd_args& a, hipStream_t s) {
  // streamed operands are addressed through 32-bit buffer offsets (one tile past the end included)
  const double lim = 2147483648.0;
  if (((double)a.S + 64) * a.ldk * 2.0 >= lim || ((double)a.S + 64) * a.ldv * 2.0 >= lim ||
      ((double)a.R + 64) * a.ldq * 2.0 >= lim || ((double)a.R + 64) * a.lddo * 2.0 >= lim)
    return hipErrorInvalidValue;
  const float c = a.scale * 1.44269504088896340736f;
  static mavlm_per_device_once once[4];
  {
    hipError_t e = once[0].dyn_lds((const void*)attn_bwd_kernel<T, 0>, BWD_LDS);
    if (e == hipSuccess) e = once[1].dyn_lds((const void*)attn_bwd_kernel<T, 1>, BWD_LDS);
    if (e == hipSuccess) e = once[2].dyn_lds((const void*)attn_bwd_kernel<T, 2>, BWD_LDS);
    if (e == hipSuccess) e = once[3].dyn_lds((const void*)attn_bwd_kernel<T, 3>, BWD_LDS);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(attn_delta_kernel<T>, dim3((a.R + 3) / 4), dim3(256), 0, s, (const uint16_t*)a.O, a.ldo,
                     (const uint16_t*)a.dO, a.lddo, a.delta, a.R, a.H);
  const dim3 gq(((a.R + 127) / 128) * a.H), gk(((a.S + 127) / 128) * a.H);
  if (a.dQ != nullptr)
    launch_mode<T, 0>(gq, s, a.Q, a.ldq, a.dO, a.lddo, a.K, a.ldk, a.V, a.ldv, a.lse2, a.delta, a.dQ, a.lddq, a.R, a.S,
                      a.R, a.H, c, a.scale);
  if (a.dK != nullptr && a.dV != nullptr && g_mavlm_attn_bwd_fused) {
    launch_mode<T, 3>(gk, s, a.K, a.ldk, a.V, a.ldv, a.Q, a.ldq, a.dO, a.lddo, a.lse2, a.delta, a.dK, a.lddk, a.S, a.R,
                      a.R, a.H, c, a.scale, a.dV, a.lddv);
    return hipGetLastError();
  }
  if (a.dK != nullptr)
    launch_mode<T, 1>(gk, s, a.K, a.ldk, a.V, a.ldv, a.Q, a.ldq, a.dO, a.lddo, a.lse2, a.delta, a.dK, a.lddk, a.S, a.R,
                      a.R, a.H, c, a.scale);
  if (a.dV != nullptr)
    launch_mode<T, 2>(gk, s, a.K, a.ldk, nullptr, 0, a.Q, a.ldq, a.dO, a.lddo, a.lse2, a.delta, a.dV, a.lddv, a.S, a.R,
                      a.R, a.H, c, 1.0f);
  return hipGetLastError();
}

}  // namespace

int g_mavlm_attn_bwd_fused = 0;   // experiment switch: 1 = fused dK+dV kernel (512-register waves, one workgroup per CU)

hipError_t mavlm_launch_attention_bwd(const mavlm_attn_bwd_args& a, int dtype, hipStream_t s) {
  return dtype == MAVLM_F16 ? launch_all<F16>(a, s) : launch_all<BF16>(a, s);
}
