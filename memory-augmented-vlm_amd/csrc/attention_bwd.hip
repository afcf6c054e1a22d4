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

  // ---- staging through registers: thread handles chunks row = (tid>>4) + 16 i, ch = tid & 15 of both images
  const int srow = tid >> 4, sch = tid & 15;
  const int st_off = 256 * srow + 16 * (sch ^ bimg_x(srow));   // + 4096 i
  const uint16_t* yg = Y + h * BHD + sch * 8;
  const uint16_t* y2g = Y2 + h * BHD + sch * 8;
  u32x4 yreg[4], y2reg[4];
  float sreg = 0.f;
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = t * BKT + srow + 16 * i;
      row = row < NY ? row : NY - 1;
      yreg[i] = *(const u32x4*)(yg + (size_t)row * ldy);
      y2reg[i] = *(const u32x4*)(y2g + (size_t)row * ldy2);
    }
    if (MODE != 0 && tid < 128) {                 // per-row statistics of the streamed queries (MODE 1-3)
      const int q = t * BKT + (tid & 63);
      if (tid < 64) sreg = q < NY ? lse2[(size_t)h * R + q] : INFINITY;    // masked query: exp2(-inf) = 0
      else sreg = q < NY ? delta[(size_t)h * R + q] : 0.f;
    }
  };
  auto store_tile = [&](int buf) {
    char* yb = smem + buf * 2 * BTILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(yb + st_off + 4096 * i) = yreg[i];
      *(u32x4*)(yb + BTILE + st_off + 4096 * i) = y2reg[i];
    }
    if (MODE != 0 && tid < 128) *(float*)(smem + BSTAT + buf * 512 + 4 * tid) = sreg;
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

  load_tile(0);
  store_tile(0);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    asm volatile("" : "+v"(xf[ks]));
    if (MODE != 2) asm volatile("" : "+v"(x2f[ks]));
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) load_tile(t + 1);
    const char* yb = smem + cur * 2 * BTILE;                   // Y image; Y2 image at + BTILE
    const char* zb = (MODE == 2) ? yb + BTILE : yb;            // image read transposed
    const unsigned zbase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)zb;
    const unsigned z2base = (unsigned)(uintptr_t)(MAVLM_LDS const char*)(yb + BTILE);   // MODE 3: dO image, transposed
    const float* stat = (const float*)(smem + BSTAT + cur * 512);
    const bool ragged = (t == nt - 1) && (NY & (BKT - 1));

#pragma unroll
    for (int b = 0; b < 2; ++b) {
      // ---- T^T = Y . X^T and (dQ, dK) dP^T = Y2 . X2^T for the 32 streamed rows of half b
      f32x16 tt, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { tt[i] = 0.f; dp[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const typename T::vec8 yf = *(const typename T::vec8*)(yb + yaddr[ks] + 8192 * b);
        tt = T::mfma32(yf, xf[ks], tt);
        if (MODE != 2) {
          const typename T::vec8 y2f = *(const typename T::vec8*)(yb + BTILE + yaddr[ks] + 8192 * b);
          dp = T::mfma32(y2f, x2f[ks], dp);
        }
      }
      // ---- E = P (dV) or dS = P o (dP - delta) (dQ, dK); streamed row of element i: 32b + (i&3) + 8(i>>2) + 4hh
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 l4, d4;
        if (MODE != 0) {
          l4 = *(const f32x4*)(stat + 32 * b + 8 * g + 4 * hh);
          if (MODE == 1 || MODE == 3) d4 = *(const f32x4*)(stat + 64 + 32 * b + 8 * g + 4 * hh);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * g + j;
          const float l = (MODE == 0) ? lse_l : l4[j];
          float p = __builtin_amdgcn_exp2f(tt[i] * c - l);
          if (MODE == 0 && ragged && t * BKT + 32 * b + j + 8 * g + 4 * hh >= NY) p = 0.f;   // key past the end
          if (MODE == 2) tt[i] = p;
          else tt[i] = p * (dp[i] - ((MODE == 0) ? del_l : d4[j]));
          if (MODE == 3) dp[i] = p;                                   // dP is consumed: its registers carry P
        }
      }
      typename T::vec8 ef[2], pf[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2<T>(tt[8 * s + 2 * j], tt[8 * s + 2 * j + 1]);
        ef[s] = __builtin_bit_cast(typename T::vec8, w);
        if (MODE == 3) {
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = pack2<T>(dp[8 * s + 2 * j], dp[8 * s + 2 * j + 1]);
          pf[s] = __builtin_bit_cast(typename T::vec8, w);
        }
      }
      // ---- A^T += Z^T . E^T
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const typename T::vec4 lo = T::ds_read_tr(zbase + zaddr[db][0] + 256 * (32 * b + 16 * s));
          const typename T::vec4 hi = T::ds_read_tr(zbase + zaddr[db][1] + 256 * (32 * b + 16 * s));
          const typename T::vec8 zf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc[db] = T::mfma32(zf, ef[s], acc[db]);
          if (MODE == 3) {
            const typename T::vec4 lo2 = T::ds_read_tr(z2base + zaddr[db][0] + 256 * (32 * b + 16 * s));
            const typename T::vec4 hi2 = T::ds_read_tr(z2base + zaddr[db][1] + 256 * (32 * b + 16 * s));
            const typename T::vec8 z2f = __builtin_shufflevector(lo2, hi2, 0, 1, 2, 3, 4, 5, 6, 7);
            acc2[db] = T::mfma32(z2f, pf[s], acc2[db]);
          }
        }
    }

    if (t + 1 < nt) store_tile(cur ^ 1);
    __syncthreads();
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
