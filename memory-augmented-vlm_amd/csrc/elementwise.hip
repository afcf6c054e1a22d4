// Row-wise kernels of the memory path (HBM-bound; 16-byte vector accesses, wave-level reductions).
//
//   layernorm_kernel  Residual's LayerNorm (llava/model/memory_module/MemoryController.py:24,28): fp32 row in
//                     (dense + bias + residual, written by the GEMM epilogue), 16-bit row out.
//   row_add_kernel    out[t,p,:] = x[src[t],p,:] + table[idx[t],:]  - temporal positional-encoding add
//                     (position_encoding.py:58,64), fine-frame gather + token-type add (llava_arch.py:513-524,554).
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

// one wave per row, NV float4 per lane (D <= 256*NV)
template <typename T, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const uint16_t* __restrict__ res,
                                                        int ldr, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ out,
                                                        int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = D >> 2;
  const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
  f32x4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = i * 64 + lane;
    if (j < nvec) {
      v[i] = xr[j];
      if (res != nullptr) {      // residual add of the Residual block, in fp32 (MemoryController.py:28)
        const u16x4 rv = *(const u16x4*)(res + (size_t)row * ldr + 4 * j);
        v[i][0] += T::to_f32(rv[0]); v[i][1] += T::to_f32(rv[1]); v[i][2] += T::to_f32(rv[2]); v[i][3] += T::to_f32(rv[3]);
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    } else {
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = i * 64 + lane;
    if (j < nvec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[i][e] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  uint16_t* orow = out + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int j = i * 64 + lane;
    if (j < nvec) {
      const f32x4 g = ((const f32x4*)gamma)[j];
      const f32x4 b = ((const f32x4*)beta)[j];
      *(u32x2*)(orow + 4 * j) = pack4<T>((v[i][0] - mean) * rstd * g[0] + b[0], (v[i][1] - mean) * rstd * g[1] + b[1],
                                         (v[i][2] - mean) * rstd * g[2] + b[2], (v[i][3] - mean) * rstd * g[3] + b[3]);
    }
  }
}

// D % 8 == 0: a lane owns chunks of 8 consecutive elements (two float4 in, ONE 16-byte residual load and ONE 16-byte
// store per chunk - the 4-element form above moves the 16-bit data 8 bytes per lane)
// SK: x = `splits` fp32 planes of a split-K GEMM ([splits][rows][D]) + bias, added in the order of splitk_reduce_kernel (planes in
// order, then the bias): the reduction pass and the LayerNorm in one kernel, same bits as the two
template <typename T, int NC, bool SK = false>
__global__ __launch_bounds__(256) void layernorm8_kernel(const float* __restrict__ x, const uint16_t* __restrict__ res,
                                                         int ldr, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, uint16_t* __restrict__ out,
                                                         int rows, int D, float eps, int splits = 1,
                                                         const float* __restrict__ bias = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = D >> 3;
  const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
  f32x4 v[NC][2];
  u32x4 rv[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 64 + lane;
    if (j < nch) {
      v[i][0] = xr[2 * j];
      v[i][1] = xr[2 * j + 1];
      if constexpr (SK) {
        const size_t plane4 = (size_t)rows * D / 4;
        for (int k = 1; k < splits; ++k) {
          const f32x4 a = xr[(size_t)k * plane4 + 2 * j], b = xr[(size_t)k * plane4 + 2 * j + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[i][0][e] += a[e]; v[i][1][e] += b[e]; }
        }
        if (bias != nullptr) {
          const f32x4 a = ((const f32x4*)bias)[2 * j], b = ((const f32x4*)bias)[2 * j + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[i][0][e] += a[e]; v[i][1][e] += b[e]; }
        }
      }
      if (res != nullptr) rv[i] = *(const u32x4*)(res + (size_t)row * ldr + 8 * j);
    } else {
      v[i][0] = v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 64 + lane;
    if (j < nch) {
      if (res != nullptr) {      // residual add of the Residual block, in fp32 (MemoryController.py:28)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e >> 1][2 * (e & 1)] += T::to_f32((uint16_t)(rv[i][e] & 0xffffu));
          v[i][e >> 1][2 * (e & 1) + 1] += T::to_f32((uint16_t)(rv[i][e] >> 16));
        }
      }
      s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 64 + lane;
    if (j < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e >> 2][e & 3] - mean;
        ss += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
  uint16_t* orow = out + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 64 + lane;
    if (j < nch) {
      u32x4 o;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x4 g = ((const f32x4*)gamma)[2 * j + hf];
        const f32x4 b = ((const f32x4*)beta)[2 * j + hf];
        const u32x2 w = pack4<T>((v[i][hf][0] - mean) * rstd * g[0] + b[0], (v[i][hf][1] - mean) * rstd * g[1] + b[1],
                                 (v[i][hf][2] - mean) * rstd * g[2] + b[2], (v[i][hf][3] - mean) * rstd * g[3] + b[3]);
        o[2 * hf] = w[0];
        o[2 * hf + 1] = w[1];
      }
      *(u32x4*)(orow + 8 * j) = o;
    }
  }
}

// Wide rows (D > 2048, e.g. 3584 at the OneVision-7B width): ONE ROW PER WORKGROUP instead of per wave.  With a wave per row a
// lane holds 7-8 chunks (~110 VGPRs, 4 waves per SIMD) and 6 272 rows are 1.5 rounds of the chip's wave slots: 2.4 TB/s
// algorithmic.  Four waves per row hold <= 2 chunks per lane (~40 VGPRs); the two row reductions go through LDS (fixed order:
// lanes by butterfly, then the four waves in order).
template <typename T, int NC, bool SK = false>
__global__ __launch_bounds__(256) void layernorm8_block_kernel(const float* __restrict__ x, const uint16_t* __restrict__ res,
                                                               int ldr, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, uint16_t* __restrict__ out,
                                                               int rows, int D, float eps, int splits = 1,
                                                               const float* __restrict__ bias = nullptr) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int row = blockIdx.x;
  const int nch = D >> 3;
  const f32x4* xr = (const f32x4*)(x + (size_t)row * D);
  f32x4 v[NC][2];
  u32x4 rv[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 256 + tid;
    if (j < nch) {
      v[i][0] = xr[2 * j];
      v[i][1] = xr[2 * j + 1];
      if constexpr (SK) {
        const size_t plane4 = (size_t)rows * D / 4;
        for (int k = 1; k < splits; ++k) {
          const f32x4 a = xr[(size_t)k * plane4 + 2 * j], b = xr[(size_t)k * plane4 + 2 * j + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[i][0][e] += a[e]; v[i][1][e] += b[e]; }
        }
        if (bias != nullptr) {
          const f32x4 a = ((const f32x4*)bias)[2 * j], b = ((const f32x4*)bias)[2 * j + 1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[i][0][e] += a[e]; v[i][1][e] += b[e]; }
        }
      }
      if (res != nullptr) rv[i] = *(const u32x4*)(res + (size_t)row * ldr + 8 * j);
    } else {
      v[i][0] = v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 256 + tid;
    if (j < nch) {
      if (res != nullptr) {      // residual add of the Residual block, in fp32 (MemoryController.py:28)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e >> 1][2 * (e & 1)] += T::to_f32((uint16_t)(rv[i][e] & 0xffffu));
          v[i][e >> 1][2 * (e & 1) + 1] += T::to_f32((uint16_t)(rv[i][e] >> 16));
        }
      }
      s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[0][wave] = s;
  __syncthreads();
  const float mean = (((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]) / (float)D;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 256 + tid;
    if (j < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[i][e >> 2][e & 3] - mean;
        ss += d * d;
      }
    }
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[1][wave] = ss;
  __syncthreads();
  const float rstd = rsqrtf((((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]) / (float)D + eps);
  uint16_t* orow = out + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int j = i * 256 + tid;
    if (j < nch) {
      u32x4 o;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const f32x4 g = ((const f32x4*)gamma)[2 * j + hf];
        const f32x4 b = ((const f32x4*)beta)[2 * j + hf];
        const u32x2 w = pack4<T>((v[i][hf][0] - mean) * rstd * g[0] + b[0], (v[i][hf][1] - mean) * rstd * g[1] + b[1],
                                 (v[i][hf][2] - mean) * rstd * g[2] + b[2], (v[i][hf][3] - mean) * rstd * g[3] + b[3]);
        o[2 * hf] = w[0];
        o[2 * hf + 1] = w[1];
      }
      *(u32x4*)(orow + 8 * j) = o;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void row_add_kernel(const uint16_t* __restrict__ x, const int64_t* __restrict__ src,
                                                      const uint16_t* __restrict__ table,
                                                      const int64_t* __restrict__ idx, uint16_t* __restrict__ out,
                                                      int T_, int P, int D) {
  // one thread = 8 elements; grid-stride over T*P*D/8 vectors
  const int dv = D >> 3;
  const size_t total = (size_t)T_ * P * dv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dv);
    const size_t rowi = i / dv;
    const int t = (int)(rowi / P);
    const int p = (int)(rowi % P);
    const int64_t st = src ? src[t] : t;
    const int64_t it = idx ? idx[t] : 0;
    const u16x8 a = *(const u16x8*)(x + ((size_t)st * P + p) * D + 8 * c);
    const u16x8 b = *(const u16x8*)(table + (size_t)it * D + 8 * c);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = T::from_f32(T::to_f32(a[e]) + T::to_f32(b[e]));
    *(u16x8*)(out + rowi * D + 8 * c) = o;
  }
}

// row_add_kernel for the videos of a row batch in ONE launch (blockIdx.y = video): the fine-frame gather + token-type add of
// mavlm_fuse_emit_batch (llava_arch.py:620-629), one source pointer per video, the videos' outputs vstride elements apart
struct row_add_srcs { const uint16_t* p[16]; };
template <typename T>
__global__ __launch_bounds__(256) void row_add_batch_kernel(row_add_srcs xs, const int64_t* __restrict__ src,
                                                            const uint16_t* __restrict__ table, uint16_t* __restrict__ out,
                                                            long long vstride, int T_, int P, int D) {
  const uint16_t* __restrict__ x = xs.p[blockIdx.y];
  out += (size_t)blockIdx.y * vstride;
  const int dv = D >> 3;
  const size_t total = (size_t)T_ * P * dv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % dv);
    const size_t rowi = i / dv;
    const int t = (int)(rowi / P);
    const int p = (int)(rowi % P);
    const int64_t st = src ? src[t] : t;
    const u16x8 a = *(const u16x8*)(x + ((size_t)st * P + p) * D + 8 * c);
    const u16x8 b = *(const u16x8*)(table + 8 * c);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = T::from_f32(T::to_f32(a[e]) + T::to_f32(b[e]));
    *(u16x8*)(out + rowi * D + 8 * c) = o;
  }
}

// The literal rows of a video's token block (llava_arch.py:541-543,559-566: memory prompt, image_newline, frame prompt,
// image_newline) - up to four short row runs copied to fixed rows of every video's block in ONE launch.
struct copy_rows_args {
  const uint16_t* src[4];
  int n[4];            // rows of run i (0 = unused)
  long long dst[4];    // first row of run i inside a video's block
};
__global__ __launch_bounds__(256) void copy_rows_kernel(copy_rows_args a, uint16_t* __restrict__ out, long long vstride, int D) {
  int r = blockIdx.x, i = 0;
  while (i < 3 && r >= a.n[i]) r -= a.n[i++];
  const u16x8* s = (const u16x8*)(a.src[i] + (size_t)r * D);
  u16x8* d = (u16x8*)(out + (size_t)blockIdx.y * vstride + (size_t)(a.dst[i] + r) * D);
  for (int c = threadIdx.x; c < (D >> 3); c += 256) d[c] = s[c];
}

// get_2dPool, bilinear branch (llava/model/llava_arch.py:277-297): F.interpolate(size=ceil(side/stride), mode='bilinear',
// align_corners=False) over the [side, side] token grid, channels last.  One workgroup per output token, 8 channels
// per lane (16-B loads of the 4 source tokens), fp32 lerp, ONE rounding to 16 bits; optionally followed by the
// temporal PE add (position_encoding.py:58,64) with its own rounding, exactly as the two reference ops round.
template <typename T>
__global__ __launch_bounds__(128) void pool_bilinear_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                            const uint16_t* __restrict__ table,
                                                            const int64_t* __restrict__ idx, int side, int oside, int D,
                                                            float scale) {
  const int f = blockIdx.y;
  const int o = blockIdx.x;
  const int oy = o / oside, ox = o % oside;
  // area_pixel_compute_source_index(scale, dst, align_corners=false, cubic=false): max(scale*(dst+0.5)-0.5, 0)
  const float sy = fmaxf(scale * (oy + 0.5f) - 0.5f, 0.f), sx = fmaxf(scale * (ox + 0.5f) - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < side - 1), x1 = x0 + (x0 < side - 1);
  const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
  const uint16_t* base = x + (size_t)f * side * side * D;
  const uint16_t* p00 = base + (size_t)(y0 * side + x0) * D;
  const uint16_t* p01 = base + (size_t)(y0 * side + x1) * D;
  const uint16_t* p10 = base + (size_t)(y1 * side + x0) * D;
  const uint16_t* p11 = base + (size_t)(y1 * side + x1) * D;
  uint16_t* op = out + ((size_t)f * oside * oside + o) * D;
  const uint16_t* pe = table ? table + (size_t)idx[f] * D : nullptr;
  for (int c = threadIdx.x * 8; c < D; c += blockDim.x * 8) {
    const u16x8 a = *(const u16x8*)(p00 + c), b = *(const u16x8*)(p01 + c), cc = *(const u16x8*)(p10 + c),
                d = *(const u16x8*)(p11 + c);
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = hy * (hx * T::to_f32(a[e]) + lx * T::to_f32(b[e])) + ly * (hx * T::to_f32(cc[e]) + lx * T::to_f32(d[e]));
      r[e] = T::from_f32(v);
    }
    if (pe) {
      const u16x8 q = *(const u16x8*)(pe + c);
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = T::from_f32(T::to_f32(r[e]) + T::to_f32(q[e]));
    }
    *(u16x8*)(op + c) = r;
  }
}

template <typename T, int NV>
void ln_launch(const float* x, const void* res, int ldr, const float* g, const float* b, void* out, int rows, int D,
               float eps, hipStream_t s) {
  hipLaunchKernelGGL((layernorm_kernel<T, NV>), dim3((rows + 3) / 4), dim3(256), 0, s, x, (const uint16_t*)res, ldr, g, b,
                     (uint16_t*)out, rows, D, eps);
}

template <typename T, int NC>
void ln8_launch(const float* x, const void* res, int ldr, const float* g, const float* b, void* out, int rows, int D,
                float eps, hipStream_t s) {
  hipLaunchKernelGGL((layernorm8_kernel<T, NC>), dim3((rows + 3) / 4), dim3(256), 0, s, x, (const uint16_t*)res, ldr, g, b,
                     (uint16_t*)out, rows, D, eps);
}

template <typename T>
hipError_t ln_dispatch(const float* x, const void* res, int ldr, const float* g, const float* b, void* out, int rows, int D,
                       float eps, hipStream_t s) {
  if ((D & 7) == 0 && (ldr & 7) == 0 && D <= 4096) {       // 16-byte accesses to the 16-bit data
    const int nc = (D / 8 + 63) / 64;
    if (nc <= 1) ln8_launch<T, 1>(x, res, ldr, g, b, out, rows, D, eps, s);
    else if (nc <= 2) ln8_launch<T, 2>(x, res, ldr, g, b, out, rows, D, eps, s);
    else if (nc <= 4) ln8_launch<T, 4>(x, res, ldr, g, b, out, rows, D, eps, s);
    else       // D > 2048: a workgroup per row (<= 2 chunks per lane; the wave-per-row form needs 246 VGPRs there: 1 wave per SIMD)
      hipLaunchKernelGGL((layernorm8_block_kernel<T, 2>), dim3(rows), dim3(256), 0, s, x, (const uint16_t*)res, ldr, g, b,
                         (uint16_t*)out, rows, D, eps);
    return hipGetLastError();
  }
  const int nv = (D / 4 + 63) / 64;
  if (nv <= 1) ln_launch<T, 1>(x, res, ldr, g, b, out, rows, D, eps, s);
  else if (nv <= 2) ln_launch<T, 2>(x, res, ldr, g, b, out, rows, D, eps, s);
  else if (nv <= 4) ln_launch<T, 4>(x, res, ldr, g, b, out, rows, D, eps, s);
  else if (nv <= 8) ln_launch<T, 8>(x, res, ldr, g, b, out, rows, D, eps, s);
  else if (nv <= 16) ln_launch<T, 16>(x, res, ldr, g, b, out, rows, D, eps, s);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace

hipError_t mavlm_launch_layernorm(const float* x, const void* res, int ldr, const float* gamma, const float* beta,
                                  void* out, int rows, int D, float eps, int dtype, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (!x || !gamma || !beta || !out || D <= 0 || (D & 3) || (res && (ldr & 3))) return hipErrorInvalidValue;
  mavlm_prof_scope prof(MAVLM_K_LN, 0.0, (res ? 8.0 : 6.0) * rows * (double)D, s);
  return dtype == MAVLM_F16 ? ln_dispatch<F16>(x, res, ldr, gamma, beta, out, rows, D, eps, s)
                            : ln_dispatch<BF16>(x, res, ldr, gamma, beta, out, rows, D, eps, s);
}

hipError_t mavlm_launch_layernorm_planes(const float* planes, int splits, const float* bias, const void* res, int ldr,
                                         const float* gamma, const float* beta, void* out, int rows, int D, float eps, int dtype,
                                         hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  if (!planes || !gamma || !beta || !out || splits < 1 || D <= 0 || (D & 7) || D > 4096 || (res && (ldr & 7))) return hipErrorInvalidValue;
  mavlm_prof_scope prof(MAVLM_K_LN, 0.0, (res ? 4.0 : 2.0) * rows * (double)D + 4.0 * splits * rows * (double)D, s);
  const int nc = (D / 8 + 63) / 64;
#define LNP(TT)                                                                                                              \
  do {                                                                                                                       \
    if (nc <= 1) hipLaunchKernelGGL((layernorm8_kernel<TT, 1, true>), dim3((rows + 3) / 4), dim3(256), 0, s, planes, (const uint16_t*)res, ldr, gamma, beta, (uint16_t*)out, rows, D, eps, splits, bias); \
    else if (nc <= 2) hipLaunchKernelGGL((layernorm8_kernel<TT, 2, true>), dim3((rows + 3) / 4), dim3(256), 0, s, planes, (const uint16_t*)res, ldr, gamma, beta, (uint16_t*)out, rows, D, eps, splits, bias); \
    else if (nc <= 4) hipLaunchKernelGGL((layernorm8_kernel<TT, 4, true>), dim3((rows + 3) / 4), dim3(256), 0, s, planes, (const uint16_t*)res, ldr, gamma, beta, (uint16_t*)out, rows, D, eps, splits, bias); \
    else hipLaunchKernelGGL((layernorm8_block_kernel<TT, 2, true>), dim3(rows), dim3(256), 0, s, planes, (const uint16_t*)res, ldr, gamma, beta, (uint16_t*)out, rows, D, eps, splits, bias); \
  } while (0)
  if (dtype == MAVLM_F16) LNP(F16); else LNP(BF16);
#undef LNP
  return hipGetLastError();
}

hipError_t mavlm_launch_row_add(const void* x, const int64_t* src, const void* table, const int64_t* idx, void* out,
                                int T_, int P, int D, int dtype, hipStream_t s) {
  if (T_ <= 0) return hipSuccess;
  if (!x || !table || !out || P <= 0 || D <= 0 || (D & 7)) return hipErrorInvalidValue;
  const size_t total = (size_t)T_ * P * (D >> 3);
  mavlm_prof_scope prof(MAVLM_K_ROWADD, 0.0, 4.0 * T_ * (double)P * D, s);
  size_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(row_add_kernel<F16>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)x, src,
                       (const uint16_t*)table, idx, (uint16_t*)out, T_, P, D);
  else
    hipLaunchKernelGGL(row_add_kernel<BF16>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)x, src,
                       (const uint16_t*)table, idx, (uint16_t*)out, T_, P, D);
  return hipGetLastError();
}

hipError_t mavlm_launch_row_add_batch(const void* const* x, const int64_t* src, const void* table_row, void* out,
                                      long long vstride, int B, int T_, int P, int D, int dtype, hipStream_t s) {
  if (T_ <= 0 || B <= 0) return hipSuccess;
  if (!x || !table_row || !out || P <= 0 || D <= 0 || (D & 7)) return hipErrorInvalidValue;
  const size_t total = (size_t)T_ * P * (D >> 3);
  mavlm_prof_scope prof(MAVLM_K_ROWADD, 0.0, 4.0 * B * T_ * (double)P * D, s);
  size_t blocks = (total + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  for (int b0 = 0; b0 < B; b0 += 16) {
    const int nb = B - b0 < 16 ? B - b0 : 16;
    row_add_srcs xs = {};
    for (int i = 0; i < nb; ++i) {
      if (!x[b0 + i]) return hipErrorInvalidValue;
      xs.p[i] = (const uint16_t*)x[b0 + i];
    }
    uint16_t* o = (uint16_t*)out + (size_t)b0 * vstride;
    if (dtype == MAVLM_F16)
      hipLaunchKernelGGL(row_add_batch_kernel<F16>, dim3((unsigned)blocks, (unsigned)nb), dim3(256), 0, s, xs, src,
                         (const uint16_t*)table_row, o, vstride, T_, P, D);
    else
      hipLaunchKernelGGL(row_add_batch_kernel<BF16>, dim3((unsigned)blocks, (unsigned)nb), dim3(256), 0, s, xs, src,
                         (const uint16_t*)table_row, o, vstride, T_, P, D);
  }
  return hipGetLastError();
}

hipError_t mavlm_launch_copy_rows(const void* const* src, const int* n, const long long* dst, int runs, void* out,
                                  long long vstride, int B, int D, hipStream_t s) {
  if (runs < 0 || runs > 4 || !out || B <= 0 || D <= 0 || (D & 7)) return hipErrorInvalidValue;
  copy_rows_args a = {};
  int total = 0, k = 0;
  for (int i = 0; i < runs; ++i) {
    if (n[i] <= 0) continue;
    if (!src[i]) return hipErrorInvalidValue;
    a.src[k] = (const uint16_t*)src[i]; a.n[k] = n[i]; a.dst[k] = dst[i];
    total += n[i];
    ++k;
  }
  if (!total) return hipSuccess;
  mavlm_prof_scope prof(MAVLM_K_ROWADD, 0.0, 4.0 * total * (double)B * D, s);
  hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)total, (unsigned)B), dim3(256), 0, s, a, (uint16_t*)out, vstride, D);
  return hipGetLastError();
}

hipError_t mavlm_launch_pool_bilinear(const void* x, void* out, const void* table, const int64_t* idx, int F, int side,
                                      int stride, int D, int dtype, hipStream_t s) {
  if (F <= 0) return hipSuccess;
  if (!x || !out || side <= 0 || stride <= 0 || D <= 0 || (D & 7) || (table && !idx)) return hipErrorInvalidValue;
  const int oside = (side + stride - 1) / stride;
  const float scale = (float)side / (float)oside;
  mavlm_prof_scope prof(MAVLM_K_ROWADD, 0.0, 2.0 * F * ((double)side * side + oside * oside) * D, s);
  dim3 grid(oside * oside, F);
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(pool_bilinear_kernel<F16>, grid, dim3(128), 0, s, (const uint16_t*)x, (uint16_t*)out,
                       (const uint16_t*)table, idx, side, oside, D, scale);
  else
    hipLaunchKernelGGL(pool_bilinear_kernel<BF16>, grid, dim3(128), 0, s, (const uint16_t*)x, (uint16_t*)out,
                       (const uint16_t*)table, idx, side, oside, D, scale);
  return hipGetLastError();
}
