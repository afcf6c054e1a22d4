// Kernels of the reference's INACTIVE variants (dead code in the shipped reference, kept for
// drop-in completeness - correctness first, no tuning):
//
//   frame_mean_kernel        v[f,:] = mean_p x[f,p,:]          per-frame pooling, bigru.py:50, segment.py:268
//   adjacent_cosine_kernel   s[i] = cos(v[i], v[i+1])          torch.cosine_similarity(eps), segment.py:33
//   gru_seq_kernel           the recurrent half of nn.GRU (one workgroup per direction; the input half
//                            W_ih x + b_ih of all time steps is ONE MFMA GEMM)          bigru.py:26-32,68
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

// grid (ceil(D/512), F); 4 waves split the patches, lane = 8 consecutive channels
template <typename T>
__global__ __launch_bounds__(256) void frame_mean_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out16,
                                                         float* __restrict__ out32, int P, int D) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int f = blockIdx.y;
  const int c0 = blockIdx.x * 512 + lane * 8;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c0 < D) {
    for (int p = w; p < P; p += 4) {
      const u16x8 v = *(const u16x8*)(x + ((size_t)f * P + p) * D + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += T::to_f32(v[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[w][lane * 8 + e] = s[e];
  __syncthreads();
  if (w == 0 && c0 < D) {
    const float inv = 1.0f / (float)P;
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = lane * 8 + e;
      const float m = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) * inv;
      o[e] = T::from_f32(m);
      if (out32) out32[(size_t)f * D + c0 + e] = m;
    }
    if (out16) *(u16x8*)(out16 + (size_t)f * D + c0) = o;
  }
}

// one wave per adjacent pair; fp32 input rows
__global__ __launch_bounds__(256) void adjacent_cosine_kernel(const float* __restrict__ v, float* __restrict__ out, int n,
                                                              int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n - 1) return;
  const float* a = v + (size_t)i * D;
  const float* b = a + D;
  float ab = 0.f, aa = 0.f, bb = 0.f;
  for (int j = lane; j < D; j += 64) {
    const float x = a[j], y = b[j];
    ab += x * y; aa += x * x; bb += y * y;
  }
  ab = wave_sum(ab); aa = wave_sum(aa); bb = wave_sum(bb);
  if (lane == 0) out[i] = ab / (fmaxf(sqrtf(aa), eps) * fmaxf(sqrtf(bb), eps));   // x.y / (max(|x|,eps) max(|y|,eps))
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// grid = directions; 1024 threads.  xg fp32 [F, ndir*3H] (W_ih x + b_ih, gate order r,z,n per direction),
// whh 16-bit [ndir][3H][H], bhh fp32 [ndir][3H]; out 16-bit [F, ndir*H].  H <= 512, H % 8 == 0.
template <typename T>
__global__ __launch_bounds__(1024) void gru_seq_kernel(const float* __restrict__ xg, const uint16_t* __restrict__ whh,
                                                       const float* __restrict__ bhh, uint16_t* __restrict__ out, int F,
                                                       int H, int ndir) {
  __shared__ float h[512];
  __shared__ float g[3 * 512];
  const int dir = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const uint16_t* W = whh + (size_t)dir * 3 * H * H;
  const float* bh = bhh + (size_t)dir * 3 * H;
  if (tid < 512) h[tid] = 0.f;
  __syncthreads();
  const int c0 = lane * 8;
  for (int step = 0; step < F; ++step) {
    const int t = dir == 0 ? step : F - 1 - step;
    float hv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) hv[e] = c0 + e < H ? h[c0 + e] : 0.f;
    for (int row = w; row < 3 * H; row += 16) {
      float s = 0.f;
      if (c0 < H) {
        const u16x8 wv = *(const u16x8*)(W + (size_t)row * H + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += T::to_f32(wv[e]) * hv[e];
      }
      s = wave_sum(s);
      if (lane == 0) g[row] = s + bh[row];
    }
    __syncthreads();
    if (tid < H) {
      const float* xr = xg + (size_t)t * ndir * 3 * H + (size_t)dir * 3 * H;
      const float r = sigmoidf_(xr[tid] + g[tid]);
      const float z = sigmoidf_(xr[H + tid] + g[H + tid]);
      const float n = tanhf(xr[2 * H + tid] + r * g[2 * H + tid]);
      const float hn = (1.0f - z) * n + z * h[tid];
      h[tid] = hn;
      out[(size_t)t * ndir * H + (size_t)dir * H + tid] = T::from_f32(hn);
    }
    __syncthreads();
  }
}

}  // namespace

hipError_t mavlm_launch_frame_mean(const void* x, void* out16, float* out32, int F, int P, int D, int dtype, hipStream_t s) {
  if (F <= 0 || P <= 0 || D <= 0 || (D & 7)) return hipErrorInvalidValue;
  mavlm_prof_scope prof(MAVLM_K_MISC, 0.0, 2.0 * F * (double)P * D, s);
  const dim3 grid((D + 511) / 512, F);
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(frame_mean_kernel<F16>, grid, dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)out16, out32, P, D);
  else
    hipLaunchKernelGGL(frame_mean_kernel<BF16>, grid, dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)out16, out32, P, D);
  return hipGetLastError();
}

hipError_t mavlm_launch_adjacent_cosine(const float* v, float* out, int n, int D, float eps, hipStream_t s) {
  if (n < 2 || D <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(adjacent_cosine_kernel, dim3((n - 1 + 3) / 4), dim3(256), 0, s, v, out, n, D, eps);
  return hipGetLastError();
}

hipError_t mavlm_launch_gru_seq(const float* xg, const void* whh, const float* bhh, void* out, int F, int H, int ndir,
                                int dtype, hipStream_t s) {
  if (F <= 0 || H <= 0 || H > 512 || (H & 7) || ndir < 1 || ndir > 2) return hipErrorInvalidValue;
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(gru_seq_kernel<F16>, dim3(ndir), dim3(1024), 0, s, xg, (const uint16_t*)whh, bhh, (uint16_t*)out, F, H, ndir);
  else
    hipLaunchKernelGGL(gru_seq_kernel<BF16>, dim3(ndir), dim3(1024), 0, s, xg, (const uint16_t*)whh, bhh, (uint16_t*)out, F, H, ndir);
  return hipGetLastError();
}
