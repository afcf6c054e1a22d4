// Common device helpers for the MI355X (gfx950 / CDNA4) memory-path kernels.
// Wave = 64 lanes everywhere; MFMA fragment maps per cdna_hip_programming.md §3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One-time kernel attributes (hipFuncAttributeMaxDynamicSharedMemorySize) are per DEVICE, not per process: a process
// that drives a second GPU must set them there as well, or its 64-128 KiB-LDS kernels fail to launch on that device.
struct mavlm_per_device_once {
  // one bit per device ordinal (256 ordinals).  Atomic: a host may drive the library from several threads (one per device
  // or per stream); setting the attribute twice is harmless, a torn read-modify-write of the mask is not.
  unsigned long long seen[4] = {0, 0, 0, 0};
  // returns hipSuccess once `bytes` of dynamic LDS are allowed for `fn` on the CURRENT device
  hipError_t dyn_lds(const void* fn, int bytes) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (d & 63);
    unsigned long long* word = &seen[(d >> 6) & 3];
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(word, bit, __ATOMIC_RELEASE);
    return e;
  }
};


typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;

#define MAVLM_LDS __attribute__((address_space(3)))
#define MAVLM_GLOBAL __attribute__((address_space(1)))

enum { MAVLM_BF16 = 0, MAVLM_F16 = 1 };

// 16-bit element traits: storage is always uint16_t; arithmetic is fp32.
struct BF16 {
  using vec8 = bf16x8;
  using vec4 = bf16x4;
  static __device__ __forceinline__ float to_f32(uint16_t u) {
    return __builtin_bit_cast(float, (unsigned)u << 16);
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {   // RNE, v_cvt_pk_bf16_f32
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  }
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ vec4 ds_read_tr(unsigned lds_byte_off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MAVLM_LDS vec4*)(uintptr_t)lds_byte_off);
  }
};

struct F16 {
  using vec8 = f16x8;
  using vec4 = f16x4;
  static __device__ __forceinline__ float to_f32(uint16_t u) {
    return (float)__builtin_bit_cast(_Float16, u);
  }
  static __device__ __forceinline__ uint16_t from_f32(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
  static __device__ __forceinline__ f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ vec4 ds_read_tr(unsigned lds_byte_off) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 h4;
    return __builtin_bit_cast(vec4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((MAVLM_LDS h4*)(uintptr_t)lds_byte_off));
  }
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// two fp32 -> one dword of two 16-bit values (lo = a), one v_cvt_pk_* instruction
template <typename T>
__device__ __forceinline__ unsigned pack2(float a, float b);
template <>
__device__ __forceinline__ unsigned pack2<BF16>(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
template <>
__device__ __forceinline__ unsigned pack2<F16>(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
}

template <typename T>
__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
  u32x2 r;
  r[0] = pack2<T>(a, b);
  r[1] = pack2<T>(c, d);
  return r;
}

// max / sum of a value across the two 32-lane halves of a wave (lane l <-> lane l^32), no LDS round trip
// v_permlane32_swap vdst, src: lanes 32-63 of vdst <-> lanes 0-31 of src.  Fed two copies of v it leaves
// {own | other-half} in one register and {other-half | own} in the other, so max / sum of the two is the cross-half
// reduction on every lane (no LDS round trip, unlike __shfl_xor(v, 32) = ds_bpermute).
// Inline asm on purpose: with __builtin_amdgcn_permlane32_swap, hipcc (ROCm 7.2, -O3) folds the SECOND result into
// the first (InstCombine; correct at -O0) - measured as wrong softmax row sums.  The 2 wait states the ISA wants
// between a VALU write of an operand and the swap are inside the string (hipcc pads nothing inside asm).
__device__ __forceinline__ void permlane32_swap(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float xhalf_max(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xhalf_sum(float v) {
  float a = v, b = v;
  permlane32_swap(a, b);
  return a + b;
}

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so
// give each XCD a contiguous range of logical tiles (cdna_hip_programming.md T1).  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}

// GELU(x) = x * Phi(x) with the exact-erf definition (nn.GELU() default, llava_arch.py:134).  Phi through the
// complementary error function in Abramowitz-Stegun 7.1.26 form, q = 0.5 erfc(a) ~ 0.5 (a1 t + ... + a5 t^5) exp(-a^2),
// a = |x| / sqrt2, t = 1/(1 + p a): |erf error| <= 6e-7 in fp32, |GELU error| <= 3.5e-7 absolute (checked against float64 on a
// dense grid, |x| <= 8) - four orders below the 16-bit output grid.  GELU(x) = max(x, 0) - |x| q (x < 0: x q; x >= 0: x - x q): no
// compare / select.  Round 4: constants folded so that the whole evaluation is 2 transcendentals (rcp, exp2) + 9 fma / mul
// + abs + max, written with explicit fma so that the scalar form and the PACKED form (two values per v_pk_fma_f32 /
// v_pk_mul_f32: the epilogues are VALU-bound, packed math halves the non-transcendental part) give the same bits:
//   a' = |x| k, k = sqrt(log2(e) / 2)  ->  exp(-a^2) = exp2(-a'^2);  t = 1 / (1 + (p / sqrt(log2 e)) a');
//   q' = q / k = t (b1 + t (b2 + ...)) exp2(-a'^2), b_i = 0.5 a_i / k;  GELU = fma(-a', q', max(x, 0)).
#define MAVLM_GELU_K 0.8493218002880191f        /* sqrt(log2(e) / 2) */
#define MAVLM_GELU_P 0.2727374808792225f        /* 0.3275911 / sqrt(log2 e) */
#define MAVLM_GELU_B1 0.1500194578271646f          /* 0.5 * 0.254829592 / k */
#define MAVLM_GELU_B2 (-0.1674846541696695f)       /* 0.5 * -0.284496736 / k */
#define MAVLM_GELU_B3 0.8367933923973075f          /* 0.5 * 1.421413741 / k */
#define MAVLM_GELU_B4 (-0.8554778804142388f)       /* 0.5 * -1.453152027 / k */
#define MAVLM_GELU_B5 0.6248546950284685f          /* 0.5 * 1.061405429 / k */
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float a = fabsf(x) * MAVLM_GELU_K;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(a, MAVLM_GELU_P, 1.0f));
  float p = __builtin_fmaf(t, MAVLM_GELU_B5, MAVLM_GELU_B4);
  p = __builtin_fmaf(t, p, MAVLM_GELU_B3);
  p = __builtin_fmaf(t, p, MAVLM_GELU_B2);
  p = __builtin_fmaf(t, p, MAVLM_GELU_B1);
  p = t * p;
  const float e = __builtin_amdgcn_exp2f(-(a * a));
  return __builtin_fmaf(-a, p * e, fmaxf(x, 0.f));
}
// the same, two values per instruction (bit-identical per element to gelu_erf_fast)
__device__ __forceinline__ f32x2 gelu_erf_fast2(f32x2 x) {
  const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2 a = ax * MAVLM_GELU_K;
  const f32x2 d = __builtin_elementwise_fma(a, (f32x2)(MAVLM_GELU_P), (f32x2)(1.0f));
  const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2 p = __builtin_elementwise_fma(t, (f32x2)(MAVLM_GELU_B5), (f32x2)(MAVLM_GELU_B4));
  p = __builtin_elementwise_fma(t, p, (f32x2)(MAVLM_GELU_B3));
  p = __builtin_elementwise_fma(t, p, (f32x2)(MAVLM_GELU_B2));
  p = __builtin_elementwise_fma(t, p, (f32x2)(MAVLM_GELU_B1));
  p = t * p;
  const f32x2 na2 = -(a * a);
  const f32x2 e = {__builtin_amdgcn_exp2f(na2[0]), __builtin_amdgcn_exp2f(na2[1])};
  const f32x2 r = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
  return __builtin_elementwise_fma(-a, p * e, r);
}
__device__ __forceinline__ f32x4 gelu_erf_fast4(f32x4 x) {
  const f32x2 lo = gelu_erf_fast2(f32x2{x[0], x[1]}), hi = gelu_erf_fast2(f32x2{x[2], x[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}

// Widened 16-bit epilogue store (after cdna_hip_programming.md T21, with the 16-lane-row swap):
// lane group fq = lane>>4 holds 4 consecutive 16-bit outputs (2 dwords) of column block j in `a` and of block j+1 in
// `b`.  v_permlane16_swap exchanges the ODD rows of its first operand with the EVEN rows of its second, after which
// {a, b} of every lane are 8 CONSECUTIVE outputs (16 bytes): fq 0 -> block j cols 0-7, fq 1 -> block j+1 cols 0-7,
// fq 2 -> block j cols 8-15, fq 3 -> block j+1 cols 8-15.  One 16-byte store per lane replaces two 8-byte stores.
__device__ __forceinline__ u32x4 widen_pair(u32x2 a, u32x2 b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a[0]), "+v"(b[0]));
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a[1]), "+v"(b[1]));
  u32x4 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
  return r;
}

// max(a, b, c) in ONE instruction.  fmaxf() on MFMA results makes hipcc emit a canonicalising v_max(x, x) per
// operand first (3 instructions for what v_max3_f32 does in one); the asm form has no such prologue.
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
