// Fused multi-head cross-attention for the memory path, head_dim = 128, no mask, no dropout.
//
// Replaces the materialised score tensor of the reference `Attention.forward`
// (llava/model/memory_module/MemoryController.py:51-54: matmul / sqrt(d) -> softmax -> matmul -> merge heads)
// with a flash-style kernel: the [H,R,S] probabilities never touch HBM.
//
//   attn_fwd_kernel     ctx[q, h*128+d] = sum_k softmax_k(q.k/sqrt(d)) v[k, d]       (+ lse2[h][q])
//   attn_colsum_kernel  part[h][k] = sum_q exp2(s[h,q,k]*c - lse2[h][q])             (second pass; only the
//                       last formation layer needs it: MemoryController.py:135-139 frame scores)
//
// MFMA mapping (v_mfma_f32_32x32x16, cdna_hip_programming.md §3):
//   S^T[key][query] = K . Q^T      A = K rows from LDS (ds_read_b128), B = Q rows held in registers.
//     -> every lane owns ONE query column (lane&31) and 16 of the 32 keys: the softmax row reduction is
//        in-register plus one cross-half exchange.
//   O^T[d][query]   = V^T . P^T    B = the S^T accumulator registers converted to 16-bit in place
//     ("accumulator tile as the next MFMA's operand"), A = V^T read with ds_read_b64_tr_b16 from the
//     row-major V image.  The query stays on the lane, so the online-softmax rescale of O is lane-local.
//
// LDS image for K, V (and Q in the colsum pass): [64 rows][128 x 16-bit], 256-B rows, 16-B chunk `ch` of row
// `row` at byte 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3))) - conflict-free for both the b128 row reads
// and the transposed reads of the 32x32x16 operands (cdna_hip_programming.md T10, image (b)).
//
// Block = 4 waves x 32 queries; KV tile = 64 keys; K/V staged through registers (issue the global loads for
// tile t+1 before computing tile t, write them to the other LDS stage afterwards; one barrier per tile).
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

constexpr int HD = 128;             // head dim
constexpr int KT = 64;              // keys per tile
constexpr int TILE = KT * HD * 2;   // 16 KiB
constexpr int ATTN_LDS = 4 * TILE;  // 2 stages x (K,V)
constexpr int CS_LDS = 2 * TILE;    // 2 stages x Q
constexpr float RESCALE_LOG2 = 8.0f;   // deferred-rescale threshold in log2 units (mirrored by the oracle)

__device__ __forceinline__ int img_x(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// Diagnostic build only (-DMAVLM_ATTN_STAMPS): per-phase cycle sums of every wave, read back with
// mavlm_debug_read_stamps().  No stamp executes in the product build.
#ifdef MAVLM_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[8 * 4096];
#define STAMP(var)                                                                  \
  unsigned long long var;                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");        \
  __builtin_amdgcn_sched_barrier(0);
#else
#define STAMP(var)
#endif

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const uint16_t* __restrict__ Q, int ldq,
                                                          const uint16_t* __restrict__ K, int ldk,
                                                          const uint16_t* __restrict__ V, int ldv,
                                                          uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2,
                                                          int R, int S, int H, float c /* scale*log2(e) */) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid: head = bid % H.  Blocks b and b+8 share an XCD (round-robin dispatch), so with H = 8 every XCD's L2
  // serves the K/V of ONE head (3.2 MB at S = 6272) instead of all of them.  Speed only.
  const int h = blockIdx.x % H;
  const int q0 = (blockIdx.x / H) * 128 + wave * 32;
  const int r = lane & 31, hh = lane >> 5;

  // ---- Q fragments: B operand, lane holds Q[q0+r][h*128 + 16ks + 8hh + 0..7]
  typename T::vec8 qf[8];
  {
    int qrow = q0 + r;
    qrow = qrow < R ? qrow : R - 1;
    const uint16_t* qp = Q + (size_t)qrow * ldq + h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const typename T::vec8*)(qp + 16 * ks);
  }

  // ---- staging geometry: thread handles chunks id = tid + 256 i -> row = (tid>>4) + 16 i, ch = tid & 15
  const int srow = tid >> 4, sch = tid & 15;
  const int st_off = 256 * srow + 16 * (sch ^ img_x(srow));   // + 4096 i
  const uint16_t* kg = K + h * HD + sch * 8;
  const uint16_t* vg = V + h * HD + sch * 8;
  u32x4 kreg[4], vreg[4];

  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = t * KT + srow + 16 * i;
      row = row < S ? row : S - 1;
      kreg[i] = *(const u32x4*)(kg + (size_t)row * ldk);
      vreg[i] = *(const u32x4*)(vg + (size_t)row * ldv);
    }
  };
  auto store_tile = [&](int buf) {
    char* kb = smem + buf * 2 * TILE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(kb + st_off + 4096 * i) = kreg[i];
      *(u32x4*)(kb + TILE + st_off + 4096 * i) = vreg[i];
    }
  };

  // ---- fragment read geometry
  const int xr = img_x(r);
  const int k_rd = 256 * r;                             // + 8192 b + 16*((2ks+hh) ^ xr)
  // V^T transposed read: lane i=4q+p of its 16-lane group g; rows r0+q, r0 = 32b+16s+8jj+4hh
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  // byte = 256*(32b+16s+8jj+4hh+tq) + 16*(((db^tq)<<2) | ((tg1^jj)<<1) | ((tp>>1)^hh)) + 8*(tp&1)
  const int v_rd = 256 * (4 * hh + tq) + 8 * (tp & 1) + 16 * ((tp >> 1) ^ hh);

  f32x16 ot[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;

  const int nt = (S + KT - 1) / KT;
#ifdef MAVLM_ATTN_STAGGER
  // experiment: de-phase the two waves that share a SIMD (they come from two co-resident workgroups running the
  // same program and otherwise fall into lockstep: both in the MFMA phase, then both in the VALU phase)
  if (__builtin_amdgcn_s_getreg(0x1804) & 1) __builtin_amdgcn_s_sleep(MAVLM_ATTN_STAGGER);   // HW_ID.wave_id bit 0
#endif
  load_tile(0);
  store_tile(0);
  // Pin the loop-invariant Q fragments in registers BEFORE the loop: left alone, hipcc sinks their loads to the
  // loop entry and then carries counted vmcnt waits for them into the QK^T chain, where (vmcnt counts in issue
  // order) they end up waiting for the NEXT tile's loads on every iteration.
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
  __syncthreads();

#ifdef MAVLM_ATTN_STAMPS
  unsigned long long acc_qk = 0, acc_sm = 0, acc_pv = 0, acc_st = 0, acc_bar = 0;
#endif
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    STAMP(t0)
    if (t + 1 < nt) load_tile(t + 1);     // lands during this tile's compute; written to LDS at the end
    const char* kb = smem + cur * 2 * TILE;
    const unsigned vb = (unsigned)(uintptr_t)(MAVLM_LDS const char*)(kb + TILE);   // LDS byte address of the V image

    // ---- S^T = K . Q^T  (2 key blocks x 8 k-steps).  Step i = (ks = i>>1, b = i&1): consecutive MFMAs hit different
    // accumulators, and the K fragments are read KPF steps ahead so that several ds_read_b128 are in flight
    // (one read in flight exposes the LDS latency on every MFMA).
    f32x16 st[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) st[b][i] = 0.f;
    {
      constexpr int KPF = 4;
      typename T::vec8 kfr[16];
#pragma unroll
      for (int i = 0; i < KPF; ++i)
        kfr[i] = *(const typename T::vec8*)(kb + k_rd + 8192 * (i & 1) + 16 * ((2 * (i >> 1) + hh) ^ xr));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + KPF < 16) {
          const int j = i + KPF;
          kfr[j] = *(const typename T::vec8*)(kb + k_rd + 8192 * (j & 1) + 16 * ((2 * (j >> 1) + hh) ^ xr));
        }
        st[i & 1] = T::mfma32(kfr[i], qf[i >> 1], st[i & 1]);
        __builtin_amdgcn_sched_barrier(0);   // keep the read-ahead distance: hipcc otherwise sinks the reads to their use
      }
    }

    STAMP(t1)
    // ---- mask the ragged tail of the last tile: key = 64t + 32b + (i&3) + 8(i>>2) + 4hh
    if (t == nt - 1 && (S & (KT - 1))) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = t * KT + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (key >= S) st[b][i] = -INFINITY;
        }
    }

    // ---- online softmax (raw-score units; c folds 1/sqrt(d) and log2 e)
    float mx = st[0][0];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[b][i]);
    mx = xhalf_max(mx);
    // Deferred rescale (cdna_hip_programming.md T13): the running reference m_run only moves when some query of this
    // wave saw its maximum grow by more than 2^RESCALE_LOG2; otherwise P stays relative to the old reference
    // (p <= 2^RESCALE_LOG2, harmless in 16-bit floating point) and the O / l rescale is skipped.  The decision is
    // wave-uniform and taken BEFORE this tile's P is exponentiated, so everything is scaled exactly once.
    const float m_new = fmaxf(m_run, mx);
    if (__any((m_new - m_run) * c > RESCALE_LOG2)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
    }
    const float mc = m_run * c;
    float psum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(st[b][i] * c - mc);
        st[b][i] = p;
        psum += p;
      }
    l_run += psum;

    // ---- P^T fragments (B operand): registers 8s..8s+7 of key block b
    typename T::vec8 pf[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = pack2<T>(st[b][8 * s + 2 * j], st[b][8 * s + 2 * j + 1]);
        pf[b][s] = __builtin_bit_cast(typename T::vec8, w);
      }

    STAMP(t2)
    // ---- O^T += V^T . P^T
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const unsigned a0 = vb + v_rd + 256 * (32 * b + 16 * s) + 16 * (((db ^ tq) << 2) | ((tg1 ^ 0) << 1));
          const unsigned a1 = vb + v_rd + 256 * (32 * b + 16 * s + 8) + 16 * (((db ^ tq) << 2) | ((tg1 ^ 1) << 1));
          const typename T::vec4 lo = T::ds_read_tr(a0);
          const typename T::vec4 hi = T::ds_read_tr(a1);
          const typename T::vec8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          ot[db] = T::mfma32(vf, pf[b][s], ot[db]);
        }
    }

    STAMP(t3)
    if (t + 1 < nt) store_tile(cur ^ 1);
    STAMP(t4)
    __syncthreads();
#ifdef MAVLM_ATTN_STAMPS
    STAMP(t5)
    acc_qk += t1 - t0; acc_sm += t2 - t1; acc_pv += t3 - t2; acc_st += t4 - t3; acc_bar += t5 - t4;
#endif
  }
#ifdef MAVLM_ATTN_STAMPS
  if (lane == 0) {
    const int wid = (blockIdx.x * 4 + wave) & 4095;
    unsigned long long* d = g_attn_stamps + 8 * wid;
    d[0] = acc_qk; d[1] = acc_sm; d[2] = acc_pv; d[3] = acc_st; d[4] = acc_bar; d[5] = nt;
    d[6] = __builtin_amdgcn_s_getreg(0x1804) | (__builtin_amdgcn_s_getreg((5 << 11) | (8 << 6) | 4) << 8);
  }
#endif

  // ---- epilogue: O[q][h*128 + 32db + 8g + 4hh + 0..3] = O^T / l
  const float l_tot = xhalf_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (q < R) {
    uint16_t* op = O + (size_t)q * ldo + h * HD + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(u32x2*)(op + 32 * db + 8 * g) = pack4<T>(ot[db][4 * g] * inv, ot[db][4 * g + 1] * inv,
                                                   ot[db][4 * g + 2] * inv, ot[db][4 * g + 3] * inv);
    if (lse2 != nullptr && hh == 0) lse2[(size_t)h * R + q] = m_run * c + log2f(l_tot);
  }
}

// ------------------------------------------------------------------------------------------------
// Column sums of the normalised probabilities.  A wave keeps 32 keys (A operand) in registers and streams
// the queries through LDS; S^T[key][query] puts the query on the lane, so the per-key sums accumulate in
// registers over the whole query loop and are reduced across lanes once at the end.
template <typename T>
__global__ __launch_bounds__(256, 2) void attn_colsum_kernel(const uint16_t* __restrict__ Q, int ldq,
                                                             const uint16_t* __restrict__ K, int ldk,
                                                             const float* __restrict__ lse2, float* __restrict__ part,
                                                             int R, int S, int H, float c) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x % H;
  const int k0 = (blockIdx.x / H) * 128 + wave * 32;
  const int r = lane & 31, hh = lane >> 5;

  typename T::vec8 kf[8];
  {
    int krow = k0 + r;
    krow = krow < S ? krow : S - 1;
    const uint16_t* kp = K + (size_t)krow * ldk + h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *(const typename T::vec8*)(kp + 16 * ks);
  }

  const int srow = tid >> 4, sch = tid & 15;
  const int st_off = 256 * srow + 16 * (sch ^ img_x(srow));
  const uint16_t* qg = Q + h * HD + sch * 8;
  u32x4 qreg[4];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int row = t * KT + srow + 16 * i;
      row = row < R ? row : R - 1;
      qreg[i] = *(const u32x4*)(qg + (size_t)row * ldq);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(smem + buf * TILE + st_off + 4096 * i) = qreg[i];
  };

  const int xr = img_x(r);
  const int q_rd = 256 * r;
  const float* lrow = lse2 + (size_t)h * R;

  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  const int nt = (R + KT - 1) / KT;
  load_tile(0);
  store_tile(0);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(kf[ks]));   // K fragments resident (see attn_fwd_kernel)
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) load_tile(t + 1);
    const char* qb_ = smem + cur * TILE;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int qi = t * KT + 32 * qb + r;
      const bool ok = qi < R;
      const float l2 = lrow[ok ? qi : R - 1];
      f32x16 st;
#pragma unroll
      for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const typename T::vec8 qf = *(const typename T::vec8*)(qb_ + q_rd + 8192 * qb + 16 * ((2 * ks + hh) ^ xr));
        st = T::mfma32(kf[ks], qf, st);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float p = __builtin_amdgcn_exp2f(st[i] * c - l2);
        acc[i] += ok ? p : 0.f;
      }
    }
    if (t + 1 < nt) store_tile(cur ^ 1);
    __syncthreads();
  }

  // reduce over the 32 query lanes of each half; key = k0 + (i&3) + 8(i>>2) + 4hh
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float v = acc[i];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    acc[i] = v;
  }
  if (r == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (key < S) part[(size_t)h * S + key] = acc[i];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void frame_scores_kernel(const float* __restrict__ part, int planes, int H, int S, int P,
                                                           void* __restrict__ out, int out_f32) {
  // one workgroup per frame; the planes of the column-sum pass are added first, in plane order (as
  // colsum_planes_reduce_kernel); fixed summation order: per thread, per wave (butterfly), then the four waves in order
  __shared__ float wsum[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const size_t plane_stride = (size_t)H * S;
  float s = 0.f;
  for (int e = tid; e < H * P; e += 256) {
    const int hh = e / P, p = e - hh * P;
    const size_t i = (size_t)hh * S + (size_t)f * P + p;
    float v = part[i];
    for (int pl = 1; pl < planes; ++pl) v += part[(size_t)pl * plane_stride + i];
    s += v;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    s = (((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) / (float)P;
    if (out_f32) ((float*)out)[f] = s;
    else ((uint16_t*)out)[f] = T::from_f32(s);
  }
}

template <typename K>
hipError_t set_lds(K kern, int bytes) {
  return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

}  // namespace

static hipError_t launch_attention2(const mavlm_attn_args& a, int dtype, hipStream_t s, float c, dim3 grid);

static bool attn_args_ok(const void* q, int ldq, const void* k, int ldk, int R, int S, int H) {
  return q && k && R > 0 && S > 0 && H > 0 && (ldq & 7) == 0 && (ldk & 7) == 0 && ldq >= H * HD && ldk >= HD;
}

int g_mavlm_attn_impl = 0;

hipError_t mavlm_launch_attention(const mavlm_attn_args& a, int dtype, hipStream_t s) {
  const int nb = a.nb > 0 ? a.nb : 1;
  if (a.H % nb != 0 || !attn_args_ok(a.Q, a.ldq, a.K, a.ldk, a.R, a.S, a.H / nb) || !a.V || !a.O || (a.ldv & 7) || (a.ldo & 3))
    return hipErrorInvalidValue;
  const float c = a.scale * 1.44269504088896340736f;
  dim3 grid(((a.R + 127) / 128) * a.H);
  if (g_mavlm_attn_impl != 2) return mavlm_launch_attention3(a, dtype, s);     // (brackets its kernels itself)
  if (nb != 1) return hipErrorInvalidValue;                                     // the register-staged kernel: single videos only
  mavlm_prof_scope prof(MAVLM_K_ATTN, 4.0 * a.R * (double)a.S * a.H * HD,
                        2.0 * HD * a.H * (2.0 * a.R + 2.0 * a.S), s);
  return launch_attention2(a, dtype, s, c, grid);
}

static hipError_t launch_attention2(const mavlm_attn_args& a, int dtype, hipStream_t s, float c, dim3 grid) {
  static mavlm_per_device_once once[2];
  if (dtype == MAVLM_F16) {
    { hipError_t e = once[1].dyn_lds((const void*)attn_fwd_kernel<F16>, ATTN_LDS); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(attn_fwd_kernel<F16>, grid, dim3(256), ATTN_LDS, s, (const uint16_t*)a.Q, a.ldq,
                       (const uint16_t*)a.K, a.ldk, (const uint16_t*)a.V, a.ldv, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.S, a.H, c);
  } else {
    { hipError_t e = once[0].dyn_lds((const void*)attn_fwd_kernel<BF16>, ATTN_LDS); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(attn_fwd_kernel<BF16>, grid, dim3(256), ATTN_LDS, s, (const uint16_t*)a.Q, a.ldq,
                       (const uint16_t*)a.K, a.ldk, (const uint16_t*)a.V, a.ldv, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.S, a.H, c);
  }
  return hipGetLastError();
}

hipError_t mavlm_launch_colsum(const mavlm_colsum_args& a, int dtype, hipStream_t s) {
  if (!attn_args_ok(a.Q, a.ldq, a.K, a.ldk, a.R, a.S, a.H) || !a.lse2 || !a.part) return hipErrorInvalidValue;
  const float c = a.scale * 1.44269504088896340736f;
  dim3 grid(((a.S + 127) / 128) * a.H);
  if (g_mavlm_attn_impl != 2) return mavlm_launch_colsum3(a, dtype, s);     // (brackets its kernels itself)
  mavlm_prof_scope prof(MAVLM_K_COLSUM, 2.0 * a.R * (double)a.S * a.H * HD, 2.0 * HD * a.H * ((double)a.R + a.S), s);
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(attn_colsum_kernel<F16>, grid, dim3(256), CS_LDS, s, (const uint16_t*)a.Q, a.ldq,
                       (const uint16_t*)a.K, a.ldk, a.lse2, a.part, a.R, a.S, a.H, c);
  else
    hipLaunchKernelGGL(attn_colsum_kernel<BF16>, grid, dim3(256), CS_LDS, s, (const uint16_t*)a.Q, a.ldq,
                       (const uint16_t*)a.K, a.ldk, a.lse2, a.part, a.R, a.S, a.H, c);
  return hipGetLastError();
}

hipError_t mavlm_launch_frame_scores(const float* part, int planes, int H, int S, int F, int P, void* out, int out_f32,
                                     int dtype, hipStream_t s) {
  if (!part || !out || F <= 0 || F * P > S || planes < 1) return hipErrorInvalidValue;
  mavlm_prof_scope prof(MAVLM_K_MISC, 0.0, 4.0 * planes * H * (double)S, s);
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(frame_scores_kernel<F16>, dim3(F), dim3(256), 0, s, part, planes, H, S, P, out, out_f32);
  else
    hipLaunchKernelGGL(frame_scores_kernel<BF16>, dim3(F), dim3(256), 0, s, part, planes, H, S, P, out, out_f32);
  return hipGetLastError();
}

#ifdef MAVLM_ATTN_STAMPS
extern "C" int mavlm_debug_read_stamps(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_attn_stamps), sizeof(unsigned long long) * n);
}
#endif
