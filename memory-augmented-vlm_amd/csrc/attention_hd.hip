// Flash attention for WIDE heads (head_dim = 448: LLaVA-OneVision-7B, hidden 3584 / 8 heads; also instantiated at
// 128 so the machinery can be cross-checked against attention3.hip).  Same math and rounding points as
// attn_fwd3_kernel; different tiling, because a 448-wide O accumulator does not fit a 32-query wave:
//
//   * 16 queries per wave on v_mfma_f32_16x16x32: O^T[d][query] = HD/16 tiles x 4 registers (112 VGPRs at 448),
//     Q fragments HD/32 x 4 (56 VGPRs), S^T of a 32-key tile = 2 tiles x 4 registers.
//   * S^T[key][query] = K.Q^T puts the query on lane&15 and 4 consecutive keys of each 16-key block on the
//     registers of lane group g = lane>>4; the row reduction is 8 local values + two lane-group exchanges
//     (v_permlane16_swap, v_permlane32_swap).
//   * The S^T accumulators become the B operand of O^T = V^T.P^T without lane movement: k-slot j of group g is
//     key 4g+j of block 0 (j < 4) or block 1 (j >= 4); the V^T fragment is read with two ds_read_b64_tr_b16 that
//     follow the same key order.
//   * QG = 2 (round 3, head_dim 448): a wave takes TWO groups of 16 queries and feeds every K / V^T fragment it reads from
//     LDS to both (2 MFMAs per fragment read).  The 16-query form is LDS-read bound - every wave reads the whole 32-key K and
//     V tiles, 16 flop per LDS byte = the MFMA peak at the full LDS rate - the 32-query form halves the LDS bytes per flop.
//     Costs registers: O^T 2 x 112 (accumulator file) + Q fragments 2 x 56 -> 4 waves of 32 queries per workgroup, one wave
//     per SIMD (512-register budget).
//   * 8 waves (128 queries) per workgroup, 32-key K/V tiles by LDS-DMA into a 2-stage ring.  A tile row is padded to
//     HDP = ceil(HD/128)*128 columns and stored as HDP/128 sub-images of [32 keys][256 B] with the swizzle of
//     attention.hip (2-way conflicts for the 16x16x32 operand reads, as the guide documents for this image).
#include "mavlm_common.h"
#include "mavlm_kernels.h"
#include <type_traits>
#include <utility>

namespace {

template <int V>
struct HIC { static constexpr int value = V; };
template <int... I, typename F>
__device__ __forceinline__ void hd_for_each(std::integer_sequence<int, I...>, F&& f) { (f(HIC<I>{}), ...); }

constexpr int KTH = 32;                                   // keys per tile
constexpr float RESCALE_H_LOG2 = 8.0f;

__device__ __forceinline__ int imgh_x(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ float xgroup_max(float v) {    // max over the 4 lanes l, l^16, l^32, l^48
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  float m = fmaxf(a, b);
  a = m; b = m;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float xgroup_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  float m = a + b;
  a = m; b = m;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return a + b;
}

template <typename T, int HD, int QG>
__global__ __launch_bounds__(512 / QG, 2 / QG) void attn_fwd_hd_kernel(const uint16_t* __restrict__ Q, int ldq,
                                                             const uint16_t* __restrict__ K, int ldk,
                                                             const uint16_t* __restrict__ V, int ldv,
                                                             uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2,
                                                             int R, int S, int H, float c, float* __restrict__ Opart,
                                                             float* __restrict__ lse_part, int tps, long long kv_bs) {
  // row batch (mavlm_attn_args::nb): blockIdx.z = video; its queries / outputs are the rows [z R, (z+1) R), its keys start
  // kv_bs elements after the previous video's, its log-sum-exp rows are [z H, (z+1) H)  (never combined with split-KV)
  {
    const size_t vb = blockIdx.z;
    Q += vb * R * ldq;
    O += vb * R * ldo;
    K += vb * kv_bs;
    V += vb * kv_bs;
    if (lse2 != nullptr) lse2 += vb * H * R;
  }
  // split-KV for small grids (as attn_fwd3_kernel): blockIdx.y owns the keys [y*tps*32, (y+1)*tps*32), writes a
  // normalised fp32 partial + its log-sum-exp; attn_combine_kernel (attention3.hip) merges them
  const int split = blockIdx.y;
  if (tps > 0) {
    const int k0 = split * tps * KTH;
    K += (size_t)k0 * ldk;
    V += (size_t)k0 * ldv;
    S = (S - k0 < tps * KTH) ? S - k0 : tps * KTH;
  }
  constexpr int NSUB = (HD + 127) / 128;                  // 128-column sub-images per tile row
  constexpr int SUB = KTH * 256;                          // 8 KiB per sub-image
  constexpr int TILE = NSUB * SUB;                        // one K or V tile
  constexpr int KS = HD / 32;                             // k-steps of S^T = K.Q^T
  constexpr int DB = HD / 16;                             // 16-row blocks of O^T
  constexpr int NCH = HD / 8;                             // valid 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [stage][K tile | V tile]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NWV = 8 / QG;                             // waves per workgroup (128 queries either way)
  const int h = blockIdx.x % H;
  const int q0 = (blockIdx.x / H) * 128 + wave * 16 * QG;  // + 16 qg
  const int qi = lane & 15, g = lane >> 4;
  const int nt = (S + KTH - 1) / KTH;

  // ---- Q fragments (B operand): lane holds Q[q0 + 16 qg + qi][h*HD + 32ks + 8g + 0..7]
  typename T::vec8 qf[QG][KS];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    int qrow = q0 + 16 * qg + qi;
    qrow = qrow < R ? qrow : R - 1;
    const uint16_t* qp = Q + (size_t)qrow * ldq + h * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qg][ks] = *(const typename T::vec8*)(qp + 32 * ks);
  }

  // ---- LDS-DMA: a tile is NSUB*8 instructions of 1 KiB (4 rows of one sub-image); wave w issues instruction ids
  // w, w+8, ...  Instruction id = sub*8 + rg writes rows 4rg..4rg+3 of sub-image `sub`.
  auto dma_tile = [&](const uint16_t* base, int ld, int t, char* dst) {
#pragma unroll
    for (int k = 0; k < NSUB * QG; ++k) {
      const int id = wave + NWV * k;                      // wave-uniform
      const int sub = id >> 3, rg = id & 7;
      const int row = 4 * rg + (lane >> 4);
      int ch = (lane & 15) ^ imgh_x(row);                 // logical chunk of the sub-image stored at physical lane&15
      int gch = sub * 16 + ch;
      gch = gch < NCH ? gch : (gch & 7);                  // pad chunks of the last sub-image: any in-bounds source
      int krow = t * KTH + row;
      krow = krow < S ? krow : S - 1;
      const uint16_t* p = base + (size_t)krow * ld + h * HD + gch * 8;
      __builtin_amdgcn_global_load_lds((const MAVLM_GLOBAL void*)p, (MAVLM_LDS void*)(dst + sub * SUB + rg * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read geometry
  // K row read (A operand 16x16x32): row = 16 kb + qi, global chunk 4 ks + g -> sub-image (4ks+g)>>4, chunk &15
  const int xq = imgh_x(qi);                              // x(row) for row = 16 kb + qi
  const int k_rd = 256 * qi;
  // V^T transposed read: lane i = 4q+p of group g reads row 4g+q (+16 for the second block), columns 4p..4p+3 of a
  // 16-column d block db: global chunk 2 db + (p>>1)
  const int tq = (lane & 15) >> 2, tp = lane & 3;
  const int vrow = 4 * g + tq;                            // + 16 for block 1
  const int vx = imgh_x(vrow);                            // same for vrow + 16
  const unsigned vbase_off = 256 * vrow + 8 * (tp & 1);

  f32x4 ot[QG][DB];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
#pragma unroll
    for (int d = 0; d < DB; ++d) ot[qg][d] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run[qg] = -1e30f;
    l_run[qg] = 0.f;
  }

  dma_tile(K, ldk, 0, smem);
  dma_tile(V, ldv, 0, smem + TILE);
#pragma unroll
  for (int qg = 0; qg < QG; ++qg)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[qg][ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    char* kb = smem + cur * 2 * TILE;
    char* vb = kb + TILE;
    if (t + 1 < nt) {                                     // next tile into the other stage (dead since the last barrier)
      dma_tile(K, ldk, t + 1, smem + (cur ^ 1) * 2 * TILE);
      dma_tile(V, ldv, t + 1, smem + (cur ^ 1) * 2 * TILE + TILE);
    }

    // ---- S^T = K.Q^T : 2 key blocks x KS k-steps; every K fragment feeds the QG query groups of the wave
    f32x4 st[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      st[qg][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      st[qg][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int gch = 4 * ks;                             // + g ; sub-image = gch>>4 is compile-time, (gch&15)+g < 16
      const int off = (gch >> 4) * SUB + k_rd + 16 * (((gch & 15) + g) ^ xq);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const typename T::vec8 kf = *(const typename T::vec8*)(kb + off + 4096 * b);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) st[qg][b] = T::mfma16(kf, qf[qg][ks], st[qg][b]);
      }
    }
    if (t == nt - 1 && (S & (KTH - 1))) {                 // ragged tail: key = 32t + 16b + 4g + r
#pragma unroll
      for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * KTH + 16 * b + 4 * g + r >= S) st[qg][b][r] = -INFINITY;
    }

    // ---- online softmax with deferred rescale (same rule as attn_fwd3_kernel; the decision is uniform per 16-query group:
    // a group's queries sit on all 64 lanes, so `__any` per group - as the 16-query wave of the QG = 1 form decides)
    typename T::vec8 pf[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      float mx = fmaxf(fmaxf(fmaxf(st[qg][0][0], st[qg][0][1]), fmaxf(st[qg][0][2], st[qg][0][3])),
                       fmaxf(fmaxf(st[qg][1][0], st[qg][1][1]), fmaxf(st[qg][1][2], st[qg][1][3])));
      mx = xgroup_max(mx);
      const float m_new = fmaxf(m_run[qg], mx);
      if (__any((m_new - m_run[qg]) * c > RESCALE_H_LOG2)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[qg] - m_new) * c);
        m_run[qg] = m_new;
        l_run[qg] *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          f32x4 v = ot[qg][d];
          if constexpr (QG > 1) asm volatile("" : "+a"(v));   // O^T lives in the accumulator file: without this pin hipcc
          v *= alpha;                                         // hoists the 224 v_accvgpr_read out of this rare branch to
          ot[qg][d] = v;                                      // the top of every tile (and spills the Q fragments for it)
        }
      }
      const float mc = m_run[qg] * c;
      float psum = 0.f;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          st[qg][b][r] = __builtin_amdgcn_exp2f(st[qg][b][r] * c - mc);
          psum += st[qg][b][r];
        }
      l_run[qg] += psum;                                  // per-lane partial; lane groups are summed at the end
      u32x4 pw;
      pw[0] = pack2<T>(st[qg][0][0], st[qg][0][1]); pw[1] = pack2<T>(st[qg][0][2], st[qg][0][3]);
      pw[2] = pack2<T>(st[qg][1][0], st[qg][1][1]); pw[3] = pack2<T>(st[qg][1][2], st[qg][1][3]);
      pf[qg] = __builtin_bit_cast(typename T::vec8, pw);
    }

    // ---- O^T += V^T.P^T : DB blocks of 16 columns, one 32-key k-step.  The transposed reads are inline asm with
    // hand-counted lgkmcnt waits (as attn_fwd3_kernel): through the builtin hipcc puts an s_waitcnt vmcnt(0) in front of
    // the first one (it cannot prove that the read does not alias the LDS-DMA in flight), which drained the next tile's
    // DMAs in the middle of this one.  Step db issues the two reads of step db + 2, then waits until only the younger
    // reads are outstanding.  A V^T fragment feeds the QG query groups.
    const unsigned vbl = (unsigned)(uintptr_t)(MAVLM_LDS const char*)vb;
    {
      u32x2 vlo[DB], vhi[DB];
      auto vrd = [&](auto ic) {
        constexpr int db = decltype(ic)::value;
        constexpr int gch = 2 * db;                           // + (tp>>1); sub-image = gch>>4 compile-time
        const unsigned a0 = vbl + vbase_off + 16 * (((gch & 15) + (tp >> 1)) ^ vx);
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "i"((gch >> 4) * SUB));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a0), "i"((gch >> 4) * SUB + 4096));   // rows + 16
        vlo[db] = lo; vhi[db] = hi;
      };
      auto vstep = [&](auto ic) {
        constexpr int db = decltype(ic)::value;
        if constexpr (db + 2 < DB) vrd(HIC<(db + 2 < DB ? db + 2 : DB - 1)>{});
        constexpr int ahead = (DB - 1 - db) < 2 ? (DB - 1 - db) : 2;
        if constexpr (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                    // keep the MFMA below the wait
        u32x4 both;
        both[0] = vlo[db][0]; both[1] = vlo[db][1]; both[2] = vhi[db][0]; both[3] = vhi[db][1];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) ot[qg][db] = T::mfma16(__builtin_bit_cast(typename T::vec8, both), pf[qg], ot[qg][db]);
        __builtin_amdgcn_sched_barrier(0);
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // nothing older than the reads below is outstanding
      vrd(HIC<0>{});
      if constexpr (DB > 1) vrd(HIC<1>{});
      __builtin_amdgcn_sched_barrier(0);
      hd_for_each(std::make_integer_sequence<int, DB>{}, vstep);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  // ---- epilogue: O[q][h*HD + 16 db + 4g + 0..3] = O^T / l
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const float l_tot = xgroup_sum(l_run[qg]);
    const float inv = 1.0f / l_tot;
    const int q = q0 + 16 * qg + qi;
    if (q < R) {
      if (tps > 0) {
        float* pp = Opart + ((size_t)split * R + q) * (H * HD) + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db)
          *(f32x4*)(pp + 16 * db) = f32x4{ot[qg][db][0] * inv, ot[qg][db][1] * inv, ot[qg][db][2] * inv, ot[qg][db][3] * inv};
        if (g == 0) lse_part[((size_t)split * H + h) * R + q] = m_run[qg] * c + log2f(l_tot);
      } else {
        uint16_t* op = O + (size_t)q * ldo + h * HD + 4 * g;
#pragma unroll
        for (int db = 0; db < DB; ++db)
          *(u32x2*)(op + 16 * db) = pack4<T>(ot[qg][db][0] * inv, ot[qg][db][1] * inv, ot[qg][db][2] * inv, ot[qg][db][3] * inv);
        if (lse2 != nullptr && g == 0) lse2[(size_t)h * R + q] = m_run[qg] * c + log2f(l_tot);
      }
    }
  }
}

// Column sums for wide heads: 32 keys per wave stationary (A operand of v_mfma_f32_32x32x16), Q streamed through LDS
// in 32-query tiles by LDS-DMA; same structure as attn_colsum_kernel with the sub-image tile layout above.
template <typename T, int HD>
__global__ __launch_bounds__(256, 1) void attn_colsum_hd_kernel(const uint16_t* __restrict__ Q, int ldq,
                                                                const uint16_t* __restrict__ K, int ldk,
                                                                const float* __restrict__ lse2, float* __restrict__ part,
                                                                int R, int S, int H, float c) {
  constexpr int NSUB = (HD + 127) / 128;
  constexpr int SUB = KTH * 256;
  constexpr int TILE = NSUB * SUB;
  constexpr int KS = HD / 16;                             // k-steps of the 32x32x16 MFMA
  constexpr int NCH = HD / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 Q slots
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = blockIdx.x % H;
  const int k0 = (blockIdx.x / H) * 128 + wave * 32;
  const int r = lane & 31, hh = lane >> 5;
  const int nt = (R + KTH - 1) / KTH;

  typename T::vec8 kf[KS];
  {
    int krow = k0 + r;
    krow = krow < S ? krow : S - 1;
    const uint16_t* kp = K + (size_t)krow * ldk + h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) kf[ks] = *(const typename T::vec8*)(kp + 16 * ks);
  }
  auto dma_q = [&](int t, char* dst) {                    // 4 waves: instruction ids wave, wave+4, ...
#pragma unroll
    for (int k = 0; k < 2 * NSUB; ++k) {
      const int id = wave + 4 * k;
      const int sub = id >> 3, rg = id & 7;
      const int row = 4 * rg + (lane >> 4);
      int gch = sub * 16 + ((lane & 15) ^ imgh_x(row));
      gch = gch < NCH ? gch : (gch & 7);
      int qrow = t * KTH + row;
      qrow = qrow < R ? qrow : R - 1;
      const uint16_t* p = Q + (size_t)qrow * ldq + h * HD + gch * 8;
      __builtin_amdgcn_global_load_lds((const MAVLM_GLOBAL void*)p, (MAVLM_LDS void*)(dst + sub * SUB + rg * 1024), 16, 0, 0);
    }
  };
  const int xr = imgh_x(r);
  const int q_rd = 256 * r;
  const float* lrow = lse2 + (size_t)h * R;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  dma_q(0, smem);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(kf[ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  for (int t = 0; t < nt; ++t) {
    const char* qb = smem + (t & 1) * TILE;
    const int qidx = t * KTH + r;
    const bool ok = qidx < R;
    const float l2 = ok ? lrow[qidx] : INFINITY;          // issued BEFORE the DMAs: waiting for it does not drain them
    if (t + 1 < nt) dma_q(t + 1, smem + ((t + 1) & 1) * TILE);
    f32x16 st;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int gch = 2 * ks;                             // + hh
      const typename T::vec8 qf = *(const typename T::vec8*)(qb + (gch >> 4) * SUB + q_rd + 16 * (((gch & 15) + hh) ^ xr));
      st = T::mfma32(kf[ks], qf, st);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += __builtin_amdgcn_exp2f(st[i] * c - l2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
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

template <typename T, int HD, int QG>
hipError_t launch_fwd_hd(const mavlm_attn_args& a, hipStream_t s) {
  constexpr int LDS = 4 * ((HD + 127) / 128) * KTH * 256;
  auto kern = attn_fwd_hd_kernel<T, HD, QG>;
  static mavlm_per_device_once once;
  {
    hipError_t e = once.dyn_lds((const void*)kern, LDS);
    if (e != hipSuccess) return e;
  }
  const float c = a.scale * 1.44269504088896340736f;
  const int nb = a.nb > 0 ? a.nb : 1;                     // row batch: a.H counts the heads of ALL videos
  if (a.H % nb != 0) return hipErrorInvalidValue;
  const int Hv = a.H / nb;
  int tps = 0;
  int ns = (a.split_ws != nullptr && nb == 1) ? mavlm_attention_hd_splits(a.R, a.S, a.H, &tps) : 1;
  if (ns <= 1) { ns = 1; tps = 0; }
  float* opart = a.split_ws;
  float* lpart = ns > 1 ? a.split_ws + (size_t)ns * a.R * a.H * HD : nullptr;
  mavlm_prof_scope prof(MAVLM_K_ATTN, 4.0 * a.R * (double)a.S * a.H * HD, 2.0 * HD * a.H * (2.0 * a.R + 2.0 * a.S), s);
  hipLaunchKernelGGL(kern, dim3(((a.R + 127) / 128) * Hv, ns, nb), dim3(512 / QG), LDS, s, (const uint16_t*)a.Q, a.ldq,
                     (const uint16_t*)a.K, a.ldk, (const uint16_t*)a.V, a.ldv, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.S, Hv, c,
                     opart, lpart, tps, (long long)a.kv_bstride);
  if (ns > 1) return mavlm_launch_attention_combine(opart, lpart, a.O, a.ldo, a.lse2, a.R, a.H, HD, ns, std::is_same<T, F16>::value ? MAVLM_F16 : MAVLM_BF16, s);
  return hipGetLastError();
}

template <typename T, int HD>
hipError_t launch_colsum_hd(const mavlm_colsum_args& a, hipStream_t s) {
  constexpr int LDS = 2 * ((HD + 127) / 128) * KTH * 256;
  auto kern = attn_colsum_hd_kernel<T, HD>;
  static mavlm_per_device_once once;
  {
    hipError_t e = once.dyn_lds((const void*)kern, LDS);
    if (e != hipSuccess) return e;
  }
  const float c = a.scale * 1.44269504088896340736f;
  hipLaunchKernelGGL(kern, dim3(((a.S + 127) / 128) * a.H), dim3(256), LDS, s, (const uint16_t*)a.Q, a.ldq, (const uint16_t*)a.K,
                     a.ldk, a.lse2, a.part, a.R, a.S, a.H, c);
  return hipGetLastError();
}

}  // namespace

// Split-KV plan of the wide-head kernel (128-query workgroups of 8 waves, at most 2 per CU; 32-key tiles)
int mavlm_attention_hd_splits(int R, int S, int H, int* tiles_per_split) {
  const int items = ((R + 127) / 128) * H;
  const int nt = (S + KTH - 1) / KTH;
  int ns = 1;
  if (items < 200 && nt >= 32) {
    ns = 400 / items;
    if (ns > 8) ns = 8;
    if (ns > nt / 16) ns = nt / 16;
    if (ns < 2) ns = 1;
  }
  int tps = (nt + ns - 1) / ns;
  ns = (nt + tps - 1) / tps;
  if (tiles_per_split) *tiles_per_split = ns > 1 ? tps : 0;
  return ns;
}

size_t mavlm_attention_hd_split_ws_floats(int R, int S, int H, int head_dim) {
  const int ns = mavlm_attention_hd_splits(R, S, H, nullptr);
  return ns > 1 ? (size_t)ns * R * H * head_dim + (size_t)ns * H * R : 0;
}

int g_mavlm_attn_hd_qg = 0;      // tuning hook: 0 = automatic (two query groups per wave at head_dim 448), 1 / 2 = forced
hipError_t mavlm_launch_attention_hd(const mavlm_attn_args& a, int head_dim, int dtype, hipStream_t s) {
  const bool f16 = dtype == MAVLM_F16;
  if (head_dim == 448) {
    // same arithmetic per query either way (a 16-query group decides its rescales alone): results are bit-identical
    if (g_mavlm_attn_hd_qg != 1) return f16 ? launch_fwd_hd<F16, 448, 2>(a, s) : launch_fwd_hd<BF16, 448, 2>(a, s);
    return f16 ? launch_fwd_hd<F16, 448, 1>(a, s) : launch_fwd_hd<BF16, 448, 1>(a, s);
  }
  if (head_dim == 128) return f16 ? launch_fwd_hd<F16, 128, 1>(a, s) : launch_fwd_hd<BF16, 128, 1>(a, s);
  // 4-head encoder of the inactive MemoryFuser variant (MemoryFuser.py:12-19): hidden 1024 / 896 over nhead = 4
  if (head_dim == 256) return f16 ? launch_fwd_hd<F16, 256, 1>(a, s) : launch_fwd_hd<BF16, 256, 1>(a, s);
  if (head_dim == 224) return f16 ? launch_fwd_hd<F16, 224, 1>(a, s) : launch_fwd_hd<BF16, 224, 1>(a, s);
  return hipErrorInvalidValue;
}

hipError_t mavlm_launch_colsum_hd(const mavlm_colsum_args& a, int head_dim, int dtype, hipStream_t s) {
  const bool f16 = dtype == MAVLM_F16;
  if (head_dim == 448) return f16 ? launch_colsum_hd<F16, 448>(a, s) : launch_colsum_hd<BF16, 448>(a, s);
  if (head_dim == 128) return f16 ? launch_colsum_hd<F16, 128>(a, s) : launch_colsum_hd<BF16, 128>(a, s);
  return hipErrorInvalidValue;
}
