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

template <typename T, int HD>
__global__ __launch_bounds__(512, 2) void attn_fwd_hd_kernel(const uint16_t* __restrict__ Q, int ldq,
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
  const int h = blockIdx.x % H;
  const int q0 = (blockIdx.x / H) * 128 + wave * 16;
  const int qi = lane & 15, g = lane >> 4;
  const int nt = (S + KTH - 1) / KTH;

  // ---- Q fragments (B operand): lane holds Q[q0+qi][h*HD + 32ks + 8g + 0..7]
  typename T::vec8 qf[KS];
  {
    int qrow = q0 + qi;
    qrow = qrow < R ? qrow : R - 1;
    const uint16_t* qp = Q + (size_t)qrow * ldq + h * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const typename T::vec8*)(qp + 32 * ks);
  }

  // ---- LDS-DMA: a tile is NSUB*8 instructions of 1 KiB (4 rows of one sub-image); wave w issues instruction ids
  // w, w+8, ...  Instruction id = sub*8 + rg writes rows 4rg..4rg+3 of sub-image `sub`.
  auto dma_tile = [&](const uint16_t* base, int ld, int t, char* dst) {
#pragma unroll
    for (int k = 0; k < NSUB; ++k) {
      const int id = wave + 8 * k;                        // wave-uniform
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

  f32x4 ot[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d) ot[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  dma_tile(K, ldk, 0, smem);
  dma_tile(V, ldv, 0, smem + TILE);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
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

    // ---- S^T = K.Q^T : 2 key blocks x KS k-steps
    f32x4 st[2];
    st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
    st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int gch = 4 * ks;                             // + g ; sub-image = gch>>4 is compile-time, (gch&15)+g < 16
      const int off = (gch >> 4) * SUB + k_rd + 16 * (((gch & 15) + g) ^ xq);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const typename T::vec8 kf = *(const typename T::vec8*)(kb + off + 4096 * b);
        st[b] = T::mfma16(kf, qf[ks], st[b]);
      }
    }
    if (t == nt - 1 && (S & (KTH - 1))) {                 // ragged tail: key = 32t + 16b + 4g + r
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t * KTH + 16 * b + 4 * g + r >= S) st[b][r] = -INFINITY;
    }

    // ---- online softmax with deferred rescale (same rule as attn_fwd3_kernel; wave-uniform decision)
    float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])),
                     fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
    mx = xgroup_max(mx);
    const float m_new = fmaxf(m_run, mx);
    if (__any((m_new - m_run) * c > RESCALE_H_LOG2)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d) ot[d] *= alpha;
    }
    const float mc = m_run * c;
    float psum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        st[b][r] = __builtin_amdgcn_exp2f(st[b][r] * c - mc);
        psum += st[b][r];
      }
    l_run += psum;                                        // per-lane partial; lane groups are summed at the end
    u32x4 pw;
    pw[0] = pack2<T>(st[0][0], st[0][1]); pw[1] = pack2<T>(st[0][2], st[0][3]);
    pw[2] = pack2<T>(st[1][0], st[1][1]); pw[3] = pack2<T>(st[1][2], st[1][3]);
    const typename T::vec8 pf = __builtin_bit_cast(typename T::vec8, pw);

    // ---- O^T += V^T.P^T : DB blocks of 16 columns, one 32-key k-step.  The transposed reads are inline asm with
    // hand-counted lgkmcnt waits (as attn_fwd3_kernel): through the builtin hipcc puts an s_waitcnt vmcnt(0) in front of
    // the first one (it cannot prove that the read does not alias the LDS-DMA in flight), which drained the next tile's
    // DMAs in the middle of this one.  Step db issues the two reads of step db + 2, then waits until only the younger
    // reads are outstanding.
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
        ot[db] = T::mfma16(__builtin_bit_cast(typename T::vec8, both), pf, ot[db]);
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
  const float l_tot = xgroup_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int q = q0 + qi;
  if (q < R) {
    if (tps > 0) {
      float* pp = Opart + ((size_t)split * R + q) * (H * HD) + h * HD + 4 * g;
#pragma unroll
      for (int db = 0; db < DB; ++db)
        *(f32x4*)(pp + 16 * db) = f32x4{ot[db][0] * inv, ot[db][1] * inv, ot[db][2] * inv, ot[db][3] * inv};
      if (g == 0) lse_part[((size_t)split * H + h) * R + q] = m_run * c + log2f(l_tot);
    } else {
      uint16_t* op = O + (size_t)q * ldo + h * HD + 4 * g;
#pragma unroll
      for (int db = 0; db < DB; ++db)
        *(u32x2*)(op + 16 * db) = pack4<T>(ot[db][0] * inv, ot[db][1] * inv, ot[db][2] * inv, ot[db][3] * inv);
      if (lse2 != nullptr && g == 0) lse2[(size_t)h * R + q] = m_run * c + log2f(l_tot);
    }
  }
}

// Levelled stream-K plan of attn_fwd_hd2_kernel - the schedule of attn_fwd3_kernel (attention3.hip: after `full` whole units
// per workgroup, the remaining units are cut into 2^k equal key ranges level by level) over 128-query units, 32-key tiles and
// G = 256 workgroups (one per CU).
struct hd_sk_plan {
  int wgs;           // G (0 = schedule not used)
  int full;          // whole units per workgroup
  int nlev;
  int k[6];          // level: units are cut into 2^k key ranges
  int base[6];       // first unit of the level
  int nun[6];        // units of the level (<= G >> k)
  int slot[6];       // first partial slot of the level (+ virtual workgroup id)
};

// ---------------------------------------------------------------------------------------------------------------------
// Round 3: the head_dim-448 forward with 32-QUERY waves on v_mfma_f32_32x32x16, software-pipelined like attn_fwd3_kernel.
//
// Why.  In attn_fwd_hd_kernel every wave reads the whole 32-key K and V tiles from LDS for 16 queries: 16 flop per LDS byte,
// the kernel sat at 0.6-0.7 PFLOP/s with the LDS pipe as the bound (and 2-way bank conflicts on the 16x16x32 K reads).  A
// 32-query wave halves the LDS bytes per flop, but its state - O^T 32 x 448 fp32 = 224 accumulator registers, the Q fragments
// 112 - leaves room for ONE wave per SIMD, and a lone wave issues one instruction per 4 clocks whatever its kind.  A first
// form of this kernel (two 16-query groups per wave on 16x16x32: 112 MFMAs + ~460 other instructions per tile) ran 0.88
// PFLOP/s at R = 12 544, S = 6 272 with the matrix pipe 47 % busy inside the loop (rocprofv3: SQ_ACTIVE_INST_ANY 49 %,
// SQ_WAIT_INST_ANY 29 %, SQ_WAIT_ANY 21 % of the wave cycles; LDS 36 % busy, half of it bank conflicts): bound by instruction
// issue, not by LDS or latency (read-ahead depth and DMA placement: no effect).  The 32x32x16 shape does the same flops in
// HALF the MFMAs (56 per tile, 32 clocks each, 8 of them holding the issue port): six free issue slots per MFMA instead of
// two, conflict-free K reads; with ~165 / ~145 other instructions in the two phases of a tile (K pieces in [A], V pieces in
// [B]; sub-image offsets as instruction immediates; accumulator read + fma + exp as one asm statement; the ragged-tail mask a
// real branch) it runs 0.94-0.96 PFLOP/s at that shape as a plain grid - 784 workgroups, 3.06 rounds of 256 CUs - the matrix
// pipe 59 % busy inside the loop (profiles/r03_wide_head_pmc.txt), the rest waits (SQ_WAIT_ANY 27 %: an exposed LDS round trip
// when phase [A] opens, the barrier per tile) and issue stalls; 1.07 PFLOP/s on the levelled stream-K plan below.
//
// Mapping = attn_fwd3_kernel's, a 128-column sub-image standing where its head stood:
//   S^T[key][query] = K.Q^T : 28 k-steps of 16 dims, one ds_read_b128 K row fragment per MFMA (lane = key row, half = lane>>5);
//                     accumulator register i of lane (query r, half hh) is key (i & 3) + 8 (i >> 2) + 4 hh of the tile
//   O^T[d][query] += V^T.P^T : 14 blocks of 32 columns x 2 k-steps of 16 keys; the S^T registers are the B operand as they are
//                     (k-slot j of half hh = key 16 bs + {0-3, 8-11}[j] + 4 hh), the V^T fragment is two ds_read_b64_tr_b16
//   [A]  S'(t+1) = K(t+1).Q^T   ||  p = exp2(s c - m c), 16-bit converts and row sums of tile t, LDS-DMA of K(t+2), V(t+1)
//   [B]  O^T += V(t)^T.P(t)^T   ||  row maxima of tile t+1
// with hand-counted lgkmcnt waits around inline-asm fragment reads, buffer-load LDS-DMA (one loop-invariant lane offset per
// operand, tile and piece in the scalar offset; rows past S read as zeros) and K / V in separate 2-slot rings (K runs one
// tile ahead of V).  Same image layout, swizzle and rounding points as attn_fwd_hd_kernel; 4 waves x 32 queries per workgroup.
//
// FT = 1: the frame-score variant (last formation layer at the OneVision-7B width, MemoryController.py:135-139; round 4 - the column-sum
// pass attn_colsum_hd_kernel recomputed Q.K^T for it: 4 % of the step).  The keys are frames of FPK consecutive patches.  For every
// (query row, 32-key tile) the kernel writes ONE 8-byte entry: the log2 masses (relative to nothing: log2 sum p + m c) of the tile's keys
// before and behind the frame boundary inside it (FPK % 4 == 0: a boundary never cuts one of the 4-key groups a lane holds; FPK > 32:
// at most one boundary per tile; no boundary: the second value is -inf).  A (row, tile) has exactly one writer under every schedule -
// plain grid, split-KV, levelled stream-K: pieces own disjoint tile ranges - so nothing has to be merged; frame_tiles_kernel, which runs
// after the row's final log-sum-exp is known, turns the entries into partial frame sums, frame_finish_kernel (attention3.hip) adds
// them in a fixed order.  O and lse2 are bit-identical to the FT = 0 kernel (the row sums keep their order).
template <typename T, int HD, int KPF, int VPF, int FT = 0>
__global__ __launch_bounds__(256, 1) void attn_fwd_hd2_kernel(const uint16_t* __restrict__ Qa, int ldq,
                                                              const uint16_t* __restrict__ Ka, int ldk,
                                                              const uint16_t* __restrict__ Va, int ldv,
                                                              uint16_t* __restrict__ Oa, int ldo, float* __restrict__ lse2a,
                                                              int R, int S_all, int H, float c, float* __restrict__ Opart,
                                                              float* __restrict__ lse_part, int tps, long long kv_bs,
                                                              hd_sk_plan plan, float* __restrict__ fent = nullptr, int FPK = 0,
                                                              int HA = 0) {
  constexpr int NSUB = (HD + 127) / 128;
  constexpr int SUB = KTH * 256;
  constexpr int TILE = NSUB * SUB;
  constexpr int KS = HD / 16;                               // k-steps of S^T = K.Q^T (32x32x16)
  constexpr int DB = HD / 32;                               // 32-column blocks of O^T
  constexpr int NB = 2 * DB;                                // MFMA steps of phase [B]: (16-key step bs, block)
  constexpr int NPW = NSUB * 2;                             // 1 KiB DMA pieces per wave and tile (NSUB * 8 over 4 waves)
  static_assert(HD % 32 == 0 && 2 * TILE <= 65536 && 2 * NPW <= KS && KS >= 22 && NB >= 16, "phase schedules assume a wide head");
  static_assert(KPF >= 1 && KPF <= 6 && VPF >= 1 && VPF <= 4, "extend the wait tables");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K slots at 0, TILE ; V slots at 2 TILE, 3 TILE
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int nt_all = (S_all + KTH - 1) / KTH;
  const int nqb = (R + 127) / 128;

  // ---- schedule (H = heads per video; "global head" hg = video * H + head).  A workgroup runs one or more SEGMENTS: a unit
  // (global head, 128-query block) x a contiguous range of its key tiles, with a fresh online-softmax state.
  //   * plain (plan.wgs == 0, tps == 0): blockIdx = (unit of the video, -, video), one whole-unit segment.
  //   * split-KV (tps > 0; small single-video grids): blockIdx.y owns the key tiles [y tps, (y+1) tps); normalised fp32 partial
  //     + log-sum-exp per split, merged by attn_combine_kernel (attention3.hip).
  //   * levelled stream-K (plan.wgs = G > 0; more units than CUs): G persistent workgroups take plan.full whole units each, then
  //     the levels - unit ul of level (k, base) is cut into 2^k key ranges, piece p going to virtual workgroup (ul << k) + p; XCD
  //     x owns the virtual ids [x G/8, (x+1) G/8): contiguous units of the head-major order, and the 2^k workgroups that share
  //     a unit sit on one XCD.  Pieces write partials into slot plan.slot[level] + virtual id; attn_combine_hd_sk_kernel merges.
  const int sk_v = plan.wgs > 0 ? xcd_remap((int)blockIdx.x, plan.wgs) : 0;
  const int nseg = plan.wgs > 0 ? plan.full + plan.nlev : 1;
  for (int si = 0; si < nseg; ++si) {
  int hg, qblk, t_lo, nt, out_kind, split = 0, sk_slot = 0;   // out_kind 0: O / lse2, 1: split-KV partial, 2: stream-K partial
  if (plan.wgs > 0) {
    int u;
    if (si < plan.full) {
      u = si * plan.wgs + sk_v;
      t_lo = 0;
      nt = nt_all;
      out_kind = 0;
    } else {
      const int lv = si - plan.full;
      int lk = plan.k[0], lbase = plan.base[0], lnun = plan.nun[0], lslot = plan.slot[0];
#pragma unroll
      for (int j = 1; j < 6; ++j)                              // (static indices: the plan lives in scalar registers)
        if (lv == j) { lk = plan.k[j]; lbase = plan.base[j]; lnun = plan.nun[j]; lslot = plan.slot[j]; }
      const int ul = sk_v >> lk;
      if (ul >= lnun) continue;                                // this workgroup has no unit on this (partial) level
      const int piece = sk_v & ((1 << lk) - 1);
      u = lbase + ul;
      t_lo = (int)(((long long)piece * nt_all) >> lk);
      nt = (int)(((long long)(piece + 1) * nt_all) >> lk) - t_lo;
      out_kind = 2;
      sk_slot = lslot + sk_v;
      if (nt == 0) {                                           // fewer key tiles than pieces: a neutral partial (weight 0)
        float* pp = Opart + ((size_t)sk_slot * 128 + wave * 32 + r) * HD + (HD / 2) * hh;
#pragma unroll
        for (int g = 0; g < HD / 8; ++g) *(f32x4*)(pp + 4 * g) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hh == 0) lse_part[(size_t)sk_slot * 128 + wave * 32 + r] = -INFINITY;
        continue;
      }
    }
    hg = u / nqb;
    qblk = u - hg * nqb;
  } else {
    hg = (int)blockIdx.z * H + (int)(blockIdx.x % H);
    qblk = blockIdx.x / H;
    split = blockIdx.y;
    t_lo = tps > 0 ? split * tps : 0;
    nt = tps > 0 ? ((nt_all - t_lo < tps) ? nt_all - t_lo : tps) : nt_all;
    out_kind = tps > 0 ? 1 : 0;
  }
  const int vb = hg / H, h = hg - vb * H;                      // video of the row batch, head inside it
  const uint16_t* const Q = Qa + (size_t)vb * R * ldq;
  uint16_t* const O = Oa + (size_t)vb * R * ldo;
  const uint16_t* const K = Ka + (size_t)vb * kv_bs + (size_t)t_lo * KTH * ldk;
  const uint16_t* const V = Va + (size_t)vb * kv_bs + (size_t)t_lo * KTH * ldv;
  float* const lse2 = lse2a != nullptr ? lse2a + (size_t)vb * H * R : nullptr;
  const int S = (S_all - t_lo * KTH < nt * KTH) ? S_all - t_lo * KTH : nt * KTH;      // keys of this segment
  const int q0 = qblk * 128 + wave * 32;
  // frame-tile entries (FT): [global head][tile of the unit][row] x 8 bytes (a wave's 32 rows of a tile are 256 contiguous bytes: one
  // coalesced store); rows past R (clamped duplicates of row R-1) store behind the descriptor's end, which the hardware drops
  __amdgpu_buffer_rsrc_t ers;
  int e_voff = 0, f_end = 0;
  if constexpr (FT != 0) {
    const uintptr_t ea = (uintptr_t)fent;
    const uint32_t elo = __builtin_amdgcn_readfirstlane((uint32_t)ea);
    const uint32_t ehi = __builtin_amdgcn_readfirstlane((uint32_t)(ea >> 32));
    const uint32_t ebytes = __builtin_amdgcn_readfirstlane((uint32_t)HA * (uint32_t)R * (uint32_t)nt_all * 8u);   // HA: heads of all videos
    ers = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)ehi << 32) | elo), 0, ebytes, 0x00020000);
    e_voff = q0 + r < R ? ((hg * nt_all + t_lo) * R + q0 + r) * 8 : 0x7ffffff0;
    f_end = ((t_lo * KTH) / FPK + 1) * FPK - t_lo * KTH;      // end of the frame that holds the segment's first key, piece-local
  }

  // ---- Q fragments (B operand): lane holds Q[q0 + r][h*HD + 16 ks + 8 hh + 0..7]
  typename T::vec8 qf[KS];
  {
    int qrow = q0 + r;
    qrow = qrow < R ? qrow : R - 1;
    const uint16_t* qp = Q + (size_t)qrow * ldq + h * HD + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const typename T::vec8*)(qp + 16 * ks);
  }

  // ---- LDS-DMA: piece k (0..NPW-1) of wave w is instruction id = w + 4k of the tile: 1 KiB = rows 4 rg .. 4 rg + 3 of
  // sub-image sub (id = 8 sub + rg), at LDS offset id * 1024 of the slot.
  auto head_rsrc = [&](const uint16_t* base, int ld) {
    const uintptr_t a = (uintptr_t)(base + h * HD);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(S - 1) * (uint32_t)ld * 2u + (uint32_t)HD * 2u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t krs = head_rsrc(K, ldk), vrs = head_rsrc(V, ldv);
  // The swizzle term of a row, x(row) = ((row & 3) << 2) | ((row >> 2) & 3) with row = 4 rg + (lane >> 4), rg = w + 4 (k & 1),
  // is ((lane >> 4) << 2) | w for every piece: the lane's source chunk inside a sub-image does not depend on k.  ONE lane
  // offset per operand; the piece adds the uniform (k & 1) 16 rows + (k >> 1) 256 bytes to the scalar offset.  The pad chunks
  // of the last sub-image (columns HD .. 128 NSUB) then read whatever follows the head's columns - inside the descriptor it is
  // neighbouring data, behind its end zeros - and nothing ever reads them from LDS.
  const int drow = 4 * wave + (lane >> 4);
  const int dch = (lane & 15) ^ (((lane >> 4) << 2) | wave);
  const int koff = (drow * ldk + dch * 8) * 2, voff = (drow * ldv + dch * 8) * 2;
  const unsigned lds_wave = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem + wave * 1024;
  auto dma_piece = [&](__amdgpu_buffer_rsrc_t rs, int off, int ld, int toff, int slot_off, auto kc) {
    constexpr int k = decltype(kc)::value;
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));            // M0 = base + constant stays a one-instruction recompute (no hoisted SGPRs)
    // the sub-image's column offset rides in the instruction's immediate offset (no scalar add per piece); the hardware adds
    // that immediate to the LDS address as well (LDS address = M0 + immediate + 16 lane), so it is taken out of M0 again
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (MAVLM_LDS void*)(uintptr_t)(base + slot_off + k * 4096 - (k >> 1) * 256), 16, off,
                                             toff + (k & 1) * 32 * ld, (k >> 1) * 256, 0);
  };
  auto dma_tile = [&](__amdgpu_buffer_rsrc_t rs, int off, int ld, int t, int slot_off) {
    hd_for_each(std::make_integer_sequence<int, NPW>{}, [&](auto kc) { dma_piece(rs, off, ld, t * KTH * ld * 2, slot_off, kc); });
  };

  // ---- fragment read geometry (attn_fwd3_kernel's, per 128-column sub-image): loop-invariant 32-bit LDS addresses; slot,
  // sub-image and 16-key step are immediates
  const int xr = imgh_x(r);
  const unsigned sbase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)smem;
  unsigned kad[8];                                            // k-step ks reads kad[ks & 7] + (ks >> 3) SUB + slot TILE
#pragma unroll
  for (int j = 0; j < 8; ++j) kad[j] = sbase + 256 * r + 16 * ((2 * j + hh) ^ xr);
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  const int v_rd = 256 * (4 * hh + tq) + 8 * (tp & 1) + 16 * ((tp >> 1) ^ hh);
  unsigned vad[4][2];                                         // block dbi reads vad[dbi & 3][jj] + (dbi >> 2) SUB + 4096 bs + slot TILE
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      vad[db][jj] = sbase + 2 * TILE + v_rd + 256 * 8 * jj + 16 * (((db ^ tq) << 2) | ((tg1 ^ jj) << 1));

  f32x16 ot[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
  f32x16 st[2];                                               // [parity]: S^T of one 32-key tile
  float m_run = -1e30f, l_run = 0.f;

  auto mask_tail = [&](auto par, int t) {                     // key = 32 t + (i & 3) + 8 (i >> 2) + 4 hh
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (t * KTH + (i & 3) + 8 * (i >> 2) + 4 * hh >= S) st[P][i] = -INFINITY;
  };
  // deferred rescale against the row maxima `mx` (this lane's half of the keys) of the next tile.  The decision needs no
  // exchange between the two lane halves of a query row: the row maximum exceeds the threshold iff the maximum of one half
  // does, and `__any` looks at all 64 lanes; the exchange sits in the (rare) rescale branch.
  auto rescale = [&](float mx) {
    if (__any((fmaxf(m_run, mx) - m_run) * c > RESCALE_H_LOG2)) {
      mx = xhalf_max(mx);
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        // O^T lives in the accumulator file.  The pins keep each read - multiply - write of 16 registers together and inside
        // this rare branch: left alone hipcc hoists all 224 v_accvgpr_read to the top of every tile, or runs every read
        // before the first write - either way the copies push Q fragments out to scratch.
        f32x16 v = ot[d];
        asm volatile("" : "+a"(v));
        v *= alpha;
        asm volatile("" : "+a"(v));
        ot[d] = v;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- prologue: K(0), V(0), K(1) ; S(0) ; reference maxima of tile 0
  dma_tile(krs, koff, ldk, 0, 0);
  dma_tile(vrs, voff, ldv, 0, 2 * TILE);
  if (nt > 1) dma_tile(krs, koff, ldk, 1, TILE);
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) st[0][i] = st[1][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const typename T::vec8 kf = *(const typename T::vec8*)(smem + (kad[ks & 7] - sbase) + (ks >> 3) * SUB);
      st[0] = T::mfma32(kf, qf[ks], st[0]);
    }
    if (nt == 1 && (S & (KTH - 1))) mask_tail(HIC<0>{}, 0);
    float mx = st[0][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, st[0][i]);
    rescale(mx);
  }

  // ---- one pipelined iteration: S(t) in st[P] (reference maximum decided), S'(t+1) into st[P^1]
  auto iteration = [&](auto par, int t) {
    constexpr int P = decltype(par)::value;
    constexpr int N = P ^ 1;
    const bool has_next = t + 1 < nt;
    const int ktile = (t + 2) * KTH * ldk * 2, vtile = (t + 1) * KTH * ldv * 2;   // scalar byte offsets of K(t+2), V(t+1)
    const float mc = m_run * c;
    typename T::vec8 pf[2];
    float psum = 0.f;
    // (FT) keys [0, ft_kb) of this tile belong to the current frame; ft_cut: the frame ends inside the tile (or at its end)
    const int ft_kb = f_end - t * KTH;
    const bool ft_cut = FT != 0 && ft_kb <= KTH;
    float ft_lo = 0.f;
    // element e of tile t.  The score is pinned to ITS step, and v_exp_f32 is inline asm: a volatile statement keeps its place
    // between the MFMAs without the register copy the "+v" pin around the builtin cost (a v_mov per probability).  Its readers -
    // converts and row sums in cvt() - come >= 2 steps after the last exp (the transcendental-result hazard needs one
    // independent instruction).
    auto expo = [&](int e) {
      // S^T sits in the accumulator file: fetched in ITS step (left to hipcc all 16 reads open the phase), scaled and
      // exponentiated in ONE asm statement (hipcc puts an s_nop behind every inline-asm register read it cannot see into)
      float y;
      asm volatile("v_accvgpr_read_b32 %0, %1\n\tv_fma_f32 %0, %2, %0, -%3\n\tv_exp_f32 %0, %0"
                   : "=&v"(y) : "a"(st[P][e]), "s"(c), "v"(mc));
      st[P][e] = y;
    };
    auto cvt = [&](int bs) {                                  // P^T fragment of keys 16 bs .. 16 bs + 15 (this lane's 8), row sum
      float ps = st[P][8 * bs];
#pragma unroll
      for (int j = 1; j < 8; ++j) ps += st[P][8 * bs + j];    // fp32, element order
      asm volatile("" : "+v"(ps));                            // (summed here, not at the top of phase [B])
      psum += ps;
      if constexpr (FT != 0) {
        // a frame ends inside this tile: the mass of this lane's 4-key groups (keys 8 g + 4 hh + 0..3, g = 2 bs, 2 bs + 1) below
        // the boundary.  (A scalar branch: five tiles of six have no boundary.)
        if (ft_cut) {
          const float g0 = (st[P][8 * bs] + st[P][8 * bs + 1]) + (st[P][8 * bs + 2] + st[P][8 * bs + 3]);
          const float g1 = (st[P][8 * bs + 4] + st[P][8 * bs + 5]) + (st[P][8 * bs + 6] + st[P][8 * bs + 7]);
          ft_lo += (16 * bs + 4 * hh < ft_kb ? g0 : 0.f) + (16 * bs + 8 + 4 * hh < ft_kb ? g1 : 0.f);
        }
      }
      u32x4 pw;
#pragma unroll
      for (int j = 0; j < 4; ++j) pw[j] = pack2<T>(st[P][8 * bs + 2 * j], st[P][8 * bs + 2 * j + 1]);
      asm volatile("" : "+v"(pw));                            // converted HERE (phase [A]); the fp32 values die with it
      pf[bs] = __builtin_bit_cast(typename T::vec8, pw);
    };
    // transposed V(t) fragments (V slot P; landed before the previous barrier).  Step i of phase [B] = (bs = i / DB, block
    // dbi = i % DB).  The first VPF steps' reads are issued at the END of phase [A], behind its last K read, so that phase [B]
    // does not open with an exposed LDS round trip.
    u32x2 vlo[NB], vhi[NB];
    auto vrd = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int bs = i / DB, dbi = i % DB;
      constexpr int off = P * TILE + (dbi >> 2) * SUB + 4096 * bs;
      const unsigned a0 = vad[dbi & 3][0], a1 = vad[dbi & 3][1];
      u32x2 lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "i"(off));
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "i"(off));
      vlo[i] = lo; vhi[i] = hi;
    };
    auto vopen = [&]() {
      vrd(HIC<0>{});
      if constexpr (VPF > 1) vrd(HIC<1>{});
      if constexpr (VPF > 2) vrd(HIC<2>{});
      if constexpr (VPF > 3) vrd(HIC<3>{});
    };
    if (has_next) {
      // [A] K(t+1) sits in K slot N.  Step ks: one K row fragment, one MFMA.  The K reads are the only LGKM operations in
      // flight here: step i issues the read of step i + KPF, then waits until only the younger ones are outstanding.
#pragma unroll
      for (int i = 0; i < 16; ++i) st[N][i] = 0.f;
      u32x4 kfr[KS];
      auto kread = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int off = N * TILE + (i >> 3) * SUB;
        const unsigned a = kad[i & 7];
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(off));
        kfr[i] = v;
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      kread(HIC<0>{});
      if constexpr (KPF > 1) kread(HIC<1>{});
      if constexpr (KPF > 2) kread(HIC<2>{});
      if constexpr (KPF > 3) kread(HIC<3>{});
      if constexpr (KPF > 4) kread(HIC<4>{});
      if constexpr (KPF > 5) kread(HIC<5>{});
      __builtin_amdgcn_sched_barrier(0);
      auto astep = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ahead = (KS - 1 - i) < KPF ? (KS - 1 - i) : KPF;
        if constexpr (i + KPF < KS) kread(HIC<(i + KPF < KS ? i + KPF : KS - 1)>{});
        if constexpr (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if constexpr (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if constexpr (ahead == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if constexpr (ahead == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                    // keep the MFMA below the wait
        st[N] = T::mfma32(__builtin_bit_cast(typename T::vec8, kfr[i]), qf[i], st[N]);
        // K(t+2) -> K slot P (last read before the previous barrier).  A lone wave issues one instruction per 4 clocks: this
        // phase carries ~220 instructions beside its 28 MFMAs of 32 clocks, phase [B] ~120 - the V pieces go there.
        if constexpr (i < 2 * NPW && (i & 1) == 0) dma_piece(krs, koff, ldk, ktile, P * TILE, HIC<((i >> 1) < NPW ? (i >> 1) : 0)>{});
        if constexpr (i < 16) expo(i);                        // the 16 probabilities of tile t, one per step
        if constexpr (i == 18) cvt(0);
        if constexpr (i == 20) cvt(1);
        if constexpr (i == KS - 1) vopen();                   // (every K read has returned: the wait of this step was lgkmcnt(0))
        __builtin_amdgcn_sched_barrier(0);
      };
      hd_for_each(std::make_integer_sequence<int, KS>{}, astep);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // nothing older than the reads below is outstanding
      vopen();
#pragma unroll
      for (int e = 0; e < 16; ++e) expo(e);
      __builtin_amdgcn_sched_barrier(0);
      cvt(0);
      cvt(1);
    }

    // [B] O^T += V(t)^T.P(t)^T : V(t) sits in V slot P.  Step i: two transposed reads, one MFMA; the row maxima of tile t+1
    // ride along (one score per step).
    if (has_next && t + 1 == nt - 1 && (S & (KTH - 1))) {
      // (the pins are side effects the selects depend on: they keep this a scalar branch - if-converted, its 45 compare /
      //  select instructions ran in every tile, a tenth of the issue slots of phase [B])
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = st[N][i];
        asm volatile("" : "+v"(v));
        st[N][i] = ((t + 1) * KTH + (i & 3) + 8 * (i >> 2) + 4 * hh >= S) ? -INFINITY : v;
      }
    }
    float mx = -INFINITY;
    {
      auto vstep = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int bs = i / DB, dbi = i % DB;
        if constexpr (i + VPF < NB) vrd(HIC<(i + VPF < NB ? i + VPF : NB - 1)>{});
        constexpr int ahead = (NB - 1 - i) < VPF ? (NB - 1 - i) : VPF;
        if constexpr (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if constexpr (ahead == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        u32x4 both;
        both[0] = vlo[i][0]; both[1] = vlo[i][1]; both[2] = vhi[i][0]; both[3] = vhi[i][1];
        ot[dbi] = T::mfma32(__builtin_bit_cast(typename T::vec8, both), pf[bs], ot[dbi]);
        // V(t+1) -> V slot N (last read before the previous barrier): in the FIRST steps, so that the vmcnt(0) that closes the
        // tile finds them landed (one piece per 3 steps, the last of them 7 steps before the wait: -3.5 %)
        if constexpr (i < NPW) dma_piece(vrs, voff, ldv, vtile, (2 + N) * TILE, HIC<(i < NPW ? i : 0)>{});
        if constexpr (i >= 2 && i < 18) {                     // score i - 2 of tile t+1 (stale values when there is none: mx unused)
          // fetched from the accumulator file here, one per step.  (Inline asm hides the MFMA -> v_accvgpr_read hazard from
          // hipcc: two MFMAs of this phase have issued since the last one that wrote S^T - the matrix pipe is in order.)
          float sv;
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(sv) : "a"(st[N][i - 2]));
          mx = fmaxf(mx, sv);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      __builtin_amdgcn_sched_barrier(0);
      hd_for_each(std::make_integer_sequence<int, NB>{}, vstep);
    }
    if constexpr (FT != 0) {
      // (psum and ft_lo are summed in different orders: where nothing lies behind the boundary their difference is rounding noise of
      //  either sign - clamp it, log2 of a negative value is NaN)
      float alo = ft_cut ? ft_lo : psum, ahi = fmaxf(psum - ft_lo, 0.f);
      alo = xhalf_sum(alo);                                   // both key halves of the row
      float vhi = -INFINITY;
      if (ft_cut) {
        ahi = ft_kb < KTH ? xhalf_sum(ahi) : 0.f;             // (a boundary at the tile's end: nothing behind it)
        vhi = __builtin_amdgcn_logf(ahi) + mc;                // (v_log_f32 = log2; log2(0) = -inf: weight 0)
        f_end += FPK;
      }
      const float vlo = __builtin_amdgcn_logf(alo) + mc;
      if (hh == 0) {
        u32x2 e;
        e[0] = __builtin_bit_cast(unsigned, vlo);
        e[1] = __builtin_bit_cast(unsigned, vhi);
        __builtin_amdgcn_raw_buffer_store_b64(e, ers, e_voff, t * R * 8, 0);
      }
    }
    l_run += psum;                                            // per-lane partial (this half's keys); halves are summed at the end
    if (has_next) rescale(mx);                                // reference maximum for tile t+1 (after P.V(t): it touches O)
    // (tried: the barrier VPF steps before the end of phase [B] - legal once every V(t) fragment is in registers - with the first
    //  K fragments of the next tile read behind it, under the last MFMAs and the rescale check: 940 -> 906 TFLOP/s, reverted)
    // this wave's pieces of K(t+2), V(t+1) have landed.  (FT: the entry store of this tile is the YOUNGEST vector-memory operation
    // and stays in flight - waiting for its round trip cost attn_fwd3_kernel ~20 us per launch.)
    if constexpr (FT != 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  int t = 0;
  for (; t + 1 < nt; t += 2) {
    iteration(HIC<0>{}, t);
    iteration(HIC<1>{}, t + 1);
  }
  if (t < nt) iteration(HIC<0>{}, t);

  // ---- epilogue: O[q][h*HD + 32 dbi + 8 g + 4 hh + 0..3] = O^T / l
  const float l_tot = xhalf_sum(l_run);
  const float inv = 1.0f / l_tot;
  const float lse = __builtin_fmaf(m_run, c, log2f(l_tot));
  const int q = q0 + r;
  if (out_kind == 2) {                                        // stream-K partial: [slot][128 rows][HD] + [slot][128]
    float* pp = Opart + ((size_t)sk_slot * 128 + wave * 32 + r) * HD + 4 * hh;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(pp + 32 * d + 8 * g) = f32x4{ot[d][4 * g] * inv, ot[d][4 * g + 1] * inv, ot[d][4 * g + 2] * inv,
                                               ot[d][4 * g + 3] * inv};
    if (hh == 0) lse_part[(size_t)sk_slot * 128 + wave * 32 + r] = lse;
  } else if (q < R) {
    if (out_kind == 1) {
      float* pp = Opart + ((size_t)split * R + q) * (H * HD) + h * HD + 4 * hh;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(pp + 32 * d + 8 * g) = f32x4{ot[d][4 * g] * inv, ot[d][4 * g + 1] * inv, ot[d][4 * g + 2] * inv,
                                                 ot[d][4 * g + 3] * inv};
      if (hh == 0) lse_part[((size_t)split * H + h) * R + q] = lse;
    } else {
      uint16_t* op = O + (size_t)q * ldo + h * HD + 4 * hh;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(u32x2*)(op + 32 * d + 8 * g) = pack4<T>(ot[d][4 * g] * inv, ot[d][4 * g + 1] * inv, ot[d][4 * g + 2] * inv,
                                                    ot[d][4 * g + 3] * inv);
      if (lse2 != nullptr && hh == 0) lse2[(size_t)h * R + q] = lse;
    }
  }
  }  // segments
}

// Frame-tile entries [head][tile][row] -> partial frame sums (attn_fwd_hd2_kernel<.., FT = 1>).  One workgroup per (64-row block,
// global head), lane = row; wave w takes the frames w, w + 4, ...: a frame's mass in a row is the sum over the tiles that hold keys of
// it - the SECOND value of the tile the frame starts in the middle of, the FIRST value of every tile whose first key is in the frame -
// of 2^(v - lse2[row]), added in tile order; the (at most 8) entries of a frame are independent coalesced 512-byte loads.  The 64 row
// sums are added across the wave in a fixed tree order: fout[video][(head, block)][frame], which frame_finish_kernel sums in a fixed
// order.  (A first form walked all tiles of a row in one wave, carrying the running frame: 196 dependent steps, 48 us for 79 MB.)
__global__ __launch_bounds__(256) void frame_tiles_kernel(const float* __restrict__ fent, const float* __restrict__ lse2,
                                                          float* __restrict__ fout, int R, int Hv, int nt_all, int FPK, int FN,
                                                          int KT) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x, hg = blockIdx.y, nblk = gridDim.x;
  const int row = blk * 64 + lane;
  const bool ok = row < R;
  const float lse = ok ? lse2[(size_t)hg * R + row] : 0.f;
  const float2* e = (const float2*)fent + (size_t)hg * nt_all * R + (ok ? row : 0);
  const int vb = hg / Hv, h = hg - vb * Hv;
  float* const fo = fout + ((size_t)vb * Hv * nblk + (size_t)h * nblk + blk) * FN;
  for (int f = wave; f < FN; f += 4) {
    const int t0 = (f * FPK) / KT, t1 = (f * FPK + FPK - 1) / KT;        // tiles (of KT keys) that hold keys of frame f
    float rs = 0.f;
    for (int tb = t0; tb <= t1; tb += 8) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (tb + j <= t1 && tb + j < nt_all) ? e[(size_t)(tb + j) * R] : float2{-INFINITY, -INFINITY};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = tb + j;
        if (t <= t1 && t < nt_all) rs += __builtin_amdgcn_exp2f((((t * KT) / FPK == f) ? v[j].x : v[j].y) - lse);
      }
    }
    const float tot = wave_sum(ok ? rs : 0.f);
    if (lane == 0) fo[f] = tot;
  }
}

// Stream-K merge for attn_fwd_hd2_kernel: unit `ul` of level j (cut into NP = 2^k key ranges) has its partials in the slots
// plan.slot[j] + (ul << k) + p, p = 0 .. NP - 1 in key order: O = sum_p w_p O_p, w_p = 2^(lse_p - lse), lse = log2 sum_p 2^lse_p
// (an empty range carries lse = -inf: weight 0).  One workgroup per 16 rows of a cut unit: a partial row is HD / 4 lanes x
// 16 bytes, two rows per pass of 256 threads.  NP is a compile-time constant per level so that the loads of a row are issued back
// to back (attention3.hip: as run-time loops the merge was latency-bound).
template <typename T, int HD, int NP>
__device__ __forceinline__ void combine_hd_sk_rows(const float* __restrict__ Opart, const float* __restrict__ lse_part, size_t s0,
                                                   uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2, int R, int H,
                                                   int hg, int qblk, int part) {
  const int c4 = threadIdx.x & 127, rsub = threadIdx.x >> 7;
  const int vb = hg / H, h = hg - vb * H;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = part * 16 + it * 2 + rsub;
    const int q = qblk * 128 + row;
    if (q >= R || c4 >= HD / 4) continue;
    float l[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) l[p] = lse_part[(s0 + p) * 128 + row];
    float mx = l[0];
#pragma unroll
    for (int p = 1; p < NP; ++p) mx = fmaxf(mx, l[p]);
    float den = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) den += __builtin_amdgcn_exp2f(l[p] - mx);
    const float lse = mx + log2f(den);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int CH = NP < 8 ? NP : 8;                        // partial rows in flight per lane
#pragma unroll
    for (int p0 = 0; p0 < NP; p0 += CH) {
      f32x4 a[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) a[j] = *(const f32x4*)(Opart + ((s0 + p0 + j) * 128 + row) * HD + 4 * c4);
#pragma unroll
      for (int j = 0; j < CH; ++j) {                           // key order
        const float w = __builtin_amdgcn_exp2f(l[p0 + j] - lse);
        acc[0] += w * a[j][0]; acc[1] += w * a[j][1]; acc[2] += w * a[j][2]; acc[3] += w * a[j][3];
      }
    }
    *(u32x2*)(O + ((size_t)vb * R + q) * ldo + h * HD + 4 * c4) = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
    if (lse2 != nullptr && c4 == 0) lse2[(size_t)hg * R + q] = lse;
  }
}

template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_combine_hd_sk_kernel(const float* __restrict__ Opart, const float* __restrict__ lse_part,
                                                                 uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2,
                                                                 int R, int H, hd_sk_plan plan) {
  int b = blockIdx.x >> 3, lk = 0, lbase = 0, lslot = 0;      // 8 workgroups per cut unit (16 query rows each)
  const int part = blockIdx.x & 7;
  bool found = false;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (!found && j < plan.nlev) {
      if (b < plan.nun[j]) { lk = plan.k[j]; lbase = plan.base[j]; lslot = plan.slot[j]; found = true; }
      else b -= plan.nun[j];
    }
  }
  if (!found) return;
  const int nqb = (R + 127) / 128;
  const int u = lbase + b;
  const int hg = u / nqb, qblk = u - hg * nqb;
  const size_t s0 = (size_t)lslot + ((size_t)b << lk);
  switch (lk) {
    case 1: combine_hd_sk_rows<T, HD, 2>(Opart, lse_part, s0, O, ldo, lse2, R, H, hg, qblk, part); break;
    case 2: combine_hd_sk_rows<T, HD, 4>(Opart, lse_part, s0, O, ldo, lse2, R, H, hg, qblk, part); break;
    case 3: combine_hd_sk_rows<T, HD, 8>(Opart, lse_part, s0, O, ldo, lse2, R, H, hg, qblk, part); break;
    case 4: combine_hd_sk_rows<T, HD, 16>(Opart, lse_part, s0, O, ldo, lse2, R, H, hg, qblk, part); break;
    default: combine_hd_sk_rows<T, HD, 32>(Opart, lse_part, s0, O, ldo, lse2, R, H, hg, qblk, part); break;
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

// Levelled stream-K plan of the 32-query-wave kernel (see hd_sk_plan): used when the units (128-query blocks x heads of all
// videos) exceed the 256 workgroups of one-per-CU, a plain grid would leave more than 5 % of its last round empty and a unit has
// at least 2 g_mavlm_attn_sk_min_tiles key tiles of 32 (the same key count as attention3.hip's rule).  Pure function of the shape
// (mirrored by oracle/memory_path.py streamk_plan_wide).
constexpr int HD_SK_WGS = 256;
static hd_sk_plan hd2_plan(int R, int S, int H) {
  hd_sk_plan p = {};
  const int G = HD_SK_WGS;
  const long units = (long)((R + 127) / 128) * H;
  if (units <= G || (S + KTH - 1) / KTH < 2 * g_mavlm_attn_sk_min_tiles) return p;
  const long rounds = (units + G - 1) / G;
  if ((double)units / (double)(rounds * G) >= 0.95) return p;
  p.wgs = G;
  p.full = (int)(units / G);
  int rem = (int)(units % G), base = p.full * G, slot = 0;
  for (int k = 1; k <= 4; ++k)
    if (rem >= (G >> k)) {
      p.k[p.nlev] = k; p.base[p.nlev] = base; p.nun[p.nlev] = G >> k; p.slot[p.nlev] = slot;
      ++p.nlev; base += G >> k; rem -= G >> k; slot += G;
    }
  while (rem > 0) {                                           // < G/16 units left: 32-way levels of up to G/32 units
    const int n = rem < (G >> 5) ? rem : (G >> 5);
    p.k[p.nlev] = 5; p.base[p.nlev] = base; p.nun[p.nlev] = n; p.slot[p.nlev] = slot;
    ++p.nlev; base += n; rem -= n; slot += G;
  }
  return p;
}
static size_t hd2_plan_floats(const hd_sk_plan& p, int head_dim) {
  return p.wgs > 0 ? (size_t)p.wgs * p.nlev * ((size_t)128 * head_dim + 128) : 0;
}

template <typename T, int HD, int QG, int FT = 0>
hipError_t launch_fwd_hd(const mavlm_attn_args& a, hipStream_t s) {
  constexpr int LDS = 4 * ((HD + 127) / 128) * KTH * 256;
  const float c = a.scale * 1.44269504088896340736f;
  const int nb = a.nb > 0 ? a.nb : 1;                     // row batch: a.H counts the heads of ALL videos
  if (a.H % nb != 0) return hipErrorInvalidValue;
  const int Hv = a.H / nb;
  int tps = 0;
  int ns = (a.split_ws != nullptr && nb == 1) ? mavlm_attention_hd_splits(a.R, a.S, a.H, &tps) : 1;
  if (ns <= 1) { ns = 1; tps = 0; }
  float* opart = a.split_ws;
  float* lpart = ns > 1 ? a.split_ws + (size_t)ns * a.R * a.H * HD : nullptr;
  static mavlm_per_device_once once;
  if constexpr (QG == 2) {
    auto kern = attn_fwd_hd2_kernel<T, HD, 3, 3, FT>;
    {
      hipError_t e = once.dyn_lds((const void*)kern, LDS);
      if (e != hipSuccess) return e;
    }
    hd_sk_plan plan = {};
    if (a.split_ws != nullptr) plan = hd2_plan(a.R, a.S, a.H);      // (the caller sized split_ws with mavlm_attention_hd_split_ws_floats)
    dim3 grid(((a.R + 127) / 128) * Hv, ns, nb);
    if (plan.wgs > 0) {
      grid = dim3(plan.wgs, 1, 1);
      ns = 1; tps = 0;
      lpart = a.split_ws + (size_t)plan.wgs * plan.nlev * 128 * HD;
    }
    {
      mavlm_prof_scope prof(MAVLM_K_ATTN, 4.0 * a.R * (double)a.S * a.H * HD, 2.0 * HD * a.H * (2.0 * a.R + 2.0 * a.S), s);
      hipLaunchKernelGGL(kern, grid, dim3(256), LDS, s, (const uint16_t*)a.Q, a.ldq, (const uint16_t*)a.K, a.ldk,
                         (const uint16_t*)a.V, a.ldv, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.S, Hv, c, opart, lpart, tps,
                         (long long)a.kv_bstride, plan, FT ? a.frame_scr : (float*)nullptr, FT ? a.frame_keys : 0, a.H);
    }
    if (plan.wgs > 0) {
      int cut = 0;
      for (int j = 0; j < plan.nlev; ++j) cut += plan.nun[j];
      mavlm_prof_scope prof(MAVLM_K_ATTN_MERGE, 0.0, (double)cut * 128 * HD * 4.0 * 2.0, s);
      hipLaunchKernelGGL((attn_combine_hd_sk_kernel<T, HD>), dim3(cut * 8), dim3(256), 0, s, opart, lpart, (uint16_t*)a.O, a.ldo,
                         a.lse2, a.R, Hv, plan);
      return hipGetLastError();
    }
  } else {
    auto kern = attn_fwd_hd_kernel<T, HD>;
    {
      hipError_t e = once.dyn_lds((const void*)kern, LDS);
      if (e != hipSuccess) return e;
    }
    mavlm_prof_scope prof(MAVLM_K_ATTN, 4.0 * a.R * (double)a.S * a.H * HD, 2.0 * HD * a.H * (2.0 * a.R + 2.0 * a.S), s);
    hipLaunchKernelGGL(kern, dim3(((a.R + 127) / 128) * Hv, ns, nb), dim3(512), LDS, s, (const uint16_t*)a.Q, a.ldq,
                       (const uint16_t*)a.K, a.ldk, (const uint16_t*)a.V, a.ldv, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.S, Hv, c,
                       opart, lpart, tps, (long long)a.kv_bstride);
  }
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

// Split-KV plan of the wide-head kernels (128-query workgroups, ONE per CU - 128 KiB of LDS; 32-key tiles).  Small grids
// (fewer than 200 units, at least 32 key tiles) cut every unit into ns equal key ranges: ns = the count (2 .. 8, at least 16
// tiles per range) that minimises  rounds(units x ns on 256 CUs) / ns  + a charge of 2 % of a unit per split for the fp32
// partials and their merge; ties -> fewer splits.  (Rounds 1-2 used ns = min(8, 400 / units): at the OneVision-7B shape - 104
// units - that is 3 splits = 312 workgroups = TWO rounds of a third each; 2 splits = 208 workgroups finish in ONE round of a
// half.)  Pure function of the shape, mirrored by oracle/memory_path.py split_plan_wide.
int mavlm_attention_hd_splits(int R, int S, int H, int* tiles_per_split) {
  const int items = ((R + 127) / 128) * H;
  const int nt = (S + KTH - 1) / KTH;
  int ns = 1;
  if (items < 200 && nt >= 32) {
    int cap = nt / 16;
    if (cap > 8) cap = 8;
    long best = 5040 + 0;                                   // ns = 1: one round of whole units (items < 256), no charge
    for (int c = 2; c <= cap; ++c) {
      const long rounds = ((long)items * c + 255) / 256;
      const long score = rounds * 5040 / c + 100L * c;      // 5040 = lcm(1..8): exact
      if (score < best) { best = score; ns = c; }
    }
  }
  int tps = (nt + ns - 1) / ns;
  ns = (nt + tps - 1) / tps;
  if (tiles_per_split) *tiles_per_split = ns > 1 ? tps : 0;
  return ns;
}

// fp32 floats of attention partials the wide-head forward of this shape wants (H = heads of ALL videos of a row batch): the
// stream-K plan of the 32-query-wave kernel (head_dim 448) or the split-KV form of the small single-video grids
size_t mavlm_attention_hd_split_ws_floats(int R, int S, int H, int head_dim) {
  if (head_dim == 448 && g_mavlm_attn_hd_qg != 1) {
    const size_t v = hd2_plan_floats(hd2_plan(R, S, H), head_dim);
    if (v) return v;
  }
  const int ns = mavlm_attention_hd_splits(R, S, H, nullptr);
  return ns > 1 ? (size_t)ns * R * H * head_dim + (size_t)ns * H * R : 0;
}
// workgroups of the stream-K plan (0 = the plan is not used for this shape); info[3] = {G, whole units per workgroup, levels}
int mavlm_attention_hd_streamk(int R, int S, int H, int head_dim, int info[3]) {
  hd_sk_plan p = {};
  if (head_dim == 448 && g_mavlm_attn_hd_qg != 1) p = hd2_plan(R, S, H);
  if (info) { info[0] = p.wgs; info[1] = p.full; info[2] = p.nlev; }
  return p.wgs;
}

int g_mavlm_attn_hd_qg = 0;      // tuning hook: 0 = automatic (32-query waves at head_dim 448), 1 = 16-query waves, 2 = 32-query
hipError_t mavlm_launch_attention_hd(const mavlm_attn_args& a, int head_dim, int dtype, hipStream_t s) {
  const bool f16 = dtype == MAVLM_F16;
  if (head_dim == 448) {
    if (g_mavlm_attn_hd_qg != 1) return f16 ? launch_fwd_hd<F16, 448, 2>(a, s) : launch_fwd_hd<BF16, 448, 2>(a, s);
    return f16 ? launch_fwd_hd<F16, 448, 1>(a, s) : launch_fwd_hd<BF16, 448, 1>(a, s);
  }
  if (head_dim == 128) return f16 ? launch_fwd_hd<F16, 128, 1>(a, s) : launch_fwd_hd<BF16, 128, 1>(a, s);
  // 4-head encoder of the inactive MemoryFuser variant (MemoryFuser.py:12-19): hidden 1024 / 896 over nhead = 4
  if (head_dim == 256) return f16 ? launch_fwd_hd<F16, 256, 1>(a, s) : launch_fwd_hd<BF16, 256, 1>(a, s);
  if (head_dim == 224) return f16 ? launch_fwd_hd<F16, 224, 1>(a, s) : launch_fwd_hd<BF16, 224, 1>(a, s);
  return hipErrorInvalidValue;
}

// Forward + frame scores in one pass at head_dim 448 (attn_fwd_hd2_kernel<.., FT = 1> + frame_tiles_kernel; the caller finishes with
// mavlm_launch_frame_finish over mavlm_attention_hd_frames_rows_per_video rows per video).  Runs whatever schedule the plain forward
// of the shape runs: context and log-sum-exp are bit-identical to mavlm_launch_attention_hd's.
bool mavlm_attention_hd_frames_supported(int R, int S, int H, int head_dim, int frame_keys) {
  // frame_keys > 32: at most one frame boundary per 32-key tile; % 4: a boundary never cuts a lane's 4-key group; the entries are
  // addressed through 32-bit offsets below the "dropped store" offset 0x7ffffff0
  return head_dim == 448 && frame_keys > KTH && (frame_keys & 3) == 0 && R > 0 && S > 0 && S % frame_keys == 0 &&
         S / frame_keys <= 64 && (double)H * R * ((S + KTH - 1) / KTH) * 8.0 < 2147483000.0;
}
size_t mavlm_attention_hd_frames_scr_floats(int R, int S, int H) { return (size_t)H * R * ((S + KTH - 1) / KTH) * 2; }
int mavlm_attention_hd_frames_rows_per_video(int R, int Hv) { return Hv * ((R + 63) / 64); }
size_t mavlm_attention_hd_frames_out_floats(int R, int S, int H, int frame_keys) {
  return (size_t)H * ((R + 63) / 64) * (S / frame_keys);
}
hipError_t mavlm_launch_attention_hd_frames(const mavlm_attn_args& a, int head_dim, int dtype, hipStream_t s) {
  if (!a.frame_scr || !a.frame_out || !a.lse2 || !mavlm_attention_hd_frames_supported(a.R, a.S, a.H, head_dim, a.frame_keys))
    return hipErrorInvalidValue;
  const int nb = a.nb > 0 ? a.nb : 1;
  if (a.H % nb != 0) return hipErrorInvalidValue;
  hipError_t e = dtype == MAVLM_F16 ? launch_fwd_hd<F16, 448, 2, 1>(a, s) : launch_fwd_hd<BF16, 448, 2, 1>(a, s);
  if (e != hipSuccess) return e;
  return mavlm_launch_frame_tiles(a.frame_scr, a.lse2, a.frame_out, a.R, a.H, a.H / nb, (a.S + KTH - 1) / KTH, KTH, a.frame_keys,
                                  a.S / a.frame_keys, s);
}

// entries [H][nt_all][R] of the tile-entry frame-score forms (this file: 32-key tiles; attention3.hip FR = 2: 64-key tiles) ->
// fout[video][(head, 64-row block)][frame]
hipError_t mavlm_launch_frame_tiles(const float* fent, const float* lse2, float* fout, int R, int H, int Hv, int nt_all,
                                    int tile_keys, int frame_keys, int FN, hipStream_t s) {
  if (!fent || !lse2 || !fout || R <= 0 || H <= 0 || Hv <= 0 || FN <= 0 || FN > 64) return hipErrorInvalidValue;
  mavlm_prof_scope prof(MAVLM_K_MISC, 0.0, (double)H * R * nt_all * 8.0, s);
  hipLaunchKernelGGL(frame_tiles_kernel, dim3((R + 63) / 64, H), dim3(256), 0, s, fent, lse2, fout, R, Hv, nt_all, frame_keys, FN,
                     tile_keys);
  return hipGetLastError();
}

hipError_t mavlm_launch_colsum_hd(const mavlm_colsum_args& a, int head_dim, int dtype, hipStream_t s) {
  const bool f16 = dtype == MAVLM_F16;
  if (head_dim == 448) return f16 ? launch_colsum_hd<F16, 448>(a, s) : launch_colsum_hd<BF16, 448>(a, s);
  if (head_dim == 128) return f16 ? launch_colsum_hd<F16, 128>(a, s) : launch_colsum_hd<BF16, 128>(a, s);
  return hipErrorInvalidValue;
}
