// attn_fwd3_kernel: software-pipelined flash attention (head_dim 128) - same math, MFMA mapping and LDS image as
// attn_fwd_kernel (attention.hip), restructured around what the in-kernel stamps of that kernel showed
// (cycles per 64-key tile per wave: QK^T 1100, softmax 880, P.V 1180, register->LDS staging 630):
//
//   * K/V tiles arrive by LDS-DMA (16-byte global_load_lds; the XOR swizzle of the image is applied to the per-lane
//     SOURCE address) - no VGPR staging, no ds_write phase.
//   * Each wave overlaps its own matrix and vector work: while the 16 MFMAs of S(t+1) = K(t+1).Q^T issue, the
//     exp2/fma of tile t's softmax run in their shadow; while the 16 MFMAs of O += V(t)^T.P(t)^T issue, the row
//     sums and 16-bit converts of tile t and the row maximum of tile t+1 run.  Two S tiles are live (the registers
//     the DMA staging freed).
//   * The tile loop is unrolled by two, so every LDS address is a loop-invariant VGPR plus an immediate.
//
// Ring: K(t+1) is consumed one iteration before V(t+1), so K and V each keep two 16 KiB slots:
//   iteration t reads K[(t+1)&1] and V[t&1]; at its top it issues the DMA of K(t+2) -> K[t&1] and V(t+1) -> V[(t+1)&1]
//   (both slots were last read in iteration t-1, before the closing barrier); it ends with vmcnt(0) + s_barrier.
#include "mavlm_common.h"
#include "mavlm_kernels.h"

namespace {

constexpr int HD3 = 128, KT3 = 64;
constexpr int TILE3 = KT3 * HD3 * 2;          // 16 KiB
constexpr int ATTN3_LDS = 4 * TILE3;          // K0 K1 V0 V1
constexpr float RESCALE3_LOG2 = 8.0f;         // same deferred-rescale rule as attn_fwd_kernel (mirrored by the oracle)
// LDS read-ahead, in MFMA steps, of the K fragments (phase A) and of the transposed V fragments (phase B).  One MFMA
// step is 32-64 clocks of matrix pipe, an LDS read under load returns after 100-200: with a read-ahead of ONE step every
// MFMA waited for its own operand (the ISA showed `s_waitcnt lgkmcnt(0)` in front of each of the 32 MFMAs of a tile and
// the kernel ran at 5.1k clocks per tile and wave against 1.0k of MFMA work).  4 VGPRs per step of read-ahead.
#ifndef MAVLM_ATTN3_KPF
#define MAVLM_ATTN3_KPF 3
#endif
#ifndef MAVLM_ATTN3_VPF
#define MAVLM_ATTN3_VPF 3
#endif

__device__ __forceinline__ int img3_x(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <int V>
struct IC { static constexpr int value = V; };

// A workgroup runs one or more SEGMENTS.  A segment = one unit (head h, 128-query block) x a contiguous range of its key
// tiles, computed with a fresh online-softmax state; a segment that covers all key tiles of its unit writes O / lse2, any
// other writes a NORMALISED fp32 partial + its log-sum-exp for a merge kernel.  Three schedules (plan.wgs / tps):
//   * plain (tps == 0, plan.wgs == 0): grid = units, one whole-unit segment per workgroup.
//   * split-KV (tps > 0; small grids): blockIdx.y owns the key tiles [y*tps, (y+1)*tps); attn_combine_kernel merges.
//   * levelled stream-K (plan.wgs = G > 0; more units than the chip has workgroup slots, long key sequences).  784 units on
//     512 slots leave the second round of a plain grid at 53 % occupancy.  Here G persistent workgroups first take
//     floor(U / G) whole units each; the remaining r = U mod G units are written as r = sum_k d_k G / 2^k (binary digits):
//     level k gives G / 2^k units to ALL G workgroups, each unit cut into 2^k equal key ranges - every workgroup is busy
//     for (1 + r / G) unit-times in total.  Unlike equal contiguous ranges of the global tile sequence (the textbook form:
//     every workgroup at a different key position, so a head's K / V beyond the 4 MiB L2 is re-fetched by everyone - measured
//     3.4 GB per launch at 12 544 keys, and -11 % at 125 k keys), the pieces of a level start at 2^k key positions only and
//     the workgroups sharing one advance together: the L2 holds 2^k moving windows.  Static schedule, no atomics, no queue:
//     deterministic; attn_combine_sk_kernel merges the 2^k partials of a cut unit in key order; the oracle mirrors the plan.
struct attn3_sk_plan {
  int wgs;           // G (0 = schedule not used)
  int qb;            // queries per unit: 128 (4-wave workgroups, G = 512) or 256 (8-wave workgroups, G = 256)
  int full;          // whole units per workgroup
  int nlev;          // levels in use
  int k[6];          // level: units are cut into 2^k key ranges
  int base[6];       // first unit of the level
  int nun[6];        // units of the level (<= G >> k)
  int slot[6];       // first partial slot of the level (+ virtual workgroup id)
  int affine;        // 1 = XCD-affine unit order (round 4, default), 0 = position p runs unit p (rounds 1-3; tuning hook)
};

// XCD-affine unit order (round 4).  The schedule hands out POSITIONS: whole-round position (si, v) = the v-th virtual workgroup's
// unit of round si; level position (lv, ul) = the ul-th unit of level lv (cut into 2^k pieces, run by the virtual workgroups
// [ul << k, (ul + 1) << k)).  Virtual ids are XCD-major (xcd_remap): XCD x owns v in [x W, (x + 1) W), W = G / 8.  Rounds 1-3
// mapped position p to unit p (head-major order): the 32 units an XCD works on at a time then straddled two heads in 15 of
// 24 (XCD, round) pairs of the bench launch - two heads' K / V (6.4 MB) against a 4 MiB L2, each head fetched by 2-3 XCDs
// (253 MB of reads per launch for 103 MB of operands).  Now the units are dealt to the XCDs FIRST: XCD x owns the
// contiguous unit range [C(x), C(x + 1)), C(x) = x full W + sum_lv min(nun_lv, x w_lv) (w_lv = W >> k_lv units of a level per
// XCD), and walks it in the order of its positions - its rounds, then its units of level 0, 1, ...  Bench launch (784 units,
// G = 256): C(x) = 98 x = two (video, head) pairs per XCD, one head at a time except in the round that crosses from the
// first to the second.  A bijection of the positions onto the units: which unit is cut (and how) changes, the plan does
// not; the merge kernel and the oracle (streamk_unit_of_position) apply the same map.
__host__ __device__ __forceinline__ int attn3_xcd_first_unit(const attn3_sk_plan& p, int x) {
  const int W = p.wgs >> 3;
  int c = x * p.full * W;
#pragma unroll
  for (int j = 0; j < 6; ++j)
    if (j < p.nlev) {
      const int t = x * (W >> p.k[j]);
      c += p.nun[j] < t ? p.nun[j] : t;
    }
  return c;
}
// whole-round position: round si, virtual workgroup v
__host__ __device__ __forceinline__ int attn3_unit_of_round(const attn3_sk_plan& p, int si, int v) {
  if (!p.affine) return si * p.wgs + v;
  const int W = p.wgs >> 3, x = v / W;
  return attn3_xcd_first_unit(p, x) + si * W + (v - x * W);
}
// level position: ul-th unit of level lv
__host__ __device__ __forceinline__ int attn3_unit_of_level(const attn3_sk_plan& p, int lv, int ul) {
  const int W = p.wgs >> 3;
  int k = p.k[0], b0 = p.base[0];
#pragma unroll
  for (int j = 1; j < 6; ++j)
    if (lv == j) { k = p.k[j]; b0 = p.base[j]; }
  if (!p.affine) return b0 + ul;
  const int w = W >> k, x = ul / w;
  int u = attn3_xcd_first_unit(p, x) + p.full * W + (ul - x * w);
#pragma unroll
  for (int j = 0; j < 6; ++j)
    if (j < lv) {                                              // this XCD's units of the earlier levels
      const int wj = W >> p.k[j];
      int n = p.nun[j] - x * wj;
      n = n < 0 ? 0 : (n > wj ? wj : n);
      u += n;
    }
  return u;
}

// NW = waves per workgroup: 4 (128 queries per unit, two workgroups per CU) or 8 (256 queries per unit, one workgroup per
// CU, used with the stream-K schedule): the 32 KiB of K / V per tile are then staged once for eight waves instead of
// twice for four each - half the LDS-DMA pieces per wave and per MFMA (the vector-memory path takes ~16 clocks per 1 KiB
// piece and every piece cost its issuing wave ~115 clocks in the in-kernel stamps).
// FR = 1: the frame-score variant (last formation layer, MemoryController.py:135-139).  The keys of a chunk are F frames
// of FP consecutive patches; the score of a frame is the mean over its patches of the column sums of the normalised
// probabilities.  Instead of a second pass that recomputes Q.K^T (attn_colsum3_kernel, 10 % of a video), every query row
// keeps the running mass a = sum_{keys of the current frame} exp2(s c - m c) next to its row sum, and writes (a, m) to a
// scratch entry [h][q][frame] when the key loop crosses a frame boundary (FP % 4 == 0: a boundary never cuts one of the
// 4-key groups a lane holds).  With the final log-sum-exp of the row known, the wave turns its 32 x F entries into F
// partial frame sums, sum_q a 2^(m c - lse2); frame_finish_kernel adds the partial sums of all waves in a fixed order.
// On the levelled stream-K schedule a cut unit's pieces start in the middle of a frame: piece p writes the part of frame
// f it saw to entry f + p of the row (a row has FNE = FN + 31 entries: pieces and frames both ascend along the keys, so
// f + p is unique), flushes its last, unfinished frame after its key loop, and attn_combine_sk_kernel - which knows the
// row's final log-sum-exp - turns the entries of the cut units into partial frame sums in the same [unit][wave] layout.
// An entry has exactly one writer: rows past R (clamped duplicates of row R-1) store to an offset behind the buffer
// descriptor's end, which the hardware drops.
//
// Row batch (HB < H): the launch serves B = H / HB independent videos.  "Head" index h = b HB + hv: queries / outputs of
// video b are the rows [b R, (b+1) R) of Q / O, its keys start kv_bs elements after video b-1's; lse2 is [B HB, R].
template <typename T, int NW, int FR = 0>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd3_kernel(const uint16_t* __restrict__ Q, int ldq,
                                                           const uint16_t* __restrict__ Kall, int ldk,
                                                           const uint16_t* __restrict__ Vall, int ldv,
                                                           uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2,
                                                           int R, int S_all, int H, float c, float* __restrict__ Opart,
                                                           float* __restrict__ lse_part, int tps, attn3_sk_plan plan,
                                                           int HB, long long kv_bs,
                                                           float* __restrict__ fscr = nullptr,
                                                           float* __restrict__ fout = nullptr, int FP = 0, int FN = 0,
                                                           int FNE = 0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int nt_all = (S_all + KT3 - 1) / KT3;
  constexpr int QB = 32 * NW;                                 // queries per unit
  constexpr int NPW = 16 / NW;                                // K (and V) pieces of a tile this wave stages: 4 or 2
  const int nqb = (R + QB - 1) / QB;
  const int w4 = wave & 3, wq = wave >> 2;                    // piece j of this wave = rows 16 (wq + (NW/4) j) + 4 w4 + ..

  // ---- schedule.  Levelled stream-K: XCD x owns the virtual ids [x G/8, (x+1) G/8), i.e. contiguous units of the
  // head-major order - its L2 holds the K / V of ~H/8 heads, and the 2^k workgroups that share a unit sit on one XCD.
  const int sk_v = plan.wgs > 0 ? xcd_remap((int)blockIdx.x, plan.wgs) : 0;
  const int nseg = plan.wgs > 0 ? plan.full + plan.nlev : 1;
  // static priority for the second-dispatched half of an 8-wave workgroup (MI355X_MICROARCH.md "Two waves per SIMD", item 4:
  // waves 4-7 are the arbitration losers of every segment otherwise): + 0.2-0.4 % measured, same box, interleaved
  // (tools/diag_ab_attn_prio.sh)
  if (NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  for (int si = 0; si < nseg; ++si) {
  // ---- this segment: unit (h, q-block), key tiles [t_lo, t_lo + nt)
  int h, qblk, t_lo, nt, out_kind, split = 0;                 // out_kind 0: O/lse2, 1: split-KV partial, 2: stream-K partial
  int sk_slot = 0, sk_piece = 0;
  if (plan.wgs > 0) {
    int u;
    if (si < plan.full) {
      u = attn3_unit_of_round(plan, si, sk_v);
      t_lo = 0;
      nt = nt_all;
      out_kind = 0;
    } else {
      const int lv = si - plan.full;
      int lk = plan.k[0], lnun = plan.nun[0], lslot = plan.slot[0];
#pragma unroll
      for (int j = 1; j < 6; ++j)                              // (static indices: the plan lives in scalar registers)
        if (lv == j) { lk = plan.k[j]; lnun = plan.nun[j]; lslot = plan.slot[j]; }
      const int ul = sk_v >> lk;
      if (ul >= lnun) continue;                                // this workgroup has no unit on this (partial) level
      const int piece = sk_v & ((1 << lk) - 1);
      u = attn3_unit_of_level(plan, lv, ul);
      t_lo = (int)(((long long)piece * nt_all) >> lk);
      nt = (int)(((long long)(piece + 1) * nt_all) >> lk) - t_lo;
      out_kind = 2;
      sk_slot = lslot + sk_v;
      sk_piece = piece;
      if (nt == 0) {                                           // fewer key tiles than pieces: a neutral partial (weight 0)
        float* pp = Opart + ((size_t)sk_slot * QB + wave * 32 + r) * HD3 + 64 * hh;
#pragma unroll
        for (int g = 0; g < 16; ++g) *(f32x4*)(pp + 4 * g) = f32x4{0.f, 0.f, 0.f, 0.f};
        if (hh == 0) lse_part[(size_t)sk_slot * QB + wave * 32 + r] = -INFINITY;
        continue;
      }
    }
    h = u / nqb;
    qblk = u - h * nqb;
  } else {
    h = blockIdx.x % H;                                       // one head per XCD L2 when H == 8
    qblk = blockIdx.x / H;
    split = blockIdx.y;
    t_lo = tps > 0 ? split * tps : 0;
    nt = tps > 0 ? ((nt_all - t_lo < tps) ? nt_all - t_lo : tps) : nt_all;
    out_kind = tps > 0 ? 1 : 0;
  }
  const int hb = h / HB, hv = h - hb * HB;                    // video of the row batch, head inside it
  const uint16_t* K = Kall + (size_t)hb * kv_bs + (size_t)t_lo * KT3 * ldk;
  const uint16_t* V = Vall + (size_t)hb * kv_bs + (size_t)t_lo * KT3 * ldv;
  const int S = (S_all - t_lo * KT3 < nt * KT3) ? S_all - t_lo * KT3 : nt * KT3;      // keys of this segment
  const int q0 = qblk * QB + wave * 32;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q0+r][h*128 + 16ks + 8hh + 0..7]
  typename T::vec8 qf[8];
  {
    int qrow = q0 + r;
    qrow = qrow < R ? qrow : R - 1;
    const uint16_t* qp = Q + ((size_t)hb * R + qrow) * ldq + hv * HD3 + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const typename T::vec8*)(qp + 16 * ks);
  }

  // ---- LDS-DMA geometry: instruction i (0..3) of wave w writes the 1 KiB block of rows 16 i + 4 w + (lane>>4).
  // Physical 16-B chunk p = lane & 15 of row `row` holds logical chunk p ^ x(row), x = ((row&3)<<2) | ((row>>2)&3)
  // = ((lane>>4)<<2) | w: independent of i, so the four sources of a tile differ by the uniform stride 16*ld.
  //
  // The DMAs are BUFFER loads (`buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`): a wave-uniform descriptor of this
  // head's K (V) columns, ONE loop-invariant 32-bit lane offset, and a scalar offset that carries the tile and the piece.
  // Issuing a piece is then `s_mov m0` + the load - no vector address arithmetic.  (In-kernel stamps of the previous
  // form - `global_load_lds` with 64-bit lane addresses rebuilt by VALU for every piece, scalars reloaded from spill
  // lanes - showed 980 of a wave's 3 700 clocks per tile going into ISSUING the eight DMAs.)  The descriptor ends after
  // the last valid row: rows past S of a ragged last tile read as zeros (they are masked to -inf / multiplied by p = 0).
  const int drow = 4 * w4 + (lane >> 4);                      // + 16 i
  const int dch = (lane & 15) ^ (((lane >> 4) << 2) | w4);
  auto head_rsrc = [&](const uint16_t* base, int ld) {
    const uintptr_t a = (uintptr_t)(base + hv * HD3);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint32_t bytes = __builtin_amdgcn_readfirstlane((uint32_t)(S - 1) * (uint32_t)ld * 2u + (uint32_t)HD3 * 2u);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, bytes, 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t krs = head_rsrc(K, ldk), vrs = head_rsrc(V, ldv);
  // lane offsets (bytes) of the four pieces of a tile: loop-invariant VGPRs (registers the kernel has to spare - scalar
  // registers it has not: with per-piece scalar offsets and LDS addresses hoisted out of the loop the compiler spilled
  // 64 SGPRs to vector lanes and read 38 of them back per tile)
  int koff[4], voff[4];   // (4, not NPW: hipcc drops the HOST stub of a kernel whose lambdas take value-dependent array refs)
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int i = wq + (NW / 4) * j;
    koff[j] = ((drow + 16 * i) * ldk + dch * 8) * 2;
    voff[j] = ((drow + 16 * i) * ldv + dch * 8) * 2;
  }
  const unsigned lds_wave = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem + w4 * 1024 + wq * 4096;   // + slot + j * 4096 * NW/4
  // piece i (0..3) of the tile at scalar byte offset `toff`: rows 16 i .. 16 i + 15.  A tile past the last one lies
  // entirely behind the descriptor's end: issuing it writes zeros into a slot nobody reads again, so the steady-state loop
  // needs no "is there a next tile" branch.
  auto dma_piece = [&](__amdgpu_buffer_rsrc_t rs, const int (&off)[4], int toff, int slot_off, int j) {
    unsigned base = lds_wave;
    asm volatile("" : "+s"(base));            // keep M0 = base + constant a one-instruction recompute (no hoisting)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (MAVLM_LDS void*)(uintptr_t)(base + slot_off + j * (NW / 4) * 4096), 16, off[j],
                                             toff, 0, 0);
  };
  auto dma_tile = [&](__amdgpu_buffer_rsrc_t rs, const int (&off)[4], int ld, int t, int slot_off) {
#pragma unroll
    for (int j = 0; j < NPW; ++j) dma_piece(rs, off, t * KT3 * ld * 2, slot_off, j);
  };
  char* const kbuf = smem;                 // K slots at 0, TILE3
  char* const vbuf = smem + 2 * TILE3;     // V slots at 2*TILE3, 3*TILE3

  // ---- fragment read geometry (loop-invariant VGPRs; stage / block / k-step offsets are immediates)
  const int xr = img3_x(r);
  int kaddr[8];                                               // K row read: + TILE3*slot + 8192*b
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kaddr[ks] = 256 * r + 16 * ((2 * ks + hh) ^ xr);
  unsigned kad[8];                                            // the same as 32-bit LDS addresses (inline-asm reads)
  const unsigned kbase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)kbuf;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) kad[ks] = kbase + (unsigned)kaddr[ks];
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg1 = (lane >> 4) & 1;
  const int v_rd = 256 * (4 * hh + tq) + 8 * (tp & 1) + 16 * ((tp >> 1) ^ hh);
  unsigned vaddr[4][2];                                       // [db][jj]: + TILE3*slot + 256*(32b+16s)
  const unsigned vbase = (unsigned)(uintptr_t)(MAVLM_LDS const char*)vbuf;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
      vaddr[db][jj] = vbase + v_rd + 256 * 8 * jj + 16 * (((db ^ tq) << 2) | ((tg1 ^ jj) << 1));

  f32x16 ot[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int i = 0; i < 16; ++i) ot[d][i] = 0.f;
  f32x16 st[2][2];                                            // [parity][key block]
  float m_run = -1e30f, l_run = 0.f;
  // frame-score state (FR): mass of the current frame (this lane's half of the keys), the frame and its end key
  float a_cur = 0.f;
  int f_cur = 0, f_end = FP;
  __amdgpu_buffer_rsrc_t frs;
  int f_voff = 0;
  if constexpr (FR != 0) {
    const uintptr_t fa = (uintptr_t)fscr;
    const uint32_t flo = __builtin_amdgcn_readfirstlane((uint32_t)fa);          // (uint32_t: readfirstlane returns int - no
    const uint32_t fhi = __builtin_amdgcn_readfirstlane((uint32_t)(fa >> 32));  //  sign extension into the high word)
    const uint32_t fbytes = __builtin_amdgcn_readfirstlane((uint32_t)H * (uint32_t)R * (uint32_t)(FR == 2 ? nt_all : FNE) * 8u);
    frs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)fhi << 32) | flo), 0, fbytes, 0x00020000);
    // entry [h][q][f + piece] = (a, m): 8 bytes.  Rows past R are clamped duplicates of row R-1 in every other respect; their
    // entry stores must land NOWHERE (a whole wave of duplicates can take a rescale the owning wave does not, and would
    // race with it): an offset behind the descriptor's end is dropped by the hardware, and the store stays issued, so the
    // vmcnt accounting of the tile loop does not change.
    f_voff = q0 + r < R ? ((h * R + q0 + r) * FNE + sk_piece) * 8 : 0x7ffffff0;
    // FR = 2 (the small grids that split their keys, round 4): one entry per (row, 64-key TILE) instead - [head][tile][row] x 8
    // bytes: log2 masses of the tile's keys before / behind the frame boundary inside it.  A (row, tile) has one writer under
    // every schedule, nothing is merged; frame_tiles_kernel (attention_hd.hip) adds them per frame once lse2 is final.
    if constexpr (FR == 2) f_voff = q0 + r < R ? ((h * nt_all + t_lo) * R + q0 + r) * 8 : 0x7ffffff0;
    // a piece of a cut unit starts at key t_lo * 64 of its unit: the frame that holds that key, and where it ends in
    // piece-local key numbers
    f_cur = (t_lo * KT3) / FP;
    f_end = (f_cur + 1) * FP - t_lo * KT3;
  }

  // S(t) = K(t).Q^T into st[P] (no overlap; used for tile 0 only)
  auto qk_plain = [&](auto par, int slot) {
    constexpr int P = decltype(par)::value;
    const char* kb = kbuf + slot * TILE3;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) st[P][b][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const typename T::vec8 kf = *(const typename T::vec8*)(kb + kaddr[i >> 1] + 8192 * (i & 1));
      st[P][i & 1] = T::mfma32(kf, qf[i >> 1], st[P][i & 1]);
    }
  };
  auto mask_tail = [&](auto par, int t) {
    constexpr int P = decltype(par)::value;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int key = t * KT3 + 32 * b + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (key >= S) st[P][b][i] = -INFINITY;
      }
  };
  // row maximum of S in st[P] across the wave's 64 keys; deferred-rescale decision (wave-uniform), applied to O and l
  auto max_and_rescale = [&](auto par) {
    constexpr int P = decltype(par)::value;
    float mx = st[P][0][0];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, st[P][b][i]);
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    if (__any((m_new - m_run) * c > RESCALE3_LOG2)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
    }
  };

  // ---- prologue: K(0), V(0), K(1) ; S(0) ; reference maximum for tile 0
  dma_tile(krs, koff, ldk, 0, 0);
  dma_tile(vrs, voff, ldv, 0, 2 * TILE3);
  if (nt > 1) dma_tile(krs, koff, ldk, 1, TILE3);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));   // Q resident before the loop (see attention.hip)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  qk_plain(IC<0>{}, 0);
  if (nt == 1 && (S & (KT3 - 1))) mask_tail(IC<0>{}, 0);
  max_and_rescale(IC<0>{});

  // ---- one pipelined iteration: S in st[P] (tile t, reference m_run already decided), S' of tile t+1 into st[P^1]
  auto iteration = [&](auto par, int t) {
    constexpr int P = decltype(par)::value;
    constexpr int N = P ^ 1;
    const bool has_next = t + 1 < nt;
    const int ktile = (t + 2) * KT3 * ldk * 2, vtile = (t + 1) * KT3 * ldv * 2;     // scalar byte offsets of K(t+2), V(t+1)
    // K(t+2) -> K slot (t&1) == P ; V(t+1) -> V slot N.  Both slots were last read before the previous barrier.  The
    // eight DMA pieces are issued ONE PER TWO MFMA STEPS of phase A, not as a burst at the top: the vector-memory path of
    // a CU takes a 1 KiB piece per ~16 clocks, and four waves issuing eight pieces each right after the barrier stalled
    // in issue for ~1 000 clocks per tile (in-kernel stamps) before their first MFMA.

    const float mc = m_run * c;
    // [A] S'(t+1) = K(t+1).Q^T  ||  p = exp2(s*c - mc) for tile t (2 elements per MFMA).  The order is pinned with
    // sched_barrier(0): left alone hipcc hoists all 32 v_exp in front of the MFMA chain (no overlap at all) and sinks
    // the fragment reads to their use; K fragments are read KPF steps ahead (MAVLM_ATTN3_KPF).
    if (has_next) {
      // K(t+1) sits in slot (t+1)&1 == N.  Like the V reads of [B] the K fragment reads are inline asm with hand-counted
      // lgkmcnt waits: left to hipcc, a read-ahead deeper than one step still ends in `s_waitcnt lgkmcnt(0)` (it drains
      // the reads it has just issued).  These are the only LGKM operations in flight in this block (one per step), so
      // the counts are exact: step i issues the read of step i + KPF, then waits until only the younger ones are out.
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[N][b][i] = 0.f;
      constexpr int KPF = MAVLM_ATTN3_KPF;
      static_assert(KPF >= 1 && KPF <= 15, "lgkmcnt is a 4-bit counter");
      u32x4 kfr[16];
      auto kread = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int off = N * TILE3 + 8192 * (i & 1);
        const unsigned a = kad[i >> 1];
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(off));
        kfr[i] = v;
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      kread(IC<0>{});
      if constexpr (KPF > 1) kread(IC<1>{});
      if constexpr (KPF > 2) kread(IC<2>{});
      if constexpr (KPF > 3) kread(IC<3>{});
      if constexpr (KPF > 4) kread(IC<4>{});
      if constexpr (KPF > 5) kread(IC<5>{});
      __builtin_amdgcn_sched_barrier(0);
      auto astep = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int ahead = (15 - i) < KPF ? (15 - i) : KPF;
        if constexpr (i + KPF < 16) kread(IC<(i + KPF < 16 ? i + KPF : 15)>{});
        if constexpr (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if constexpr (ahead == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if constexpr (ahead == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if constexpr (ahead == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        static_assert(KPF <= 6, "extend the wait table");
        __builtin_amdgcn_sched_barrier(0);                    // keep the MFMA below the wait (rule 18)
        st[N][i & 1] = T::mfma32(__builtin_bit_cast(typename T::vec8, kfr[i]), qf[i >> 1], st[N][i & 1]);
        constexpr int GAP = 8 / NPW;                          // one piece per GAP steps: K in steps 0-7, V in steps 8-15
        if constexpr (i % GAP == 0 && i < 8) dma_piece(krs, koff, ktile, P * TILE3, i / GAP);
        if constexpr (i % GAP == 0 && i >= 8) dma_piece(vrs, voff, vtile, (2 + N) * TILE3, (i - 8) / GAP);
        constexpr int e0 = 2 * i, e1 = 2 * i + 1;             // elements (b = e>>4, idx = e&15) of tile t
        // The two empty asm statements are ordered against sched_barrier (both have side effects); the pure
        // fma/exp between them cannot be hoisted in front of the MFMA chain or sunk behind it.
        float x0 = st[P][e0 >> 4][e0 & 15], x1 = st[P][e1 >> 4][e1 & 15];
        asm volatile("" : "+v"(x0), "+v"(x1));
        x0 = __builtin_amdgcn_exp2f(x0 * c - mc);
        x1 = __builtin_amdgcn_exp2f(x1 * c - mc);
        asm volatile("" : "+v"(x0), "+v"(x1));
        st[P][e0 >> 4][e0 & 15] = x0;
        st[P][e1 >> 4][e1 & 15] = x1;
        __builtin_amdgcn_sched_barrier(0);
      };
      astep(IC<0>{}); astep(IC<1>{}); astep(IC<2>{}); astep(IC<3>{}); astep(IC<4>{}); astep(IC<5>{}); astep(IC<6>{});
      astep(IC<7>{}); astep(IC<8>{}); astep(IC<9>{}); astep(IC<10>{}); astep(IC<11>{}); astep(IC<12>{});
      astep(IC<13>{}); astep(IC<14>{}); astep(IC<15>{});
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) st[P][b][i] = __builtin_amdgcn_exp2f(st[P][b][i] * c - mc);
    }

    // [B] O^T += V(t)^T.P(t)^T  ||  16-bit converts + row sums of tile t, row maximum of tile t+1
    if (has_next && t + 1 == nt - 1 && (S & (KT3 - 1))) mask_tail(IC<N>{}, t + 1);
    float psum = 0.f;
    float mx = has_next ? st[N][0][0] : 0.f;
    {
      // The transposed V reads go through inline asm with hand-counted lgkmcnt waits: through the builtin, hipcc puts
      // an s_waitcnt vmcnt(0) in front of the first one (it cannot prove that the read does not alias the LDS-DMA in
      // flight), which drains this iteration's DMAs of K(t+2) / V(t+1) in the middle of the tile.  These are the only
      // LGKM operations outstanding in this block (the K reads of [A] were consumed by its MFMAs), so the counts
      // are exact: step i issues the two reads of step i+1, then waits until only those two are outstanding.
      u32x2 vlo[16], vhi[16];
      typename T::vec8 pf[4];
      auto vread = [&](auto ic) {                             // step i = (bs = i>>2, db = i&3)
        constexpr int i = decltype(ic)::value;
        constexpr int bs = i >> 2, db = i & 3;
        constexpr int off = P * TILE3 + 256 * (32 * (bs >> 1) + 16 * (bs & 1));
        // (asm operands do not trigger implicit lambda capture: go through locals)
        const unsigned a0 = vaddr[db][0], a1 = vaddr[db][1];
        u32x2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a0), "i"(off));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a1), "i"(off));
        vlo[i] = lo;
        vhi[i] = hi;
      };
      auto cvt = [&](int bs) {                                // P^T fragment of keys 16 bs .. 16 bs + 15
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          w[j] = pack2<T>(st[P][bs >> 1][8 * (bs & 1) + 2 * j], st[P][bs >> 1][8 * (bs & 1) + 2 * j + 1]);
        pf[bs] = __builtin_bit_cast(typename T::vec8, w);
      };
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // nothing older than the reads below is outstanding
      constexpr int VPF = MAVLM_ATTN3_VPF;                    // read-ahead in steps (2 reads per step, lgkmcnt <= 15)
      static_assert(VPF >= 1 && VPF <= 7, "lgkmcnt is a 4-bit counter");
      vread(IC<0>{});
      if constexpr (VPF > 1) vread(IC<1>{});
      if constexpr (VPF > 2) vread(IC<2>{});
      if constexpr (VPF > 3) vread(IC<3>{});
      if constexpr (VPF > 4) vread(IC<4>{});
      if constexpr (VPF > 5) vread(IC<5>{});
      if constexpr (VPF > 6) vread(IC<6>{});
      cvt(0);
      __builtin_amdgcn_sched_barrier(0);
      auto step = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int bs = i >> 2, db = i & 3;
        // after issuing the reads of step i + VPF, the reads of steps i+1 .. min(i+VPF, 15) may stay outstanding
        constexpr int ahead = (15 - i) < VPF ? (15 - i) : VPF;
        if constexpr (i + VPF < 16) vread(IC<(i + VPF < 16 ? i + VPF : 15)>{});
        if constexpr (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if constexpr (ahead == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else if constexpr (ahead == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else if constexpr (ahead == 5) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
        else if constexpr (ahead == 6) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                    // keep the MFMA below the wait (rule 18)
        u32x4 both;
        both[0] = vlo[i][0]; both[1] = vlo[i][1]; both[2] = vhi[i][0]; both[3] = vhi[i][1];
        const typename T::vec8 vf = __builtin_bit_cast(typename T::vec8, both);
        ot[db] = T::mfma32(vf, pf[bs], ot[db]);
        if (db == 1 && bs < 3) cvt(bs + 1);                   // next group's converts, 2 MFMAs ahead of their use
        const int e0 = 8 * bs + 2 * db, e1 = e0 + 1;          // 2 of the 32 p values / 2 of the next 32 scores per MFMA
        psum += st[P][e0 >> 4][e0 & 15];
        psum += st[P][e1 >> 4][e1 & 15];
        if (has_next) mx = max3_asm(mx, st[N][e0 >> 4][e0 & 15], st[N][e1 >> 4][e1 & 15]);
        __builtin_amdgcn_sched_barrier(0);
      };
      step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{});
      step(IC<7>{}); step(IC<8>{}); step(IC<9>{}); step(IC<10>{}); step(IC<11>{}); step(IC<12>{}); step(IC<13>{});
      step(IC<14>{}); step(IC<15>{});
    }
    l_run += psum;
    bool flushed = false;                                     // (wave-uniform)
    if constexpr (FR == 2) {
      flushed = true;                                         // (an entry store in every tile)
      const int kofs = f_end - t * KT3;                       // keys [0, kofs) of the tile belong to the current frame
      float plo = psum, phi = 0.f;
      const bool cut = kofs <= KT3;                           // the frame ends inside this tile (or at its end)
      if (cut) {
        // as FR = 1 below: only the 32-key block that holds the boundary is looked at
        if (kofs <= 32) {
          plo = 0.f;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float s4 = (st[P][0][4 * g4] + st[P][0][4 * g4 + 1]) + (st[P][0][4 * g4 + 2] + st[P][0][4 * g4 + 3]);
            plo += (8 * g4 + 4 * hh < kofs) ? s4 : 0.f;
          }
          phi = fmaxf(psum - plo, 0.f);
        } else {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float s4 = (st[P][1][4 * g4] + st[P][1][4 * g4 + 1]) + (st[P][1][4 * g4 + 2] + st[P][1][4 * g4 + 3]);
            phi += (32 + 8 * g4 + 4 * hh >= kofs) ? s4 : 0.f;
          }
          plo = fmaxf(psum - phi, 0.f);
        }
        f_end += FP;
      }
      const float mc2 = m_run * c;
      const float vlo = __builtin_amdgcn_logf(xhalf_sum(plo)) + mc2;        // (v_log_f32 = log2; log2(0) = -inf: weight 0)
      float vhi = -INFINITY;
      if (cut && kofs < KT3) vhi = __builtin_amdgcn_logf(xhalf_sum(phi)) + mc2;
      if (hh == 0) {
        u32x2 e;
        e[0] = __builtin_bit_cast(unsigned, vlo);
        e[1] = __builtin_bit_cast(unsigned, vhi);
        __builtin_amdgcn_raw_buffer_store_b64(e, frs, f_voff, t * R * 8, 0);
      }
    }
    if constexpr (FR == 1) {
      const int k_end = (t + 1) * KT3;
      if (f_end <= k_end) {                                   // the current frame ends inside (or at the end of) this tile
        flushed = true;
        const int kofs = f_end - t * KT3;                     // keys [0, kofs) of the tile belong to it: 4 <= kofs <= 64
        // plo = this lane's mass of the keys below the boundary.  Only the 32-key block that holds the boundary is
        // looked at (4 of the lane's 8 four-key groups: keys 32 b + 8 g4 + 4 hh + 0..3); the other block is wholly on
        // one side and comes out of psum.
        float plo;
        if (kofs <= 32) {                                     // boundary in block 0: plo = its groups below the boundary
          plo = 0.f;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float s4 = (st[P][0][4 * g4] + st[P][0][4 * g4 + 1]) + (st[P][0][4 * g4 + 2] + st[P][0][4 * g4 + 3]);
            plo += (8 * g4 + 4 * hh < kofs) ? s4 : 0.f;
          }
        } else {                                              // boundary in block 1: psum minus its groups at / above it
          float phi = 0.f;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float s4 = (st[P][1][4 * g4] + st[P][1][4 * g4 + 1]) + (st[P][1][4 * g4 + 2] + st[P][1][4 * g4 + 3]);
            phi += (32 + 8 * g4 + 4 * hh >= kofs) ? s4 : 0.f;
          }
          plo = psum - phi;
        }
        const float a_done = xhalf_sum(a_cur + plo);          // both key halves of the row
        if (hh == 0) {
          u32x2 e;
          e[0] = __builtin_bit_cast(unsigned, a_done);
          e[1] = __builtin_bit_cast(unsigned, m_run);
          __builtin_amdgcn_raw_buffer_store_b64(e, frs, f_voff, f_cur * 8, 0);
        }
        a_cur = psum - plo;
        f_cur += 1;
        f_end += FP;
      } else {
        a_cur += psum;
      }
    }

    // reference maximum for tile t+1 (after P.V(t): the rescale touches O)
    if (has_next) {
      // the decision needs no exchange between the two lane halves of a query row: the row maximum exceeds the threshold iff
      // the maximum of one half does, and `__any` looks at all 64 lanes; the exchange sits in the (rare) rescale branch
      if (__any((fmaxf(m_run, mx) - m_run) * c > RESCALE3_LOG2)) {
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
        if constexpr (FR == 1) a_cur *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int i = 0; i < 16; ++i) ot[d][i] *= alpha;
      }
    }
    // this wave's DMAs of K(t+2), V(t+1) have landed.  A frame entry stored in this tile is the YOUNGEST vector-memory
    // operation (the DMAs were issued in phase A) and retires last: leaving it in flight keeps its ~1 000-clock round trip
    // off the critical path of the barrier (waiting for it cost ~20 us per launch).
    if (FR != 0 && flushed) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  int t = 0;
  for (; t + 1 < nt; t += 2) {
    iteration(IC<0>{}, t);
    iteration(IC<1>{}, t + 1);
  }
  if (t < nt) iteration(IC<0>{}, t);

  if constexpr (FR == 1) {
    // a piece that ends inside a frame: the mass of that frame's keys seen so far is still in a_cur (whole units end on a
    // frame boundary: S % FP == 0, the last flush happened in the loop and the next frame starts at key S)
    if (f_end - FP < S) {
      const float a_done = xhalf_sum(a_cur);
      if (hh == 0) {
        u32x2 e;
        e[0] = __builtin_bit_cast(unsigned, a_done);
        e[1] = __builtin_bit_cast(unsigned, m_run);
        __builtin_amdgcn_raw_buffer_store_b64(e, frs, f_voff, f_cur * 8, 0);
      }
    }
  }

  // ---- epilogue: O[q][h*128 + 32db + 8g + 4hh + 0..3] = O^T / l
  const float l_tot = xhalf_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int q = q0 + r;
  if (out_kind == 2) {                                        // stream-K partial: [slot][128 rows][128] + [slot][128]
    float* pp = Opart + ((size_t)sk_slot * QB + wave * 32 + r) * HD3 + 4 * hh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(f32x4*)(pp + 32 * db + 8 * g) = f32x4{ot[db][4 * g] * inv, ot[db][4 * g + 1] * inv, ot[db][4 * g + 2] * inv,
                                                ot[db][4 * g + 3] * inv};
    if (hh == 0) lse_part[(size_t)sk_slot * QB + wave * 32 + r] = m_run * c + log2f(l_tot);
  } else if (q < R) {
    if (out_kind == 1) {
      float* pp = Opart + ((size_t)split * R + q) * (H * HD3) + h * HD3 + 4 * hh;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(f32x4*)(pp + 32 * db + 8 * g) = f32x4{ot[db][4 * g] * inv, ot[db][4 * g + 1] * inv, ot[db][4 * g + 2] * inv,
                                                  ot[db][4 * g + 3] * inv};
      if (hh == 0) lse_part[((size_t)split * H + h) * R + q] = m_run * c + log2f(l_tot);
    } else {
      uint16_t* op = O + ((size_t)hb * R + q) * ldo + hv * HD3 + 4 * hh;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(u32x2*)(op + 32 * db + 8 * g) = pack4<T>(ot[db][4 * g] * inv, ot[db][4 * g + 1] * inv,
                                                     ot[db][4 * g + 2] * inv, ot[db][4 * g + 3] * inv);
      if (lse2 != nullptr && hh == 0) lse2[(size_t)h * R + q] = m_run * c + log2f(l_tot);
    }
  }
  if (FR == 1 && out_kind == 0) {                             // whole unit: the row's final log-sum-exp is known here
    // Lane j takes frame j (FN <= 64): for each of the wave's 32 queries it reads the (a, m) entry [h][q][j] - one 8-byte
    // load per lane, a query's F entries are contiguous - and adds a 2^(m c - lse2[q]).  The entries were written by this
    // wave (all stores retired by the vmcnt(0) that ended the last tile); nobody else has touched these lines.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every entry of this wave is in memory
    const float my_lse = m_run * c + log2f(l_tot);            // lane (r, hh): query q0 + r
    float fsum = 0.f;
    const int fj = lane < FN ? lane : FN - 1;
    // all 64 loads are issued before the first use (the accumulators are dead, registers are free): as a loop of dependent
    // batches this epilogue cost ~10 us per unit
    float ea[32], em[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      int qi = q0 + i;
      qi = qi < R ? qi : R - 1;
      // (two dword loads: with the b64 form of the builtin hipcc emitted ONE buffer_load_dword and used it for both halves)
      const int eoff = ((h * R + qi) * FNE + fj) * 8;
      ea[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, eoff, 0, 1 /* glc: not from L1 */));
      em[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(frs, eoff + 4, 0, 1));
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const float lse_i = __shfl(my_lse, i);                  // lanes i and i + 32 hold the same row
      const bool ok = q0 + i < R && lane < FN;
      fsum += ok ? ea[i] * __builtin_amdgcn_exp2f(em[i] * c - lse_i) : 0.f;
    }
    if (lane < FN) fout[((size_t)(h * nqb + qblk) * NW + wave) * FN + lane] = fsum;
  }
  }   // segment loop (every tile iteration ends with a barrier: the LDS slots are free for the next segment's prologue)
}

// Stream-K merge: one workgroup per 32 query rows of a cut unit (QB = 128 or 256 rows).  Unit `ul` of level j (cut into 2^k key ranges) has its partials in the slots
// plan.slot[j] + (ul << k) + p, p = 0 .. 2^k - 1 in key order: O = sum_p w_p O_p, w_p = 2^(lse_p - lse), lse = log2 sum_p
// 2^lse_p (an empty range carries lse = -inf: weight 0).  A 512-byte partial row = 32 lanes x 16 bytes: a wave-instruction
// covers two whole rows, the workgroup eight.
// (NP = 2^k is a compile-time constant per level so that the 2^k log-sum-exp loads and the 2^k partial-row loads of a query
// row are issued back to back: as run-time loops each load waited for the previous one - 96 dependent round trips per row
// at k = 5, 283 us for the whole merge against 600 us for the attention itself.)
// FR: the frame-score variant (see attn_fwd3_kernel).  The pieces of a cut unit left (a, m) entries for the parts of the
// frames they saw; with the merged log-sum-exp of a row known here, lane c4 of the row's 32 lanes takes frames c4 and c4 + 32
// and adds a 2^(m c - lse) over the pieces that hold keys of the frame; the workgroup's 32 rows are then added in a fixed
// order (LDS) into ONE partial frame sum per (unit, 32-row group) - the slot a whole unit's wave fills in the main kernel.
struct attn3_frames_args {
  const float* scr;    // [H][R][FNE] entries (a, m)
  float* out;          // [units][QB / 32][FN] partial frame sums
  int FP, FN, FNE, S;  // keys per frame, frames, entries per row, keys
  float c;
};

template <typename T, int NP, int FR>
__device__ __forceinline__ void combine_sk_rows(const float* __restrict__ Opart, const float* __restrict__ lse_part, size_t s0,
                                                uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2, int R, int h, int qblk,
                                                int quarter, int QB, int HB, int lk, int u, const attn3_frames_args& fa,
                                                float* fsh) {
  const int c4 = threadIdx.x & 31, rsub = threadIdx.x >> 5;
  const int hb = h / HB, hv = h - hb * HB;
  float fs[2] = {0.f, 0.f};
  // frame f = c4 + 32 ff: the pieces [fp0, fp1] that hold its keys (piece p owns the key tiles [(p nt) >> lk, ((p+1) nt) >> lk))
  int fp0[2] = {0, 0}, fp1[2] = {-1, -1};
  const int nt_all_ = FR != 0 ? (fa.S + KT3 - 1) / KT3 : 1;
  if constexpr (FR != 0) {
    const int nt_all = (fa.S + KT3 - 1) / KT3;
    auto piece_of = [&](int tile) {                            // the (non-empty) piece that owns key tile `tile`
      int p = (int)(((long long)tile << lk) / nt_all);
      while (p + 1 < (1 << lk) && (int)(((long long)(p + 1) * nt_all) >> lk) <= tile) ++p;
      return p;
    };
#pragma unroll
    for (int ff = 0; ff < 2; ++ff) {
      const int f = c4 + 32 * ff;
      if (f < fa.FN) {
        fp0[ff] = piece_of((f * fa.FP) / KT3);
        fp1[ff] = piece_of(((f + 1) * fa.FP - 1) / KT3);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {                             // a workgroup merges 32 of the unit's rows
    const int row = quarter * 32 + it * 8 + rsub;
    const int q = qblk * QB + row;
    if (q >= R) continue;
    float l[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) l[p] = lse_part[(s0 + p) * QB + row];
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
      for (int j = 0; j < CH; ++j) a[j] = *(const f32x4*)(Opart + ((s0 + p0 + j) * QB + row) * HD3 + 4 * c4);
#pragma unroll
      for (int j = 0; j < CH; ++j) {                           // key order
        const float w = __builtin_amdgcn_exp2f(l[p0 + j] - lse);
        acc[0] += w * a[j][0]; acc[1] += w * a[j][1]; acc[2] += w * a[j][2]; acc[3] += w * a[j][3];
      }
    }
    *(u32x2*)(O + ((size_t)hb * R + q) * ldo + hv * HD3 + 4 * c4) = pack4<T>(acc[0], acc[1], acc[2], acc[3]);
    if (lse2 != nullptr && c4 == 0) lse2[(size_t)h * R + q] = lse;
    if constexpr (FR != 0) {
      const float* er = fa.scr + ((size_t)h * R + q) * fa.FNE * 2;
#pragma unroll
      for (int ff = 0; ff < 2; ++ff) {
        const int f = c4 + 32 * ff;
        for (int p = fp0[ff]; p <= fp1[ff]; ++p) {             // (empty for f >= FN)
          // pieces without key tiles (fewer tiles than pieces) wrote no entry
          if ((int)(((long long)(p + 1) * nt_all_) >> lk) == (int)(((long long)p * nt_all_) >> lk)) continue;
          const f32x2 e = *(const f32x2*)(er + 2 * (f + p));
          fs[ff] += e[0] * __builtin_amdgcn_exp2f(e[1] * fa.c - lse);
        }
      }
    }
  }
  if constexpr (FR != 0) {
    fsh[rsub * 64 + c4] = fs[0];
    fsh[rsub * 64 + 32 + c4] = fs[1];
    __syncthreads();
    if (threadIdx.x < 64 && (int)threadIdx.x < fa.FN) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += fsh[i * 64 + threadIdx.x];
      fa.out[((size_t)u * (QB / 32) + quarter) * fa.FN + threadIdx.x] = t;
    }
  }
}

template <typename T, int FR>
__global__ __launch_bounds__(256) void attn_combine_sk_kernel(const float* __restrict__ Opart, const float* __restrict__ lse_part,
                                                              uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2,
                                                              int R, int H, attn3_sk_plan plan, int QB, int HB,
                                                              attn3_frames_args fa) {
  __shared__ float fsh[FR != 0 ? 512 : 1];
  const int wpu = QB / 32;                                     // workgroups per cut unit (32 query rows each)
  int b = blockIdx.x / wpu, lk = 0, lv = 0, lslot = 0;
  const int quarter = blockIdx.x - b * wpu;
  bool found = false;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if (!found && j < plan.nlev) {
      if (b < plan.nun[j]) { lk = plan.k[j]; lv = j; lslot = plan.slot[j]; found = true; }
      else b -= plan.nun[j];
    }
  }
  if (!found) return;
  const int nqb = (R + QB - 1) / QB;
  const int u = attn3_unit_of_level(plan, lv, b);              // (the XCD-affine unit order of attn_fwd3_kernel)
  const int h = u / nqb, qblk = u - h * nqb;
  const size_t s0 = (size_t)lslot + ((size_t)b << lk);
  switch (lk) {
    case 1: combine_sk_rows<T, 2, FR>(Opart, lse_part, s0, O, ldo, lse2, R, h, qblk, quarter, QB, HB, lk, u, fa, fsh); break;
    case 2: combine_sk_rows<T, 4, FR>(Opart, lse_part, s0, O, ldo, lse2, R, h, qblk, quarter, QB, HB, lk, u, fa, fsh); break;
    case 3: combine_sk_rows<T, 8, FR>(Opart, lse_part, s0, O, ldo, lse2, R, h, qblk, quarter, QB, HB, lk, u, fa, fsh); break;
    case 4: combine_sk_rows<T, 16, FR>(Opart, lse_part, s0, O, ldo, lse2, R, h, qblk, quarter, QB, HB, lk, u, fa, fsh); break;
    default: combine_sk_rows<T, 32, FR>(Opart, lse_part, s0, O, ldo, lse2, R, h, qblk, quarter, QB, HB, lk, u, fa, fsh); break;
  }
}

// O[q, h*128+d] = sum_s w_s Opart[s][q][h*128+d], w_s = 2^(lse_s - lse), lse = log2 sum_s 2^lse_s.  One wave per query
// row, 16 columns per lane and 1024-column chunk (8 lanes per head).
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ Opart, const float* __restrict__ lse_part,
                                                           uint16_t* __restrict__ O, int ldo, float* __restrict__ lse2,
                                                           int R, int H, int hd, int ns) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= R) return;
  const int W = H * hd;                                      // hd % 16 == 0: a lane's 16 columns lie in one head
  for (int c0 = 0; c0 < W; c0 += 1024) {
    const int col = c0 + lane * 16;
    if (col >= W) continue;
    const int h = col / hd;
    float mx = -INFINITY;
    for (int sp = 0; sp < ns; ++sp) mx = fmaxf(mx, lse_part[((size_t)sp * H + h) * R + q]);
    float den = 0.f;
    for (int sp = 0; sp < ns; ++sp) den += __builtin_amdgcn_exp2f(lse_part[((size_t)sp * H + h) * R + q] - mx);
    const float lse = mx + log2f(den);
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int sp = 0; sp < ns; ++sp) {
      const float wgt = __builtin_amdgcn_exp2f(lse_part[((size_t)sp * H + h) * R + q] - lse);
      const float* pp = Opart + ((size_t)sp * R + q) * W + col;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const f32x4 x = *(const f32x4*)(pp + 4 * v);
        acc[4 * v] += wgt * x[0]; acc[4 * v + 1] += wgt * x[1]; acc[4 * v + 2] += wgt * x[2]; acc[4 * v + 3] += wgt * x[3];
      }
    }
    uint16_t* op = O + (size_t)q * ldo + col;
#pragma unroll
    for (int v = 0; v < 4; ++v) *(u32x2*)(op + 4 * v) = pack4<T>(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
    if (lse2 != nullptr && col % hd == 0) lse2[(size_t)h * R + q] = lse;
  }
}

}  // namespace

hipError_t mavlm_launch_attention_combine(const float* opart, const float* lpart, void* O, int ldo, float* lse2, int R, int H,
                                          int hd, int ns, int dtype, hipStream_t s) {
  if (dtype == MAVLM_F16)
    hipLaunchKernelGGL(attn_combine_kernel<F16>, dim3((R + 3) / 4), dim3(256), 0, s, opart, lpart, (uint16_t*)O, ldo, lse2, R, H,
                       hd, ns);
  else
    hipLaunchKernelGGL(attn_combine_kernel<BF16>, dim3((R + 3) / 4), dim3(256), 0, s, opart, lpart, (uint16_t*)O, ldo, lse2, R, H,
                       hd, ns);
  return hipGetLastError();
}

// Split-KV plan: grids that fill less than ~60 % of the 512 workgroup slots (2 per CU) split the keys over
// blockIdx.y.  Deterministic function of the shape: the fused step and the stand-alone operator take the same path.
int mavlm_attention_splits(int R, int S, int H, int* tiles_per_split) {
  const int items = ((R + 127) / 128) * H;
  const int nt = (S + KT3 - 1) / KT3;
  int ns = 1;
  if (items < 320 && nt >= 16) {
    ns = 512 / items;
    if (ns > 8) ns = 8;
    if (ns > nt / 8) ns = nt / 8;
    if (ns < 2) ns = 1;
  }
  int tps = (nt + ns - 1) / ns;
  ns = (nt + tps - 1) / tps;
  if (tiles_per_split) *tiles_per_split = ns > 1 ? tps : 0;
  return ns;
}

// Levelled stream-K plan (see attn_fwd3_kernel).  Used when the units (128-query blocks x heads) exceed the ATTN3_SK_WGS
// workgroup slots of the chip (256 CUs x 2), a plain grid would leave more than 5 % of its last round empty, and a unit has
// at least g_mavlm_attn_sk_min_tiles key tiles.  The schedule costs a prologue per segment, 64 KiB of fp32 partial per cut
// and the merge kernel, and the plain grid's under-filled last round already runs ~1.6x faster per workgroup (one per
// CU): measured on one device at 784 units, S = 6272 / 12 544 / 25 088 / 62 720 / 125 440 keys: 362 -> 334, 683 -> 612,
// 1331 -> 1172, 3404 -> 2883, 6964 -> 5710 us (889 -> 964 ... 926 -> 1129 TFLOP/s); below ~64 tiles per unit the fixed
// costs win.
// The remainder r = U mod G is served by levels k = 1 .. 5 (2-way .. 32-way cuts); what is left after level 4 (fewer than
// G/16 units) goes into one or two 32-way levels.  Pure function of the shape (mirrored by oracle/memory_path.py).
constexpr int ATTN3_SK_WGS = 512;
int g_mavlm_attn_sk_min_tiles = 64;          // tuning / test hook (mavlm_set_attention_streamk_min_tiles)
int g_mavlm_attn_sk_waves = 0;                // tuning hook: 0 = automatic, 4 / 8 = waves per stream-K workgroup
int g_mavlm_attn_unit_order = 1;              // tuning hook: 1 = XCD-affine unit order, 0 = position order (rounds 1-3)
static attn3_sk_plan attn3_plan_for(int R, int S, int H, int waves) {
  attn3_sk_plan p = {};
  const int QB = 32 * waves, G = ATTN3_SK_WGS * 4 / waves;    // 512 four-wave or 256 eight-wave workgroups fill 256 CUs
  const long units = (long)((R + QB - 1) / QB) * H;
  if (units <= G || (S + KT3 - 1) / KT3 < g_mavlm_attn_sk_min_tiles) return p;
  const long rounds = (units + G - 1) / G;
  if ((double)units / (double)(rounds * G) >= 0.95) return p;
  p.wgs = G;
  p.qb = QB;
  p.affine = g_mavlm_attn_unit_order;
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
static attn3_sk_plan attn3_plan(int R, int S, int H) {
  if (g_mavlm_attn_sk_waves != 4) {
    const attn3_sk_plan p8 = attn3_plan_for(R, S, H, 8);
    if (p8.wgs > 0 || g_mavlm_attn_sk_waves == 8) return p8;
  }
  return attn3_plan_for(R, S, H, 4);
}
int mavlm_attention_streamk_wgs(int R, int S, int H) { return attn3_plan(R, S, H).wgs; }
void mavlm_attention_plan_info(int R, int S, int H, int info[4]) {
  const attn3_sk_plan pl = attn3_plan(R, S, H);
  info[0] = pl.wgs > 0 && pl.qb == 256 ? 8 : 4;
  info[1] = pl.wgs;
  info[2] = pl.nlev;
  info[3] = pl.wgs > 0 ? 1 : mavlm_attention_splits(R, S, H, nullptr);
  if (info[3] < 1) info[3] = 1;
}

// unit (head-major index h * nqb + q-block) at a schedule position: lv < 0: whole round `a`, virtual workgroup `b`; lv >= 0:
// the b-th unit of level lv.  -1 = no stream-K plan / out of range.  (CPU mirror test of the XCD-affine unit order.)
int mavlm_attention_plan_unit_(int R, int S, int H, int lv, int a, int b) {
  const attn3_sk_plan pl = attn3_plan(R, S, H);
  if (pl.wgs <= 0) return -1;
  if (lv < 0) return (a >= 0 && a < pl.full && b >= 0 && b < pl.wgs) ? attn3_unit_of_round(pl, a, b) : -1;
  return (lv < pl.nlev && b >= 0 && b < pl.nun[lv]) ? attn3_unit_of_level(pl, lv, b) : -1;
}

size_t mavlm_attention_split_ws_floats(int R, int S, int H) {
  const attn3_sk_plan pl = attn3_plan(R, S, H);
  if (pl.wgs > 0) return (size_t)pl.wgs * pl.nlev * ((size_t)pl.qb * HD3 + pl.qb);
  const int ns = mavlm_attention_splits(R, S, H, nullptr);
  return ns > 1 ? (size_t)ns * R * H * HD3 + (size_t)ns * H * R : 0;
}

// workspace that covers the plan of either workgroup shape (4 / 8 waves): for buffers sized before the tuning hook
// mavlm_set_attention_streamk_waves may change
size_t mavlm_attention_split_ws_floats_max(int R, int S, int H) {
  size_t m = mavlm_attention_split_ws_floats(R, S, H);
  for (int w = 4; w <= 8; w += 4) {
    const attn3_sk_plan pl = attn3_plan_for(R, S, H, w);
    const size_t v = pl.wgs > 0 ? (size_t)pl.wgs * pl.nlev * ((size_t)pl.qb * HD3 + pl.qb) : 0;
    if (v > m) m = v;
  }
  return m;
}

namespace {
struct attn3_launch_args {
  dim3 grid;
  float c;
  float* opart;
  float* lpart;
  int tps, ns, cut_units;
  attn3_sk_plan plan;
  int HB;              // heads per video (a.H / a.nb)
  int FNE;             // frame-score variant: entries per row of the scratch
};
template <typename T, int NW, int FR>
hipError_t attn3_launch(const mavlm_attn_args& a, const attn3_launch_args& la, hipStream_t s) {
  auto kern = attn_fwd3_kernel<T, NW, FR>;
  static mavlm_per_device_once once;
  hipError_t e = once.dyn_lds((const void*)kern, ATTN3_LDS);
  if (e != hipSuccess) return e;
  const int FN = FR ? a.S / a.frame_keys : 0;
  {  // (bench.py's instrumented pass: the main kernel and the merge are bracketed separately, so the main kernel's average
     // is directly comparable with its line in a rocprofv3 summary)
    mavlm_prof_scope prof(FR ? MAVLM_K_ATTN_FRAMES : MAVLM_K_ATTN, 4.0 * a.R * (double)a.S * a.H * HD3,
                          2.0 * HD3 * a.H * (2.0 * a.R + 2.0 * a.S), s);
    hipLaunchKernelGGL(kern, la.grid, dim3(64 * NW), ATTN3_LDS, s, (const uint16_t*)a.Q, a.ldq, (const uint16_t*)a.K, a.ldk,
                       (const uint16_t*)a.V, a.ldv, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.S, a.H, la.c, la.opart, la.lpart,
                       la.tps, la.plan, la.HB, (long long)a.kv_bstride, FR ? a.frame_scr : (float*)nullptr,
                       FR ? a.frame_out : (float*)nullptr, FR ? a.frame_keys : 0, FN, la.FNE);
  }
  double parts = la.ns > 1 ? (double)la.ns * a.R * a.H : 0.0;      // fp32 partial rows the merge reads
  if (la.plan.wgs > 0) {
    parts = 0.0;
    for (int j = 0; j < la.plan.nlev; ++j) parts += (double)la.plan.nun[j] * (1 << la.plan.k[j]) * la.plan.qb;
  }
  mavlm_prof_scope prof(la.plan.wgs > 0 ? (la.cut_units > 0 ? MAVLM_K_ATTN_MERGE : -1) : (la.ns > 1 ? MAVLM_K_ATTN_MERGE : -1),
                        0.0, parts * 4.0 * (HD3 + 1), s);
  if (la.plan.wgs > 0 && la.cut_units > 0) {
    attn3_frames_args fa = {};
    constexpr int FRM = FR == 1 ? 1 : 0;                      // (FR = 2: the tile entries need no merge)
    if (FRM) fa = attn3_frames_args{a.frame_scr, a.frame_out, a.frame_keys, FN, la.FNE, a.S, la.c};
    hipLaunchKernelGGL((attn_combine_sk_kernel<T, FRM>), dim3(la.cut_units * (la.plan.qb / 32)), dim3(256), 0, s, la.opart,
                       la.lpart, (uint16_t*)a.O, a.ldo, a.lse2, a.R, a.H, la.plan, la.plan.qb, la.HB, fa);
  } else if (la.ns > 1) {
    hipLaunchKernelGGL(attn_combine_kernel<T>, dim3((a.R + 3) / 4), dim3(256), 0, s, la.opart, la.lpart, (uint16_t*)a.O, a.ldo,
                       a.lse2, a.R, a.H, HD3, la.ns);
  }
  if constexpr (FR == 2) {                                    // lse2 is final: tile entries -> partial frame sums
    hipError_t e2 = mavlm_launch_frame_tiles(a.frame_scr, a.lse2, a.frame_out, a.R, a.H, la.HB, (a.S + KT3 - 1) / KT3, KT3,
                                             a.frame_keys, FN, s);
    if (e2 != hipSuccess) return e2;
  }
  return hipGetLastError();
}
}  // namespace

int g_mavlm_frame_score_mode = 1;   // mavlm_step: 1 = frame scores fused into the last layer's forward, 0 = column-sum pass

// scores[b][f] = (1 / P) * sum of the partial frame sums of video b (rows [b rows, (b+1) rows) of fout), added in a fixed
// order (one workgroup per (frame, video))
template <typename T>
__global__ __launch_bounds__(256) void frame_finish_kernel(const float* __restrict__ fout, int rows, int F, int P,
                                                           void* __restrict__ out, int out_f32) {
  __shared__ float wsum[4];
  const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* fo = fout + (size_t)b * rows * F;
  float s = 0.f;
  for (int i = tid; i < rows; i += 256) s += fo[(size_t)i * F + f];
  s = wave_sum(s);
  if ((tid & 63) == 0) wsum[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    s = (((wsum[0] + wsum[1]) + wsum[2]) + wsum[3]) / (float)P;
    if (out_f32) ((float*)out)[(size_t)b * F + f] = s;
    else ((uint16_t*)out)[(size_t)b * F + f] = T::from_f32(s);
  }
}

// Frame-score variant of the forward (fused step, last formation layer).  Runs whatever schedule the plain forward of the
// same shape runs (plain grid or levelled stream-K - never the split-KV form of the small grids, which keep the column-sum
// pass): context and log-sum-exp are bit-identical to mavlm_launch_attention3's.
constexpr int ATTN3_FR_EXTRA = 31;      // extra entries per row: a unit is cut into at most 32 pieces
bool mavlm_attention_frames_supported(int R, int S, int H, int frame_keys) {
  // frame_keys >= 64: at most one frame boundary per 64-key tile; % 4: a boundary never cuts a lane's 4-key group;
  // the scratch is addressed through 32-bit offsets below the "dropped store" offset 0x7ffffff0
  return frame_keys >= KT3 && (frame_keys & 3) == 0 && S > 0 && S % frame_keys == 0 && S / frame_keys <= 64 &&
         (double)H * R * (64.0 + ATTN3_FR_EXTRA) * 8.0 < 2147483000.0;
}
size_t mavlm_attention_frames_scr_floats(int R, int S, int H, int frame_keys) {
  return (size_t)H * R * (S / frame_keys + ATTN3_FR_EXTRA) * 2;
}
// one partial frame sum per (unit, 32-query group): units of 128 or 256 queries cover ceil(R/128)*4 or ceil(R/256)*8 groups
size_t mavlm_attention_frames_out_rows(int R, int H) {
  const size_t g4 = (size_t)((R + 127) / 128) * 4, g8 = (size_t)((R + 255) / 256) * 8;
  return (size_t)H * (g4 > g8 ? g4 : g8);
}
size_t mavlm_attention_frames_out_floats(int R, int S, int H, int frame_keys) {
  return mavlm_attention_frames_out_rows(R, H) * (S / frame_keys);
}

hipError_t mavlm_launch_frame_finish(const float* fout, int rows, int nb, int F, int P, void* out, int out_f32, int dtype,
                                     hipStream_t s) {
  if (!fout || !out || rows <= 0 || F <= 0 || P <= 0 || nb <= 0) return hipErrorInvalidValue;
  mavlm_prof_scope prof(MAVLM_K_MISC, 0.0, 4.0 * rows * (double)F * nb, s);
  if (dtype == MAVLM_F16) hipLaunchKernelGGL(frame_finish_kernel<F16>, dim3(F, nb), dim3(256), 0, s, fout, rows, F, P, out, out_f32);
  else hipLaunchKernelGGL(frame_finish_kernel<BF16>, dim3(F, nb), dim3(256), 0, s, fout, rows, F, P, out, out_f32);
  return hipGetLastError();
}

// rows of `frame_out` per video that mavlm_launch_attention3_frames fills for this shape (the argument of the finish kernel)
int mavlm_attention_frames_rows_per_video(const mavlm_attn_args& a) {
  const int nb = a.nb > 0 ? a.nb : 1;
  attn3_sk_plan plan = {};
  if (a.split_ws != nullptr) plan = attn3_plan(a.R, a.S, a.H);
  const int qb = plan.wgs > 0 ? plan.qb : 128;
  return (a.H / nb) * ((a.R + qb - 1) / qb) * (qb / 32);
}

static hipError_t attn3_dispatch(const mavlm_attn_args& a, int dtype, int frames, hipStream_t s) {      // frames: 0 / 1 (a, m) / 2 tile entries
  // K / V are addressed through 32-bit buffer offsets (one descriptor per head): the key block must span < 4 GiB
  // (2 GiB: the scalar offset of the tile after the last one must not wrap either)
  if ((double)a.S * a.ldk * 2.0 >= 2147483648.0 || (double)a.S * a.ldv * 2.0 >= 2147483648.0) return hipErrorInvalidValue;
  const int nb = a.nb > 0 ? a.nb : 1;
  if (a.H % nb != 0) return hipErrorInvalidValue;
  if (frames == 1 && (!a.frame_scr || !a.frame_out || !mavlm_attention_frames_supported(a.R, a.S, a.H, a.frame_keys)))
    return hipErrorInvalidValue;
  if (frames == 2 && (!a.frame_scr || !a.frame_out || !a.lse2 || !mavlm_attention_frame_tiles_supported(a.R, a.S, a.H, a.frame_keys)))
    return hipErrorInvalidValue;
  const float c = a.scale * 1.44269504088896340736f;
  int tps = 0, ns = 1;
  attn3_sk_plan plan = {};
  if (a.split_ws != nullptr) {
    plan = attn3_plan(a.R, a.S, a.H);
    // the split-KV form of the small grids: single videos only, and never with the frame masses (one writer per entry)
    if (plan.wgs == 0 && nb == 1 && frames != 1) ns = mavlm_attention_splits(a.R, a.S, a.H, &tps);
  }
  const int skg = plan.wgs;
  int cut_units = 0;
  for (int j = 0; j < plan.nlev; ++j) cut_units += plan.nun[j];
  if (ns <= 1) { ns = 1; tps = 0; }
  float* opart = a.split_ws;
  float* lpart = nullptr;
  if (skg > 0) lpart = a.split_ws + (size_t)skg * plan.nlev * plan.qb * HD3;
  else if (ns > 1) lpart = a.split_ws + (size_t)ns * a.R * a.H * HD3;
  const int units = ((a.R + 127) / 128) * a.H;
  const dim3 grid(skg > 0 ? skg : units, skg > 0 ? 1 : ns);
  const bool w8 = skg > 0 && plan.qb == 256;
  const attn3_launch_args la = {grid, c, opart, lpart, tps, ns, cut_units, plan, a.H / nb,
                                frames == 1 ? a.S / a.frame_keys + ATTN3_FR_EXTRA : 0};
  const bool h16 = dtype == MAVLM_F16;
  if (frames == 2) {
    if (h16) return w8 ? attn3_launch<F16, 8, 2>(a, la, s) : attn3_launch<F16, 4, 2>(a, la, s);
    return w8 ? attn3_launch<BF16, 8, 2>(a, la, s) : attn3_launch<BF16, 4, 2>(a, la, s);
  }
  if (frames) {
    if (h16) return w8 ? attn3_launch<F16, 8, 1>(a, la, s) : attn3_launch<F16, 4, 1>(a, la, s);
    return w8 ? attn3_launch<BF16, 8, 1>(a, la, s) : attn3_launch<BF16, 4, 1>(a, la, s);
  }
  if (h16) return w8 ? attn3_launch<F16, 8, 0>(a, la, s) : attn3_launch<F16, 4, 0>(a, la, s);
  return w8 ? attn3_launch<BF16, 8, 0>(a, la, s) : attn3_launch<BF16, 4, 0>(a, la, s);
}

hipError_t mavlm_launch_attention3_frames(const mavlm_attn_args& a, int dtype, hipStream_t s) { return attn3_dispatch(a, dtype, 1, s); }
hipError_t mavlm_launch_attention3(const mavlm_attn_args& a, int dtype, hipStream_t s) { return attn3_dispatch(a, dtype, 0, s); }
// the tile-entry form (FR = 2): any schedule, incl. the split-KV form of the small grids
bool mavlm_attention_frame_tiles_supported(int R, int S, int H, int frame_keys) {
  return frame_keys >= KT3 && (frame_keys & 3) == 0 && R > 0 && S > 0 && S % frame_keys == 0 && S / frame_keys <= 64 &&
         (double)H * R * ((S + KT3 - 1) / KT3) * 8.0 < 2147483000.0;
}
size_t mavlm_attention_frame_tiles_scr_floats(int R, int S, int H) { return (size_t)H * R * ((S + KT3 - 1) / KT3) * 2; }
size_t mavlm_attention_frame_tiles_out_floats(int R, int S, int H, int frame_keys) { return (size_t)H * ((R + 63) / 64) * (S / frame_keys); }
hipError_t mavlm_launch_attention3_frame_tiles(const mavlm_attn_args& a, int dtype, hipStream_t s) { return attn3_dispatch(a, dtype, 2, s); }

// ------------------------------------------------------------------------------------------------------------
// attn_colsum3_kernel: column sums of the normalised probabilities (frame scores, MemoryController.py:135-139).
// A wave keeps 32 keys in registers (A operand) and streams the queries: 64-query Q tiles and their 64 lse2 values arrive
// by LDS-DMA into a 2-slot ring; S^T(qb) for the next 32-query block is computed (8 MFMAs) while exp2(s*c - lse2) of the
// previous block is accumulated per key in registers.
//   * Q fragments are read from LDS QPF steps ahead of their MFMA with hand-counted lgkmcnt waits (inline asm: the ISA of
//     the plain-C++ form waited `lgkmcnt(0)` for every fragment right after issuing its read).
//   * The DMAs are buffer loads (descriptor + 32-bit lane offset + scalar tile offset) issued between MFMA steps, not as
//     a burst (see attn_fwd3_kernel).
//   * Balanced schedule.  A unit = (128-key block, head) x all query tiles; (key blocks x heads) rarely matches the chip's
//     workgroup slots (392 units at S = 6272: whole units leave the CUs with 1 or 2 workgroups, two halves with 3 or 4 -
//     measured 820 TFLOP/s there against 904 at S = 6144 and 975 at S = 8192 where the slots fill evenly).  The grid is
//     therefore G workgroups over the FLATTENED (unit, query tile) space: workgroup g owns tiles [g*T/G, (g+1)*T/G) - every
//     workgroup the same count (+-1) - and walks the at most few units its range touches (segment loop).
//   * Deterministic output without atomics or a memset: the pieces of a unit are consecutive workgroups, piece j of a unit
//     stores its column sums into plane j of `part` ([planes, H, S]); piece 0 also zeroes the planes its unit does not use.
//     The consumer adds the planes in order (frame_scores_kernel / colsum_planes_reduce_kernel).
namespace {

constexpr int CS3_LSE = 2 * TILE3;             // 2 x 256 B of lse2 behind the two Q slots
constexpr int CS3_LDS = 2 * TILE3 + 512;
constexpr int CS3_QPF = 2;                     // LDS read-ahead of the Q fragments, in MFMA steps (4 VGPRs each; the
                                               // kernel must stay within 128 VGPRs for four workgroups per CU)

template <typename T>
__global__ __launch_bounds__(256, 4) void attn_colsum3_kernel(const uint16_t* __restrict__ Q, int ldq,
                                                              const uint16_t* __restrict__ K, int ldk,
                                                              const float* __restrict__ lse2, float* __restrict__ part,
                                                              int R, int S, int H, float c, int nkb, int nplanes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int ntq = (R + KT3 - 1) / KT3;
  // workgroup g owns tiles [b(g), b(g+1)) of the flattened (unit, query tile) space, b(g) = g*q + min(g, rem): the first
  // `rem` workgroups take q+1 tiles, the others q  (32-bit arithmetic; the launcher checks units * tiles < 2^31)
  const int ttot = nkb * H * ntq, G = (int)gridDim.x, gq = ttot / G, grem = ttot - gq * G;
  const int g = (int)blockIdx.x;
  int b0 = g * gq + (g < grem ? g : grem);
  const int b1 = b0 + gq + (g < grem ? 1 : 0);
  const size_t plane_stride = (size_t)H * S;
  auto owner = [&](int x) { return x < grem * (gq + 1) ? x / (gq + 1) : (x - grem) / gq; };   // largest g with b(g) <= x

  for (bool first_seg = true; b0 < b1; first_seg = false) {
  const int u = b0 / ntq;                                       // unit = (key block, head)
  const int tq0 = b0 - u * ntq;                                 // first query tile of this segment
  const int nt = (ntq - tq0 < b1 - b0) ? ntq - tq0 : b1 - b0;   // >= 1
  const int h = u % H, kb = u / H;
  const int g_first = owner(u * ntq), g_last = owner((u + 1) * ntq - 1);   // the workgroups that touch this unit
  const int plane = g - g_first;
  const int k0 = kb * 128 + wave * 32;
  if (!first_seg) {             // every wave is done with the LDS slots of the previous segment (and has no DMA in flight)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }

  typename T::vec8 kf[8];
  {
    int krow = k0 + r;
    krow = krow < S ? krow : S - 1;
    const uint16_t* kp = K + (size_t)krow * ldk + h * HD3 + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = *(const typename T::vec8*)(kp + 16 * ks);
  }

  // Q tile / lse2 DMA: same LDS image as the K tiles of attn_fwd3_kernel; rows past R read as zeros (masked below)
  auto rsrc_of = [&](const void* base, uint32_t bytes) {
    const uintptr_t a = (uintptr_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes),
                                             0x00020000);
  };
  const __amdgpu_buffer_rsrc_t qrs = rsrc_of(Q + h * HD3, (uint32_t)(R - 1) * (uint32_t)ldq * 2u + (uint32_t)HD3 * 2u);
  const __amdgpu_buffer_rsrc_t lrs = rsrc_of(lse2 + (size_t)h * R, (uint32_t)R * 4u);
  const int drow = 4 * wave + (lane >> 4);
  const int dch = (lane & 15) ^ (((lane >> 4) << 2) | wave);
  const int qo = (drow * ldq + dch * 8) * 2;                   // lane offset; tile and piece go into the scalar offset
  const unsigned lds0 = (unsigned)(uintptr_t)(MAVLM_LDS char*)smem;   // (this kernel is short of VGPRs, not of SGPRs)
  auto dma_piece = [&](int t, int slot_off, int i) {            // t: tile index inside this workgroup's range
    unsigned base = lds0 + wave * 1024;
    asm volatile("" : "+s"(base));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(qrs, (MAVLM_LDS void*)(uintptr_t)(base + slot_off + i * 4096), 16, qo,
                                             ((tq0 + t) * KT3 + 16 * i) * ldq * 2, 0, 0);
  };
  auto dma_lse = [&](int t, int slot) {                         // 64 lse2 values of the tile: 4 bytes per lane, wave 0
    if (wave == 0) {
      unsigned base = lds0;
      asm volatile("" : "+s"(base));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (MAVLM_LDS void*)(uintptr_t)(base + CS3_LSE + slot * 256), 4, lane * 4,
                                               (tq0 + t) * KT3 * 4, 0, 0);
    }
  };
  auto dma_q = [&](int t, int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_piece(t, slot * TILE3, i);
    dma_lse(t, slot);
  };

  const int xr = img3_x(r);
  unsigned qad[8];                                              // LDS byte address of fragment ks: + slot + block
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) qad[ks] = lds0 + 256 * r + 16 * ((2 * ks + hh) ^ xr);

  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  f32x16 st[2];

  // One phase = the 8 MFMAs of S^T for the 32-query block at LDS offset OFF into st[P^1], with the exp2 / accumulate of the
  // block in st[P] (2 elements per MFMA) in their shadow.  DSLOT >= 0: the five DMAs of tile `tdma` into slot DSLOT are
  // issued after MFMA steps 0, 2, 4, 6 (Q pieces) and 1 (lse2).  FIRST: no previous block to accumulate.
  auto phase = [&](auto pc, auto offc, auto dslotc, auto firstc, float l2, bool ok, int tdma) {
    constexpr int P = decltype(pc)::value, N = P ^ 1, OFF = decltype(offc)::value, DSLOT = decltype(dslotc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) st[N][i] = 0.f;
    const float l2m = ok ? l2 : INFINITY;                       // masked query: exp2(-inf) = 0
    u32x4 qf[8];
    auto rd = [&](auto ic) {
      constexpr int ks = decltype(ic)::value;
      const unsigned a = qad[ks];
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(OFF));
      qf[ks] = v;
    };
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    rd(IC<0>{});
    if constexpr (CS3_QPF > 1) rd(IC<1>{});
    if constexpr (CS3_QPF > 2) rd(IC<2>{});
    static_assert(CS3_QPF >= 1 && CS3_QPF <= 3, "read-ahead prologue / wait table are written for 1..3");
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](auto ic) {
      constexpr int ks = decltype(ic)::value;
      constexpr int ahead = (7 - ks) < CS3_QPF ? (7 - ks) : CS3_QPF;
      if constexpr (ks + CS3_QPF < 8) rd(IC<(ks + CS3_QPF < 8 ? ks + CS3_QPF : 7)>{});
      if constexpr (ahead == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else if constexpr (ahead == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      else if constexpr (ahead == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      st[N] = T::mfma32(kf[ks], __builtin_bit_cast(typename T::vec8, qf[ks]), st[N]);
      if constexpr (DSLOT >= 0) {
        if constexpr ((ks & 1) == 0) dma_piece(tdma, DSLOT * TILE3, ks >> 1);
        if constexpr (ks == 1) dma_lse(tdma, DSLOT);
      }
      if constexpr (!FIRST) {
        float x0 = st[P][2 * ks], x1 = st[P][2 * ks + 1];
        asm volatile("" : "+v"(x0), "+v"(x1));
        x0 = __builtin_amdgcn_exp2f(x0 * c - l2m);
        x1 = __builtin_amdgcn_exp2f(x1 * c - l2m);
        asm volatile("" : "+v"(x0), "+v"(x1));
        acc[2 * ks] += x0;
        acc[2 * ks + 1] += x1;
        asm volatile("" : "+v"(acc[2 * ks]), "+v"(acc[2 * ks + 1]));   // add here (else 16 exp results stay live to the end)
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    step(IC<0>{}); step(IC<1>{}); step(IC<2>{}); step(IC<3>{}); step(IC<4>{}); step(IC<5>{}); step(IC<6>{}); step(IC<7>{});
  };
  auto drain = [&](auto pc, float l2, bool ok) {                // last block: no MFMAs to hide behind
    constexpr int P = decltype(pc)::value;
    const float l2m = ok ? l2 : INFINITY;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += __builtin_amdgcn_exp2f(st[P][i] * c - l2m);
  };

  dma_q(0, 0);
  if (nt > 1) dma_q(1, 1);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(kf[ks]));
  if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile 0 landed; up to 4 DMAs of tile 1 stay in flight
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  phase(IC<1>{}, IC<0>{}, IC<-1>{}, IC<1>{}, 0.f, false, 0);    // block (0, 0) -> st[0]

  // tile t (index inside this workgroup's query range) in slot SLOT; its block 0 already sits in st[0]
  auto tile = [&](auto slotc, int t) {
    constexpr int SLOT = decltype(slotc)::value;
    const float* ls = (const float*)(smem + CS3_LSE + SLOT * 256);
    float l0 = ls[r], l1 = ls[32 + r];
    // no LDS read of this wave may be in flight when the hand-counted reads of the phases start
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l0), "+v"(l1)::"memory");
    const int q0 = (tq0 + t) * KT3 + r;
    const bool ok0 = q0 < R, ok1 = q0 + 32 < R;
    // block (t,0) accumulated under the MFMAs of block (t,1)
    phase(IC<0>{}, IC<SLOT * TILE3 + 8192>{}, IC<-1>{}, IC<0>{}, l0, ok0, 0);
    if (t + 1 < nt) {
      // tile t+1 must be visible before its block 0 is read: this wave's DMAs + barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      // every wave has passed the barrier, i.e. finished reading tile t (its block 1 fed the MFMAs above, its lse2
      // values are in registers): the slot is refilled (tile t+2) while block (t,1) is accumulated under the MFMAs of
      // block (t+1,0).  No DMA is issued that the loop would not wait for (a wave must not end with one in flight).
      if (t + 2 < nt) phase(IC<1>{}, IC<(SLOT ^ 1) * TILE3>{}, IC<SLOT>{}, IC<0>{}, l1, ok1, t + 2);
      else phase(IC<1>{}, IC<(SLOT ^ 1) * TILE3>{}, IC<-1>{}, IC<0>{}, l1, ok1, 0);
    } else {
      drain(IC<1>{}, l1, ok1);
    }
  };
  int t = 0;
  for (; t + 1 < nt; t += 2) {
    tile(IC<0>{}, t);
    tile(IC<1>{}, t + 1);
  }
  if (t < nt) tile(IC<0>{}, t);

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
      if (key < S) {
        float* o = part + (size_t)h * S + key;
        o[(size_t)plane * plane_stride] = acc[i];
        if (plane == 0)
          for (int pz = g_last - g_first + 1; pz < nplanes; ++pz) o[(size_t)pz * plane_stride] = 0.f;
      }
    }
  }
  b0 += nt;
  }  // segment loop
}

// part[0] += part[1] + ... in plane order (stand-alone operator: the result is the first H*S floats)
__global__ void colsum_planes_reduce_kernel(float* __restrict__ part, size_t n, int nplanes) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = part[i];
  for (int p = 1; p < nplanes; ++p) v += part[(size_t)p * n + i];
  part[i] = v;
}

}  // namespace

// Plan of the column-sum pass (pure function of the shape): G workgroups over the flattened (unit, query tile) space and
// the number of planes of `part` = the most pieces a unit can have.  G = 2 workgroups per CU measured best (longer query
// streams per workgroup, one prologue per ~150 tiles); never more workgroups than tiles.
int g_mavlm_colsum_wgs = 0;       // diagnostics: 0 = automatic
int mavlm_colsum_plan(int R, int S, int H, int* planes) {
  const long nkb = (S + 127) / 128, ntq = (R + KT3 - 1) / KT3;
  const long ttot = nkb * H * ntq;
  long G = g_mavlm_colsum_wgs > 0 ? g_mavlm_colsum_wgs : 512;       // (the setter admits 64 .. 1024)
  {   // short ranges only pay prologues: at least ~8 query tiles per workgroup, but never fewer workgroups than units
    const long units = nkb * H, floor_ = units < G ? units : G, by_len = ttot / 8;
    const long cap = by_len > floor_ ? by_len : floor_;
    if (G > cap) G = cap;
  }
  if (G > ttot) G = ttot;
  if (planes) {                                          // the most workgroups any unit's tiles are spread over (exact)
    const long q = ttot / G, rem = ttot - q * G;
    auto owner = [&](long x) { return x < rem * (q + 1) ? x / (q + 1) : (x - rem) / q; };     // as in the kernel
    long most = 1;
    for (long u = 0; u < nkb * H; ++u) {
      const long n = owner((u + 1) * ntq - 1) - owner(u * ntq) + 1;
      if (n > most) most = n;
    }
    *planes = (int)most;
  }
  return (int)G;
}
size_t mavlm_colsum_part_floats(int R, int S, int H) {      // buffer size: covers every setting of the diagnostics hooks
  const long nkb = (S + 127) / 128, ntq = (R + KT3 - 1) / KT3;
  const long ttot = nkb * H * ntq;
  const long minlen = ttot / (ttot < 1024 ? ttot : 1024);
  return (size_t)((ntq - 1 + minlen - 1) / minlen + 1) * H * S;
}
int mavlm_colsum_planes(int R, int S, int H) {
  int planes = 1;
  if (g_mavlm_attn_impl != 2) mavlm_colsum_plan(R, S, H, &planes);
  return planes;
}

hipError_t mavlm_launch_colsum3(const mavlm_colsum_args& a, int dtype, hipStream_t s) {
  if ((double)a.R * a.ldq * 2.0 >= 2147483648.0) return hipErrorInvalidValue;      // 32-bit buffer offsets
  const float c = a.scale * 1.44269504088896340736f;
  const int nkb = (a.S + 127) / 128;
  if ((double)nkb * a.H * ((a.R + KT3 - 1) / KT3) >= 2.0e9) return hipErrorInvalidValue;   // 32-bit schedule arithmetic
  int planes = 1;
  const int G = mavlm_colsum_plan(a.R, a.S, a.H, &planes);
  {
    mavlm_prof_scope prof(MAVLM_K_COLSUM, 2.0 * a.R * (double)a.S * a.H * HD3, 2.0 * HD3 * a.H * ((double)a.R + a.S), s);
    if (dtype == MAVLM_F16)
      hipLaunchKernelGGL(attn_colsum3_kernel<F16>, dim3(G), dim3(256), CS3_LDS, s, (const uint16_t*)a.Q, a.ldq,
                         (const uint16_t*)a.K, a.ldk, a.lse2, a.part, a.R, a.S, a.H, c, nkb, planes);
    else
      hipLaunchKernelGGL(attn_colsum3_kernel<BF16>, dim3(G), dim3(256), CS3_LDS, s, (const uint16_t*)a.Q, a.ldq,
                         (const uint16_t*)a.K, a.ldk, a.lse2, a.part, a.R, a.S, a.H, c, nkb, planes);
  }
  if (!a.keep_planes && planes > 1) {
    const size_t n = (size_t)a.H * a.S;
    mavlm_prof_scope prof(MAVLM_K_MISC, 0.0, 4.0 * (planes + 1) * (double)n, s);
    hipLaunchKernelGGL(colsum_planes_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.part, n, planes);
  }
  return hipGetLastError();
}
