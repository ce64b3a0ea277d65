// 256x256x64 "8-phase" bf16/fp16 GEMM for CDNA4 (gfx950): the main-path GEMM of the hot path
// (q/k/v/o projections and both FusedMLP stages at M >= a few thousand rows).
//
// Same math/epilogue as gemm_bias_act_kernel (gemm_kernel.h); different pipeline:
//   * 8 waves = 2 (M) x 4 (N); wave (wr, wc) owns output rows wr*128..+127, columns wc*64..+63
//     (4 x 2 tiles of 32x32, 128 accumulator registers).
//   * LDS = 2 K-tile buffers x {XA, WA, WB, XB} half-tiles of 128 rows x 64 k (16 KiB each) = 128 KiB.
//     Half-tiles are grouped by WHEN they are consumed, not by row range: XA / XB = the rows of the first /
//     second 64-row m-half of both wave rows, WA / WB = the rows of the first / second 32-column n-half of
//     all four wave columns; phase 0 reads XA+WA, phase 1 WB, phase 2 XB, phase 3 WA again.
//   * Each K-tile is cut into 4 phases; a phase computes one 64x32 quadrant of the wave tile over the
//     whole BK = 64 (8 MFMAs) and issues ONE half-tile of direct-to-LDS prefetch (2 x 1 KiB per wave)
//     for the NEXT K-tile, in consumption order.  Loads stay in flight across barriers: each phase ends its
//     load segment with a counted `s_waitcnt vmcnt(4)` (two half-tiles stay in flight; never 0 in steady
//     state), which retires the half-tile issued two phases earlier -- the one the next phase reads.
//     Barriers are raw s_barrier (a __syncthreads() would drain the DMA queue).
//   * The two waves that share a SIMD (w and w+4) run staggered by one barrier: while one is in its
//     MFMA segment the other is in its LDS-read / prefetch-issue segment, so the matrix pipe alternates
//     between them instead of both stalling on the same barrier.
//
// Ordering contract (placement-independent, see the in-loop comments):
//   RAW  a half-tile is read only after (a) every wave's counted vmcnt that covers its pieces of that
//        half-tile and (b) at least one barrier after the latest such wait.
//   WAR  a half-tile slot is re-filled only after every wave that reads it has passed the lgkmcnt(0)
//        that retires those reads plus one barrier.
#pragma once
#include <type_traits>

#include "gemm_kernel.h"

constexpr int G8_HALF = 128 * GEMM_BK * 2;   // 16 KiB: 128 rows x 64 k x 2 B
constexpr int G8_BUF = 4 * G8_HALF;          // X0, X1, W0, W1
constexpr int G8_SMEM = 2 * G8_BUF;          // 128 KiB

// VAR: timing-only ablation bits for tuning (0 in the shipped dispatch): 1 = no stagger, 2 = no s_setprio,
// 4 = no prefetch issue (wrong results), 8 = fragment reads only in K-tile 0 (wrong results).
template <typename T, int ACT, int VAR = 0>
__global__ __launch_bounds__(512) void gemm8p_kernel(const GemmDev p) {
  using X8 = typename DT<T>::x8;
  static_assert(ACT != MIO_ACT_SWIGLU, "dual-B GEMM uses gemm_bias_act_kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r = lane & 31, h = lane >> 5;

  int tm, tn;
  gemm_tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256;
  const int n0 = tn * 256;
  const int nk = (p.K + GEMM_BK - 1) / GEMM_BK;

  // ---- prefetch addressing: a half-tile is 16 pieces of 8 rows; wave w owns pieces 2w, 2w+1.
  // lane -> (row in piece, stored chunk); the swizzle c ^ ((row>>1)&7) is applied to the SOURCE chunk.
  const int prow = lane >> 3, pcs = lane & 7;
  int64_t goff[4][2];  // element offset of this lane's 16-byte chunk at k0 = 0, per half-tile / piece
  int kchunk[2];       // source chunk index (0..7) per piece (same for every half-tile)
  // half-tile ids: 0 = XA, 1 = WA, 2 = WB, 3 = XB (issue / consumption order)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int lrow = (wave * 2 + i) * 8 + prow;  // 0..127 inside the half-tile
    kchunk[i] = pcs ^ ((lrow >> 1) & 7);
    const int xr = (lrow >> 6) * 128 + (lrow & 63);  // tile row of XA's local row (XB: + 64)
    const int wrw = (lrow >> 5) * 64 + (lrow & 31);  // tile row of WA's local row (WB: + 32)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int64_t gm = m0 + xr + j * 64;
      if (gm > p.M - 1) gm = p.M - 1;
      goff[j ? 3 : 0][i] = gm * p.ldx + 8 * kchunk[i];
      int gn = n0 + wrw + j * 32;
      if (gn > p.N - 1) gn = p.N - 1;
      goff[1 + j][i] = (int64_t)gn * p.ldw + 8 * kchunk[i];
    }
  }
  const T* xg = (const T*)p.x;
  const T* wg = (const T*)p.w;

  // issue half-tile j (0 = XA, 1 = WA, 2 = WB, 3 = XB) of K-tile kt; no-op past the last K-tile
  auto issue = [&](int kt, auto J) {
    constexpr int j = decltype(J)::value;
    if ((VAR & 4) ? (kt < 1) : (kt < nk)) {
      char* dst = smem + (kt & 1) * G8_BUF + j * G8_HALF;
      const int k0 = kt * GEMM_BK;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const T* src = ((j == 0 || j == 3) ? xg : wg) + goff[j][i] + k0;
        if (k0 + 8 * kchunk[i] >= p.K) src = (const T*)mio_zero16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (MIO_LDS void*)(dst + (wave * 2 + i) * 1024), 16, 0, 0);
      }
    }
  };

  f32x16_t acc[2][4];  // [nt][mt]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

  // ---- fragment read offsets inside a K-tile buffer (bytes).  Rows are base + r with base % 32 == 0,
  // so the swizzle term (row>>1)&7 == (r>>1)&7 is a per-lane constant.
  const int sw = (r >> 1) & 7;
  int coff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) coff[ks] = ((2 * ks + h) ^ sw) * 16;
  const int xrow_base = (wr * 64 + r) * 128;  // local row wr*64 + mt*32 + r of XA (slot 0) / XB (slot 3)
  const int wrow_base = (wc * 32 + r) * 128;  // local row wc*32 + r of WA (slot 1) / WB (slot 2)

  X8 xf[2][4];  // [mt within the m-half][ks]
  X8 wf[4];     // [ks]

  // ---- prologue: the four half-tiles of K-tile 0; XA + WA must have landed before phase 0
  issue(0, std::integral_constant<int, 0>{});
  issue(0, std::integral_constant<int, 1>{});
  issue(0, std::integral_constant<int, 2>{});
  issue(0, std::integral_constant<int, 3>{});
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (!(VAR & 1) && wr == 1) __builtin_amdgcn_s_barrier();  // stagger: waves 4-7 run one barrier behind waves 0-3

  // one phase = [LDS fragment reads + one half-tile of prefetch] | barrier | [8 MFMAs] | barrier
  auto phase = [&](int kt, auto P) {
    constexpr int ph = decltype(P)::value;
    const char* buf = smem + (kt & 1) * G8_BUF;
    const bool do_reads = !(VAR & 8) || kt == 0;
    if (do_reads) {
      if constexpr (ph == 0 || ph == 2) {  // X rows of m-half ph/2: slot XA (0) or XB (3)
        constexpr int slot = (ph == 0) ? 0 : 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            xf[mt][ks] = __builtin_bit_cast(
                X8, *(const u32x4_t*)(buf + slot * G8_HALF + xrow_base + mt * 32 * 128 + coff[ks]));
      }
      if constexpr (ph != 2) {  // W rows: phases 0,3 -> WA (slot 1); phase 1 -> WB (slot 2, kept through phase 2)
        constexpr int slot = (ph == 1) ? 2 : 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          wf[ks] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + slot * G8_HALF + wrow_base + coff[ks]));
      }
    }
    // prefetch: phase p issues half-tile p of K-tile kt+1 (XA, WA, WB, XB = consumption order) into the
    // other buffer.  WAR: that slot was last read during K-tile kt-1, in phase 0 / 3 / 1 / 2 respectively;
    // the closest case (WA: read in phase 3 of kt-1, re-filled in phase 1 of kt) is four barriers apart.
    issue(kt + 1, P);
    // RAW: retire the half-tile issued two phases ago (it is read in the NEXT phase, after this phase's two
    // barriers; the trailing wave group's wait falls one barrier later, still before that read).  Two
    // half-tiles (4 loads) stay in flight.  In the last K-tile nothing new is issued: drain instead.
    if (kt + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & 2)) __builtin_amdgcn_s_setprio(1);
    constexpr int nt = (ph == 0 || ph == 3) ? 0 : 1;
    constexpr int mh = ph / 2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      acc[nt][mh * 2 + 0] = DT<T>::mfma32(wf[ks], xf[0][ks], acc[nt][mh * 2 + 0]);
      acc[nt][mh * 2 + 1] = DT<T>::mfma32(wf[ks], xf[1][ks], acc[nt][mh * 2 + 1]);
    }
    if (!(VAR & 2)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  for (int kt = 0; kt < nk; ++kt) {
    phase(kt, std::integral_constant<int, 0>{});
    phase(kt, std::integral_constant<int, 1>{});
    phase(kt, std::integral_constant<int, 2>{});
    phase(kt, std::integral_constant<int, 3>{});
  }
  if (!(VAR & 1) && wr == 0) __builtin_amdgcn_s_barrier();  // match the stagger barrier of waves 4-7

  f32x16_t dummy[1][1];
  gemm_epilogue<T, ACT, 2, 4>(p, acc, dummy, m0 + wr * 128, n0 + wc * 64, r, h);
}
