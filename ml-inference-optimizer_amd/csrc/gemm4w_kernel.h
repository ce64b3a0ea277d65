// 256x256 bf16/fp16 GEMM, 4 waves per workgroup, ONE wave per SIMD with the whole 512-register file
// (gfx950: VGPR + AGPR unified) -- the main-path GEMM of the hot path.
//
// Same math/epilogue as gemm_bias_act_kernel (gemm_kernel.h).  Structure:
//   * waves 2 (M) x 2 (N); wave (wr, wc) owns a 128 x 128 output tile = 4 x 4 MFMA tiles of 32x32
//     (256 accumulator registers).  The bigger wave tile cuts LDS fragment traffic per MFMA by a third
//     against 128x64 wave tiles, and with one wave per SIMD nothing competes for the matrix pipe: the wave
//     keeps it busy by itself, software-pipelined (2 MFMAs, 1 fragment read, 1 prefetch piece per micro-step).
//   * A 256x256 tile needs 64 KiB of operands per 64 k, i.e. ~32 B/clk/CU at full MFMA rate, and the fill
//     path of one CU is latency x concurrency bound (measured ~35 GB/s with one K-tile in flight, i.e. half
//     the need).  So the K-tile is only 32 deep and the LDS holds FOUR of them (4 x 32 KiB = 128 KiB):
//     three K-tiles (96 KiB) are in flight behind the one being multiplied, loads are never drained
//     (counted s_waitcnt vmcnt(16)), and a K-tile has two full K-tile times to arrive.
//   * direct-to-LDS loads (global_load_lds_dwordx4): a 1-KiB piece = 16 rows x 64 B; the bank swizzle
//     (16-B chunk c of row r stored at c ^ ((r>>2)&3)) is applied to the per-lane SOURCE address.
//   * ONE barrier per K-tile, placed before the LAST k-step's MFMAs: by then every wave has its last
//     fragments of the current stage in registers (WAR for the refill) and has waited for its pieces of the
//     next K-tile (RAW); the first fragments of the next K-tile are read right after the barrier and their
//     latency hides under the 16 MFMAs still queued.
#pragma once
#include <type_traits>

#include "gemm_kernel.h"

constexpr int G4_BK = 32;
constexpr int G4_STAGES = 4;
constexpr int G4_XT = 256 * G4_BK * 2;       // 16 KiB: one operand tile (256 rows x 64 B)
constexpr int G4_BUF = 2 * G4_XT;            // X + W
constexpr int G4_SMEM = G4_STAGES * G4_BUF;  // 128 KiB

// VAR: timing-only ablation bits (0 in the shipped dispatch): 4 = no prefetch issue in the loop (wrong results),
// 8 = no fragment reads in the loop (wrong results).
template <typename T, int ACT, int VAR = 0>
__global__ __launch_bounds__(256) void gemm4w_kernel(const GemmDev p) {
  using X8 = typename DT<T>::x8;
  static_assert(ACT != MIO_ACT_SWIGLU, "dual-B GEMM uses gemm_bias_act_kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  int tm, tn;
  gemm_tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256;
  const int n0 = tn * 256;
  const int nk = (p.K + G4_BK - 1) / G4_BK;

  // ---- prefetch addressing: an operand tile is 16 pieces of 16 rows x 64 B; wave w owns pieces 4w..4w+3 of X and W.
  const int prow = lane >> 2, pcs = lane & 3;
  int64_t xoff[4], woff[4];
  int kch[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 16 + prow;  // 0..255
    kch[i] = pcs ^ ((row >> 2) & 3);
    int64_t gm = m0 + row;
    if (gm > p.M - 1) gm = p.M - 1;
    xoff[i] = gm * p.ldx + 8 * kch[i];
    int gn = n0 + row;
    if (gn > p.N - 1) gn = p.N - 1;
    woff[i] = (int64_t)gn * p.ldw + 8 * kch[i];
  }
  const T* xg = (const T*)p.x;
  const T* wg = (const T*)p.w;
  const bool ktail = (p.K % G4_BK) != 0;

  // issue ONE 1-KiB piece (piece i of X if which == 0, of W if which == 1) of K-tile kt into stage kt % 4.
  // Branch-free: past the last K-tile the last one is simply re-fetched into an idle stage.
  auto issue_one = [&](int kt, auto I, auto WHICH) {
    constexpr int i = decltype(I)::value, which = decltype(WHICH)::value;
    const int kte = kt < nk ? kt : nk - 1;
    char* dst = smem + (kt & (G4_STAGES - 1)) * G4_BUF + which * G4_XT + (wave * 4 + i) * 1024;
    const int k0 = kte * G4_BK;
    const T* src = (which ? wg + woff[i] : xg + xoff[i]) + k0;
    if (ktail && k0 + 8 * kch[i] >= p.K) src = (const T*)mio_zero16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (MIO_LDS void*)dst, 16, 0, 0);
  };

  f32x16_t acc[4][4];  // [nt][mt]
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;

  // fragment read offsets (bytes inside a K-tile buffer); rows are base + r with base % 32 == 0
  const int sw = (r >> 2) & 3;
  const int xbase = (wr * 128 + r) * 64;
  const int wbase = G4_XT + (wc * 128 + r) * 64;

  X8 fx[2][4], fw[2][4];  // [register buffer][tile]
  // fragment #idx of a k-step, in the order the MFMAs (nt-major) first need them: w0 x0 x1 x2 x3 w1 w2 w3
  auto read_one = [&](const char* buf, int ks, auto RB, auto IDX) {
    constexpr int rb = decltype(RB)::value, idx = decltype(IDX)::value;
    const int co = ((2 * ks + h) ^ sw) * 16;
    if constexpr (idx == 0) fw[rb][0] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + wbase + co));
    else if constexpr (idx <= 4) fx[rb][idx - 1] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + xbase + (idx - 1) * 32 * 64 + co));
    else fw[rb][idx - 4] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + wbase + (idx - 4) * 32 * 64 + co));
  };
  auto mfma_pair = [&](auto RB, auto J) {  // MFMAs 2j, 2j+1 of the k-step (nt-major order)
    constexpr int rb = decltype(RB)::value, j = decltype(J)::value;
    constexpr int nt = j / 2, mt0 = (j % 2) * 2;
    acc[nt][mt0] = DT<T>::mfma32(fw[rb][nt], fx[rb][mt0], acc[nt][mt0]);
    acc[nt][mt0 + 1] = DT<T>::mfma32(fw[rb][nt], fx[rb][mt0 + 1], acc[nt][mt0 + 1]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // One k-step = 8 micro-steps of {2 MFMAs, 1 fragment read for the next k-step, optionally 1 prefetch piece}.
  // The order is pinned with sched_barrier(0): hipcc otherwise clusters the loads between MFMA bursts and the
  // matrix pipe idles (one wave per SIMD: nobody else fills it).
  auto kstep = [&](auto RBv, const char* rbuf, int rks, int pf_kt, auto PF_I0, auto PFv) {
    using RB = decltype(RBv);
    constexpr int rb = RB::value;
    constexpr int pf_i0 = decltype(PF_I0)::value;
    constexpr bool pf = decltype(PFv)::value;
    using NRB = std::integral_constant<int, rb ^ 1>;
#define G4_MICRO(J)                                                                      \
    mfma_pair(RB{}, std::integral_constant<int, J>{});                                   \
    if constexpr (!(VAR & 8)) read_one(rbuf, rks, NRB{}, std::integral_constant<int, J>{}); \
    if constexpr (pf && !(VAR & 4))                                                      \
      issue_one(pf_kt, std::integral_constant<int, pf_i0 + J / 2>{}, std::integral_constant<int, J % 2>{}); \
    __builtin_amdgcn_sched_barrier(0);
    G4_MICRO(0) G4_MICRO(1) G4_MICRO(2) G4_MICRO(3) G4_MICRO(4) G4_MICRO(5) G4_MICRO(6) G4_MICRO(7)
#undef G4_MICRO
  };

  // ---- prologue: K-tiles 0, 1, 2 in flight (stages 0..2); K-tile 0 must have landed
#define G4_ISSUE_TILE(KT)                                                                              \
  issue_one(KT, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});                   \
  issue_one(KT, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});                   \
  issue_one(KT, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});                   \
  issue_one(KT, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
  G4_ISSUE_TILE(0)
  G4_ISSUE_TILE(1)
  G4_ISSUE_TILE(2)
#undef G4_ISSUE_TILE
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    const int co = (h ^ sw) * 16;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      fx[0][t] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + xbase + t * 32 * 64 + co));
      fw[0][t] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + wbase + t * 32 * 64 + co));
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  using TRUE_ = std::true_type;
  using FALSE_ = std::false_type;
  for (int kt = 0; kt < nk; ++kt) {
    const char* buf = smem + (kt & (G4_STAGES - 1)) * G4_BUF;
    const char* nbuf = smem + ((kt + 1) & (G4_STAGES - 1)) * G4_BUF;
    // k-step 0 (register buffer 0): read the fragments of k-step 1 and issue the 8 prefetch pieces of K-tile
    // kt+3 into the stage that K-tile kt-1 occupied (released by the barrier of the previous iteration)
    kstep(I0{}, buf, 1, kt + 3, I0{}, TRUE_{});
    // this wave's last fragments of `buf` are in registers (lgkmcnt) and its pieces of K-tile kt+1 have landed
    // (all but the 16 newest loads = K-tiles kt+2, kt+3); after the barrier that holds for every wave
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1 (register buffer 1): read the first fragments of K-tile kt+1
    kstep(I1{}, nbuf, 0, 0, I0{}, FALSE_{});
  }

  f32x16_t dummy[1][1];
  gemm_epilogue<T, ACT, 4, 4>(p, acc, dummy, m0 + wr * 128, n0 + wc * 128, r, h);
}
