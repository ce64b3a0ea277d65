// 256x256 bf16/fp16 GEMM on v_mfma_f32_16x16x32, 4 waves per workgroup, ONE wave per SIMD with the whole
// 512-register file (gfx950: VGPR + AGPR unified) -- the main-path GEMM of the hot path.
//
// Why this shape: on random operands MI355X lowers its clock under matrix load, and how far depends on the
// MFMA shape.  Measured on this chip with bare register-operand loops (tools/micro/mfma_peak.hip), one wave
// per SIMD: v_mfma_f32_32x32x16_bf16 sustains 1.53 PFLOP/s, v_mfma_f32_16x16x32_bf16 1.88 PFLOP/s (zeros: 2.4
// for both).  The 16x16x32 form is worth ~1.2x before anything else.
//
// Same math/epilogue as gemm_bias_act_kernel (gemm_kernel.h).  Structure:
//   * waves 2 (M) x 2 (N); wave (wr, wc) owns a 128 x 128 output tile = 8 x 8 MFMA tiles of 16x16
//     (256 accumulator registers); K-tile = 32 = one MFMA k-step = 64 MFMAs per wave.
//   * The weight tile is the A operand and the activation tile the B operand, so a lane of the accumulator
//     holds ONE output row (m = lane&15) and 4 consecutive output columns: bias / activation / residual are
//     lane-local and the stores are 8-byte row segments.
//   * LDS holds FOUR K-tiles (4 x (X[256][32] + W[256][32]) = 128 KiB): a 256x256 tile needs ~32 B/clk/CU of
//     operands at full rate and one CU's fill path is latency x concurrency bound, so three K-tiles (96 KiB) stay
//     in flight behind the one being multiplied; loads are never drained (counted s_waitcnt vmcnt(16)).
//   * direct-to-LDS loads (buffer_load_dwordx4 ... lds): a 1-KiB piece = 16 rows x 64 B.  Bank swizzle for the
//     16x16x32 fragment read (lane -> row l&15, 16-byte chunk l>>4): chunk c of row r is stored at
//     c ^ f((r>>2)&3), f = {0,2,3,1}, applied to the per-lane SOURCE address (LDS-DMA writes lane-linear).
//   * ONE barrier per K-tile, in the middle of its 64 MFMAs: first half = MFMAs + the 8 prefetch pieces of
//     K-tile kt+3; then wait (counted) for K-tile kt+1 + barrier; second half = MFMAs + the 16 fragment reads of
//     K-tile kt+1 into the other register buffer.  Micro-step order is pinned with sched_barrier(0) (hipcc
//     otherwise clusters the loads between MFMA bursts and the only wave on the SIMD leaves the pipe idle).
#pragma once
#include <type_traits>

#include "gemm_kernel.h"
#include "gemm4w16_acc.inc"

constexpr int G6_BK = 32;
constexpr int G6_STAGES = 4;
constexpr int G6_XT = 256 * G6_BK * 2;       // 16 KiB: one operand tile (256 rows x 64 B)
constexpr int G6_BUF = 2 * G6_XT;            // X + W
constexpr int G6_SMEM = G6_STAGES * G6_BUF;  // 128 KiB

__device__ __forceinline__ int g6_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// Epilogue for 16x16 accumulator tiles: lane (c = lane&15, g = lane>>4) holds output row mbase + mt*16 + c and
// columns nbase + nt*16 + 4g + (0..3) of tile (nt, mt) -- 8 bytes.  Two row-adjacent tiles A = (nt, mt) and
// B = (nt, mt+1) are exchanged with v_permlane16_swap so that every lane ends up with 16 contiguous bytes of ONE
// row (even g: tile A's row, odd g: tile B's row; columns 8*(g>>1)..+7): half the store instructions for the
// same bytes -- the store tail of an MFMA epilogue is issue-bound, not bandwidth-bound.
template <typename T, int ACT>
__device__ __forceinline__ void gemm_epilogue16(const GemmDev& p, int64_t mbase, int nbase, int c16, int g) {
  using X4 = typename DT<T>::x4;
  int ncol[8];
  X4 bv[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const int n = nbase + nt * 16 + 4 * g;
    ncol[nt] = (n < p.N) ? n : (p.N - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[nt][e] = (T)0.f;
  }
  if (p.bias) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) bv[nt] = *(const X4*)((const T*)p.bias + ncol[nt]);
  }
  auto row_pair = [&](auto MTP) {
    constexpr int mt = 2 * decltype(MTP)::value;
    const int64_t mA = mbase + mt * 16 + c16, mB = mA + 16;
    const int64_t mAc = mA < p.M ? mA : (p.M - 1), mBc = mB < p.M ? mB : (p.M - 1);
    X4 rA[8], rB[8];
    if (p.res) {
      const T* ra = (const T*)p.res + mAc * p.ldr;
      const T* rb = (const T*)p.res + mBc * p.ldr;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        rA[nt] = *(const X4*)(ra + ncol[nt]);
        rB[nt] = *(const X4*)(rb + ncol[nt]);
      }
    }
    const int64_t mrow = (g & 1) ? mB : mA;  // the row this lane stores after the exchange
    const bool mok = mrow < p.M;
    T* yrow = (T*)p.y + (mok ? mrow : 0) * p.ldy;
    auto col_block = [&](auto NT) {
      constexpr int nt = decltype(NT)::value;
      const f32x4_t a = G6AccIO<nt * 8 + mt>::read();
      const f32x4_t b = G6AccIO<nt * 8 + mt + 1>::read();
      float va[4], vb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        va[e] = gemm_act<ACT>(a[e] + (float)bv[nt][e]);
        vb[e] = gemm_act<ACT>(b[e] + (float)bv[nt][e]);
      }
      if (p.res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] += (float)rA[nt][e];
          vb[e] += (float)rB[nt][e];
        }
      }
      const auto s0 = __builtin_amdgcn_permlane16_swap(pack2<T>(va[0], va[1]), pack2<T>(vb[0], vb[1]), false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pack2<T>(va[2], va[3]), pack2<T>(vb[2], vb[3]), false, false);
      const int n = nbase + nt * 16 + 8 * (g >> 1);
      if (mok && n < p.N) {
        u32x4_t o = {s0[0], s1[0], s0[1], s1[1]};
        *(u32x4_t*)(yrow + n) = o;
      }
    };
    col_block(std::integral_constant<int, 0>{}); col_block(std::integral_constant<int, 1>{});
    col_block(std::integral_constant<int, 2>{}); col_block(std::integral_constant<int, 3>{});
    col_block(std::integral_constant<int, 4>{}); col_block(std::integral_constant<int, 5>{});
    col_block(std::integral_constant<int, 6>{}); col_block(std::integral_constant<int, 7>{});
  };
  row_pair(std::integral_constant<int, 0>{}); row_pair(std::integral_constant<int, 1>{});
  row_pair(std::integral_constant<int, 2>{}); row_pair(std::integral_constant<int, 3>{});
}

// VAR (timing-only ablations, 0 in the shipped dispatch): 4 = no prefetch issue in the loop, 8 = no fragment
// reads in the loop, 16 = no per-K-tile barrier (all: wrong results), 32 = in-kernel stamps (diagnostic build)
template <typename T, int ACT, int VAR = 0>
__global__ __launch_bounds__(256) void gemm4w16_kernel(const GemmDev p) {
  using X8 = typename DT<T>::x8;
  static_assert(ACT != MIO_ACT_SWIGLU, "dual-B GEMM uses gemm_bias_act_kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int c16 = lane & 15, g = lane >> 4;

  int tm, tn;
  gemm_tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * 256;
  const int n0 = tn * 256;
  const int nk = (p.K + G6_BK - 1) / G6_BK;

  // ---- prefetch addressing: an operand tile is 16 pieces of 16 rows x 64 B; wave w owns pieces 4w..4w+3 of X and W.
  // Loads are buffer_load ... lds with a per-workgroup resource (base = first row of this tile), a per-lane
  // 32-bit byte offset fixed for the whole kernel (clamped row * ld + swizzled chunk) and the K offset in an
  // SGPR: no vector ALU work per load (one wave per SIMD: every VALU instruction between two MFMAs delays the
  // second one -- measured 31 cycles per load with 64-bit per-lane addresses).  Requires K % 32 == 0.
  const int prow = lane >> 2, pcs = lane & 3;
  int xvo[4], wvo[4];
  const int64_t mrem = p.M - m0;          // rows of x left from this tile's first row (>= 1)
  const int nrem = p.N - n0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 16 + prow;  // 0..255
    const int kch = pcs ^ g6_swz(row);
    const int xr = (row < mrem) ? row : (int)(mrem - 1);
    const int wrw = (row < nrem) ? row : (nrem - 1);
    xvo[i] = (p.x_blk ? xr * 64 : xr * (int)p.ldx * 2) + 16 * kch;
    wvo[i] = (p.w_blk ? wrw * 64 : wrw * (int)p.ldw * 2) + 16 * kch;
  }
  const int xkstep = p.x_blk ? 16384 : G6_BK * 2;  // bytes from one K-tile of x / w to the next
  const int wkstep = p.w_blk ? 16384 : G6_BK * 2;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      p.x_blk ? (void*)((const char*)p.x + ((int64_t)tm * nk << 14)) : (void*)((const T*)p.x + m0 * p.ldx), 0, 0x7fffffff,
      0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      p.w_blk ? (void*)((const char*)p.w + ((int64_t)tn * nk << 14)) : (void*)((const T*)p.w + (int64_t)n0 * p.ldw), 0,
      0x7fffffff, 0x00020000);

  // issue ONE 1-KiB piece (piece i of X if which == 0, of W if which == 1) of K-tile kt into stage kt % 4.
  // Branch-free: past the last K-tile the last one is simply re-fetched into an idle stage.
  auto issue_one = [&](int kt, auto I, auto WHICH) {
    constexpr int i = decltype(I)::value, which = decltype(WHICH)::value;
    const int kte = kt < nk ? kt : nk - 1;
    char* dst = smem + (kt & (G6_STAGES - 1)) * G6_BUF + which * G6_XT + (wave * 4 + i) * 1024;
    const int koff = kte * (which ? wkstep : xkstep);  // bytes
    __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? wrs : xrs, (MIO_LDS void*)dst, 16, which ? wvo[i] : xvo[i], koff,
                                             0, 0);
  };

  // 64 accumulator tiles of 4 registers live in the accumulator file a[0:255], owned by inline asm (tile
  // (nt, mt) = a[4*(8*nt+mt) ..+3]): left to hipcc, 64 independent 4-register accumulators get shuffled
  // between AGPRs and VGPRs inside the loop (4 v_accvgpr moves per MFMA).
  // (the first K-tile starts them with C = 0: no zeroing pass)

  // fragment read offsets: lane reads row base + c16, stored chunk g ^ f(row)
  const int co = (g ^ g6_swz(c16)) * 16;
  const int xbase = (wr * 128 + c16) * 64 + co;
  const int wbase = G6_XT + (wc * 128 + c16) * 64 + co;

  // Activation fragments (B operand, shared by all 8 weight tiles) are double-buffered per K-tile; weight
  // fragments (A operand, each used by 8 consecutive MFMAs) are read just in time, two micro-steps ahead.
  X8 fx[2][8];  // [register buffer][mt]
  X8 fw[2];     // rotating pair, indexed by nt & 1
  auto read_x = [&](const char* buf, auto RB, auto MT) {
    constexpr int rb = decltype(RB)::value, mt = decltype(MT)::value;
    fx[rb][mt] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + xbase + mt * 16 * 64));
  };
  auto read_w = [&](const char* buf, auto NT) {
    constexpr int nt = decltype(NT)::value;
    fw[nt & 1] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + wbase + nt * 16 * 64));
  };
  auto mfma4 = [&](auto RB, auto J, auto FIRST_) {  // MFMAs 4j .. 4j+3 of the K-tile (nt-major): nt = j/2, mt = (j%2)*4 ..
    constexpr int rb = decltype(RB)::value, j = decltype(J)::value;
    constexpr int nt = j / 2, mt0 = (j % 2) * 4;
    if constexpr (decltype(FIRST_)::value != 0) {
      G6Acc<T, nt * 8 + mt0 + 0>::mfma0(fw[nt & 1], fx[rb][mt0 + 0]);
      G6Acc<T, nt * 8 + mt0 + 1>::mfma0(fw[nt & 1], fx[rb][mt0 + 1]);
      G6Acc<T, nt * 8 + mt0 + 2>::mfma0(fw[nt & 1], fx[rb][mt0 + 2]);
      G6Acc<T, nt * 8 + mt0 + 3>::mfma0(fw[nt & 1], fx[rb][mt0 + 3]);
    } else {
      G6Acc<T, nt * 8 + mt0 + 0>::mfma(fw[nt & 1], fx[rb][mt0 + 0]);
      G6Acc<T, nt * 8 + mt0 + 1>::mfma(fw[nt & 1], fx[rb][mt0 + 1]);
      G6Acc<T, nt * 8 + mt0 + 2>::mfma(fw[nt & 1], fx[rb][mt0 + 2]);
      G6Acc<T, nt * 8 + mt0 + 3>::mfma(fw[nt & 1], fx[rb][mt0 + 3]);
    }
  };

  unsigned long long t_begin = 0, t_loop0 = 0, t_loop1 = 0, r_begin = 0;
  if constexpr (VAR & 32) {
    t_begin = __builtin_amdgcn_s_memtime();
    r_begin = __builtin_amdgcn_s_memrealtime();
  }
  // ---- prologue: K-tiles 0, 1, 2 in flight (stages 0..2); K-tile 0 must have landed
#define G6_ISSUE_TILE(KT)                                                                              \
  issue_one(KT, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});                   \
  issue_one(KT, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});                   \
  issue_one(KT, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{});                   \
  issue_one(KT, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});                   \
  issue_one(KT, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
  G6_ISSUE_TILE(0)
  G6_ISSUE_TILE(1)
  G6_ISSUE_TILE(2)
#undef G6_ISSUE_TILE
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < 8; ++t) fx[0][t] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + xbase + t * 16 * 64));
  fw[0] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + wbase));
  __builtin_amdgcn_sched_barrier(0);

  // One K-tile (64 MFMAs = 16 micro-steps of 4) on activation register buffer RB.
  //   micro-steps 0..3 : MFMAs + just-in-time weight fragments
  //   wait (counted) for this wave's pieces of K-tile kt+1, barrier
  //   micro-steps 4..15: MFMAs + weight fragments + the 8 prefetch pieces of K-tile kt+3 + the 8 activation
  //                      fragments of K-tile kt+1 (other register buffer) + weight fragment 0 of K-tile kt+1
  // RAW: K-tile kt+1 is read only after every wave's counted wait and the barrier.  WAR: the refilled stage
  // (kt+3)%4 == (kt-1)%4 was last read in iteration kt-1, which every wave has left once it passes this barrier.
  auto ktile = [&](auto RBv, int kt, auto FIRSTv) {
    using RB = decltype(RBv);
    using NRB = std::integral_constant<int, RB::value ^ 1>;
    using FIRST = decltype(FIRSTv);
    const char* buf = smem + (kt & (G6_STAGES - 1)) * G6_BUF;
    const char* nbuf = smem + ((kt + 1) & (G6_STAGES - 1)) * G6_BUF;
#define IC(N) std::integral_constant<int, N>{}
#define G6_MFMA_W(J)                                                   \
    mfma4(RB{}, IC(J), FIRST{});                                       \
    if constexpr (!(VAR & 8) && (J) % 2 == 0 && (J) / 2 + 1 < 8) read_w(buf, IC((J) / 2 + 1));
    G6_MFMA_W(0) __builtin_amdgcn_sched_barrier(0);
    G6_MFMA_W(1) __builtin_amdgcn_sched_barrier(0);
    G6_MFMA_W(2) __builtin_amdgcn_sched_barrier(0);
    G6_MFMA_W(3) __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // outstanding: K-tile kt+1 (8 loads), kt+2 (8)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(VAR & 16)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define G6_MICRO_B(J, P)                                                                       \
    G6_MFMA_W(J)                                                                               \
    if constexpr ((P) < 8) {                                                                   \
      if constexpr (!(VAR & 4)) issue_one(kt + 3, IC((P) / 2), IC((P) % 2));                   \
      if constexpr (!(VAR & 8)) read_x(nbuf, NRB{}, IC((P) < 8 ? (P) : 0));                    \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);
    G6_MICRO_B(4, 0) G6_MICRO_B(5, 1) G6_MICRO_B(6, 2) G6_MICRO_B(7, 3) G6_MICRO_B(8, 4) G6_MICRO_B(9, 5)
    G6_MICRO_B(10, 6) G6_MICRO_B(11, 7) G6_MICRO_B(12, 8) G6_MICRO_B(13, 8)
    G6_MFMA_W(14)
    if constexpr (!(VAR & 8)) read_w(nbuf, IC(0));  // weight fragment 0 of K-tile kt+1 (fw[0] was last used by nt = 6, micro-steps 12-13)
    __builtin_amdgcn_sched_barrier(0);
    G6_MFMA_W(15) __builtin_amdgcn_sched_barrier(0);
#undef G6_MICRO_B
#undef G6_MFMA_W
#undef IC
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if constexpr (VAR & 32) t_loop0 = __builtin_amdgcn_s_memtime();
  ktile(I0{}, 0, I1{});  // first K-tile: accumulators := a.b
  int kt = 1;
  for (; kt + 1 < nk; kt += 2) {
    ktile(I1{}, kt, I0{});
    ktile(I0{}, kt + 1, I0{});
  }
  if (kt < nk) ktile(I1{}, kt, I0{});
  if constexpr (VAR & 32) t_loop1 = __builtin_amdgcn_s_memtime();

  // the last MFMAs must have retired before the accumulator file is read (no interlock for asm readers)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  gemm_epilogue16<T, ACT>(p, m0 + wr * 128, n0 + wc * 128, c16, g);
  if constexpr (VAR & 32) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    const unsigned long long r_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && p.dbg != nullptr) {
      unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
      d[0] = t_begin; d[1] = t_loop0; d[2] = t_loop1; d[3] = t_end; d[4] = r_begin; d[5] = r_end;
      d[6] = __builtin_amdgcn_s_getreg((6 << 11) | (0 << 6) | 20 /* HW_REG_XCC_ID */);
      d[7] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4 /* HW_REG_HW_ID */);
    }
  }
}
