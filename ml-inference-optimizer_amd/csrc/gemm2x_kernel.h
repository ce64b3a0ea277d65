// 256x128 bf16/fp16 GEMM on v_mfma_f32_16x16x32, 4 waves per workgroup, TWO workgroups per CU.
//
// Sibling of gemm4w16_kernel (same operand orientation, LDS swizzle, buffer_load-to-LDS prefetch, asm-owned
// accumulators, epilogue) with half the tile in N, so that a workgroup needs 256 registers per wave
// (128 accumulators + <= 128 VGPRs) and 72 KiB of LDS (3 K-tile stages) and two of them share a CU:
//   * the two waves on a SIMD belong to DIFFERENT output tiles and are not synchronised with each other, so
//     one wave's fragment reads / prefetch issue / waits fill behind the other's MFMAs (measured on the
//     one-wave-per-SIMD kernel: 19.4 cycles per 16-cycle MFMA, 16.4 with the non-MFMA work removed), and
//   * one workgroup's epilogue (accumulator read-out, bias/activation, 16-byte stores: ~20 % of a K=1024 tile,
//     HBM-write bound when every CU stores at once) overlaps the other workgroup's main loop once the two drift
//     out of phase.
// Wave (wr, wc) owns x rows wr*128..+127 and w rows wc*64..+63: 8 x 4 accumulator tiles of 16x16 in a[0:127].
#pragma once
#include <type_traits>

#include "gemm4w16_kernel.h"

constexpr int G2_BK = 32;
constexpr int G2_STAGES = 3;
constexpr int G2_XT = 256 * G2_BK * 2;  // 16 KiB
constexpr int G2_WT = 128 * G2_BK * 2;  // 8 KiB
constexpr int G2_BUF = G2_XT + G2_WT;   // 24 KiB
constexpr int G2_SMEM = G2_STAGES * G2_BUF;  // 72 KiB

// Epilogue shared with gemm4w16_kernel, parameterised on the number of 16-column tiles per wave.
template <typename T, int ACT, int NT_>
__device__ __forceinline__ void gemm_epilogue16n(const GemmDev& p, int64_t mbase, int nbase, int c16, int g) {
  using X4 = typename DT<T>::x4;
  int ncol[NT_];
  X4 bv[NT_];
#pragma unroll
  for (int nt = 0; nt < NT_; ++nt) {
    const int n = nbase + nt * 16 + 4 * g;
    ncol[nt] = (n < p.N) ? n : (p.N - 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[nt][e] = (T)0.f;
  }
  if (p.bias) {
#pragma unroll
    for (int nt = 0; nt < NT_; ++nt) bv[nt] = *(const X4*)((const T*)p.bias + ncol[nt]);
  }
  auto row_pair = [&](auto MTP) {
    constexpr int mt = 2 * decltype(MTP)::value;
    const int64_t mA = mbase + mt * 16 + c16, mB = mA + 16;
    const int64_t mAc = mA < p.M ? mA : (p.M - 1), mBc = mB < p.M ? mB : (p.M - 1);
    X4 rA[NT_], rB[NT_];
    if (p.res) {
      const T* ra = (const T*)p.res + mAc * p.ldr;
      const T* rb = (const T*)p.res + mBc * p.ldr;
#pragma unroll
      for (int nt = 0; nt < NT_; ++nt) {
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
    if constexpr (NT_ == 8) {
      col_block(std::integral_constant<int, 4 % NT_>{}); col_block(std::integral_constant<int, 5 % NT_>{});
      col_block(std::integral_constant<int, 6 % NT_>{}); col_block(std::integral_constant<int, 7 % NT_>{});
    }
  };
  row_pair(std::integral_constant<int, 0>{}); row_pair(std::integral_constant<int, 1>{});
  row_pair(std::integral_constant<int, 2>{}); row_pair(std::integral_constant<int, 3>{});
}


template <typename T, int ACT>
__global__ __launch_bounds__(256, 2) void gemm2x_kernel(const GemmDev p) {
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
  const int n0 = tn * 128;
  const int nk = p.K / G2_BK;

  // ---- prefetch addressing (see gemm4w16_kernel): wave w owns X pieces 4w..4w+3 and W pieces 2w, 2w+1
  const int prow = lane >> 2, pcs = lane & 3;
  int xvo[4], wvo[2];
  const int64_t mrem = p.M - m0;
  const int nrem = p.N - n0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 16 + prow;  // 0..255
    const int xr = (row < mrem) ? row : (int)(mrem - 1);
    xvo[i] = xr * (int)p.ldx * 2 + 16 * (pcs ^ g6_swz(row));
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 16 + prow;  // 0..127
    const int wrw = (row < nrem) ? row : (nrem - 1);
    wvo[i] = wrw * (int)p.ldw * 2 + 16 * (pcs ^ g6_swz(row));
  }
  const __amdgpu_buffer_rsrc_t xrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.x + m0 * p.ldx), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.w + (int64_t)n0 * p.ldw), 0, 0x7fffffff, 0x00020000);

  // piece q of this wave for K-tile kt: q = 0..3 -> X pieces, 4..5 -> W pieces; stage kt % 3.
  // Branch-free: past the last K-tile the last one is re-fetched into an idle stage.
  auto issue_one = [&](int kt, auto Q) {
    constexpr int q = decltype(Q)::value;
    const int kte = kt < nk ? kt : nk - 1;
    char* st = smem + (kt % G2_STAGES) * G2_BUF;
    const int koff = kte * (G2_BK * 2);
    if constexpr (q < 4)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (MIO_LDS void*)(st + (wave * 4 + q) * 1024), 16, xvo[q], koff, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (MIO_LDS void*)(st + G2_XT + (wave * 2 + (q - 4)) * 1024), 16,
                                               wvo[q - 4], koff, 0, 0);
  };

  {  // zero the 32 accumulator tiles a[0:127]
    auto zero_all = [&](auto self, auto K) {
      constexpr int k = decltype(K)::value;
      G6AccIO<k>::zero();
      if constexpr (k + 1 < 32) self(self, std::integral_constant<int, k + 1>{});
    };
    zero_all(zero_all, std::integral_constant<int, 0>{});
  }

  const int co = (g ^ g6_swz(c16)) * 16;
  const int xbase = (wr * 128 + c16) * 64 + co;
  const int wbase = G2_XT + (wc * 64 + c16) * 64 + co;

  X8 fx[2][8];
  X8 fw[2];
  auto read_x = [&](const char* buf, auto RB, auto MT) {
    constexpr int rb = decltype(RB)::value, mt = decltype(MT)::value;
    fx[rb][mt] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + xbase + mt * 16 * 64));
  };
  auto read_w = [&](const char* buf, auto NT) {
    constexpr int nt = decltype(NT)::value;
    fw[nt & 1] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + wbase + nt * 16 * 64));
  };
  auto mfma4 = [&](auto RB, auto J) {  // MFMAs 4j..4j+3 (nt-major): nt = j/2, mt = (j%2)*4..; tile k = nt*8 + mt
    constexpr int rb = decltype(RB)::value, j = decltype(J)::value;
    constexpr int nt = j / 2, mt0 = (j % 2) * 4;
    G6Acc<T, nt * 8 + mt0 + 0>::mfma(fw[nt & 1], fx[rb][mt0 + 0]);
    G6Acc<T, nt * 8 + mt0 + 1>::mfma(fw[nt & 1], fx[rb][mt0 + 1]);
    G6Acc<T, nt * 8 + mt0 + 2>::mfma(fw[nt & 1], fx[rb][mt0 + 2]);
    G6Acc<T, nt * 8 + mt0 + 3>::mfma(fw[nt & 1], fx[rb][mt0 + 3]);
  };
#define IC(N) std::integral_constant<int, N>{}

  // ---- prologue: K-tiles 0 and 1 in flight; K-tile 0 must have landed
  issue_one(0, IC(0)); issue_one(0, IC(1)); issue_one(0, IC(2)); issue_one(0, IC(3)); issue_one(0, IC(4)); issue_one(0, IC(5));
  issue_one(1, IC(0)); issue_one(1, IC(1)); issue_one(1, IC(2)); issue_one(1, IC(3)); issue_one(1, IC(4)); issue_one(1, IC(5));
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < 8; ++t) fx[0][t] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + xbase + t * 16 * 64));
  fw[0] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + wbase));
  __builtin_amdgcn_sched_barrier(0);

  // One K-tile (32 MFMAs = 8 micro-steps of 4) on activation register buffer RB:
  //   micro-steps 0,1 : MFMAs + just-in-time weight fragment
  //   wait for this wave's pieces of K-tile kt+1 (all but the newest 0 loads: only kt+1 is outstanding), barrier
  //   micro-steps 2..7: MFMAs + weight fragments + the 6 prefetch pieces of K-tile kt+2 (stage (kt-1)%3, which every
  //                     wave has left once it passes this barrier) + the 8 activation fragments of K-tile kt+1
  auto ktile = [&](auto RBv, int kt) {
    using RB = decltype(RBv);
    using NRB = std::integral_constant<int, RB::value ^ 1>;
    const char* buf = smem + (kt % G2_STAGES) * G2_BUF;
    const char* nbuf = smem + ((kt + 1) % G2_STAGES) * G2_BUF;
    mfma4(RB{}, IC(0)); read_w(buf, IC(1)); __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(1)); __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(2)); read_w(buf, IC(2)); issue_one(kt + 2, IC(0)); read_x(nbuf, NRB{}, IC(0)); read_x(nbuf, NRB{}, IC(1));
    __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(3)); issue_one(kt + 2, IC(1)); read_x(nbuf, NRB{}, IC(2));
    __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(4)); read_w(buf, IC(3)); issue_one(kt + 2, IC(2)); read_x(nbuf, NRB{}, IC(3));
    __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(5)); issue_one(kt + 2, IC(3)); read_x(nbuf, NRB{}, IC(4)); read_x(nbuf, NRB{}, IC(5));
    __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(6)); read_w(nbuf, IC(0)); issue_one(kt + 2, IC(4)); read_x(nbuf, NRB{}, IC(6));
    __builtin_amdgcn_sched_barrier(0);
    mfma4(RB{}, IC(7)); issue_one(kt + 2, IC(5)); read_x(nbuf, NRB{}, IC(7));
    __builtin_amdgcn_sched_barrier(0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    ktile(I0{}, kt);
    ktile(I1{}, kt + 1);
  }
  if (kt < nk) ktile(I0{}, kt);
#undef IC

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  gemm_epilogue16n<T, ACT, 4>(p, m0 + wr * 128, n0 + wc * 64, c16, g);
}
