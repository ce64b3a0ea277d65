// bf16/fp16 GEMM with fused epilogue for CDNA4 (gfx950):
//   y[M,N] = act(x[M,K] @ w[N,K]^T + bias) (+ residual)            (ACT != SWIGLU)
//   y[M,N] = silu(x @ wg^T + bg) * (x @ w^T + b)                   (ACT == SWIGLU, dual-B)
// This is the contraction under FusedMLP (reference kernels/triton/mlp_kernels.py:27-641: fc1 -> act
// -> fc2 tiled GEMM nests at :91-126 and :171-198) and under the q/k/v/o projections.
//
// Structure: BMxBN output tile per workgroup, BK = 64, both operands K-contiguous ("NT").
//   * global -> LDS with direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per wave-instruction,
//     no VGPR staging).  The LDS image is lane-linear, so the bank-conflict swizzle
//     (16-B chunk c of row r stored at c ^ ((r>>1)&7)) is applied to the per-lane SOURCE address
//     and again on the ds_read_b128 side.
//   * v_mfma_f32_32x32x16 with the WEIGHT tile as the A operand and the activation tile as the B
//     operand: the accumulator then has the output ROW (m) on the lane and 4 consecutive output
//     columns per register group, so bias/activation/residual are lane-local and the stores are
//     8-byte row segments.
//   * K tails (K % 64 != 0, K % 8 == 0) and M/N edges: out-of-range chunks are sourced from a
//     16-byte zero block / clamped rows, so there is no separate edge kernel.
#pragma once
#include "mio_common.h"

struct GemmDev {
  const void* x;
  const void* w;
  const void* wg;
  const void* bias;
  const void* bias_g;
  const void* res;
  void* y;
  int64_t M, ldx, ldw, ldy, ldr;
  int N, K;
  int tiles_m, tiles_n;
  // Blocked activation layout (FusedMLP's intermediate only, set by mio_fused_mlp_fwd): element (m, k) lives at byte
  // ((m / 256 * (K / 32) + k / 32) * 256 + m % 256) * 64 + (k % 32) * 2, i.e. every (256-row, 32-column) K-tile of
  // the consuming GEMM is one contiguous 16 KiB block.  With the plain [M, I] layout a K-tile is 256 pieces of 64 B
  // at a stride of 2 I bytes: at I = 4096 that access pattern, not the matrix pipe, bounds the second GEMM.
  int x_blk, y_blk;
  // Blocked weight (mio_weight_block, one-time repack): the same layout with n in the place of m, rows padded to 256.
  int w_blk;
  unsigned long long* dbg;  // diagnostic builds only (in-kernel stamps); nullptr otherwise
  // Column scale (persistent kernel only): output columns [cs_lo, cs_hi) are multiplied by cs_val in fp32 AFTER bias /
  // activation and BEFORE the rounding to 16 bits; cs_lo, cs_hi multiples of 128 (a wave's column range is inside or
  // outside).  Use: the fused QKV projection hands the attention kernel K * softmax_scale * log2(e) with ONE rounding
  // (mio_fa3_fwd k_prescaled).  cs_lo >= cs_hi: off.
  int cs_lo, cs_hi;
  float cs_val;
  int group_m;  // row tiles that share each weight tile inside an XCD's run of tiles (0: the default 8)
  // LayerNorm folded into the GEMMs on either side of it (gemm8w_kernel FOLD, mio_gemm_ln_bw; reference
  // kernels/triton/fused_layernorm_qkv.py:37-420, layernorm_kernels.py:35-188):
  //   producer (the GEMM that writes the residual stream, FOLD = 2): stats_out[slot][row] = (sum, sum of squares) of the ROUNDED
  //     output row over the tile column `slot` (N / 256 slots, rows padded to whole 256-row tiles);
  //   consumer (the projection behind the LayerNorm, FOLD = 1): w holds the gamma-scaled weights with every row's mean over k
  //     subtracted ((x - mean 1) . w = x . (w - mean(w) 1): the centring moves from the activations to the weights), bias the
  //     beta-folded bias; the read-out computes acc * rstd + bias, rstd from ln_stats (ln_slots slots).
  int res_blk;             // the residual is in the blocked activation layout (width N)
  const float* ln_stats;   // consumer: [ln_slots][tiles_m * 256][2]
  float* stats_out;        // producer: [N / 256][tiles_m * 256][2]
  int ln_slots;
  float ln_eps;
};

constexpr int GEMM_LN_SLOTS_MAX = 8;  // 256-column statistic slots a folded LayerNorm may span (row width <= 2048): LDS region of gemm8w_kernel

static inline void gemm_dev_defaults(GemmDev& p) {
  p.wg = nullptr; p.bias = nullptr; p.bias_g = nullptr; p.res = nullptr;
  p.tiles_m = p.tiles_n = 0;
  p.x_blk = p.y_blk = p.w_blk = 0;
  p.dbg = nullptr;
  p.cs_lo = p.cs_hi = 0; p.cs_val = 1.f; p.group_m = 0;
  p.res_blk = 0; p.ln_stats = nullptr; p.stats_out = nullptr; p.ln_slots = 0; p.ln_eps = 0.f;
}

int mio_gemm_impl();  // MIO_GEMM_IMPL override (0 = default dispatch); defined in gemm_api.hip

constexpr int GEMM_BK = 64;

static __device__ __attribute__((aligned(16))) const uint32_t mio_zero16[4] = {0, 0, 0, 0};

template <int ACT>
__device__ __forceinline__ float gemm_act(float v) {
  if (ACT == MIO_ACT_GELU_TANH) {
    // 0.5 v (1 + tanh(u)) = v * sigmoid(2u), u = sqrt(2/pi) (v + 0.044715 v^3)   (mlp_kernels.py:144-161)
    const float u2 = 2.0f * 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return v * fast_rcp(1.0f + fast_exp2(-u2 * 1.4426950408889634f));
  } else if (ACT == MIO_ACT_GELU_ERF) {
    return 0.5f * v * (1.0f + erff(v * 0.7071067811865476f));
  } else if (ACT == MIO_ACT_RELU) {
    return fmaxf(v, 0.f);
  } else if (ACT == MIO_ACT_SILU) {
    return v * fast_rcp(1.0f + fast_exp2(-v * 1.4426950408889634f));
  }
  return v;
}

// Two values at once with packed fp32 math (v_pk_mul/fma/add_f32): used where no MFMA is in flight (the read-out of
// the persistent kernel), where packed ops run at twice the rate of the scalar forms.  Same formulas as gemm_act,
// constants folded: gelu_tanh(v) = v / (1 + 2^(-v (c0 + c1 v^2))), c0 = 2 sqrt(2/pi) log2(e), c1 = 0.044715 c0.
// SCALAR = true: the plain one-value formulas (A/B only: wrong stores first blamed on the packed forms turned out to be store
// data overwritten behind the store instruction, gemm8w_kernel.h).
template <int ACT, bool SCALAR = false>
__device__ __forceinline__ f32x2_t gemm_act2(f32x2_t v) {
  if constexpr (SCALAR) {
    return (f32x2_t){gemm_act<ACT>(v[0]), gemm_act<ACT>(v[1])};
  } else if constexpr (ACT == MIO_ACT_GELU_TANH) {
    constexpr float c0 = 2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
    const f32x2_t t = v * v;
    const f32x2_t z = -(v * (t * c1 + c0));
    const f32x2_t d = (f32x2_t){fast_exp2(z[0]), fast_exp2(z[1])} + 1.0f;
    return v * (f32x2_t){fast_rcp(d[0]), fast_rcp(d[1])};
  } else if constexpr (ACT == MIO_ACT_SILU) {
    const f32x2_t z = v * -1.4426950408889634f;
    const f32x2_t d = (f32x2_t){fast_exp2(z[0]), fast_exp2(z[1])} + 1.0f;
    return v * (f32x2_t){fast_rcp(d[0]), fast_rcp(d[1])};
  } else {
    return (f32x2_t){gemm_act<ACT>(v[0]), gemm_act<ACT>(v[1])};
  }
}

// XCD-aware, grouped tile order: consecutive ids (round-robin over the 8 XCDs) are folded so each
// XCD walks a contiguous run of tiles; inside a run GROUP_M row-tiles share each weight tile.
__device__ __forceinline__ void gemm_tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn, int GROUP_M = 8) {
  const int nwg = tiles_m * tiles_n;
  const int q = nwg >> 3, rr = nwg & 7;
  const int xcd = id & 7, idx = id >> 3;
  const int pid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  const int per_group = GROUP_M * tiles_n;
  const int g = pid / per_group;
  const int first_m = g * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int within = pid - g * per_group;
  tm = first_m + within % gsz;
  tn = within / gsz;
}

// ---- shared epilogue --------------------------------------------------------------------------------
// Accumulator layout (weight tile = A operand, activation tile = B operand): lane (r, h) holds output row
// mbase + mt*32 + r and, per register group g, the 4 consecutive columns nbase + nt*32 + 8g + 4h + (0..3).
// All bias / residual loads of a row block are issued before the first use (clamped addresses instead of
// branches), so the epilogue pays one memory round trip per row block instead of one per 8-byte store.
template <typename T, int ACT, int NT, int MT>
__device__ __forceinline__ void gemm_epilogue(const GemmDev& p, f32x16_t (&acc)[NT][MT],
                                              f32x16_t (&accg)[(ACT == MIO_ACT_SWIGLU) ? NT : 1][(ACT == MIO_ACT_SWIGLU) ? MT : 1],
                                              int64_t mbase, int nbase, int r, int h) {
  using X4 = typename DT<T>::x4;
  constexpr bool GATE = (ACT == MIO_ACT_SWIGLU);
  int ncol[NT][4];
  bool nok[NT][4];
  X4 bv[NT][4], bgv[GATE ? NT : 1][GATE ? 4 : 1];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nbase + nt * 32 + 8 * g + 4 * h;
      nok[nt][g] = n < p.N;
      ncol[nt][g] = nok[nt][g] ? n : (p.N - 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[nt][g][e] = (T)0.f;
    }
  if (p.bias) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[nt][g] = *(const X4*)((const T*)p.bias + ncol[nt][g]);
  }
  if constexpr (GATE) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bgv[nt][g][e] = (T)0.f;
        if (p.bias_g) bgv[nt][g] = *(const X4*)((const T*)p.bias_g + ncol[nt][g]);
      }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int64_t m = mbase + mt * 32 + r;
    const bool mok = m < p.M;
    const int64_t mc = mok ? m : (p.M - 1);
    T* yrow = (T*)p.y + mc * p.ldy;
    X4 rv[NT][4];
    if (p.res) {
      const T* rrow = (const T*)p.res + mc * p.ldr;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) rv[nt][g] = *(const X4*)(rrow + ncol[nt][g]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][4 * g + e] + (float)bv[nt][g][e];
        if constexpr (GATE) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = gemm_act<MIO_ACT_SILU>(accg[nt][mt][4 * g + e] + (float)bgv[nt][g][e]) * v[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gemm_act<ACT>(v[e]);
        }
        if (p.res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)rv[nt][g][e];
        }
        if (mok && nok[nt][g]) {
          u32x2_t o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
          *(u32x2_t*)(yrow + ncol[nt][g]) = o;
        }
      }
  }
}

template <typename T, int BM, int BN, int WM, int WN, int ACT>
__global__ __launch_bounds__(WM* WN * 64) void gemm_bias_act_kernel(const GemmDev p) {
  using X8 = typename DT<T>::x8;
  constexpr bool GATE = (ACT == MIO_ACT_SWIGLU);
  constexpr int NWAVE = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;  // per-wave output tile
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int XB = BM * GEMM_BK * 2;       // bytes of one x tile
  constexpr int WB = BN * GEMM_BK * 2;       // bytes of one w tile
  constexpr int STAGE = XB + WB * (GATE ? 2 : 1);
  constexpr int X_PIECES = BM / 8 / NWAVE;   // 1-KiB direct-to-LDS pieces per wave per tile
  constexpr int W_PIECES = BN / 8 / NWAVE;
  static_assert(BM % (8 * NWAVE) == 0 && BN % (8 * NWAVE) == 0, "tile/wave mismatch");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  int tm, tn;
  gemm_tile_coords(blockIdx.x, p.tiles_m, p.tiles_n, tm, tn);
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = tn * BN;

  // ---- direct-to-LDS staging: lane -> (row in 8-row piece, stored chunk); source chunk un-swizzled
  const int prow = lane >> 3, pcs = lane & 7;
  const T* xg = (const T*)p.x;
  const T* wg_ = (const T*)p.w;
  const T* gg = (const T*)p.wg;

  auto stage = [&](int kt, int buf) {
    char* sx = smem + buf * STAGE;
    char* sw = sx + XB;
    const int k0 = kt * GEMM_BK;
#pragma unroll
    for (int i = 0; i < X_PIECES; ++i) {
      const int piece = wave * X_PIECES + i;
      const int row = piece * 8 + prow;
      const int c = pcs ^ ((row >> 1) & 7);
      int64_t gr = m0 + row;
      if (gr > p.M - 1) gr = p.M - 1;
      const T* src = xg + gr * p.ldx + k0 + 8 * c;
      if (k0 + 8 * c >= p.K) src = (const T*)mio_zero16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (MIO_LDS void*)(sx + piece * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
      const int piece = wave * W_PIECES + i;
      const int row = piece * 8 + prow;
      const int c = pcs ^ ((row >> 1) & 7);
      int gr = n0 + row;
      if (gr > p.N - 1) gr = p.N - 1;
      const bool kz = (k0 + 8 * c >= p.K);
      const T* src = wg_ + (int64_t)gr * p.ldw + k0 + 8 * c;
      if (kz) src = (const T*)mio_zero16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (MIO_LDS void*)(sw + piece * 1024), 16, 0, 0);
      if (GATE) {
        const T* srcg = gg + (int64_t)gr * p.ldw + k0 + 8 * c;
        if (kz) srcg = (const T*)mio_zero16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)srcg,
                                         (MIO_LDS void*)(sw + WB + piece * 1024), 16, 0, 0);
      }
    }
  };

  f32x16_t acc[NT][MT];
  f32x16_t accg[GATE ? NT : 1][GATE ? MT : 1];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[nt][mt][i] = 0.f;
        if (GATE) accg[nt][mt][i] = 0.f;
      }

  const int nk = (p.K + GEMM_BK - 1) / GEMM_BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // per-lane fragment read offsets: row (.. + r), chunk (2*ks + h) ^ swz(row)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(kt + 1, cur ^ 1);
    const char* sx = smem + cur * STAGE;
    const char* sw = sx + XB;
#pragma unroll
    for (int ks = 0; ks < GEMM_BK / 16; ++ks) {
      X8 xf[MT], wf[NT], gf[GATE ? NT : 1];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = wm * TM + mt * 32 + r;
        const int c = (2 * ks + h) ^ ((row >> 1) & 7);
        xf[mt] = __builtin_bit_cast(X8, *(const u32x4_t*)(sx + row * 128 + c * 16));
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int row = wn * TN + nt * 32 + r;
        const int c = (2 * ks + h) ^ ((row >> 1) & 7);
        wf[nt] = __builtin_bit_cast(X8, *(const u32x4_t*)(sw + row * 128 + c * 16));
        if (GATE) gf[nt] = __builtin_bit_cast(X8, *(const u32x4_t*)(sw + WB + row * 128 + c * 16));
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[nt][mt] = DT<T>::mfma32(wf[nt], xf[mt], acc[nt][mt]);
          if (GATE) accg[nt][mt] = DT<T>::mfma32(gf[nt], xf[mt], accg[nt][mt]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  gemm_epilogue<T, ACT, NT, MT>(p, acc, accg, m0 + wm * TM, n0 + wn * TN, r, h);
}

// Host launcher for one dtype; defined per translation unit (gemm_inst.hip).
template <typename T>
int gemm_launch(GemmDev p, int act, hipStream_t stream);
