// FlashAttention-3 style prefill kernel, second structure: ONE wave per SIMD, 64 query rows per wave.
//
// Same algorithm, layouts and C-ABI semantics as fa3_fwd_kernel (fa3_fwd_kernel.h; no user masks here).
// Why a second structure: at head_dim 64 the first kernel is bound by the SIMD's instruction ISSUE, not by
// the matrix pipe (measured: MFMA pipe 23 % busy; per 32x64 score tile a wave issues ~24 LDS fragment reads
// and ~200 VALU instructions for 16 MFMAs, and the two waves of a SIMD together saturate its issue slots).
// Giving each wave TWO 32-row query sub-tiles halves the K / V fragment reads per MFMA (each fragment feeds
// two MFMAs), and the two sub-tiles are independent instruction streams the scheduler can interleave
// (softmax VALU of one under the MFMAs of the other).
//
//   * workgroup = 4 waves = 256 query rows of one (batch, head); KV tile = 64 keys; K/V staging, LDS images,
//     swapped QK^T, P^T-as-B-operand and transposed V reads exactly as in fa3_fwd_kernel.
//   * registers: the score tiles, P fragments, Q fragments and staging registers need ~230 arch VGPRs, so the
//     O^T accumulators (QT x D/32 tiles of 16 registers) live in the accumulator file a[0:..], owned by inline asm
//     (only MFMAs touch them in steady state; the deferred rescale and the epilogue go through v_accvgpr moves).
#pragma once
#include <type_traits>

#include "fa3_fwd_kernel.h"

constexpr int FA2_QT = 2;             // 32-row query sub-tiles per wave
constexpr int FA2_BM = 4 * 32 * FA2_QT;  // 256 query rows per workgroup

// Accumulator tile k (16 registers) = a[16k : 16k+15], asm-owned (see gemm4w16_acc.inc for the pattern).
template <typename T, int K>
struct Fa2Acc;
#define FA2_CL(B) "a" #B
#define FA2_DEF(K, R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11, R12, R13, R14, R15)                          \
  template <>                                                                                                     \
  struct Fa2Acc<__bf16, K> {                                                                                      \
    static __device__ __forceinline__ void mfma(bf16x8_t a, bf16x8_t b) {                                         \
      asm volatile("v_mfma_f32_32x32x16_bf16 a[" #R0 ":" #R15 "], %0, %1, a[" #R0 ":" #R15 "]"                     \
                   :                                                                                              \
                   : "v"(a), "v"(b)                                                                               \
                   : FA2_CL(R0), FA2_CL(R1), FA2_CL(R2), FA2_CL(R3), FA2_CL(R4), FA2_CL(R5), FA2_CL(R6), FA2_CL(R7), \
                     FA2_CL(R8), FA2_CL(R9), FA2_CL(R10), FA2_CL(R11), FA2_CL(R12), FA2_CL(R13), FA2_CL(R14),      \
                     FA2_CL(R15));                                                                                \
    }                                                                                                             \
  };                                                                                                              \
  template <>                                                                                                     \
  struct Fa2Acc<_Float16, K> {                                                                                    \
    static __device__ __forceinline__ void mfma(f16x8_t a, f16x8_t b) {                                           \
      asm volatile("v_mfma_f32_32x32x16_f16 a[" #R0 ":" #R15 "], %0, %1, a[" #R0 ":" #R15 "]"                      \
                   :                                                                                              \
                   : "v"(a), "v"(b)                                                                               \
                   : FA2_CL(R0), FA2_CL(R1), FA2_CL(R2), FA2_CL(R3), FA2_CL(R4), FA2_CL(R5), FA2_CL(R6), FA2_CL(R7), \
                     FA2_CL(R8), FA2_CL(R9), FA2_CL(R10), FA2_CL(R11), FA2_CL(R12), FA2_CL(R13), FA2_CL(R14),      \
                     FA2_CL(R15));                                                                                \
    }                                                                                                             \
  };                                                                                                              \
  template <>                                                                                                     \
  struct Fa2AccIO<K> {                                                                                            \
    template <int G>                                                                                              \
    static __device__ __forceinline__ f32x4_t read4() { /* registers 4G..4G+3 of the tile */                      \
      float x0, x1, x2, x3;                                                                                       \
      if constexpr (G == 0)                                                                                       \
        asm volatile("v_accvgpr_read_b32 %0, a" #R0 "\n\tv_accvgpr_read_b32 %1, a" #R1 "\n\tv_accvgpr_read_b32 %2, a" #R2 \
                     "\n\tv_accvgpr_read_b32 %3, a" #R3 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));                    \
      else if constexpr (G == 1)                                                                                  \
        asm volatile("v_accvgpr_read_b32 %0, a" #R4 "\n\tv_accvgpr_read_b32 %1, a" #R5 "\n\tv_accvgpr_read_b32 %2, a" #R6 \
                     "\n\tv_accvgpr_read_b32 %3, a" #R7 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));                    \
      else if constexpr (G == 2)                                                                                  \
        asm volatile("v_accvgpr_read_b32 %0, a" #R8 "\n\tv_accvgpr_read_b32 %1, a" #R9 "\n\tv_accvgpr_read_b32 %2, a" #R10 \
                     "\n\tv_accvgpr_read_b32 %3, a" #R11 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));                   \
      else                                                                                                        \
        asm volatile("v_accvgpr_read_b32 %0, a" #R12 "\n\tv_accvgpr_read_b32 %1, a" #R13 "\n\tv_accvgpr_read_b32 %2, a" #R14 \
                     "\n\tv_accvgpr_read_b32 %3, a" #R15 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));                   \
      return (f32x4_t){x0, x1, x2, x3};                                                                           \
    }                                                                                                             \
    template <int G>                                                                                              \
    static __device__ __forceinline__ void write4(f32x4_t v) {                                                    \
      if constexpr (G == 0)                                                                                       \
        asm volatile("v_accvgpr_write_b32 a" #R0 ", %0\n\tv_accvgpr_write_b32 a" #R1 ", %1\n\tv_accvgpr_write_b32 a" #R2 \
                     ", %2\n\tv_accvgpr_write_b32 a" #R3 ", %3" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])          \
                     : FA2_CL(R0), FA2_CL(R1), FA2_CL(R2), FA2_CL(R3));                                            \
      else if constexpr (G == 1)                                                                                  \
        asm volatile("v_accvgpr_write_b32 a" #R4 ", %0\n\tv_accvgpr_write_b32 a" #R5 ", %1\n\tv_accvgpr_write_b32 a" #R6 \
                     ", %2\n\tv_accvgpr_write_b32 a" #R7 ", %3" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])          \
                     : FA2_CL(R4), FA2_CL(R5), FA2_CL(R6), FA2_CL(R7));                                            \
      else if constexpr (G == 2)                                                                                  \
        asm volatile("v_accvgpr_write_b32 a" #R8 ", %0\n\tv_accvgpr_write_b32 a" #R9 ", %1\n\tv_accvgpr_write_b32 a" #R10 \
                     ", %2\n\tv_accvgpr_write_b32 a" #R11 ", %3" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])         \
                     : FA2_CL(R8), FA2_CL(R9), FA2_CL(R10), FA2_CL(R11));                                          \
      else                                                                                                        \
        asm volatile("v_accvgpr_write_b32 a" #R12 ", %0\n\tv_accvgpr_write_b32 a" #R13 ", %1\n\tv_accvgpr_write_b32 a" #R14 \
                     ", %2\n\tv_accvgpr_write_b32 a" #R15 ", %3" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])         \
                     : FA2_CL(R12), FA2_CL(R13), FA2_CL(R14), FA2_CL(R15));                                        \
    }                                                                                                             \
  };
template <int K>
struct Fa2AccIO;
FA2_DEF(0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
FA2_DEF(1, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
FA2_DEF(2, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47)
FA2_DEF(3, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63)
FA2_DEF(4, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79)
FA2_DEF(5, 80, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 93, 94, 95)
FA2_DEF(6, 96, 97, 98, 99, 100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110, 111)
FA2_DEF(7, 112, 113, 114, 115, 116, 117, 118, 119, 120, 121, 122, 123, 124, 125, 126, 127)
// tiles 8..15 = a[128:255]: the top of the accumulator file, where fa3_fwd2 / fa3_fwd3 keep their accumulators, out of
// the way of the low registers the allocator hands out first
FA2_DEF(8, 128, 129, 130, 131, 132, 133, 134, 135, 136, 137, 138, 139, 140, 141, 142, 143)
FA2_DEF(9, 144, 145, 146, 147, 148, 149, 150, 151, 152, 153, 154, 155, 156, 157, 158, 159)
FA2_DEF(10, 160, 161, 162, 163, 164, 165, 166, 167, 168, 169, 170, 171, 172, 173, 174, 175)
FA2_DEF(11, 176, 177, 178, 179, 180, 181, 182, 183, 184, 185, 186, 187, 188, 189, 190, 191)
FA2_DEF(12, 192, 193, 194, 195, 196, 197, 198, 199, 200, 201, 202, 203, 204, 205, 206, 207)
FA2_DEF(13, 208, 209, 210, 211, 212, 213, 214, 215, 216, 217, 218, 219, 220, 221, 222, 223)
FA2_DEF(14, 224, 225, 226, 227, 228, 229, 230, 231, 232, 233, 234, 235, 236, 237, 238, 239)
FA2_DEF(15, 240, 241, 242, 243, 244, 245, 246, 247, 248, 249, 250, 251, 252, 253, 254, 255)
#undef FA2_DEF

// compile-time loop helper: f(integral_constant<int, I>) for I in [0, N)
template <int N, typename F>
__device__ __forceinline__ void fa2_for(F&& f) {
  if constexpr (N > 0) {
    fa2_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// (Forcing two waves per SIMD at D = 64 -- 192 VGPRs + 64 accumulator registers would fit 256 -- makes hipcc split
// the file 128/128 and spill 86 VGPRs; left at one wave per SIMD.)
template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(256) void fa3_fwd2_kernel(const FaDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  using SM = FaSmem<D>;
  constexpr int QT = FA2_QT;
  constexpr int KS = D / 16;
  constexpr int DT_ = D / 32;
  constexpr int CPR = D / 8;
  constexpr int NLD = D / 32;
  static_assert(QT * DT_ <= 8, "accumulator tiles");
  // The O^T tiles sit at the TOP of the accumulator file (tile KO + k): the allocator hands out a0, a1, ... first for
  // its own values -- under VGPR pressure (D = 96 / 128) it parks temporaries there, including between two asm
  // statements of the rare rescale path, where registers only protected by clobber lists look free to it.
  // tests/test_host_logic.py checks after every build that no compiler-generated instruction touches the owned range.
  constexpr int KO = 16 - QT * DT_;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  int bh, qi;
  {
    const int id = blockIdx.x;
    if (p.xcd_remap & 1) {
      const int xcd = id & 7, slot = id >> 3;
      if (p.xcd_remap & 4) {  // query-block-major: all (b,h) of one query block before the next block
        const int nbh8 = (p.B * p.H) >> 3;
        qi = slot / nbh8;
        bh = (slot % nbh8) * 8 + xcd;
      } else {
        bh = (slot / p.qgrid) * 8 + xcd;
        qi = slot % p.qgrid;
      }
    } else {
      bh = id / p.qgrid;
      qi = id % p.qgrid;
    }
  }
  const int b = bh / p.H, head = bh % p.H;
  const int kvh = head / (p.H / p.Hkv);
  // Causal: this workgroup handles query block nqblk-1-qi (heavy) and then block qi (light): equal work for every
  // workgroup, and the fixed per-block cost (Q load, first K/V tile, epilogue) is paid by half as many, fuller
  // workgroups.  Non-causal: one block.
  const int npass = (CAUSAL && (p.nqblk - 1 - qi) != qi) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  const int qblk = CAUSAL ? (pass == 0 ? p.nqblk - 1 - qi : qi) : qi;
  const int q0 = qblk * FA2_BM;
  const int wrow0 = q0 + wave * (32 * QT);  // first query row of this wave
  int qrow[QT];
  bool q_ok[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    qrow[qt] = wrow0 + 32 * qt + r;
    q_ok[qt] = qrow[qt] < p.Sq;
  }

  X8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const T* qp = (const T*)p.q + b * p.qs_b + head * p.qs_h + (int64_t)(q_ok[qt] ? qrow[qt] : 0) * p.qs_s;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * h;
      u32x4_t raw = *(const u32x4_t*)(qp + (d0 < p.D ? d0 : 0));
      if (!(q_ok[qt] && d0 < p.D)) raw = (u32x4_t){0, 0, 0, 0};
      qf[qt][ks] = __builtin_bit_cast(X8, raw);
    }
  }

  // ---- running state: (m, l) per query sub-tile in registers, O^T in a[16*(qt*DT_+dt) ..]
  float m_i[QT], l_i[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_i[qt] = -INFINITY;
    l_i[qt] = 0.f;
  }
  fa2_for<QT * DT_>([&](auto K) {
    constexpr int k = decltype(K)::value;
    constexpr int qt = k / DT_, dt = k % DT_;
    f32x4_t z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (p.carry_in && q_ok[qt]) {
      const float* oa = p.o_acc + (((int64_t)b * p.Sq + qrow[qt]) * p.H + head) * p.D;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * dt + 8 * g + 4 * h;
        if (d0 < p.D) z[g] = *(const f32x4_t*)(oa + d0);
      }
    }
    Fa2AccIO<KO + k>::template write4<0>(z[0]);
    Fa2AccIO<KO + k>::template write4<1>(z[1]);
    Fa2AccIO<KO + k>::template write4<2>(z[2]);
    Fa2AccIO<KO + k>::template write4<3>(z[3]);
  });
  if (p.carry_in) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
      if (q_ok[qt]) {
        const float lse_in = p.lse[((int64_t)b * p.H + head) * p.Sq + qrow[qt]];
        if (lse_in != -INFINITY) {
          m_i[qt] = lse_in * FA_LOG2E;
          l_i[qt] = (h == 0) ? 1.f : 0.f;
        }
      }
  }
  // consume the Q fragments / carried scalars here so hipcc waits for those loads outside the tile loop
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[qt][ks]));
    asm volatile("" ::"v"(m_i[qt]));
  }

  int n_tiles;
  if (CAUSAL) {
    int kmax = q0 + FA2_BM - 1 + p.q_offset - p.k_offset;
    if (kmax > p.Sk - 1) kmax = p.Sk - 1;
    n_tiles = kmax < 0 ? 0 : kmax / FA_BN + 1;
  } else {
    n_tiles = (p.Sk + FA_BN - 1) / FA_BN;
  }

  // ---- K/V staging (as fa3_fwd_kernel: unconditional asm loads, zeroing at LDS-write time for partial tiles)
  const T* kbase = (const T*)p.k + b * p.ks_b + kvh * p.ks_h;
  const T* vbase = (const T*)p.v + b * p.vs_b + kvh * p.vs_h;
  u32x4_t kreg[NLD], vreg[NLD];
  const int d_chunks = p.D >> 3;
  auto stage_load = [&](int tile) {
    const int kv0 = tile * FA_BN;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = tid + 256 * j;
      const int row = i / CPR, c = i % CPR;
      int kv = kv0 + row;
      kv = kv < p.Sk ? kv : p.Sk - 1;
      const int cc = c < d_chunks ? c : d_chunks - 1;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kreg[j]) : "v"(kbase + (int64_t)kv * p.ks_s + 8 * cc) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vreg[j]) : "v"(vbase + (int64_t)kv * p.vs_s + 8 * cc) : "memory");
    }
  };
  auto stage_write = [&](int buf, int tile) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[j]), "+v"(vreg[j])::"memory");
    char* kb = smem + buf * SM::STAGE;
    char* vb = kb + SM::K_BYTES;
    const int kv0 = tile * FA_BN;
    const bool partial = (kv0 + FA_BN > p.Sk) || (d_chunks != CPR);
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = tid + 256 * j;
      const int row = i / CPR, c = i % CPR;
      u32x4_t kk = kreg[j], vv = vreg[j];
      if (partial && !((kv0 + row < p.Sk) && (c < d_chunks))) {
        kk = (u32x4_t){0, 0, 0, 0};
        vv = (u32x4_t){0, 0, 0, 0};
      }
      *(u32x4_t*)(kb + row * SM::KROW + 16 * c) = kk;
      *(u32x4_t*)(vb + ((row >> 3) * DT_ + (c >> 2)) * 512 + (row & 7) * 64 + (c & 3) * 16) = vv;
    }
  };

  if (n_tiles > 0) {
    stage_load(0);
    stage_write(0, 0);
  }
  __syncthreads();

  const int k_rd = r * SM::KROW + 16 * h;
  const int g16 = lane >> 4, i16 = lane & 15;
  const int v_rd = (4 * h + (i16 >> 2)) * 64 + 32 * (g16 & 1) + 8 * (i16 & 3);
  const float c2 = p.scale_log2e;

  auto process_tile = [&](int t, auto EDGE_) {
    constexpr bool EDGE = decltype(EDGE_)::value;
    const int cur = t & 1;
    const int kv0 = t * FA_BN;
    const char* kb = smem + cur * SM::STAGE;
    const char* vb = kb + SM::K_BYTES;

    // ---- S^T[qt] = K . Q[qt]^T: every K fragment feeds QT MFMAs
    f32x16_t sc[QT][2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        sc[qt][0][i] = 0.f;
        sc[qt][1][i] = 0.f;
      }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const X8 a0 = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd + 32 * ks));
      const X8 a1 = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd + 32 * SM::KROW + 32 * ks));
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        sc[qt][0] = DT<T>::mfma32(a0, qf[qt][ks], sc[qt][0]);
        sc[qt][1] = DT<T>::mfma32(a1, qf[qt][ks], sc[qt][1]);
      }
    }

    // ---- V^T fragments for the whole tile, issued NOW (one wave per SIMD: nobody else hides an LDS round trip;
    // left inline, every pair of transposed reads stalls the two MFMAs it feeds): they fly under the softmax VALU
    X8 vfr[4][DT_];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt) {
        const X4 lo = DT<T>::ds_read_tr(vb + v_rd + ((2 * s + 0) * DT_ + dt) * 512);
        const X4 hi = DT<T>::ds_read_tr(vb + v_rd + ((2 * s + 1) * DT_ + dt) * 512);
        vfr[s][dt][0] = lo[0]; vfr[s][dt][1] = lo[1]; vfr[s][dt][2] = lo[2]; vfr[s][dt][3] = lo[3];
        vfr[s][dt][4] = hi[0]; vfr[s][dt][5] = hi[1]; vfr[s][dt][6] = hi[2]; vfr[s][dt][7] = hi[3];
      }
    __builtin_amdgcn_sched_barrier(0);

    // ---- softmax per query sub-tile -> P^T fragments
    X8 pf[QT][4];
    fa2_for<QT>([&](auto QTI) {
      constexpr int qt = decltype(QTI)::value;
      float mx;
      if constexpr (!EDGE) {
        mx = fmaxf(fmaxf(sc[qt][0][0], sc[qt][0][1]), sc[qt][0][2]);
#pragma unroll
        for (int i = 3; i + 1 < 16; i += 2) mx = fmaxf(fmaxf(mx, sc[qt][0][i]), sc[qt][0][i + 1]);
        mx = fmaxf(fmaxf(mx, sc[qt][0][15]), sc[qt][1][0]);
#pragma unroll
        for (int i = 1; i + 1 < 16; i += 2) mx = fmaxf(fmaxf(mx, sc[qt][1][i]), sc[qt][1][i + 1]);
        mx = fmaxf(mx, sc[qt][1][15]) * c2;
      } else {
        const int q_pos = qrow[qt] + p.q_offset;
        mx = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int kv = kv0 + 32 * tt + (i & 3) + 8 * (i >> 2) + 4 * h;
            float tv = sc[qt][tt][i] * c2;
            if ((kv >= p.Sk) || (CAUSAL && (kv + p.k_offset > q_pos))) tv = -INFINITY;
            sc[qt][tt][i] = tv;
            mx = fmaxf(mx, tv);
          }
      }
      mx = fmaxf(mx, other_half(mx));

      float m_sub;
      if (__builtin_amdgcn_ballot_w64(mx > m_i[qt] + FA_RESCALE_THR) != 0) {
        const float m_new = fmaxf(m_i[qt], mx);
        m_sub = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = fast_exp2(m_i[qt] - m_sub);
        l_i[qt] *= alpha;
        m_i[qt] = m_new;
        // rescale this sub-tile's O^T accumulators through the VALU (rare: deferred-rescale threshold)
        fa2_for<DT_>([&](auto DTI) {
          constexpr int k = qt * DT_ + decltype(DTI)::value;
          f32x4_t v[4] = {Fa2AccIO<KO + k>::template read4<0>(), Fa2AccIO<KO + k>::template read4<1>(),
                          Fa2AccIO<KO + k>::template read4<2>(), Fa2AccIO<KO + k>::template read4<3>()};
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[gq][e] *= alpha;
          Fa2AccIO<KO + k>::template write4<0>(v[0]);
          Fa2AccIO<KO + k>::template write4<1>(v[1]);
          Fa2AccIO<KO + k>::template write4<2>(v[2]);
          Fa2AccIO<KO + k>::template write4<3>(v[3]);
        });
      } else {
        m_sub = m_i[qt];
      }
      float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = sc[qt][s >> 1][8 * (s & 1) + j];
          e[j] = EDGE ? fast_exp2(x - m_sub) : fast_exp2(__builtin_fmaf(x, c2, -m_sub));
        }
        rs0 += (e[0] + e[1]) + (e[2] + e[3]);
        rs1 += (e[4] + e[5]) + (e[6] + e[7]);
        u32x4_t w;
        w[0] = pack2<T>(e[0], e[1]);
        w[1] = pack2<T>(e[2], e[3]);
        w[2] = pack2<T>(e[4], e[5]);
        w[3] = pack2<T>(e[6], e[7]);
        pf[qt][s] = __builtin_bit_cast(X8, w);
      }
      l_i[qt] += rs0 + rs1;
    });
    // P fragments were just written by the VALU and are read by MFMAs inside asm statements: hipcc inserts no
    // wait states there, so pad once
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 4" ::: "memory");

    // ---- O^T[qt] += V^T . P^T[qt]: every V fragment feeds QT MFMAs
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fa2_for<DT_>([&](auto DTI) {
        constexpr int dt = decltype(DTI)::value;
        Fa2Acc<T, KO + 0 * DT_ + dt>::mfma(vfr[s][dt], pf[0][s]);
        Fa2Acc<T, KO + 1 * DT_ + dt>::mfma(vfr[s][dt], pf[1][s]);
      });
    }
  };

  for (int t = 0; t < n_tiles; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1 < n_tiles);
    if (more) stage_load(t + 1);
    const int kv0 = t * FA_BN;
    bool skip = false, edge = (kv0 + FA_BN > p.Sk);
    if (CAUSAL) {
      skip = (kv0 + p.k_offset) > (wrow0 + 32 * QT - 1 + p.q_offset);
      edge = edge || ((kv0 + FA_BN - 1 + p.k_offset) > (wrow0 + p.q_offset));
    }
    if (!skip) {
      if (edge) process_tile(t, std::true_type{});
      else process_tile(t, std::false_type{});
    }
    if (more) stage_write(cur ^ 1, t + 1);
    __syncthreads();
  }

  // ---- epilogue
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMAs retired before the accumulator file is read
  fa2_for<QT>([&](auto QTI) {
    constexpr int qt = decltype(QTI)::value;
    const float l_tot = l_i[qt] + other_half(l_i[qt]);
    const float inv = (l_tot > 0.f) ? fast_rcp(l_tot) : 0.f;
    if (q_ok[qt]) {
      if (p.lse != nullptr && h == 0) {
        const float lse = (l_tot > 0.f) ? (m_i[qt] + fast_log2(l_tot)) * FA_LN2 : -INFINITY;
        p.lse[((int64_t)b * p.H + head) * p.Sq + qrow[qt]] = lse;
      }
    }
    T* op = (p.o != nullptr) ? ((T*)p.o + b * p.os_b + head * p.os_h + (int64_t)(q_ok[qt] ? qrow[qt] : 0) * p.os_s) : nullptr;
    float* oa = (p.o_acc != nullptr)
                    ? (p.o_acc + (((int64_t)b * p.Sq + (q_ok[qt] ? qrow[qt] : 0)) * p.H + head) * p.D)
                    : nullptr;
    fa2_for<DT_>([&](auto DTI) {
      constexpr int dt = decltype(DTI)::value;
      constexpr int k = qt * DT_ + dt;
      const f32x4_t v[4] = {Fa2AccIO<KO + k>::template read4<0>(), Fa2AccIO<KO + k>::template read4<1>(),
                            Fa2AccIO<KO + k>::template read4<2>(), Fa2AccIO<KO + k>::template read4<3>()};
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * dt + 8 * g + 4 * h;
        if (q_ok[qt] && d0 < p.D) {
          const float x0 = v[g][0] * inv, x1 = v[g][1] * inv, x2 = v[g][2] * inv, x3 = v[g][3] * inv;
          if (op != nullptr) {
            u32x2_t w = {pack2<T>(x0, x1), pack2<T>(x2, x3)};
            *(u32x2_t*)(op + d0) = w;
          }
          if (oa != nullptr) {
            f32x4_t w = {x0, x1, x2, x3};
            *(f32x4_t*)(oa + d0) = w;
          }
        }
      }
    });
  });
  }  // pass
}
