// FlashAttention-3 style prefill kernel for CDNA4 (gfx950), written for 64-wide wavefronts.
//
// Replaces _flash_attention_forward_kernel (reference kernels/triton/flash_attention_kernels.py:38-325)
// and _ring_attention_forward_kernel (kernels/triton/attention_kernels.py:35-202).
//
// Structure (one workgroup = 4 waves = 128 query rows of one (batch, head); KV tile = 64 keys):
//   * Q fragments live in registers for the whole kernel (B operand of the QK^T MFMA).
//   * K/V tiles are staged global -> registers -> LDS, double buffered; the global loads of tile
//     t+1 are issued before the MFMA work of tile t and written to LDS after it (issue-early /
//     write-late), one barrier per tile.
//   * "Swapped" QK^T:  S^T[key][query] = K . Q^T with v_mfma_f32_32x32x16.  The accumulator then
//     has the QUERY on the lane (col = lane&31) and 16 keys in registers, so the row max / row sum
//     are in-register reductions plus ONE cross-half exchange (v_permlane32_swap), and the
//     softmax'd tile is already the B operand of the next MFMA:
//   * O^T[d][query] += V^T[d][key] . P^T[key][query].  P^T needs no LDS round trip or lane
//     shuffle (accumulator registers 8s..8s+7 -> bf16 = the k-step-s fragment); V^T fragments come
//     from a row-major V tile through the hardware transposing read ds_read_b64_tr_b16.
//     The running (m, l) are per-lane scalars in the same layout, so the rescale is lane-local.
//   * LDS images: K rows padded by 16 B (conflict-free ds_read_b128 for any D), V as
//     [key/8][d/32][8][32] sub-tiles (each half-wave's transposed read covers one 256-B bank row).
#pragma once
#include <type_traits>

#include "mio_common.h"

struct FaDev {
  const void* q;
  const void* k;
  const void* v;
  void* o;
  float* lse;
  float* o_acc;
  const void* mask;
  int64_t qs_b, qs_s, qs_h;
  int64_t ks_b, ks_s, ks_h;
  int64_t vs_b, vs_s, vs_h;
  int64_t os_b, os_s, os_h;
  int64_t ms_b, ms_h, ms_q, ms_k;
  int B, Sq, Sk, H, Hkv, D;
  int carry_in, q_offset, k_offset;
  int nqblk, xcd_remap;
  int qgrid;  // workgroups per (batch, head): nqblk, or ceil(nqblk/2) when causal blocks are paired (fa3_fwd2)
  float scale_log2e;  // softmax_scale * log2(e)
  int o_blk;          // o is the [B*Sq, H*D] output in the blocked activation layout (mio_fa3_o_blocked_ok)
  int k_prescaled;    // K already carries softmax_scale * log2(e) (fa3_fwd4_kernel KPRE; mio_fa3_k_prescaled_ok)
};

constexpr int FA_BM = 128;
constexpr int FA_BN = 64;
constexpr float FA_LOG2E = 1.4426950408889634f;
constexpr float FA_LN2 = 0.6931471805599453f;
constexpr float FA_NEG_FILL_LOG2 = -1.0e9f * FA_LOG2E;
constexpr float FA_RESCALE_THR = 6.0f;  // exp2 domain: P <= 64 between rescales

template <int D>
struct FaSmem {
  static constexpr int KROW = D * 2 + 16;       // bytes per K row (padded)
  static constexpr int K_BYTES = FA_BN * KROW;
  static constexpr int V_BYTES = FA_BN * D * 2;
  static constexpr int STAGE = K_BYTES + V_BYTES;
  static constexpr int TOTAL = 2 * STAGE;
};

template <typename T, int D, bool CAUSAL, int MASK>
__global__ __launch_bounds__(256) void fa3_fwd_kernel(const FaDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  using SM = FaSmem<D>;
  constexpr int KS = D / 16;       // k-steps of the QK^T product
  constexpr int DT_ = D / 32;      // 32-row d tiles of O^T
  constexpr int CPR = D / 8;       // 16-byte chunks per K/V row
  constexpr int NLD = D / 32;      // chunks per thread per operand per tile

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  // ---- workgroup -> (batch, head, query block); same (b,h) stays on one XCD for K/V reuse in L2
  int bh, qi;
  {
    const int id = blockIdx.x;
    if (p.xcd_remap) {
      const int xcd = id & 7, slot = id >> 3;
      bh = (slot / p.nqblk) * 8 + xcd;
      qi = slot % p.nqblk;
    } else {
      bh = id / p.nqblk;
      qi = id % p.nqblk;
    }
  }
  const int qblk = CAUSAL ? (p.nqblk - 1 - qi) : qi;  // heaviest causal blocks first
  const int b = bh / p.H, head = bh % p.H;
  const int kvh = head / (p.H / p.Hkv);
  const int q0 = qblk * FA_BM;
  const int qrow = q0 + wave * 32 + r;
  const bool q_ok = qrow < p.Sq;

  // ---- Q fragments (B operand: lane (r,h) holds Q[qrow][16ks + 8h .. +7])
  X8 qf[KS];
  {
    const T* qp = (const T*)p.q + b * p.qs_b + head * p.qs_h + (int64_t)qrow * p.qs_s;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int d0 = 16 * ks + 8 * h;
      u32x4_t raw = {0, 0, 0, 0};
      if (q_ok && d0 < p.D) raw = *(const u32x4_t*)(qp + d0);
      qf[ks] = __builtin_bit_cast(X8, raw);
    }
  }

  // ---- running state
  f32x16_t oacc[DT_];
  float m_i = -INFINITY, l_i = 0.f;
#pragma unroll
  for (int dt = 0; dt < DT_; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) oacc[dt][i] = 0.f;

  if (p.carry_in) {
    if (q_ok) {
      const float lse_in = p.lse[((int64_t)b * p.H + head) * p.Sq + qrow];
      if (lse_in != -INFINITY) {
        m_i = lse_in * FA_LOG2E;
        l_i = (h == 0) ? 1.f : 0.f;
      }
      const float* oa = p.o_acc + (((int64_t)b * p.Sq + qrow) * p.H + head) * p.D;
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d0 = 32 * dt + 8 * g + 4 * h;
          if (d0 < p.D) {
            const f32x4_t x = *(const f32x4_t*)(oa + d0);
            oacc[dt][4 * g + 0] = x[0];
            oacc[dt][4 * g + 1] = x[1];
            oacc[dt][4 * g + 2] = x[2];
            oacc[dt][4 * g + 3] = x[3];
          }
        }
    }
  }

  // Consume the Q fragments (and the carried state) here: hipcc then places its wait for those loads HERE instead
  // of at their first use inside the tile loop, where a vmcnt(0) would also drain the K/V prefetch of every tile.
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
  if (p.carry_in) {
#pragma unroll
    for (int dt = 0; dt < DT_; ++dt)
#pragma unroll
      for (int i = 0; i < 16; i += 4) asm volatile("" ::"v"(oacc[dt][i]));
    asm volatile("" ::"v"(m_i));
  }

  // ---- number of KV tiles this workgroup visits
  int n_tiles;
  if (CAUSAL && MASK == 0) {
    int kmax = q0 + FA_BM - 1 + p.q_offset - p.k_offset;
    if (kmax > p.Sk - 1) kmax = p.Sk - 1;
    n_tiles = kmax < 0 ? 0 : kmax / FA_BN + 1;
  } else {
    n_tiles = (p.Sk + FA_BN - 1) / FA_BN;
  }

  // ---- staging helpers
  const T* kbase = (const T*)p.k + b * p.ks_b + kvh * p.ks_h;
  const T* vbase = (const T*)p.v + b * p.vs_b + kvh * p.vs_h;
  u32x4_t kreg[NLD], vreg[NLD];

  // Loads are UNCONDITIONAL (row / chunk clamped to a valid address): a load under `if (ok)` merges with the
  // zero default at the join, which is a use of the loaded register, so hipcc waits for the load right there --
  // at the top of the tile, with its whole HBM/L2 latency exposed.  Out-of-range rows / chunks are zeroed when
  // the registers are written to LDS (after the tile's MFMA work), and only for tiles that need it.
  const int d_chunks = p.D >> 3;  // valid 16-byte chunks per row
  auto stage_load = [&](int tile) {
    const int kv0 = tile * FA_BN;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
      const int i = tid + 256 * j;
      const int row = i / CPR, c = i % CPR;
      int kv = kv0 + row;
      kv = kv < p.Sk ? kv : p.Sk - 1;
      const int cc = c < d_chunks ? c : d_chunks - 1;
      // asm loads: invisible to hipcc's s_waitcnt insertion, which would otherwise drain them (vmcnt(0)) at the
      // first basic-block join of the tile body, i.e. before the first MFMA.  Retired by fa_wait_loads() below.
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(kreg[j]) : "v"(kbase + (int64_t)kv * p.ks_s + 8 * cc) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(vreg[j]) : "v"(vbase + (int64_t)kv * p.vs_s + 8 * cc) : "memory");
    }
  };
  auto wait_loads = [&]() {  // every staged register is an in/out operand: no consumer can be scheduled above this
#pragma unroll
    for (int j = 0; j < NLD; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(kreg[j]), "+v"(vreg[j])::"memory");
  };
  auto stage_write = [&](int buf, int tile) {
    wait_loads();
    char* kb = smem + buf * SM::STAGE;
    char* vb = kb + SM::K_BYTES;
    const int kv0 = tile * FA_BN;
    const bool partial = (kv0 + FA_BN > p.Sk) || (d_chunks != CPR);  // wave-uniform
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

  // per-lane LDS read offsets
  const int k_rd = r * SM::KROW + 16 * h;                         // + 32*KROW*tt + 32*ks
  const int g16 = lane >> 4, i16 = lane & 15;                     // transposed-read group / lane in group
  const int v_rd = (4 * h + (i16 >> 2)) * 64 + 32 * (g16 & 1) + 8 * (i16 & 3);  // + ((2s+u)*DT_ + dt)*512
  const int q_pos = qrow + p.q_offset;
  const float c2 = p.scale_log2e;

  // One KV tile for this wave.  EDGE = the tile needs per-element masking (causal diagonal, keys past Sk, or any
  // user mask); interior tiles (the vast majority) compile to the bare online-softmax update with no compares.
  auto process_tile = [&](int t, auto EDGE_) {
    constexpr bool EDGE = decltype(EDGE_)::value;
    const int cur = t & 1;
    const int kv0 = t * FA_BN;
    const char* kb = smem + cur * SM::STAGE;
    const char* vb = kb + SM::K_BYTES;

    // ---- S^T = K . Q^T   (two 32-key tiles).  Everything below works on the two accumulator tuples in place
    // (sc[tt][reg]); copying them into a scalar array costs 32 v_mov_b64 per tile in a VALU-bound loop.
    f32x16_t sc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sc[0][i] = 0.f;
      sc[1][i] = 0.f;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const X8 a0 = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd + 32 * ks));
      const X8 a1 = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd + 32 * SM::KROW + 32 * ks));
      sc[0] = DT<T>::mfma32(a0, qf[ks], sc[0]);
      sc[1] = DT<T>::mfma32(a1, qf[ks], sc[1]);
    }

    // ---- V^T fragments for the whole tile, issued NOW: the transposed LDS reads fly under the softmax VALU work
    // below instead of each stalling the P.V MFMA it feeds (hipcc otherwise sinks every read to just before its use)
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

    // ---- masks (edge tiles only) and the row maximum, in the exp2 domain
    float mx;
    if constexpr (!EDGE) {
      mx = fmaxf(fmaxf(sc[0][0], sc[0][1]), sc[0][2]);
#pragma unroll
      for (int i = 3; i + 1 < 16; i += 2) mx = fmaxf(fmaxf(mx, sc[0][i]), sc[0][i + 1]);
      mx = fmaxf(fmaxf(mx, sc[0][15]), sc[1][0]);
#pragma unroll
      for (int i = 1; i + 1 < 16; i += 2) mx = fmaxf(fmaxf(mx, sc[1][i]), sc[1][i + 1]);
      mx = fmaxf(mx, sc[1][15]) * c2;
    } else {
      const int64_t mrow = (MASK == 0) ? 0
                                       : ((int64_t)b * p.ms_b + (int64_t)head * p.ms_h + (int64_t)(q_ok ? qrow : 0) * p.ms_q);
      mx = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int kv = kv0 + 32 * tt + (i & 3) + 8 * (i >> 2) + 4 * h;
          float tv = sc[tt][i] * c2;
          if (MASK == 0) {
            if ((kv >= p.Sk) || (CAUSAL && (kv + p.k_offset > q_pos))) tv = -INFINITY;
          } else {
            if (CAUSAL && (kv + p.k_offset > q_pos)) tv = FA_NEG_FILL_LOG2;
            if (kv < p.Sk) {
              if (MASK == MIO_MASK_KEEP_U8) {
                const uint8_t keep = ((const uint8_t*)p.mask)[mrow + (int64_t)kv * p.ms_k];
                if (!keep) tv = FA_NEG_FILL_LOG2;
              } else if (MASK == MIO_MASK_ADD_F32) {
                tv += ((const float*)p.mask)[mrow + (int64_t)kv * p.ms_k] * FA_LOG2E;
              }
            } else {
              tv = -INFINITY;
            }
          }
          sc[tt][i] = tv;
          mx = fmaxf(mx, tv);
        }
    }
    mx = fmaxf(mx, other_half(mx));

    // ---- online softmax update (flash_attention_kernels.py:276-298 with exp -> exp2).  The running maximum
    // is only raised -- and O rescaled -- when some row of the wave outgrows it by more than FA_RESCALE_THR
    // (exp2 domain): until then P is computed against the stale maximum and is at most 2^THR, exact in the
    // fp32 row sum and harmless in bf16/fp16 (their relative precision does not depend on the scale).
    float m_sub;
    if (__builtin_amdgcn_ballot_w64(mx > m_i + FA_RESCALE_THR) != 0) {
      const float m_new = fmaxf(m_i, mx);
      m_sub = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = fast_exp2(m_i - m_sub);
      l_i *= alpha;
      m_i = m_new;
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt)
#pragma unroll
        for (int i = 0; i < 16; ++i) oacc[dt][i] *= alpha;
    } else {
      m_sub = m_i;  // finite here: a row with m_i == -inf always takes the branch above
    }
    // P = exp2(score - m), row sum, and the P^T fragments: k-step s (16 keys) = registers 8(s&1)..+7 of tile s>>1
    float rs0 = 0.f, rs1 = 0.f;
    X8 pf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = sc[s >> 1][8 * (s & 1) + j];
        e[j] = EDGE ? fast_exp2(x - m_sub) : fast_exp2(__builtin_fmaf(x, c2, -m_sub));
      }
      rs0 += (e[0] + e[1]) + (e[2] + e[3]);
      rs1 += (e[4] + e[5]) + (e[6] + e[7]);
      u32x4_t w;
      w[0] = pack2<T>(e[0], e[1]);
      w[1] = pack2<T>(e[2], e[3]);
      w[2] = pack2<T>(e[4], e[5]);
      w[3] = pack2<T>(e[6], e[7]);
      pf[s] = __builtin_bit_cast(X8, w);
    }
    l_i += rs0 + rs1;

    // ---- O^T += V^T . P^T
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt) oacc[dt] = DT<T>::mfma32(vfr[s][dt], pf[s], oacc[dt]);
  };

  for (int t = 0; t < n_tiles; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1 < n_tiles);
    if (more) stage_load(t + 1);

    const int kv0 = t * FA_BN;
    bool skip = false, edge = (MASK != 0) || (kv0 + FA_BN > p.Sk);
    if (CAUSAL && MASK == 0) {
      // every key of this tile is in the future of every row of this wave
      skip = (kv0 + p.k_offset) > (q0 + wave * 32 + 31 + p.q_offset);
      edge = edge || ((kv0 + FA_BN - 1 + p.k_offset) > (q0 + wave * 32 + p.q_offset));
    }
    if (!skip) {
      if (edge) process_tile(t, std::true_type{});
      else process_tile(t, std::false_type{});
    }

    if (more) stage_write(cur ^ 1, t + 1);
    __syncthreads();
  }

  // ---- epilogue: normalise, write O (and the fp32 carry state / lse)
  const float l_tot = l_i + other_half(l_i);
  const float inv = (l_tot > 0.f) ? fast_rcp(l_tot) : 0.f;
  if (q_ok) {
    if (p.lse != nullptr && h == 0) {
      const float lse = (l_tot > 0.f) ? (m_i + fast_log2(l_tot)) * FA_LN2 : -INFINITY;
      p.lse[((int64_t)b * p.H + head) * p.Sq + qrow] = lse;
    }
    T* op = (p.o != nullptr) ? ((T*)p.o + b * p.os_b + head * p.os_h + (int64_t)qrow * p.os_s) : nullptr;
    float* oa = (p.o_acc != nullptr) ? (p.o_acc + (((int64_t)b * p.Sq + qrow) * p.H + head) * p.D) : nullptr;
#pragma unroll
    for (int dt = 0; dt < DT_; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * dt + 8 * g + 4 * h;
        if (d0 < p.D) {
          const float x0 = oacc[dt][4 * g + 0] * inv, x1 = oacc[dt][4 * g + 1] * inv;
          const float x2 = oacc[dt][4 * g + 2] * inv, x3 = oacc[dt][4 * g + 3] * inv;
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
  }
}

// Host launcher for one (dtype, padded D); defined per translation unit (fa3_fwd_inst.hip).
template <typename T, int D>
int fa3_launch(const FaDev& p, int causal, int mask_kind, hipStream_t stream);
