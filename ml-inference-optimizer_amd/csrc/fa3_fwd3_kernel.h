// FlashAttention forward, third structure (head dim 64, no user mask): software-pipelined across KV tiles.
//
// fa3_fwd2_kernel (one wave per SIMD, 64 query rows per wave) runs each tile as QK^T -> softmax -> PV in sequence, so
// the matrix core idles during the softmax and the vector ALU idles during the MFMAs: 2375 cycles per tile against
// 1024 of MFMA work (tools/fa_ab.py, rocprofv3 SQ counters).  Here, per wave and tile t:
//   phase 1   S(t+1) = K(t+1) . Q^T                  16 MFMAs   ||   P(t) = exp2(S(t))                  64 v_exp + 32 v_cvt_pk
//   phase 2   O^T += V(t)^T . P(t)^T, L += ones . P(t)^T  24 MFMAs   ||   S(t+1) := S(t+1)*c - ref, row max    64 v_fma + 32 v_max3, DMA issue
// with S double-buffered in VGPRs.  What makes the two sides balance:
//   * the scale-and-subtract of tile t+1 (exp2 domain, against the CURRENT reference) and its row maximum ride in the
//     vector slack of the PV phase, so the exp phase is exp + convert only;
//   * row sums come from the matrix core: L^T += ones(32x16) . P^T into a 32x32 accumulator whose registers all hold
//     the row sum (no v_add per element);
//   * every MFMA is an asm statement on asm-owned registers: O^T, L, Q, ones live in the accumulator file (MFMA
//     A/B operands may come from there), which leaves the 256 architectural VGPRs to S (128), P (32) and the K / V
//     fragments -- and every phase is a fixed sequence of micro-steps (MFMA + its share of the vector work) pinned
//     with sched_barrier(0).  (Compiler-visible "+a"/"a" operands were tried: accumulator tuples get copied at branch
//     joins right behind an asm MFMA whose latency the compiler does not know -- lost updates -- and "a" inputs are
//     re-copied from VGPRs before every use.)
// The reference only moves when a row outgrows it by 2^FA_RESCALE_THR (flash_attention_kernels.py:276-298 is the
// algorithm: online softmax with running max / sum; exp -> exp2).  K/V tiles go global -> LDS by DMA through a 4-stage
// ring (K(t+1) and V(t) are read while tiles t+2 and t+3 are in flight: a tile has two iterations to land), one
// barrier per tile.
#pragma once
#include "fa3_fwd2_kernel.h"

constexpr int FA3_BM = 256;     // query rows per workgroup (4 waves x 64)
constexpr int FA3_STAGES = 4;
// Accumulator-file map (all asm-owned, at the TOP of the file: the allocator hands out a0, a1, ... for its own
// values first; tests/test_host_logic.py verifies after every build that no compiler-generated instruction touches
// Fa3Map<D>::A_Q and up):
//   L tile qt = Fa2Acc tile 14 + qt = a[224:255]             O^T tile qt*DT+dt = Fa2Acc tile T_O + qt*DT+dt, right below
//   ones (A operand, all ones) = the 4 registers below O^T    Q fragment (qt, ks) = a[A_Q + 4 KS qt + 4 ks : +3], below
template <int D>
struct Fa3Map {
  static constexpr int KS = D / 16, DT = D / 32;
  static constexpr int T_L = 14, T_O = 14 - 2 * DT;
  static constexpr int A_ONES = 16 * T_O - 4, A_Q = A_ONES - 8 * KS;  // D = 64: 156 / 124, 96: 124 / 76, 128: 92 / 28
  static constexpr int CPRK = D / 8 + 1;          // 16-byte chunks per padded K row
  static constexpr int KU = CPRK, VU = D / 8;     // 1-KiB DMA units of the K / V tile of a stage
  static constexpr int NU = KU + VU, UPW = (NU + 3) / 4;  // units per stage, per wave
};

// write one accumulator register a[R] (R >= Fa3Map<D>::A_Q: the asm-owned range; the MFMA statements' clobber lists
// make the whole file part of the kernel's allocation)
template <int R>
struct Fa3AW {
  static __device__ __forceinline__ void w(uint32_t v) {
    asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(v), "n"(R));
  }
};

template <typename T>
struct Fa3Ops;
#define FA3_OPS(TY, SUF)                                                                                              \
  template <>                                                                                                         \
  struct Fa3Ops<TY> {                                                                                                 \
    using X8 = typename DT<TY>::x8;                                                                                   \
    /* S (+)= K fragment (VGPR) . Q fragment a[R:R+3]; FIRST: S = ... (C = 0) */                                     \
    template <int R, bool FIRST>                                                                                      \
    static __device__ __forceinline__ void qk(f32x16_t& acc, const X8& kf) {                                          \
      if constexpr (FIRST)                                                                                            \
        asm volatile("v_mfma_f32_32x32x16_" SUF " %0, %1, a[%2:%3], 0" : "=v"(acc) : "v"(kf), "n"(R), "n"(R + 3));    \
      else                                                                                                            \
        asm volatile("v_mfma_f32_32x32x16_" SUF " %0, %1, a[%2:%3], %0" : "+v"(acc) : "v"(kf), "n"(R), "n"(R + 3));   \
    }                                                                                                                 \
    /* first k-step with the running reference as the C operand: S = K fragment . Q fragment + c (KPRE) */           \
    template <int R>                                                                                                  \
    static __device__ __forceinline__ void qk_c(f32x16_t& acc, const X8& kf, const f32x16_t& c) {                    \
      asm volatile("v_mfma_f32_32x32x16_" SUF " %0, %1, a[%2:%3], %4" : "=&v"(acc) : "v"(kf), "n"(R), "n"(R + 3), "v"(c)); \
    }                                                                                                                 \
    template <int RO>                                                                                                 \
    static __device__ __forceinline__ void lsum0(const X8& pf) {                                                      \
      asm volatile("v_mfma_f32_32x32x16_" SUF " a[224:239], a[%1:%2], %0, a[224:239]"                                 \
                   :                                                                                                  \
                   : "v"(pf), "n"(RO), "n"(RO + 3)                                                                    \
                   : "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235",  \
                     "a236", "a237", "a238", "a239");                                                                 \
    }                                                                                                                 \
    template <int RO>                                                                                                 \
    static __device__ __forceinline__ void lsum1(const X8& pf) {                                                      \
      asm volatile("v_mfma_f32_32x32x16_" SUF " a[240:255], a[%1:%2], %0, a[240:255]"                                 \
                   :                                                                                                  \
                   : "v"(pf), "n"(RO), "n"(RO + 3)                                                                    \
                   : "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251",  \
                     "a252", "a253", "a254", "a255");                                                                 \
    }                                                                                                                 \
  };
FA3_OPS(__bf16, "bf16")
FA3_OPS(_Float16, "f16")
#undef FA3_OPS

// Phase 2 gives every micro-step (= one MFMA, 32 matrix-pipe cycles of which the issue port is held for 8) exactly ONE
// piece of side work, so that the step stays inside the MFMA's shadow: the V fragment read of the next k-step (3 steps,
// fixed by the fragment ring), one of the 16 scale / max groups (4 fma + 2 max3 = 24 issue cycles), or one DMA unit.
// Stacked on the same step (as they were: scale / max on steps 2..17, reads on 1 / 7 / 13, DMA on every fourth) the
// dense steps overflow the shadow and the matrix pipe waits.  role(j): 0 nothing, 1 + i scale / max group i,
// 32 + k DMA unit k.  Order: a DMA unit first, then four groups and a unit alternately, the remaining units last.
template <int NS2, int PS, int UPW>
struct Fa3P2Role {
  static constexpr int role(int j) {
    int n_v = 0, n_d = 0;
    for (int jj = 0; jj < NS2; ++jj) {
      const int s = jj / PS, m = jj % PS;
      const bool rd = (m == 1 && s + 1 < 4);
      int r = 0;
      if (!rd) {
        const bool want_d = n_d < UPW && (n_d == 0 || n_v >= 4 * n_d || n_v == 16);
        if (want_d) {
          r = 32 + n_d;
          ++n_d;
        } else if (n_v < 16) {
          r = 1 + n_v;
          ++n_v;
        }
      }
      if (jj == j) return r;
    }
    return 0;
  }
  static constexpr int count(int lo, int hi) {  // steps with a role in [lo, hi)
    int n = 0;
    for (int jj = 0; jj < NS2; ++jj) n += (role(jj) >= lo && role(jj) < hi) ? 1 : 0;
    return n;
  }
  static_assert(count(1, 17) == 16 && count(32, 32 + UPW) == UPW, "phase 2: not every scale/max group or DMA unit has a step");
};

// ABL (diagnostic build only, timing-only ablations with WRONG results -- what each piece of the tile loop costs):
//   1 no scale-and-subtract / max in phase 2     2 no exp (P = converted S)     4 no row-sum MFMAs
//   8 no DMA issue in the tile loop              16 no reference test / update  32 no edge masks
// KPRE (FaDev::k_prescaled; see fa3_fwd4_kernel.h): K carries softmax_scale * log2(e).  The reference enters the QK^T
// product as the C operand of its first k-step (one 16-register tuple per query sub-tile), phase 2 loses its 64 v_fma +
// 32 v_max3 per tile, and the rescale test is bit 14 of the OR of the tile's packed P words (move_ref below).
template <typename T>
struct Fa3Margin { static constexpr float value = 5.0f; };
template <>
struct Fa3Margin<_Float16> { static constexpr float value = 2.0f; };

template <typename T, int D, bool CAUSAL, bool STAMP = false, int ABL = 0, bool KPRE = false>
__global__ __launch_bounds__(256) void fa3_fwd3_kernel(const FaDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  using OPS = Fa3Ops<T>;
  using MAP = Fa3Map<D>;
  constexpr int QT = 2, KS = MAP::KS, DT_ = MAP::DT, UPW = MAP::UPW;
  constexpr int FA3_T_O = MAP::T_O, FA3_T_L = MAP::T_L, FA3_A_Q = MAP::A_Q, FA3_A_ONES = MAP::A_ONES;
  using SM = FaSmem<D>;
#define IC(N) std::integral_constant<int, (N)> {}

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
      bh = (slot / p.qgrid) * 8 + xcd;
      qi = slot % p.qgrid;
    } else {
      bh = id / p.qgrid;
      qi = id % p.qgrid;
    }
  }
  const int b = bh / p.H, head = bh % p.H;
  const int kvh = head / (p.H / p.Hkv);

  // constant MFMA operand: ones = A operand of all ones (row sums)
  {
    const uint32_t o2 = pack2<T>(1.f, 1.f);
    Fa3AW<FA3_A_ONES + 0>::w(o2); Fa3AW<FA3_A_ONES + 1>::w(o2); Fa3AW<FA3_A_ONES + 2>::w(o2); Fa3AW<FA3_A_ONES + 3>::w(o2);
  }
  const float c2 = p.scale_log2e;

  // per-lane LDS read offsets (layouts: fa3_fwd_kernel.h)
  const int k_rd = r * SM::KROW + 16 * h;
  const int g16 = lane >> 4, i16 = lane & 15;
  const int v_rd = (4 * h + (i16 >> 2)) * 64 + 32 * (g16 & 1) + 8 * (i16 & 3);

  // Causal: this workgroup handles query block nqblk-1-qi (heavy) and then block qi (light): equal work for every
  // workgroup.  Non-causal: one block.
  const int npass = (CAUSAL && (p.nqblk - 1 - qi) != qi) ? 2 : 1;
  unsigned long long st_all[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // diagnostic build: sums over both passes
  if constexpr (STAMP) st_all[7] = __builtin_amdgcn_s_memtime();
  for (int pass = 0; pass < npass; ++pass) {
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0, pt4 = 0;
    if constexpr (STAMP) pt0 = __builtin_amdgcn_s_memtime();
    const int qblk = CAUSAL ? (pass == 0 ? p.nqblk - 1 - qi : qi) : qi;
    const int q0 = qblk * FA3_BM;
    const int wrow0 = q0 + wave * (32 * QT);  // first query row of this wave
    int qrow[QT];
    bool q_ok[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      qrow[qt] = wrow0 + 32 * qt + r;
      q_ok[qt] = qrow[qt] < p.Sq;
    }

    // ---- running state per query sub-tile: m_i = reference the probabilities are taken against (exp2 domain; -inf =
    // no finite score yet, the reference is then 0), negref = -reference as used by the scale-and-subtract
    float m_i[QT], negref[QT], lcarry[QT];
    // KPRE state per query sub-tile: ref = the reference subtracted through the C operand (0 while the row is fresh),
    // nref16 = -ref in all 16 registers; orw = OR of the tile's packed P words; fresh_any is wave-uniform
    float ref[QT] = {0.f, 0.f};
    bool fresh[QT] = {true, true};
    bool fresh_any = true;
    uint32_t orw = 0u;
    f32x16_t nref16[QT];
    if constexpr (KPRE) {
#pragma unroll
      for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int i = 0; i < 16; ++i) nref16[qt][i] = 0.f;
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      m_i[qt] = -INFINITY;
      negref[qt] = 0.f;
      lcarry[qt] = 0.f;
      if (p.carry_in && q_ok[qt]) {
        const float lse_in = p.lse[((int64_t)b * p.H + head) * p.Sq + qrow[qt]];
        if (lse_in != -INFINITY) {
          m_i[qt] = lse_in * FA_LOG2E;
          negref[qt] = -m_i[qt];
          lcarry[qt] = 1.f;
        }
      }
    }
    // ---- tiles: the workgroup walks n_tiles (barriers, staging); this wave computes the first n_w of them
    int n_tiles, n_w;
    if (CAUSAL) {
      int kmax = q0 + FA3_BM - 1 + p.q_offset - p.k_offset;
      if (kmax > p.Sk - 1) kmax = p.Sk - 1;
      n_tiles = kmax < 0 ? 0 : kmax / FA_BN + 1;
      int kw = wrow0 + 32 * QT - 1 + p.q_offset - p.k_offset;
      if (kw > p.Sk - 1) kw = p.Sk - 1;
      n_w = kw < 0 ? 0 : kw / FA_BN + 1;
    } else {
      n_tiles = (p.Sk + FA_BN - 1) / FA_BN;
      n_w = n_tiles;
    }
    const int n_tiles_dma = n_tiles > 0 ? n_tiles : 1;
    // last key visible to each query row of this lane, and the first tile of this wave that needs masks
    int klim[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      klim[qt] = p.Sk - 1;
      if (CAUSAL) {
        const int c = qrow[qt] + p.q_offset - p.k_offset;
        klim[qt] = c < klim[qt] ? c : klim[qt];
      }
    }
    // tile t is an edge tile iff its last key is past the limit of the wave's FIRST row (limits grow with the row)
    int lim0 = p.Sk - 1;
    if (CAUSAL) {
      const int c = wrow0 + p.q_offset - p.k_offset;
      lim0 = c < lim0 ? c : lim0;
    }
    const int first_edge = (lim0 + 1) / FA_BN;  // tiles t >= first_edge contain a key > lim0  (lim0 + 1 >= 0 here
                                                // whenever n_w > 0 ... negative limits give first_edge <= 0: all edge)
    auto is_edge = [&](int t) -> bool { return t >= first_edge; };

    // ---- K/V staging: global -> LDS by DMA (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB of lane-linear LDS per
    // wave-instruction, no staging registers).  A stage is 17 such units: 0..8 the K tile (64 rows x 144 B: 9 chunks
    // per row, the 9th is padding), 9..16 the V tile ([key/8][d/32][8][32] sub-tiles of 512 B); the per-lane SOURCE
    // address realises the layout.  Every wave moves exactly UPW = 5 units per tile (slot assignment below; spare
    // slots repeat the wave's first unit), which is what the counted waits assume.  Rows past Sk and chunks
    // past D are clamped to valid data instead of zeroed: such keys are masked to -inf (edge tiles) and Q~ is zero
    // past D, so the values only need to be finite.
    const T* kbase = (const T*)p.k + b * p.ks_b + kvh * p.ks_h;
    const T* vbase = (const T*)p.v + b * p.vs_b + kvh * p.vs_h;
    const int d_chunks = p.D >> 3;
    // Slot i of a wave: slots 0 .. KSL-1 carry K units (wave + 4i; the spare ones of the last K slot repeat the wave's
    // first unit), the rest V units -- K or V is a compile-time property of the slot, and so is its LDS offset from
    // the wave's first unit (bar the repeat).  Per-lane source offsets (row * stride + chunk) are fixed for the whole
    // kernel; only the last tile of the sequence can be partial, and it gets its own set with the rows clamped to the
    // last valid one.  What is left per unit in the tile loop: one select, m0, the load (was ~12 instructions: on a
    // one-wave-per-SIMD kernel every scalar instruction is issue time, and phase 2 is issue-bound).
    constexpr int KSL = (MAP::KU + 3) / 4;
    const int ks2 = (int)p.ks_s * 2, vs2 = (int)p.vs_s * 2;  // row strides in bytes
    const int last_tile = (p.Sk - 1) >> 6, last_row = (p.Sk - 1) & (FA_BN - 1);
    int st_off[UPW], st_offl[UPW];  // per slot: byte offset of this lane's 16-B chunk from the tile's first row (full / last tile)
#pragma unroll
    for (int i = 0; i < UPW; ++i) {
      int u, row, c;
      if (i < KSL) {
        u = (wave + 4 * i < MAP::KU) ? wave + 4 * i : wave;
        const int u16 = 64 * u + lane;
        row = u16 / MAP::CPRK;
        c = u16 % MAP::CPRK;
      } else {
        u = wave + 4 * (i - KSL);
        const int blk = 2 * u + (lane >> 5);
        row = 8 * (blk / DT_) + ((lane & 31) >> 2);
        c = 4 * (blk % DT_) + (lane & 3);
      }
      c = c < d_chunks ? c : d_chunks - 1;
      const int rowl = row < last_row ? row : last_row;
      st_off[i] = row * (i < KSL ? ks2 : vs2) + 16 * c;
      st_offl[i] = rowl * (i < KSL ? ks2 : vs2) + 16 * c;
    }
    const int kl_imm = (wave + 4 * (KSL - 1) < MAP::KU) ? 4096 * (KSL - 1) : 0;  // LDS offset of the last K slot's unit
    // one DMA unit of tile `tile` (clamped to the last tile: a run past the end re-fetches valid data into a dead
    // stage, which keeps the number of loads per iteration -- and the counted waits -- the same for every iteration)
    const char* dma_kb = nullptr;  // scalar: first row of the K / V tile being fetched (set by dma_tile_base)
    const char* dma_vb = nullptr;
    bool dma_is_last = false;      // that tile is the (possibly partial) last one of the sequence
    uint32_t dma_lds = 0;          // LDS address of this wave's first unit in the stage being filled
    auto dma_tile_base = [&](int tile_) {
      const int tile = tile_ < n_tiles_dma ? tile_ : n_tiles_dma - 1;
      const int kv0 = tile * FA_BN;
      // (32-bit offsets: the launcher sends sequences whose K / V rows span 4 GiB or more to fa3_fwd_kernel; 8
      //  scalar instructions per 64-bit multiply-add otherwise, and scalar instructions are issue time here)
      const uint32_t ko = __builtin_amdgcn_readfirstlane((uint32_t)kv0 * (uint32_t)ks2);
      const uint32_t vo = __builtin_amdgcn_readfirstlane((uint32_t)kv0 * (uint32_t)vs2);
      dma_kb = (const char*)kbase + ko;
      dma_vb = (const char*)vbase + vo;
      dma_is_last = (tile == last_tile);
      dma_lds = (uint32_t)(size_t)((MIO_LDS char*)(smem + (tile_ & (FA3_STAGES - 1)) * SM::STAGE)) + 1024 * wave;
    };
    auto dma_unit = [&](auto I_, int) {
      constexpr int i = decltype(I_)::value;
      const int off = dma_is_last ? st_offl[i] : st_off[i];
      const char* base = (i < KSL) ? dma_kb : dma_vb;
      // asm: invisible to the compiler's wait-count insertion, which otherwise drains the DMA (vmcnt(0)) in front of
      // the next LDS read it cannot prove disjoint -- the V fragments of the tile being computed
      if constexpr (i == KSL - 1) {
        asm volatile("s_add_i32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                     :
                     : "s"(dma_lds), "s"(kl_imm), "v"(off), "s"(base)
                     : "memory", "m0", "scc");  // s_add writes SCC
      } else {
        constexpr int imm = i < KSL ? 4096 * i : 1024 * MAP::KU + 4096 * (i - KSL);
        asm volatile("s_add_i32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                     :
                     : "s"(dma_lds), "n"(imm), "v"(off), "s"(base)
                     : "memory", "m0", "scc");  // s_add writes SCC
      }
    };
    auto stage_dma = [&](int tile) {
      dma_tile_base(tile);
      fa2_for<UPW>([&](auto I_) { dma_unit(I_, tile); });
    };

    if constexpr (STAMP) pt1 = __builtin_amdgcn_s_memtime();
    __syncthreads();  // the previous pass is done with every LDS stage
    stage_dma(0);
    stage_dma(1);
    stage_dma(2);
    // ---- Q rows (lane (r,h) holds Q[row][16ks + 8h .. +7]), requested right behind the first tiles: loads only, from
    // clamped addresses, all issued before anything waits (rows past Sq and chunks past D are zeroed when the fragments
    // are committed: a conditional next to the load becomes a branch with a vmcnt(0) in it, and an accumulator-file
    // write per load serialises the eight latencies -- 5k cycles per pass measured, tools/fa_stamps.py)
    u32x4_t qraw[QT * KS];
    fa2_for<QT * KS>([&](auto QK_) {
      constexpr int qt = decltype(QK_)::value / KS, ks = decltype(QK_)::value % KS;
      const T* qp = (const T*)p.q + b * p.qs_b + head * p.qs_h + (int64_t)(q_ok[qt] ? qrow[qt] : 0) * p.qs_s;
      const int d0 = 16 * ks + 8 * h;
      qraw[qt * KS + ks] = *(const u32x4_t*)(qp + (d0 < p.D ? d0 : 0));
    });
    // ---- O^T and L start from zero (or the carried state): set up under the latency of those requests
    fa2_for<QT>([&](auto QTI) {
      constexpr int qt = decltype(QTI)::value;
      const f32x4_t lv = {lcarry[qt], lcarry[qt], lcarry[qt], lcarry[qt]};
      Fa2AccIO<FA3_T_L + qt>::template write4<0>(lv);
      Fa2AccIO<FA3_T_L + qt>::template write4<1>(lv);
      Fa2AccIO<FA3_T_L + qt>::template write4<2>(lv);
      Fa2AccIO<FA3_T_L + qt>::template write4<3>(lv);
    });
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
      Fa2AccIO<FA3_T_O + k>::template write4<0>(z[0]);
      Fa2AccIO<FA3_T_O + k>::template write4<1>(z[1]);
      Fa2AccIO<FA3_T_O + k>::template write4<2>(z[2]);
      Fa2AccIO<FA3_T_O + k>::template write4<3>(z[3]);
    });

    // ---- Q as MFMA B fragments in the accumulator file
    fa2_for<QT * KS>([&](auto QK_) {
      constexpr int qt = decltype(QK_)::value / KS, ks = decltype(QK_)::value % KS;
      constexpr int R = FA3_A_Q + 4 * KS * qt + 4 * ks;
      u32x4_t raw = qraw[qt * KS + ks];
      const uint32_t keep = (q_ok[qt] && 16 * ks + 8 * h < p.D) ? 0xffffffffu : 0u;
      raw[0] &= keep; raw[1] &= keep; raw[2] &= keep; raw[3] &= keep;
      Fa3AW<R + 0>::w(raw[0]); Fa3AW<R + 1>::w(raw[1]); Fa3AW<R + 2>::w(raw[2]); Fa3AW<R + 3>::w(raw[3]);
    });
    asm volatile("s_nop 7" ::: "memory");  // accumulator-file writes settle before the first MFMA reads them
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");  // tiles 0 and 1 have landed, tile 2 may still fly
    __syncthreads();
    if constexpr (STAMP) pt2 = __builtin_amdgcn_s_memtime();

    f32x16_t S[2][QT][2];   // score tiles: buffer (t & 1), query sub-tile, 32-key half
    u32x4_t pfw[QT][4];     // P^T fragments of the tile in phase 2: k-step s (16 keys)
    X8 vfr[2][DT_];         // V^T fragments, ring over k-steps
    float mx[QT];

    auto read_v = [&](const char* vb, auto S_) {
      constexpr int s = decltype(S_)::value;
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt) {
        const X4 lo = DT<T>::ds_read_tr(vb + v_rd + ((2 * s + 0) * DT_ + dt) * 512);
        const X4 hi = DT<T>::ds_read_tr(vb + v_rd + ((2 * s + 1) * DT_ + dt) * 512);
        X8 f;
        f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
        f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
        vfr[s & 1][dt] = f;
      }
    };

    // ---- phase 1: S[cb ^ 1] = RAW scores of the next tile (K at kb)  ||  P = exp2(S[cb]), row sums of P on the matrix
    // core as soon as a fragment is complete (+ the first V fragments from vb).  The QK^T half always runs (after the
    // last tile of the wave it produces scores nobody reads): a run-time "has next" would put the shared vector half
    // under two branches, and the compiler then hoists it out of the pinned micro-steps.
    auto phase1 = [&](auto CB_, auto DO_EXP_, const char* kb, const char* vb) {
      constexpr int cb = decltype(CB_)::value, nb = cb ^ 1;
      constexpr bool DO_EXP = decltype(DO_EXP_)::value != 0;
      X8 kf[2][2];
      auto read_k = [&](auto KS_) {
        constexpr int ks = decltype(KS_)::value;
        kf[ks & 1][0] = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd + 32 * ks));
        kf[ks & 1][1] = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd + 32 * SM::KROW + 32 * ks));
      };
      read_k(IC(0));
      read_k(IC(1));
      constexpr int NS1 = 4 * KS;  // one QK^T MFMA per step; the 16 exp half-units are spread evenly over the steps
      auto step = [&](auto J_) {
        constexpr int j = decltype(J_)::value;
        {
          constexpr int ks = j >> 2, qt = (j & 3) >> 1, tt = j & 1;
          if constexpr (KPRE && ks == 0) OPS::template qk_c<FA3_A_Q + 4 * KS * qt>(S[nb][qt][tt], kf[0][tt], nref16[qt]);
          else OPS::template qk<FA3_A_Q + 4 * KS * qt + 4 * ks, ks == 0>(S[nb][qt][tt], kf[ks & 1][tt]);
          if constexpr ((j & 3) == 3 && ks + 2 < KS) read_k(IC(ks + 2));  // the ring slot of k-step ks is free again
        }
        if constexpr (DO_EXP) {
          constexpr int hu = (j * 16) / NS1, hu_prev = j == 0 ? -1 : ((j - 1) * 16) / NS1;
          if constexpr (hu != hu_prev) {
            constexpr int u = hu >> 1, half = hu & 1, qt = u >> 2, s = u & 3;
            constexpr int base = 8 * (s & 1) + 4 * half;
            auto ex = [](float x) { return (ABL & 2) ? x : fast_exp2(x); };
            const float e0 = ex(S[cb][qt][s >> 1][base + 0]);
            const float e1 = ex(S[cb][qt][s >> 1][base + 1]);
            const float e2 = ex(S[cb][qt][s >> 1][base + 2]);
            const float e3 = ex(S[cb][qt][s >> 1][base + 3]);
            const uint32_t w0 = pack2<T>(e0, e1), w1 = pack2<T>(e2, e3);
            if constexpr (KPRE) orw |= w0 | w1;
            asm volatile("" ::"v"(w0), "v"(w1));  // a use in THIS block: keeps the exp / cvt work from sinking to phase 2
            pfw[qt][s][2 * half + 0] = w0;
            pfw[qt][s][2 * half + 1] = w1;
          }
          // (the first V fragments of phase 2, early: issued in the last step their LDS latency opens phase 2)
          if constexpr (j == NS1 / 2) read_v(vb, IC(0));
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      fa2_for<NS1>(step);
    };

    // ---- masks of an edge tile (causal diagonal, keys past Sk) on S[nb]; first key kv0n.  Rare: not overlapped.
    // Key kv0n + c + 4h (c = 32 tt + (i & 3) + 8 (i >> 2), a compile-time constant per register) is visible to query
    // row q iff it is <= klim[q]: one compare against a per-lane threshold per element, in groups of 8 so that the
    // compare results do not pile up in scalar registers.
    auto mask_tile = [&](auto NB_, int kv0n) {
      constexpr int nb = decltype(NB_)::value;
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results (asm) are read by the vector ALU next
      fa2_for<QT>([&](auto QTI) {
        constexpr int qt = decltype(QTI)::value;
        const int thr = klim[qt] - kv0n - 4 * h;
        fa2_for<4>([&](auto G_) {
          constexpr int tt = decltype(G_)::value >> 1, i0 = 8 * (decltype(G_)::value & 1);
#pragma unroll
          for (int i = i0; i < i0 + 8; ++i) {
            const int c = 32 * tt + (i & 3) + 8 * (i >> 2);
            if (c > thr) S[nb][qt][tt][i] = -INFINITY;
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    };

    // ---- phase 2: O^T += V^T . P^T, L += ones . P^T   ||   S[nb] := S[nb] * c2 - reference (exp2 domain) and its
    // row max, DMA issue
    auto phase2 = [&](auto NB_, auto DO_PV_, const char* vb, int dma_tile) {
      constexpr int nb = decltype(NB_)::value;
      constexpr bool DO_PV = decltype(DO_PV_)::value != 0;
      constexpr int PS = 2 * DT_ + 2, NS2 = 4 * PS;  // per 16-key k-step: 2 DT PV MFMAs + 2 row-sum MFMAs
      using ROLE = Fa3P2Role<NS2, PS, UPW>;
      auto step = [&](auto J_) {
        constexpr int j = decltype(J_)::value;
        constexpr int s = j / PS, m = j % PS;
        if constexpr (DO_PV) {
          if constexpr (m < 2 * DT_) {
            constexpr int dt = m >> 1, qt = m & 1;
            Fa2Acc<T, FA3_T_O + qt * DT_ + dt>::mfma(vfr[s & 1][dt], __builtin_bit_cast(X8, pfw[qt][s]));
          } else if constexpr (m == 2 * DT_) {
            if constexpr (!(ABL & 4)) OPS::template lsum0<FA3_A_ONES>(__builtin_bit_cast(X8, pfw[0][s]));
          } else {
            if constexpr (!(ABL & 4)) OPS::template lsum1<FA3_A_ONES>(__builtin_bit_cast(X8, pfw[1][s]));
          }
          if constexpr (m == 1 && s + 1 < 4) read_v(vb, IC(s + 1));  // slot (s+1)&1 was last read by k-step s-1
          if constexpr (ROLE::role(j) >= 32 && !(ABL & 8)) dma_unit(IC(ROLE::role(j) - 32), dma_tile);
        }
        // scale-and-subtract + max of 4 scores per group, in the order the QK^T MFMAs of phase 1 finished writing
        // them: sub-tile qt = i / 8, 32-key half tt = (i / 4) & 1, registers 4 (i & 3) .. +3
        if constexpr (!KPRE && ROLE::role(j) >= 1 && ROLE::role(j) <= 16 && !(DO_PV && (ABL & 1))) {
          constexpr int i = ROLE::role(j) - 1, qt = i / 8, tt = (i / 4) & 1, r0 = 4 * (i & 3);
          // scalar fmas: this translation unit is compiled with -fno-slp-vectorize -- SLP packs adjacent scalar fmas into
          // v_pk_fma_f32, which costs more issue time beside MFMAs than the two scalar forms (MI355X_MICROARCH.md,
          // packed f32 VALU); written as single-instruction asm instead, hipcc puts an s_nop between each group's
          // fmas and the max that reads them (4 of the step's 32 cycles)
          auto fma1 = [&](float x) {
            return __builtin_fmaf(x, c2, negref[qt]);
          };
          float v0 = fma1(S[nb][qt][tt][r0 + 0]);
          float v1 = fma1(S[nb][qt][tt][r0 + 1]);
          float v2 = fma1(S[nb][qt][tt][r0 + 2]);
          float v3 = fma1(S[nb][qt][tt][r0 + 3]);
          S[nb][qt][tt][r0 + 0] = v0;
          S[nb][qt][tt][r0 + 1] = v1;
          S[nb][qt][tt][r0 + 2] = v2;
          S[nb][qt][tt][r0 + 3] = v3;
          if constexpr ((i & 7) == 0) mx[qt] = fmaxf(fmaxf(fmaxf(v0, v1), v2), v3);
          else mx[qt] = fmaxf(fmaxf(fmaxf(fmaxf(mx[qt], v0), v1), v2), v3);  // two v_max3
          asm volatile("" : "+v"(mx[qt]));  // pins this step's share of the work to this micro-step
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      fa2_for<NS2>(step);
    };

    // ---- reference update for the tile in S[nb] (rare after the first tiles: deferred-rescale threshold).  One
    // wave-uniform test for both query sub-tiles; inside, a row that does not need to move gets delta = 0, alpha = 1.
    auto update = [&](auto NB_) {
      constexpr int nb = decltype(NB_)::value;
      // per-lane test on the half-row maxima (the two lane halves of a row are only combined inside the rare branch)
      const bool trig = (mx[0] > FA_RESCALE_THR) || (m_i[0] == -INFINITY) || (mx[1] > FA_RESCALE_THR) || (m_i[1] == -INFINITY);
      if (__builtin_amdgcn_ballot_w64(trig) != 0) {
        float mxr[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) mxr[qt] = fmaxf(mx[qt], other_half(mx[qt]));
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs retired before the file is read
        fa2_for<QT>([&](auto QTI) {
          constexpr int qt = decltype(QTI)::value;
          const bool fresh = (m_i[qt] == -INFINITY);
          const float ref_old = fresh ? 0.f : m_i[qt];
          const float m_new = fmaxf(m_i[qt], mxr[qt] + ref_old);
          const float ref_new = (m_new == -INFINITY) ? 0.f : m_new;
          const float delta = ref_new - ref_old;
          const float alpha = fresh ? 1.f : fast_exp2(-delta);
          m_i[qt] = m_new;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            S[nb][qt][0][i] -= delta;
            S[nb][qt][1][i] -= delta;
          }
          // (alpha = 1 in every lane -- the first tile of a pass without carried state, or a trigger caused by the
          //  other sub-tile -- leaves O and L as they are: skip the pass over the accumulator file)
          if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
            auto rescale = [&](auto KI) {
              constexpr int k = decltype(KI)::value;
              f32x4_t v[4] = {Fa2AccIO<k>::template read4<0>(), Fa2AccIO<k>::template read4<1>(),
                              Fa2AccIO<k>::template read4<2>(), Fa2AccIO<k>::template read4<3>()};
  #pragma unroll
              for (int gq = 0; gq < 4; ++gq)
  #pragma unroll
                for (int e = 0; e < 4; ++e) v[gq][e] *= alpha;
              Fa2AccIO<k>::template write4<0>(v[0]);
              Fa2AccIO<k>::template write4<1>(v[1]);
              Fa2AccIO<k>::template write4<2>(v[2]);
              Fa2AccIO<k>::template write4<3>(v[3]);
            };
            fa2_for<DT_>([&](auto DTI) { rescale(IC(FA3_T_O + qt * DT_ + decltype(DTI)::value)); });
            rescale(IC(FA3_T_L + qt));
          }
          negref[qt] = -ref_new;
        });
        asm volatile("s_nop 7" ::: "memory");
      }
    };

    // ---- KPRE: move the reference of rows that need it (fresh rows that now see a finite score; rows with some P >= 2) for
    // the tile whose scores sit in S[cb] at the OLD reference.  WHEN = 0: tile 0 of a pass (nothing exponentiated yet);
    // WHEN = 1: after phase 1 -- also shift S[cb ^ 1] (the next tile's scores, produced with the old C operand) and recompute
    // the tile's P.  Rare: not scheduled.
    auto move_ref = [&](auto CB_, auto WHEN_) {
      constexpr int cb = decltype(CB_)::value;
      constexpr int WHEN = decltype(WHEN_)::value;
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results (asm) are read by the vector ALU next
      fa2_for<QT>([&](auto QTI) {
        constexpr int qt = decltype(QTI)::value;
        float mxl = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) mxl = fmaxf(mxl, fmaxf(S[cb][qt][0][i], S[cb][qt][1][i]));
        const float mxr = fmaxf(mxl, other_half(mxl));
        const bool need = fresh[qt] ? (mxr != -INFINITY) : (mxr >= 1.0f);
        const float delta = need ? mxr + Fa3Margin<T>::value : 0.f;
        const float alpha = (need && !fresh[qt]) ? fast_exp2(-delta) : 1.f;  // a fresh row's O and L are still zero
        if (need) fresh[qt] = false;
        ref[qt] += delta;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          S[cb][qt][0][i] -= delta;
          S[cb][qt][1][i] -= delta;
          if constexpr (WHEN != 0) {
            S[cb ^ 1][qt][0][i] -= delta;
            S[cb ^ 1][qt][1][i] -= delta;
          }
          nref16[qt][i] = -ref[qt];
        }
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
          auto rescale = [&](auto KI) {
            constexpr int k = decltype(KI)::value;
            f32x4_t v[4] = {Fa2AccIO<k>::template read4<0>(), Fa2AccIO<k>::template read4<1>(),
                            Fa2AccIO<k>::template read4<2>(), Fa2AccIO<k>::template read4<3>()};
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
              for (int e = 0; e < 4; ++e) v[gq][e] *= alpha;
            Fa2AccIO<k>::template write4<0>(v[0]);
            Fa2AccIO<k>::template write4<1>(v[1]);
            Fa2AccIO<k>::template write4<2>(v[2]);
            Fa2AccIO<k>::template write4<3>(v[3]);
          };
          fa2_for<DT_>([&](auto DTI) { rescale(IC(FA3_T_O + qt * DT_ + decltype(DTI)::value)); });
          rescale(IC(FA3_T_L + qt));
        }
      });
      if constexpr (WHEN == 1) {
        orw = 0u;
        fa2_for<16>([&](auto HU_) {
          constexpr int hu = decltype(HU_)::value, u = hu >> 1, half = hu & 1, qt = u >> 2, s2 = u & 3;
          constexpr int base = 8 * (s2 & 1) + 4 * half;
          const float e0 = fast_exp2(S[cb][qt][s2 >> 1][base + 0]);
          const float e1 = fast_exp2(S[cb][qt][s2 >> 1][base + 1]);
          const float e2 = fast_exp2(S[cb][qt][s2 >> 1][base + 2]);
          const float e3 = fast_exp2(S[cb][qt][s2 >> 1][base + 3]);
          const uint32_t w0 = pack2<T>(e0, e1), w1 = pack2<T>(e2, e3);
          orw |= w0 | w1;
          pfw[qt][s2][2 * half + 0] = w0;
          pfw[qt][s2][2 * half + 1] = w1;
        });
      }
      fresh_any = __builtin_amdgcn_ballot_w64(fresh[0] || fresh[1]) != 0;
      asm volatile("s_nop 7" ::: "memory");
    };

    // ---- scores, masks, maximum and reference of tile 0
    if (n_w > 0) {
      phase1(IC(1), IC(0), smem, smem);
      if (is_edge(0)) mask_tile(IC(0), 0);
      else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      if constexpr (KPRE) {
        move_ref(IC(0), IC(0));
      } else {
        phase2(IC(0), IC(0), smem, 0);
        update(IC(0));
      }
    }

    // ---- tiles this wave computes (two per trip: the score buffers alternate), then the tiles it only helps to
    // move (other waves of the workgroup still need them: causal, this wave's rows end earlier)
    auto land = [&]() {  // end of an iteration: this wave's share of tile t + 2 has landed (tile t + 3 may still fly)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");
      __syncthreads();
    };
    unsigned long long st_sum[5] = {0, 0, 0, 0, 0};  // diagnostic build: cycles in phase 1 / mask / phase 2 / update / land
    auto iter = [&](int t, auto CB_) {
      constexpr int cb = decltype(CB_)::value;
      unsigned long long c1 = 0, c2 = 0, c2b = 0, c3 = 0, c4 = 0, c5 = 0;
      if constexpr (STAMP) c1 = __builtin_amdgcn_s_memtime();
      const char* kb_n = smem + ((t + 1) & 3) * SM::STAGE;
      const char* vb_c = smem + (t & 3) * SM::STAGE + SM::K_BYTES;
      const bool has_next = (t + 1 < n_w);
      if constexpr (KPRE) orw = 0u;
      phase1(CB_, IC(1), kb_n, vb_c);
      if constexpr (KPRE) {
        // some P >= 2 (bit 14 of a packed 16-bit word; inf and NaN included), or a row still waiting for its first score
        if (__builtin_amdgcn_ballot_w64((orw & 0x40004000u) != 0u) != 0 || fresh_any) move_ref(CB_, IC(1));
      }
      if constexpr (STAMP) c2 = __builtin_amdgcn_s_memtime();
      if constexpr (!(ABL & 32))
        if (has_next && is_edge(t + 1)) mask_tile(IC(cb ^ 1), (t + 1) * FA_BN);
      // (P words written by the vector ALU late in phase 1 are first read by an MFMA many steps into phase 2)
      if constexpr (STAMP) c2b = __builtin_amdgcn_s_memtime();
      dma_tile_base(t + 3);
      phase2(IC(cb ^ 1), IC(1), vb_c, t + 3);
      if constexpr (STAMP) c3 = __builtin_amdgcn_s_memtime();
      if constexpr (!KPRE && !(ABL & 16))
        if (has_next) update(IC(cb ^ 1));
      if constexpr (STAMP) c4 = __builtin_amdgcn_s_memtime();
      land();
      if constexpr (STAMP) {
        c5 = __builtin_amdgcn_s_memtime();
        st_sum[0] += c2 - c1; st_sum[1] += c2b - c2; st_sum[2] += c3 - c2b; st_sum[3] += c4 - c3; st_sum[4] += c5 - c4;
      }
    };
    if constexpr (STAMP) pt3 = __builtin_amdgcn_s_memtime();
    int t = 0;
    for (; t + 1 < n_w; t += 2) {
      iter(t, IC(0));
      iter(t + 1, IC(1));
    }
    if (t < n_w) {
      iter(t, IC(0));
      ++t;
    }
    for (; t < n_tiles; ++t) {  // tiles this wave only helps to move
      stage_dma(t + 3);
      land();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the next pass starts / the wave ends

    if constexpr (STAMP) {
      pt4 = __builtin_amdgcn_s_memtime();
      st_all[8] += pt1 - pt0;   // state init
      st_all[9] += pt2 - pt1;   // first K/V tiles and Q requested and landed, accumulator-file set-up
      st_all[10] += pt3 - pt2;  // tile 0 scores / masks / max
      st_all[11] += pt4 - pt3;  // tile loop + helper iterations + drain
      st_all[12] -= pt4;        // (+ end of epilogue below)
#pragma unroll
      for (int i = 0; i < 5; ++i) st_all[i] += st_sum[i];
      st_all[5] += n_w;
      st_all[6] += n_tiles;
    }
    // ---- epilogue
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last MFMAs retired before the accumulator file is read
    fa2_for<QT>([&](auto QTI) {
      constexpr int qt = decltype(QTI)::value;
      const float l_tot = Fa2AccIO<FA3_T_L + qt>::template read4<0>()[0];
      const float inv = (l_tot > 0.f) ? fast_rcp(l_tot) : 0.f;
      if (q_ok[qt]) {
        if (p.lse != nullptr && h == 0) {
          const float lse = (l_tot > 0.f) ? ((KPRE ? ref[qt] : m_i[qt]) + fast_log2(l_tot)) * FA_LN2 : -INFINITY;
          p.lse[((int64_t)b * p.H + head) * p.Sq + qrow[qt]] = lse;
        }
      }
      T* op = (p.o != nullptr) ? ((T*)p.o + b * p.os_b + head * p.os_h + (int64_t)(q_ok[qt] ? qrow[qt] : 0) * p.os_s) : nullptr;
      float* oa = (p.o_acc != nullptr)
                      ? (p.o_acc + (((int64_t)b * p.Sq + (q_ok[qt] ? qrow[qt] : 0)) * p.H + head) * p.D)
                      : nullptr;
      fa2_for<DT_>([&](auto DTI) {
        constexpr int dt = decltype(DTI)::value;
        constexpr int k = FA3_T_O + qt * DT_ + dt;
        const f32x4_t v[4] = {Fa2AccIO<k>::template read4<0>(), Fa2AccIO<k>::template read4<1>(),
                              Fa2AccIO<k>::template read4<2>(), Fa2AccIO<k>::template read4<3>()};
        // 16-bit output: the two lanes of a row (h = 0 / 1) hold d = 8g + 4h .. +3 -- one v_permlane32_swap per word
        // gives lane h the 8 contiguous columns 16 gp + 8h .. +7 of g pair gp: 16-byte stores instead of 8-byte ones
        if (op != nullptr) {
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            uint32_t w[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const f32x4_t x = v[2 * gp + k];
              w[k][0] = pack2<T>(x[0] * inv, x[1] * inv);
              w[k][1] = pack2<T>(x[2] * inv, x[3] * inv);
            }
            const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
            const int d0 = 32 * dt + 16 * gp + 8 * h;
            if (q_ok[qt] && d0 < p.D) *(u32x4_t*)(op + d0) = (u32x4_t){s0[0], s1[0], s0[1], s1[1]};
          }
        }
        if (oa != nullptr) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * dt + 8 * g + 4 * h;
            if (q_ok[qt] && d0 < p.D) {
              const f32x4_t w = {v[g][0] * inv, v[g][1] * inv, v[g][2] * inv, v[g][3] * inv};
              *(f32x4_t*)(oa + d0) = w;
            }
          }
        }
      });
    });
    if constexpr (STAMP) st_all[12] += __builtin_amdgcn_s_memtime();  // epilogue
  }  // pass
  if constexpr (STAMP) {  // p.mask doubles as the stamp buffer: [block][wave][16] u64; [7] = whole workgroup lifetime
    if (lane == 0 && p.mask != nullptr) {
      unsigned long long* d = (unsigned long long*)p.mask + ((size_t)blockIdx.x * 4 + wave) * 16;
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 7; ++i) d[i] = st_all[i];
      d[7] = t_end - st_all[7];
#pragma unroll
      for (int i = 8; i < 13; ++i) d[i] = st_all[i];
    }
  }
#undef IC
}
