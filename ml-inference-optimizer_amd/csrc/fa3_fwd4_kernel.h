// FlashAttention forward, fourth structure (head dim <= 64, no user mask, plain output): TWO waves per SIMD.
//
// fa3_fwd3_kernel (one wave per SIMD) is bound by its own instruction stream: at head dim 64 a 64-key tile carries as
// much softmax work (64 v_exp, 64 v_fma, 32 v_cvt_pk, 32 v_max3 per lane) as at 128 but only half the MFMAs to hide it
// under, and with ONE wave on a SIMD every LDS read, DMA issue, scalar instruction, wait and barrier is issue time of
// that same stream (timing-only ablations, tools/fa_ablate.py: even with all softmax work, DMA, masks and the reference
// test removed a tile takes 1.5 x its 1024 matrix-pipe cycles).  Here a workgroup is 8 waves = 256 query rows, 32 rows
// per wave, 256 registers per wave: the two waves of a SIMD share its matrix pipe and vector issue, and one wave's LDS /
// DMA / scalar / wait instructions issue beside the other wave's vector work.
//
//   * same algorithm, LDS-DMA staging and software pipelining across KV tiles as fa3_fwd3_kernel (S double-buffered:
//     QK^T of tile t+1 beside exp of tile t, PV of tile t beside scale / max of tile t+1), one barrier per tile;
//   * K/V tile = 16 DMA units of 1 KiB, two per wave (K rows XOR-swizzled instead of padded: 16 units, not 17);
//   * everything lives in architectural VGPRs through the MFMA builtins (no asm-owned accumulator file): the compiler
//     schedules inside a phase, the hardware interleaves the two waves of a SIMD.
// Not handled here (the launcher keeps those on fa3_fwd3_kernel / fa3_fwd_kernel): user masks, the fp32 (o_acc, lse)
// carry of ring attention, head dims above 64.
#pragma once
#include "fa3_fwd2_kernel.h"

constexpr int FA4_BM = 256;      // query rows per workgroup (8 waves x 32)
constexpr int FA4_STAGES = 4;
constexpr int FA4_KBYTES = FA_BN * 128;        // K tile: 64 rows x 128 B, chunk c of row r at position c ^ ((r >> 1) & 7)
constexpr int FA4_STAGE = 2 * FA4_KBYTES;      // + V tile [key/8][d/32][8][32] sub-tiles of 512 B
constexpr int FA4_SMEM = FA4_STAGES * FA4_STAGE;

// ABL (diagnostic build, timing-only, WRONG results): 1 no barrier per tile, 2 no LDS fragment reads, 4 no exp / convert,
// 8 no scale / max, 16 no MFMAs, 32 no DMA issue in the loop; 64 (correct results) row sums on the vector ALU (fp32 adds of
// the un-rounded exp values, per lane over its half of the keys) instead of the ones-MFMA.
// (A staggered form -- waves 4-7 half a tile behind waves 0-3, PV of tile t-1 before QK^T / exp of tile t, 8 LDS stages --
// was built and measured 4.5 % SLOWER than lockstep, 0.340 vs 0.325 ms)
//
// KPRE (FaDev::k_prescaled): K arrives already multiplied by softmax_scale * log2(e) -- applied in fp32 in the epilogue of
// the GEMM that produced it, before its one rounding to 16 bits (ops.gemm_bias_act col_scale=; scaling a 16-bit K or Q
// afterwards would add a rounding and costs 3-5e-3 of lse accuracy).  Then the running reference enters the QK^T product
// as the MFMA's C operand (a 16-register tuple holding -reference: one query per lane), S = K~ . Q^T - reference is
// already the exp2 argument, and the 32 v_fma + 16 v_max3 per lane and tile of the scale / max pass are gone.  The rescale
// trigger moves behind the exp: the reference is kept FA4_MARGIN above the running maximum, so P <= 2^-MARGIN normally, and
// "some P >= 2" (the row outgrew the maximum by 2^(MARGIN + 1)) is bit 14 of a packed 16-bit word -- one v_or3_b32 per four
// values; the rare branch recomputes the tile's P from the still intact scores.
template <typename T>
struct Fa4Margin { static constexpr float value = 5.0f; };   // bf16: exponent range of fp32
template <>
struct Fa4Margin<_Float16> { static constexpr float value = 2.0f; };  // fp16: keep the small probabilities out of the subnormals

template <typename T, bool CAUSAL, int ABL = 0, bool KPRE = false>
__global__ __launch_bounds__(512) void fa3_fwd4_kernel(const FaDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  constexpr int KS = 4, DT_ = 2, UPW = 2;
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
  const float c2 = p.scale_log2e;
  const X8 ones = __builtin_bit_cast(X8, (u32x4_t){pack2<T>(1.f, 1.f), pack2<T>(1.f, 1.f), pack2<T>(1.f, 1.f), pack2<T>(1.f, 1.f)});

  // per-lane LDS read offsets.  K: lane (r, h) reads row 32 tt + r, chunk 2 ks + h at position (2 ks + h) ^ ((r >> 1) & 7)
  const int kx = (r >> 1) & 7;
  int k_rd[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) k_rd[ks] = r * 128 + 16 * ((2 * ks + h) ^ kx);
  const int g16 = lane >> 4, i16 = lane & 15;
  const int v_rd = FA4_KBYTES + (4 * h + (i16 >> 2)) * 64 + 32 * (g16 & 1) + 8 * (i16 & 3);

  const T* kbase = (const T*)p.k + b * p.ks_b + kvh * p.ks_h;
  const T* vbase = (const T*)p.v + b * p.vs_b + kvh * p.vs_h;
  const int d_chunks = p.D >> 3;
  const int ks2 = (int)p.ks_s * 2, vs2 = (int)p.vs_s * 2;  // row strides in bytes
  const int last_tile = (p.Sk - 1) >> 6, last_row = (p.Sk - 1) & (FA_BN - 1);
  // DMA: wave w moves K unit w (rows 8w .. 8w+7) and V unit w (key groups 2w / 2w+1 ... see the V image) of every tile.
  // Per-lane source offsets from the tile's first row, for a full tile and for the (possibly partial) last tile, whose
  // rows past Sk are clamped to the last valid one (such keys are masked; the values only need to be finite).
  int offk, offkl, offv, offvl;
  {
    const int krow = 8 * wave + (lane >> 3);
    int kc = (lane & 7) ^ ((krow >> 1) & 7);
    kc = kc < d_chunks ? kc : d_chunks - 1;
    const int krowl = krow < last_row ? krow : last_row;
    offk = krow * ks2 + 16 * kc;
    offkl = krowl * ks2 + 16 * kc;
    const int blk = 2 * wave + (lane >> 5);
    const int vrow = 8 * (blk / DT_) + ((lane & 31) >> 2);
    int vc = 4 * (blk % DT_) + (lane & 3);
    vc = vc < d_chunks ? vc : d_chunks - 1;
    const int vrowl = vrow < last_row ? vrow : last_row;
    offv = vrow * vs2 + 16 * vc;
    offvl = vrowl * vs2 + 16 * vc;
  }

  const int npass = (CAUSAL && (p.nqblk - 1 - qi) != qi) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
    const int qblk = CAUSAL ? (pass == 0 ? p.nqblk - 1 - qi : qi) : qi;
    const int q0 = qblk * FA4_BM;
    const int wrow0 = q0 + wave * 32;
    const int qrow = wrow0 + r;
    const bool q_ok = qrow < p.Sq;

    int n_tiles, n_w;
    if (CAUSAL) {
      int kmax = q0 + FA4_BM - 1 + p.q_offset - p.k_offset;
      if (kmax > p.Sk - 1) kmax = p.Sk - 1;
      n_tiles = kmax < 0 ? 0 : kmax / FA_BN + 1;
      int kw = wrow0 + 31 + p.q_offset - p.k_offset;
      if (kw > p.Sk - 1) kw = p.Sk - 1;
      n_w = kw < 0 ? 0 : kw / FA_BN + 1;
    } else {
      n_tiles = (p.Sk + FA_BN - 1) / FA_BN;
      n_w = n_tiles;
    }
    const int n_tiles_dma = n_tiles > 0 ? n_tiles : 1;
    int klim = p.Sk - 1, lim0 = p.Sk - 1;
    if (CAUSAL) {
      const int c = qrow + p.q_offset - p.k_offset;
      klim = c < klim ? c : klim;
      const int c0 = wrow0 + p.q_offset - p.k_offset;
      lim0 = c0 < lim0 ? c0 : lim0;
    }
    const int first_edge = (lim0 + 1) / FA_BN;  // tiles t >= first_edge contain a key past the limit of the wave's first row

    // one tile's two DMA units of this wave (tile index clamped: a run past the end re-fetches the last tile into a dead
    // stage, which keeps the loads per iteration -- and the counted waits -- the same for every iteration)
    auto stage_dma = [&](int tile_) {
      const int tile = tile_ < n_tiles_dma ? tile_ : n_tiles_dma - 1;
      const uint32_t ko = __builtin_amdgcn_readfirstlane((uint32_t)(tile * FA_BN) * (uint32_t)ks2);
      const uint32_t vo = __builtin_amdgcn_readfirstlane((uint32_t)(tile * FA_BN) * (uint32_t)vs2);
      const char* kb = (const char*)kbase + ko;
      const char* vb = (const char*)vbase + vo;
      const bool lastt = (tile == last_tile);
      const uint32_t lds = (uint32_t)(size_t)((MIO_LDS char*)(smem + (tile_ & (FA4_STAGES - 1)) * FA4_STAGE)) + 1024 * wave;
      const int ok_ = lastt ? offkl : offk, ov_ = lastt ? offvl : offv;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(ok_), "s"(kb) : "memory", "m0");
      asm volatile("s_add_i32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                   :
                   : "s"(lds), "n"(FA4_KBYTES), "v"(ov_), "s"(vb)
                   : "memory", "m0", "scc");
    };

    __syncthreads();  // the previous pass is done with every LDS stage
    stage_dma(0);
    stage_dma(1);
    stage_dma(2);

    // ---- Q fragments (B operand: lane (r, h) holds Q[qrow][16 ks + 8 h .. +7]); rows past Sq / chunks past D are zero
    X8 qf[KS];
    {
      const T* qp = (const T*)p.q + b * p.qs_b + head * p.qs_h + (int64_t)(q_ok ? qrow : 0) * p.qs_s;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int d0 = 16 * ks + 8 * h;
        u32x4_t raw = *(const u32x4_t*)(qp + (d0 < p.D ? d0 : 0));
        const uint32_t keep = (q_ok && d0 < p.D) ? 0xffffffffu : 0u;
        raw[0] &= keep; raw[1] &= keep; raw[2] &= keep; raw[3] &= keep;
        qf[ks] = __builtin_bit_cast(X8, raw);
      }
    }
    f32x16_t O[DT_], L;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      O[0][i] = 0.f; O[1][i] = 0.f; L[i] = 0.f;
    }
    // m_i: reference the probabilities are taken against (exp2 domain; -inf = no finite score yet, the reference is then
    // 0); negref = -reference as used by the scale-and-subtract
    float m_i = -INFINITY, negref = 0.f, mx = -INFINITY;
    float lsum = 0.f;  // (ABL & 64) this lane's partial row sum: its 32 of the tile's 64 keys
    // KPRE state: ref = the reference subtracted through the MFMA's C operand (running maximum at the last move + margin;
    // 0 while the row has seen no finite score), nref16 = -ref in every register, orw = OR of the tile's packed P words
    float ref = 0.f;
    bool fresh = true;
    bool fresh_any = true;  // wave-uniform: some row of the wave is still fresh
    uint32_t orw = 0u;
    f32x16_t nref16;
#pragma unroll
    for (int i = 0; i < 16; ++i) nref16[i] = 0.f;

    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");  // tiles 0 and 1 have landed, tile 2 may still fly
    __syncthreads();

    f32x16_t S[2][2];  // score tiles: buffer (t & 1), 32-key half
    u32x4_t pfw[4];    // P^T fragments of the tile in its PV phase: k-step s (16 keys)

    // S[nb] = raw scores of the tile whose K image starts at kb.  The eight K fragments are requested up front (their LDS
    // latency runs under whatever the scheduler places next), then the MFMAs.
    auto qk = [&](auto NB_, const char* kb) {
      constexpr int nb = decltype(NB_)::value;
      X8 kf[KS][2];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          if constexpr (ABL & 2) kf[ks][tt] = qf[(ks + tt) & 3];
          else kf[ks][tt] = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd[ks] + 4096 * tt));
        }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          if constexpr (ABL & 16) {
            S[nb][tt][ks] = __uint_as_float(__builtin_bit_cast(u32x4_t, kf[ks][tt])[0]);
          } else if (ks == 0) {
            f32x16_t z;
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] = 0.f;
            S[nb][tt] = DT<T>::mfma32(kf[ks][tt], qf[ks], KPRE ? nref16 : z);
          } else {
            S[nb][tt] = DT<T>::mfma32(kf[ks][tt], qf[ks], S[nb][tt]);
          }
        }
      }
    };
    X8 vfr[2][DT_];
    auto read_v = [&](const char* vb, auto S_) {
      constexpr int s = decltype(S_)::value;
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt) {
        if constexpr (ABL & 2) {
          vfr[s & 1][dt] = qf[(s + dt) & 3];
        } else {
          const X4 lo = DT<T>::ds_read_tr(vb + v_rd + ((2 * s + 0) * DT_ + dt) * 512);
          const X4 hi = DT<T>::ds_read_tr(vb + v_rd + ((2 * s + 1) * DT_ + dt) * 512);
          X8 f;
          f[0] = lo[0]; f[1] = lo[1]; f[2] = lo[2]; f[3] = lo[3];
          f[4] = hi[0]; f[5] = hi[1]; f[6] = hi[2]; f[7] = hi[3];
          vfr[s & 1][dt] = f;
        }
      }
    };
    // one exp / convert half-unit of P = exp2(S[cb]): 4 v_exp + 2 v_cvt_pk -> two words of the k-step fragment
    auto half_unit = [&](auto CB_, auto HU_) {
      constexpr int cb = decltype(CB_)::value;
      constexpr int hu = decltype(HU_)::value, s = hu >> 1, half = hu & 1;
      constexpr int base = 8 * (s & 1) + 4 * half;
      auto ex = [](float x) { return (ABL & 4) ? x : fast_exp2(x); };
      const float e0 = ex(S[cb][s >> 1][base + 0]);
      const float e1 = ex(S[cb][s >> 1][base + 1]);
      const float e2 = ex(S[cb][s >> 1][base + 2]);
      const float e3 = ex(S[cb][s >> 1][base + 3]);
      const uint32_t w0 = (ABL & 4) ? __float_as_uint(e0) ^ __float_as_uint(e1) : pack2<T>(e0, e1);
      const uint32_t w1 = (ABL & 4) ? __float_as_uint(e2) ^ __float_as_uint(e3) : pack2<T>(e2, e3);
      if constexpr (ABL & 64) {
        lsum += (e0 + e1) + (e2 + e3);
        asm volatile("" : "+v"(lsum));
      }
      if constexpr (KPRE) orw |= w0 | w1;
      asm volatile("" ::"v"(w0), "v"(w1));  // a use in THIS step: keeps the work from sinking to its consumer in phase 2
      pfw[s][2 * half + 0] = w0;
      pfw[s][2 * half + 1] = w1;
    };
    // ---- phase 1: S[cb ^ 1] = raw scores of the next tile (K image at kb)  ||  P = exp2(S[cb]).  A fixed sequence of
    // micro-steps pinned with sched_barrier(0): all eight K fragments are requested first, the first exp / convert
    // half-unit (4 exp + 2 cvt) runs under their LDS latency, then one MFMA + one half-unit per step.
    auto phase1 = [&](auto CB_, const char* kb, const char* vb) {
      constexpr int cb = decltype(CB_)::value, nb = cb ^ 1;
      X8 kf[KS][2];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          if constexpr (ABL & 2) kf[ks][tt] = qf[(ks + tt) & 3];
          else kf[ks][tt] = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + k_rd[ks] + 4096 * tt));
        }
      __builtin_amdgcn_sched_barrier(0);
      half_unit(CB_, IC(0));
      __builtin_amdgcn_sched_barrier(0);
      fa2_for<8>([&](auto J_) {
        constexpr int j = decltype(J_)::value, ks = j >> 1, tt = j & 1;
        if constexpr (ABL & 16) {
          S[nb][tt][ks] = __uint_as_float(__builtin_bit_cast(u32x4_t, kf[ks][tt])[0]);
        } else if constexpr (ks == 0) {
          f32x16_t z;
#pragma unroll
          for (int i = 0; i < 16; ++i) z[i] = 0.f;
          S[nb][tt] = DT<T>::mfma32(kf[ks][tt], qf[ks], KPRE ? nref16 : z);
        } else {
          S[nb][tt] = DT<T>::mfma32(kf[ks][tt], qf[ks], S[nb][tt]);
        }
        // half-units 6 / 7 (k-step 3) ride in phase 2's fragment-read steps; KPRE: all eight here -- the rescale test
        // needs the whole tile's P before any of it enters O
        if constexpr (KPRE ? (j < 7) : (j < 5)) half_unit(CB_, IC(j + 1));
        if constexpr (j == 5) read_v(vb, IC(0));  // the first V fragments of phase 2, early: their LDS latency runs under steps 6 / 7
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // ---- phase 2: O^T += V^T . P^T, L += ones . P^T (V image at vb)  ||  S[nb] := S[nb] * c2 - reference (exp2 domain) and
    // the lane's half-row maximum.  12 micro-steps (k-step s: O tile 0, O tile 1, L): the V fragments of k-step s + 1 are
    // requested in the first step of k-step s (ring of two), eight of the other steps carry one scale / max group
    // (4 fma + 2 max3) each.
    auto phase2 = [&](auto NB_, const char* vb, auto DS_) {
      constexpr int nb = decltype(NB_)::value;
      constexpr bool DO_SCALE = decltype(DS_)::value != 0;
      auto CB_ = IC(nb ^ 1);
      fa2_for<12>([&](auto J_) {
        constexpr int j = decltype(J_)::value, s = j / 3, m = j % 3;
        const X8 pf = __builtin_bit_cast(X8, pfw[s]);
        if constexpr (ABL & 16) {
          if constexpr (m < 2) O[m][s] += __uint_as_float(__builtin_bit_cast(u32x4_t, vfr[s & 1][m])[0] ^ pfw[s][m]);
        } else if constexpr (m < 2) {
          O[m] = DT<T>::mfma32(vfr[s & 1][m], pf, O[m]);
        } else if constexpr (!(ABL & 64)) {
          L = DT<T>::mfma32(ones, pf, L);
        }
        constexpr bool rd = (m == 0 && s < 3);
        if constexpr (rd) read_v(vb, IC(s + 1));
        if constexpr (!KPRE && j == 0) half_unit(CB_, IC(6));
        if constexpr (!KPRE && j == 3) half_unit(CB_, IC(7));
        // scale / max groups ride in the steps without a fragment read: group index = rank of this step among them
        constexpr int g = j - (s < 3 ? s + 1 : 3);  // steps before j that read: min(s + 1, 3) when m > 0 ... (m == 0: s)
        constexpr int grp = rd ? -1 : (m == 0 ? j - 3 : g);
        if constexpr (!KPRE && DO_SCALE && !(ABL & 8) && grp >= 0 && grp < 8) {
          constexpr int tt = grp >> 2, r0 = 4 * (grp & 3);
          const float v0 = __builtin_fmaf(S[nb][tt][r0 + 0], c2, negref);
          const float v1 = __builtin_fmaf(S[nb][tt][r0 + 1], c2, negref);
          const float v2 = __builtin_fmaf(S[nb][tt][r0 + 2], c2, negref);
          const float v3 = __builtin_fmaf(S[nb][tt][r0 + 3], c2, negref);
          S[nb][tt][r0 + 0] = v0; S[nb][tt][r0 + 1] = v1; S[nb][tt][r0 + 2] = v2; S[nb][tt][r0 + 3] = v3;
          if constexpr (grp == 0) mx = fmaxf(fmaxf(fmaxf(v0, v1), v2), v3);
          else mx = fmaxf(fmaxf(fmaxf(fmaxf(mx, v0), v1), v2), v3);
          asm volatile("" : "+v"(mx));  // pins this step's share of the work to this micro-step
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // scale / max alone (tile 0 of a pass)
    auto scale_max = [&](auto NB_) {
      constexpr int nb = decltype(NB_)::value;
      float m0 = -INFINITY, m1 = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const float a = __builtin_fmaf(S[nb][tt][i], c2, negref);
          const float bq = __builtin_fmaf(S[nb][tt][i + 1], c2, negref);
          S[nb][tt][i] = a;
          S[nb][tt][i + 1] = bq;
          if (tt == 0) m0 = fmaxf(fmaxf(m0, a), bq);
          else m1 = fmaxf(fmaxf(m1, a), bq);
        }
      mx = fmaxf(m0, m1);
    };
    // masks of an edge tile (causal diagonal, keys past Sk) on S[nb]; first key kv0n.  Key kv0n + c + 4h with
    // c = 32 tt + (i & 3) + 8 (i >> 2) is visible to this lane's query row iff it is <= klim.
    auto mask_tile = [&](auto NB_, int kv0n) {
      constexpr int nb = decltype(NB_)::value;
      const int thr = klim - kv0n - 4 * h;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int c = 32 * tt + (i & 3) + 8 * (i >> 2);
          if (c > thr) S[nb][tt][i] = -INFINITY;
        }
    };
    // reference update for the tile in S[nb] (rare after the first tiles: deferred-rescale threshold)
    auto update = [&](auto NB_) {
      constexpr int nb = decltype(NB_)::value;
      const bool trig = (mx > FA_RESCALE_THR) || (m_i == -INFINITY);
      if (__builtin_amdgcn_ballot_w64(trig) != 0) {
        const float mxr = fmaxf(mx, other_half(mx));
        const bool fresh = (m_i == -INFINITY);
        const float ref_old = fresh ? 0.f : m_i;
        const float m_new = fmaxf(m_i, mxr + ref_old);
        const float ref_new = (m_new == -INFINITY) ? 0.f : m_new;
        const float delta = ref_new - ref_old;
        const float alpha = fresh ? 1.f : fast_exp2(-delta);
        m_i = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          S[nb][0][i] -= delta;
          S[nb][1][i] -= delta;
        }
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            O[0][i] *= alpha;
            O[1][i] *= alpha;
            L[i] *= alpha;
          }
          lsum *= alpha;
        }
        negref = -ref_new;
      }
    };
    // KPRE: move the reference of rows that need it (fresh rows that now see a finite score; rows with some P >= 2), for
    // the tile whose scores sit in S[cb] at the OLD reference.  REDO: also recompute that tile's P and shift S[cb ^ 1] (the
    // next tile's scores, produced with the old C operand).  Rare: not scheduled.
    auto move_ref = [&](auto CB_, auto REDO_) {
      constexpr int cb = decltype(CB_)::value;
      constexpr bool REDO = decltype(REDO_)::value != 0;
      float mxl = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) mxl = fmaxf(mxl, fmaxf(S[cb][0][i], S[cb][1][i]));
      const float mxr = fmaxf(mxl, other_half(mxl));
      const bool need = fresh ? (mxr != -INFINITY) : (mxr >= 1.0f);
      const float delta = need ? mxr + Fa4Margin<T>::value : 0.f;
      const float alpha = (need && !fresh) ? fast_exp2(-delta) : 1.f;  // a fresh row's O and L are still zero
      if (need) fresh = false;
      ref += delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        S[cb][0][i] -= delta;
        S[cb][1][i] -= delta;
        if constexpr (REDO) {
          S[cb ^ 1][0][i] -= delta;
          S[cb ^ 1][1][i] -= delta;
        }
        nref16[i] = -ref;
      }
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          O[0][i] *= alpha;
          O[1][i] *= alpha;
          L[i] *= alpha;
        }
      }
      if constexpr (REDO) {
        orw = 0u;
        fa2_for<8>([&](auto HU_) { half_unit(CB_, HU_); });
      }
      fresh_any = __builtin_amdgcn_ballot_w64(fresh) != 0;
    };
    auto is_edge = [&](int t) -> bool { return t >= first_edge; };
    auto land = [&]() {  // this wave's share of tile t + 2 has landed (tile t + 3 may still fly); then everyone's
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(UPW) : "memory");
      if constexpr (!(ABL & 1)) __syncthreads();
    };

    auto stg = [&](int tile) -> const char* { return smem + (tile & (FA4_STAGES - 1)) * FA4_STAGE; };
    int t = 0;
    // ---- scores, masks, maximum and reference of tile 0
    if (n_w > 0) {
      qk(IC(0), stg(0));
      if (is_edge(0)) mask_tile(IC(0), 0);
      if constexpr (KPRE) {
        move_ref(IC(0), IC(0));
      } else {
        scale_max(IC(0));
        update(IC(0));
      }
    }
    // ---- tile loop (two tiles per trip: the score buffers alternate)
    auto iter = [&](int t, auto CB_) {
      constexpr int cb = decltype(CB_)::value;
      const bool has_next = (t + 1 < n_w);
      if constexpr (KPRE) orw = 0u;
      phase1(CB_, stg(t + 1), stg(t));   // (after the last tile of the wave the QK^T half produces scores nobody reads)
      if constexpr (KPRE) {
        // some P >= 2 (bit 14 of a packed 16-bit word; inf and NaN included), or a row still waiting for its first score
        if (__builtin_amdgcn_ballot_w64((orw & 0x40004000u) != 0u) != 0 || fresh_any) move_ref(CB_, IC(1));
      }
      if (has_next && is_edge(t + 1)) mask_tile(IC(cb ^ 1), (t + 1) * FA_BN);
      if constexpr (!(ABL & 32)) stage_dma(t + 3);
      phase2(IC(cb ^ 1), stg(t), IC(1));
      if constexpr (!KPRE && !(ABL & 8))
        if (has_next) update(IC(cb ^ 1));
      land();
    };
    for (; t + 1 < n_w; t += 2) {
      iter(t, IC(0));
      iter(t + 1, IC(1));
    }
    if (t < n_w) {
      iter(t, IC(0));
      ++t;
    }
    for (; t < n_tiles; ++t) {  // tiles this wave only helps to move (causal: its rows end earlier)
      stage_dma(t + 3);
      land();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may still be writing LDS when the next pass starts / the wave ends

    // ---- epilogue
    {
      const float l_tot = (ABL & 64) ? lsum + other_half(lsum) : L[0];
      const float inv = (l_tot > 0.f) ? fast_rcp(l_tot) : 0.f;
      if (q_ok && p.lse != nullptr && h == 0) {
        const float lse = (l_tot > 0.f) ? ((KPRE ? ref : m_i) + fast_log2(l_tot)) * FA_LN2 : -INFINITY;
        p.lse[((int64_t)b * p.H + head) * p.Sq + qrow] = lse;
      }
      T* op = (T*)p.o + b * p.os_b + head * p.os_h + (int64_t)(q_ok ? qrow : 0) * p.os_s;
#pragma unroll
      for (int dt = 0; dt < DT_; ++dt) {
        // the two lanes of a row (h = 0 / 1) hold d = 8g + 4h .. +3 -- one v_permlane32_swap per word gives lane h the 8
        // contiguous columns 16 gp + 8h .. +7 of g pair gp: 16-byte stores
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          uint32_t w[2][2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int i0 = 4 * (2 * gp + k);
            w[k][0] = pack2<T>(O[dt][i0 + 0] * inv, O[dt][i0 + 1] * inv);
            w[k][1] = pack2<T>(O[dt][i0 + 2] * inv, O[dt][i0 + 3] * inv);
          }
          const auto s0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
          const int d0 = 32 * dt + 16 * gp + 8 * h;
          if (q_ok && d0 < p.D) *(u32x4_t*)(op + d0) = (u32x4_t){s0[0], s1[0], s0[1], s1[1]};
        }
      }
    }
  }  // pass
#undef IC
}
