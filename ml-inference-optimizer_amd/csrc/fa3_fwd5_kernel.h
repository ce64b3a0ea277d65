// FlashAttention forward, fifth structure: fa3_fwd4_kernel's two-waves-per-SIMD skeleton on v_mfma_f32_16x16x32 tiles
// (head dim <= 64, k_prescaled launches only: K carries softmax_scale * log2(e), fa3_fwd4_kernel.h KPRE).
//
// Why: the attention kernels are bound by the power cap, so what counts is energy per tile.
//   * On random operands the chip sustains ~1.2 x the FLOP/s on 16x16x32 that it does on 32x32x16 (tools/micro/
//     mfma_peak.hip: 1.88 vs 1.53 PFLOP/s, same cycles per FLOP, higher clock).
//   * The row sum "ones . P^T" is one MFMA per (16 queries, 32 keys): 4 of 36 MFMAs of 16 cycles per wave-tile instead of
//     4 of 20 of 32 cycles -- 1/9 of the matrix-pipe time instead of 1/5.
// The swapped product carries over to the 16x16 shape:
//   S^T tile (16 keys x 16 queries) = K tile (A: lane (r, g) = K[key 16 kt + r][d 32 ds + 8 g .. +7], one ds_read_b128)
//                                     . Q^T (B: lane (c, g) = Q[query 16 qg + c][d 32 ds + 8 g .. +7]); accumulator: query
//                                     c on the lane, keys 16 kt + 4 g + i in registers i = 0..3;
//   P^T as the B operand of O^T += V^T . P^T needs k = 8 g + j on lane group g: the exp'd registers of key tiles 2 s and
//     2 s + 1 ARE that fragment for the 32-key step s if k <-> key is read as j < 4: 32 s + 4 g + j, j >= 4: 32 s + 16 + 4 g
//     + (j - 4) -- the contraction order is free as long as the A operand uses the same map;
//   V^T tile (A: 16 d rows x 32 keys) in that map = two ds_read_b64_tr_b16 of 4 consecutive keys x 16 d from a ROW-MAJOR V
//     image (lane group g: keys 32 s (+16) + 4 g .., lane i of the group receives column d = 16 dt + i).
// LDS images (both filled by DMA, swizzle on the per-lane SOURCE address): K rows of 128 B with chunk c at c ^ ((row >> 1) &
// 7) as in fwd4 (conflict-free ds_read_b128 for this lane -> (row, chunk) map too); V rows of 128 B with the 32-byte block b
// at b ^ ((row >> 1) & 3) (the 8 rows of a half-wave's transposed read then cover 8 different 32-byte blocks of the 256-byte
// bank row).
//
// Software pipeline.  No LDS read sits in front of its consumer: the fragments are requested one half-iteration ahead, one
// per micro-step, so the eight waves' 128 KB of LDS reads per tile spread over the whole iteration instead of bursting behind
// the barrier (tools/fa5_stamps.py: with the reads at the head of their phase the second-dispatched waves waited 580 cycles
// for their first K fragment and the PV half ran at the LDS latency, 60 cycles per MFMA pair).
//   iteration t:  half 1: S(t+1) = K(t+1) fragments (registers) . Q^T - ref || P(t) = exp2(S(t)) || request V(t) fragments
//                 reference test (rare: move), edge masks of S(t+1)
//                 t even: wait for this wave's DMA shares, barrier (tiles <= t+3 visible), DMA of the next two tiles
//                 half 2: O^T += V(t) fragments . P(t), row sums || request K(t+2) fragments
// A K fragment dies in the step that a V fragment is born in and vice versa: ~36 fragment registers live.  Tiles t .. t+5
// are live or in flight: 8 LDS stages of 16 KB, ONE barrier per two tiles.  The KV tiles of both causal passes (same head)
// are one stream of "virtual" tiles: the heavy pass' last iterations request the light pass' first tiles.
#pragma once
#include "fa3_fwd4_kernel.h"

constexpr int FA5_STAGES = 8;
// Waves 4..7 (the second wave of each SIMD) meet the barrier BEFORE the QK^T half of an iteration, waves 0..3 behind it: the
// two waves of a SIMD then run opposite halves (vector-heavy QK^T || exp beside matrix-only PV) instead of queueing for the
// same unit
constexpr bool FA5_STAGGER = true;
constexpr int FA5_SMEM = FA5_STAGES * FA4_STAGE;

// CARRY: the ring form -- (o_acc fp32 [B, Sq, H, D], lse) carried in (p.carry_in) and written back; p.o may be null.
// OBLK: the 16-bit output goes to the GEMMs' blocked activation layout (FaDev::o_blk launches; not with CARRY).
// KPRE = false: K arrives as it is (the reference's functional entry point, triton_flash_attention(q, k, v)): the scores
// and the running reference stay in RAW units (q . k), the C operand subtracts the raw reference, and softmax_scale * log2(e)
// is applied in fp32 on the way into exp2 (one v_pk_mul per two scores: 16 vector instructions per wave-tile more than the
// pre-scaled form, no extra rounding of Q or K).
template <typename T, bool CAUSAL, bool STAMP = false, int ABL = 0, bool CARRY = false, bool OBLK = false, bool KPRE = true>  // ABL: timing-only ablations (diagnostic build)
__global__ __launch_bounds__(512) void fa3_fwd5_kernel(const FaDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  constexpr int NKT = 4, NQG = 2, NDS = 2, NDT = 4, NS = 2;
#define IC(N) std::integral_constant<int, (N)> {}

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;

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
  X8 ones = __builtin_bit_cast(X8, (u32x4_t){pack2<T>(1.f, 1.f), pack2<T>(1.f, 1.f), pack2<T>(1.f, 1.f), pack2<T>(1.f, 1.f)});
  asm volatile("" : "+v"(ones));  // stays in four VGPRs (hipcc otherwise rebuilds it from SGPRs in front of every row-sum pair)

  // per-lane LDS read offsets
  int k_rd[NDS];  // K fragment (kt, ds): row 16 kt + c16, chunk 4 ds + g at position (4 ds + g) ^ ((c16 >> 1) & 7)
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) k_rd[ds] = c16 * 128 + 16 * ((4 * ds + g) ^ ((c16 >> 1) & 7));
  int v_rd[NDT];  // V fragment (dt, s, hf): row 32 s + 16 hf + 4 g + q, 32-byte block dt ^ x, 8 bytes at 8 p2
  {
    const int q = c16 >> 2, p2 = c16 & 3, x = ((4 * g + q) >> 1) & 3;
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) v_rd[dt] = FA4_KBYTES + (4 * g + q) * 128 + ((dt ^ x) * 32) + 8 * p2;
  }

  const T* kbase = (const T*)p.k + b * p.ks_b + kvh * p.ks_h;
  const T* vbase = (const T*)p.v + b * p.vs_b + kvh * p.vs_h;
  const int d_chunks = p.D >> 3;
  const int ks2 = (int)p.ks_s * 2, vs2 = (int)p.vs_s * 2;
  const int last_tile = (p.Sk - 1) >> 6, last_row = (p.Sk - 1) & (FA_BN - 1);
  // DMA: wave w moves rows 8 w .. 8 w + 7 of the K tile and of the V tile (one 1-KiB unit each)
  int offk, offkl, offv, offvl;
  {
    const int row = 8 * wave + (lane >> 3), pos = lane & 7;
    const int rowl = row < last_row ? row : last_row;
    int kc = pos ^ ((row >> 1) & 7);
    kc = kc < d_chunks ? kc : d_chunks - 1;
    offk = row * ks2 + 16 * kc;
    offkl = rowl * ks2 + 16 * kc;
    int vc = (((pos >> 1) ^ ((row >> 1) & 3)) << 1) | (pos & 1);
    vc = vc < d_chunks ? vc : d_chunks - 1;
    offv = row * vs2 + 16 * vc;
    offvl = rowl * vs2 + 16 * vc;
  }

  // diagnostic build (STAMP): cycles per region summed over both passes, p.mask doubles as the record buffer
  unsigned long long st_all[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (STAMP) st_all[7] = __builtin_amdgcn_s_memtime();

  const int npass = (CAUSAL && (p.nqblk - 1 - qi) != qi) ? 2 : 1;
  auto pass_q0 = [&](int pass) __attribute__((always_inline)) -> int {
    return (CAUSAL ? (pass == 0 ? p.nqblk - 1 - qi : qi) : qi) * FA4_BM;
  };
  auto pass_tiles = [&](int pass) __attribute__((always_inline)) -> int {  // KV tiles the workgroup walks in that pass
    if (!CAUSAL) return (p.Sk + FA_BN - 1) / FA_BN;
    int kmax = pass_q0(pass) + FA4_BM - 1 + p.q_offset - p.k_offset;
    if (kmax > p.Sk - 1) kmax = p.Sk - 1;
    return kmax < 0 ? 0 : kmax / FA_BN + 1;
  };
  // Q fragments (B operand: lane (c16, g) holds Q[row][32 ds + 8 g .. +7]); rows past Sq / chunks past D are zero
  auto load_q = [&](int pass, X8 (&dst)[NQG][NDS]) __attribute__((always_inline)) {
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
      const int row = pass_q0(pass) + wave * 32 + 16 * qg + c16;
      const bool ok = row < p.Sq;
      const T* qp = (const T*)p.q + b * p.qs_b + head * p.qs_h + (int64_t)(ok ? row : 0) * p.qs_s;
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        const int d0 = 32 * ds + 8 * g;
        u32x4_t raw = *(const u32x4_t*)(qp + (d0 < p.D ? d0 : 0));
        const uint32_t keep = (ok && d0 < p.D) ? 0xffffffffu : 0u;
        raw[0] &= keep; raw[1] &= keep; raw[2] &= keep; raw[3] &= keep;
        dst[qg][ds] = __builtin_bit_cast(X8, raw);
      }
    }
  };
  // The KV tiles of both passes (same head, so the same K / V rows) form ONE stream of "virtual" tiles: tbase = virtual index
  // of the current pass' tile 0, vnext = next virtual tile to request, vseen = every virtual tile below it has landed and is
  // visible to all waves.  Tile v lives in LDS stage v % FA5_STAGES.  The first tiles of the second pass are requested by the
  // last iterations of the first, and its Q rows in front of the first pass' epilogue.
  // static priority for the half of the waves that leads (waves 0..3 run half an iteration ahead): -1 % measured; the other
  // half at priority 1 instead: +3 %.  (Diagnostic library: bits 4 / 5 of xcd_remap = none / the other half.)
#ifdef MIO_DIAG
  if (p.xcd_remap & 32) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  } else if (!(p.xcd_remap & 16)) {
    if (wave < 4) __builtin_amdgcn_s_setprio(1);
  }
#else
  if (wave < 4) __builtin_amdgcn_s_setprio(1);
#endif
  int tbase = 0, vnext = 0, vseen = 0;
  X8 qf_next[NQG][NDS];
  load_q(0, qf_next);

  for (int pass = 0; pass < npass; ++pass) {
    unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0;
    if constexpr (STAMP) pt0 = __builtin_amdgcn_s_memtime();
    const int q0 = pass_q0(pass);
    const int wrow0 = q0 + wave * 32;
    int qrow[NQG];
    bool q_ok[NQG];
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
      qrow[qg] = wrow0 + 16 * qg + c16;
      q_ok[qg] = qrow[qg] < p.Sq;
    }

    const int n_tiles = pass_tiles(pass);
    const int n_tiles_next = pass + 1 < npass ? pass_tiles(pass + 1) : 0;
    int n_w = n_tiles;
    if (CAUSAL) {
      int kw = wrow0 + 31 + p.q_offset - p.k_offset;
      if (kw > p.Sk - 1) kw = p.Sk - 1;
      n_w = kw < 0 ? 0 : kw / FA_BN + 1;
    }
    int klim[NQG], lim0 = p.Sk - 1;
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
      klim[qg] = p.Sk - 1;
      if (CAUSAL) {
        const int c = qrow[qg] + p.q_offset - p.k_offset;
        klim[qg] = c < klim[qg] ? c : klim[qg];
      }
    }
    if (CAUSAL) {
      const int c0 = wrow0 + p.q_offset - p.k_offset;
      lim0 = c0 < lim0 ? c0 : lim0;
    }
    const int first_edge = (lim0 + 1) / FA_BN;
    const bool late = wave >= 4;

    auto stage_dma = [&](int tile_) __attribute__((always_inline)) {  // tile_: virtual index
      int tile = tile_ - tbase;
      if (tile >= n_tiles) tile = (tile - n_tiles < n_tiles_next) ? tile - n_tiles : -1;  // the next pass' tile, or none
      if (tile < 0) return;
      const uint32_t ko = __builtin_amdgcn_readfirstlane((uint32_t)(tile * FA_BN) * (uint32_t)ks2);
      const uint32_t vo = __builtin_amdgcn_readfirstlane((uint32_t)(tile * FA_BN) * (uint32_t)vs2);
      const char* kb = (const char*)kbase + ko;
      const char* vb = (const char*)vbase + vo;
      const bool lastt = (tile == last_tile);
      const uint32_t lds = (uint32_t)(size_t)((MIO_LDS char*)(smem + (tile_ & (FA5_STAGES - 1)) * FA4_STAGE)) + 1024 * wave;
      const int ok_ = lastt ? offkl : offk, ov_ = lastt ? offvl : offv;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(lds), "v"(ok_), "s"(kb) : "memory", "m0");
      asm volatile("s_add_i32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3"
                   :
                   : "s"(lds), "n"(FA4_KBYTES), "v"(ov_), "s"(vb)
                   : "memory", "m0", "scc");
    };

    X8 qf[NQG][NDS];
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg)
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) qf[qg][ds] = qf_next[qg][ds];
    // tiles 0 .. 3 of this pass are requested (normally by the previous pass) and tiles 0, 1 have landed
    while (vnext < tbase + 4) {
      stage_dma(vnext);
      ++vnext;
    }
    if (vseen < tbase + 2) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      vseen = vnext;
    }

    f32x4_t O[NDT][NQG], L[NQG];
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
      L[qg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) O[dt][qg] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    // per query group: ref = the reference subtracted through the C operand (running maximum at the last move + margin; 0
    // while the row is fresh), nref4 = -ref in all four registers; orw = OR of the tile's packed P words
    const float cs = KPRE ? 1.f : p.scale_log2e, ics = KPRE ? 1.f : 1.f / p.scale_log2e;  // score units -> base-2 exponent
    float ref[NQG] = {0.f, 0.f};
    bool fresh[NQG] = {true, true};
    bool fresh_any = true;
    uint32_t orw = 0u;
    f32x4_t nref4[NQG] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
    if constexpr (CARRY) {
      // a carried row continues from (o_acc, lse): reference = lse in base 2 (every earlier score lies below it), row sum 1,
      // O = the normalised carry; rows that have seen no key yet (lse = -inf) start fresh
      if (p.carry_in) {
#pragma unroll
        for (int qg = 0; qg < NQG; ++qg) {
          const float lse_in = q_ok[qg] ? p.lse[((int64_t)b * p.H + head) * p.Sq + qrow[qg]] : -INFINITY;
          if (lse_in != -INFINITY) {
            ref[qg] = lse_in * FA_LOG2E * ics;
            fresh[qg] = false;
            L[qg] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
            const float* oa = p.o_acc + (((int64_t)b * p.Sq + qrow[qg]) * p.H + head) * p.D;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
              const int d0 = 16 * dt + 4 * g;
              if (d0 < p.D) O[dt][qg] = *(const f32x4_t*)(oa + d0);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) nref4[qg][i] = -ref[qg];
        }
        fresh_any = __builtin_amdgcn_ballot_w64(fresh[0] || fresh[1]) != 0;
      }
    }

    f32x4_t S[2][NKT][NQG];  // score tiles: buffer (t & 1), 16-key tile, query group
    u32x4_t pfw[NS][NQG];    // P^T fragments: 32-key step s, query group
    X8 kf[NKT * NDS];        // K fragments of the next score tile: index 2 kt + ds
    X8 vf[NS * NDT];         // V^T fragments of the current tile: index 4 s + dt

    auto read_k = [&](const char* kb, auto F_) __attribute__((always_inline)) {  // fragment f = 2 kt + ds
      constexpr int f = decltype(F_)::value;
      kf[f] = __builtin_bit_cast(X8, *(const u32x4_t*)(kb + 2048 * (f >> 1) + k_rd[f & 1]));
    };
    auto qk_pair = [&](auto NB_, auto F_) __attribute__((always_inline)) {  // the two MFMAs (query groups 0, 1) of K fragment f = 2 kt + ds
      constexpr int nb = decltype(NB_)::value, f = decltype(F_)::value, kt = f >> 1, ds = f & 1;
#pragma unroll
      for (int qg = 0; qg < NQG; ++qg) {
        if constexpr (ds == 0) S[nb][kt][qg] = DT<T>::mfma16(kf[f], qf[qg][0], nref4[qg]);
        else S[nb][kt][qg] = DT<T>::mfma16(kf[f], qf[qg][1], S[nb][kt][qg]);
      }
    };
    // one exp / convert unit: the 4 scores of (key tile kt, query group qg) -> two words of P^T fragment (kt / 2, qg)
    auto exp_unit = [&](auto CB_, auto U_) __attribute__((always_inline)) {
      constexpr int cb = decltype(CB_)::value, u = decltype(U_)::value, kt = u >> 1, qg = u & 1;
      float x0 = S[cb][kt][qg][0], x1 = S[cb][kt][qg][1], x2 = S[cb][kt][qg][2], x3 = S[cb][kt][qg][3];
      if constexpr (!KPRE) {
        const f32x2_t c2 = {cs, cs};
        const f32x2_t a = (f32x2_t){x0, x1} * c2, bb = (f32x2_t){x2, x3} * c2;
        x0 = a[0]; x1 = a[1]; x2 = bb[0]; x3 = bb[1];
      }
      const float e0 = fast_exp2(x0);
      const float e1 = fast_exp2(x1);
      const float e2 = fast_exp2(x2);
      const float e3 = fast_exp2(x3);
      const uint32_t w0 = pack2<T>(e0, e1), w1 = pack2<T>(e2, e3);
      orw |= w0 | w1;
      asm volatile("" ::"v"(w0), "v"(w1));  // a use in THIS step: keeps the work from sinking to its consumer in phase 2
      pfw[kt >> 1][qg][2 * (kt & 1) + 0] = w0;
      pfw[kt >> 1][qg][2 * (kt & 1) + 1] = w1;
    };
    auto read_v = [&](const char* vb, auto F_) __attribute__((always_inline)) {  // fragment f = 4 s + dt
      constexpr int f = decltype(F_)::value, s = f >> 2, dt = f & 3;
      const X4 lo = DT<T>::ds_read_tr(vb + 4096 * s + v_rd[dt]);
      const X4 hi = DT<T>::ds_read_tr(vb + 4096 * s + 2048 + v_rd[dt]);
      X8 x;
      x[0] = lo[0]; x[1] = lo[1]; x[2] = lo[2]; x[3] = lo[3];
      x[4] = hi[0]; x[5] = hi[1]; x[6] = hi[2]; x[7] = hi[3];
      vf[f] = x;
    };
    // ---- phase 1: S[cb ^ 1] = scores of the next tile (its K fragments are in kf) minus the reference (C operand)  ||
    // P = exp2(S[cb])  ||  the V fragments of the current tile (image at vb) are requested.  One exp unit, then two MFMAs +
    // one unit + one fragment request per step.
    auto phase1 = [&](auto CB_, auto DO_EXP_, const char* vb) __attribute__((always_inline)) {
      constexpr int cb = decltype(CB_)::value, nb = cb ^ 1;
      constexpr bool DO_EXP = decltype(DO_EXP_)::value != 0;
      if constexpr (DO_EXP) {
        exp_unit(CB_, IC(0));
        __builtin_amdgcn_sched_barrier(0);
      }
      fa2_for<8>([&](auto ST_) __attribute__((always_inline)) {
        constexpr int st = decltype(ST_)::value;
        qk_pair(IC(nb), ST_);
        if constexpr (DO_EXP && st < 7) exp_unit(CB_, IC(st + 1));
        if constexpr (DO_EXP) read_v(vb, ST_);
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    // ---- phase 2: O^T += V^T . P^T, L += ones . P^T: 10 steps of two MFMAs (8 V fragments x 2 query groups, the row sums
    // behind each 32-key step)  ||  the K fragments of the tile after the next (image at kb) are requested
    auto phase2 = [&](const char* kb) __attribute__((always_inline)) {
      fa2_for<NS * NDT>([&](auto F_) __attribute__((always_inline)) {
        constexpr int f = decltype(F_)::value, s = f >> 2, dt = f & 3;
        O[dt][0] = DT<T>::mfma16(vf[f], __builtin_bit_cast(X8, pfw[s][0]), O[dt][0]);
        O[dt][1] = DT<T>::mfma16(vf[f], __builtin_bit_cast(X8, pfw[s][1]), O[dt][1]);
        read_k(kb, F_);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (dt == NDT - 1) {
          L[0] = DT<T>::mfma16(ones, __builtin_bit_cast(X8, pfw[s][0]), L[0]);
          L[1] = DT<T>::mfma16(ones, __builtin_bit_cast(X8, pfw[s][1]), L[1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    };
    // masks of an edge tile on S[nb]; first key kv0n.  Key kv0n + 16 kt + 4 g + i is visible to query group qg's row iff <= klim
    auto mask_tile = [&](auto NB_, int kv0n) __attribute__((always_inline)) {
      constexpr int nb = decltype(NB_)::value;
#pragma unroll
      for (int qg = 0; qg < NQG; ++qg) {
        const int thr = klim[qg] - kv0n - 4 * g;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (16 * kt + i > thr) S[nb][kt][qg][i] = -INFINITY;
      }
    };
    // move the reference of rows that need it for the tile whose scores sit in S[cb] at the OLD reference (fa3_fwd4 KPRE).
    // WHEN = 0: tile 0 of a pass; 1: after phase 1 -- also shift S[cb ^ 1] and recompute the tile's P.  Rare.
    auto move_ref = [&](auto CB_, auto WHEN_) __attribute__((always_inline)) {
      constexpr int cb = decltype(CB_)::value;
      constexpr int WHEN = decltype(WHEN_)::value;
#pragma unroll
      for (int qg = 0; qg < NQG; ++qg) {
        float mxl = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) mxl = fmaxf(mxl, S[cb][kt][qg][i]);
        mxl = fmaxf(mxl, __shfl_xor(mxl, 16, 64));  // the four lanes c16 + 16 g of a query hold its 64 keys
        const float mxr = fmaxf(mxl, __shfl_xor(mxl, 32, 64));
        const bool need = fresh[qg] ? (mxr != -INFINITY) : (mxr >= ics);  // some P >= 2
        const float delta = need ? mxr + Fa4Margin<T>::value * ics : 0.f;
        const float alpha = (need && !fresh[qg]) ? fast_exp2(-delta * cs) : 1.f;
        if (need) fresh[qg] = false;
        ref[qg] += delta;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            S[cb][kt][qg][i] -= delta;
            if constexpr (WHEN != 0) S[cb ^ 1][kt][qg][i] -= delta;
          }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          nref4[qg][i] = -ref[qg];
          if constexpr (WHEN != 0 || CARRY) {  // (tile 0 without a carry: every row is fresh, O and L are still zero)
            L[qg][i] *= alpha;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) O[dt][qg][i] *= alpha;
          }
        }
      }
      if constexpr (WHEN == 1) {
        orw = 0u;
        fa2_for<8>([&](auto U_) __attribute__((always_inline)) { exp_unit(CB_, U_); });
      }
      fresh_any = __builtin_amdgcn_ballot_w64(fresh[0] || fresh[1]) != 0;
    };
    auto is_edge = [&](int t) __attribute__((always_inline)) -> bool { return t >= first_edge; };
    // every second iteration (t even) a wave waits for its DMA shares, meets the other waves (every virtual tile requested
    // so far is then visible: pass-local tiles up to t + 3) and requests the next two tiles
    auto sync_and_dma = [&](int t, auto EVEN_) __attribute__((always_inline)) {  // EVEN_: 1 / 0 = t is even / odd, 2 = look
      constexpr int EVEN = decltype(EVEN_)::value;
      if constexpr (EVEN != 0) {
        if (EVEN == 1 || !(t & 1)) {
          if constexpr (ABL & 4) asm volatile("s_barrier" ::: "memory");
          else if constexpr (ABL & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
          vseen = vnext;
          if constexpr (!(ABL & 8)) {
            stage_dma(vnext);
            stage_dma(vnext + 1);
          }
          vnext += 2;
        }
      }
    };
    auto stg = [&](int tile) __attribute__((always_inline)) -> const char* { return smem + ((tbase + tile) & (FA5_STAGES - 1)) * FA4_STAGE; };

    if constexpr (STAMP) pt1 = __builtin_amdgcn_s_memtime();
    int t = 0;
    // ---- scores, masks and reference of tile 0; K fragments of tile 1
    if (n_w > 0) {
      fa2_for<8>([&](auto F_) __attribute__((always_inline)) { read_k(stg(0), F_); });
      phase1(IC(1), IC(0), stg(0));
      fa2_for<8>([&](auto F_) __attribute__((always_inline)) { read_k(stg(1), F_); });  // (land under the reference set-up)
      if (is_edge(0)) mask_tile(IC(0), 0);
      move_ref(IC(0), IC(0));
    }
    unsigned long long st_sum[6] = {0, 0, 0, 0, 0, 0};
    auto iter = [&](int t, auto CB_) __attribute__((always_inline)) {
      constexpr int cb = decltype(CB_)::value;
      unsigned long long c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
      if constexpr (STAMP) c1 = __builtin_amdgcn_s_memtime();
      const bool has_next = (t + 1 < n_w);
      if constexpr (FA5_STAGGER) {
        if (late) sync_and_dma(t, IC(cb ^ 1));
      }
      orw = 0u;
      phase1(CB_, IC(1), stg(t));
      if constexpr (STAMP) c2 = __builtin_amdgcn_s_memtime();
      if constexpr (ABL & 1) asm volatile("" ::"v"(orw));
      else if (__builtin_expect(__builtin_amdgcn_ballot_w64((orw & 0x40004000u) != 0u) != 0 || fresh_any, 0)) move_ref(CB_, IC(1));
      // one register home for the C-operand tuples on both paths (hipcc otherwise copies them on the COMMON path)
      asm volatile("" : "+v"(nref4[0]), "+v"(nref4[1]));
      if (__builtin_expect(has_next && is_edge(t + 1), 0)) mask_tile(IC(cb ^ 1), (t + 1) * FA_BN);
      if constexpr (STAMP) c3 = __builtin_amdgcn_s_memtime();
      if constexpr (FA5_STAGGER) {
        if (!late) sync_and_dma(t, IC(cb ^ 1));
      } else {
        sync_and_dma(t, IC(cb ^ 1));
      }
      if constexpr (STAMP) c4 = __builtin_amdgcn_s_memtime();
      phase2(stg(t + 2));
      if constexpr (STAMP) {
        c5 = __builtin_amdgcn_s_memtime();
        st_sum[0] += c2 - c1; st_sum[1] += c3 - c2; st_sum[2] += c5 - c4; st_sum[3] += c4 - c3;
      }
    };
    if constexpr (STAMP) pt2 = __builtin_amdgcn_s_memtime();
    for (; t + 1 < n_w; t += 2) {
      iter(t, IC(0));
      iter(t + 1, IC(1));
    }
    if (t < n_w) {
      iter(t, IC(0));
      ++t;
    }
    for (; t < n_tiles; ++t) sync_and_dma(t, IC(2));  // tiles this wave only helps to move
    tbase += n_tiles;
    if (pass + 1 < npass) load_q(pass + 1, qf_next);  // in front of the epilogue's stores
    if constexpr (STAMP) {
      pt3 = __builtin_amdgcn_s_memtime();
      st_all[8] += pt1 - pt0;   // Q load, first tiles requested and landed
      st_all[9] += pt2 - pt1;   // tile 0 scores / masks / reference
      st_all[10] += pt3 - pt2;  // tile loop + helper iterations + drain
      st_all[11] -= pt3;        // (+ end of epilogue below)
#pragma unroll
      for (int i = 0; i < 5; ++i) st_all[i] += st_sum[i];
      st_all[5] += n_w;
      st_all[6] += n_tiles;
    }

    // ---- epilogue: lane (c16, g) holds O[query qrow[qg]][d = 16 dt + 4 g + i], 8 bytes per (dt, qg).  The two query groups
    // are exchanged between lane rows g and g ^ 1 (v_permlane16_swap per dword): even rows end up with 16 contiguous bytes
    // (d = 16 dt + 4 g .. + 7) of query group 0, odd rows with 16 bytes (d = 16 dt + 4 (g - 1) ..) of query group 1 -- half the
    // store instructions for the same bytes (the store tail is issue-bound: 73 cycles per wave-instruction).
    float inv2[NQG];
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
      const float l_tot = L[qg][0];
      inv2[qg] = (l_tot > 0.f) ? fast_rcp(l_tot) : 0.f;
      if (q_ok[qg] && p.lse != nullptr && g == 0) {
        const float lse = (l_tot > 0.f) ? (ref[qg] * cs + fast_log2(l_tot)) * FA_LN2 : -INFINITY;
        p.lse[((int64_t)b * p.H + head) * p.Sq + qrow[qg]] = lse;
      }
    }
    bool wide = (NQG == 2 && !CARRY);
#ifdef MIO_DIAG
    wide = wide && !(p.xcd_remap & 64);  // (A/B: bit 6 keeps the 8-byte stores)
#endif
    if (wide) {
      const int mq = g & 1;                       // the query group this lane stores after the exchange
      const int dofs = 4 * (g & ~1);              // first of its 8 head-dim columns inside a 16-column tile
      const bool ok = q_ok[mq];
      const int qr = ok ? qrow[mq] : 0;
      u32x4_t keep[NDT];
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        uint32_t a0 = pack2<T>(O[dt][0][0] * inv2[0], O[dt][0][1] * inv2[0]), a1 = pack2<T>(O[dt][0][2] * inv2[0], O[dt][0][3] * inv2[0]);
        uint32_t b0 = pack2<T>(O[dt][1][0] * inv2[1], O[dt][1][1] * inv2[1]), b1 = pack2<T>(O[dt][1][2] * inv2[1], O[dt][1][3] * inv2[1]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(a1, b1, false, false);
        // even rows: {own a0, own a1, a0 / a1 of row g + 1}; odd rows: {b0 / b1 of row g - 1, own b0, own b1}
        const u32x4_t w = {s0[0], s1[0], s0[1], s1[1]};
        keep[dt] = w;
        const int d0 = 16 * dt + dofs;
        if constexpr (OBLK) {
          const int64_t m = (int64_t)b * p.Sq + qr;
          char* ob = (char*)p.o + (((m >> 8) * ((p.H * p.D) >> 5)) << 14) + ((m & 255) << 6);
          const int c = head * p.D + d0;
          if (ok && d0 < p.D) *(u32x4_t*)(ob + ((int64_t)(c >> 5) << 14) + ((c & 31) << 1)) = w;
        } else {
          T* op = (T*)p.o + b * p.os_b + head * p.os_h + (int64_t)qr * p.os_s;
          if (ok && d0 < p.D) *(u32x4_t*)(op + d0) = w;
        }
      }
      // the stored registers stay untouched until the stores have fetched them (gemm8w_kernel.h: store data must outlive ...)
      if constexpr (NDT == 4) asm volatile("s_nop 15\n\ts_nop 15" : : "v"(keep[0]), "v"(keep[1]), "v"(keep[2]), "v"(keep[3]) : "memory");
      else if constexpr (NDT == 2) asm volatile("s_nop 15\n\ts_nop 15" : : "v"(keep[0]), "v"(keep[1]) : "memory");
    } else {
#pragma unroll
    for (int qg = 0; qg < NQG; ++qg) {
      const float inv = inv2[qg];
      if (!CARRY || p.o != nullptr) {
        if constexpr (OBLK) {
          // element (m, c) of the [B*Sq, H*D] matrix in the GEMMs' blocked activation layout:
          // ((m / 256) * (H*D / 32) + c / 32) * 16 KiB + (m % 256) * 64 B + (c % 32) * 2 B
          const int64_t m = (int64_t)b * p.Sq + (q_ok[qg] ? qrow[qg] : 0);
          char* ob = (char*)p.o + (((m >> 8) * ((p.H * p.D) >> 5)) << 14) + ((m & 255) << 6);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = 16 * dt + 4 * g, c = head * p.D + d0;
            const u32x2_t w = {pack2<T>(O[dt][qg][0] * inv, O[dt][qg][1] * inv), pack2<T>(O[dt][qg][2] * inv, O[dt][qg][3] * inv)};
            if (q_ok[qg] && d0 < p.D) *(u32x2_t*)(ob + ((int64_t)(c >> 5) << 14) + ((c & 31) << 1)) = w;
          }
        } else {
          T* op = (T*)p.o + b * p.os_b + head * p.os_h + (int64_t)(q_ok[qg] ? qrow[qg] : 0) * p.os_s;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            const u32x2_t w = {pack2<T>(O[dt][qg][0] * inv, O[dt][qg][1] * inv), pack2<T>(O[dt][qg][2] * inv, O[dt][qg][3] * inv)};
            if (q_ok[qg] && d0 < p.D) *(u32x2_t*)(op + d0) = w;
          }
        }
      }
      if constexpr (CARRY) {
        if (p.o_acc != nullptr) {
          float* oa = p.o_acc + (((int64_t)b * p.Sq + (q_ok[qg] ? qrow[qg] : 0)) * p.H + head) * p.D;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = 16 * dt + 4 * g;
            const f32x4_t w = {O[dt][qg][0] * inv, O[dt][qg][1] * inv, O[dt][qg][2] * inv, O[dt][qg][3] * inv};
            if (q_ok[qg] && d0 < p.D) *(f32x4_t*)(oa + d0) = w;
          }
        }
      }
    }
    }  // (the 8-byte form: ring carry launches, which also write the fp32 state)
    if constexpr (STAMP) st_all[11] += __builtin_amdgcn_s_memtime();
  }  // pass
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no DMA may still be writing LDS when the wave ends
  if constexpr (STAMP) {  // [block][wave][16] u64; [7] = wave lifetime
    if (lane == 0 && p.mask != nullptr) {
      unsigned long long* d = (unsigned long long*)p.mask + ((size_t)blockIdx.x * 8 + wave) * 16;
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < 7; ++i) d[i] = st_all[i];
      d[7] = t_end - st_all[7];
#pragma unroll
      for (int i = 8; i < 12; ++i) d[i] = st_all[i];
    }
  }
#undef IC
}
