// Persistent form of gemm4w16_kernel with an OVERLAPPED epilogue (no residual operand).
//
// Measured anatomy of gemm4w16_kernel at K = 1024 (tools/gemm_stamps.py): prologue 3.0k, main loop 40k, epilogue
// 11.5k cycles, and the epilogue is HBM-write bound because all 256 CUs finish a tile together (32 MiB of output
// in ~7 us while every matrix pipe idles).  Here a workgroup walks its tiles in a loop and the stores of tile i
// trickle out during tile i+1:
//   * end of a tile: the 256 accumulator registers are read out, bias + activation applied, packed to 16 bits,
//     row-pair exchanged (v_permlane16_swap) into 32 x 16-byte chunks per lane held in VGPRs (four 32-dword
//     register arrays ob0..ob3, one per dword of a chunk); the next tile's first K-tile restarts them with C = 0;
//   * next tile: every K-tile issues SPK (1, 2 or 4) buffer_store_dwordx4 of the previous tile from its first
//     micro-steps, chunk index = kt*SPK + s read with VGPR-relative addressing (the index is wave-uniform).  Row /
//     column edges and "all chunks already sent" are handled by the buffer range check: such a lane gets an
//     out-of-range offset and its store is dropped -- no branches inside the MFMA stream.  Spreading the stores
//     over the whole tile matters: vmcnt retires in order, so a burst of stores in front of the prefetch loads
//     stalls the K loop on HBM write bandwidth (first version of this kernel: 32 stores in two K-tiles, no gain);
//   * vmcnt counts stores too, but stores are NOT retired in order with loads (a dropped or fast store leaves the
//     counter early; a first version that allowed "8 loads + the stores issued since" in flight read K-tiles that
//     had not landed).  The counted wait therefore stays vmcnt(8): with at most 8 operations of any kind in
//     flight, the loads still in flight are a suffix of at most 8 loads = the K-tile after the one being
//     retired.  Stores are issued from micro-steps 12..15, right after the 8 prefetch loads of the K-tile, so by
//     the next wait (8 micro-steps later) they have normally been acknowledged and the wait does not see them.
//   * the load stream never drains between tiles: the last three K-tiles of a tile prefetch K-tiles 0..2 of the
//     workgroup's NEXT tile (the LDS stage index runs on across tiles), so the only per-tile bubble left is the
//     ~4k-cycle accumulator read-out, during which those loads land.  The bias row of the tile is fetched (into
//     LDS) before those three K-tiles.
// Everything else (tile shape, LDS stages and swizzle, buffer_load-to-LDS prefetch, asm-owned accumulators,
// micro-step order) is gemm4w16_kernel's.  Requires K % 64 == 0 (an even number of K-tiles keeps the fragment
// double-buffer parity across tiles) and SPK * (K / 32) >= 32; no residual operand.
#pragma once
#include <type_traits>

#include "gemm4w16_kernel.h"

typedef __attribute__((ext_vector_type(32))) uint32_t u32x32_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

constexpr int G6P_BIAS_OFF = G6_SMEM;          // 4 waves x 256 B: each wave's slice of the tile's bias row
constexpr int G6P_SMEM = G6_SMEM + 4 * 256;

template <typename T, int ACT, int SPK, bool STAMP = false>
__global__ __launch_bounds__(256) void gemm4w16p_kernel(const GemmDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  static_assert(ACT != MIO_ACT_SWIGLU, "dual-B GEMM uses gemm_bias_act_kernel");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int c16 = lane & 15, g = lane >> 4;
  const int nk = p.K / G6_BK;
  const int ntiles = p.tiles_m * p.tiles_n;

  // fragment read offsets (tile independent)
  const int co = (g ^ g6_swz(c16)) * 16;
  const int xbase = (wr * 128 + c16) * 64 + co;
  const int wbase = G6_XT + (wc * 128 + c16) * 64 + co;

  // pending output of the previous tile: 32 chunks of 16 bytes per lane (dword c of chunk j = obc[j]) + where they go
  u32x32_t ob0, ob1, ob2, ob3;
#pragma unroll
  for (int j = 0; j < 32; ++j) ob0[j] = ob1[j] = ob2[j] = ob3[j] = 0u;
  __amdgpu_buffer_rsrc_t yrs_prev = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0, 0x00020000);
  // after the exchange a lane holds row (2*mtp + (g&1))*16 + c16 of the wave tile, columns 8*(g>>1) .. +7 of
  // 16-column tile nt: byte offset of chunk (mtp, nt) = yoff0 + mtp * 64*ldy + nt * 32
  // (GemmDev::y_blk: the tile's 8 K-tile blocks of the consumer are 16 KiB apart, a row is 64 B inside a block)
  const int yrow0 = wr * 128 + (g & 1) * 16 + c16;
  const int yoff0 = p.y_blk ? (wc * 4 << 14) + yrow0 * 64 + 16 * (g >> 1)
                            : (yrow0 * (int)p.ldy + wc * 128 + 8 * (g >> 1)) * 2;
  const int ystep = p.y_blk ? 32 * 64 : 64 * (int)p.ldy;
  int nchunks_prev = 0;    // 32 once a tile is pending (wave-uniform)
  bool full_prev = false;  // previous tile lies completely inside M x N: no per-lane masks (wave-uniform)
  int mrem_prev = 0;       // rows of the previous tile inside M
  int nvalid_prev = 0;     // chunks nt < nvalid_prev are inside N (per lane)

  X8 fx[2][8];
  X8 fw[2];

  // ---- operand addressing of the tile being computed / prefetched
  int xvo[4], wvo[4];
  __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t wrs = xrs;
  int64_t m0 = 0;
  int n0 = 0, nrem = 0;
  int64_t mrem = 0;
  // lane id recomputed from an opaque instruction wherever lane-constant addresses are built inside the tile loop:
  // hoisted to kernel entry they are spilled around the loop, and the reload's compiler-inserted vmcnt(0) drains the
  // hand-placed prefetch (cdna_hip_programming.md, attention pitfalls)
  auto lane_now = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto setup = [&](int tile) {
    int tm, tn;
    gemm_tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn);
    m0 = (int64_t)tm * 256;
    n0 = tn * 256;
    mrem = p.M - m0;
    nrem = p.N - n0;
    const int ln = lane_now(), prow = ln >> 2, pcs = ln & 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 16 + prow;
      const int kch = pcs ^ g6_swz(row);
      const int xr = (row < mrem) ? row : (int)(mrem - 1);
      const int wrw = (row < nrem) ? row : (nrem - 1);
      xvo[i] = (p.x_blk ? xr * 64 : xr * (int)p.ldx * 2) + 16 * kch;
      wvo[i] = (p.w_blk ? wrw * 64 : wrw * (int)p.ldw * 2) + 16 * kch;
    }
    xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.x_blk ? (void*)((const char*)p.x + ((int64_t)tm * nk << 14)) : (void*)((const T*)p.x + m0 * p.ldx), 0, 0x7fffffff,
        0x00020000);
    wrs = __builtin_amdgcn_make_buffer_rsrc(
        p.w_blk ? (void*)((const char*)p.w + ((int64_t)tn * nk << 14)) : (void*)((const T*)p.w + (int64_t)n0 * p.ldw), 0,
        0x7fffffff, 0x00020000);
  };

#define IC(N) std::integral_constant<int, N>{}
  const int krot = ((int)(blockIdx.x & 7) * nk) >> 3;  // XCD = blockIdx.x % 8
  const int xkstep = p.x_blk ? 16384 : G6_BK * 2;  // bytes from one K-tile of x / w to the next (GemmDev::x_blk, w_blk)
  const int wkstep = p.w_blk ? 16384 : G6_BK * 2;
  int sbase = 0;  // LDS stage of the current tile's K-tile 0 (the stage index runs on across tiles)
  // request piece (i, which) of K-tile `kl` of the tile `setup` describes into stage (sbase + ks) & 3
  auto issue_one = [&](int ks, int kl, auto I, auto WHICH) {
    constexpr int i = decltype(I)::value, which = decltype(WHICH)::value;
    char* dst = smem + ((sbase + ks) & (G6_STAGES - 1)) * G6_BUF + which * G6_XT + (wave * 4 + i) * 1024;
    // K-tiles are visited in a rotated order, a different rotation per XCD (the sum over k does not care): with a
    // row stride of 8 KiB (K = 4096) every workgroup on the chip otherwise asks the same few memory channels for the
    // same k at the same time
    const int kr = kl + krot;
    const int koff = (kr >= nk ? kr - nk : kr) * (which ? wkstep : xkstep);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? wrs : xrs, (MIO_LDS void*)dst, 16, which ? wvo[i] : xvo[i],
                                             koff, 0, 0);
  };
  // K-tiles 0, 1, 2 of the current `setup` in flight (24 loads per lane)
  auto issue_prologue = [&]() {
#define G6P_ISSUE_TILE(KT)                                                                                  \
  issue_one(KT, KT, IC(0), IC(0)); issue_one(KT, KT, IC(0), IC(1)); issue_one(KT, KT, IC(1), IC(0));       \
  issue_one(KT, KT, IC(1), IC(1)); issue_one(KT, KT, IC(2), IC(0)); issue_one(KT, KT, IC(2), IC(1));       \
  issue_one(KT, KT, IC(3), IC(0)); issue_one(KT, KT, IC(3), IC(1));
    G6P_ISSUE_TILE(0)
    G6P_ISSUE_TILE(1)
    G6P_ISSUE_TILE(2)
#undef G6P_ISSUE_TILE
  };
  auto read_x = [&](const char* buf, auto RB, auto MT) {
    constexpr int rb = decltype(RB)::value, mt = decltype(MT)::value;
    fx[rb][mt] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + xbase + mt * 16 * 64));
  };
  auto read_w = [&](const char* buf, auto NT) {
    constexpr int nt = decltype(NT)::value;
    fw[nt & 1] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + wbase + nt * 16 * 64));
  };
  // FIRST: the first K-tile of an output tile starts the accumulators (C = 0): no zeroing pass over the file
  auto mfma4 = [&](auto RB, auto J, auto FIRST_) {
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
  // store chunk `chunk` (wave-uniform; = mtp*8 + nt) of the previous tile; scalar branches only.  Three phases so
  // that each fits the shadow of one 4-MFMA micro-step: address, VGPR-relative read of the chunk, store.
  int st_off = 0;
  u32x4_t st_d = {0u, 0u, 0u, 0u};
  auto store_phase = [&](int chunk, auto PH) {
    constexpr int ph = decltype(PH)::value;
    if (chunk < nchunks_prev) {
      const int mtp = chunk >> 3, nt = chunk & 7;
      if constexpr (ph == 0 || ph == 3) {
        int off = yoff0 + mtp * ystep;
        if (!full_prev) {
          const bool ok = (nt < nvalid_prev) && (yrow0 + 32 * mtp < mrem_prev);
          off = ok ? off : 0x7fffffff;  // out of range: dropped by the buffer range check
        }
        st_off = off;
      }
      if constexpr (ph == 1 || ph == 3) st_d = (u32x4_t){ob0[chunk], ob1[chunk], ob2[chunk], ob3[chunk]};
      if constexpr (ph == 2 || ph == 3)
        __builtin_amdgcn_raw_buffer_store_b128(st_d, yrs_prev, st_off, p.y_blk ? ((nt >> 1) << 14) + ((nt & 1) << 5) : nt * 32, 0);
    }
  };
  // One K-tile (see gemm4w16_kernel) + SPK stores of the previous tile (SPK = 1: phases in micro-steps 12, 13, 14;
  // otherwise one whole store in each of micro-steps 12..12+SPK-1).  TAIL: the K-tile prefetched three ahead lies
  // past this tile -> K-tile kt+3-nk of the tile `setup` now describes (the next tile, or this one again when the
  // workgroup has no next tile: harmless, never read).
  auto ktile = [&](auto RBv, int kt, auto TAILv) {  // TAILv: 0 plain, 1 tail, 2 first K-tile of the output tile
    using RB = decltype(RBv);
    using NRB = std::integral_constant<int, RB::value ^ 1>;
    constexpr bool TAIL = decltype(TAILv)::value == 1;
    using FIRST = std::integral_constant<int, decltype(TAILv)::value == 2 ? 1 : 0>;
    const char* buf = smem + ((sbase + kt) & (G6_STAGES - 1)) * G6_BUF;
    const char* nbuf = smem + ((sbase + kt + 1) & (G6_STAGES - 1)) * G6_BUF;
    const int kl = TAIL ? kt + 3 - nk : kt + 3;
#define G6P_STEP(J)                                                                        \
    mfma4(RB{}, IC(J), FIRST{});                                                           \
    if constexpr ((J) % 2 == 0 && (J) / 2 + 1 < 8) read_w(buf, IC((J) / 2 + 1));          \
    if constexpr (SPK == 1) {                                                              \
      if constexpr ((J) >= 12 && (J) <= 14) store_phase(kt, IC((J) - 12));                 \
    } else {                                                                               \
      if constexpr ((J) >= 12 && (J) - 12 < SPK) store_phase(kt * SPK + (J) - 12, IC(3)); \
    }
    G6P_STEP(0) __builtin_amdgcn_sched_barrier(0);
    G6P_STEP(1) __builtin_amdgcn_sched_barrier(0);
    G6P_STEP(2) __builtin_amdgcn_sched_barrier(0);
    G6P_STEP(3) __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#define G6P_STEP_B(J, P)                                                                   \
    G6P_STEP(J)                                                                            \
    if constexpr ((P) < 8) {                                                               \
      issue_one(kt + 3, kl, IC((P) / 2), IC((P) % 2));                                     \
      read_x(nbuf, NRB{}, IC((P) < 8 ? (P) : 0));                                          \
    }                                                                                      \
    __builtin_amdgcn_sched_barrier(0);
    G6P_STEP_B(4, 0) G6P_STEP_B(5, 1) G6P_STEP_B(6, 2) G6P_STEP_B(7, 3) G6P_STEP_B(8, 4) G6P_STEP_B(9, 5)
    G6P_STEP_B(10, 6) G6P_STEP_B(11, 7) G6P_STEP_B(12, 8) G6P_STEP_B(13, 8)
    G6P_STEP(14)
    read_w(nbuf, IC(0));
    __builtin_amdgcn_sched_barrier(0);
    G6P_STEP(15) __builtin_amdgcn_sched_barrier(0);
#undef G6P_STEP_B
#undef G6P_STEP
  };

  int tile = blockIdx.x;  // the launcher keeps gridDim.x <= ntiles
  setup(tile);
  issue_prologue();
  // K-tile 0 has landed (at most K-tiles 1 and 2 = 16 loads still in flight) for every wave
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int t = 0; t < 8; ++t) fx[0][t] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + xbase + t * 16 * 64));
  fw[0] = __builtin_bit_cast(X8, *(const u32x4_t*)(smem + wbase));
  __builtin_amdgcn_sched_barrier(0);
  for (;;) {
    unsigned long long st0 = 0, st1 = 0, st2 = 0, sr0 = 0;
    if constexpr (STAMP) {
      st0 = st1 = __builtin_amdgcn_s_memtime();
      sr0 = __builtin_amdgcn_s_memrealtime();
    }
    // K-tiles 0 .. nk-4 prefetch inside this tile (nk is even and >= 8)
    ktile(IC(0), 0, IC(2));
    ktile(IC(1), 1, IC(0));
    int kt = 2;
    for (; kt < nk - 4; kt += 2) {
      ktile(IC(0), kt, IC(0));
      ktile(IC(1), kt + 1, IC(0));
    }
    ktile(IC(0), kt, IC(0));  // kt = nk - 4: requests this tile's last K-tile

    // ---- this tile's output coordinates and bias row; from here on `setup` describes the next tile
    const int64_t om0 = m0;
    const int on0 = n0, onrem = nrem;
    const int omrem = mrem < 256 ? (int)mrem : 256;
    // bias: each wave DMAs the 256 B slice of the bias row under its own 128 columns into its private LDS slot (one
    // global_load_lds_dword: 64 lanes x 4 B) and reads it back in the read-out.  Held in registers from here the 16
    // values do not fit beside the output buffers and fragments: hipcc spills them with a vmcnt(0) in front of the
    // tail K-tiles.  The request is older than the 24 loads the tail issues, so the counted waits there cover it.
    if (p.bias != nullptr) {
      const int ln = lane_now();
      const int n = on0 + wc * 128 + 2 * ln;
      const int nc = n < p.N - 2 ? n : p.N - 2;  // columns past N are never stored
      const uint32_t lds = (uint32_t)(size_t)((MIO_LDS char*)(smem + G6P_BIAS_OFF + wave * 256));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2"
                   :
                   : "s"(lds), "v"(nc * 2), "s"(p.bias)
                   : "memory", "m0");
    }
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < ntiles;
    if (has_next) setup(next);
    __builtin_amdgcn_sched_barrier(0);
    ktile(IC(1), kt + 1, IC(1));
    ktile(IC(0), kt + 2, IC(1));
    ktile(IC(1), kt + 3, IC(1));
    sbase = (sbase + nk) & (G6_STAGES - 1);
    if constexpr (STAMP) st2 = __builtin_amdgcn_s_memtime();

    // ---- read-out: accumulators -> bias/activation -> 16-bit -> row-pair exchange -> ob0..3
    // (the last MFMAs must have retired before the accumulator file is read: no interlock for asm readers)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    {
      u32x2_t bq[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) bq[nt] = (u32x2_t){0u, 0u};
      if (p.bias != nullptr) {
        const char* bl = smem + G6P_BIAS_OFF + wave * 256 + 8 * (lane_now() >> 4);
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) bq[nt] = *(const u32x2_t*)(bl + nt * 32);
      }
      // column scale: wave-uniform (the wave's 128 columns lie inside or outside [cs_lo, cs_hi))
      const int ncol0 = on0 + wc * 128;
      const float csv = (ncol0 >= p.cs_lo && ncol0 < p.cs_hi) ? p.cs_val : 1.0f;
      const f32x2_t cs2 = {csv, csv};
      auto pair = [&](auto MTP) {
        constexpr int mtp = decltype(MTP)::value, mt = 2 * mtp;
        auto col = [&](auto NT) {
          constexpr int nt = decltype(NT)::value;
          const X4 bv = __builtin_bit_cast(X4, bq[nt]);
          const f32x4_t a = G6AccIO<nt * 8 + mt>::read();
          const f32x4_t b = G6AccIO<nt * 8 + mt + 1>::read();
          const f32x2_t b01 = {(float)bv[0], (float)bv[1]}, b23 = {(float)bv[2], (float)bv[3]};
          const f32x2_t a01 = gemm_act2<ACT>((f32x2_t){a[0], a[1]} + b01) * cs2, a23 = gemm_act2<ACT>((f32x2_t){a[2], a[3]} + b23) * cs2;
          const f32x2_t c01 = gemm_act2<ACT>((f32x2_t){b[0], b[1]} + b01) * cs2, c23 = gemm_act2<ACT>((f32x2_t){b[2], b[3]} + b23) * cs2;
          const auto s0 = __builtin_amdgcn_permlane16_swap(pack2<T>(a01[0], a01[1]), pack2<T>(c01[0], c01[1]), false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pack2<T>(a23[0], a23[1]), pack2<T>(c23[0], c23[1]), false, false);
          ob0[mtp * 8 + nt] = s0[0];
          ob1[mtp * 8 + nt] = s1[0];
          ob2[mtp * 8 + nt] = s0[1];
          ob3[mtp * 8 + nt] = s1[1];
        };
        col(IC(0)); col(IC(1)); col(IC(2)); col(IC(3)); col(IC(4)); col(IC(5)); col(IC(6)); col(IC(7));
      };
      pair(IC(0)); pair(IC(1)); pair(IC(2)); pair(IC(3));
    }
    if constexpr (STAMP) {  // per (tile, wave): tile start, loop start, loop end, read-out end; 100 MHz start / end
      const unsigned long long st3 = __builtin_amdgcn_s_memtime();
      const unsigned long long sr1 = __builtin_amdgcn_s_memrealtime();
      if (lane == 0 && p.dbg != nullptr) {
        unsigned long long* d = p.dbg + ((size_t)tile * 4 + wave) * 8;
        d[0] = st0; d[1] = st1; d[2] = st2; d[3] = st3; d[4] = sr0; d[5] = sr1; d[6] = blockIdx.x; d[7] = 0;
      }
    }
    // where this tile's chunks go
    {  // (wave-uniform 64-bit offset, forced into scalar registers: as vector values hipcc computes it before the tail
       //  K-tiles and spills it around them)
      const int64_t yb = p.y_blk ? (((om0 >> 8) * (int64_t)(p.N >> 5) + (on0 >> 5)) << 14) : (om0 * p.ldy + on0) * 2;
      const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)yb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(yb >> 32));
      yrs_prev = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.y + (((uint64_t)hi << 32) | lo)), 0, 0x7fffffff, 0x00020000);
    }
    nchunks_prev = 32;
    full_prev = (omrem == 256) && (onrem >= 256);
    mrem_prev = omrem;
    {
      const int c0 = wc * 128 + 8 * (g >> 1);  // first column of this lane's chunk in tile nt = 0
      int nv = (onrem - c0 + 15) / 16;
      nvalid_prev = nv < 0 ? 0 : (nv > 8 ? 8 : nv);
    }
    if (!has_next) break;
    tile = next;
  }

  // ---- flush the last tile's output
  {
    auto flush = [&](auto self, auto Jc) {
      constexpr int j = decltype(Jc)::value;
      constexpr int mtp = j / 8, nt = j % 8;
      const bool ok = (nt < nvalid_prev) && (yrow0 + 32 * mtp < mrem_prev);
      const int off = ok ? yoff0 + mtp * ystep : 0x7fffffff;
      const u32x4_t d = {ob0[j], ob1[j], ob2[j], ob3[j]};
      __builtin_amdgcn_raw_buffer_store_b128(d, yrs_prev, off, p.y_blk ? ((nt >> 1) << 14) + ((nt & 1) << 5) : nt * 32, 0);
      if constexpr (j + 1 < 32) self(self, IC(j + 1));
    };
    flush(flush, IC(0));
  }
#undef IC
}
