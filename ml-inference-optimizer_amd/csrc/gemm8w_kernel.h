// 256x256 bf16/fp16 GEMM on v_mfma_f32_16x16x32 with EIGHT waves per workgroup: two waves on every SIMD, run as a
// ping-pong -- while one wave of a SIMD issues its MFMAs the other reads its next fragments from LDS and requests the
// K-tile three ahead (cdna_hip_programming.md, "The 256^2 8-phase template"; MI355X_MICROARCH.md, "Two waves per SIMD").
// Persistent: a workgroup walks its output tiles and the load stream runs on across tiles.
//
// Why (round-3 measurements, tools/gemm8_ksweep.py, M 32768 x N 3072, blocked operands): the one-wave-per-SIMD kernels pay
// issue time for everything that is not an MFMA.  Per K-tile of 32 and workgroup, zero operands (clock 2.39 GHz) / random:
//   gemm4w16_kernel  555 / 696 ns     gemm4w16p_kernel (stores trickled through the K loop)  618 / 750 ns
//   this loop        520 / 678 ns     (1024 matrix-pipe cycles = 428 ns)
// and timing-only ablations of this loop: without its LDS-DMA requests 436 ns, without its fragment reads unchanged, with
// every request re-reading L1-resident bytes 444 ns, without the counted wait unchanged: what is left between the loop and
// the matrix pipe is the L2 -> CU fetch rate (32 KiB per K-tile), not instruction issue and not latency; blocked (contiguous
// 1-KiB) pieces are fetched 17 % faster than 64-byte row pieces.  Per tile the one-tile kernels lose another 5.6 - 7.6 us
// (workgroup launch, first tiles' latency, read-out, store drain): at K = 1024 that is 25 % of the time, hence persistent.
//
// Structure (operand formats are gemm4w16_kernel's: K-tile = 32, four 32-KiB LDS stages X[256][32] + W[256][32], 64-byte
// rows with the g6_swz chunk swizzle on the DMA source address, blocked or row-major operands):
//   * wave w: group = w >> 2 (waves w and w + 4 share a SIMD), column quarter = w & 3; wave tile = 128 rows (the
//     group's half of the tile) x 64 columns = 8 x 4 accumulator tiles of 16x16 (128 registers).  Weight tile = MFMA A
//     operand, activation tile = B operand: a lane holds ONE output row and 4 consecutive columns (lane-local epilogue).
//   * a K-tile is two SEGMENTS per wave, separated by s_barrier: [12 fragment reads + 4 DMA pieces of the K-tile three
//     ahead + counted wait] | [32 MFMAs].  Group 1 runs one barrier behind group 0 (one extra s_barrier in front of its
//     loop, one behind group 0's), so at any time one wave of a SIMD is in its MFMA segment and the other in its read
//     segment.  (Two phases of 16 MFMAs per K-tile, the guide's template granularity, measured 2.5 % slower.)
//   * hazards, counted in a wave's own segments (group 1's segment s runs during group 0's segment s + 1):
//       RAW  a K-tile's pieces are waited for (counted vmcnt, own pieces) in a read segment L and first read in read
//            segment L + 2: every wave of both groups has passed its wait and a barrier behind it by then;
//       WAR  a read segment ends with lgkmcnt(0) IN FRONT of its barrier, so a stage is restaged from the second
//            segment after its last read (the K-tile three ahead reuses the stage of the K-tile before this one).
//     Three K-tiles stay in flight (vmcnt(8) = the two youngest K-tiles of 4 pieces per wave).
//   * tile boundary: the last three K-tiles of a tile request K-tiles 0..2 of the workgroup's next tile (the stage index
//     runs on); the last read segment waits vmcnt(4) (next K-tiles 0 AND 1 landed), so the next tile's first read segment
//     needs no wait.  The first K-tile of a tile multiplies with C = 0: no zeroing pass.
//   * output: the read-out (bias / activation / column scale / residual, v_permlane16_swap row-pair exchange) stores
//     16 chunks of 16 bytes per lane at the tile boundary.  The CU's store path takes 16 bytes per clock (73 cycles per
//     buffer_store_dwordx4 wave-instruction whatever its address pattern): 8.2k cycles per 256x256 tile, 4.7k per group,
//     the two groups one after the other, against a 34k-cycle K loop at K = 1024.  Trickling the stores through the read
//     segments of the next tile's K loop (output parked in 32..64 registers, read back by wave-uniform index) was built
//     and measured SLOWER (QKV 0.191 ms against 0.167 for the burst; with the trickled stores' data dropped by the range
//     check still 0.181, with no trickled store at all 0.161): a fifth vector-memory instruction in a read segment whose
//     four LDS-DMA requests already take ~400 of the partner's 535 MFMA cycles queues behind every wave's loads.
//   * store data must outlive the store: `buffer_store_dwordx4 v[88:91], v93, s[16:19], s0 offen` followed at once by a
//     VALU write of v88/v89 (hipcc reuses the dead registers for the next chunk) reached memory with the NEW contents in
//     lanes 12..15 / 44..47 of dword 1 (round 3, MI355X, ROCm 7.2: deterministic, tools/dbg/gemm8_dump.py; LLVM assumes
//     the register-soffset form has no such hazard).  Every stored chunk is therefore kept live (an empty asm use) until
//     all 16 stores of the read-out have been issued and 64 wait states have passed.
//   * activation fragments rotate: the read segment fetches the weight fragments and row tiles 0..3; row tile t + 4 is
//     read into tile t's registers right behind tile t's four MFMAs (its pieces were requested by this wave's own group,
//     so the stage is not restaged under those reads): 32 fragment registers instead of 48 pay for the parked output.
//   * GLU (SwiGLU, reference kernels/triton/mlp_kernels.py:417-641): the weight tile interleaves 32 gate rows and 32 up
//     rows per wave (mio_weight_block_glu), so accumulator columns 0..31 / 32..63 of a wave are gate / up of the SAME 32
//     output columns and silu(gate) * up is lane-local; the output tile is 256 x 128.
#pragma once
#include <type_traits>

#include "gemm4w16_kernel.h"

constexpr int G8_THREADS = 512;
constexpr int G8_BIAS_OFF = G6_SMEM;  // 8 waves x 256 B: each wave's slice of the bias row(s)
constexpr int G8_SMEM = G6_SMEM + 8 * 256;
// LayerNorm fold (FOLD != 0): per-wave rstd table, the row-statistics region (consumer: [group][slot <= 8][128 rows]
// (sum, sum of squares); producer: [group][column quarter][128 rows]) and the producer's arrival counters
constexpr int G8_RS_OFF = G8_SMEM;  // consumer: per wave, rstd of its group's 128 rows (512 B)
constexpr int G8_STAT_OFF = G8_RS_OFF + 8 * 512;
constexpr int G8_CNT_OFF = G8_STAT_OFF + 2 * 8192;
constexpr int G8_SMEM_FOLD = G8_CNT_OFF + 64;
static_assert(GEMM_LN_SLOTS_MAX * 1024 <= 8192, "a group's statistics region holds GEMM_LN_SLOTS_MAX slots of 128 rows x 8 bytes");

// VAR (timing-only ablations, diagnostic library): 4 = no prefetch issue in the loop, 16 = no counted wait,
// 64 = every piece re-reads K-tile 0, 128 = in-kernel stamps (tools/gemm8_stamps.py), 256 = scalar activation math,
// 2048 = all eight activation fragments read in the read segment
// FOLD (GemmDev, "LayerNorm folded into the GEMMs on either side of it"): 1 = consumer (the weights are centred, so the read-out
// normalises with acc * rstd + bias; row statistics from p.ln_stats), 2 = producer (the read-out also writes the rounded output rows' (sum, sum
// of squares) per tile column: the four column-quarter waves of a group leave their partials in LDS, the last one to arrive adds
// them in a fixed order -- deterministic -- and stores 128 rows x 8 bytes)
template <typename T, int ACT, bool RES, int VAR = 0, int FOLD = 0>
__global__ __launch_bounds__(G8_THREADS) void gemm8w_kernel(const GemmDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  constexpr bool GLU = (ACT == MIO_ACT_SWIGLU);
  constexpr bool SCALAR_ACT = (VAR & 256) != 0;  // gemm_kernel.h gemm_act2 (ablation)
  constexpr int BN = GLU ? 128 : 256;  // output columns per tile
  constexpr int WN = BN / 4;           // output columns per wave
  constexpr int ONT = WN / 16;         // output column tiles per wave
  constexpr int NCH = 4 * ONT;         // 16-byte output chunks per lane and tile: chunk j = (row pair j / ONT, column tile j % ONT)
  static_assert(!(GLU && RES), "no residual on the gated stage");
  static_assert(FOLD != 2 || !GLU, "the gated stage has no residual epilogue, hence no row statistics");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wn = wave & 3;
  const int c16 = lane & 15, g = lane >> 4;
  const int nk = p.K / G6_BK;
  const int ntiles = p.tiles_m * p.tiles_n;

  // ---- operand addressing of the tile being computed / prefetched (gemm4w16p_kernel's, 2 + 2 pieces per wave)
  int xvo[2], wvo[2];
  __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
  __amdgpu_buffer_rsrc_t wrs = xrs;
  int64_t m0 = 0, mrem = 0;
  int n0 = 0, nrem = 0;  // in OUTPUT columns
  // lane id recomputed from an opaque instruction wherever lane-constant addresses are built inside the tile loop: hoisted
  // out of it (LICM) they are kept live and spilled (gemm4w16p_kernel.h)
  auto lane_now = [&]() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto setup = [&](int tile) {
    int tm, tn;
    gemm_tile_coords(tile, p.tiles_m, p.tiles_n, tm, tn, p.group_m > 0 ? p.group_m : 4);  // 4: measured end to end
    m0 = (int64_t)tm * 256;
    n0 = tn * BN;
    mrem = p.M - m0;
    nrem = p.N - n0;
    const int ln = lane_now(), prow = ln >> 2, pcs = ln & 3;
    const int wrows = GLU ? 256 : nrem;  // GLU weights are blocked and padded: every row of the tile exists
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 16 + prow;  // 0..255: waves 0..3 request rows 0..127, waves 4..7 rows 128..255
      const int kch = pcs ^ g6_swz(row);
      const int xr = (row < mrem) ? row : (int)(mrem - 1);
      const int wrw = (row < wrows) ? row : (wrows - 1);
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
  const int xkstep = p.x_blk ? 16384 : G6_BK * 2;
  const int wkstep = p.w_blk ? 16384 : G6_BK * 2;

#define IC(N) std::integral_constant<int, N>{}
  int sbase = 0;  // LDS stage of the current tile's K-tile 0 (the stage index runs on across tiles)
  // piece i (0 / 1) of X (which = 0) or W (which = 1): K-tile `kl` of the tile `setup` describes into stage (sbase + ks) & 3
  auto issue_one = [&](int ks, int kl, auto I, auto WHICH) {
    constexpr int i = decltype(I)::value, which = decltype(WHICH)::value;
    char* dst = smem + ((sbase + ks) & (G6_STAGES - 1)) * G6_BUF + which * G6_XT + (wave * 2 + i) * 1024;
    int koff = kl * (which ? wkstep : xkstep);
    if constexpr ((VAR & 64) != 0) koff = 0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(which ? wrs : xrs, (MIO_LDS void*)dst, 16, which ? wvo[i] : xvo[i], koff, 0, 0);
  };
  auto issue_tile = [&](int ks, int kl) {
    issue_one(ks, kl, IC(0), IC(0));
    issue_one(ks, kl, IC(0), IC(1));
    issue_one(ks, kl, IC(1), IC(0));
    issue_one(ks, kl, IC(1), IC(1));
  };

  const int co = (g ^ g6_swz(c16)) * 16;
  const int xbase = (grp * 128 + c16) * 64 + co;
  const int wbase = G6_XT + (wn * 64 + c16) * 64 + co;

  constexpr bool ROT = (VAR & 2048) == 0;  // (ablation 2048: all eight activation fragments read in the read segment)
  X8 fx[ROT ? 4 : 8];
  X8 fw[4];
  f32x4_t acc[8][4];

  const int ystep = p.y_blk ? 32 * 64 : 64 * (int)p.ldy;  // bytes per row pair (32 rows)
  const int ychi = p.y_blk ? 16384 : 64;  // bytes from column tile nt to nt + 2 (blocked: the next 32-column block)

  // one K-tile: kt = index inside the current tile (stage (sbase + kt) & 3), kl = the K-tile requested three ahead (of the
  // tile `setup` describes), WAIT = the counted wait of the read segment (8: the next K-tile has landed; 4: the next two)
  // FOLD 1: the row statistics of the group's 128 rows of the CURRENT tile (wave wn requests rows 32 wn .. 32 wn + 31 of every slot:
  // 256 contiguous bytes per slot; 4 requests for <= 4 slots, else 8 -- a fixed count, so the counted waits stay immediates;
  // surplus requests repeat the last slot).  Requested in K-tile 1's read segment BEHIND that K-tile's operand requests: every
  // wave has left the previous tile's read-out by then (K-tile 0's barriers: the region is free), and K-tiles 1 and 2 wait with
  // the request count added (the statistics are the youngest entries behind K-tile 4's pieces, so vmcnt(8 + n) still means "the
  // next K-tile has landed"); from K-tile 3 on they are two K-tiles old and the plain vmcnt(8) covers them.  (Requested behind
  // K-tile 0 with plain waits, K-tile 1's vmcnt(8) also drained K-tile 3's just-issued pieces: one memory latency per tile, QKV
  // 171.6 us; in front of the tail: 176.5.)
  auto issue_stats = [&]() {
    const int l = lane_now();
    const int64_t fm0 = m0 + grp * 128;
    const int64_t mpad = (int64_t)p.tiles_m * 256;
    const float* ssrc = p.ln_stats + (fm0 + 32 * wn) * 2 + l;
    const uint32_t slds = (uint32_t)(size_t)((MIO_LDS char*)(smem + G8_STAT_OFF + grp * 8192 + wn * 256));
    const int nreq = p.ln_slots <= 4 ? 4 : 8;
    for (int s = 0; s < nreq; ++s) {
      const int ss = s < p.ln_slots ? s : p.ln_slots - 1;
      const float* src = ssrc + (int64_t)ss * mpad * 2;
      const uint32_t dst = slds + ss * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(dst), "v"(src) : "memory", "m0");
    }
  };
  // EXTRA (FOLD 1): 1 = request the row statistics behind this K-tile's operand requests, 1 / 2 = count them in the wait
  auto ktile = [&](int kt, int kl, auto FIRST, auto WAIT, auto EXTRA) {
    const char* buf = smem + ((sbase + kt) & (G6_STAGES - 1)) * G6_BUF;
    // -- read segment
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fw[nt] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + wbase + nt * 1024));
#pragma unroll
    for (int t = 0; t < (ROT ? 4 : 8); ++t) fx[t] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + xbase + t * 1024));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((VAR & 4) == 0) issue_tile(kt + 3, kl);
    if constexpr (decltype(EXTRA)::value == 1) issue_stats();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (decltype(EXTRA)::value != 0) {
      if (p.ln_slots <= 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    if constexpr ((VAR & 16) == 0 && decltype(WAIT)::value == 8 && decltype(EXTRA)::value == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if constexpr ((VAR & 16) == 0 && decltype(WAIT)::value == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read has returned before the barrier (WAR rule above)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // -- MFMA segment: row tile t's four MFMAs, then row tile t + 4 into its fragment registers
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if constexpr (decltype(FIRST)::value != 0)
          acc[t][nt] = DT<T>::mfma16(fw[nt], fx[ROT ? (t & 3) : t], (f32x4_t){0.f, 0.f, 0.f, 0.f});
        else
          acc[t][nt] = DT<T>::mfma16(fw[nt], fx[ROT ? (t & 3) : t], acc[t][nt]);
      }
      if (ROT && t < 4) {
        __builtin_amdgcn_sched_barrier(0);
        fx[t] = __builtin_bit_cast(X8, *(const u32x4_t*)(buf + xbase + (t + 4) * 1024));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  if constexpr (FOLD == 2) {  // arrival counters (LDS is not cleared between workgroups); the prologue's barrier publishes them
    if (tid < 2) ((MIO_LDS unsigned*)(smem + G8_CNT_OFF))[tid] = 0u;
  }
  int tile = blockIdx.x;  // the launcher keeps gridDim.x <= ntiles
  setup(tile);
  issue_tile(0, 0);
  issue_tile(1, 1);
  issue_tile(2, 2);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // K-tiles 0 and 1 landed for this wave
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one segment behind group 0
  __builtin_amdgcn_sched_barrier(0);

  for (;;) {
    unsigned long long st[6] = {0, 0, 0, 0, 0, 0}, sr0 = 0;
    if constexpr ((VAR & 128) != 0) {
      st[0] = __builtin_amdgcn_s_memtime();
      sr0 = __builtin_amdgcn_s_memrealtime();
    }
    // K-tiles 0 .. nk-4 prefetch inside this tile (nk >= 4)
    ktile(0, 3, IC(1), IC(0), IC(0));
    if constexpr ((VAR & 128) != 0) st[1] = __builtin_amdgcn_s_memtime();
    if constexpr (FOLD == 1) {  // (nk >= 8: the fold needs K % 256 == 0)
      ktile(1, 4, IC(0), IC(8), IC(1));
      ktile(2, 5, IC(0), IC(8), IC(2));
      for (int kt = 3; kt < nk - 3; ++kt) ktile(kt, kt + 3, IC(0), IC(8), IC(0));
    } else {
      for (int kt = 1; kt < nk - 3; ++kt) ktile(kt, kt + 3, IC(0), IC(8), IC(0));
    }
    if constexpr ((VAR & 128) != 0) st[3] = __builtin_amdgcn_s_memtime();

    // ---- this wave's output coordinates; from here on `setup` describes the next tile
    const int64_t om0 = m0 + grp * 128;
    const int on0 = n0 + wn * WN;
    const int64_t omrem = mrem - grp * 128;  // rows of the wave tile inside M (may be <= 0)
    const int onrem = nrem - wn * WN;
    // bias: each wave DMAs the slice of the bias row(s) under its own columns into its private LDS slot and reads it back
    // in the read-out (held in registers across the tail K-tiles hipcc would drain the prefetch in front of the first use).
    // Slot dwords 0..31 = the wave's 64 accumulator columns; GLU: 0..15 gate bias (bias_g), 16..31 up bias (bias) of the
    // wave's 32 output columns.  The request is older than the 12 loads the tail issues: their counted waits cover it.
    const bool have_bias = GLU ? (p.bias != nullptr || p.bias_g != nullptr) : (p.bias != nullptr);
    if (have_bias) {
      const int l = lane_now() & 31, lg = GLU ? (l & 15) : l;
      const int n = on0 + 2 * lg;
      const int nc = n < p.N - 2 ? n : p.N - 2;  // columns past N are never stored
      const T* base = (const T*)p.bias;
      if constexpr (GLU) {
        base = (l < 16) ? (const T*)p.bias_g : (const T*)p.bias;
        if (base == nullptr) base = (const T*)(p.bias ? p.bias : p.bias_g);  // (that half is ignored by the read-out)
      }
      const T* src = base + nc;
      const uint32_t lds = (uint32_t)(size_t)((MIO_LDS char*)(smem + G8_BIAS_OFF + wave * 256));
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(lds), "v"(src) : "memory", "m0");
    }
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < ntiles;
    if (has_next) setup(next);  // (no next tile: K-tiles 0..2 of this one are fetched again into idle stages, never read)
    __builtin_amdgcn_sched_barrier(0);
    ktile(nk - 3, 0, IC(0), IC(8), IC(0));
    ktile(nk - 2, 1, IC(0), IC(8), IC(0));
    ktile(nk - 1, 2, IC(0), IC(4), IC(0));
    sbase = (sbase + nk) & (G6_STAGES - 1);
    if constexpr ((VAR & 128) != 0) st[4] = __builtin_amdgcn_s_memtime();

    // ---- read-out: bias / activation / column scale (/ residual), row-pair exchange, 16-byte stores.
    // Residual loads and stores are buffer instructions on a per-tile descriptor (base = the wave tile's first element): a 32-bit lane offset fixed per tile, the row-pair
    // step in the scalar offset, the column tile in the immediate, and rows / columns past M / N get an out-of-range
    // offset that the range check drops -- no branches, no 64-bit math.
    {
      const int ln = lane_now(), c16 = ln & 15, g = ln >> 4;  // (shadows the kernel-scope values on purpose)
      u32x2_t bq[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bq[nt] = (u32x2_t){0u, 0u};
      if (have_bias) {
        const char* bl = smem + G8_BIAS_OFF + wave * 256 + 8 * g;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bq[nt] = *(const u32x2_t*)(bl + nt * 32);
        if constexpr (GLU) {
          if (p.bias_g == nullptr) bq[0] = bq[1] = (u32x2_t){0u, 0u};
          if (p.bias == nullptr) bq[2] = bq[3] = (u32x2_t){0u, 0u};
        }
      }
      const int ncs = on0 & ~127;  // column scale: a wave's columns lie inside one 128-column unit
      const float csv = (ncs >= p.cs_lo && ncs < p.cs_hi) ? p.cs_val : 1.0f;
      const f32x2_t cs2 = {csv, csv};
      const bool full = (omrem >= 128) && (onrem >= WN);  // wave-uniform
      const int rS = c16 + 16 * (g & 1);  // the row (inside a row pair) this lane stores after the exchange
      // output: y_blk = blocked activation layout of the consuming GEMM ((256-row, 32-column) blocks of 16 KiB)
      const int64_t yb = p.y_blk ? ((((om0 >> 8) * (int64_t)(p.N >> 5) + (on0 >> 5)) << 14) + (om0 & 255) * 64) : (om0 * p.ldy + on0) * 2;
      const uint32_t ylo = __builtin_amdgcn_readfirstlane((uint32_t)yb), yhi = __builtin_amdgcn_readfirstlane((uint32_t)(yb >> 32));
      const __amdgpu_buffer_rsrc_t yrs =
          __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.y + (((uint64_t)yhi << 32) | ylo)), 0, 0x7fffffff, 0x00020000);
      const int yvo = p.y_blk ? rS * 64 + 16 * (g >> 1) : (rS * (int)p.ldy + 8 * (g >> 1)) * 2;
      __amdgpu_buffer_rsrc_t rrs = yrs;
      int rvo = 0, rstep = 0;
      if constexpr (RES) {
        // (a wave tile that lies completely past M or N stores nothing: its loads read the matrix' first rows / columns)
        const int64_t rm = omrem > 0 ? om0 : 0;
        const int rn = onrem > 0 ? on0 : 0;
        const int64_t rb = p.res_blk ? ((((rm >> 8) * (int64_t)(p.N >> 5) + (rn >> 5)) << 14) + (rm & 255) * 64) : (rm * p.ldr + rn) * 2;
        const uint32_t rlo = __builtin_amdgcn_readfirstlane((uint32_t)rb), rhi = __builtin_amdgcn_readfirstlane((uint32_t)(rb >> 32));
        rrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.res + (((uint64_t)rhi << 32) | rlo)), 0, 0x7fffffff, 0x00020000);
        rvo = p.res_blk ? c16 * 64 + 8 * g : (c16 * (int)p.ldr + 4 * g) * 2;  // accumulator layout: row c16 (+16 for the pair's second tile), columns 4 g ..
        rstep = p.res_blk ? 2048 : 64 * (int)p.ldr;
      }
      const int rchi = (RES && p.res_blk) ? 16384 : 64;   // bytes from column tile nt to nt + 2 of the residual
      const int rrow = (RES && p.res_blk) ? 64 : (int)p.ldr * 2;  // bytes per residual row
      // FOLD 1: rstd of this lane's eight rows (16 r + c16).  The weights are centred (mio_ln_fold_weight: every row of gamma o W
      // has its mean over k subtracted), so x . w'^T already equals (x - mean) . (gamma o W)^T and the read-out is acc * rstd +
      // bias: one packed fma where the plain kernel has a packed add.  Each wave turns the slots' (sum, sum of squares) of its
      // group's 128 rows into rstd once (two rows per lane) in its private table and reads back the eight it needs.
      float rsv[8];
      if constexpr (FOLD == 1) {
        const MIO_LDS char* sl = (const MIO_LDS char*)(smem + G8_STAT_OFF + grp * 8192);
        MIO_LDS float* mine = (MIO_LDS float*)(smem + G8_RS_OFF + wave * 512);
        const float ik = 1.f / (float)p.K;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int r = ln + 64 * h;
          float sm = 0.f, sq = 0.f;
          for (int s = 0; s < p.ln_slots; ++s) {
            const f32x2_t v = *(const MIO_LDS f32x2_t*)(sl + s * 1024 + r * 8);
            sm += v[0];
            sq += v[1];
          }
          const float mu = sm * ik;
          mine[r] = __builtin_amdgcn_rsqf(fmaxf(sq * ik - mu * mu, 0.f) + p.ln_eps);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) rsv[r] = mine[16 * r + c16];
      }
      u32x4_t keep[NCH];
      // RES: the residual pieces of the wave tile are requested ahead of the stores (NBUF row pairs of 16 registers: the fragment
      // registers are dead here; the producer form, which also carries its statistics, keeps 3 in flight and requests the fourth
      // behind the first pair's stores).  Requested per row pair, each batch queued behind the previous pair's stores and its wait
      // then also waited for those stores' acknowledgements: four serialised memory round trips per read-out.
      constexpr int NBUF = RES ? (FOLD == 2 ? 3 : 4) : 1;
      u32x2_t resBuf[NBUF][2][ONT];
      auto request_res = [&](auto MTP) {
        constexpr int mtp = decltype(MTP)::value, bi = mtp % NBUF;
        // rows past M: clamped to the tile's first row (their results are never stored)
        const int oa = (full || mtp * 32 + c16 < omrem) ? rvo + mtp * rstep : rvo - c16 * rrow;
        const int ob = (full || mtp * 32 + 16 + c16 < omrem) ? rvo + mtp * rstep + (rstep >> 1) : rvo - c16 * rrow;
#pragma unroll
        for (int nt = 0; nt < ONT; ++nt) {
          const int cofs = (full || nt * 16 + 4 * g < onrem) ? (nt & 1) * 32 + (nt >> 1) * rchi : -8 * g;
          resBuf[bi][0][nt] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rrs, oa + cofs, 0, 0));
          resBuf[bi][1][nt] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rrs, ob + cofs, 0, 0));
        }
      };
      if constexpr (RES) {
        request_res(IC(0));
        request_res(IC(1));
        request_res(IC(2));
        if constexpr (NBUF == 4) request_res(IC(3));
        __builtin_amdgcn_sched_barrier(0);  // (the requests stay in front of the first row pair's arithmetic and stores)
      }
#pragma unroll
      for (int mtp = 0; mtp < 4; ++mtp) {
        const u32x2_t* resA = resBuf[RES ? mtp % NBUF : 0][0];
        const u32x2_t* resB = resBuf[RES ? mtp % NBUF : 0][1];
        const bool mok = full || (mtp * 32 + rS < omrem);
        f32x2_t rsA = {1.f, 1.f}, rsB = {1.f, 1.f};
        if constexpr (FOLD == 1) {
          rsA = (f32x2_t){rsv[2 * mtp], rsv[2 * mtp]};
          rsB = (f32x2_t){rsv[2 * mtp + 1], rsv[2 * mtp + 1]};
        }
        float stA = 0.f, sqA = 0.f, stB = 0.f, sqB = 0.f;  // FOLD 2: this lane's share of the two rows' (sum, sum of squares)
#pragma unroll
        for (int nt = 0; nt < ONT; ++nt) {
          const f32x4_t a = acc[2 * mtp][nt], b = acc[2 * mtp + 1][nt];
          const X4 bv = __builtin_bit_cast(X4, bq[nt]);
          const f32x2_t b01 = {(float)bv[0], (float)bv[1]}, b23 = {(float)bv[2], (float)bv[3]};
          // pre-activation value of rows A / B: acc + bias, or (FOLD 1) acc * rstd + bias
          f32x2_t va01, va23, vb01, vb23;
          if constexpr (FOLD == 1) {
            va01 = (f32x2_t){a[0], a[1]} * rsA + b01;
            va23 = (f32x2_t){a[2], a[3]} * rsA + b23;
            vb01 = (f32x2_t){b[0], b[1]} * rsB + b01;
            vb23 = (f32x2_t){b[2], b[3]} * rsB + b23;
          } else {
            va01 = (f32x2_t){a[0], a[1]} + b01;
            va23 = (f32x2_t){a[2], a[3]} + b23;
            vb01 = (f32x2_t){b[0], b[1]} + b01;
            vb23 = (f32x2_t){b[2], b[3]} + b23;
          }
          f32x2_t a01, a23, c01, c23;
          if constexpr (GLU) {
            const f32x4_t au = acc[2 * mtp][nt + 2], bu = acc[2 * mtp + 1][nt + 2];
            const X4 uv = __builtin_bit_cast(X4, bq[nt + 2]);
            const f32x2_t u01 = {(float)uv[0], (float)uv[1]}, u23 = {(float)uv[2], (float)uv[3]};
            // (va / vb: the gate's pre-activation values; the up half takes the same rstd under FOLD 1)
            f32x2_t ua01, ua23, ub01, ub23;
            if constexpr (FOLD == 1) {
              ua01 = (f32x2_t){au[0], au[1]} * rsA + u01;
              ua23 = (f32x2_t){au[2], au[3]} * rsA + u23;
              ub01 = (f32x2_t){bu[0], bu[1]} * rsB + u01;
              ub23 = (f32x2_t){bu[2], bu[3]} * rsB + u23;
            } else {
              ua01 = (f32x2_t){au[0], au[1]} + u01;
              ua23 = (f32x2_t){au[2], au[3]} + u23;
              ub01 = (f32x2_t){bu[0], bu[1]} + u01;
              ub23 = (f32x2_t){bu[2], bu[3]} + u23;
            }
            a01 = gemm_act2<MIO_ACT_SILU, SCALAR_ACT>(va01) * ua01;
            a23 = gemm_act2<MIO_ACT_SILU, SCALAR_ACT>(va23) * ua23;
            c01 = gemm_act2<MIO_ACT_SILU, SCALAR_ACT>(vb01) * ub01;
            c23 = gemm_act2<MIO_ACT_SILU, SCALAR_ACT>(vb23) * ub23;
          } else {
            a01 = gemm_act2<ACT, SCALAR_ACT>(va01) * cs2;
            a23 = gemm_act2<ACT, SCALAR_ACT>(va23) * cs2;
            c01 = gemm_act2<ACT, SCALAR_ACT>(vb01) * cs2;
            c23 = gemm_act2<ACT, SCALAR_ACT>(vb23) * cs2;
          }
          if constexpr (RES) {
            const X4 ra = __builtin_bit_cast(X4, resA[nt]), rb = __builtin_bit_cast(X4, resB[nt]);
            a01 += (f32x2_t){(float)ra[0], (float)ra[1]};
            a23 += (f32x2_t){(float)ra[2], (float)ra[3]};
            c01 += (f32x2_t){(float)rb[0], (float)rb[1]};
            c23 += (f32x2_t){(float)rb[2], (float)rb[3]};
          }
          const uint32_t pa01 = pack2<T>(a01[0], a01[1]), pa23 = pack2<T>(a23[0], a23[1]);
          const uint32_t pc01 = pack2<T>(c01[0], c01[1]), pc23 = pack2<T>(c23[0], c23[1]);
          if constexpr (FOLD == 2) {  // statistics of what is STORED (the rounded values), as a LayerNorm reading y would see them
            using X2 = typename DT<T>::x2;
            const X2 r0 = __builtin_bit_cast(X2, pa01), r1 = __builtin_bit_cast(X2, pa23);
            const X2 r2 = __builtin_bit_cast(X2, pc01), r3 = __builtin_bit_cast(X2, pc23);
            const float f0 = (float)r0[0], f1 = (float)r0[1], f2 = (float)r1[0], f3 = (float)r1[1];
            const float h0 = (float)r2[0], h1 = (float)r2[1], h2 = (float)r3[0], h3 = (float)r3[1];
            stA += (f0 + f1) + (f2 + f3);
            sqA += (f0 * f0 + f1 * f1) + (f2 * f2 + f3 * f3);
            stB += (h0 + h1) + (h2 + h3);
            sqB += (h0 * h0 + h1 * h1) + (h2 * h2 + h3 * h3);
          }
          const auto s0 = __builtin_amdgcn_permlane16_swap(pa01, pc01, false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(pa23, pc23, false, false);
          const u32x4_t o = {s0[0], s1[0], s0[1], s1[1]};
          keep[mtp * ONT + nt] = o;
          const bool ok = mok && (full || nt * 16 + 8 * (g >> 1) < onrem);
          const int off = ok ? yvo : 0x7fffffff;  // out of range: dropped by the buffer range check
          __builtin_amdgcn_raw_buffer_store_b128(o, yrs, off, mtp * ystep + (nt & 1) * 32 + (nt >> 1) * ychi, 0);
        }
        if constexpr (RES && NBUF == 3) {
          if (mtp == 0) request_res(IC(3));  // into row pair 0's registers, behind its stores
        }
        if constexpr (FOLD == 2) {
          stA += __shfl_xor(stA, 16, 64); sqA += __shfl_xor(sqA, 16, 64); stB += __shfl_xor(stB, 16, 64); sqB += __shfl_xor(sqB, 16, 64);
          stA += __shfl_xor(stA, 32, 64); sqA += __shfl_xor(sqA, 32, 64); stB += __shfl_xor(stB, 32, 64); sqB += __shfl_xor(sqB, 32, 64);
          if (g == 0) {
            MIO_LDS char* pl = (MIO_LDS char*)(smem + G8_STAT_OFF + grp * 8192 + wn * 1024) + (mtp * 32 + c16) * 8;
            *(MIO_LDS f32x2_t*)pl = (f32x2_t){stA, sqA};
            *(MIO_LDS f32x2_t*)(pl + 128) = (f32x2_t){stB, sqB};
          }
        }
      }
      if constexpr (FOLD == 2) {
        // the last of the group's four column-quarter waves to get here adds the partials (fixed order) and stores the slot
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        MIO_LDS unsigned* cnt = (MIO_LDS unsigned*)(smem + G8_CNT_OFF) + grp;
        unsigned old = 0;
        if (ln == 0) old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        asm volatile("" ::: "memory");  // the partials below are read AFTER the counter says the other three waves have written theirs
        if (old == 3u) {
          const MIO_LDS char* pb = (const MIO_LDS char*)(smem + G8_STAT_OFF + grp * 8192);
          const int64_t mpad = (int64_t)p.tiles_m * 256;
          float* dst = p.stats_out + ((int64_t)(on0 >> 8) * mpad + om0) * 2;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int r = ln + 64 * h;
            f32x2_t t = *(const MIO_LDS f32x2_t*)(pb + r * 8);
#pragma unroll
            for (int w = 1; w < 4; ++w) t += *(const MIO_LDS f32x2_t*)(pb + w * 1024 + r * 8);
            *(f32x2_t*)(dst + 2 * r) = t;
            asm volatile("s_nop 7" : : "v"(t) : "memory");  // (store data stays live behind the store, header note)
          }
          if (ln == 0) *cnt = 0u;
        }
      }
      // no register a store reads is written before the stores have fetched their data (header: store data must outlive ...)
      if constexpr (NCH == 16)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                     :
                     : "v"(keep[0]), "v"(keep[1]), "v"(keep[2]), "v"(keep[3]), "v"(keep[4]), "v"(keep[5]), "v"(keep[6]), "v"(keep[7]),
                       "v"(keep[8]), "v"(keep[9]), "v"(keep[10]), "v"(keep[11]), "v"(keep[12]), "v"(keep[13]), "v"(keep[14]), "v"(keep[15])
                     : "memory");
      else
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                     :
                     : "v"(keep[0]), "v"(keep[1]), "v"(keep[2]), "v"(keep[3]), "v"(keep[4]), "v"(keep[5]), "v"(keep[6]), "v"(keep[7])
                     : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((VAR & 128) != 0) {  // per (tile, wave): start, K-tile 0 / 1 done, tail start, loop end, read-out end; 100 MHz
      st[5] = __builtin_amdgcn_s_memtime();
      const unsigned long long sr1 = __builtin_amdgcn_s_memrealtime();
      if (lane_now() == 0 && p.dbg != nullptr) {
        unsigned long long* d = p.dbg + ((size_t)tile * 8 + wave) * 8;
        d[0] = st[0]; d[1] = st[1]; d[2] = st[2]; d[3] = st[3]; d[4] = st[4]; d[5] = st[5]; d[6] = sr0; d[7] = sr1;
      }
    }
    if (!has_next) break;
    tile = next;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#undef IC
}
