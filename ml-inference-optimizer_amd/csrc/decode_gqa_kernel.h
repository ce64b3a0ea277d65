// Paged decode for grouped-query attention on the matrix core (gfx950): the (H / Hkv) * q_len <= 16 query vectors that
// attend through one kv head are the 16 columns of v_mfma_f32_16x16x32 tiles, so a cached key / value element is fetched
// once and used by all of them in ONE instruction.  decode_rows_kernel / decode_paged_kernel spend ~240 vector instructions
// per 16-byte cache chunk at 8 queries per key (1.04 TB/s at H 32 / Hkv 4 / D 128); here the per-chunk work is 16 MFMAs and
// ~60 vector instructions per 32 keys, and the kernel is bound by HBM again.
//
// Same paged layout and semantics as decode_paged_kernel (reference attention_kernels.py:628-808): cache
// [num_blocks, num_layers, block_size, Hkv, D], token t of sequence b in block block_tables[b, t / block_size].
//
//   workgroup = (sequence b, kv head, context split), 4 waves; wave w owns the 32-key chunks w, w + 4, ... of the split and its
//   own online-softmax state; the four states are merged through LDS at the end, the splits by decode_reduce_kernel.
//   K and V of a chunk go global -> LDS by DMA (global_load_lds_dwordx4: no VGPRs, no compiler-visible vector loads, so
//   every s_waitcnt vmcnt in the loop is ours), two stages per wave: ~1.5 chunks (K + V: 16 KiB at D 128) in flight per wave.
//   The products are fa3_fwd5_kernel's swapped forms:
//     S^T tile (16 keys x 16 queries) = K (A: lane (r, g) = K[key r][32 ds + 8 g .. +7]) . Q^T (B: lane (c, g) = Q[c][same d]);
//       accumulator: query c on the lane, keys 4 g + i in registers i;
//     O^T (16 d x 16 queries) += V^T (A: two ds_read_b64_tr_b16 of 4 keys x 16 d from the row-major V image) . P^T (B: the
//       lane's own exp'd registers of key tiles 0 / 1: k = 8 g + j <-> key 4 g + j (j < 4), 16 + 4 g + j - 4 (j >= 4)).
//   LDS images carry an XOR swizzle applied on the DMA's per-lane SOURCE address (K: 16-byte chunk ^ row, V: 32-byte block
//   ^ row) so the fragment reads are bank-conflict free.
//   The block-table slice of the split is staged in LDS once (ds_read lookups: no vector-memory traffic beside the DMA).
#pragma once
#include "mio_common.h"

constexpr int DG_BT_MAX = 2048;  // block-table entries a split may span (8 KiB of LDS)
constexpr int DG_NST = 2;        // (K, V) stages per wave

template <int D>
constexpr int dg_smem_bytes() {
  return 4 * DG_NST * (2 * 32 * D * 2) + DG_BT_MAX * 4;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void decode_gqa_kernel(const DecDev p) {
  using X8 = typename DT<T>::x8;
  using X4 = typename DT<T>::x4;
  constexpr int NDS = D / 32;        // 32-wide d steps of q . k
  constexpr int NDT = D / 16;        // 16-row d tiles of O^T
  constexpr int ROWB = D * 2;        // bytes per cached head row
  constexpr int HALF = 32 * ROWB;    // bytes of a 32-key K (or V) image
  constexpr int STAGE = 2 * HALF;
  constexpr int LPR = ROWB / 16;     // lanes (16-byte chunks) per row: 16 / 8
  constexpr int RPI = 64 / LPR;      // rows per DMA instruction: 4 / 8
  constexpr int NDMA = 32 / RPI;     // DMA instructions per image: 8 / 4
  constexpr int NL = 2 * NDMA;       // vector-memory instructions per chunk
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float s_m[4][16], s_l[4][16];
  MIO_LDS int* bt_s = (MIO_LDS int*)(smem + 4 * DG_NST * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.Hkv, kvh = blockIdx.x % p.Hkv, split = blockIdx.y;
  const int rep = p.H / p.Hkv, QN = rep * p.q_len;
  const int ctx = p.cl[b];
  const int begin = split * p.split_len;
  int end = begin + p.split_len;
  if (end > ctx) end = ctx;
  {
    const int64_t cap = (int64_t)p.max_blocks * p.bs;  // a context longer than the block-table row: the tail is masked
    if (end > cap) end = (int)cap;
  }
  const int nkeys = end > begin ? end - begin : 0;
  const int nch = (nkeys + 31) >> 5;
  const int last = end - 1;
  const int blk0 = begin / p.bs;

  // ---- block-table slice -> LDS
  if (nkeys > 0) {
    const int nb = last / p.bs - blk0 + 1;  // <= DG_BT_MAX (launcher)
    const int32_t* btrow = p.bt + (int64_t)b * p.max_blocks;
    for (int i = tid; i < nb; i += 256) bt_s[i] = btrow[blk0 + i];
  }

  // ---- Q fragments (B operand): lane (c16, g) holds Q[query c16][32 ds + 8 g .. +7]; queries past QN are zero
  X8 qf[NDS];
  {
    const bool ok = c16 < QN;
    const int j = ok ? c16 : 0;
    const T* qp = (const T*)p.q + b * p.qs_b + (int64_t)(kvh * rep + j / p.q_len) * p.qs_h + (int64_t)(j % p.q_len) * p.qs_s;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      u32x4_t raw = {0, 0, 0, 0};
      if (ok) raw = *(const u32x4_t*)(qp + 32 * ds + 8 * g);
      qf[ds] = __builtin_bit_cast(X8, raw);
    }
  }
  __syncthreads();  // bt_s visible
  // the compiler's own vector loads end here: it must not wait for "its" loads inside the loop (its vmcnt(0) would drain the DMA)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ds = 0; ds < NDS; ++ds) asm volatile("" : "+v"(qf[ds]));

  const int64_t tok_bytes = (int64_t)p.Hkv * D * 2;
  const char* kbase = (const char*)p.kc + (int64_t)kvh * ROWB;
  const char* vbase = (const char*)p.vc + (int64_t)kvh * ROWB;
  char* ring = smem + wave * (DG_NST * STAGE);
  const uint32_t ring_lds = (uint32_t)(size_t)((MIO_LDS char*)ring);

  // DMA of chunk j into stage st: instruction i moves rows RPI i .. RPI i + RPI - 1 of the K image and of the V image
  const int drow = lane / LPR, dpos = lane % LPR;
  auto issue = [&](int j, int st) __attribute__((always_inline)) {
    const uint32_t lds = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)(st * STAGE));
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int key = RPI * i + drow;  // row of the chunk image
      int pos = begin + 32 * j + key;
      pos = pos < last ? pos : last;   // addresses stay inside the split; the scores of the padding are masked
      const int blk = bt_s[pos / p.bs - blk0];
      const int64_t row = ((int64_t)blk * p.L + p.layer) * p.bs + pos % p.bs;
      int kc, vc;
      if constexpr (D == 128) {
        kc = dpos ^ (key & 15);
        vc = (((dpos >> 1) ^ (key & 7)) << 1) | (dpos & 1);
      } else {
        kc = dpos ^ ((key >> 1) & 7);
        vc = (((dpos >> 1) ^ ((key >> 1) & 3)) << 1) | (dpos & 1);
      }
      const char* ks = kbase + row * tok_bytes + 16 * kc;
      const char* vs = vbase + row * tok_bytes + 16 * vc;
      const uint32_t lk = lds + 1024 * i, lv = lds + HALF + 1024 * i;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lk), "v"(ks) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lv), "v"(vs) : "memory", "m0");
    }
  };

  // per-lane LDS read offsets
  int k_rd[NDS], v_rd[NDT];
  {
    const int q4 = c16 >> 2, p2 = c16 & 3, vrow = 4 * g + q4;
#pragma unroll
    for (int ds = 0; ds < NDS; ++ds) {
      if constexpr (D == 128) k_rd[ds] = c16 * ROWB + 16 * ((4 * ds + g) ^ c16);
      else k_rd[ds] = c16 * ROWB + 16 * ((4 * ds + g) ^ ((c16 >> 1) & 7));
    }
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      if constexpr (D == 128) v_rd[dt] = HALF + vrow * ROWB + ((dt ^ (vrow & 7)) * 32) + 8 * p2;
      else v_rd[dt] = HALF + vrow * ROWB + ((dt ^ ((vrow >> 1) & 3)) * 32) + 8 * p2;
    }
  }

  const float sl2 = p.scale * LOG2E;
  float m = -INFINITY, l = 0.f;  // m: running maximum of the query's scaled scores (log2 units), the same on its 4 lanes
  f32x4_t o[NDT];
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  int j = wave, st = 0;
  if (j < nch) issue(j, 0);
  if (j + 4 < nch) issue(j + 4, 1);
  for (; j < nch; j += 4, st ^= 1) {
    if (j + 4 < nch) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const char* sb = ring + st * STAGE;
    // ---- S^T = K . Q^T for the two 16-key tiles
    f32x4_t s2[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < NDS; ++ds) {
        const X8 kf = __builtin_bit_cast(X8, *(const MIO_LDS u32x4_t*)(sb + kt * 16 * ROWB + k_rd[ds]));
        acc = DT<T>::mfma16(kf, qf[ds], acc);
      }
      s2[kt] = acc;
    }
    // ---- scale, mask, online softmax (a query's keys sit on 4 lanes: c16, c16 + 16, + 32, + 48)
    const int kpos = begin + 32 * j + 4 * g;
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = (kpos + 16 * kt + i < end) ? s2[kt][i] * sl2 : -INFINITY;
        s2[kt][i] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = fast_exp2(m - m_ref);
    m = m_new;
    uint32_t pk[4];
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float e0 = fast_exp2(s2[kt][2 * h2] - m_ref), e1 = fast_exp2(s2[kt][2 * h2 + 1] - m_ref);
        const typename DT<T>::x2 r = __builtin_convertvector((f32x2_t){e0, e1}, typename DT<T>::x2);
        psum += (float)r[0] + (float)r[1];  // the row sum counts what the product multiplies: the rounded weights
        pk[2 * kt + h2] = __builtin_bit_cast(uint32_t, r);
      }
    l = l * alpha + psum;
    const X8 pf = __builtin_bit_cast(X8, (u32x4_t){pk[0], pk[1], pk[2], pk[3]});
    // ---- O^T = O^T * alpha + V^T . P^T
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) {
      const X4 lo = DT<T>::ds_read_tr(sb + v_rd[dt]);
      const X4 hi = DT<T>::ds_read_tr(sb + 16 * ROWB + v_rd[dt]);
      const X8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      f32x4_t acc = o[dt];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] *= alpha;
      o[dt] = DT<T>::mfma16(vf, pf, acc);
    }
    // the stage is free once its fragment reads have returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (j + 8 < nch) issue(j + 8, st);
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);

  // ---- merge the four waves' states: O^T as [wave][query][d] fp32 in the (now idle) ring space
  __syncthreads();
  MIO_LDS float* ob = (MIO_LDS float*)smem;
#pragma unroll
  for (int dt = 0; dt < NDT; ++dt) *(MIO_LDS f32x4_t*)(ob + (wave * 16 + c16) * D + 16 * dt + 4 * g) = o[dt];
  if (g == 0) {
    s_m[wave][c16] = m;
    s_l[wave][c16] = l;
  }
  __syncthreads();
  constexpr int CPQ = D / 8;  // 8-column pieces per query
  if (tid < 16 * CPQ) {
    const int qj = tid / CPQ, c8 = tid % CPQ;
    if (qj < QN) {
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < 4; ++w) M = fmaxf(M, s_m[w][qj]);
      float Ls = 0.f, acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      if (M != -INFINITY) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const float wgt = fast_exp2(s_m[w][qj] - M);
          Ls += s_l[w][qj] * wgt;
          const MIO_LDS float* src = ob + (w * 16 + qj) * D + 8 * c8;
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[i] += src[i] * wgt;
        }
      }
      const float inv = (Ls > 0.f) ? 1.f / Ls : 0.f;  // empty context -> 0 (attention_kernels.py:802)
      const int h = kvh * rep + qj / p.q_len, qi = qj % p.q_len;
      const int64_t row = ((int64_t)b * p.H + h) * p.q_len + qi;
      if (p.nsplit == 1) {
        uint32_t w4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w4[i] = pack2<T>(acc[2 * i] * inv, acc[2 * i + 1] * inv);
        *(u32x4_t*)((T*)p.o + b * p.os_b + h * p.os_h + (int64_t)qi * p.os_s + 8 * c8) = (u32x4_t){w4[0], w4[1], w4[2], w4[3]};
      } else {
        float* wo = p.ws_o + (row * p.nsplit + split) * p.D + 8 * c8;
#pragma unroll
        for (int i = 0; i < 8; ++i) wo[i] = acc[i] * inv;
        if (c8 == 0) p.ws_lse[row * p.nsplit + split] = (Ls > 0.f) ? (M + fast_log2(Ls)) * LN2 : -INFINITY;
      }
    }
  }
}
