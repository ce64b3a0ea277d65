// Paged-KV decode attention + cache scatter for CDNA4 (gfx950).  HBM-bandwidth-bound: K/V rows go
// straight from global memory to VGPRs (16 B per lane), no LDS staging, split over the context so
// a small batch still fills 256 CUs; partial (o, lse) states are merged by a second tiny kernel.
//
// Replaces _paged_attention_fwd_kernel (reference kernels/triton/attention_kernels.py:628-808, wrapper
// :1206-1311) and _reshape_and_cache_kernel (:811-905, wrapper :1314-1407).
// Cache layout [num_blocks, num_layers, block_size, Hkv, D]; token t of sequence b lives in physical
// block block_tables[b, t / block_size] at slot t % block_size (:728-751).
#include <cstdlib>
#include <mutex>

#include "mio_common.h"

struct DecDev {
  const void* q;
  void* o;
  const void* kc;
  const void* vc;
  const int32_t* bt;
  const int32_t* cl;
  float* ws_o;    // [rows, nsplit, D]
  float* ws_lse;  // [rows, nsplit]
  int64_t qs_b, qs_h, qs_s, os_b, os_h, os_s;
  int B, H, Hkv, q_len, D, L, layer, bs, max_blocks, nsplit, split_len;
  float scale;
};

#include "decode_gqa_kernel.h"

// decode_gqa_kernel (matrix-core form): one workgroup per (sequence, kv head, split); the split count aims at one
// (D 128: 136 KiB of LDS) or two (D 64) workgroups per CU, 128-key granularity (4 waves x 32-key chunks)
static inline int dec_nsplit_gqa(int64_t units, int max_ctx, int bs, int D = 128) {
  // workgroups aimed for = what is resident at once (D 128: 136 KiB of LDS, one per CU; D 64: two per CU); a second round of
  // workgroups costs its tail (tools/dbg/dec_gqa_ab.py, B 64 H 32 Hkv 4 D 128: 256 -> 5.68 TB/s, 512 -> 5.37, 2048 -> 4.50)
  int target = D > 64 ? 256 : 512;
#ifdef MIO_DIAG
  if (mio_dbg_get(2) > 0) target = mio_dbg_get(2);
#endif
  int want = (int)((target + units - 1) / units);
  int cap = (max_ctx + 255) / 256;  // >= 256 keys per split
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  // a split's block-table slice must fit its LDS image
  while (((int64_t)(max_ctx + want - 1) / want + 127) / bs + 2 > DG_BT_MAX) ++want;
  return want;
}
// Picked when 2 .. 16 query vectors share a cached key ((H / Hkv) * q_len): the vector-ALU kernels are HBM-bound with one
// (5.7-5.9 TB/s at MHA) and fall off from there -- tools/dbg/dec_gqa_small_qn.py, shipped-before vs this kernel: 2 vectors
// 5.52 -> 6.16 TB/s (D 128, B 64), 2.31 -> 3.25 (D 64, B 8), 5.46 -> 5.25 (D 64, two query positions: the one loss); 3 vectors
// 2.36 -> 5.75; 4 vectors 3.32 -> 5.48 and 1.67 -> 4.89 at B 1 x ctx 131072; 8 vectors 1.04 -> 5.4-5.7.  With one vector it
// is 8 % slower at D 64 (5.20 vs 5.68: left to the row kernels) and 2 % faster at D 128 (taken).
static inline bool dec_gqa_ok(int B, int H, int Hkv, int q_len, int D, int max_ctx, int bs, const int64_t* os, const void* o) {
  const int qn = (H / Hkv) * q_len;
  if (qn > 16 || (qn < 2 && D != 128)) return false;  // one query vector per key: only at D 128 (5.90 -> 6.05, 5.49 -> 5.62 TB/s)
  if (D != 64 && D != 128) return false;
  if (max_ctx < 1) return false;
  if (os[0] % 8 != 0 || os[1] % 8 != 0 || os[2] % 8 != 0 || !mio_aligned16(o)) return false;  // 16-byte output stores
  return true;
}

static inline int dec_nsplit(int B, int H, int q_len, int max_ctx) {
  const int64_t rows = (int64_t)B * H * q_len;
#ifdef MIO_DIAG
  static const int wgs = [] {  // MIO_DEC_WGS: workgroups the split aims for (tuning aid)
    const char* e = std::getenv("MIO_DEC_WGS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 512;
  }();
#else
  constexpr int wgs = 512;  // workgroups the split aims for: 2 per CU
#endif
  int want = (int)((wgs + rows - 1) / rows);
  int cap = (max_ctx + 255) / 256;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return want;
}

// whole-token-row kernel (decode_rows_kernel): one workgroup per (sequence, split) -> the split count aims at the same
// number of workgroups with B sequences instead of B * H * q_len rows
static inline int dec_nsplit_rows(int B, int max_ctx) {
  int target = 512;  // workgroups aimed for: 2 per CU (sweep 256 .. 4096 at B 8 / 32 / 64 / 256: tools/dbg/dec_rows_sweep.py)
#ifdef MIO_DIAG
  if (mio_dbg_get(2) > 0) target = mio_dbg_get(2);  // tuning sweep (tools/dbg/dec_rows_sweep.py)
#endif
  int want = (target + B - 1) / B;
  int cap = (max_ctx + 63) / 64;
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return want;
}
// Picked for B >= 16: at B 8 (128 MiB of cache, Infinity-Cache resident between launches) the per-head kernel's 512
// small workgroups run 28 us against 40-46 us here; from B 32 on (streams from HBM) the whole-row reads win:
// B 64 H 16 D 64 187 -> 183 us (5.87 TB/s), B 256 6.06 TB/s (tools/dbg/dec_rows_sweep.py).
static inline bool dec_rows_ok(int B, int H, int Hkv, int q_len, int D, int max_ctx) {
  // a cache that fits the 256 MiB Infinity Cache between steps (B 8 at ctx 4096: 128 MiB) is read faster by the per-head
  // kernel's many small workgroups (28 vs 40-46 us); one that streams from HBM goes through whole token rows from B 8 on
  // (round 3, B 8 x ctx 32768: 5.49 vs 5.42 TB/s; B 4 x ctx 65536: 5.02 vs 5.41 -- too few sequences per split column)
  const double kv_bytes = 4.0 * B * (double)max_ctx * Hkv * D;
  if (B < 16 && (B < 8 || kv_bytes <= 200.0 * 1048576.0)) return false;
  if (D != 64 && D != 128) return false;
  const int cpt = Hkv * (D / 8), qn = (H / Hkv) * q_len;
  return cpt >= 16 && cpt <= 256 && (cpt & (cpt - 1)) == 0 && qn == 1;  // 2 .. 16 query vectors per key: decode_gqa_kernel
}

// CPRP = chunks-per-row padded to a power of two (8 for D <= 64, 16 for D <= 128)
template <typename T, int CPRP, int U>
__global__ __launch_bounds__(256) void decode_paged_kernel(const DecDev p) {
  constexpr int TPI = 64 / CPRP;  // tokens per wave-iteration
  constexpr int NSTATE = 4 * TPI;
  __shared__ float s_o[NSTATE][CPRP * 8 + 1];
  __shared__ float s_m[NSTATE], s_l[NSTATE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = lane / CPRP, c = lane % CPRP;
  const int row = blockIdx.x;  // (b, h, qi)
  const int split = blockIdx.y;
  const int qi = row % p.q_len;
  const int h = (row / p.q_len) % p.H;
  const int b = row / (p.q_len * p.H);
  const int kvh = h / (p.H / p.Hkv);
  const int ctx = p.cl[b];
  const int begin = split * p.split_len;
  int end = begin + p.split_len;
  if (end > ctx) end = ctx;
  const bool c_ok = (8 * c < p.D);

  float qf[8];
  {
    u32x4_t raw = {0, 0, 0, 0};
    if (c_ok) raw = *(const u32x4_t*)((const T*)p.q + b * p.qs_b + h * p.qs_h + (int64_t)qi * p.qs_s + 8 * c);
    const typename DT<T>::x8 v = __builtin_bit_cast(typename DT<T>::x8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[i] = (float)v[i] * p.scale;
  }

  float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = 0.f;

  const int64_t tok_stride = (int64_t)p.Hkv * p.D;
  const int64_t blk_stride = (int64_t)p.L * p.bs * tok_stride;

  // U wave-iterations (U * TPI tokens per wave) per batch, two batches in flight: the K/V rows of batch i+1 and the
  // block-table entries of batch i+2 are requested before batch i is reduced, and one max / rescale serves the U
  // tokens of a batch.  Measured (tools/dbg/dec_sweep.sh): U = 1 .. 4 are within 2 % of each other, U = 8 is 3-5 %
  // slower -- the kernel is bound by the 128-byte-pieces-at-token-stride access pattern (5.2-5.4 TB/s at B 64), not
  // by loads in flight.
  constexpr int STEP = 4 * TPI;  // tokens the workgroup's four waves cover per iteration
  // Loads are unconditional (addresses clamped to the split's last token / the row's first chunk, values masked in
  // `reduce`): predicated loads become branches, and hipcc drains vmcnt at every join, which serialises the batches.
  const int last = end - 1;  // >= begin here: empty splits skip the loop
  const int coff = c_ok ? 8 * c : 0;
  const int64_t lay_off = (int64_t)p.layer * p.bs * tok_stride + (int64_t)kvh * p.D + coff;
  const int32_t* btrow = p.bt + (int64_t)b * p.max_blocks;
  auto load_pb = [&](int pos0, int (&pb)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int pos = min(pos0 + j * STEP + t, last);
      pb[j] = btrow[min(pos / p.bs, p.max_blocks - 1)];
    }
  };
  auto load_kv = [&](int pos0, const int (&pb)[U], u32x4_t (&kr)[U], u32x4_t (&vr)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int pos = min(pos0 + j * STEP + t, last);
      const int64_t off = (int64_t)pb[j] * blk_stride + lay_off + (int64_t)(pos % p.bs) * tok_stride;
      kr[j] = *(const u32x4_t*)((const T*)p.kc + off);
      vr[j] = *(const u32x4_t*)((const T*)p.vc + off);
    }
  };
  auto reduce = [&](int pos0, const int (&pb)[U], const u32x4_t (&kr)[U], const u32x4_t (&vr)[U]) {
    float sc[U];
    float m_new = m;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const typename DT<T>::x8 kv = __builtin_bit_cast(typename DT<T>::x8, kr[j]);
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += qf[i] * (float)kv[i];  // qf = 0 in the padding chunks (c_ok false)
#pragma unroll
      for (int x = 1; x < CPRP; x <<= 1) s += __shfl_xor(s, x, 64);
      const int pos = pos0 + j * STEP + t;
      sc[j] = (pos < end && pos / p.bs < p.max_blocks) ? s : -INFINITY;  // uniform within the token's lane group
      m_new = fmaxf(m_new, sc[j]);
    }
    const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;  // nothing seen yet: every weight below is exp(-inf) = 0
    const float alpha = __expf(m - m_ref);
    l *= alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] *= alpha;
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const typename DT<T>::x8 vv = __builtin_bit_cast(typename DT<T>::x8, vr[j]);
      const float pe = __expf(sc[j] - m_ref);
      l += pe;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += pe * (float)vv[i];
    }
    m = m_new;
  };
  {
    constexpr int BATCH = U * STEP;
    int pbA[U], pbB[U], pbC[U];  // block ids of the batch being reduced, the next one, and the one after
    u32x4_t kA[U], vA[U], kB[U], vB[U];
    auto shift = [&]() {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        pbA[j] = pbB[j];
        pbB[j] = pbC[j];
      }
    };
    int pos0 = begin + wave * TPI;
    if (begin >= end) pos0 = end;  // empty split: no loads at all
    else {
    load_pb(pos0, pbA);
    load_pb(pos0 + BATCH, pbB);
    load_kv(pos0, pbA, kA, vA);
    }
    while (pos0 < end) {
      load_pb(pos0 + 2 * BATCH, pbC);
      load_kv(pos0 + BATCH, pbB, kB, vB);
      reduce(pos0, pbA, kA, vA);
      pos0 += BATCH;
      if (pos0 >= end) break;
      shift();
      load_pb(pos0 + 2 * BATCH, pbC);
      load_kv(pos0 + BATCH, pbB, kA, vA);
      reduce(pos0, pbA, kB, vB);
      pos0 += BATCH;
      shift();
    }
  }

  // ---- merge the NSTATE per-(wave, token-slot) states
  const int g = wave * TPI + t;
#pragma unroll
  for (int i = 0; i < 8; ++i) s_o[g][8 * c + i] = o[i];
  if (c == 0) {
    s_m[g] = m;
    s_l[g] = l;
  }
  __syncthreads();
  if (tid < p.D) {
    float M = -INFINITY;
    for (int j = 0; j < NSTATE; ++j) M = fmaxf(M, s_m[j]);
    float Lsum = 0.f, acc = 0.f;
    if (M != -INFINITY) {
      for (int j = 0; j < NSTATE; ++j) {
        const float w = __expf(s_m[j] - M);
        Lsum += s_l[j] * w;
        acc += s_o[j][tid] * w;
      }
    }
    const float val = (Lsum > 0.f) ? acc / Lsum : 0.f;  // empty context -> 0 (attention_kernels.py:802)
    if (p.nsplit == 1) {
      ((T*)p.o)[b * p.os_b + h * p.os_h + (int64_t)qi * p.os_s + tid] = (T)val;
    } else {
      p.ws_o[((int64_t)row * p.nsplit + split) * p.D + tid] = val;
      if (tid == 0) p.ws_lse[(int64_t)row * p.nsplit + split] = (Lsum > 0.f) ? M + __logf(Lsum) : -INFINITY;
    }
  }
}

// ---- whole-token-row variant -------------------------------------------------------------------------------------------
// decode_paged_kernel reads 128- / 256-byte pieces (one head of one token) at the token stride Hkv * D * 2 bytes: every
// piece is a separate HBM burst in a random physical block, and the other heads' workgroups fetch the neighbouring pieces
// at other times (5.1-5.4 TB/s at B 64, H 16, D 64).  Here ONE workgroup owns a (sequence, context split) for ALL heads:
// a wave-load is 64 lanes x 16 B = 1 KiB of ONE token row (contiguous: the cache stores [token][Hkv][D]), a lane keeps the
// running softmax state of its own (kv head, 16-byte chunk) for the QN = (H / Hkv) * q_len query vectors that attend
// through that kv head, and consecutive tokens of a block are consecutive 2-KiB rows -- 32 KiB contiguous per block.
//   CPR  = chunks per head row (D / 8: 8 or 16)          CPT = Hkv * CPR chunks per token row (16 .. 256, power of two)
//   CPT >= 64: the row is NPART = CPT / 64 wave-loads; wave w takes part w % NPART of tokens (w / NPART) + k * (4 / NPART)
//   CPT <  64: one wave-load holds TPL = 64 / CPT tokens; wave w takes token slots 4 k + w
// Same clamped-address / masked-score loop as decode_paged_kernel, batches of U token slots, double-buffered.
template <typename T, int CPR, int QN>
__global__ __launch_bounds__(256) void decode_rows_kernel(const DecDev p) {
  constexpr int U = 2;  // token slots per batch (U = 4 measured the same within 2 %)
  __shared__ float s_st[4][64][QN][10];  // per (wave, lane, query): o[8], m, l

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x, split = blockIdx.y;
  const int CPT = p.Hkv * CPR;
  const int npart = CPT >= 64 ? CPT / 64 : 1;
  const int tpl = CPT >= 64 ? 1 : 64 / CPT;             // tokens per wave-load
  const int part = wave % npart, tslot = wave / npart;  // this wave's slice of the row / token slot
  const int wpp = 4 / npart;                            // waves per part
  const int tl = CPT >= 64 ? 0 : lane / CPT;            // token inside the wave-load
  const int cidx = CPT >= 64 ? part * 64 + lane : lane % CPT;  // 16-byte chunk of the token row
  const int kvh = cidx / CPR, c = cidx % CPR;
  const int rep = p.H / p.Hkv;
  const int ctx = p.cl[b];
  const int begin = split * p.split_len;
  int end = begin + p.split_len;
  if (end > ctx) end = ctx;

  float qf[QN][8];
#pragma unroll
  for (int j = 0; j < QN; ++j) {
    const int h = kvh * rep + j / p.q_len, qi = j % p.q_len;
    const u32x4_t raw = *(const u32x4_t*)((const T*)p.q + b * p.qs_b + h * p.qs_h + (int64_t)qi * p.qs_s + 8 * c);
    const typename DT<T>::x8 v = __builtin_bit_cast(typename DT<T>::x8, raw);
#pragma unroll
    for (int i = 0; i < 8; ++i) qf[j][i] = (float)v[i] * p.scale;
  }
  float m[QN], l[QN], o[QN][8];
#pragma unroll
  for (int j = 0; j < QN; ++j) {
    m[j] = -INFINITY;
    l[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[j][i] = 0.f;
  }

  const int64_t tok_stride = (int64_t)p.Hkv * p.D;
  const int64_t blk_stride = (int64_t)p.L * p.bs * tok_stride;
  const int64_t lay_off = (int64_t)p.layer * p.bs * tok_stride + (int64_t)cidx * 8;
  const int32_t* btrow = p.bt + (int64_t)b * p.max_blocks;
  const int step = wpp * tpl;  // tokens the workgroup covers per slot
  const int last = end - 1;
  auto tok = [&](int pos0, int j) { return pos0 + j * step + tl; };
  auto load_pb = [&](int pos0, int (&pb)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int pos = min(tok(pos0, j), last);
      pb[j] = btrow[min(pos / p.bs, p.max_blocks - 1)];
    }
  };
  auto load_kv = [&](int pos0, const int (&pb)[U], u32x4_t (&kr)[U], u32x4_t (&vr)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int pos = min(tok(pos0, j), last);
      const int64_t off = (int64_t)pb[j] * blk_stride + lay_off + (int64_t)(pos % p.bs) * tok_stride;
      kr[j] = *(const u32x4_t*)((const T*)p.kc + off);
      vr[j] = *(const u32x4_t*)((const T*)p.vc + off);
    }
  };
  auto reduce = [&](int pos0, const u32x4_t (&kr)[U], const u32x4_t (&vr)[U]) {
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      float sc[U];
      float m_new = m[q];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const typename DT<T>::x8 kv = __builtin_bit_cast(typename DT<T>::x8, kr[j]);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += qf[q][i] * (float)kv[i];
#pragma unroll
        for (int x = 1; x < CPR; x <<= 1) s += __shfl_xor(s, x, 64);
        const int pos = tok(pos0, j);
        sc[j] = (pos < end && pos / p.bs < p.max_blocks) ? s : -INFINITY;
        m_new = fmaxf(m_new, sc[j]);
      }
      const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;
      const float alpha = __expf(m[q] - m_ref);
      l[q] *= alpha;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[q][i] *= alpha;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const typename DT<T>::x8 vv = __builtin_bit_cast(typename DT<T>::x8, vr[j]);
        const float pe = __expf(sc[j] - m_ref);
        l[q] += pe;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[q][i] += pe * (float)vv[i];
      }
      m[q] = m_new;
    }
  };
  {
    const int BATCH = U * step;
    int pbA[U], pbB[U], pbC[U];
    u32x4_t kA[U], vA[U], kB[U], vB[U];
    auto shift = [&]() {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        pbA[j] = pbB[j];
        pbB[j] = pbC[j];
      }
    };
    int pos0 = begin + tslot * tpl;
    if (begin >= end) pos0 = end;  // empty split: no loads at all
    else {
      load_pb(pos0, pbA);
      load_pb(pos0 + BATCH, pbB);
      load_kv(pos0, pbA, kA, vA);
    }
    while (pos0 < end) {
      load_pb(pos0 + 2 * BATCH, pbC);
      load_kv(pos0 + BATCH, pbB, kB, vB);
      reduce(pos0, kA, vA);
      pos0 += BATCH;
      if (pos0 >= end) break;
      shift();
      load_pb(pos0 + 2 * BATCH, pbC);
      load_kv(pos0 + BATCH, pbB, kA, vA);
      reduce(pos0, kB, vB);
      pos0 += BATCH;
      shift();
    }
  }
  // ---- merge the states of one (chunk of the row, query) held by several waves / token slots
#pragma unroll
  for (int q = 0; q < QN; ++q) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s_st[wave][lane][q][i] = o[q][i];
    s_st[wave][lane][q][8] = m[q];
    s_st[wave][lane][q][9] = l[q];
  }
  __syncthreads();
  if (tslot == 0 && tl == 0) {  // one lane per chunk of the row: its own state first, then the others'
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      float M = -INFINITY;
      for (int w = part; w < 4; w += npart)
        for (int t2 = 0; t2 < tpl; ++t2) M = fmaxf(M, s_st[w][(CPT >= 64 ? lane : t2 * CPT + cidx)][q][8]);
      float Ls = 0.f, acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = 0.f;
      if (M != -INFINITY) {
        for (int w = part; w < 4; w += npart)
          for (int t2 = 0; t2 < tpl; ++t2) {
            const float* st = s_st[w][(CPT >= 64 ? lane : t2 * CPT + cidx)][q];
            const float wgt = __expf(st[8] - M);
            Ls += st[9] * wgt;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += st[i] * wgt;
          }
      }
      const float inv = (Ls > 0.f) ? 1.f / Ls : 0.f;  // empty context -> 0 (attention_kernels.py:802)
      const int h = kvh * rep + q / p.q_len, qi = q % p.q_len;
      const int64_t row = ((int64_t)b * p.H + h) * p.q_len + qi;
      if (p.nsplit == 1) {
        T* op = (T*)p.o + b * p.os_b + h * p.os_h + (int64_t)qi * p.os_s + 8 * c;
#pragma unroll
        for (int i = 0; i < 8; ++i) op[i] = (T)(acc[i] * inv);
      } else {
        float* wo = p.ws_o + (row * p.nsplit + split) * p.D + 8 * c;
#pragma unroll
        for (int i = 0; i < 8; ++i) wo[i] = acc[i] * inv;
        if (c == 0) p.ws_lse[row * p.nsplit + split] = (Ls > 0.f) ? M + __logf(Ls) : -INFINITY;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(128) void decode_reduce_kernel(const DecDev p) {
  const int row = blockIdx.x, d = threadIdx.x;
  if (d >= p.D) return;
  const int qi = row % p.q_len;
  const int h = (row / p.q_len) % p.H;
  const int b = row / (p.q_len * p.H);
  float M = -INFINITY;
  for (int s = 0; s < p.nsplit; ++s) M = fmaxf(M, p.ws_lse[(int64_t)row * p.nsplit + s]);
  float W = 0.f, acc = 0.f;
  if (M != -INFINITY) {
    for (int s = 0; s < p.nsplit; ++s) {
      const float w = __expf(p.ws_lse[(int64_t)row * p.nsplit + s] - M);
      W += w;
      acc += w * p.ws_o[((int64_t)row * p.nsplit + s) * p.D + d];
    }
  }
  ((T*)p.o)[b * p.os_b + h * p.os_h + (int64_t)qi * p.os_s + d] = (T)((W > 0.f) ? acc / W : 0.f);
}

// wave-iterations per double-buffered batch: 2 (a sweep of 1 / 2 / 4 / 8 moved the kernel by <= 3 %, 8 slower; the
// diagnostic build keeps the sweep behind MIO_DEC_U, read once)
template <typename T, int CPRP>
static void dec_launch_u(const DecDev& p, dim3 grid, hipStream_t st) {
#ifdef MIO_DIAG
  static const int u = [] { const char* e = std::getenv("MIO_DEC_U"); return e ? std::atoi(e) : 2; }();
  switch (u) {
    case 1: hipLaunchKernelGGL((decode_paged_kernel<T, CPRP, 1>), grid, dim3(256), 0, st, p); return;
    case 4: hipLaunchKernelGGL((decode_paged_kernel<T, CPRP, 4>), grid, dim3(256), 0, st, p); return;
    case 8: hipLaunchKernelGGL((decode_paged_kernel<T, CPRP, 8>), grid, dim3(256), 0, st, p); return;
    default: break;
  }
#endif
  hipLaunchKernelGGL((decode_paged_kernel<T, CPRP, 2>), grid, dim3(256), 0, st, p);
}

template <typename T>
static void dec_launch(const DecDev& p, dim3 grid, hipStream_t st) {
  if (p.D <= 64) dec_launch_u<T, 8>(p, grid, st);
  else dec_launch_u<T, 16>(p, grid, st);
  if (p.nsplit > 1) hipLaunchKernelGGL(decode_reduce_kernel<T>, dim3(grid.x), dim3(128), 0, st, p);
}

template <typename T>
static void dec_launch_rows(const DecDev& p, int qn, unsigned rows, hipStream_t st) {
  const dim3 grid((unsigned)p.B, (unsigned)p.nsplit);
#define MIO_DEC_ROWS(CPR_, QN_) hipLaunchKernelGGL((decode_rows_kernel<T, CPR_, QN_>), grid, dim3(256), 0, st, p)
  (void)qn;  // always 1 (dec_rows_ok); the kernel template keeps QN for the diagnostic sweeps
  if (p.D == 64) MIO_DEC_ROWS(8, 1);
  else MIO_DEC_ROWS(16, 1);
#undef MIO_DEC_ROWS
  if (p.nsplit > 1) hipLaunchKernelGGL(decode_reduce_kernel<T>, dim3(rows), dim3(128), 0, st, p);
}

template <typename T>
static int dec_launch_gqa(const DecDev& p, unsigned rows, hipStream_t st) {
  const dim3 grid((unsigned)(p.B * p.Hkv), (unsigned)p.nsplit);
  static std::once_flag once;
  static hipError_t ea = hipSuccess;
  std::call_once(once, [&] {
    ea = hipFuncSetAttribute((const void*)decode_gqa_kernel<T, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, dg_smem_bytes<128>());
    if (ea == hipSuccess)
      ea = hipFuncSetAttribute((const void*)decode_gqa_kernel<T, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, dg_smem_bytes<64>());
  });
  if (ea != hipSuccess) return mio_fail(std::string("decode_gqa: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  if (p.D == 128) hipLaunchKernelGGL((decode_gqa_kernel<T, 128>), grid, dim3(256), dg_smem_bytes<128>(), st, p);
  else hipLaunchKernelGGL((decode_gqa_kernel<T, 64>), grid, dim3(256), dg_smem_bytes<64>(), st, p);
  if (p.nsplit > 1) hipLaunchKernelGGL(decode_reduce_kernel<T>, dim3(rows), dim3(128), 0, st, p);
  return 0;
}

extern "C" size_t mio_fa3_decode_workspace_bytes(int32_t B, int32_t H, int32_t q_len, int32_t D, int32_t max_ctx) {
  const int ns_a = dec_nsplit(B, H, q_len, max_ctx), ns_b = dec_nsplit_rows(B, max_ctx);
  const int ns_c = dec_nsplit_gqa(B, max_ctx, 1, 64);  // fewest units (Hkv 1), smallest block size, larger target: the largest split count
  int ns = ns_a > ns_b ? ns_a : ns_b;  // covers whichever kernel the launch picks (it does not know Hkv here)
  if (ns_c > ns) ns = ns_c;
  return (size_t)B * H * q_len * ns * (size_t)(D + 1) * sizeof(float) + 256;
}

extern "C" int mio_fa3_decode_paged(const void* q, void* o, const void* k_cache, const void* v_cache,
                                    const int32_t* block_tables, const int32_t* context_lengths,
                                    const int64_t q_stride[3], const int64_t o_stride[3], int32_t B, int32_t H,
                                    int32_t Hkv, int32_t q_len, int32_t D, int32_t num_layers, int32_t layer_idx,
                                    int32_t block_size, int32_t max_blocks_per_seq, int32_t max_ctx, float scale,
                                    int32_t dtype, void* workspace, void* stream) {
  MIO_CHECK(q && o && k_cache && v_cache && block_tables && context_lengths, "mio_fa3_decode_paged: null pointer");
  MIO_CHECK(B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0 && q_len > 0, "mio_fa3_decode_paged: bad sizes");
  MIO_CHECK(D >= 8 && D <= 128 && D % 8 == 0, "mio_fa3_decode_paged: head_dim must be a multiple of 8 in [8,128]");
  MIO_CHECK(layer_idx >= 0 && layer_idx < num_layers, "mio_fa3_decode_paged: layer_idx out of range");
  MIO_CHECK(block_size > 0 && max_blocks_per_seq > 0 && max_ctx >= 0, "mio_fa3_decode_paged: bad cache geometry");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_fa3_decode_paged: dtype must be bf16 or fp16");
  MIO_CHECK(q_stride[0] % 8 == 0 && q_stride[1] % 8 == 0 && q_stride[2] % 8 == 0 && mio_aligned16(q) &&
                mio_aligned16(k_cache) && mio_aligned16(v_cache),
            "mio_fa3_decode_paged: q/cache rows must be 16-byte aligned");
  DecDev p;
  p.q = q; p.o = o; p.kc = k_cache; p.vc = v_cache; p.bt = block_tables; p.cl = context_lengths;
  p.qs_b = q_stride[0]; p.qs_h = q_stride[1]; p.qs_s = q_stride[2];
  p.os_b = o_stride[0]; p.os_h = o_stride[1]; p.os_s = o_stride[2];
  p.B = B; p.H = H; p.Hkv = Hkv; p.q_len = q_len; p.D = D; p.L = num_layers; p.layer = layer_idx;
  p.bs = block_size; p.max_blocks = max_blocks_per_seq; p.scale = scale;
  bool rows_kernel = dec_rows_ok(B, H, Hkv, q_len, D, max_ctx);
  bool gqa_kernel = dec_gqa_ok(B, H, Hkv, q_len, D, max_ctx, block_size, o_stride, o);
#ifdef MIO_DIAG
  if (mio_dbg_get(6) == 1) rows_kernel = gqa_kernel = false;  // A/B: the per-head kernel (tools/dbg/dec_rows_ab.py)
  if (mio_dbg_get(6) == 2) gqa_kernel = false;                // A/B: rows kernel where it applies
  if (mio_dbg_get(6) == 3)                                    // A/B: the matrix-core kernel for 1 .. 4 query vectors too
    gqa_kernel = (H / Hkv) * q_len <= 16 && (D == 64 || D == 128) && o_stride[0] % 8 == 0 && o_stride[1] % 8 == 0 &&
                 o_stride[2] % 8 == 0 && mio_aligned16(o);
#endif
  if (gqa_kernel) rows_kernel = false;
  p.nsplit = gqa_kernel ? dec_nsplit_gqa((int64_t)B * Hkv, max_ctx, block_size, D)
                        : rows_kernel ? dec_nsplit_rows(B, max_ctx) : dec_nsplit(B, H, q_len, max_ctx);
  int sl = (max_ctx + p.nsplit - 1) / p.nsplit;
  const int gran = gqa_kernel ? 128 : 32;
  sl = (sl + gran - 1) / gran * gran;
  if (sl < gran) sl = gran;
  p.split_len = sl;
  MIO_CHECK(p.nsplit == 1 || workspace != nullptr, "mio_fa3_decode_paged: workspace required");
  const int64_t rows = (int64_t)B * H * q_len;
  p.ws_o = (float*)workspace;
  p.ws_lse = p.ws_o ? p.ws_o + rows * p.nsplit * D : nullptr;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)rows, (unsigned)p.nsplit), block(256);
  if (gqa_kernel) {
    const int rc = (dtype == MIO_BF16) ? dec_launch_gqa<__bf16>(p, (unsigned)rows, st) : dec_launch_gqa<_Float16>(p, (unsigned)rows, st);
    if (rc != 0) return rc;
  } else if (rows_kernel) {
    if (dtype == MIO_BF16) dec_launch_rows<__bf16>(p, (H / Hkv) * q_len, (unsigned)rows, st);
    else dec_launch_rows<_Float16>(p, (H / Hkv) * q_len, (unsigned)rows, st);
  } else if (dtype == MIO_BF16) dec_launch<__bf16>(p, grid, st);
  else dec_launch<_Float16>(p, grid, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("decode_paged launch: ") + hipGetErrorString(e));
  return 0;
}

// ---- reshape_and_cache: one workgroup per sequence, 16-byte chunks over (Hkv, D) --------------------
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(const uint16_t* __restrict__ key,
                                                                const uint16_t* __restrict__ value,
                                                                uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                                const int32_t* __restrict__ bt,
                                                                const int32_t* __restrict__ cl, int64_t ks_b,
                                                                int64_t ks_h, int64_t vs_b, int64_t vs_h, int Hkv,
                                                                int D, int L, int layer, int bs, int max_blocks) {
  const int b = blockIdx.x;
  const int pos = cl[b] - 1;  // write position (attention_kernels.py:858)
  if (pos < 0 || pos / bs >= max_blocks) return;  // empty sequence / context longer than the block table row: nothing written
  const int pb = bt[(int64_t)b * max_blocks + pos / bs];
  const int64_t tok_stride = (int64_t)Hkv * D;
  const int64_t dst = ((int64_t)pb * L + layer) * bs * tok_stride + (int64_t)(pos % bs) * tok_stride;
  const int cpr = D >> 3;
  for (int i = threadIdx.x; i < Hkv * cpr; i += 256) {
    const int hh = i / cpr, c = i % cpr;
    const u32x4_t kk = *(const u32x4_t*)(key + b * ks_b + hh * ks_h + 8 * c);
    const u32x4_t vv = *(const u32x4_t*)(value + b * vs_b + hh * vs_h + 8 * c);
    *(u32x4_t*)(kc + dst + (int64_t)hh * D + 8 * c) = kk;
    *(u32x4_t*)(vc + dst + (int64_t)hh * D + 8 * c) = vv;
  }
}

extern "C" int mio_reshape_and_cache(const void* key, const void* value, void* k_cache, void* v_cache,
                                     const int32_t* block_tables, const int32_t* context_lengths,
                                     const int64_t k_stride[2], const int64_t v_stride[2], int32_t B, int32_t Hkv,
                                     int32_t D, int32_t num_layers, int32_t layer_idx, int32_t block_size,
                                     int32_t max_blocks_per_seq, int32_t dtype, void* stream) {
  MIO_CHECK(key && value && k_cache && v_cache && block_tables && context_lengths, "mio_reshape_and_cache: null pointer");
  MIO_CHECK(B > 0 && Hkv > 0 && D >= 8 && D % 8 == 0, "mio_reshape_and_cache: bad sizes");
  MIO_CHECK(layer_idx >= 0 && layer_idx < num_layers && block_size > 0, "mio_reshape_and_cache: bad cache geometry");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_reshape_and_cache: dtype must be bf16 or fp16");
  MIO_CHECK(k_stride[0] % 8 == 0 && k_stride[1] % 8 == 0 && v_stride[0] % 8 == 0 && v_stride[1] % 8 == 0 &&
                mio_aligned16(key) && mio_aligned16(value) && mio_aligned16(k_cache) && mio_aligned16(v_cache),
            "mio_reshape_and_cache: 16-byte alignment");
  hipLaunchKernelGGL(reshape_and_cache_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)key, (const uint16_t*)value, (uint16_t*)k_cache, (uint16_t*)v_cache,
                     block_tables, context_lengths, k_stride[0], k_stride[1], v_stride[0], v_stride[1], Hkv, D,
                     num_layers, layer_idx, block_size, max_blocks_per_seq);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("reshape_and_cache launch: ") + hipGetErrorString(e));
  return 0;
}
