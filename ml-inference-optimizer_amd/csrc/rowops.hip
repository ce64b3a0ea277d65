// HBM-bound row kernels either side of the MFMA kernels: LayerNorm / residual+LayerNorm and the
// (o, lse) merge of two partial attention states.  One wave per row, 16-byte vector accesses.
//   LayerNorm: reference kernels/triton/layernorm_kernels.py:35-188 (_layernorm_fwd_kernel,
//              _layernorm_residual_fwd_kernel; wrapper triton_layernorm :191-276).
//   merge:     the (alpha, beta) rescale of kernels/triton/attention_kernels.py:1573-1585 on
//              normalised states.
#include "mio_common.h"

template <typename T>
__device__ __forceinline__ void unpack8(const u32x4_t raw, float (&f)[8]) {
  const typename DT<T>::x8 v = __builtin_bit_cast(typename DT<T>::x8, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T>
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  u32x4_t w = {pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3]), pack2<T>(f[4], f[5]), pack2<T>(f[6], f[7])};
  return w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// One wave per row; each lane owns chunks lane, lane+64, ... of 8 elements (cols % 8 == 0, cols <= 8*64*CH).
template <typename T, int CH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                        const T* __restrict__ w, const T* __restrict__ bias,
                                                        T* __restrict__ y, T* __restrict__ sum_out, int64_t rows,
                                                        int cols, float eps, float alpha, int yblk) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = cols >> 3;
  float v[CH][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      unpack8<T>(*(const u32x4_t*)(x + row * cols + 8 * c), v[j]);
      if (res != nullptr) {
        float rr[8];
        unpack8<T>(*(const u32x4_t*)(res + row * cols + 8 * c), rr);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[j][i] += alpha * rr[i];
        if (sum_out != nullptr) {
          // the stored sum is rounded to T; normalise the rounded value so y == LN(sum_out) exactly
          const u32x4_t packed = pack8<T>(v[j]);
          *(u32x4_t*)(sum_out + row * cols + 8 * c) = packed;
          unpack8<T>(packed, v[j]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[j][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[j][i] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[j][i] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)cols + eps);
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int c = lane + 64 * j;
    if (c < nch) {
      float wv[8], bv[8], o[8];
      unpack8<T>(*(const u32x4_t*)(w + 8 * c), wv);
      if (bias != nullptr) unpack8<T>(*(const u32x4_t*)(bias + 8 * c), bv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = (v[j][i] - mean) * rstd * wv[i] + (bias != nullptr ? bv[i] : 0.f);
      // yblk: blocked activation layout (gemm_kernel.h GemmDev::x_blk): chunk c of the row is chunk c & 3 of the 64-byte
      // row segment in K-tile block c >> 2
      const int64_t yo = yblk ? (((row >> 8) * (cols >> 5) + (c >> 2)) * 256 + (row & 255)) * 32 + (c & 3) * 8
                              : row * cols + 8 * c;
      *(u32x4_t*)(y + yo) = pack8<T>(o);
    }
  }
}

template <typename T>
static int ln_launch(const void* x, const void* res, const void* w, const void* b, void* y, void* sum_out, int64_t rows,
                     int cols, float eps, float alpha, hipStream_t st, int yblk) {
  const int nch = cols / 8;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define LN_GO(CH)                                                                                             \
  hipLaunchKernelGGL((layernorm_kernel<T, CH>), grid, block, 0, st, (const T*)x, (const T*)res, (const T*)w, \
                     (const T*)b, (T*)y, (T*)sum_out, rows, cols, eps, alpha, yblk)
  if (nch <= 64) LN_GO(1);
  else if (nch <= 128) LN_GO(2);
  else if (nch <= 256) LN_GO(4);
  else if (nch <= 512) LN_GO(8);
  else return mio_fail("mio_layernorm_fwd: cols > 4096 not supported");
#undef LN_GO
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("layernorm launch: ") + hipGetErrorString(e));
  return 0;
}

static int ln_entry(const char* who, const void* x, const void* residual, const void* weight, const void* bias, void* y,
                    void* sum_out, int64_t rows, int32_t cols, float eps, float alpha, int32_t dtype, void* stream, int yblk) {
  (void)who;
  MIO_CHECK(x && weight && y, "mio_layernorm_fwd: x, weight, y must be non-null");
  MIO_CHECK(rows >= 0 && cols > 0 && cols % 8 == 0, "mio_layernorm_fwd: cols must be a positive multiple of 8");
  MIO_CHECK(!yblk || cols % 32 == 0, "mio_layernorm_fwd_bx: cols must be a multiple of 32");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_layernorm_fwd: dtype must be bf16 or fp16");
  MIO_CHECK(mio_aligned16(x) && mio_aligned16(residual) && mio_aligned16(weight) && mio_aligned16(bias) &&
                mio_aligned16(y) && mio_aligned16(sum_out),
            "mio_layernorm_fwd: pointers must be 16-byte aligned");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MIO_BF16)
    return ln_launch<__bf16>(x, residual, weight, bias, y, sum_out, rows, cols, eps, alpha, st, yblk);
  return ln_launch<_Float16>(x, residual, weight, bias, y, sum_out, rows, cols, eps, alpha, st, yblk);
}

extern "C" int mio_layernorm_fwd(const void* x, const void* residual, const void* weight, const void* bias, void* y,
                                 void* sum_out, int64_t rows, int32_t cols, float eps, float alpha, int32_t dtype,
                                 void* stream) {
  return ln_entry("mio_layernorm_fwd", x, residual, weight, bias, y, sum_out, rows, cols, eps, alpha, dtype, stream, 0);
}

// y in the blocked activation layout (include/mio_hip.h): ceil(rows / 256) * 256 x cols elements
extern "C" int mio_layernorm_fwd_bx(const void* x, const void* residual, const void* weight, const void* bias, void* yb,
                                    void* sum_out, int64_t rows, int32_t cols, float eps, float alpha, int32_t dtype,
                                    void* stream) {
  return ln_entry("mio_layernorm_fwd_bx", x, residual, weight, bias, yb, sum_out, rows, cols, eps, alpha, dtype, stream, 1);
}

// ---- merge of two normalised partial attention states ------------------------------------------
// rows laid out [B, Sq, H]; lse laid out [B, H, Sq].  One thread per (row, 4 elements of D).
template <typename T>
__global__ __launch_bounds__(256) void attn_merge_kernel(float* __restrict__ o_a, float* __restrict__ lse_a,
                                                         const float* __restrict__ o_b,
                                                         const float* __restrict__ lse_b, T* __restrict__ o_out,
                                                         int B, int Sq, int H, int D) {
  const int dq = D >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Sq * H * dq;
  if (idx >= total) return;
  const int64_t row = idx / dq;
  const int c = (int)(idx - row * dq);
  const int hh = (int)(row % H);
  const int64_t bs = row / H;
  const int s = (int)(bs % Sq);
  const int b = (int)(bs / Sq);
  const int64_t li = ((int64_t)b * H + hh) * Sq + s;
  const float la = lse_a[li], lb = lse_b[li];
  const float mx = fmaxf(la, lb);
  float wa = 0.f, wb = 0.f;  // lse itself is updated by attn_merge_lse_kernel after all rows are read
  if (mx != -INFINITY) {
    const float ea = __expf(la - mx), eb = __expf(lb - mx);
    const float inv = 1.0f / (ea + eb);
    wa = ea * inv;
    wb = eb * inv;
  }
  const f32x4_t a = *(const f32x4_t*)(o_a + row * D + 4 * c);
  const f32x4_t bb = *(const f32x4_t*)(o_b + row * D + 4 * c);
  f32x4_t o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = wa * a[i] + wb * bb[i];
  *(f32x4_t*)(o_a + row * D + 4 * c) = o;
  if (o_out != nullptr) {
    u32x2_t w = {pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3])};
    *(u32x2_t*)(o_out + row * D + 4 * c) = w;
  }
}

__global__ __launch_bounds__(256) void attn_merge_lse_kernel(float* __restrict__ lse_a,
                                                             const float* __restrict__ lse_b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float la = lse_a[i], lb = lse_b[i];
  const float mx = fmaxf(la, lb);
  lse_a[i] = (mx == -INFINITY) ? -INFINITY : mx + __logf(__expf(la - mx) + __expf(lb - mx));
}

extern "C" int mio_attn_merge(float* o_a, float* lse_a, const float* o_b, const float* lse_b, void* o_out, int32_t B,
                              int32_t Sq, int32_t H, int32_t D, int32_t dtype, void* stream) {
  MIO_CHECK(o_a && lse_a && o_b && lse_b, "mio_attn_merge: null state pointer");
  MIO_CHECK(B > 0 && Sq >= 0 && H > 0 && D > 0 && D % 4 == 0, "mio_attn_merge: bad sizes (D % 4 == 0)");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_attn_merge: dtype must be bf16 or fp16");
  MIO_CHECK(mio_aligned16(o_a) && mio_aligned16(o_b) && mio_aligned16(o_out), "mio_attn_merge: 16-byte alignment");
  if (Sq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)B * Sq * H * (D / 4);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (dtype == MIO_BF16)
    hipLaunchKernelGGL(attn_merge_kernel<__bf16>, grid, block, 0, st, o_a, lse_a, o_b, lse_b, (__bf16*)o_out, B, Sq, H, D);
  else
    hipLaunchKernelGGL(attn_merge_kernel<_Float16>, grid, block, 0, st, o_a, lse_a, o_b, lse_b, (_Float16*)o_out, B, Sq, H, D);
  const int64_t n = (int64_t)B * H * Sq;
  hipLaunchKernelGGL(attn_merge_lse_kernel, dim3((unsigned)((n + 255) / 256)), block, 0, st, lse_a, lse_b, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("attn_merge launch: ") + hipGetErrorString(e));
  return 0;
}
