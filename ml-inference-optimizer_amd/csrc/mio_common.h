// Shared device/host helpers for libmio_hip.so (gfx950 only; no portability layers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "mio_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define MIO_LDS __attribute__((address_space(3)))

// ---- dtype traits: everything the kernels need to be generic over bf16 / fp16 ---------------
template <typename T>
struct DT;

template <>
struct DT<__bf16> {
  using elem = __bf16;
  using x8 = bf16x8_t;
  using x4 = bf16x4_t;
  using x2 = bf16x2_t;
  static __device__ __forceinline__ f32x16_t mfma32(x8 a, x8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4_t mfma16(x8 a, x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 ds_read_tr(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((MIO_LDS x4*)(lds_ptr));
  }
};

template <>
struct DT<_Float16> {
  using elem = _Float16;
  using x8 = f16x8_t;
  using x4 = f16x4_t;
  using x2 = f16x2_t;
  static __device__ __forceinline__ f32x16_t mfma32(x8 a, x8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4_t mfma16(x8 a, x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 ds_read_tr(const void* lds_ptr) {
    typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_raw;
    return __builtin_bit_cast(x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((MIO_LDS fp16x4_raw*)(lds_ptr)));
  }
};

// two fp32 -> packed pair of T (RNE); bit pattern returned as u32
template <typename T>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  typename DT<T>::x2 v = __builtin_convertvector((f32x2_t){a, b}, typename DT<T>::x2);
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T>
__device__ __forceinline__ float to_f32(T x) {
  return (float)x;
}

// value held by the other 32-lane half of the wave (lane ^ 32)
__device__ __forceinline__ float other_half(float x) {
  uint32_t u = __float_as_uint(x);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  // r[0]: lanes 0-31 keep own, lanes 32-63 get lower's; r[1]: lanes 0-31 get upper's, 32-63 keep own
  return (threadIdx.x & 32) ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---- host-side error plumbing -----------------------------------------------------------------
void mio_set_error(const std::string& msg);
int mio_fail(const std::string& msg);  // sets the message, returns -1

#define MIO_CHECK(cond, msg) \
  do {                       \
    if (!(cond)) return mio_fail(msg); \
  } while (0)

#define MIO_HIP_OK(expr)                                                                  \
  do {                                                                                    \
    hipError_t e__ = (expr);                                                              \
    if (e__ != hipSuccess) return mio_fail(std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

#ifdef MIO_DIAG
// diagnostic build only (libmio_hip_dbg.so): integer knobs the tools set between launches of one process
extern "C" void mio_dbg_set(int key, int value);
extern "C" int mio_dbg_get(int key);
#endif

static inline bool mio_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
