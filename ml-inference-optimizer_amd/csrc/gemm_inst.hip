// One translation unit per dtype: compiled with -DGEMM_TYPE_ID={0,1}.
#include <cstdio>
#include <cstdlib>

#include "gemm2x_kernel.h"
#include "gemm4w16_kernel.h"
#include "gemm4w16p_kernel.h"
#include "gemm4w_kernel.h"
#include "gemm8p_kernel.h"

#if GEMM_TYPE_ID == 0
using GT = __bf16;
#else
using GT = _Float16;
#endif

template <int BM, int BN, int WM, int WN, int ACT>
static int launch_cfg(GemmDev p, hipStream_t stream) {
  constexpr bool GATE = (ACT == MIO_ACT_SWIGLU);
  constexpr size_t smem = 2 * (size_t)(BM * GEMM_BK * 2 + BN * GEMM_BK * 2 * (GATE ? 2 : 1));
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.tiles_n = (p.N + BN - 1) / BN;
  auto kern = gemm_bias_act_kernel<GT, BM, BN, WM, WN, ACT>;
  static bool attr_set = false;  // > 64 KiB dynamic LDS needs the opt-in once per kernel
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) return mio_fail(std::string("gemm: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm launch: ") + hipGetErrorString(e));
  return 0;
}

static int gemm_var() {  // MIO_GEMM_VAR=<bits>: timing-only ablations of the ACT_NONE 8-phase kernel (tuning aid)
  static const int v = [] {
    const char* e = std::getenv("MIO_GEMM_VAR");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}

template <int ACT, int VAR = 0>
static int launch_8p(GemmDev p, hipStream_t stream) {
  if constexpr (ACT == MIO_ACT_NONE && VAR == 0) {
    switch (gemm_var()) {
      case 1: return launch_8p<ACT, 1>(p, stream);
      case 2: return launch_8p<ACT, 2>(p, stream);
      case 4: return launch_8p<ACT, 4>(p, stream);
      case 8: return launch_8p<ACT, 8>(p, stream);
      case 12: return launch_8p<ACT, 12>(p, stream);
      default: break;
    }
  }
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 255) / 256;
  auto kern = gemm8p_kernel<GT, ACT, VAR>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G8_SMEM);
    if (e != hipSuccess) return mio_fail(std::string("gemm8p: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), G8_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm8p launch: ") + hipGetErrorString(e));
  return 0;
}

static int gemm_var();
template <int ACT, int VAR = 0>
static int launch_4w(GemmDev p, hipStream_t stream) {
  if constexpr (ACT == MIO_ACT_NONE && VAR == 0) {
    switch (gemm_var()) {
      case 4: return launch_4w<ACT, 4>(p, stream);
      case 8: return launch_4w<ACT, 8>(p, stream);
      case 12: return launch_4w<ACT, 12>(p, stream);
      default: break;
    }
  }
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 255) / 256;
  auto kern = gemm4w_kernel<GT, ACT, VAR>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G4_SMEM);
    if (e != hipSuccess) return mio_fail(std::string("gemm4w: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), G4_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm4w launch: ") + hipGetErrorString(e));
  return 0;
}

static int gemm_var();
template <int ACT, int VAR = 0>
static int launch_4w16(GemmDev p, hipStream_t stream) {
  if constexpr (ACT == MIO_ACT_NONE && VAR == 0) {
    if (p.dbg != nullptr) {
      switch (gemm_var()) {
        case 4: return launch_4w16<ACT, 36>(p, stream);
        case 8: return launch_4w16<ACT, 40>(p, stream);
        case 12: return launch_4w16<ACT, 44>(p, stream);
        case 16: return launch_4w16<ACT, 48>(p, stream);
        case 28: return launch_4w16<ACT, 60>(p, stream);
        default: return launch_4w16<ACT, 32>(p, stream);
      }
    }
    if (gemm_var() == 16) return launch_4w16<ACT, 16>(p, stream);
  }
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 255) / 256;
  auto kern = gemm4w16_kernel<GT, ACT, VAR>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G6_SMEM);
    if (e != hipSuccess) return mio_fail(std::string("gemm4w16: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), G6_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm4w16 launch: ") + hipGetErrorString(e));
  return 0;
}

// persistent variant with the overlapped epilogue (no residual operand, K >= 128)
template <int ACT, int SPK = 0>
static int launch_4w16p(GemmDev p, hipStream_t stream) {
  if constexpr (SPK == 0) {  // stores per K-tile so that all 32 chunks of a tile leave during the next one
    const int nk = p.K / 32;
    if (nk >= 32) return launch_4w16p<ACT, 1>(p, stream);
    if (nk >= 16) return launch_4w16p<ACT, 2>(p, stream);
    return launch_4w16p<ACT, 4>(p, stream);
  }
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 255) / 256;
  constexpr bool CAN_STAMP = (ACT == MIO_ACT_NONE && SPK == 1);
  auto kern = gemm4w16p_kernel<GT, ACT, (SPK == 0 ? 1 : SPK)>;
  if constexpr (CAN_STAMP) {
    if (p.dbg != nullptr) kern = gemm4w16p_kernel<GT, ACT, 1, true>;
  }
  static bool attr_set = false;
  static int ncu = 256;
  if (!attr_set || p.dbg != nullptr) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G6P_SMEM);
    if (e != hipSuccess) return mio_fail(std::string("gemm4w16p: hipFuncSetAttribute: ") + hipGetErrorString(e));
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n & ~7;  // whole XCD groups, so tile % 8 keeps naming the XCD
    attr_set = true;
  }
  const int tiles = p.tiles_m * p.tiles_n;
  hipLaunchKernelGGL(kern, dim3(tiles < ncu ? tiles : ncu), dim3(256), G6P_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm4w16p launch: ") + hipGetErrorString(e));
  return 0;
}

template <int ACT>
static int launch_2x(GemmDev p, hipStream_t stream) {
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 127) / 128;
  auto kern = gemm2x_kernel<GT, ACT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM);
    if (e != hipSuccess) return mio_fail(std::string("gemm2x: hipFuncSetAttribute: ") + hipGetErrorString(e));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), G2_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm2x launch: ") + hipGetErrorString(e));
  return 0;
}

static int gemm_impl() { return mio_gemm_impl(); }  // MIO_GEMM_IMPL (gemm_api.hip)

template <int ACT>
static int launch_act(const GemmDev& p, hipStream_t stream) {
  // Big tiles when they still fill the chip (>= 256 workgroups), else 128x128.
  constexpr bool GATE = (ACT == MIO_ACT_SWIGLU);
  if constexpr (GATE) {
    const int64_t big = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    if (big >= 256) return launch_cfg<256, 128, 2, 4, ACT>(p, stream);
    return launch_cfg<128, 64, 2, 2, ACT>(p, stream);
  } else {
    const int64_t big = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (big >= 256) {
      if (gemm_impl() == 1) return launch_cfg<256, 256, 2, 4, ACT>(p, stream);
      if (gemm_impl() == 2) return launch_8p<ACT>(p, stream);
      if (gemm_impl() == 3) return launch_4w<ACT>(p, stream);
      // the 16x16x32 kernel addresses operands with 32-bit per-tile byte offsets and needs whole K-tiles
      const bool fits = (p.K % 32 == 0) && (p.ldx * 512 < (int64_t)0x7fffffff) && (p.ldw * 512 < (int64_t)0x7fffffff);
      if (!fits) return launch_cfg<256, 256, 2, 4, ACT>(p, stream);
      if (gemm_impl() == 4) return launch_2x<ACT>(p, stream);
      // no residual: persistent kernel with the overlapped epilogue (MIO_GEMM_IMPL=4w16 keeps the one-tile kernel)
      if (gemm_impl() != 6 && p.res == nullptr && p.K >= 256 && p.K % 64 == 0 && p.ldy * 512 < (int64_t)0x7fffffff)
        return launch_4w16p<ACT>(p, stream);
      return launch_4w16<ACT>(p, stream);
    }
    return launch_cfg<128, 128, 2, 2, ACT>(p, stream);
  }
}

template <>
int gemm_launch<GT>(GemmDev p, int act, hipStream_t stream) {
  if (const char* e = std::getenv("MIO_GEMM_DBG_PTR")) {
    p.dbg = (unsigned long long*)std::strtoull(e, nullptr, 0);
    static int once = 0;
    if (!once++) fprintf(stderr, "[mio] gemm stamps -> %p (act %d)\n", (void*)p.dbg, act);
  }
  switch (act) {
    case MIO_ACT_NONE: return launch_act<MIO_ACT_NONE>(p, stream);
    case MIO_ACT_GELU_TANH: return launch_act<MIO_ACT_GELU_TANH>(p, stream);
    case MIO_ACT_GELU_ERF: return launch_act<MIO_ACT_GELU_ERF>(p, stream);
    case MIO_ACT_RELU: return launch_act<MIO_ACT_RELU>(p, stream);
    case MIO_ACT_SILU: return launch_act<MIO_ACT_SILU>(p, stream);
    case MIO_ACT_SWIGLU: return launch_act<MIO_ACT_SWIGLU>(p, stream);
  }
  return mio_fail("gemm: unknown activation");
}
