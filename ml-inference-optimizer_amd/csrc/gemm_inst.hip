// One translation unit per dtype: compiled with -DGEMM_TYPE_ID={0,1}.
// The product library reads no environment variable; A/B switches and stamp instantiations: -DMIO_DIAG (make dbg).
#include <cstdlib>
#include <mutex>

#include "gemm4w16_kernel.h"
#include "gemm4w16p_kernel.h"
#include "gemm8w_kernel.h"

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
  static std::once_flag once;  // > 64 KiB dynamic LDS needs the opt-in once per kernel
  static hipError_t ea = hipSuccess;
  std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
  if (ea != hipSuccess) return mio_fail(std::string("gemm: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm launch: ") + hipGetErrorString(e));
  return 0;
}

#ifdef MIO_DIAG
static int gemm_var() {  // MIO_GEMM_VAR=<bits>: timing-only ablations of the ACT_NONE one-tile kernel (tuning aid)
  static const int v = [] {
    const char* e = std::getenv("MIO_GEMM_VAR");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}
#endif

#ifdef MIO_DIAG  // the one-wave-per-SIMD kernels of rounds 1-2: kept in the diagnostic library for A/B runs only
template <int ACT, int VAR = 0>
static int launch_4w16(GemmDev p, hipStream_t stream) {
#ifdef MIO_DIAG
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
#endif
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + 255) / 256;
  auto kern = gemm4w16_kernel<GT, ACT, VAR>;
  static std::once_flag once;
  static hipError_t ea = hipSuccess;
  std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G6_SMEM); });
  if (ea != hipSuccess) return mio_fail(std::string("gemm4w16: hipFuncSetAttribute: ") + hipGetErrorString(ea));
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
  auto kern = gemm4w16p_kernel<GT, ACT, (SPK == 0 ? 1 : SPK)>;
  static std::once_flag once;
  static hipError_t ea = hipSuccess;
  static int ncu = 256;
  std::call_once(once, [&] {
    ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G6P_SMEM);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n & ~7;  // whole XCD groups, so tile % 8 keeps naming the XCD
  });
  if (ea != hipSuccess) return mio_fail(std::string("gemm4w16p: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  const int tiles = p.tiles_m * p.tiles_n;
#ifdef MIO_DIAG
  if constexpr (ACT == MIO_ACT_NONE && SPK == 1) {
    if (p.dbg != nullptr) {  // in-kernel stamps (tools/gemm_stamps_p.py)
      auto kd = gemm4w16p_kernel<GT, ACT, 1, true>;
      hipError_t ed = hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, G6P_SMEM);
      if (ed != hipSuccess) return mio_fail(std::string("gemm4w16p (stamps): hipFuncSetAttribute: ") + hipGetErrorString(ed));
      hipLaunchKernelGGL(kd, dim3(tiles < ncu ? tiles : ncu), dim3(256), G6P_SMEM, stream, p);
      hipError_t e2 = hipGetLastError();
      if (e2 != hipSuccess) return mio_fail(std::string("gemm4w16p (stamps) launch: ") + hipGetErrorString(e2));
      return 0;
    }
  }
#endif
  hipLaunchKernelGGL(kern, dim3(tiles < ncu ? tiles : ncu), dim3(256), G6P_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm4w16p launch: ") + hipGetErrorString(e));
  return 0;
}

#endif  // MIO_DIAG

// eight-wave ping-pong kernel (gemm8w_kernel.h), persistent; `one_tile`: one workgroup per tile instead (A/B only)
template <int ACT, bool RES, int VAR = 0, int FOLD = 0>
static int launch_8w(GemmDev p, hipStream_t stream, bool one_tile = false) {
  constexpr int BN = (ACT == MIO_ACT_SWIGLU) ? 128 : 256;
  constexpr int G8_SMEM = FOLD ? G8_SMEM_FOLD : ::G8_SMEM;  // (shadows the namespace-scope constant for this launcher)
  p.tiles_m = (int)((p.M + 255) / 256);
  p.tiles_n = (p.N + BN - 1) / BN;
  auto kern = gemm8w_kernel<GT, ACT, RES, VAR, FOLD>;
  static std::once_flag once;
  static hipError_t ea = hipSuccess;
  static int ncu = 256;
  std::call_once(once, [&] {
    ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G8_SMEM);
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      ncu = n & ~7;  // whole XCD groups, so tile % 8 keeps naming the XCD
  });
  if (ea != hipSuccess) return mio_fail(std::string("gemm8w: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  const int tiles = p.tiles_m * p.tiles_n;
  int wgs = ncu;
#ifdef MIO_DIAG  // mio_dbg_set(7, n): a workgroup budget below the CU count, for co-running with another stream's kernel
  if (mio_dbg_get(7) >= 8 && mio_dbg_get(7) < ncu) wgs = mio_dbg_get(7) & ~7;  // (tools/overlap_probe.py)
#endif
  hipLaunchKernelGGL(kern, dim3((one_tile || tiles < wgs) ? tiles : wgs), dim3(G8_THREADS), G8_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("gemm8w launch: ") + hipGetErrorString(e));
  return 0;
}
template <int ACT, int VAR = 0>
static int launch_8w_res(const GemmDev& p, hipStream_t stream, bool one_tile = false) {
  if constexpr (ACT != MIO_ACT_SWIGLU && VAR == 0) {
    if (p.res != nullptr) return launch_8w<ACT, true>(p, stream, one_tile);
  }
  return launch_8w<ACT, false, VAR>(p, stream, one_tile);
}

static int gemm_impl() { return mio_gemm_impl(); }  // MIO_GEMM_IMPL (gemm_api.hip)

template <int ACT>
static int launch_act(const GemmDev& p, hipStream_t stream) {
  // Big tiles when they still fill the chip (>= 256 workgroups), else 128x128.
  constexpr bool GATE = (ACT == MIO_ACT_SWIGLU);
  if constexpr (GATE) {
    if (p.w_blk == 2) {  // interleaved blocked gate / up weight (gemm_api.hip checked the shape)
      if (p.ln_stats != nullptr) return launch_8w<ACT, false, 0, 1>(p, stream);  // LayerNorm applied in the read-out
      return launch_8w<ACT, false>(p, stream);
    }
    const int64_t big = ((p.M + 255) / 256) * ((p.N + 127) / 128);
    if (big >= 256) return launch_cfg<256, 128, 2, 4, ACT>(p, stream);
    return launch_cfg<128, 64, 2, 2, ACT>(p, stream);
  } else {
    const int64_t big = ((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (big >= 256) {
      if (gemm_impl() == 1) return launch_cfg<256, 256, 2, 4, ACT>(p, stream);
      // the 16x16x32 kernels address operands with 32-bit per-tile byte offsets and need whole K-tiles (>= 4 of them)
      const bool fits = (p.K % 32 == 0) && p.K >= 128 && (p.ldx * 512 < (int64_t)0x7fffffff) &&
                        (p.ldw * 512 < (int64_t)0x7fffffff) && (p.ldy * 512 < (int64_t)0x7fffffff) &&
                        (p.res == nullptr || p.ldr * 512 < (int64_t)0x7fffffff);
      if (!fits) return launch_cfg<256, 256, 2, 4, ACT>(p, stream);
      // LayerNorm fold (mio_gemm_ln_bw checked the shape): consumer = projection behind the LayerNorm, producer = residual GEMM
      if constexpr (ACT == MIO_ACT_NONE || ACT == MIO_ACT_GELU_TANH) {
        if (p.ln_stats != nullptr) return launch_8w<ACT, false, 0, 1>(p, stream);
      }
      if constexpr (ACT == MIO_ACT_NONE) {
        if (p.stats_out != nullptr) return launch_8w<ACT, true, 0, 2>(p, stream);
      }
#ifdef MIO_DIAG
      if constexpr (ACT == MIO_ACT_NONE) {
        if (gemm_impl() == 8 && p.dbg != nullptr) return launch_8w<ACT, false, 128>(p, stream);  // stamps
        if (gemm_impl() == 24) return launch_8w_res<ACT, 2048>(p, stream);
        if (gemm_impl() == 10) return launch_8w_res<ACT, 4>(p, stream);
        if (gemm_impl() == 13) return launch_8w_res<ACT, 16>(p, stream);
        if (gemm_impl() == 16) return launch_8w_res<ACT, 64>(p, stream);
      }
      if constexpr (ACT == MIO_ACT_GELU_TANH) {
        if (gemm_impl() == 20) return launch_8w<ACT, false, 256>(p, stream);  // scalar activation math
      }
      if (gemm_impl() == 9) return launch_8w_res<ACT>(p, stream, true);  // one workgroup per tile
      if (gemm_impl() == 5 || gemm_impl() == 6) {  // rounds 1-2: persistent where it applied, else one tile per workgroup
        if (gemm_impl() == 5 && p.res == nullptr && p.K >= 256 && p.K % 64 == 0) return launch_4w16p<ACT>(p, stream);
        return launch_4w16<ACT>(p, stream);
      }
#endif
      return launch_8w_res<ACT>(p, stream);
    }
    return launch_cfg<128, 128, 2, 2, ACT>(p, stream);
  }
}

template <>
int gemm_launch<GT>(GemmDev p, int act, hipStream_t stream) {
#ifdef MIO_DIAG
  static const char* dbg_ptr = std::getenv("MIO_GEMM_DBG_PTR");
  if (dbg_ptr != nullptr) p.dbg = (unsigned long long*)std::strtoull(dbg_ptr, nullptr, 0);
  static const char* gm = std::getenv("MIO_GEMM_GROUP_M");  // tile-order sweep (tools/dbg)
  if (gm != nullptr) p.group_m = std::atoi(gm);
  if (mio_dbg_get(5) > 0) p.group_m = mio_dbg_get(5);
#endif
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
