// One translation unit per dtype: compiled with -DGEMM_TYPE_ID={0,1}.
#include "gemm_kernel.h"

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
    if (big >= 256) return launch_cfg<256, 256, 2, 4, ACT>(p, stream);
    return launch_cfg<128, 128, 2, 2, ACT>(p, stream);
  }
}

template <>
int gemm_launch<GT>(GemmDev p, int act, hipStream_t stream) {
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
