// One translation unit per (dtype, padded head dim): compiled with -DFA_TYPE_ID={0,1} -DFA_D={64,96,128}.
#include "fa3_fwd_kernel.h"

#if FA_TYPE_ID == 0
using FaT = __bf16;
#else
using FaT = _Float16;
#endif

template <bool CAUSAL, int MASK>
static int launch_one(const FaDev& p, hipStream_t stream) {
  const int grid = p.nqblk * p.B * p.H;
  const size_t smem = FaSmem<FA_D>::TOTAL;
  auto kern = fa3_fwd_kernel<FaT, FA_D, CAUSAL, MASK>;
  static bool attr_set = false;  // > 64 KiB dynamic LDS (D = 128) needs the opt-in once per kernel
  if (!attr_set && smem > 48 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd: hipFuncSetAttribute: ") + hipGetErrorString(ea));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd launch: ") + hipGetErrorString(e));
  return 0;
}

template <>
int fa3_launch<FaT, FA_D>(const FaDev& p, int causal, int mask_kind, hipStream_t stream) {
  if (causal) {
    if (mask_kind == MIO_MASK_NONE) return launch_one<true, 0>(p, stream);
    if (mask_kind == MIO_MASK_KEEP_U8) return launch_one<true, 1>(p, stream);
    return launch_one<true, 2>(p, stream);
  }
  if (mask_kind == MIO_MASK_NONE) return launch_one<false, 0>(p, stream);
  if (mask_kind == MIO_MASK_KEEP_U8) return launch_one<false, 1>(p, stream);
  return launch_one<false, 2>(p, stream);
}
