// One translation unit per (dtype, padded head dim): compiled with -DFA_TYPE_ID={0,1} -DFA_D={64,96,128}.
#include <cstdlib>

#include "fa3_fwd2_kernel.h"
#include "fa3_fwd3_kernel.h"

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

// second structure (one wave per SIMD, 64 query rows per wave): no user mask, D % 32 == 0 instantiations
template <bool CAUSAL>
static int launch_two(FaDev p, hipStream_t stream) {
  p.nqblk = (p.Sq + FA2_BM - 1) / FA2_BM;
  // causal: a workgroup takes query blocks i and nqblk-1-i back to back -> every workgroup does the same work
  p.qgrid = CAUSAL ? (p.nqblk + 1) / 2 : p.nqblk;
  if (const char* e = std::getenv("MIO_FA_ORDER")) p.xcd_remap |= std::atoi(e);  // tuning aid: 2 = light-first, 4 = block-major
  const int grid = p.qgrid * p.B * p.H;
  const size_t smem = FaSmem<FA_D>::TOTAL;
  auto kern = fa3_fwd2_kernel<FaT, FA_D, CAUSAL>;
  static bool attr_set = false;
  if (!attr_set && smem > 48 * 1024) {
    hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd2: hipFuncSetAttribute: ") + hipGetErrorString(ea));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd2 launch: ") + hipGetErrorString(e));
  return 0;
}

// third structure (software-pipelined across KV tiles): no user mask
template <bool CAUSAL>
static int launch_three(FaDev p, hipStream_t stream) {
  p.nqblk = (p.Sq + FA3_BM - 1) / FA3_BM;
  p.qgrid = CAUSAL ? (p.nqblk + 1) / 2 : p.nqblk;
  const int grid = p.qgrid * p.B * p.H;
  const size_t smem = FA3_STAGES * FaSmem<FA_D>::STAGE;
  auto kern = fa3_fwd3_kernel<FaT, FA_D, CAUSAL>;
  if (const char* e = std::getenv("MIO_FA_DBG_PTR")) {  // diagnostic build with in-kernel phase stamps (tools/fa_stamps.py)
    kern = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, true>;
    p.mask = (const void*)std::strtoull(e, nullptr, 0);
  }
  static bool attr_set = false;
  if (!attr_set || std::getenv("MIO_FA_DBG_PTR")) {
    hipError_t ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd3: hipFuncSetAttribute: ") + hipGetErrorString(ea));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd3 launch: ") + hipGetErrorString(e));
  return 0;
}

static int fa_impl() {  // MIO_FA_IMPL=1 / 2 / 3 force one structure for A/B runs (0 = the rule in fa3_launch)
  static const int v = [] {
    const char* e = std::getenv("MIO_FA_IMPL");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}

template <>
int fa3_launch<FaT, FA_D>(const FaDev& p, int causal, int mask_kind, hipStream_t stream) {
  // Structure choice (measured on MI355X, B8 S4096, random data; MIO_FA_IMPL=1 / 2 / 3 forces one for A/B runs):
  //   no user mask, Sq > 128: the software-pipelined kernel at every head dim --
  //     D64 causal 0.359 ms (766 TFLOP/s) vs 0.442 two-waves-per-SIMD / 0.52 sequential one-wave; non-causal 0.626 vs 0.79;
  //     D128 causal 0.284 ms (967 TFLOP/s) vs 0.361 sequential one-wave; D80 non-causal 0.828 ms (830) vs 1.158;
  //   user masks and Sq <= 128: the two-waves-per-SIMD kernel.
  // (the pipelined kernel addresses K / V tiles with 32-bit byte offsets from the (batch, head) base)
  const bool span32 = (int64_t)p.Sk * p.ks_s * 2 < (1ll << 32) && (int64_t)p.Sk * p.vs_s * 2 < (1ll << 32);
  if ((fa_impl() == 3 || fa_impl() == 0) && mask_kind == MIO_MASK_NONE && p.Sq > 128 && span32)
    return causal ? launch_three<true>(p, stream) : launch_three<false>(p, stream);
  const bool two = (fa_impl() == 2);
  if (mask_kind == MIO_MASK_NONE && two && p.Sq > 128)
    return causal ? launch_two<true>(p, stream) : launch_two<false>(p, stream);
  if (causal) {
    if (mask_kind == MIO_MASK_NONE) return launch_one<true, 0>(p, stream);
    if (mask_kind == MIO_MASK_KEEP_U8) return launch_one<true, 1>(p, stream);
    return launch_one<true, 2>(p, stream);
  }
  if (mask_kind == MIO_MASK_NONE) return launch_one<false, 0>(p, stream);
  if (mask_kind == MIO_MASK_KEEP_U8) return launch_one<false, 1>(p, stream);
  return launch_one<false, 2>(p, stream);
}
