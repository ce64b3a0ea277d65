// One translation unit per (dtype, padded head dim): compiled with -DFA_TYPE_ID={0,1} -DFA_D={64,96,128}.
// The product library reads no environment variable and keeps no unsynchronised mutable state; the A/B switches and the
// in-kernel stamp instantiations exist only in the diagnostic build (make dbg: -DMIO_DIAG -> libmio_hip_dbg.so).
#include <cstdlib>
#include <mutex>

#include "fa3_fwd2_kernel.h"
#include "fa3_fwd3_kernel.h"
#if FA_D == 64
#include "fa3_fwd4_kernel.h"
#include "fa3_fwd5_kernel.h"
#endif

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
  if (smem > 48 * 1024) {  // > 64 KiB dynamic LDS (D = 128) needs the opt-in once per kernel
    static std::once_flag once;
    static hipError_t ea = hipSuccess;
    std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd launch: ") + hipGetErrorString(e));
  return 0;
}

#ifdef MIO_DIAG
// second structure (one wave per SIMD, 64 query rows per wave): no user mask, D % 32 == 0 instantiations
template <bool CAUSAL>
static int launch_two(FaDev p, hipStream_t stream) {
  p.nqblk = (p.Sq + FA2_BM - 1) / FA2_BM;
  // causal: a workgroup takes query blocks i and nqblk-1-i back to back -> every workgroup does the same work
  p.qgrid = CAUSAL ? (p.nqblk + 1) / 2 : p.nqblk;
  static const int order = [] { const char* e = std::getenv("MIO_FA_ORDER"); return e ? std::atoi(e) : 0; }();
  p.xcd_remap |= order;  // tuning aid: 2 = light-first, 4 = block-major
  const int grid = p.qgrid * p.B * p.H;
  const size_t smem = FaSmem<FA_D>::TOTAL;
  auto kern = fa3_fwd2_kernel<FaT, FA_D, CAUSAL>;
  if (smem > 48 * 1024) {
    static std::once_flag once;
    static hipError_t ea = hipSuccess;
    std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd2: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd2 launch: ") + hipGetErrorString(e));
  return 0;
}
#endif

// third structure (software-pipelined across KV tiles): no user mask
template <bool CAUSAL, bool KPRE = false>
static int launch_three(FaDev p, hipStream_t stream) {
  p.nqblk = (p.Sq + FA3_BM - 1) / FA3_BM;
  p.qgrid = CAUSAL ? (p.nqblk + 1) / 2 : p.nqblk;
  const int grid = p.qgrid * p.B * p.H;
  const size_t smem = FA3_STAGES * FaSmem<FA_D>::STAGE;
  auto kern = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 0, KPRE>;
#ifdef MIO_DIAG
#if FA_D == 64 && FA_TYPE_ID == 0
  if constexpr (CAUSAL && !KPRE) {  // timing-only ablations (tools/fa_ablate.py): mio_dbg_set(0, bits)
    void (*ka)(const FaDev) = nullptr;
    switch (mio_dbg_get(0)) {
      case 1: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 1>; break;
      case 2: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 2>; break;
      case 3: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 3>; break;
      case 4: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 4>; break;
      case 8: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 8>; break;
      case 16: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 16>; break;
      case 32: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 32>; break;
      case 48: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 48>; break;
      case 57: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 57>; break;
      case 59: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 59>; break;
      case 63: ka = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, false, 63>; break;
      default: break;
    }
    if (ka != nullptr) {
      hipError_t ed = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      if (ed != hipSuccess) return mio_fail(std::string("fa3_fwd3 (ablation): hipFuncSetAttribute: ") + hipGetErrorString(ed));
      hipLaunchKernelGGL(ka, dim3(grid), dim3(256), smem, stream, p);
      return 0;
    }
  }
#endif
  static const char* dbg_ptr = std::getenv("MIO_FA_DBG_PTR");  // in-kernel phase stamps (tools/fa_stamps.py)
  if (dbg_ptr != nullptr && !KPRE) {
    auto kd = fa3_fwd3_kernel<FaT, FA_D, CAUSAL, true>;
    p.mask = (const void*)std::strtoull(dbg_ptr, nullptr, 0);
    hipError_t ed = hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (ed != hipSuccess) return mio_fail(std::string("fa3_fwd3 (stamps): hipFuncSetAttribute: ") + hipGetErrorString(ed));
    hipLaunchKernelGGL(kd, dim3(grid), dim3(256), smem, stream, p);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) return mio_fail(std::string("fa3_fwd3 (stamps) launch: ") + hipGetErrorString(e2));
    return 0;
  }
#endif
  {
    static std::once_flag once;
    static hipError_t ea = hipSuccess;
    std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); });
    if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd3: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd3 launch: ") + hipGetErrorString(e));
  return 0;
}

// fourth structure (two waves per SIMD, 8 waves x 32 query rows, 32x32x16 tiles): head dim <= 64, no user mask, plain output.
// Diagnostic library only since round 3: fa3_fwd5_kernel takes every launch it used to (with and without pre-scaled K).
#if FA_D == 64 && defined(MIO_DIAG)
template <bool CAUSAL, bool KPRE = false>
static int launch_four(FaDev p, hipStream_t stream) {
  p.nqblk = (p.Sq + FA4_BM - 1) / FA4_BM;
  p.qgrid = CAUSAL ? (p.nqblk + 1) / 2 : p.nqblk;
  const int grid = p.qgrid * p.B * p.H;
  void (*kern)(const FaDev) = fa3_fwd4_kernel<FaT, CAUSAL, 0, KPRE>;
#if defined(MIO_DIAG) && FA_TYPE_ID == 0
  if constexpr (CAUSAL && !KPRE) {  // timing-only ablations (tools/fa4_ablate.py)
    void (*ka)(const FaDev) = nullptr;
    switch (mio_dbg_get(0)) {
      case 1: ka = fa3_fwd4_kernel<FaT, CAUSAL, 1>; break;
      case 2: ka = fa3_fwd4_kernel<FaT, CAUSAL, 2>; break;
      case 4: ka = fa3_fwd4_kernel<FaT, CAUSAL, 4>; break;
      case 8: ka = fa3_fwd4_kernel<FaT, CAUSAL, 8>; break;
      case 12: ka = fa3_fwd4_kernel<FaT, CAUSAL, 12>; break;
      case 16: ka = fa3_fwd4_kernel<FaT, CAUSAL, 16>; break;
      case 32: ka = fa3_fwd4_kernel<FaT, CAUSAL, 32>; break;
      case 35: ka = fa3_fwd4_kernel<FaT, CAUSAL, 35>; break;
      case 47: ka = fa3_fwd4_kernel<FaT, CAUSAL, 47>; break;
      case 64: ka = fa3_fwd4_kernel<FaT, CAUSAL, 64>; break;
      default: break;
    }
    if (ka != nullptr) {
      hipError_t ed = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, FA4_SMEM);
      if (ed != hipSuccess) return mio_fail(std::string("fa3_fwd4 (ablation): hipFuncSetAttribute: ") + hipGetErrorString(ed));
      hipLaunchKernelGGL(ka, dim3(grid), dim3(512), FA4_SMEM, stream, p);
      return 0;
    }
  }
#endif
  static std::once_flag once;
  static hipError_t ea = hipSuccess;
  std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FA4_SMEM); });
  if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd4: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), FA4_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd4 launch: ") + hipGetErrorString(e));
  return 0;
}
#endif

// fifth structure (fa3_fwd4's skeleton on 16x16x32 MFMA tiles): k_prescaled launches, head dim <= 64
#if FA_D == 64
template <bool CAUSAL, bool CARRY = false, bool OBLK = false, bool KPRE = true>
static int launch_five(FaDev p, hipStream_t stream) {
  p.nqblk = (p.Sq + FA4_BM - 1) / FA4_BM;
  p.qgrid = CAUSAL ? (p.nqblk + 1) / 2 : p.nqblk;
  const int grid = p.qgrid * p.B * p.H;
  void (*kern)(const FaDev) = fa3_fwd5_kernel<FaT, CAUSAL, false, 0, CARRY, OBLK, KPRE>;
#if defined(MIO_DIAG) && FA_TYPE_ID == 0
  if constexpr (!CARRY && !OBLK && KPRE) {
  p.xcd_remap |= (mio_dbg_get(3) & 7) << 4;  // wave-priority probe (tools/fa5_ablate.py)
  static const char* dbg_ptr = std::getenv("MIO_FA_DBG_PTR");  // in-kernel phase stamps (tools/fa5_stamps.py)
  if (dbg_ptr != nullptr) {
    auto kd = fa3_fwd5_kernel<FaT, CAUSAL, true>;
    p.mask = (const void*)std::strtoull(dbg_ptr, nullptr, 0);
    hipError_t ed = hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, FA5_SMEM);
    if (ed != hipSuccess) return mio_fail(std::string("fa3_fwd5 (stamps): hipFuncSetAttribute: ") + hipGetErrorString(ed));
    hipLaunchKernelGGL(kd, dim3(grid), dim3(512), FA5_SMEM, stream, p);
    return 0;
  }
  if constexpr (CAUSAL) {  // timing-only ablations (tools/fa5_ablate.py): mio_dbg_set(0, bits)
    void (*ka)(const FaDev) = nullptr;
    switch (mio_dbg_get(0)) {
      case 1: ka = fa3_fwd5_kernel<FaT, CAUSAL, false, 1>; break;
      case 2: ka = fa3_fwd5_kernel<FaT, CAUSAL, false, 2>; break;
      case 4: ka = fa3_fwd5_kernel<FaT, CAUSAL, false, 4>; break;
      case 8: ka = fa3_fwd5_kernel<FaT, CAUSAL, false, 8>; break;
      case 15: ka = fa3_fwd5_kernel<FaT, CAUSAL, false, 15>; break;
      default: break;
    }
    if (ka != nullptr) {
      hipError_t ed = hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, FA5_SMEM);
      if (ed != hipSuccess) return mio_fail(std::string("fa3_fwd5 (ablation): hipFuncSetAttribute: ") + hipGetErrorString(ed));
      hipLaunchKernelGGL(ka, dim3(grid), dim3(512), FA5_SMEM, stream, p);
      return 0;
    }
  }
  }
#endif
  static std::once_flag once;
  static hipError_t ea = hipSuccess;
  std::call_once(once, [&] { ea = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, FA5_SMEM); });
  if (ea != hipSuccess) return mio_fail(std::string("fa3_fwd5: hipFuncSetAttribute: ") + hipGetErrorString(ea));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), FA5_SMEM, stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("fa3_fwd5 launch: ") + hipGetErrorString(e));
  return 0;
}
#endif

#ifdef MIO_DIAG
static int fa_impl() {  // MIO_FA_IMPL=1 / 2 / 3 force one structure for A/B runs (0 = the rule in fa3_launch)
  static const int v = [] {
    const char* e = std::getenv("MIO_FA_IMPL");
    return e ? std::atoi(e) : 0;
  }();
  return v;
}
#else
static constexpr int fa_impl() { return 0; }
#endif

template <>
int fa3_launch<FaT, FA_D>(const FaDev& p, int causal, int mask_kind, hipStream_t stream) {
  // Structure choice (measured on MI355X, B8 S4096, random data; MIO_FA_IMPL=1 / 2 / 3 forces one for A/B runs):
  //   no user mask, Sq > 128: the software-pipelined kernel at every head dim --
  //     D64 causal 0.359 ms (766 TFLOP/s) vs 0.442 two-waves-per-SIMD / 0.52 sequential one-wave; non-causal 0.626 vs 0.79;
  //     D128 causal 0.284 ms (967 TFLOP/s) vs 0.361 sequential one-wave; D80 non-causal 0.828 ms (830) vs 1.158;
  //   user masks and Sq <= 128: the two-waves-per-SIMD kernel.
  // (the pipelined kernel addresses K / V tiles with 32-bit byte offsets from the (batch, head) base)
  const bool span32 = (int64_t)p.Sk * p.ks_s * 2 < (1ll << 32) && (int64_t)p.Sk * p.vs_s * 2 < (1ll << 32);
#if FA_D == 64
  // head dim <= 64, causal, plain output: the two-waves-per-SIMD kernel (0.325 vs 0.336 ms at B8 S4096 H16 interleaved on
  // one box; non-causal it is 1 % behind fa3_fwd3 and stays there).  Diagnostic build: MIO_FA_IMPL=3 / mio_dbg_set(1, 3)
  // keep fa3_fwd3, = 4 force fa3_fwd4 for non-causal launches too.
  {
    const bool plain = mask_kind == MIO_MASK_NONE && p.Sq > 128 && span32 && p.o != nullptr && p.o_acc == nullptr && !p.carry_in;
    // k_prescaled (mio_fa3_fwd has checked mio_fa3_k_prescaled_ok), head dim <= 64: fa3_fwd5 (16x16x32 MFMA tiles).  Same
    // box, interleaved, B8 S4096 H16 bf16: causal 0.3047 ms vs 0.3305 fa3_fwd4 KPRE / 0.3401 fa3_fwd3 KPRE / 0.3570 fa3_fwd3;
    // non-causal 0.5494 vs 0.6162 / 0.5862 / 0.6273.  Diagnostic build: mio_dbg_set(1, 3 | 4) select the other KPRE forms.
    // ... and its ring form: (o_acc, lse) carried in / written back, bf16 output optional
    if (p.k_prescaled && !plain && mask_kind == MIO_MASK_NONE && p.Sq > 128 && span32 && p.o_acc != nullptr)
      return causal ? launch_five<true, true>(p, stream) : launch_five<false, true>(p, stream);
    if (p.k_prescaled && plain) {
      int which = 5;
#ifdef MIO_DIAG
      if (mio_dbg_get(1) == 3 || mio_dbg_get(1) == 4) which = mio_dbg_get(1);
#endif
      if (which == 5 && p.o_blk) return causal ? launch_five<true, false, true>(p, stream) : launch_five<false, false, true>(p, stream);
      if (which == 5) return causal ? launch_five<true>(p, stream) : launch_five<false>(p, stream);
#ifdef MIO_DIAG
      if (which == 4) return causal ? launch_four<true, true>(p, stream) : launch_four<false, true>(p, stream);
#endif
    }
    // plain K (the functional entry point and every module that does not own the K projection's epilogue): the same
    // structure with the scale applied in fp32 on the way into exp2
    if (!p.k_prescaled && plain && fa_impl() == 0
#ifdef MIO_DIAG
        && mio_dbg_get(1) == 0
#endif
    )
      return causal ? launch_five<true, false, false, false>(p, stream) : launch_five<false, false, false, false>(p, stream);
#ifdef MIO_DIAG
    bool four = false;
    if (plain && (fa_impl() == 4 || mio_dbg_get(1) == 4)) four = true;
    if (four) return causal ? launch_four<true>(p, stream) : launch_four<false>(p, stream);
#endif
  }
#endif
  if (p.k_prescaled) {  // head dim 65 .. 96 (and the diagnostic A/B at 64): fa3_fwd3's KPRE form.  Not at 128: the two
                         // reference tuples (32 VGPRs) do not fit beside the score / P / fragment registers there -- hipcc
                         // parks values in accumulator registers the kernel owns (tools/check_agpr.py catches it)
#if FA_D < 128
    if (mask_kind == MIO_MASK_NONE && p.Sq > 128 && span32 && p.o != nullptr && p.o_acc == nullptr && !p.carry_in)
      return causal ? launch_three<true, true>(p, stream) : launch_three<false, true>(p, stream);
#endif
    return mio_fail("fa3_fwd: k_prescaled launch outside the kernels that support it");
  }
  if ((fa_impl() == 3 || fa_impl() == 0) && mask_kind == MIO_MASK_NONE && p.Sq > 128 && span32)
    return causal ? launch_three<true>(p, stream) : launch_three<false>(p, stream);
#ifdef MIO_DIAG
  if (mask_kind == MIO_MASK_NONE && fa_impl() == 2 && p.Sq > 128)
    return causal ? launch_two<true>(p, stream) : launch_two<false>(p, stream);
#endif
  if (causal) {
    if (mask_kind == MIO_MASK_NONE) return launch_one<true, 0>(p, stream);
    if (mask_kind == MIO_MASK_KEEP_U8) return launch_one<true, 1>(p, stream);
    return launch_one<true, 2>(p, stream);
  }
  if (mask_kind == MIO_MASK_NONE) return launch_one<false, 0>(p, stream);
  if (mask_kind == MIO_MASK_KEEP_U8) return launch_one<false, 1>(p, stream);
  return launch_one<false, 2>(p, stream);
}
