// C-ABI entry points mio_gemm_bias_act / mio_fused_mlp_fwd (see include/mio_hip.h).
#include "gemm_kernel.h"

extern template int gemm_launch<__bf16>(GemmDev, int, hipStream_t);
extern template int gemm_launch<_Float16>(GemmDev, int, hipStream_t);

static int gemm_dispatch(const GemmDev& p, int act, int dtype, hipStream_t st) {
  if (dtype == MIO_BF16) return gemm_launch<__bf16>(p, act, st);
  return gemm_launch<_Float16>(p, act, st);
}

extern "C" int mio_gemm_bias_act(const void* x, const void* w, const void* bias, const void* w_gate,
                                 const void* bias_gate, const void* residual, void* y, int64_t M, int32_t N,
                                 int32_t K, int64_t ldx, int64_t ldw, int64_t ldy, int64_t ldr, int32_t act,
                                 int32_t dtype, void* stream) {
  MIO_CHECK(x && w && y, "mio_gemm_bias_act: x, w, y must be non-null");
  MIO_CHECK(M >= 0 && N > 0 && K > 0, "mio_gemm_bias_act: bad sizes");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_gemm_bias_act: dtype must be bf16 or fp16");
  MIO_CHECK(act >= MIO_ACT_NONE && act <= MIO_ACT_SWIGLU, "mio_gemm_bias_act: unknown activation");
  MIO_CHECK((act == MIO_ACT_SWIGLU) == (w_gate != nullptr), "mio_gemm_bias_act: w_gate is required iff act == SWIGLU");
  MIO_CHECK(K % 8 == 0 && N % 8 == 0, "mio_gemm_bias_act: N and K must be multiples of 8");
  MIO_CHECK(ldx % 8 == 0 && ldw % 8 == 0 && ldy % 8 == 0 && (residual == nullptr || ldr % 8 == 0),
            "mio_gemm_bias_act: row strides must be multiples of 8 elements");
  MIO_CHECK(ldx >= K && ldw >= K && ldy >= N, "mio_gemm_bias_act: row stride smaller than row length");
  MIO_CHECK(mio_aligned16(x) && mio_aligned16(w) && mio_aligned16(y) && mio_aligned16(w_gate) &&
                mio_aligned16(residual) && mio_aligned16(bias) && mio_aligned16(bias_gate),
            "mio_gemm_bias_act: pointers must be 16-byte aligned");
  if (M == 0) return 0;
  GemmDev p;
  p.x = x; p.w = w; p.wg = w_gate; p.bias = bias; p.bias_g = bias_gate; p.res = residual; p.y = y;
  p.M = M; p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.ldr = ldr; p.N = N; p.K = K;
  p.tiles_m = p.tiles_n = 0;
  p.dbg = nullptr;
  return gemm_dispatch(p, act, dtype, (hipStream_t)stream);
}

extern "C" size_t mio_fused_mlp_workspace_bytes(int64_t M, int32_t d, int32_t I, int32_t act) {
  (void)d; (void)act;
  return (size_t)M * (size_t)I * 2;
}

extern "C" int mio_fused_mlp_fwd(const void* x, const void* w1, const void* b1, const void* wg, const void* bg,
                                 const void* w2, const void* b2, const void* residual, void* y, void* workspace,
                                 int64_t M, int32_t d, int32_t I, int32_t act, int32_t dtype, void* stream) {
  MIO_CHECK(workspace != nullptr || M == 0, "mio_fused_mlp_fwd: workspace must be non-null");
  MIO_CHECK(act != MIO_ACT_NONE, "mio_fused_mlp_fwd: an activation is required");
  // stage 1: h = act(x w1^T + b1) [* silu-gate], written once in the storage dtype
  int rc = mio_gemm_bias_act(x, w1, b1, wg, bg, nullptr, workspace, M, I, d, d, d, I, 0, act, dtype, stream);
  if (rc != 0) return rc;
  // stage 2: y = h w2^T + b2 (+ residual)
  return mio_gemm_bias_act(workspace, w2, b2, nullptr, nullptr, residual, y, M, d, I, I, I, d, d, MIO_ACT_NONE,
                           dtype, stream);
}
