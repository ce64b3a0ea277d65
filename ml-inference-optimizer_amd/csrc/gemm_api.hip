// C-ABI entry points mio_gemm_bias_act / mio_fused_mlp_fwd (see include/mio_hip.h).
#include <cstdlib>
#include <string>

#include "gemm_kernel.h"

extern template int gemm_launch<__bf16>(GemmDev, int, hipStream_t);
extern template int gemm_launch<_Float16>(GemmDev, int, hipStream_t);

// Diagnostic build only: MIO_GEMM_IMPL=v1|4wp|4w16 forces one pipeline for A/B comparisons (read once).  The product
// library always takes the default dispatch (0) and reads no environment variable.
int mio_gemm_impl() {
#ifdef MIO_DIAG
  if (mio_dbg_get(4) != 0) return mio_dbg_get(4);  // run-time override for same-process A/B (tools/gemm8_ab.py)
  static const int v = [] {
    const char* e = std::getenv("MIO_GEMM_IMPL");
    if (e == nullptr) return 0;
    const std::string s(e);
    if (s == "v1") return 1;
    if (s == "4wp") return 5;
    if (s == "4w16") return 6;
    if (s == "8w1") return 9;   // gemm8w_kernel, one workgroup per tile
    return 0;
  }();
  return v;
#else
  return 0;
#endif
}

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
  gemm_dev_defaults(p);
  p.x = x; p.w = w; p.wg = w_gate; p.bias = bias; p.bias_g = bias_gate; p.res = residual; p.y = y;
  p.M = M; p.ldx = ldx; p.ldw = ldw; p.ldy = ldy; p.ldr = ldr; p.N = N; p.K = K;
  p.tiles_m = p.tiles_n = 0;
  p.x_blk = p.y_blk = 0;
  p.w_blk = 0;
  p.dbg = nullptr;
  p.cs_lo = p.cs_hi = 0; p.cs_val = 1.f; p.group_m = 0;
  return gemm_dispatch(p, act, dtype, (hipStream_t)stream);
}

// Both GEMMs of the MLP take a 256x256-tile 16x16x32 kernel (gemm_inst.hip launch_act) and stage 1 the persistent one:
// then the intermediate can use the blocked layout (GemmDev::x_blk / y_blk).
static bool mlp_blocked_ok(int64_t M, int32_t d, int32_t I, int32_t act, bool residual) {
  (void)residual;
  if (mio_gemm_impl() != 0 && mio_gemm_impl() < 5) return false;
  const int64_t tm = (M + 255) / 256;
  // (SwiGLU: stage 1 computes 256 x 128 output tiles from 256 interleaved gate / up weight rows, gemm8w_kernel.h)
  const bool big1 = tm * ((I + 255) / 256) >= 256, big2 = tm * ((d + 255) / 256) >= 256;
  const bool fits = (int64_t)d * 512 < 0x7fffffff && (int64_t)I * 512 < 0x7fffffff;
  return big1 && big2 && fits && d % 64 == 0 && d >= 256 && I % 256 == 0 && d % 8 == 0;
}

extern "C" size_t mio_fused_mlp_workspace_bytes(int64_t M, int32_t d, int32_t I, int32_t act) {
  (void)d; (void)act;
  const int64_t mp = (M + 255) / 256 * 256;  // whole 256-row blocks (blocked intermediate layout)
  return (size_t)mp * (size_t)I * 2;
}

static int fused_mlp_impl(const void* x, const void* w1, const void* b1, const void* wg, const void* bg, const void* w2,
                          const void* b2, const void* residual, void* y, void* workspace, int64_t M, int32_t d, int32_t I,
                          int32_t act, int32_t dtype, void* stream, int wblk, int xblk = 0) {
  MIO_CHECK(workspace != nullptr || M == 0, "mio_fused_mlp_fwd: workspace must be non-null");
  MIO_CHECK(act != MIO_ACT_NONE, "mio_fused_mlp_fwd: an activation is required");
  if (M > 0 && mlp_blocked_ok(M, d, I, act, residual != nullptr) && (act != MIO_ACT_SWIGLU || wblk)) {
    MIO_CHECK(x && w1 && w2 && y, "mio_fused_mlp_fwd: x, w1, w2, y must be non-null");
    MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_fused_mlp_fwd: dtype must be bf16 or fp16");
    MIO_CHECK(mio_aligned16(x) && mio_aligned16(w1) && mio_aligned16(w2) && mio_aligned16(y) && mio_aligned16(b1) &&
                  mio_aligned16(b2) && mio_aligned16(residual) && mio_aligned16(workspace),
              "mio_fused_mlp_fwd: pointers must be 16-byte aligned");
    MIO_CHECK(act != MIO_ACT_SWIGLU || wblk, "mio_fused_mlp_fwd: the gated 256-tile kernel takes the interleaved blocked weight "
                                            "(mio_weight_block_glu) through mio_fused_mlp_glu_fwd_bw");
    GemmDev p;
    gemm_dev_defaults(p);
    p.x = x; p.w = w1; p.wg = nullptr; p.bias = b1; p.bias_g = bg; p.res = nullptr; p.y = workspace;
    p.M = M; p.ldx = d; p.ldw = d; p.ldy = I; p.ldr = 0; p.N = I; p.K = d;
    p.tiles_m = p.tiles_n = 0;
    p.x_blk = xblk; p.y_blk = 1; p.w_blk = (act == MIO_ACT_SWIGLU) ? 2 : wblk;  // 2: gate / up rows interleaved per wave
    p.dbg = nullptr;
    p.cs_lo = p.cs_hi = 0; p.cs_val = 1.f; p.group_m = 0;
    int rc = gemm_dispatch(p, act, dtype, (hipStream_t)stream);
    if (rc != 0) return rc;
    p.x = workspace; p.w = w2; p.bias = b2; p.bias_g = nullptr; p.res = residual; p.y = y;
    p.ldx = I; p.ldw = I; p.ldy = d; p.ldr = d; p.N = d; p.K = I;
    p.tiles_m = p.tiles_n = 0;
    p.x_blk = 1; p.y_blk = 0; p.w_blk = wblk;
    return gemm_dispatch(p, MIO_ACT_NONE, dtype, (hipStream_t)stream);
  }
  MIO_CHECK(!wblk && !xblk, "mio_fused_mlp_fwd_bw: this shape does not take the blocked-weight kernels "
                   "(mio_fused_mlp_blocked_weight_ok == 0); pass the plain weights to mio_fused_mlp_fwd");
  // stage 1: h = act(x w1^T + b1) [* silu-gate], written once in the storage dtype
  int rc = mio_gemm_bias_act(x, w1, b1, wg, bg, nullptr, workspace, M, I, d, d, d, I, 0, act, dtype, stream);
  if (rc != 0) return rc;
  // stage 2: y = h w2^T + b2 (+ residual)
  return mio_gemm_bias_act(workspace, w2, b2, nullptr, nullptr, residual, y, M, d, I, I, I, d, d, MIO_ACT_NONE,
                           dtype, stream);
}

extern "C" int mio_fused_mlp_fwd(const void* x, const void* w1, const void* b1, const void* wg, const void* bg,
                                 const void* w2, const void* b2, const void* residual, void* y, void* workspace,
                                 int64_t M, int32_t d, int32_t I, int32_t act, int32_t dtype, void* stream) {
  return fused_mlp_impl(x, w1, b1, wg, bg, w2, b2, residual, y, workspace, M, d, I, act, dtype, stream, 0);
}

// ---- blocked weights -----------------------------------------------------------------------------------------------
static bool gemm_blocked_w_ok(int64_t M, int32_t N, int32_t K, int32_t act) {
  if (act == MIO_ACT_SWIGLU || (mio_gemm_impl() != 0 && mio_gemm_impl() < 5)) return false;
  const bool big = ((M + 255) / 256) * (int64_t)((N + 255) / 256) >= 256;
  return big && K % 32 == 0 && K >= 128 && (int64_t)K * 512 < 0x7fffffff && (int64_t)N * 512 < 0x7fffffff;
}

extern "C" size_t mio_weight_blocked_bytes(int32_t N, int32_t K) {
  return (size_t)((N + 255) / 256 * 256) * (size_t)K * 2;
}

// one 16-byte chunk per thread: destination unit u = ((tn * nk + kt) * 256 + row) * 4 + chunk
__global__ void weight_block_kernel(const uint16_t* __restrict__ w, int64_t ldw, uint16_t* __restrict__ wb, int N, int K,
                                    int64_t units) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int nk = K / 32;
  const int chunk = (int)(u & 3), row = (int)((u >> 2) & 255);
  const int64_t blk = u >> 10;
  const int kt = (int)(blk % nk), tn = (int)(blk / nk);
  const int n = tn * 256 + row;
  u32x4_t v = {0u, 0u, 0u, 0u};
  if (n < N) v = *(const u32x4_t*)(w + (int64_t)n * ldw + kt * 32 + chunk * 8);
  *(u32x4_t*)(wb + u * 8) = v;
}

extern "C" int mio_weight_block(const void* w, int64_t ldw, void* wb, int32_t N, int32_t K, int32_t dtype, void* stream) {
  MIO_CHECK(w && wb, "mio_weight_block: w and wb must be non-null");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_weight_block: dtype must be bf16 or fp16");
  MIO_CHECK(N > 0 && K > 0 && K % 32 == 0 && ldw >= K && ldw % 8 == 0, "mio_weight_block: need K % 32 == 0, ldw % 8 == 0");
  MIO_CHECK(mio_aligned16(w) && mio_aligned16(wb), "mio_weight_block: pointers must be 16-byte aligned");
  const int64_t units = (int64_t)((N + 255) / 256) * (K / 32) * 1024;
  hipLaunchKernelGGL(weight_block_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)w, ldw, (uint16_t*)wb, N, K, units);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("mio_weight_block launch: ") + hipGetErrorString(e));
  return 0;
}

extern "C" int32_t mio_gemm_blocked_weight_ok(int64_t M, int32_t N, int32_t K, int32_t act) {
  return gemm_blocked_w_ok(M, N, K, act) ? 1 : 0;
}

extern "C" int mio_gemm_bias_act_bw(const void* x, const void* wb, const void* bias, const void* residual, void* y,
                                    int64_t M, int32_t N, int32_t K, int64_t ldx, int64_t ldy, int64_t ldr, int32_t act,
                                    int32_t dtype, int32_t x_blocked, void* stream) {
  MIO_CHECK(x && wb && y, "mio_gemm_bias_act_bw: x, wb, y must be non-null");
  MIO_CHECK(M >= 0 && N > 0 && K > 0, "mio_gemm_bias_act_bw: bad sizes");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_gemm_bias_act_bw: dtype must be bf16 or fp16");
  MIO_CHECK(act >= MIO_ACT_NONE && act < MIO_ACT_SWIGLU, "mio_gemm_bias_act_bw: unknown / unsupported activation");
  if (x_blocked) ldx = K;  // blocked x: ceil(M / 256) * 256 x K elements, no row stride
  MIO_CHECK(N % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (residual == nullptr || ldr % 8 == 0) && ldx >= K && ldy >= N,
            "mio_gemm_bias_act_bw: bad strides");
  MIO_CHECK(mio_aligned16(x) && mio_aligned16(wb) && mio_aligned16(y) && mio_aligned16(residual) && mio_aligned16(bias),
            "mio_gemm_bias_act_bw: pointers must be 16-byte aligned");
  MIO_CHECK(ldx * 512 < (int64_t)0x7fffffff && ldy * 512 < (int64_t)0x7fffffff && (residual == nullptr || ldr * 512 < (int64_t)0x7fffffff),
            "mio_gemm_bias_act_bw: row stride too large for the blocked-weight kernels");
  MIO_CHECK(gemm_blocked_w_ok(M, N, K, act), "mio_gemm_bias_act_bw: this shape does not take the blocked-weight kernels "
                                             "(mio_gemm_blocked_weight_ok == 0); use mio_gemm_bias_act with the plain weight");
  GemmDev p;
  gemm_dev_defaults(p);
  p.x = x; p.w = wb; p.wg = nullptr; p.bias = bias; p.bias_g = nullptr; p.res = residual; p.y = y;
  p.M = M; p.ldx = ldx; p.ldw = K; p.ldy = ldy; p.ldr = ldr; p.N = N; p.K = K;
  p.tiles_m = p.tiles_n = 0;
  p.x_blk = x_blocked ? 1 : 0; p.y_blk = 0; p.w_blk = 1;
  p.dbg = nullptr;
  p.cs_lo = p.cs_hi = 0; p.cs_val = 1.f; p.group_m = 0;
  return gemm_dispatch(p, act, dtype, (hipStream_t)stream);
}

// column scale: the launch must end in the persistent kernel (gemm_inst.hip launch_act): blocked weight shape, no residual,
// K >= 256, K % 64 == 0
extern "C" int32_t mio_gemm_col_scale_ok(int64_t M, int32_t N, int32_t K, int32_t act) {
  return (gemm_blocked_w_ok(M, N, K, act) && (mio_gemm_impl() == 0 || mio_gemm_impl() >= 5) && K >= 256 && K % 64 == 0 && N % 8 == 0) ? 1 : 0;
}

extern "C" int mio_gemm_bias_act_bw_cs(const void* x, const void* wb, const void* bias, void* y, int64_t M, int32_t N,
                                       int32_t K, int64_t ldx, int64_t ldy, int32_t act, int32_t dtype, int32_t x_blocked,
                                       int32_t cs_lo, int32_t cs_hi, float cs_val, void* stream) {
  MIO_CHECK(x && wb && y, "mio_gemm_bias_act_bw_cs: x, wb, y must be non-null");
  MIO_CHECK(M >= 0 && N > 0 && K > 0, "mio_gemm_bias_act_bw_cs: bad sizes");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_gemm_bias_act_bw_cs: dtype must be bf16 or fp16");
  MIO_CHECK(act >= MIO_ACT_NONE && act < MIO_ACT_SWIGLU, "mio_gemm_bias_act_bw_cs: unknown / unsupported activation");
  if (x_blocked) ldx = K;
  MIO_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= K && ldy >= N, "mio_gemm_bias_act_bw_cs: bad strides");
  MIO_CHECK(mio_aligned16(x) && mio_aligned16(wb) && mio_aligned16(y) && mio_aligned16(bias),
            "mio_gemm_bias_act_bw_cs: pointers must be 16-byte aligned");
  MIO_CHECK(ldx * 512 < (int64_t)0x7fffffff && ldy * 512 < (int64_t)0x7fffffff, "mio_gemm_bias_act_bw_cs: row stride too large");
  MIO_CHECK(mio_gemm_col_scale_ok(M, N, K, act), "mio_gemm_bias_act_bw_cs: this shape does not run the persistent kernel "
                                                 "(mio_gemm_col_scale_ok == 0)");
  MIO_CHECK(cs_lo >= 0 && cs_hi <= N && cs_lo % 128 == 0 && cs_hi % 128 == 0 && cs_lo <= cs_hi,
            "mio_gemm_bias_act_bw_cs: [cs_lo, cs_hi) must be multiples of 128 inside [0, N]");
  GemmDev p;
  gemm_dev_defaults(p);
  p.x = x; p.w = wb; p.wg = nullptr; p.bias = bias; p.bias_g = nullptr; p.res = nullptr; p.y = y;
  p.M = M; p.ldx = ldx; p.ldw = K; p.ldy = ldy; p.ldr = 0; p.N = N; p.K = K;
  p.tiles_m = p.tiles_n = 0;
  p.x_blk = x_blocked ? 1 : 0; p.y_blk = 0; p.w_blk = 1;
  p.dbg = nullptr;
  p.cs_lo = cs_lo; p.cs_hi = cs_hi; p.cs_val = cs_val; p.group_m = 0;
  return gemm_dispatch(p, act, dtype, (hipStream_t)stream);
}

extern "C" int32_t mio_fused_mlp_blocked_weight_ok(int64_t M, int32_t d, int32_t I, int32_t act) {
  return (M > 0 && mlp_blocked_ok(M, d, I, act, false)) ? 1 : 0;
}

extern "C" int mio_fused_mlp_fwd_bw(const void* x, const void* w1b, const void* b1, const void* w2b, const void* b2,
                                    const void* residual, void* y, void* workspace, int64_t M, int32_t d, int32_t I,
                                    int32_t act, int32_t dtype, int32_t x_blocked, void* stream) {
  return fused_mlp_impl(x, w1b, b1, nullptr, nullptr, w2b, b2, residual, y, workspace, M, d, I, act, dtype, stream, 1,
                        x_blocked ? 1 : 0);
}

// ---- SwiGLU on the 256-tile kernel: gate / up weights interleaved per wave in one blocked weight ----------------------
extern "C" size_t mio_weight_blocked_glu_bytes(int32_t I, int32_t K) {
  return (size_t)((I + 127) / 128 * 256) * (size_t)K * 2;
}

// one 16-byte chunk per thread: destination unit u = ((tn * nk + kt) * 256 + row) * 4 + chunk, row = wn * 64 + h * 32 + j
// <- row tn * 128 + wn * 32 + j of the gate (h = 0) or up (h = 1) weight
__global__ void weight_block_glu_kernel(const uint16_t* __restrict__ wg, const uint16_t* __restrict__ wu, int64_t ldw,
                                        uint16_t* __restrict__ wb, int I, int K, int64_t units) {
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int nk = K / 32;
  const int chunk = (int)(u & 3), row = (int)((u >> 2) & 255);
  const int64_t blk = u >> 10;
  const int kt = (int)(blk % nk), tn = (int)(blk / nk);
  const int n = tn * 128 + (row >> 6) * 32 + (row & 31);
  const uint16_t* src = ((row >> 5) & 1) ? wu : wg;
  u32x4_t v = {0u, 0u, 0u, 0u};
  if (n < I) v = *(const u32x4_t*)(src + (int64_t)n * ldw + kt * 32 + chunk * 8);
  *(u32x4_t*)(wb + u * 8) = v;
}

extern "C" int mio_weight_block_glu(const void* w_gate, const void* w_up, int64_t ldw, void* wb, int32_t I, int32_t K,
                                    int32_t dtype, void* stream) {
  MIO_CHECK(w_gate && w_up && wb, "mio_weight_block_glu: w_gate, w_up and wb must be non-null");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_weight_block_glu: dtype must be bf16 or fp16");
  MIO_CHECK(I > 0 && K > 0 && K % 32 == 0 && ldw >= K && ldw % 8 == 0, "mio_weight_block_glu: need K % 32 == 0, ldw % 8 == 0");
  MIO_CHECK(mio_aligned16(w_gate) && mio_aligned16(w_up) && mio_aligned16(wb), "mio_weight_block_glu: pointers must be 16-byte aligned");
  const int64_t units = (int64_t)((I + 127) / 128) * (K / 32) * 1024;
  hipLaunchKernelGGL(weight_block_glu_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)w_gate, (const uint16_t*)w_up, ldw, (uint16_t*)wb, I, K, units);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("mio_weight_block_glu launch: ") + hipGetErrorString(e));
  return 0;
}

extern "C" int mio_fused_mlp_glu_fwd_bw(const void* x, const void* wgu_b, const void* b_up, const void* b_gate, const void* w2b,
                                        const void* b2, const void* residual, void* y, void* workspace, int64_t M, int32_t d,
                                        int32_t I, int32_t dtype, int32_t x_blocked, void* stream) {
  MIO_CHECK(mio_aligned16(b_gate), "mio_fused_mlp_glu_fwd_bw: pointers must be 16-byte aligned");
  MIO_CHECK(M == 0 || mlp_blocked_ok(M, d, I, MIO_ACT_SWIGLU, false),
            "mio_fused_mlp_glu_fwd_bw: this shape does not take the 256-tile kernels (mio_fused_mlp_blocked_weight_ok(.., SWIGLU) == 0); "
            "pass the plain weights to mio_fused_mlp_fwd");
  return fused_mlp_impl(x, wgu_b, b_up, nullptr, b_gate, w2b, b2, residual, y, workspace, M, d, I, MIO_ACT_SWIGLU, dtype, stream, 1,
                        x_blocked ? 1 : 0);
}

// ---- LayerNorm folded into the GEMMs on either side of it ------------------------------------------------
// Reference: kernels/triton/fused_layernorm_qkv.py:37-420 (LayerNorm as the prologue of the QKV projection) and
// layernorm_kernels.py:35-188 (residual add + LayerNorm in one pass).  Here neither a prologue nor a pass: the GEMM that
// WRITES the residual stream (out-proj / fc2, residual epilogue) also writes each output row's (sum, sum of squares) per
// 256-column tile, and the projection BEHIND the LayerNorm multiplies the raw stream with gamma-scaled weights and applies
// rstd in its read-out:  LN(x) W^T + b = rstd * (x - mean 1) (gamma o W)^T + b' = rstd * x W'^T + b',  W' = gamma o W with every
// row's mean over k subtracted (the centring moves from the activations to the weights), b' = b + W beta.
extern "C" size_t mio_ln_stats_bytes(int64_t M, int32_t width) {
  return (size_t)((width + 255) / 256) * (size_t)((M + 255) / 256 * 256) * 2 * sizeof(float);
}

extern "C" int32_t mio_gemm_ln_ok(int64_t M, int32_t N, int32_t K, int32_t act, int32_t fold_in, int32_t stats_out) {
  if (mio_gemm_impl() != 0) return 0;
  if (act == MIO_ACT_SWIGLU) {  // the gated stage (interleaved gate / up blocked weight, 256 x 128 output tiles): consumer form only
    const bool big = ((M + 255) / 256) * (int64_t)((N + 127) / 128) >= 256;
    return (big && !stats_out && N % 128 == 0 && K >= 128 && (int64_t)K * 512 < 0x7fffffff && (int64_t)N * 512 < 0x7fffffff &&
            (!fold_in || K % 256 == 0) && K % 32 == 0) ? 1 : 0;
  }
  if (!gemm_blocked_w_ok(M, N, K, act) || N % 32 != 0) return 0;
  if (fold_in && (K % 256 != 0 || (act != MIO_ACT_NONE && act != MIO_ACT_GELU_TANH))) return 0;
  if (stats_out && (N % 256 != 0 || act != MIO_ACT_NONE || fold_in)) return 0;
  return 1;
}

// one workgroup per weight row: w_scaled[n][k] = T(w[n][k] * gamma[k] - mean_k(w[n][.] * gamma[.])) (the row mean is taken over
// the unrounded products), bias_out[n] = T(bias[n] + sum_k w[n][k] * beta[k]).
// The consumer's read-out relies on sum_k w_scaled[n][k] = 0 (that is what removes the activations' mean); after rounding to 16
// bits the sum is off by ~sqrt(K / 12) ulp, and the product picks up mean(x) times that.  So the rounding DIRECTION of a few
// elements is flipped (those whose exact value sits closest to the midpoint of its two neighbours: the flip leaves their own
// error almost unchanged, weighed against how much of the sum it removes) until no flip brings the row sum closer to zero:
// <= 64 greedy steps per row, one-time weight preparation.
template <typename T, int EPT>  // EPT: elements per thread (K <= 256 * EPT)
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const T* __restrict__ w, int64_t ldw, const T* __restrict__ gamma,
                                                             const T* __restrict__ beta, const T* __restrict__ bias,
                                                             T* __restrict__ ws, T* __restrict__ bias_out, int K) {
  __shared__ float s_c[256], s_b[256];
  __shared__ int s_i[256];
  const int n = blockIdx.x, t = threadIdx.x;
  float c = 0.f, bb = 0.f;
  for (int k = t; k < K; k += 256) {
    const float wv = (float)w[(int64_t)n * ldw + k];
    c += wv * (float)gamma[k];
    if (beta != nullptr) bb += wv * (float)beta[k];
  }
  s_c[t] = c;
  s_b[t] = bb;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      s_c[t] += s_c[t + s];
      s_b[t] += s_b[t + s];
    }
    __syncthreads();
  }
  const float mean = s_c[0] / (float)K;
  if (t == 0) bias_out[n] = (T)((bias != nullptr ? (float)bias[n] : 0.f) + s_b[0]);
  __syncthreads();
  // rounded values (as 16-bit patterns) and their exact counterparts
  uint16_t bits[EPT];
  float exact[EPT];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int k = t + 256 * e;
    bits[e] = 0;
    exact[e] = 0.f;
    if (k < K) {
      exact[e] = (float)w[(int64_t)n * ldw + k] * (float)gamma[k] - mean;
      const T r = (T)exact[e];
      bits[e] = __builtin_bit_cast(uint16_t, r);
      sum += (float)r;
    }
  }
  auto val = [](uint16_t b) { return (float)__builtin_bit_cast(T, b); };
  for (int it = 0; it < 64; ++it) {
    s_c[t] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) s_c[t] += s_c[t + s];
      __syncthreads();
    }
    const float eps = s_c[0];  // sum of the rounded row (the exact row sums to zero)
    __syncthreads();
    // this thread's best flip: moves the sum towards zero without overshooting past -eps, smallest growth of its own error
    float best = 3.0e38f;
    int best_e = -1;
    uint16_t best_bits = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int k = t + 256 * e;
      if (k >= K || (bits[e] & 0x7fff) == 0) continue;
      const float cur = val(bits[e]);
      // the neighbour on the other side of the exact value
      const bool up = cur < exact[e];  // rounded down -> candidate is the next value up
      const bool neg = (bits[e] & 0x8000) != 0;
      const uint16_t nb = (uint16_t)((up != neg) ? bits[e] + 1 : bits[e] - 1);
      const float nv = val(nb);
      const float delta = nv - cur;
      if (!(delta * eps < 0.f) || fabsf(eps + delta) >= fabsf(eps)) continue;
      // what the flip takes off |sum| minus what it adds to the element's own error (lower = better; the reduction of s_b is a min)
      const float cost = fmaxf(fabsf(nv - exact[e]) - fabsf(cur - exact[e]), 0.f) - (fabsf(eps) - fabsf(eps + delta));
      if (cost < best) {
        best = cost;
        best_e = e;
        best_bits = nb;
      }
    }
    s_b[t] = best;
    s_i[t] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s && s_b[t + s] < s_b[t]) {
        s_b[t] = s_b[t + s];
        s_i[t] = s_i[t + s];
      }
      __syncthreads();
    }
    const bool any = s_b[0] < 3.0e38f;
    const int winner = s_i[0];
    __syncthreads();
    if (!any) break;
    if (t == winner) {
#pragma unroll
      for (int e = 0; e < EPT; ++e)
        if (e == best_e) {
          sum += val(best_bits) - val(bits[e]);
          bits[e] = best_bits;
        }
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int k = t + 256 * e;
    if (k < K) ws[(int64_t)n * K + k] = __builtin_bit_cast(T, bits[e]);
  }
}

extern "C" int mio_ln_fold_weight(const void* w, int64_t ldw, const void* gamma, const void* beta, const void* bias,
                                  void* w_scaled, void* bias_out, int32_t N, int32_t K, int32_t dtype, void* stream) {
  MIO_CHECK(w && gamma && w_scaled && bias_out, "mio_ln_fold_weight: w, gamma, w_scaled, bias_out must be non-null");
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_ln_fold_weight: dtype must be bf16 or fp16");
  MIO_CHECK(N > 0 && K > 0 && K <= 8192 && ldw >= K, "mio_ln_fold_weight: bad sizes (K <= 8192)");
#define MIO_LNFW(T_, E_)                                                                                                       \
  hipLaunchKernelGGL((ln_fold_weight_kernel<T_, E_>), dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, (const T_*)w, ldw, \
                     (const T_*)gamma, (const T_*)beta, (const T_*)bias, (T_*)w_scaled, (T_*)bias_out, K)
  if (dtype == MIO_BF16) {
    if (K <= 2048) MIO_LNFW(__bf16, 8); else MIO_LNFW(__bf16, 32);
  } else {
    if (K <= 2048) MIO_LNFW(_Float16, 8); else MIO_LNFW(_Float16, 32);
  }
#undef MIO_LNFW
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("mio_ln_fold_weight launch: ") + hipGetErrorString(e));
  return 0;
}

// rows wider than 2048 columns leave more than GEMM_LN_SLOTS_MAX statistic slots: summed in groups (fixed order) down to a count
// the consumer's LDS region holds -- one small launch per LayerNorm, against a LayerNorm pass over the whole stream
__global__ void ln_stats_reduce_kernel(const float* __restrict__ in, float* __restrict__ out, int per, int64_t n2, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over slots_out * rows * 2 floats
  if (i >= total) return;
  const int64_t so = i / n2, r = i % n2;
  float a = 0.f;
  for (int j = 0; j < per; ++j) a += in[(so * per + j) * n2 + r];
  out[i] = a;
}

extern "C" int mio_ln_stats_reduce(const float* stats_in, int32_t slots_in, float* stats_out, int32_t slots_out, int64_t M, void* stream) {
  MIO_CHECK(stats_in && stats_out, "mio_ln_stats_reduce: null pointer");
  MIO_CHECK(slots_in > 0 && slots_out > 0 && slots_in % slots_out == 0 && M >= 0, "mio_ln_stats_reduce: slots_in must be a multiple of slots_out");
  const int64_t n2 = (M + 255) / 256 * 256 * 2, total = n2 * slots_out;
  if (total == 0) return 0;
  hipLaunchKernelGGL(ln_stats_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, stats_in, stats_out,
                     slots_in / slots_out, n2, total);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mio_fail(std::string("mio_ln_stats_reduce launch: ") + hipGetErrorString(e));
  return 0;
}

extern "C" int mio_gemm_ln_bw(const void* x, const void* wb, const void* bias, const void* bias_gate, const void* residual, void* y, int64_t M, int32_t N,
                              int32_t K, int64_t ldx, int64_t ldy, int64_t ldr, int32_t act, int32_t dtype, int32_t flags,
                              const float* ln_stats, int32_t ln_slots, float ln_eps, float* stats_out, int32_t cs_lo, int32_t cs_hi,
                              float cs_val, void* stream) {
  const bool xb = (flags & MIO_GEMM_X_BLOCKED) != 0, yb = (flags & MIO_GEMM_Y_BLOCKED) != 0, rb = (flags & MIO_GEMM_RES_BLOCKED) != 0;
  MIO_CHECK(x && wb && y, "mio_gemm_ln_bw: x, wb, y must be non-null");
  MIO_CHECK(M >= 0 && N > 0 && K > 0, "mio_gemm_ln_bw: bad sizes");
  if (M == 0) return 0;
  MIO_CHECK(dtype == MIO_BF16 || dtype == MIO_FP16, "mio_gemm_ln_bw: dtype must be bf16 or fp16");
  MIO_CHECK(act >= MIO_ACT_NONE && act <= MIO_ACT_SWIGLU, "mio_gemm_ln_bw: unknown activation");
  MIO_CHECK((flags & ~7) == 0, "mio_gemm_ln_bw: unknown flag");
  MIO_CHECK(act == MIO_ACT_SWIGLU || bias_gate == nullptr, "mio_gemm_ln_bw: bias_gate belongs to the gated stage (act == SWIGLU)");
  MIO_CHECK(act != MIO_ACT_SWIGLU || residual == nullptr, "mio_gemm_ln_bw: the gated stage takes no residual");
  MIO_CHECK(mio_aligned16(bias_gate), "mio_gemm_ln_bw: pointers must be 16-byte aligned");
  MIO_CHECK(mio_gemm_ln_ok(M, N, K, act, ln_stats != nullptr, stats_out != nullptr),
            "mio_gemm_ln_bw: this shape / activation does not take the folded kernels (mio_gemm_ln_ok == 0)");
  MIO_CHECK(ln_stats == nullptr || residual == nullptr, "mio_gemm_ln_bw: the consumer form takes no residual");
  if (ln_stats != nullptr && ln_slots == 0) ln_slots = K / 256;
  MIO_CHECK(ln_stats == nullptr || (ln_slots >= 1 && ln_slots <= GEMM_LN_SLOTS_MAX),
            "mio_gemm_ln_bw: at most 8 statistic slots (rows wider than 2048 columns: mio_ln_stats_reduce first)");
  MIO_CHECK(stats_out == nullptr || residual != nullptr, "mio_gemm_ln_bw: the producer form is the residual epilogue");
  MIO_CHECK(!rb || residual != nullptr, "mio_gemm_ln_bw: RES_BLOCKED without a residual");
  if (xb) ldx = K;
  if (yb) ldy = N;
  if (rb) ldr = N;
  MIO_CHECK(ldx % 8 == 0 && ldy % 8 == 0 && (residual == nullptr || ldr % 8 == 0) && ldx >= K && ldy >= N &&
                (residual == nullptr || ldr >= N),
            "mio_gemm_ln_bw: bad strides");
  MIO_CHECK(ldx * 512 < (int64_t)0x7fffffff && ldy * 512 < (int64_t)0x7fffffff && (residual == nullptr || ldr * 512 < (int64_t)0x7fffffff),
            "mio_gemm_ln_bw: row stride too large");
  MIO_CHECK(mio_aligned16(x) && mio_aligned16(wb) && mio_aligned16(y) && mio_aligned16(residual) && mio_aligned16(bias) &&
                mio_aligned16(ln_stats) && mio_aligned16(stats_out),
            "mio_gemm_ln_bw: pointers must be 16-byte aligned");
  MIO_CHECK(cs_lo >= cs_hi || (residual == nullptr && cs_lo >= 0 && cs_hi <= N && cs_lo % 128 == 0 && cs_hi % 128 == 0),
            "mio_gemm_ln_bw: [cs_lo, cs_hi) must be multiples of 128 inside [0, N], without a residual");
  GemmDev p;
  gemm_dev_defaults(p);
  p.x = x; p.w = wb; p.bias = bias; p.bias_g = bias_gate; p.res = residual; p.y = y;
  p.M = M; p.ldx = ldx; p.ldw = K; p.ldy = ldy; p.ldr = ldr; p.N = N; p.K = K;
  p.x_blk = xb ? 1 : 0; p.y_blk = yb ? 1 : 0; p.w_blk = (act == MIO_ACT_SWIGLU) ? 2 : 1; p.res_blk = rb ? 1 : 0;
  p.cs_lo = cs_lo; p.cs_hi = cs_hi; p.cs_val = cs_val;
  p.ln_stats = ln_stats; p.ln_eps = ln_eps; p.ln_slots = ln_stats ? ln_slots : 0;
  p.stats_out = stats_out;
  return gemm_dispatch(p, act, dtype, (hipStream_t)stream);
}
