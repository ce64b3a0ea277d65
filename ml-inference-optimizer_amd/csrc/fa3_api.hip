// C-ABI entry point mio_fa3_fwd: argument validation + dispatch (see include/mio_hip.h).
#include <cmath>

#include "fa3_fwd_kernel.h"

extern template int fa3_launch<__bf16, 64>(const FaDev&, int, int, hipStream_t);
extern template int fa3_launch<__bf16, 96>(const FaDev&, int, int, hipStream_t);
extern template int fa3_launch<__bf16, 128>(const FaDev&, int, int, hipStream_t);
extern template int fa3_launch<_Float16, 64>(const FaDev&, int, int, hipStream_t);
extern template int fa3_launch<_Float16, 96>(const FaDev&, int, int, hipStream_t);
extern template int fa3_launch<_Float16, 128>(const FaDev&, int, int, hipStream_t);

static bool strides_ok(const int64_t s[3]) { return (s[0] % 8 == 0) && (s[1] % 8 == 0) && (s[2] % 8 == 0); }

extern "C" int32_t mio_fa3_k_prescaled_ok(const mio_fa3_fwd_params_t* a) {
  if (a == nullptr) return 0;
  const bool span32 = (int64_t)a->Sk * a->k_stride[1] * 2 < (1ll << 32) && (int64_t)a->Sk * a->v_stride[1] * 2 < (1ll << 32);
  if (!(a->D <= 96 && a->mask_kind == MIO_MASK_NONE && a->Sq > 128 && span32)) return 0;
  const bool plain = a->o != nullptr && a->o_acc == nullptr && !a->carry_in;
  // the (o_acc, lse) ring carry: fa3_fwd5_kernel only (head dim <= 64)
  return (plain || (a->D <= 64 && a->o_acc != nullptr && a->lse != nullptr)) ? 1 : 0;
}

extern "C" int32_t mio_fa3_o_blocked_ok(const mio_fa3_fwd_params_t* a) {
  if (a == nullptr || !mio_fa3_k_prescaled_ok(a)) return 0;
  const bool plain = a->o != nullptr && a->o_acc == nullptr && !a->carry_in;
  return (plain && a->D <= 64 && ((int64_t)a->H * a->D) % 32 == 0) ? 1 : 0;
}

extern "C" int mio_fa3_fwd(const mio_fa3_fwd_params_t* a, void* stream) {
  MIO_CHECK(a != nullptr, "mio_fa3_fwd: null params");
  MIO_CHECK(a->q && a->k && a->v, "mio_fa3_fwd: q/k/v must be non-null");
  MIO_CHECK(a->o != nullptr || a->o_acc != nullptr, "mio_fa3_fwd: o or o_acc must be given");
  MIO_CHECK(a->B > 0 && a->H > 0 && a->Hkv > 0 && a->Sq >= 0 && a->Sk >= 0, "mio_fa3_fwd: bad sizes");
  MIO_CHECK(a->H % a->Hkv == 0, "mio_fa3_fwd: H must be a multiple of Hkv");
  MIO_CHECK(a->D >= 8 && a->D <= 128 && a->D % 8 == 0, "mio_fa3_fwd: head_dim must be a multiple of 8 in [8,128]");
  MIO_CHECK(a->dtype == MIO_BF16 || a->dtype == MIO_FP16, "mio_fa3_fwd: dtype must be bf16 or fp16");
  MIO_CHECK(a->softmax_scale > 0.f && std::isfinite(a->softmax_scale), "mio_fa3_fwd: softmax_scale must be > 0");
  MIO_CHECK(a->mask_kind >= 0 && a->mask_kind <= 2, "mio_fa3_fwd: bad mask_kind");
  MIO_CHECK((a->mask_kind == MIO_MASK_NONE) == (a->mask == nullptr), "mio_fa3_fwd: mask pointer / mask_kind mismatch");
  MIO_CHECK(strides_ok(a->q_stride) && strides_ok(a->k_stride) && strides_ok(a->v_stride) &&
                (a->o == nullptr || a->o_blocked || strides_ok(a->o_stride)),
            "mio_fa3_fwd: strides must be multiples of 8 elements (16-byte rows)");
  MIO_CHECK(mio_aligned16(a->q) && mio_aligned16(a->k) && mio_aligned16(a->v) && mio_aligned16(a->o) &&
                mio_aligned16(a->o_acc),
            "mio_fa3_fwd: pointers must be 16-byte aligned");
  MIO_CHECK(!a->carry_in || (a->o_acc && a->lse), "mio_fa3_fwd: carry_in needs o_acc and lse");
  MIO_CHECK(a->o_acc == nullptr || a->lse != nullptr, "mio_fa3_fwd: o_acc needs lse");
  if (a->Sq == 0) return 0;

  FaDev p;
  p.q = a->q; p.k = a->k; p.v = a->v; p.o = a->o; p.lse = a->lse; p.o_acc = a->o_acc; p.mask = a->mask;
  p.qs_b = a->q_stride[0]; p.qs_s = a->q_stride[1]; p.qs_h = a->q_stride[2];
  p.ks_b = a->k_stride[0]; p.ks_s = a->k_stride[1]; p.ks_h = a->k_stride[2];
  p.vs_b = a->v_stride[0]; p.vs_s = a->v_stride[1]; p.vs_h = a->v_stride[2];
  p.os_b = a->o_stride[0]; p.os_s = a->o_stride[1]; p.os_h = a->o_stride[2];
  p.ms_b = a->mask_stride[0]; p.ms_h = a->mask_stride[1]; p.ms_q = a->mask_stride[2]; p.ms_k = a->mask_stride[3];
  p.B = a->B; p.Sq = a->Sq; p.Sk = a->Sk; p.H = a->H; p.Hkv = a->Hkv; p.D = a->D;
  p.carry_in = a->carry_in; p.q_offset = a->q_offset; p.k_offset = a->k_offset;
  p.nqblk = (a->Sq + FA_BM - 1) / FA_BM;
  p.qgrid = p.nqblk;
  p.xcd_remap = ((a->B * a->H) % 8 == 0) ? 1 : 0;
  p.scale_log2e = a->softmax_scale * FA_LOG2E;
  p.k_prescaled = a->k_prescaled ? 1 : 0;
  p.o_blk = a->o_blocked ? 1 : 0;
  MIO_CHECK(!p.o_blk || (p.k_prescaled && mio_fa3_o_blocked_ok(a)), "mio_fa3_fwd: o_blocked is not supported for this launch "
                                                                     "(needs k_prescaled and mio_fa3_o_blocked_ok != 0)");
  MIO_CHECK(!p.k_prescaled || mio_fa3_k_prescaled_ok(a), "mio_fa3_fwd: k_prescaled is not supported for this launch "
                                                         "(mio_fa3_k_prescaled_ok == 0)");

  hipStream_t st = (hipStream_t)stream;
  const int dpad = a->D <= 64 ? 64 : (a->D <= 96 ? 96 : 128);
  if (a->dtype == MIO_BF16) {
    if (dpad == 64) return fa3_launch<__bf16, 64>(p, a->causal, a->mask_kind, st);
    if (dpad == 96) return fa3_launch<__bf16, 96>(p, a->causal, a->mask_kind, st);
    return fa3_launch<__bf16, 128>(p, a->causal, a->mask_kind, st);
  }
  if (dpad == 64) return fa3_launch<_Float16, 64>(p, a->causal, a->mask_kind, st);
  if (dpad == 96) return fa3_launch<_Float16, 96>(p, a->causal, a->mask_kind, st);
  return fa3_launch<_Float16, 128>(p, a->causal, a->mask_kind, st);
}
