// Split-I fused-MLP experiment (reference kernels/triton/mlp_kernels.py:27-230 keeps a [BLOCK, I] strip on chip): the
// data flow of an MI355X version, measured before building the MFMA body.
//
// One persistent workgroup per CU owns a strip of R output rows x d columns whose fc2 partial sums stay in the accumulator
// file for the whole walk over I: R x d x 4 B <= 256 KiB (256 lanes x 256 accumulator registers) -> R = 64 at d = 1024.
// Per I-slice of 64 columns it needs the W1 slice [64, d] and the W2 slice [d, 64]: 2 x 128 KiB, i.e. ALL 16 MiB of
// weights per strip, 64 FLOP per weight byte.  M = 32768 rows = 512 strips -> 8 GiB of weight bytes have to reach the
// LDS per MLP (the two-launch blocked path moves 128 x 16 MiB = 2 GiB at 256-row tiles), in the 0.46 ms the MFMA work
// takes at the GEMM kernels' rate: 17.5 TB/s chip-wide, 68 GB/s per CU.
// This program streams exactly that: every workgroup walks the 64 contiguous 256-KiB slice pairs (best case: weights
// pre-packed per slice) through an 8 x 16 KiB LDS ring by LDS-DMA, twice (2 strips per CU), all CUs in the same order so
// that an XCD's L2 serves 31 of its 32 readers.  Variant 1 adds the MFMA load of the real kernel (16 x 16x16x32 MFMAs per
// wave per 16-KiB slot, register operands): what the stream sustains beside a busy matrix pipe.  Variant 2 also reads the
// weight fragments back from LDS the way the two halves of the real kernel would (fc1: one 1-KiB read per MFMA, the strip's
// 64 rows give no more reuse; fc2: one per 4 MFMAs).  Not included in any variant: the x strip / GELU / H hand-over through
// LDS and the epilogue, so every number is an upper bound for the fused kernel.
// Build: hipcc -O3 --offload-arch=gfx950 splitI_stream.hip -o splitI_stream ; run: ./splitI_stream
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define LDSP __attribute__((address_space(3)))

constexpr int SLOT = 16384, NSLOT = 8;

template <int MFMA>
__global__ __launch_bounds__(256) void stream_kernel(const char* w, size_t wbytes, int strips, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nslot_total = (int)(wbytes / SLOT);  // slots per pass over the weights
  f32x4 acc[16];
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    uint32_t s = threadIdx.x * 2654435761u + 12345u;
    uint32_t r[4];
    for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; r[i] = (s & 0x3f803f80u) | 0x3c003c00u; }
    a = __builtin_bit_cast(bf16x8, *(uint4*)r);
    b = a;
  }
  // each wave moves 4 x 1 KiB pieces of every 16-KiB slot; two slots stay in flight per wave (vmcnt(4) before a slot is
  // "consumed"); no barrier: a wave only ever rewrites its own quarter of a slot
  int issued = 0;
  const int total = nslot_total * strips;
  auto issue = [&](int slot_idx) {
    const int s = slot_idx % nslot_total;
    const char* src = w + (size_t)s * SLOT + wave * 4096 + lane * 16;
    const uint32_t lds = (uint32_t)(size_t)((LDSP char*)(smem + (slot_idx % NSLOT) * SLOT)) + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      asm volatile("s_add_i32 m0, %0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off"
                   :
                   : "s"(lds), "n"(0), "v"(src + i * 1024)
                   : "memory", "m0", "scc");
  };
  issue(0);
  issue(1);
  issue(2);
  issued = 3;
  for (int c = 0; c < total; ++c) {
    if (issued < total) issue(issued);
    ++issued;
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // slot c has landed (3 younger slots x 4 loads may fly)
    if (MFMA == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    if (MFMA == 2) {
      // the fc1 half of the real kernel: a wave owns 16 rows of the strip (its x fragments live in registers) and needs
      // EVERY W1 fragment of the slice -- one 1-KiB ds_read_b128 per MFMA, by all four waves; the fc2 half reuses each
      // fragment 4 times (64 rows) -- one read per 4 MFMAs
      const char* slot = smem + (c % NSLOT) * SLOT;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bf16x8 fa = __builtin_bit_cast(bf16x8, *(const uint4*)(slot + ((i * 1024 + lane * 16) & (SLOT - 1))));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, b, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int i = 8; i < 16; i += 4) {
        const bf16x8 fa = __builtin_bit_cast(bf16x8, *(const uint4*)(slot + (((i + wave) * 1024 + lane * 16) & (SLOT - 1))));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, b, acc[i + j], 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (sum == 123.456f) sink[0] = sum + *(float*)(smem + lane * 4);
}

int main() {
  const size_t wbytes = 16u << 20;  // W1 + W2 of the GPT-2-medium MLP (d 1024, I 4096, bf16)
  const int strips = 2;             // M = 32768 rows / 64 rows per strip / 256 CUs
  char* w; float* sink;
  hipMalloc(&w, wbytes); hipMalloc(&sink, 4);
  std::vector<uint16_t> h(wbytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff));
  hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    auto kern = variant == 0 ? stream_kernel<0> : (variant == 1 ? stream_kernel<1> : stream_kernel<2>);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), NSLOT * SLOT, 0, w, wbytes, strips, sink);
    hipDeviceSynchronize();
    const int n = 200;
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), NSLOT * SLOT, 0, w, wbytes, strips, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= n;
    const double bytes = 256.0 * strips * wbytes;
    const double flops = 4.0 * 32768 * 1024 * 4096;  // the MLP this stream would feed
    printf("variant %d (%s): %.3f ms per MLP-equivalent pass, %.2f TB/s into LDS chip-wide, %.1f GB/s per CU -> the fused kernel "
           "cannot beat %.0f TFLOP/s (two-launch path today: 0.472 ms = 1164 TFLOP/s)\n",
           variant, variant == 0 ? "stream only" : (variant == 1 ? "stream beside 16 MFMA / wave / slot, register operands" : "stream + MFMA + the W fragment reads from LDS"), ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
           flops / ms / 1e9);
  }
  return 0;
}
