// Bare MFMA loops on random register operands: what the chip sustains per MFMA shape / waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE, int NACC, int LB = 512, int BAR = 0>
__global__ __launch_bounds__(LB) void k(const uint4* in, float* out, int iters) {
  const int lane = threadIdx.x;
  bf16x8 a[4], b[4];
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {
    a[i] = __builtin_bit_cast(bf16x8, in[(lane * 8 + i) % 4096]);
    b[i] = __builtin_bit_cast(bf16x8, in[(lane * 8 + 4 + i) % 4096]);
  }
  if (SHAPE == 32) {
    f32x16 acc[NACC];
    _Pragma("unroll") for (int n = 0; n < NACC; ++n) { _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[n][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n & 3], b[(n >> 2) & 3], acc[n], 0, 0, 0);
      if (BAR & 1) __builtin_amdgcn_s_barrier();
      if (BAR & 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    _Pragma("unroll") for (int n = 0; n < NACC; ++n) { _Pragma("unroll") for (int i = 0; i < 16; ++i) s += acc[n][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  } else {
    f32x4 acc[NACC * 4];
    _Pragma("unroll") for (int n = 0; n < NACC * 4; ++n) { _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[n][i] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < NACC * 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[n & 3], b[(n >> 2) & 3], acc[n], 0, 0, 0);
    }
    float s = 0;
    _Pragma("unroll") for (int n = 0; n < NACC * 4; ++n) { _Pragma("unroll") for (int i = 0; i < 4; ++i) s += acc[n][i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  }
}

template <int SHAPE, int NACC, int LB = 512, int BAR = 0>
void run(const char* name, int threads, const uint4* in, float* out, int iters = 20000, int lds = 0, int reps = 2) {
  const int blocks = 256;
  if (lds) hipFuncSetAttribute((const void*)k<SHAPE, NACC, LB, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL((k<SHAPE, NACC, LB, BAR>), dim3(blocks), dim3(threads), lds, 0, in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(s);
  for (int rep = 0; rep < reps; ++rep)
    hipLaunchKernelGGL((k<SHAPE, NACC, LB, BAR>), dim3(blocks), dim3(threads), lds, 0, in, out, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); ms /= reps;
  const double flops_per = (SHAPE == 32) ? 32768.0 * NACC : 16384.0 * NACC * 4;
  const double fl = flops_per * iters * (threads / 64) * blocks;
  printf("%-34s %4d thr/blk  %8.3f ms  %7.1f TFLOP/s\n", name, threads, ms, fl / (ms * 1e-3) / 1e12);
}

int main() {
  std::vector<uint16_t> h(4096 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); v = u >> 16; }
  uint4* in; float* out;
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  run<32, 8>("32x32x16 8acc 1 wave/SIMD", 256, in, out);
  run<32, 8>("32x32x16 8acc 2 waves/SIMD", 512, in, out);
  run<32, 16, 256>("32x32x16 16acc(256 regs) 1w/SIMD", 256, in, out);
  run<16, 8>("16x16x32 32acc 1 wave/SIMD", 256, in, out);
  run<16, 8>("16x16x32 32acc 2 waves/SIMD", 512, in, out);
  run<16, 16, 256>("16x16x32 64acc(256 regs) 1w/SIMD", 256, in, out);
  run<32, 16, 256, 1>("32x32x16 16acc + s_barrier/iter", 256, in, out);
  run<32, 16, 256, 2>("32x32x16 16acc + waitcnt/iter", 256, in, out);
  run<32, 16, 256, 3>("32x32x16 16acc + both", 256, in, out);
  run<32, 16, 256, 3>("32x32 16acc both, 128KB LDS", 256, in, out, 20000, 131072);
  run<32, 16, 256, 3>("32x32 16acc both, short (256 it) x20", 256, in, out, 256, 0, 20);
  run<32, 16, 256, 3>("32x32 16acc both, short+LDS x20", 256, in, out, 256, 131072, 20);
  run<16, 16, 256, 0>("16x16 64acc short (256 it) x20", 256, in, out, 256, 0, 20);
  {  // gaussian operands like the real GEMM: A-side N(0, 0.02), B-side N(0, 1) (lane*8+i picks a/b from halves)
    auto gauss = [] { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307 * v); };
    for (size_t i = 0; i < h.size(); ++i) { float f = (float)gauss() * (((i / 8) % 8) < 4 ? 0.02f : 1.0f); uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<32, 16, 256, 3>("gauss 32x32 16acc both", 256, in, out);
    run<16, 16, 256, 0>("gauss 16x16 64acc", 256, in, out);
    for (size_t i = 0; i < h.size(); ++i) { float f = (float)gauss(); uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    run<32, 16, 256, 3>("N(0,1) both 32x32 16acc", 256, in, out);
    run<16, 16, 256, 0>("N(0,1) both 16x16 64acc", 256, in, out);
  }
  hipMemset(in, 0, h.size() * 2);
  run<32, 8>("zeros 32x32x16 1 wave/SIMD", 256, in, out);
  run<16, 8>("zeros 16x16x32 1 wave/SIMD", 256, in, out);
  return 0;
}
