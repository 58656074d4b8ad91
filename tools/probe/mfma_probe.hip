// Microbenchmark: do VALU instructions of the SAME wave hide in the shadow of its MFMAs?  (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NV, int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(0.5f + e); }
  float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.0f, v3 = 3.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      if (NV >= 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
      if (NV >= 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2));
      if (NV >= 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
      if (NV >= 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2));
      if (NV >= 6) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2)); }
      if (NV >= 8) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2)); }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v0 + v3;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * NACC);
}

template <int NV, int NACC>
void run(const char* name, int wgs_per_cu) {
  float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<NV, NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<NV, NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  const double mfma = (double)256 * wgs_per_cu * 4 * iters * NACC;       // wave-level MFMAs
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  counter cycles per (MFMA + %d VALU) = %.1f\n", name, wgs_per_cu, ms, mfma * 16384.0 / ms / 1e9, NV, cyc);
  hipFree(d);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int NACC>
__global__ __launch_bounds__(256) void probe32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(0.5f + e); }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0) / (float)(iters * NACC);
}
void run32(int wgs_per_cu) {
  float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
  const int iters = 2000; constexpr int NACC = 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe32<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe32<NACC>), dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
  const double mfma = (double)256 * wgs_per_cu * 4 * iters * NACC;
  printf("%-28s waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  counter cycles per MFMA = %.1f\n", "mfma 32x32x16 only", wgs_per_cu, ms, mfma * 32768.0 / ms / 1e9, cyc);
  hipFree(d);
}
int main() {
  run32(1); run32(2);
  for (int w = 1; w <= 2; ++w) {
    run<0, 16>("mfma only", w);
    run<1, 16>("mfma + 1 valu", w);
    run<2, 16>("mfma + 2 valu", w);
    run<3, 16>("mfma + 3 valu", w);
    run<4, 16>("mfma + 4 valu", w);
    run<6, 16>("mfma + 6 valu", w);
    run<8, 16>("mfma + 8 valu", w);
  }
  return 0;
}
