// valu_probe.hip -- issue cost of the softmax-side VALU instructions of the attention kernels on gfx950, calibrated against
// v_mfma_f32_16x16x32_bf16 (16-17 shader cycles per instruction per SIMD, MI355X_MICROARCH.md): independent chains of one opcode,
// 1 or 2 waves per SIMD (256 / 512 threads, one workgroup per CU), ticks of s_memtime per instruction per wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_probe.hip -o valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <initializer_list>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
constexpr int ITER = 512, UN = 16;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, float seed) {
  const int tid = threadIdx.x;
  f2 a[UN]; f4 acc[4]; float s[UN]; unsigned w[UN];
#pragma unroll
  for (int i = 0; i < UN; ++i) { a[i] = f2{seed + i + tid, seed - i}; s[i] = seed * i + tid; w[i] = 0; }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  const f2 c1 = {1.0001f, 0.9999f}, c2 = {1e-3f, -1e-3f};
  bf8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (MODE >= 11) {                                       // role split: waves 0-3 multiply (16 MFMAs per iteration), waves 4-7 run 48 VALU of one opcode
    if (__builtin_amdgcn_readfirstlane(tid) < 256) {
#pragma unroll 1
      for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UN; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
      }
    } else {
#pragma unroll 1
      for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 3 * UN; ++i) {
          if (MODE == 11) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 15]) : "v"(c1), "v"(c2));
          if (MODE == 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i & 15]) : "v"(c1[0]), "v"(c2[0]));
          if (MODE == 13) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i & 15]));
          if (MODE == 14) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i & 15]) : "v"(a[i & 15][0]), "v"(a[i & 15][1]));
          if (MODE == 15) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i & 15]) : "v"(c1));
          if (MODE == 16) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i & 15]) : "v"(c1[0]));
        }
      }
    }
  } else
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
      if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c1[0]), "v"(c2[0]));
      if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
      if (MODE == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(a[i][0]), "v"(a[i][1]));
      if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
      if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
      if (MODE == 6) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
      if (MODE == 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c1[0]));
      if (MODE == 17) { typedef __attribute__((ext_vector_type(4))) short s4_; acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(s4_{1, 2, 3, 4}, s4_{4, 3, 2, 1}, acc[i & 3], 0, 0, 0); }
      if (MODE == 18) { typedef __attribute__((ext_vector_type(16))) float f16_; static_assert(sizeof(f16_) == 64, ""); }
      if (MODE == 8) { asm volatile("v_exp_f32 %0, %0" : "+v"(s[i])); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2)); }      // alternate exp / pk
      if (MODE == 9) { acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);                                               // 1 MFMA + 3 pk_fma
                       asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(i + 5) & 15]) : "v"(c1), "v"(c2));
                       asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[(i + 9) & 15]) : "v"(c1), "v"(c2)); }
      if (MODE == 10) { acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);                                              // 1 MFMA + 6 plain fma (same flops as 3 pk)
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c1[0]), "v"(c2[0])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(i + 3) & 15]) : "v"(c1[0]), "v"(c2[0]));
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(i + 6) & 15]) : "v"(c1[0]), "v"(c2[0])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(i + 9) & 15]) : "v"(c1[0]), "v"(c2[0]));
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(i + 12) & 15]) : "v"(c1[0]), "v"(c2[0])); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[(i + 13) & 15]) : "v"(c1[0]), "v"(c2[0])); }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < UN; ++i) r += a[i][0] + a[i][1] + s[i] + (float)w[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
  if (r == 12345.678f) out[0] = r;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (tid == 256) cyc[256 + blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, float* out, unsigned long long* cyc) {
  for (int nt : {256, 512, 768, 1024}) {
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nt), 0, 0, out, cyc, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nt), 0, 0, out, cyc, 1.0f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), cyc, 512 * 8, hipMemcpyDeviceToHost));
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i]; s /= 256;
    double s2 = 0; for (int i = 256; i < 512; ++i) s2 += (double)h[i]; s2 /= 256;
    const double n = (double)ITER * UN;
    printf("%-34s waves/SIMD %d : %8.3f ticks per instr(-group) per wave  (%.2f ns per instr per SIMD by events; wave 4: %.3f ticks)\n", name, nt / 256, s / n, ms * 1e6 / n / (nt / 256), nt == 512 ? s2 / n : 0.0);
  }
  return 0;
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 512 * 8));
  run<6>("v_mfma_f32_16x16x32_bf16", out, cyc);
  run<17>("v_mfma_f32_16x16x16_bf16", out, cyc);
  run<1>("v_fma_f32", out, cyc);
  run<7>("v_mul_f32", out, cyc);
  run<0>("v_pk_fma_f32", out, cyc);
  run<4>("v_pk_mul_f32", out, cyc);
  run<5>("v_pk_add_f32", out, cyc);
  run<2>("v_exp_f32", out, cyc);
  run<3>("v_cvt_pk_bf16_f32", out, cyc);
  run<8>("v_exp_f32 + v_pk_fma_f32", out, cyc);
  run<9>("mfma + 3 v_pk_fma_f32", out, cyc);
  run<10>("mfma + 6 v_fma_f32", out, cyc);
  run<11>("split: mfma | 3 v_pk_fma_f32", out, cyc);
  run<12>("split: mfma | 3 v_fma_f32", out, cyc);
  run<13>("split: mfma | 3 v_exp_f32", out, cyc);
  run<14>("split: mfma | 3 v_cvt_pk_bf16_f32", out, cyc);
  run<15>("split: mfma | 3 v_pk_mul_f32", out, cyc);
  run<16>("split: mfma | 3 v_mul_f32", out, cyc);
  return 0;
}
