// valu4_probe.hip -- SATURATED issue rates of the softmax-side VALU instructions of the attention kernels on gfx950, by WALL time of
// long kernels (hipEvents over ~ms runs: the 12-us kernels of valu_probe.hip are dominated by launch overhead in their event column),
// at 1..4 waves per SIMD, for plain / packed-f32 / packed-f16 / transcendental / convert opcodes, alone and beside MFMAs in the
// proportion of the key-blocked window-attention kernels (attention_win4.hip).  Reports ns and shader cycles (at the clock the kernel
// ran at, from s_memtime / wall) per instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 valu4_probe.hip -o valu4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <initializer_list>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
constexpr int ITER = 20000, UN = 16;

// MODE: 0 v_fma_f32  1 v_pk_fma_f32  2 v_exp_f32  3 v_cvt_pk_bf16_f32  4 v_max3_f32  5 v_pk_add_f32  6 v_pk_mul_f32  7 v_add_f32
//       8 v_pk_fma_f16  9 v_pk_mul_f16  10 v_cvt_pkrtz_f16_f32  11 v_mul_f32  12 v_exp_f16 (no packed form)  13 v_dot2_f32_bf16
//       20 mfma 16x16x32 alone   21 mfma 32x32x16 alone
//       30 forward chain per 16 "elements pairs": 8 pk_fma + 16 exp + 8 cvt_pk          (= 16 score elements per lane)
//       31 the same with plain fma: 16 fma + 16 exp + 8 cvt_pk
//       32 forward block: 30 + 5 mfma 16x16x32 (the win4 forward: 10 MFMAs per 32 elements per lane)
//       33 forward block with plain fma: 31 + 5 mfma
//       34 backward chain per 16 elements: 8 pk_fma + 16 exp + 8 pk_fma(dp*ss - delta) + 8 pk_mul + 8 cvt (ds) [+ 8 cvt (p) for dkv]
//       35 = 34 + 8 mfma
template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc, float seed) {
  const int tid = threadIdx.x;
  f2 a[UN]; f4 acc[4]; float s[UN]; unsigned w[UN];
  typedef __attribute__((ext_vector_type(16))) float f16v;
  f16v big[2];
#pragma unroll
  for (int i = 0; i < UN; ++i) { a[i] = f2{seed + i + tid, seed - i}; s[i] = seed * i + tid; w[i] = 0x3c003c00u + i; }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) { big[0][i] = 0.f; big[1][i] = 0.f; }
  const f2 c1 = {1.0001f, 0.9999f}, c2 = {1e-3f, -1e-3f};
  const unsigned h1 = 0x3c003c01u, h2 = 0x10001000u;
  bf8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(c1[0]), "v"(c2[0]));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
      if (MODE == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
      if (MODE == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(a[i][0]), "v"(a[i][1]));
      if (MODE == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a[i][0]), "v"(a[i][1]));
      if (MODE == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
      if (MODE == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
      if (MODE == 7) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c2[0]));
      if (MODE == 8) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(w[i]) : "v"(h1), "v"(h2));
      if (MODE == 9) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(w[i]) : "v"(h1));
      if (MODE == 10) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(a[i][0]), "v"(a[i][1]));
      if (MODE == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c1[0]));
      if (MODE == 12) asm volatile("v_exp_f16 %0, %0" : "+v"(w[i]));
      if (MODE == 13) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(s[i]) : "v"(w[i]), "v"(h1));
      if (MODE == 20) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
      if (MODE == 21) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, big[i & 1], 0, 0, 0);
    }
    if (MODE == 30 || MODE == 32) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 32 && i < 5) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i]) : "v"(a[i][0]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i + 1]) : "v"(a[i][1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]));
      }
    }
    if (MODE == 31 || MODE == 33) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 33 && i < 5) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][0]) : "v"(c1[0]), "v"(c2[0]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i][1]) : "v"(c1[0]), "v"(c2[0]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i]) : "v"(a[i][0]));
        asm volatile("v_exp_f32 %0, %1" : "=v"(s[2 * i + 1]) : "v"(a[i][1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(s[2 * i]), "v"(s[2 * i + 1]));
      }
    }
    if (MODE == 34 || MODE == 35) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 35) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i & 3], 0, 0, 0);
        f2 e = a[i], d = a[(i + 3) & 7 | 8];
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(e) : "v"(c1), "v"(c2));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[0]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(e[1]));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(c1), "v"(c2));
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d) : "v"(e));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(d[0]), "v"(d[1]));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i + 8]) : "v"(e[0]), "v"(e[1]));
        a[i] = e; a[(i + 3) & 7 | 8] = d;
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < UN; ++i) r += a[i][0] + a[i][1] + s[i] + (float)w[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
  r += big[0][0] + big[1][5];
  if (r == 12345.678f) out[0] = r;
  if ((tid & 63) == 0) cyc[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
int run(const char* name, int per_iter, float* out, unsigned long long* cyc) {
  for (int nt : {256, 512, 768, 1024}) {
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nt), 0, 0, out, cyc, 1.0f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nt), 0, 0, out, cyc, 1.0f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    std::vector<unsigned long long> h(4096);
    CK(hipMemcpy(h.data(), cyc, 4096 * 8, hipMemcpyDeviceToHost));
    double s = 0; int n_ = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < nt / 64; ++w) { s += (double)h[b * 16 + w]; ++n_; }
    s /= n_;
    const double n = (double)ITER * per_iter, wps = nt / 256.0;
    const double ghz = s / (ms * 1e6);                       // wave ticks per wall ns ~ the shader clock (a wave spans nearly the whole kernel)
    printf("%-46s waves/SIMD %d : %7.3f cyc/instr/wave  %6.3f cyc/instr/SIMD  %6.3f ns/instr/SIMD (wall %.3f ms, ~%.2f GHz)\n", name, nt / 256, s / n, s / n / wps,
           ms * 1e6 / n / wps, ms, ghz);
  }
  return 0;
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 4096 * 8));
  run<0>("v_fma_f32", UN, out, cyc);
  run<7>("v_add_f32", UN, out, cyc);
  run<11>("v_mul_f32", UN, out, cyc);
  run<1>("v_pk_fma_f32", UN, out, cyc);
  run<5>("v_pk_add_f32", UN, out, cyc);
  run<6>("v_pk_mul_f32", UN, out, cyc);
  run<2>("v_exp_f32", UN, out, cyc);
  run<3>("v_cvt_pk_bf16_f32", UN, out, cyc);
  run<4>("v_max3_f32", UN, out, cyc);
  run<8>("v_pk_fma_f16", UN, out, cyc);
  run<9>("v_pk_mul_f16", UN, out, cyc);
  run<10>("v_cvt_pkrtz_f16_f32", UN, out, cyc);
  run<12>("v_exp_f16", UN, out, cyc);
  run<13>("v_dot2_f32_bf16", UN, out, cyc);
  run<20>("v_mfma_f32_16x16x32_bf16", UN, out, cyc);
  run<21>("v_mfma_f32_32x32x16_bf16", UN, out, cyc);
  run<30>("fwd chain: 8 pk_fma + 16 exp + 8 cvt  (per group)", 1, out, cyc);
  run<31>("fwd chain: 16 fma + 16 exp + 8 cvt    (per group)", 1, out, cyc);
  run<32>("fwd block: chain(pk) + 5 mfma         (per group)", 1, out, cyc);
  run<33>("fwd block: chain(plain) + 5 mfma      (per group)", 1, out, cyc);
  run<34>("bwd chain: 16 pk + 16 exp + 8 pk_mul + 16 cvt (per group)", 1, out, cyc);
  run<35>("bwd block: chain + 8 mfma             (per group)", 1, out, cyc);
  return 0;
}
