// Groundwork for config 5's fp8 GEMM path (gfx950): sustained rate and operand layout of the fp8 MFMAs.
//   v_mfma_f32_16x16x32_fp8_fp8            (gfx940-style, 8 fp8 per lane per operand)
//   v_mfma_scale_f32_16x16x128_f8f6f4      (gfx950, 32 fp8 per lane per operand, E8M0 block scales; scale 127 = 1.0)
// Layout hypothesis checked numerically against a CPU product: lane l holds row/col l%16 and the 32 consecutive k
// k0 = 32*(l/16) .. k0+31 in its eight operand registers (register i = bytes k0+4i .. k0+4i+3, little endian).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__host__ __device__ inline float e4m3_to_f(uint8_t v) {          // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}

__global__ void layout_kernel(const uint8_t* A, const uint8_t* B, float* D) {      // A [16][128], B [16][128] (row = m or n), D [16][16]
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *reinterpret_cast<const int*>(A + r * 128 + g * 32 + 4 * i);
    b[i] = *reinterpret_cast<const int*>(B + r * 128 + g * 32 + 4 * i);
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int j = 0; j < 4; ++j) D[(g * 4 + j) * 16 + r] = c[j];     // accumulator: row 4g+j of the A side, column r of the B side
}

template <int NACC>
__global__ __launch_bounds__(256) void rate128(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = 0x38383838 + threadIdx.x; b[e] = 0x30303030 + e; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void rate32(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  long a = 0x3838383838383838L + threadIdx.x, b = 0x3030303030303030L;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename Kn>
void time_it(const char* name, Kn kern, int wgs_per_cu, double flop_per_mfma, int nacc) {
  float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)256 * wgs_per_cu * 4 * iters * nacc;
  printf("%-44s waves/SIMD=%d  %.3f ms  %.0f TFLOP/s\n", name, wgs_per_cu, ms, mfma * flop_per_mfma / ms / 1e9);
  hipFree(d);
}

int main() {
  // ---- layout
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  const uint8_t vals[8] = {0x00, 0x30, 0x38, 0x40, 0xb8, 0x34, 0x3c, 0x28};      // 0, .5, 1, 2, -1, .75, 1.5, .25
  for (int i = 0; i < 16 * 128; ++i) { A[i] = vals[(i * 7 + i / 128) % 8]; B[i] = vals[(i * 3 + 5 * (i / 128)) % 8]; }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  std::vector<float> D(256);
  hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double s = 0;
      for (int k = 0; k < 128; ++k) s += (double)e4m3_to_f(A[m * 128 + k]) * e4m3_to_f(B[n * 128 + k]);
      maxerr = fmax(maxerr, fabs(s - D[m * 16 + n]));
    }
  printf("f8f6f4 16x16x128 layout check (lane l: row l%%16, k = 32*(l/16)..+31; D[4g+j][r]): max |err| = %g  -> %s\n", maxerr, maxerr < 1e-3 ? "layout confirmed" : "MISMATCH");
  // ---- rates
  time_it("v_mfma_f32_16x16x32_fp8_fp8", rate32<8>, 1, 16384.0, 8);
  time_it("v_mfma_f32_16x16x32_fp8_fp8", rate32<8>, 2, 16384.0, 8);
  time_it("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)", rate128<8>, 1, 65536.0, 8);
  time_it("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)", rate128<8>, 2, 65536.0, 8);
  return 0;
}
