// lds_atomic_probe.hip -- what one CU's LDS pipe sustains for the accesses the window-attention redesign weighs:
//   * ds_add_f32 (no return) conflict-free / 2-way / 4-way same-address collisions inside a wave instruction
//   * ds_read_b128 / ds_read_b64 / ds_read2_b32 gathers with the address pattern of a Toeplitz bias-table row fetch
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 lds_atomic_probe.hip -o lds_atomic_probe ; run without arguments.
// Output: cycles per wave-instruction per CU with 8 and 16 waves resident (all issuing the same instruction stream).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int ITER = 2048;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += blockDim.x) f[i] = 0.f;
  __syncthreads();
  const int r = lane & 15, g = lane >> 4;
  uint32_t a;
  if (MODE == 0) a = (uint32_t)(wave * 256 + lane * 4);                                   // add: 64 distinct banks
  if (MODE == 1) a = (uint32_t)(wave * 256 + (lane >> 1) * 4);                            // add: pairs of lanes share an address
  if (MODE == 2) a = (uint32_t)(wave * 256 + (lane >> 2) * 4);                            // add: 4 lanes share an address
  if (MODE == 3) a = (uint32_t)(wave * 256 + ((lane >> 1) * 8) % 256);                    // add: 2-way BANK conflict, distinct addresses (stride 2 dwords)
  if (MODE == 4 || MODE == 5 || MODE == 6) {
    // Toeplitz row fetch: 4 rows per wave instruction (2 q positions x 2 key positions), start s in [0, 11] inside the row
    const int lq = r >> 3, dq = r & 7, lk = g >> 1, gb = g & 1;
    const int rho = 40 + lq - lk + wave;                                                   // adjacent rows, as in natural order
    const int s = 7 - dq + 4 * gb;
    if (MODE == 4) a = (uint32_t)(rho * 192 + s * 16);                                     // 12 aligned 16-byte windows per row (b128)
    if (MODE == 5) a = (uint32_t)(rho * 64 + s * 4);                                       // 15 f32 per row, two ds_read2_b32
    if (MODE == 6) a = (uint32_t)(rho * 96 + s * 8);                                       // bf16 windows, 8 bytes each (b64)
  }
  if (MODE == 7) a = (uint32_t)(lane * 16 + wave * 1024);                                  // plain conflict-free b128
  float acc = 0.f;
  typedef __attribute__((ext_vector_type(4))) float f4;
  typedef __attribute__((ext_vector_type(2))) float f2;
  f4 v4 = {0.f, 0.f, 0.f, 0.f}; f2 v2 = {0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER / 8; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE <= 3) {
        asm volatile("ds_add_f32 %0, %1" ::"v"(a), "v"(1.0f) : "memory");
      } else if (MODE == 4 || MODE == 7) {
        f4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(a) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        v4 += t;
      } else if (MODE == 5) {
        f2 t0_, t1_;
        asm volatile("ds_read2_b32 %0, %2 offset0:0 offset1:1\n\tds_read2_b32 %1, %2 offset0:2 offset1:3" : "=&v"(t0_), "=&v"(t1_) : "v"(a) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        v2 += t0_ + t1_;
      } else if (MODE == 6) {
        f2 t;
        asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(a) : "memory");
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        v2 += t;
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  acc = v4[0] + v4[1] + v4[2] + v4[3] + v2[0] + v2[1] + f[tid];
  if (acc == 12345.678f) out[0] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
int run(const char* name, float* out, unsigned long long* cyc) {
  for (int nw : {4, 8, 16}) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nw * 64), 65536, 0, out, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(nw * 64), 65536, 0, out, cyc);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto x : h) s += (double)x;
    s /= 256;
    // s = counter ticks (100 MHz wall clock on gfx950 via readcyclecounter? printed raw) for ITER instructions per wave
    printf("%-44s waves/CU %2d : %10.1f ticks per %d wave-instructions -> %.3f ticks per wave-instr per CU\n", name, nw, s, ITER, s / (ITER * (double)nw));
  }
  return 0;
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&cyc, 256 * 8));
  // calibrate the tick: s_memtime / readcyclecounter counts shader cycles or a fixed clock; time one kernel with events too
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 65536, 0, out, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 65536, 0, out, cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h0; CK(hipMemcpy(&h0, cyc, 8, hipMemcpyDeviceToHost));
  printf("calibration: kernel %.1f us by events, %llu ticks in the timed loop of block 0 -> %.1f ticks/us\n", ms * 1e3, h0, h0 / (ms * 1e3));
  run<0>("ds_add_f32 conflict-free", out, cyc);
  run<1>("ds_add_f32 2 lanes per address", out, cyc);
  run<2>("ds_add_f32 4 lanes per address", out, cyc);
  run<3>("ds_add_f32 2-way bank conflict", out, cyc);
  run<7>("ds_read_b128 conflict-free", out, cyc);
  run<4>("ds_read_b128 Toeplitz windows (192 B rows)", out, cyc);
  run<5>("2 x ds_read2_b32 Toeplitz (64 B rows)", out, cyc);
  run<6>("ds_read_b64 Toeplitz bf16 (96 B rows)", out, cyc);
  return 0;
}
