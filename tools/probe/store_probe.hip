// store_probe.hip -- is the GEMM epilogue's store rate (about 13 B/clk/CU) a per-CU limit or a chip-wide one?
// Every workgroup streams 16-byte-per-lane stores (1 KiB per wave instruction, whole 128-byte lines) over its own region; the same
// per-workgroup byte count with 32 / 64 / 128 / 256 / 512 workgroups (one per CU up to 256, two per CU at 512), plain / nt stores,
// and the same with 16-byte loads for comparison.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/store_probe.hip -o /tmp/store_probe && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>      // 0 plain store, 1 nontemporal store, 2 load (sum kept alive)
__global__ __launch_bounds__(256) void stream_kernel(uint4* base, size_t bytes_per_wg, int reps) {
  uint4* p = base + (size_t)blockIdx.x * (bytes_per_wg / 16);
  const size_t n = bytes_per_wg / 16;                    // uint4 per workgroup
  uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3u, 4u);
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int r = 0; r < reps; ++r)
    for (size_t i = threadIdx.x; i < n; i += 256 * 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t j = i + (size_t)u * 256;
        if (j < n) {
          if (MODE == 0) p[j] = v;
          else if (MODE == 1) { typedef __attribute__((ext_vector_type(4))) unsigned u32x4; __builtin_nontemporal_store(u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(&p[j])); }
          else { const uint4 t = p[j]; acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w; }
        }
      }
    }
  if (MODE == 2 && acc.x == 0x12345678u) p[0] = acc;
}

int main() {
  const size_t per_wg = (size_t)8 << 20;                 // 8 MiB per workgroup per repetition
  uint4* buf;
  CK(hipMalloc(&buf, per_wg * 512));
  CK(hipMemset(buf, 0, per_wg * 512));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"store", "store nt", "load"};
  for (int mode = 0; mode < 3; ++mode)
    for (int nwg : {32, 64, 128, 256, 512}) {
      const int reps = 4;
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, reps);
        else if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, reps);
        else hipLaunchKernelGGL(stream_kernel<2>, dim3(nwg), dim3(256), 0, 0, buf, per_wg, reps);
      };
      launch();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; ++i) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = 3.0 * reps * (double)per_wg * nwg;
      printf("%-9s %3d workgroups (256 threads each): %7.1f GB/s total, %6.2f GB/s per workgroup\n", names[mode], nwg, bytes / ms / 1e6, bytes / ms / 1e6 / nwg);
    }
  return 0;
}
