// store_pattern_probe.hip -- what limits the GEMM epilogue's stores per CU?  One 512-thread workgroup per CU writes a [rows][pitch] bf16
// output tile-by-tile the way the 256x256 kernel does (a wave instruction = R rows x (1024 / R) contiguous bytes of ITS 128-byte-wide
// column strip, rows `pitch` bytes apart), for several pitches and row counts per instruction, against fully contiguous 1 KiB stores.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/store_pattern_probe.hip -o /tmp/spp && /tmp/spp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// each wave: `n_inst` store instructions; instruction i writes R rows x SEG bytes (SEG = 1024 / R): lane -> (row = lane / (SEG/16), 16-byte chunk)
// wave w of the workgroup owns the column strip starting at byte w * SEG_W of the tile row; successive instructions walk down the rows
template <int R>
__global__ __launch_bounds__(512) void pat_kernel(unsigned char* base, size_t pitch, int n_inst, int strip_bytes, size_t wg_stride) {
  constexpr int SEG = 1024 / R, LPR = SEG / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned char* p = base + (size_t)blockIdx.x * wg_stride + (size_t)wave * strip_bytes;
  const int row = lane / LPR, ch = lane % LPR;
  const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3u, 4u);
  for (int i = 0; i < n_inst; ++i) {
    // rows advance by R per instruction; when the strip is wider than SEG, consecutive instructions first cover the strip
    const int per_row_insts = strip_bytes / SEG;
    const size_t r0 = (size_t)(i / per_row_insts) * R, c0 = (size_t)(i % per_row_insts) * SEG;
    *reinterpret_cast<uint4*>(p + (r0 + row) * pitch + c0 + ch * 16) = v;
  }
}

int main() {
  unsigned char* buf;
  const size_t total = (size_t)6 << 30;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 0, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Cfg { const char* name; int R; size_t pitch; int strip; };
  // 8 waves per workgroup; a workgroup covers 8 strips side by side (its 1024 or 2048 byte wide column block) of `rows` rows
  const Cfg cfgs[] = {
      {"contiguous 1 KiB per instruction (pitch = strip = 1024, 1 row)", 1, 8 * 1024, 1024},
      {"8 rows x 128 B, pitch 6144 (N = 3072)", 8, 6144, 128},
      {"8 rows x 128 B, pitch 4608 (N = 2304)", 8, 4608, 128},
      {"8 rows x 128 B, pitch 1536 (N = 768)", 8, 1536, 128},
      {"8 rows x 128 B, pitch 6144 + 128", 8, 6272, 128},
      {"4 rows x 256 B, pitch 6144 (strip 256)", 4, 6144, 256},
      {"2 rows x 512 B, pitch 6144 (strip 512)", 2, 6144, 512},
      {"1 row x 1024 B, pitch 6144 (strip 1024)", 1, 6144, 1024},
      {"16 rows x 64 B, pitch 6144 (strip 128: two instructions per 128 B)", 16, 6144, 128},
  };
  for (int nwg : {1, 256}) {
    for (const Cfg& c : cfgs) {
      const int n_inst = 4096;                           // per wave: 4 MiB
      const size_t rows = (size_t)n_inst * (c.R == 16 ? 8 : c.R) / (c.strip / (1024 / c.R) > 0 ? (c.strip / (1024 / c.R)) : 1);
      size_t wg_stride = (rows + 64) * c.pitch;          // workgroups write disjoint row ranges
      if (c.pitch == 8 * 1024) wg_stride = (size_t)n_inst * 1024 * 8 + 4096;
      if (wg_stride * nwg > total) { printf("  (skipped: %s)\n", c.name); continue; }
      auto launch = [&]() {
        switch (c.R) {
          case 1: hipLaunchKernelGGL(pat_kernel<1>, dim3(nwg), dim3(512), 0, 0, buf, c.pitch, n_inst, c.strip, wg_stride); break;
          case 2: hipLaunchKernelGGL(pat_kernel<2>, dim3(nwg), dim3(512), 0, 0, buf, c.pitch, n_inst, c.strip, wg_stride); break;
          case 4: hipLaunchKernelGGL(pat_kernel<4>, dim3(nwg), dim3(512), 0, 0, buf, c.pitch, n_inst, c.strip, wg_stride); break;
          case 8: hipLaunchKernelGGL(pat_kernel<8>, dim3(nwg), dim3(512), 0, 0, buf, c.pitch, n_inst, c.strip, wg_stride); break;
          default: hipLaunchKernelGGL(pat_kernel<16>, dim3(nwg), dim3(512), 0, 0, buf, c.pitch, n_inst, c.strip, wg_stride); break;
        }
      };
      launch();
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 3; ++i) launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double bytes = 3.0 * n_inst * 1024.0 * 8 * nwg;
      printf("%3d workgroups: %-70s %7.1f GB/s total, %6.2f GB/s per CU\n", nwg, c.name, bytes / ms / 1e6, bytes / ms / 1e6 / nwg);
    }
  }
  return 0;
}
