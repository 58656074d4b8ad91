// gather_bw_probe.hip -- what HBM delivers for the fusion attention kernels' operand pattern: every workgroup pulls 432 rows x 128 B
// (one head's K rows of one sequence) out of a [tokens][2304] bf16 matrix (row stride 4 608 B), twice (K and V), against the same bytes
// laid out contiguously per (sequence, head).  Direct-to-LDS DMA, 12 waves, one workgroup per CU at a time (112 KB of LDS), no compute.
//   hipcc --offload-arch=gfx950 -O3 -o gather_bw_probe tools/probe/gather_bw_probe.hip && ./gather_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;

__global__ __launch_bounds__(768) void gather_kernel(const uint16_t* base, long unit_stride, long head_stride, int row_stride, int heads, int* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, wave_base = tid & ~63;
  const int unit = blockIdx.x, seq = unit / heads, h = unit - seq * heads;
  for (int img = 0; img < 2; ++img) {
    const uint16_t* src = base + seq * unit_stride + h * head_stride + img * (heads * head_stride);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(src), 0, 0x7fffffff, 0x00020000);
    for (int i0 = 0; i0 < 432 * 8; i0 += 768) {
      const int u = i0 + tid;
      if (u < 432 * 8) {
        const int row = u >> 3, ch = u & 7;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(smem + img * 57344 + (size_t)(i0 + wave_base) * 16), 16, (unsigned)(row * row_stride * 2 + ch * 16), 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0 && smem[unit & 1023] == 0x7f && sink) sink[0] = 1;
}

int main() {
  const int nseq = 160, heads = 12, L = 432, ld = 2304;
  const size_t n = (size_t)nseq * L * ld;
  // three buffers walked in turn: 3 x 212 MB of K / V rows do not stay in the 256 MB memory-side cache between launches
  uint16_t* bufs[3]; int* sink;
  for (auto& b : bufs) { hipMalloc(&b, n * 2); hipMemset(b, 1, n * 2); }
  hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void*)gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 57344);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char* name; long unit_stride, head_stride; int row_stride; } cases[] = {
    {"token-major rows (stride 4608 B: [token][q|k|v][head][64])", (long)L * ld, 64, ld},
    {"head-major (contiguous 55 KB per (sequence, head, k|v))", (long)L * ld, (long)L * 64, 64},
  };
  for (auto& c : cases) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(gather_kernel, dim3(nseq * heads), dim3(768), 2 * 57344, 0, bufs[i % 3], c.unit_stride, c.head_stride, c.row_stride, heads, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-62s %.1f us per launch, %.2f TB/s\n", c.name, ms / 12 * 1000.f, (double)nseq * heads * 2 * L * 128 / (ms / 12 * 1e-3) / 1e12);
    }
  }
  return 0;
}
