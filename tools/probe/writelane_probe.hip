// writelane_probe.hip -- the four 64-bit lane masks of four compares (SDWA / VCC destinations, as in the fusion attention forward)
// written into lanes 0-7 of one VGPR with inline-asm v_writelane_b32, against the same record built with selects.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 writelane_probe.hip -o writelane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__device__ __forceinline__ uint32_t wl(uint32_t v, uint32_t s, const int lane) {
  if (MODE == 0) asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "i"(lane));
  if (MODE == 1) asm volatile("s_nop 4\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "i"(lane));
  if (MODE == 2) v = ((int)(threadIdx.x & 63) == lane) ? s : v;
  if (MODE == 3) asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "i"(lane) : "vcc");
  return v;
}
template <int MODE>
__global__ void probe(const uint32_t* in, uint32_t* out, uint32_t thr, int ntile) {
  const int lane = threadIdx.x & 63;
  for (int t = 0; t < ntile; ++t) {
    const uint32_t w = in[t * 64 + lane];
    uint32_t rec = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool d = (__builtin_amdgcn_alignbit(w, w, 8 * j) & 0xffffu) < thr;
      const uint64_t m = __builtin_amdgcn_ballot_w64(d);
      rec = wl<MODE>(rec, (uint32_t)m, 2 * j);
      rec = wl<MODE>(rec, (uint32_t)(m >> 32), 2 * j + 1);
    }
    if (lane < 8) out[t * 8 + lane] = rec;
  }
}
int main() {
  const int NT = 256;
  uint32_t h[NT * 64], exp[NT * 8], got[NT * 8];
  uint32_t s = 12345;
  for (int i = 0; i < NT * 64; ++i) { s = s * 1664525u + 1013904223u; h[i] = s; }
  const uint32_t thr = 6554;
  for (int t = 0; t < NT; ++t)
    for (int j = 0; j < 4; ++j) {
      uint64_t m = 0;
      for (int l = 0; l < 64; ++l) { const uint32_t w = h[t * 64 + l]; const uint32_t f = ((w >> (8 * j)) | (j ? (w << (32 - 8 * j)) : 0)) & 0xffffu; if (f < thr) m |= 1ull << l; }
      exp[t * 8 + 2 * j] = (uint32_t)m; exp[t * 8 + 2 * j + 1] = (uint32_t)(m >> 32);
    }
  uint32_t *din, *dout;
  CK(hipMalloc(&din, sizeof(h))); CK(hipMalloc(&dout, sizeof(got)));
  CK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
  for (int mode = 0; mode < 4; ++mode) {
    CK(hipMemset(dout, 0xff, sizeof(got)));
    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, din, dout, thr, NT);
    if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(1), dim3(64), 0, 0, din, dout, thr, NT);
    if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, din, dout, thr, NT);
    if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(1), dim3(64), 0, 0, din, dout, thr, NT);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost));
    int bad[8] = {0};
    for (int i = 0; i < NT * 8; ++i) if (got[i] != exp[i]) ++bad[i & 7];
    printf("mode %d: wrong dwords per record slot:", mode);
    for (int i = 0; i < 8; ++i) printf(" %d", bad[i]);
    printf("   (first record got %08x %08x %08x %08x ... expected %08x %08x %08x %08x)\n", got[0], got[1], got[2], got[3], exp[0], exp[1], exp[2], exp[3]);
  }
  return 0;
}
