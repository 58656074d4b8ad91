// dot2_probe.hip -- what v_dot2c_f32_bf16 (__builtin_amdgcn_fdot2_f32_bf16) computes on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const unsigned* b, const float* c, float* out, float* out2) {
  const int i = threadIdx.x;
  out[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), c[i], false);
  float acc = c[i];
  const bf16x2 one2 = __builtin_bit_cast(bf16x2, 0x3f803f80u);
  for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i] + 0u * r), one2, acc, false);
  out2[i] = acc;
}
static unsigned bf(float f) { unsigned u; memcpy(&u, &f, 4); return u >> 16; }
int main() {
  unsigned ha[64], hb[64]; float hc[64], ho[64], ho2[64];
  for (int i = 0; i < 64; ++i) { ha[i] = bf(1.0f + i) | (bf(0.5f * i) << 16); hb[i] = bf(2.0f) | (bf(-3.0f) << 16); hc[i] = 100.f; }
  unsigned *a, *b; float *c, *o, *o2;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&c, 256); hipMalloc(&o, 256); hipMalloc(&o2, 256);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice); hipMemcpy(c, hc, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, b, c, o, o2);
  hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(ho2, o2, 256, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) {
    const float x0 = 1.0f + i, x1 = 0.5f * i;
    printf("i=%d lo=%g hi=%g : dot2(a,(2,-3))+100 = %g (expect %g) ; 4x dot2(a,ones)+100 = %g (expect %g)\n", i, x0, x1, ho[i], 100 + 2 * x0 - 3 * x1, ho2[i], 100 + 4 * (x0 + x1));
  }
  return 0;
}
