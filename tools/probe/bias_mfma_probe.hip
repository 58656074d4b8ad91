// probe: D = I(16x16) * Bias + C on v_mfma_f32_16x16x16_bf16 with the packed-bf16 bias registers of the window-attention kernels
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4_;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ uint32_t pack_bf2(float lo, float hi) { typedef __attribute__((ext_vector_type(2))) float f2; return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, bf16x2)); }
__global__ void k(const float* bias /*[16 rows][16 cols]*/, float* out, int variant) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  s16x4_ a;
  for (int j = 0; j < 4; ++j) a[j] = (4 * g + j == r) ? (short)0x3f80 : (short)0;
  float b4[4];
  for (int j = 0; j < 4; ++j) b4[j] = bias[(4 * g + j) * 16 + r];
  const uint32_t w0 = pack_bf2(b4[0], b4[1]), w1 = pack_bf2(b4[2], b4[3]);
  f32x4 c = {1.f, 2.f, 3.f, 4.f};
  f32x4 d;
  if (variant == 0) d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(s16x4_, u32x2_{w0, w1}), c, 0, 0, 0);
  else { s16x4_ b = {(short)(w0 & 0xffff), (short)(w0 >> 16), (short)(w1 & 0xffff), (short)(w1 >> 16)}; d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
  if (variant >= 3) {
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 hv = {0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00};
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const s16x4_ bb = __builtin_bit_cast(s16x4_, u32x2_{w0, w1});
    if (variant == 3) {          // reverse order: score first, bias accumulates onto it
      d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bb, __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hv), __builtin_bit_cast(bf16x8, hv), z, 0, 0, 0), 0, 0, 0);
    } else if (variant == 4) {   // an independent 16x16x32 between producer and consumer
      f32x4 t = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bb, z, 0, 0, 0);
      f32x4 e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hv), __builtin_bit_cast(bf16x8, hv), c, 0, 0, 0);
      asm volatile("" : "+v"(e), "+v"(t));
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hv), __builtin_bit_cast(bf16x8, hv), t, 0, 0, 0);
      d[0] += 0.f * e[0];
    } else {                     // variant 5: explicit wait states
      f32x4 t = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bb, z, 0, 0, 0);
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(t));
      d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hv), __builtin_bit_cast(bf16x8, hv), t, 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) out[(4 * g + j) * 16 + r] = d[j] - 8.0f;
    return;
  }
  if (variant == 2) {            // dependent chain: the bias product is the C operand of a 16x16x32 score MFMA (A = B = 0.5 everywhere -> +8)
    typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 hv = {0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00, 0x3f00};
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, hv), __builtin_bit_cast(bf16x8, hv),
                                                __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(s16x4_, u32x2_{w0, w1}), z, 0, 0, 0), 0, 0, 0);
    for (int j = 0; j < 4; ++j) out[(4 * g + j) * 16 + r] = d[j] - 8.0f;
    return;
  }
  for (int j = 0; j < 4; ++j) out[(4 * g + j) * 16 + r] = d[j] - c[j];
}
int main() {
  float h[256], o[256];
  for (int i = 0; i < 256; ++i) h[i] = (i % 7 == 0) ? -1.0e30f : (float)((i * 37 % 64) - 32) / 8.0f;
  float *db, *dout;
  hipMalloc(&db, 1024); hipMalloc(&dout, 1024);
  hipMemcpy(db, h, 1024, hipMemcpyHostToDevice);
  for (int v = 0; v < 6; ++v) {
    k<<<1, 64>>>(db, dout, v);
    hipMemcpy(o, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) { float want = h[i]; if (!(fabsf(o[i] - want) <= 1e-2f * fabsf(want) + 1e-6f)) { if (bad < 5) printf("  v%d [%d] got %g want %g\n", v, i, o[i], want); ++bad; } }
    printf("variant %d: %d mismatches\n", v, bad);
  }
  return 0;
}
