// common.h -- shared device helpers for the gfx950 kernels of libvmvm.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vmvm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short u16;

extern thread_local int g_vmvm_last_hip_error;

#define VMVM_CHECK_LAUNCH()                                  \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) { g_vmvm_last_hip_error = (int)e_; return VMVM_EHIP; } \
  } while (0)

__device__ __forceinline__ float bf2f(u16 v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ u16 f2bf(float f) {
  __bf16 b = (__bf16)f;                      // RNE; v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(u16, b);
}
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {          // one v_cvt_pk_bf16_f32 (RNE)
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, bf16x2));
}
__device__ __forceinline__ uint32_t pack_bf2v(f32x2 v) {          // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void unpack_bf8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack_bf8(const float* f) {
  return make_uint4(pack_bf2(f[0], f[1]), pack_bf2(f[2], f[3]), pack_bf2(f[4], f[5]), pack_bf2(f[6], f[7]));
}

// fp16 twins of pack_bf8 / unpack_bf8 (the frozen dVAE tokenizer runs in fp16 like the reference's GPU path)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_{lo, hi}, f16x2));
}
__device__ __forceinline__ uint4 pack_h8(const float* f) {
  return make_uint4(pack_h2(f[0], f[1]), pack_h2(f[2], f[3]), pack_h2(f[4], f[5]), pack_h2(f[6], f[7]));
}
__device__ __forceinline__ void unpack_h8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 h = __builtin_bit_cast(f16x2, w[i]);
    f[2 * i] = (float)h[0]; f[2 * i + 1] = (float)h[1];
  }
}
template <bool F16> __device__ __forceinline__ uint4 pack8(const float* f) { return F16 ? pack_h8(f) : pack_bf8(f); }
template <bool F16> __device__ __forceinline__ void unpack8(const uint4& v, float* f) { if (F16) unpack_h8(v, f); else unpack_bf8(v, f); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte streaming accesses: a tensor that is read once / written once per launch (LayerNorm rows, ...) should not displace the GEMM
// operand panels that live in L2 / the Infinity Cache between launches.  Round 6, measured in the step with every LayerNorm kernel on
// these (tools/scratch/build_ln_nt_variant.sh): 106.02 -> 105.56 ms (three interleaved pairs); alone the kernels are unchanged when
// their operands are cold and slower when a micro-benchmark keeps them resident -- which a training step never does.
typedef unsigned v4u_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt16(const void* p) {
  const v4u_nt v = __builtin_nontemporal_load(reinterpret_cast<const v4u_nt*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_nt16(void* p, const uint4 x) {
  const v4u_nt v = {x.x, x.y, x.z, x.w};
  __builtin_nontemporal_store(v, reinterpret_cast<v4u_nt*>(p));
}

// erf-GELU (video_swin.py:66 nn.GELU ; HF hidden_act="gelu") and its derivative.  These run in GEMM epilogues on every
// output element, where VALU time is not hidden by MFMA work, so erf is an odd minimax polynomial on |z| <= 3 (clamped to
// +-1 beyond; max abs error 9e-5, i.e. <= 2e-4 absolute on GELU -- below the bf16 rounding of the stored activation) with
// no transcendental: 8 FMAs instead of ocml's branchy erff or a rcp+exp formulation.
__device__ __forceinline__ float erf_poly(float z) {
  const float t = z * z;
  float p = -4.0375596e-07f;
  p = fmaf(p, t, 1.7119051e-05f);
  p = fmaf(p, t, -3.1437373e-04f);
  p = fmaf(p, t, 3.3201380e-03f);
  p = fmaf(p, t, -2.2705898e-02f);
  p = fmaf(p, t, 1.0779675e-01f);
  p = fmaf(p, t, -3.7335253e-01f);
  p = fmaf(p, t, 1.1279515e+00f);
  const float e = p * z;
  return fabsf(z) > 3.0f ? copysignf(1.0f, z) : fminf(fmaxf(e, -1.0f), 1.0f);
}
// two elements at a time: the polynomial is a pure FMA chain, so it runs on the packed-f32 pipe (v_pk_fma_f32) at half the
// instruction count; the |z| > 3 select becomes a clamp of the argument (v_med3_f32) plus a clamp of the result
__device__ __forceinline__ f32x2 erf_poly2(f32x2 z) {
  z = f32x2{__builtin_amdgcn_fmed3f(z[0], -3.0f, 3.0f), __builtin_amdgcn_fmed3f(z[1], -3.0f, 3.0f)};
  const f32x2 t = z * z;
  f32x2 p = f32x2{-4.0375596e-07f, -4.0375596e-07f};
  p = __builtin_elementwise_fma(p, t, f32x2{1.7119051e-05f, 1.7119051e-05f});
  p = __builtin_elementwise_fma(p, t, f32x2{-3.1437373e-04f, -3.1437373e-04f});
  p = __builtin_elementwise_fma(p, t, f32x2{3.3201380e-03f, 3.3201380e-03f});
  p = __builtin_elementwise_fma(p, t, f32x2{-2.2705898e-02f, -2.2705898e-02f});
  p = __builtin_elementwise_fma(p, t, f32x2{1.0779675e-01f, 1.0779675e-01f});
  p = __builtin_elementwise_fma(p, t, f32x2{-3.7335253e-01f, -3.7335253e-01f});
  p = __builtin_elementwise_fma(p, t, f32x2{1.1279515e+00f, 1.1279515e+00f});
  const f32x2 e = p * z;
  return f32x2{__builtin_amdgcn_fmed3f(e[0], -1.0f, 1.0f), __builtin_amdgcn_fmed3f(e[1], -1.0f, 1.0f)};
}
__device__ __forceinline__ f32x2 gelu_sigmoid2(f32x2 x, f32x2 t);
__device__ __forceinline__ f32x2 gelu2(f32x2 x);
// 8-bit code of GELU' (vmvm_gemm_desc.aux_code8): g in [-0.129, 1.129] -> round((g + 0.13) * 255 / 1.26), and back
constexpr float GC8_LO = -0.13f, GC8_STEP = 1.26f / 255.0f, GC8_INV = 255.0f / 1.26f;
// GELU and the (unrounded) code of GELU' together -- round 4: ONE shared exponential.  The normal cdf as a logistic of an odd cubic,
//   Phi(x) ~ s = 1 / (1 + 2^(x (A2 + B2 x^2))),     A2 = -a log2 e, B2 = -b log2 e,  (a, b) = (1.59982729, 0.0699463)
// (a minimax fit of GELU and GELU' against erf over |x| <= 8, tools/scratch/fit_gelu_logistic.py: max |GELU error| 3.3e-4 -- the erf
// polynomial it replaces: 4.0e-4 --, max |GELU' error| 6.7e-4, an eighth of a code step; torch's tanh form is this family with other
// constants).  Then GELU = x s and GELU' = s + x s (1 - s) u'(x) with u' = a + 3 b x^2, so the second transcendental pair of the
// erf form (its pdf term) is gone: per pair of elements 9 packed instructions + 2 v_exp_f32 + 2 v_rcp_f32 against 19 + 2 before --
// the GELU classes of the GEMM are bound by the ISSUE of their epilogue instructions, not by how they are arranged.
// Saturation is benign: 2^(...) -> inf gives s = 0, y = 0, g = 0; -> 0 gives s = 1, y = x, g = 1 (no inf * 0 anywhere).
constexpr float GL_A = 1.59982729f, GL_B = 0.0699463f;
__device__ __forceinline__ f32x2 gelu_sigmoid2(f32x2 x, f32x2 t) {     // s = Phi(x), t = x * x
  constexpr float A2 = -GL_A * 1.4426950408889634f, B2 = -GL_B * 1.4426950408889634f;
  const f32x2 u = x * __builtin_elementwise_fma(t, f32x2{B2, B2}, f32x2{A2, A2});
  const f32x2 d = f32x2{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + f32x2{1.0f, 1.0f};
  return f32x2{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) { return x * gelu_sigmoid2(x, x * x); }
__device__ __forceinline__ void gelu_and_code2(f32x2 x, f32x2& y, f32x2& cf) {
  const f32x2 t = x * x;
  const f32x2 s = gelu_sigmoid2(x, t);
  y = x * s;
  const f32x2 du = __builtin_elementwise_fma(t, f32x2{3.0f * GL_B, 3.0f * GL_B}, f32x2{GL_A, GL_A});
  const f32x2 xw = __builtin_elementwise_fma(-y, s, y);               // x s (1 - s)
  const f32x2 g = __builtin_elementwise_fma(xw, du, s);
  cf = __builtin_elementwise_fma(g, f32x2{GC8_INV, GC8_INV}, f32x2{-GC8_LO * GC8_INV, -GC8_LO * GC8_INV});
}
__device__ __forceinline__ uint32_t gelu_code4(float c0, float c1, float c2, float c3) {      // round + saturate to bytes 0..3
  uint32_t w = 0;
  w = __builtin_amdgcn_cvt_pk_u8_f32(c0, 0, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c1, 1, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c2, 2, w);
  w = __builtin_amdgcn_cvt_pk_u8_f32(c3, 3, w);
  return w;
}
__device__ __forceinline__ void gelu_decode4(uint32_t w, float* g) {
  g[0] = __builtin_fmaf((float)((w >> 0) & 0xffu), GC8_STEP, GC8_LO);
  g[1] = __builtin_fmaf((float)((w >> 8) & 0xffu), GC8_STEP, GC8_LO);
  g[2] = __builtin_fmaf((float)((w >> 16) & 0xffu), GC8_STEP, GC8_LO);
  g[3] = __builtin_fmaf((float)((w >> 24) & 0xffu), GC8_STEP, GC8_LO);
}
__device__ __forceinline__ f32x2 gelu_grad2(f32x2 x) {
  const f32x2 a = x * x * f32x2{-0.72134752044448170f, -0.72134752044448170f};           // -0.5 * log2(e) * x^2
  const f32x2 pdf = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} * f32x2{0.39894228040143268f, 0.39894228040143268f};
  const f32x2 cdf = __builtin_elementwise_fma(erf_poly2(x * f32x2{0.70710678118654752f, 0.70710678118654752f}), f32x2{0.5f, 0.5f}, f32x2{0.5f, 0.5f});
  return __builtin_elementwise_fma(x, pdf, cdf);
}
__device__ __forceinline__ float gelu_f(float x) {          // the same logistic form as gelu2 (every GELU forward of the library agrees)
  constexpr float A2 = -1.59982729f * 1.4426950408889634f, B2 = -0.0699463f * 1.4426950408889634f;
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * fmaf(x * x, B2, A2)));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return fmaf(x, pdf, 0.5f * (1.0f + erf_poly(x * 0.70710678118654752f)));
}

// Philox4x32-7 counter RNG: 4 x 32 random bits per (counter, key)
__device__ __forceinline__ uint4 philox4x32_7(uint4 c, uint2 k) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_lo_u32 + v_mul_hi_u32 pair: both forms run at a
    // quarter of the VALU rate, and the multiplies are a third of the softmax-side VALU time of the fusion attention kernels
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += W0; k.y += W1;
  }
  return c;
}
// Dropout randomness: one Philox4x32-7 evaluation per block of EIGHT consecutive elements, 16 random bits per element (compared
// as the top half of a 32-bit word against the 32-bit threshold, i.e. p is quantised to 1/65536).  dropout_bits8 serves the
// 16-byte (8-element) consumers, dropout_bits the 4-element ones: element e of the tensor gets the same bits either way.
__device__ __forceinline__ uint4 dropout_block8(uint64_t seed, uint64_t offset, uint64_t e8) {
  const uint64_t c = offset + e8;
  return philox4x32_7(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0x5eedu, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ void dropout_bits8(uint64_t seed, uint64_t offset, uint64_t e8, uint32_t (&bits)[8]) {
  const uint4 b = dropout_block8(seed, offset, e8);
  bits[0] = b.x << 16; bits[1] = b.x & 0xffff0000u; bits[2] = b.y << 16; bits[3] = b.y & 0xffff0000u;
  bits[4] = b.z << 16; bits[5] = b.z & 0xffff0000u; bits[6] = b.w << 16; bits[7] = b.w & 0xffff0000u;
}
// the 4 consecutive elements starting at element index e4*4 (= half of block e4 >> 1)
__device__ __forceinline__ uint4 dropout_bits(uint64_t seed, uint64_t offset, uint64_t e4) {
  const uint4 b = dropout_block8(seed, offset, e4 >> 1);
  const uint32_t lo = (e4 & 1) ? b.z : b.x, hi = (e4 & 1) ? b.w : b.y;
  return make_uint4(lo << 16, lo & 0xffff0000u, hi << 16, hi & 0xffff0000u);
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(fminf(fmaxf(p, 0.f), 1.f) * 4294967295.0f); }

// CUs a persistent / resident grid may use: the whole chip minus `reserve` (vmvm_gemm_desc.reserve_cus), kept a multiple of 8 so the
// XCD-aware rasterisations (workgroup b runs on XCD b % 8) still see equal shares per XCD.
static inline int vmvm_usable_cus(int reserve) {
  int r = reserve < 0 ? 0 : reserve > 128 ? 128 : reserve;
  r = (r + 7) & ~7;
  return 256 - r;
}
