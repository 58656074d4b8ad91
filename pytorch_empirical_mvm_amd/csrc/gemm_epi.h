// gemm_epi.h -- tile rasterisation and the fused epilogue shared by the GEMM kernels of libvmvm.so (gemm.hip, gemm_pp.hip)
#pragma once
#include "common.h"

namespace {

// Tile rasterisation inside one XCD's run of logical ids: walk GM consecutive M-panels for each N-tile before moving to the
// next N-tile, so the ~32-64 tiles resident on an XCD at any time share GM activation panels and (32..64)/GM weight tiles --
// a working set that fits the 4 MiB L2 (PMC before: the fc1 GEMM fetched 1.17 GB per launch, almost all of it the 4.7 MB
// weight matrix re-streamed through L2 once per M-panel).
__device__ __forceinline__ void raster(int tile, int nbm, int nbn, int GM, int& tm, int& tn) {
  const int per_group = GM * nbn;
  const int grp = tile / per_group;
  const int first_m = grp * GM;
  const int gsize = (nbm - first_m < GM) ? (nbm - first_m) : GM;
  const int in = tile - grp * per_group;
  tm = first_m + in % gsize;
  tn = in / gsize;
}

// where a split problem's units leave their bias-gradient partials: [S][M] floats BEHIND the S slabs, when the workspace holds them
// (host and kernels evaluate this same predicate; nullptr: one f32 atomic per (unit, m) as before)
__host__ __device__ inline float* colsum_parts(const vmvm_gemm_desc& d, int S) {
  if (S <= 1 || !d.workspace || !d.colsum) return nullptr;
  const size_t slabs = (size_t)S * d.M * d.N * sizeof(float), need = slabs + (size_t)S * d.M * sizeof(float);
  return (size_t)d.workspace_bytes >= need ? reinterpret_cast<float*>(reinterpret_cast<char*>(d.workspace) + slabs) : nullptr;
}
// ---- fused epilogue for 4 consecutive output columns (n..n+3) of row m (see include/vmvm.h for the order) ----------
struct EpiCtx { bool has_drop; uint32_t thr; float keep_scale; int S, slice, M, N; };
__device__ __forceinline__ void epi_store(const vmvm_gemm_desc& p, const EpiCtx& e_, float (&v)[4], int m, long dst, int n, float rs) {
  const bool has_drop = e_.has_drop; const uint32_t thr = e_.thr; const float keep_scale = e_.keep_scale;
  const int S = e_.S, slice = e_.slice, M = e_.M, N = e_.N;
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    const float bs = p.scale_bias_only ? rs : 1.0f;   // DropPath producer form: A rows already carry the scale
    v[0] += b.x * bs; v[1] += b.y * bs; v[2] += b.z * bs; v[3] += b.w * bs;
  }
  if (n < p.col_scale_n) {                            // q = (x Wq^T + bq) * scale  (video_swin.py:152)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.col_scale;
  }
  if (p.act == 1) {
    if (p.C2) {
      uint2 pre = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(p.C2) + (size_t)m * p.ldc2 + n) = pre;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = gelu_f(v[e]);
  } else if (p.act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if (p.act == 3 || p.act == 4) {
    const uint2 a2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(p.aux) + (size_t)m * p.ldaux + n);
    float u[4] = {__uint_as_float(a2.x << 16), __uint_as_float(a2.x & 0xffff0000u),
                  __uint_as_float(a2.y << 16), __uint_as_float(a2.y & 0xffff0000u)};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= (p.act == 3) ? gelu_grad_f(u[e]) : (u[e] > 0.f ? 1.f : 0.f);
  }
  if (p.row_scale && !p.scale_bias_only) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= rs;
  }
  if (has_drop) {
    const uint64_t e4 = ((uint64_t)m * (uint64_t)N + (uint64_t)n) >> 2;
    const uint4 bits = dropout_bits(p.seed, p.offset, e4);
    v[0] = bits.x < thr ? 0.f : v[0] * keep_scale;
    v[1] = bits.y < thr ? 0.f : v[1] * keep_scale;
    v[2] = bits.z < thr ? 0.f : v[2] * keep_scale;
    v[3] = bits.w < thr ? 0.f : v[3] * keep_scale;
  }
  if (p.resid) {
    const uint2 r2 = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(p.resid) + (size_t)dst * p.ldr + n);
    v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
    v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
  }
  if (S > 1) {
    if (p.workspace) {                               // split-K partial slab [slice][M][N], summed by splitk_reduce_kernel
      float* c = reinterpret_cast<float*>(p.workspace) + ((size_t)slice * M + dst) * N + n;
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
    } else {                                         // no workspace: f32 atomics into the accumulator
      float* c = reinterpret_cast<float*>(p.C) + (size_t)dst * p.ldc + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(c + e, v[e]);
    }
  } else if (p.out_fp32) {
    float* c = reinterpret_cast<float*>(p.C) + (size_t)dst * p.ldc + n;
    if (p.accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(c);
      v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
    }
    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(p.C) + (size_t)dst * p.ldc + n) =
        make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
  }
}

// ---- 8-wide epilogue: the persistent kernel permutes the N index inside each 32-column block (operand rows / transposing-
// read pieces are free to permute) so that a lane's accumulators of an MFMA tile PAIR are 8 CONSECUTIVE output columns:
// one 16-byte bf16 store (two for f32) instead of two 8-byte ones, and 16/32-byte bias / residual / aux loads.  The bf16
// epilogue was store-ISSUE bound: f32 output (2x the bytes, same instruction count) cost only +15% on HBM-bound shapes.
// Epilogue feature mask: the persistent kernel is instantiated per mask so that each instantiation carries only the code of the
// features its problems use.  (One generic epilogue inlined at 8 call sites is ~9000 instructions; streaming that through the
// instruction cache every tile cost 10-25% on the short-K shapes of this model.)  EF_ALL = everything, any descriptor.
enum : int { EF_BIAS = 1, EF_COLSCALE = 2, EF_ACT1 = 4, EF_ACT24 = 8, EF_ACT3 = 16, EF_RS = 32, EF_DROP = 64, EF_RESID = 128,
             EF_SPLIT = 256, EF_F32 = 512, EF_MAP = 1024, EF_EDGE4 = 2048, EF_ALL = 4095,
             EF_COLSUM = 4096 /* fused column sum of the m-major A operand (not part of EF_ALL: only the wgrad build carries it) */,
             EF_ARGMAX = 8192 /* act 5: (row maximum, column) pairs per 64-column group instead of the outputs (dVAE tokenizer's last conv) */,
             EF_CODE8 = 16384 /* gemm_pp.h builds of aux_code8 (8-bit GELU' codes move through 64-byte LDS rows); the 128x128 kernels test the descriptor */ };
template <int F, bool F16 = false>
__device__ __forceinline__ void epi_store8(const vmvm_gemm_desc& p, const EpiCtx& e_, float (&v)[8], int m, long dst, int n, float rs, int nvalid,
                                           const float (&bz)[8], const uint4& auxv, const uint4& resv) {
  // bz / auxv / resv: bias, saved activation and residual of this fragment, requested by the caller BEFORE the tile's first
  // store (loads cannot be hoisted over stores by the compiler: the pointers may alias)
  if ((F & EF_EDGE4) && nvalid < 8) {                    // N % 8 == 4 edge: fall back to the 4-wide path
    float a[4] = {v[0], v[1], v[2], v[3]};
    epi_store(p, e_, a, m, dst, n, rs);
    return;
  }
  if ((F & EF_BIAS) && p.bias) {
    const float bs = ((F & EF_RS) && p.scale_bias_only) ? rs : 1.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(bz[e], bs, v[e]);
  }
  if ((F & EF_COLSCALE) && n < p.col_scale_n) {          // col_scale_n is a multiple of 8 for every caller (C of qkv)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.col_scale;
  }
  if ((F & EF_ACT1) && p.act == 1) {
    if (!F16 && p.C2 && p.aux_code8) {                      // saved for the backward: an 8-bit code of GELU'(v) instead of v
      float gq[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2 y, g;
        gelu_and_code2(f32x2{v[e], v[e + 1]}, y, g);
        v[e] = y[0]; v[e + 1] = y[1]; gq[e] = g[0]; gq[e + 1] = g[1];
      }
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.C2) + (size_t)m * p.ldc2 + n) =
          make_uint2(gelu_code4(gq[0], gq[1], gq[2], gq[3]), gelu_code4(gq[4], gq[5], gq[6], gq[7]));
    } else {
      if (p.C2) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.C2) + (size_t)m * p.ldc2 + n) = pack8<F16>(v);
#pragma unroll
      for (int e = 0; e < 8; e += 2) { const f32x2 y = gelu2(f32x2{v[e], v[e + 1]}); v[e] = y[0]; v[e + 1] = y[1]; }
    }
  } else if ((F & EF_ACT24) && p.act == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if ((F & EF_ACT3) && p.act == 3) {
    if (!F16 && p.aux_code8) {                              // auxv.x / .y: the eight codes of this fragment
      float gq[8];
      gelu_decode4(auxv.x, gq); gelu_decode4(auxv.y, gq + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gq[e];
    } else {
      float u[8];
      unpack8<F16>(auxv, u);
#pragma unroll
      for (int e = 0; e < 8; e += 2) { const f32x2 gg = gelu_grad2(f32x2{u[e], u[e + 1]}); v[e] *= gg[0]; v[e + 1] *= gg[1]; }
    }
  } else if ((F & EF_ACT24) && p.act == 4) {
    float u[8];
    unpack8<F16>(auxv, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= (u[e] > 0.f ? 1.f : 0.f);
  }
  if ((F & EF_RS) && p.row_scale && !p.scale_bias_only) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= rs;
  }
  if ((F & EF_DROP) && e_.has_drop) {
    const uint64_t el = (uint64_t)m * (uint64_t)e_.N + (uint64_t)n;
    uint32_t bits[8];
    if ((F & EF_EDGE4) && (el & 7)) {                     // N % 8 != 0: this 8-run straddles two blocks
      const uint4 b0 = dropout_bits(p.seed, p.offset, el >> 2), b1 = dropout_bits(p.seed, p.offset, (el >> 2) + 1);
      bits[0] = b0.x; bits[1] = b0.y; bits[2] = b0.z; bits[3] = b0.w; bits[4] = b1.x; bits[5] = b1.y; bits[6] = b1.z; bits[7] = b1.w;
    } else {
      dropout_bits8(p.seed, p.offset, el >> 3, bits);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bits[e] < e_.thr ? 0.f : v[e] * e_.keep_scale;
  }
  if ((F & EF_RESID) && p.resid) {
    float rr[8];
    unpack8<F16>(resv, rr);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rr[e];
  }
  if ((F & EF_SPLIT) && e_.S > 1) {
    if (p.workspace) {
      float* c = reinterpret_cast<float*>(p.workspace) + ((size_t)e_.slice * e_.M + dst) * e_.N + n;
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      float* c = reinterpret_cast<float*>(p.C) + (size_t)dst * p.ldc + n;
#pragma unroll
      for (int e = 0; e < 8; ++e) atomicAdd(c + e, v[e]);
    }
  } else if ((F & EF_F32) && p.out_fp32) {
    float* c = reinterpret_cast<float*>(p.C) + (size_t)dst * p.ldc + n;
    if (p.accumulate) {
      const float4 o0 = *reinterpret_cast<const float4*>(c), o1 = *reinterpret_cast<const float4*>(c + 4);
      v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w; v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
    }
    *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
  } else {
    *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.C) + (size_t)dst * p.ldc + n) = pack8<F16>(v);
  }
}

// Math half of epi_store8 for 16-bit outputs: returns the packed result (and the packed pre-activation when act == 1) instead of
// storing -- the 256x256 kernel re-tiles them through LDS so that every store instruction writes whole 128-byte lines.
// (N % 8 == 0 only; bf16 / fp16 output, no split-K / f32 path.)
template <int F, bool F16 = false>
__device__ __forceinline__ void epi_math8(const vmvm_gemm_desc& p, const EpiCtx& e_, float (&v)[8], int m, int n, float rs,
                                          const float (&bz)[8], const uint4& auxv, const uint4& resv, uint4& out, uint4& pre) {
  if ((F & EF_BIAS) && p.bias) {
    const float bs = ((F & EF_RS) && p.scale_bias_only) ? rs : 1.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(bz[e], bs, v[e]);
  }
  if ((F & EF_COLSCALE) && n < p.col_scale_n) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= p.col_scale;
  }
  if ((F & EF_ACT1) && p.act == 1) {
    if constexpr ((F & EF_CODE8) != 0) {                   // pre.x / .y: the eight GELU' codes of this fragment
      float gq[8];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        f32x2 y, g;
        gelu_and_code2(f32x2{v[e], v[e + 1]}, y, g);
        v[e] = y[0]; v[e + 1] = y[1]; gq[e] = g[0]; gq[e + 1] = g[1];
      }
      pre = make_uint4(gelu_code4(gq[0], gq[1], gq[2], gq[3]), gelu_code4(gq[4], gq[5], gq[6], gq[7]), 0, 0);
    } else {
      pre = pack8<F16>(v);
#pragma unroll
      for (int e = 0; e < 8; e += 2) { const f32x2 y = gelu2(f32x2{v[e], v[e + 1]}); v[e] = y[0]; v[e + 1] = y[1]; }
    }
  } else if ((F & EF_ACT24) && p.act == 2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  } else if ((F & EF_ACT3) && p.act == 3) {
    if constexpr ((F & EF_CODE8) != 0) {
      float gq[8];
      gelu_decode4(auxv.x, gq); gelu_decode4(auxv.y, gq + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= gq[e];
    } else {
      float u[8];
      unpack8<F16>(auxv, u);
#pragma unroll
      for (int e = 0; e < 8; e += 2) { const f32x2 gg = gelu_grad2(f32x2{u[e], u[e + 1]}); v[e] *= gg[0]; v[e + 1] *= gg[1]; }
    }
  } else if ((F & EF_ACT24) && p.act == 4) {
    float u[8];
    unpack8<F16>(auxv, u);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= (u[e] > 0.f ? 1.f : 0.f);
  }
  if ((F & EF_RS) && p.row_scale && !p.scale_bias_only) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= rs;
  }
  if ((F & EF_DROP) && e_.has_drop) {
    uint32_t bits[8];
    dropout_bits8(p.seed, p.offset, ((uint64_t)m * (uint64_t)e_.N + (uint64_t)n) >> 3, bits);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bits[e] < e_.thr ? 0.f : v[e] * e_.keep_scale;
  }
  if ((F & EF_RESID) && p.resid) {
    float rr[8];
    unpack8<F16>(resv, rr);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += rr[e];
  }
  out = pack8<F16>(v);
}

}  // namespace
