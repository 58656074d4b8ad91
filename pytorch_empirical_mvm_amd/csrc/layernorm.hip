// layernorm.hip -- gather-LayerNorm forward/backward for gfx950 (HBM-bound, one wave per row).
//
// One 64-lane wave owns one output row; each lane holds its 16-byte chunks (8 bf16) of the row in
// registers (NCH chunks per lane), so the row is read from HBM exactly once, statistics are f32
// (two-pass in registers), and the window-shift/partition gather (Video-Swin norm1) or the 2x2
// patch-merging gather is pure address arithmetic on the read side (`src` map) -- no rolled,
// padded or concatenated tensor is ever materialised.
// Backward: a fixed grid of waves walks rows grid-stride, keeps dgamma/dbeta partials in registers,
// reduces them across the block in LDS and issues one f32 atomic per column per block.
#include "common.h"
#include <vmvm_probe_hooks.h>
#include <cstdlib>

namespace {

// 8 consecutive input elements of a row: bf16 (one 16-byte load) or f32 (two)
template <bool XF32>
__device__ __forceinline__ void load_x8(const void* X, size_t elem_off, float* f) {
  if (XF32) {
    const float* p = reinterpret_cast<const float*>(X) + elem_off;
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    unpack_bf8(ld_nt16(reinterpret_cast<const u16*>(X) + elem_off), f);
  }
}

template <int NCH, bool XF32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const vmvm_ln_fwd_desc p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long m = (long)blockIdx.x * 4 + wave;
  if (m >= p.M) return;
  const int C = p.C, nch = C >> 3, cseg = C / p.nseg;
  u16* Y = reinterpret_cast<u16*>(p.Y) + (size_t)m * p.ldy;
  long b = 0, ml = m;
  if (p.src) { b = m / p.rows_out_per_batch; ml = m - b * p.rows_out_per_batch; }

  float x[NCH][8];
  float s = 0.f;
  bool any_valid = (p.src == nullptr);
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[i][e] = 0.f;
    if (c < nch) {
      const int col = c * 8;
      long srow = m; int within = col;
      if (p.src) {
        const int seg = col / cseg; within = col - seg * cseg;
        const int sr = p.src[ml * p.nseg + seg];
        srow = sr < 0 ? -1 : (long)sr + b * p.rows_in_per_batch;
      }
      if (srow >= 0) {
        any_valid = true;
        load_x8<XF32>(p.X, (size_t)srow * p.ldx + within, x[i]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[i][e];
    }
  }
  if (p.src && p.pad_mode == 0) {
    // window map: the row is either fully valid or a pad slot (zero OUTPUT row)
    const bool valid = __any(any_valid);
    if (!valid) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) *reinterpret_cast<uint4*>(Y + c * 8) = make_uint4(0, 0, 0, 0);
      }
      if (lane == 0) { p.mean[m] = 0.f; p.rstd[m] = 0.f; }
      return;
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = x[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + p.eps);
  if (lane == 0) { p.mean[m] = mean; p.rstd[m] = rstd; }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      const int col = c * 8;
      const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.beta + col), b1 = *reinterpret_cast<const float4*>(p.beta + col + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (x[i][e] - mean) * rstd * gg[e] + bb[e];
      st_nt16(Y + col, pack_bf8(o));
    }
  }
}

template <int NCH, bool XF32>
__global__ __launch_bounds__(256, (NCH == 2 ? 4 : 1)) void ln_bwd_kernel(const vmvm_ln_bwd_desc p) {   // NCH = 2: 131 VGPRs without the hint, 4 waves per SIMD need <= 128
  extern __shared__ __attribute__((aligned(16))) float red[];   // [2][C] partial dgamma/dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = p.C, nch = C >> 3, cseg = C / p.nseg;
  const u16* dY = reinterpret_cast<const u16*>(p.dY);
  u16* dX = reinterpret_cast<u16*>(p.dX);
  const u16* ADD = reinterpret_cast<const u16*>(p.dX_add);
  u16* dX2 = reinterpret_cast<u16*>(p.dX2);
  const bool has_drop = dX2 != nullptr && p.dropout_p > 0.f;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = has_drop ? 1.f / (1.f - p.dropout_p) : 1.f;

  float dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) { dg[i][e] = 0.f; db[i][e] = 0.f; }

  const long stride = (long)gridDim.x * 4;
  for (long m = (long)blockIdx.x * 4 + wave; m < p.M; m += stride) {
    const float mean = p.mean[m], rstd = p.rstd[m];
    long b = 0, ml = m;
    if (p.src) { b = m / p.rows_out_per_batch; ml = m - b * p.rows_out_per_batch; }
    if (p.src && p.pad_mode == 0 && p.src[ml] < 0) continue;     // pad slot: constant zero output
    // bf16 inputs stay PACKED between the two passes over the row (x-hat and gamma * dy are recomputed for the output: the kernel is
    // memory-latency bound at 19 % VALU, and the 16 registers saved at NCH = 2 are the difference between 3 and 4 waves per SIMD)
    float xh[XF32 ? NCH : 1][8], gdy[XF32 ? NCH : 1][8];
    uint4 xraw[NCH], dyraw[NCH];
    long srow[NCH]; int within[NCH];
    uint4 addv[NCH];                                      // residual-path gradient, requested WITH x / dY (not after the row reduction)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      srow[i] = -1; within[i] = 0;
      xraw[i] = make_uint4(0, 0, 0, 0); dyraw[i] = make_uint4(0, 0, 0, 0);
      if (XF32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { xh[XF32 ? i : 0][e] = 0.f; gdy[XF32 ? i : 0][e] = 0.f; }
      }
      if (c < nch) {
        const int col = c * 8;
        srow[i] = m; within[i] = col;
        if (p.src) {
          const int seg = col / cseg; within[i] = col - seg * cseg;
          const int sr = p.src[ml * p.nseg + seg];
          srow[i] = sr < 0 ? -1 : (long)sr + b * p.rows_in_per_batch;
        }
        float xv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dyv[8];
        if (XF32) {
          if (srow[i] >= 0) load_x8<XF32>(p.X, (size_t)srow[i] * p.ldx + within[i], xv);
        } else {
          if (srow[i] >= 0) xraw[i] = ld_nt16(reinterpret_cast<const u16*>(p.X) + (size_t)srow[i] * p.ldx + within[i]);
          unpack_bf8(xraw[i], xv);
        }
        addv[i] = make_uint4(0, 0, 0, 0);
        if (ADD && srow[i] >= 0) addv[i] = ld_nt16(ADD + (p.add_by_out ? (size_t)m * p.ldadd + col : (size_t)srow[i] * p.ldadd + within[i]));
        dyraw[i] = ld_nt16(dY + (size_t)m * p.lddy + col);
        unpack_bf8(dyraw[i], dyv);
        const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xhe = (xv[e] - mean) * rstd, gde = dyv[e] * gg[e];
          if (XF32) { xh[XF32 ? i : 0][e] = xhe; gdy[XF32 ? i : 0][e] = gde; }
          dg[i][e] += dyv[e] * xhe;
          db[i][e] += dyv[e];
          s1 += gde;
          s2 += gde * xhe;
        }
      }
    }
    s1 = wave_sum(s1) / (float)C;
    s2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = lane + i * 64;
      if (c < nch && srow[i] >= 0) {
        float o[8];
        if (XF32) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rstd * (gdy[XF32 ? i : 0][e] - s1 - xh[XF32 ? i : 0][e] * s2);
        } else {
          float xv[8], dyv[8];
          unpack_bf8(xraw[i], xv);
          unpack_bf8(dyraw[i], dyv);
          const int col = c * 8;
          const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
          const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = rstd * (dyv[e] * gg[e] - s1 - (xv[e] - mean) * rstd * s2);
        }
        if (ADD) {
          float a[8];
          unpack_bf8(addv[i], a);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += a[e];
        }
        long drow = srow[i];
        if (p.dx_map) { const long bb = m / p.dx_map_len; drow = (long)p.dx_map[m - bb * p.dx_map_len] + bb * p.dx_map_len; }     // (identity walk: srow = m)
        st_nt16(dX + (size_t)drow * p.lddx + within[i], pack_bf8(o));
        if (dX2) {
          if (has_drop) {
            uint32_t bits[8];
            dropout_bits8(p.seed, p.offset, ((uint64_t)m * (uint64_t)C + (uint64_t)(c * 8)) >> 3, bits);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = bits[e] < thr ? 0.f : o[e] * keep_scale;
          }
          st_nt16(dX2 + (size_t)m * p.lddx2 + c * 8, pack_bf8(o));
        }
      }
    }
  }
  // block reduction of the dgamma/dbeta partials: one LDS slab per wave (plain stores, no LDS atomics), then either one partial
  // row per workgroup in the scratch buffer (summed by ln_colreduce_kernel) or, without scratch, one global atomic per column
  float* redw = red + wave * 2 * C;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = lane + i * 64;
    if (c < nch) {
      *reinterpret_cast<float4*>(redw + c * 8) = make_float4(dg[i][0], dg[i][1], dg[i][2], dg[i][3]);
      *reinterpret_cast<float4*>(redw + c * 8 + 4) = make_float4(dg[i][4], dg[i][5], dg[i][6], dg[i][7]);
      *reinterpret_cast<float4*>(redw + C + c * 8) = make_float4(db[i][0], db[i][1], db[i][2], db[i][3]);
      *reinterpret_cast<float4*>(redw + C + c * 8 + 4) = make_float4(db[i][4], db[i][5], db[i][6], db[i][7]);
    }
  }
  __syncthreads();
  float* ws = reinterpret_cast<float*>(p.workspace);
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const float v = red[i] + red[2 * C + i] + red[4 * C + i] + red[6 * C + i];
    if (ws) ws[(size_t)blockIdx.x * 2 * C + i] = v;
    else atomicAdd(i < C ? p.dgamma + i : p.dbeta + (i - C), v);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Packed variants for narrow rows (C <= 256: Video-Swin stages 1-2 and the patch embedding).  With one row per wave only C/8 of
// the 64 lanes hold data (16 of 64 at C = 128), so the kernels above run at a quarter of the HBM rate there.  Here a wave owns
// 64/LPR rows: lane = sub * LPR + cl, row = base + sub, chunk = cl; row statistics are segmented xor-shuffle sums over the LPR
// lanes of a row.  All row-level exits become per-lane predicates (every lane must reach the shuffles).
// ---------------------------------------------------------------------------------------------------------------------------
template <int LPR>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int LPR, bool XF32>
__global__ __launch_bounds__(256) void ln_fwd_pk_kernel(const vmvm_ln_fwd_desc p) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, cl = lane % LPR;
  const long m = ((long)blockIdx.x * 4 + wave) * RPW + sub;
  const int C = p.C, nch = C >> 3, cseg = C / p.nseg;
  const bool rowok = m < p.M;
  const bool act = rowok && cl < nch;
  const long mm = rowok ? m : 0;
  u16* Y = reinterpret_cast<u16*>(p.Y) + (size_t)mm * p.ldy;
  long b = 0, ml = mm;
  if (p.src) { b = mm / p.rows_out_per_batch; ml = mm - b * p.rows_out_per_batch; }
  const int col = cl * 8;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = 0.f;
  float s = 0.f, nvalid = (p.src == nullptr) ? 1.f : 0.f;
  if (act) {
    long srow = mm; int within = col;
    if (p.src) {
      const int seg = col / cseg; within = col - seg * cseg;
      const int sr = p.src[ml * p.nseg + seg];
      srow = sr < 0 ? -1 : (long)sr + b * p.rows_in_per_batch;
    }
    if (srow >= 0) {
      nvalid = 1.f;
      load_x8<XF32>(p.X, (size_t)srow * p.ldx + within, x);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s += x[e];
  }
  // window map: the row is either fully valid or a pad slot (zero OUTPUT row)
  const bool padrow = p.src && p.pad_mode == 0 && seg_sum<LPR>(nvalid) == 0.f;
  const float mean = seg_sum<LPR>(s) / (float)C;
  float q = 0.f;
  if (act) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = x[e] - mean; q += d * d; }
  }
  const float rstd = rsqrtf(seg_sum<LPR>(q) / (float)C + p.eps);
  if (rowok && cl == 0) { p.mean[m] = padrow ? 0.f : mean; p.rstd[m] = padrow ? 0.f : rstd; }
  if (act) {
    if (padrow) { *reinterpret_cast<uint4*>(Y + col) = make_uint4(0, 0, 0, 0); return; }
    const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.beta + col), b1 = *reinterpret_cast<const float4*>(p.beta + col + 4);
    const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (x[e] - mean) * rstd * gg[e] + bb[e];
    st_nt16(Y + col, pack_bf8(o));
  }
}

template <int LPR, bool XF32>
__global__ __launch_bounds__(256) void ln_bwd_pk_kernel(const vmvm_ln_bwd_desc p) {
  constexpr int RPW = 64 / LPR;
  extern __shared__ __attribute__((aligned(16))) float red[];   // 4 wave slabs x [2][C] partial dgamma/dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, cl = lane % LPR;
  const int C = p.C, nch = C >> 3, cseg = C / p.nseg;
  const u16* dY = reinterpret_cast<const u16*>(p.dY);
  u16* dX = reinterpret_cast<u16*>(p.dX);
  const u16* ADD = reinterpret_cast<const u16*>(p.dX_add);
  u16* dX2 = reinterpret_cast<u16*>(p.dX2);
  const bool has_drop = dX2 != nullptr && p.dropout_p > 0.f;
  const uint32_t thr = dropout_threshold(p.dropout_p);
  const float keep_scale = has_drop ? 1.f / (1.f - p.dropout_p) : 1.f;
  const int col = cl * 8;
  float dg[8], db[8], gg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 8; ++e) { dg[e] = 0.f; db[e] = 0.f; }
  if (cl < nch) {
    const float4 g0 = *reinterpret_cast<const float4*>(p.gamma + col), g1 = *reinterpret_cast<const float4*>(p.gamma + col + 4);
    gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
  }
  const long stride = (long)gridDim.x * 4 * RPW;
  // Two iteration orders.  Output-major (the default): row m of dY, its source row looked up in `src` -- x, the residual gradient and dX
  // are then rows scattered by the map.  SOURCE-major (`inv` given, one segment): row n of x / dX_add / dX in order, its dY / mean / rstd
  // row looked up in `inv` -- ONE scattered stream instead of three (window maps of Video-Swin stage 1-2: 256-512-byte rows; measured
  // 336 -> 324 us / 208 -> 191 us, profiles/r04_ab_ln_bwd_source_major.txt).  A source row without an output row (inv < 0) is left to the caller.
  const bool by_src = p.inv != nullptr;
  const long Mloop = by_src ? (long)p.rows_in_total : (long)p.M;
  for (long mb = ((long)blockIdx.x * 4 + wave) * RPW; mb < Mloop; mb += stride) {       // wave-uniform loop: every lane reaches the shuffles
    const long m = mb + sub;
    bool rowok = m < Mloop;
    long mm = rowok ? m : 0;                              // the dY / mean / rstd row
    long nrow = -1;                                       // source-major: the x / dX_add / dX row
    if (by_src) {
      const int n = (int)mm;
      const int bb = n / p.rows_in_per_batch, nl = n - bb * p.rows_in_per_batch;
      const int w = p.inv[nl];                            // (requesting it one iteration ahead measured slower: 324 -> 346 us)
      nrow = n;
      rowok = rowok && w >= 0;
      mm = rowok ? (long)w + (long)bb * p.rows_out_per_batch : 0;
    }
    const float mean = p.mean[mm], rstd = p.rstd[mm];
    long b = 0, ml = mm;
    if (p.src && !by_src) { b = mm / p.rows_out_per_batch; ml = mm - b * p.rows_out_per_batch; }
    const bool padrow = p.src && !by_src && p.pad_mode == 0 && p.src[ml] < 0;     // pad slot: constant zero output, no gradient
    const bool act = rowok && !padrow && cl < nch;
    float xh[8], gdy[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { xh[e] = 0.f; gdy[e] = 0.f; }
    long srow = -1; int within = col;
    uint4 addv = make_uint4(0, 0, 0, 0);
    float s1 = 0.f, s2 = 0.f;
    if (act) {
      srow = mm;
      if (by_src) srow = nrow;
      else if (p.src) {
        const int seg = col / cseg; within = col - seg * cseg;
        const int sr = p.src[ml * p.nseg + seg];
        srow = sr < 0 ? -1 : (long)sr + b * p.rows_in_per_batch;
      }
      float xv[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dyv[8];
      if (srow >= 0) load_x8<XF32>(p.X, (size_t)srow * p.ldx + within, xv);
      if (ADD && srow >= 0) addv = ld_nt16(ADD + (p.add_by_out ? (size_t)mm * p.ldadd + col : (size_t)srow * p.ldadd + within));
      unpack_bf8(ld_nt16(dY + (size_t)mm * p.lddy + col), dyv);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[e] = (xv[e] - mean) * rstd;
        gdy[e] = dyv[e] * gg[e];
        dg[e] += dyv[e] * xh[e];
        db[e] += dyv[e];
        s1 += gdy[e];
        s2 += gdy[e] * xh[e];
      }
    }
    s1 = seg_sum<LPR>(s1) / (float)C;
    s2 = seg_sum<LPR>(s2) / (float)C;
    if (act && srow >= 0) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = rstd * (gdy[e] - s1 - xh[e] * s2);
      if (ADD) {
        float a[8];
        unpack_bf8(addv, a);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += a[e];
      }
      long drow = srow;
      if (p.dx_map) { const long bb = mm / p.dx_map_len; drow = (long)p.dx_map[mm - bb * p.dx_map_len] + bb * p.dx_map_len; }     // (identity walk: srow = mm)
      st_nt16(dX + (size_t)drow * p.lddx + within, pack_bf8(o));
      if (dX2) {
        if (has_drop) {
          uint32_t bits[8];
          dropout_bits8(p.seed, p.offset, ((uint64_t)mm * (uint64_t)C + (uint64_t)col) >> 3, bits);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = bits[e] < thr ? 0.f : o[e] * keep_scale;
        }
        st_nt16(dX2 + (size_t)mm * p.lddx2 + col, pack_bf8(o));
      }
    }
  }
  // the RPW row slots of a wave hold partials of the same columns: fold them, then the per-wave slab reduction of ln_bwd_kernel
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) { dg[e] += __shfl_xor(dg[e], o, 64); db[e] += __shfl_xor(db[e], o, 64); }
  }
  float* redw = red + wave * 2 * C;
  if (sub == 0 && cl < nch) {
    *reinterpret_cast<float4*>(redw + col) = make_float4(dg[0], dg[1], dg[2], dg[3]);
    *reinterpret_cast<float4*>(redw + col + 4) = make_float4(dg[4], dg[5], dg[6], dg[7]);
    *reinterpret_cast<float4*>(redw + C + col) = make_float4(db[0], db[1], db[2], db[3]);
    *reinterpret_cast<float4*>(redw + C + col + 4) = make_float4(db[4], db[5], db[6], db[7]);
  }
  __syncthreads();
  float* ws = reinterpret_cast<float*>(p.workspace);
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const float v = red[i] + red[2 * C + i] + red[4 * C + i] + red[6 * C + i];
    if (ws) ws[(size_t)blockIdx.x * 2 * C + i] = v;
    else atomicAdd(i < C ? p.dgamma + i : p.dbeta + (i - C), v);
  }
}

// dgamma[c] += sum_rows ws[row][c], dbeta[c] += sum_rows ws[row][C + c] : 32 columns x 32 row lanes per workgroup, ONE workgroup per
// column group (round 6, run-to-run reproducibility: rounds 2-5 split the rows over blockIdx.y and combined the 8-32 partial sums of a
// column with f32 atomics; here a row lane sums rows rl, rl + 32, ... in order and lane 0 adds the 32 lane sums in order -- the few
// MB of partial rows are L2-resident, 5-7 us as before).
__global__ __launch_bounds__(1024) void ln_colreduce_kernel(const float* ws, int rows, int C, float* dgamma, float* dbeta) {
  __shared__ float part[32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;  // 32 columns x 32 row lanes
  const int col = blockIdx.x * 32 + cl;
  float acc = 0.f;
  if (col < 2 * C) {
    int r = rl;
    for (; r + 96 < rows; r += 128) {                     // four independent loads in flight
      const float a0 = ws[(size_t)r * 2 * C + col], a1 = ws[(size_t)(r + 32) * 2 * C + col];
      const float a2 = ws[(size_t)(r + 64) * 2 * C + col], a3 = ws[(size_t)(r + 96) * 2 * C + col];
      acc += a0; acc += a1; acc += a2; acc += a3;
    }
    for (; r < rows; r += 32) acc += ws[(size_t)r * 2 * C + col];
  }
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && col < 2 * C) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) v += part[k][cl];
    float* o = col < C ? dgamma + col : dbeta + (col - C);
    *o += v;                                              // single writer per column
  }
}

}  // namespace

extern "C" int vmvm_layernorm_fwd(const vmvm_ln_fwd_desc* d, void* stream) {
  if (!d || !d->X || !d->Y || !d->gamma || !d->beta || !d->mean || !d->rstd) return VMVM_EINVAL;
  if (d->M <= 0 || d->C <= 0 || (d->C & 7) || d->nseg < 1 || (d->C % d->nseg) || ((d->C / d->nseg) & 7)) return VMVM_EINVAL;
  if ((d->ldx & 7) || (d->ldy & 7)) return VMVM_EINVAL;
  if (d->x_fp32 && d->src) return VMVM_ENOSUPPORT;
  if (d->src && (d->rows_out_per_batch <= 0 || d->rows_in_per_batch <= 0)) return VMVM_EINVAL;
  if (d->C > 6 * 512) return VMVM_ENOSUPPORT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = (d->M + 3) / 4;
  if (d->C <= 256) {                                    // narrow rows: 4 (C <= 128) or 2 rows per wave
    const int rpw = d->C <= 128 ? 4 : 2;
    const int gridp = (d->M + 4 * rpw - 1) / (4 * rpw);
    if (d->x_fp32) {
      if (rpw == 4) hipLaunchKernelGGL((ln_fwd_pk_kernel<16, true>), dim3(gridp), dim3(256), 0, st, *d);
      else hipLaunchKernelGGL((ln_fwd_pk_kernel<32, true>), dim3(gridp), dim3(256), 0, st, *d);
    } else {
      if (rpw == 4) hipLaunchKernelGGL((ln_fwd_pk_kernel<16, false>), dim3(gridp), dim3(256), 0, st, *d);
      else hipLaunchKernelGGL((ln_fwd_pk_kernel<32, false>), dim3(gridp), dim3(256), 0, st, *d);
    }
  } else if (d->x_fp32) {
    if (d->C > 512) return VMVM_ENOSUPPORT;
    hipLaunchKernelGGL((ln_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, st, *d);
  } else if (d->C <= 512) hipLaunchKernelGGL((ln_fwd_kernel<1, false>), dim3(grid), dim3(256), 0, st, *d);
  else if (d->C <= 1024) hipLaunchKernelGGL((ln_fwd_kernel<2, false>), dim3(grid), dim3(256), 0, st, *d);
  else if (d->C <= 2048) hipLaunchKernelGGL((ln_fwd_kernel<4, false>), dim3(grid), dim3(256), 0, st, *d);
  else hipLaunchKernelGGL((ln_fwd_kernel<6, false>), dim3(grid), dim3(256), 0, st, *d);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

// resident grid of the backward kernels: workgroups that fit per CU at the variant's VGPR count x 256 CUs, at most one per 4 rows.
// ONE helper for the launcher and the workspace-size query, so the two plans cannot drift.
static int ln_bwd_grid(int M, int C, int reserve_cus) {
  const int per_cu = vmvm_hook::ln_bwd_per_cu(C <= 128 ? 4 : C <= 512 ? 5 : C <= 1024 ? 4 : 2);
  int grid = (M + 3) / 4;
  const int cus = vmvm_usable_cus(reserve_cus);
  if (grid > cus * per_cu) grid = cus * per_cu;
  return grid;
}

extern "C" int vmvm_layernorm_bwd(const vmvm_ln_bwd_desc* d, void* stream) {
  if (!d || !d->dY || !d->X || !d->gamma || !d->mean || !d->rstd || !d->dX || !d->dgamma || !d->dbeta) return VMVM_EINVAL;
  if (d->M <= 0 || d->C <= 0 || (d->C & 7) || d->nseg < 1 || (d->C % d->nseg) || ((d->C / d->nseg) & 7)) return VMVM_EINVAL;
  if ((d->ldx & 7) || (d->lddy & 7) || (d->lddx & 7)) return VMVM_EINVAL;
  if (d->src && (d->rows_out_per_batch <= 0 || d->rows_in_per_batch <= 0)) return VMVM_EINVAL;
  if (d->dX2 && d->src) return VMVM_ENOSUPPORT;
  if (d->C > 6 * 512) return VMVM_ENOSUPPORT;
  if (d->inv && (!d->src || d->nseg != 1 || d->C > 256 || d->rows_in_total <= 0 || d->dX2 || d->pad_mode != 0)) return VMVM_ENOSUPPORT;     // source-major order: the packed kernels, one segment, pad slots constant zero (the walk never visits rows without a source, and with pad_mode != 0 those rows contribute dY to dgamma / dbeta)
  if (d->dx_map && (d->src || d->inv || d->dx_map_len <= 0 || (d->M % d->dx_map_len))) return VMVM_EINVAL;      // scattered dX: the identity walk, whole batches
  if (d->add_by_out && (!d->src || d->nseg != 1 || !d->dX_add)) return VMVM_EINVAL;
  //  combinations nothing defines: dX goes to dx_map[m] but dX2 would stay at row m; the source-major walk indexes its
  // residual gradient by SOURCE row, add_by_out by output row.  dX must not alias dX_add under dx_map (rows are written out of order).
  if (d->dx_map && d->dX2) return VMVM_ENOSUPPORT;
  if (d->add_by_out && d->inv) return VMVM_ENOSUPPORT;
  if (d->dx_map && d->dX_add == d->dX) return VMVM_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // one resident set of workgroups (256 CUs x workgroups that fit per CU at this variant's VGPR count); each loops over rows
  // Workgroups that fit per CU at the variant's VGPR count.  Allocation granule 8: the packed C <= 128 build takes 100 -> 104
  // registers, FOUR waves per SIMD, not five (the C <= 256 build takes 96: five) -- a fifth workgroup per CU ran as a second round and cost 25 % at C = 128.
  // (Measured and dropped: one workgroup fewer per CU to trim a mostly-empty last round of rows -- slower in the step wherever the
  // memory system is not yet saturated; a second row per wave in flight -- costs the wave it was meant to replace.)
  const int Mloop = d->inv ? d->rows_in_total : d->M;   // rows the resident grid walks
  int grid = ln_bwd_grid(Mloop, d->C, d->reserve_cus);
  const size_t sm = (size_t)8 * d->C * sizeof(float);   // 4 wave slabs x [2][C]
  vmvm_ln_bwd_desc dd = *d;
  if (dd.workspace && dd.workspace_bytes < (uint64_t)grid * 2 * d->C * sizeof(float)) dd.workspace = nullptr;
#define LAUNCH_LNB(NCH, XF)                                                                                            \
  do {                                                                                                                  \
    if (sm > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_kernel<NCH, XF>),                   \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)       \
      return VMVM_EHIP;                                                                                                 \
    hipLaunchKernelGGL((ln_bwd_kernel<NCH, XF>), dim3(grid), dim3(256), sm, st, dd);                                    \
  } while (0)
  if (d->C <= 256) {
    const int rpw = d->C <= 128 ? 4 : 2;
    const int gmax = (Mloop + 4 * rpw - 1) / (4 * rpw);
    if (grid > gmax) { grid = gmax; if (dd.workspace && dd.workspace_bytes < (uint64_t)grid * 2 * d->C * sizeof(float)) dd.workspace = nullptr; }
    if (d->x_fp32) {
      if (rpw == 4) hipLaunchKernelGGL((ln_bwd_pk_kernel<16, true>), dim3(grid), dim3(256), sm, st, dd);
      else hipLaunchKernelGGL((ln_bwd_pk_kernel<32, true>), dim3(grid), dim3(256), sm, st, dd);
    } else {
      if (rpw == 4) hipLaunchKernelGGL((ln_bwd_pk_kernel<16, false>), dim3(grid), dim3(256), sm, st, dd);
      else hipLaunchKernelGGL((ln_bwd_pk_kernel<32, false>), dim3(grid), dim3(256), sm, st, dd);
    }
  } else if (d->x_fp32) {
    if (d->C > 512) return VMVM_ENOSUPPORT;
    LAUNCH_LNB(1, true);
  } else if (d->C <= 512) LAUNCH_LNB(1, false);
  else if (d->C <= 1024) LAUNCH_LNB(2, false);
  else if (d->C <= 2048) LAUNCH_LNB(4, false);
  else LAUNCH_LNB(6, false);
  if (dd.workspace) {
    VMVM_CHECK_LAUNCH();
    hipLaunchKernelGGL(ln_colreduce_kernel, dim3((2 * d->C + 31) / 32), dim3(1024), 0, st,
                       reinterpret_cast<const float*>(dd.workspace), grid, d->C, d->dgamma, d->dbeta);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

// scratch of the dgamma / dbeta reduction (one [2][C] f32 partial row per resident workgroup); without it: global atomics
extern "C" int64_t vmvm_layernorm_bwd_workspace_size(const vmvm_ln_bwd_desc* d) {
  if (!d || d->M <= 0 || d->C <= 0) return VMVM_EINVAL;
  return (int64_t)ln_bwd_grid(d->inv && d->rows_in_total > d->M ? d->rows_in_total : d->M, d->C, 0) * 2 * d->C * (int64_t)sizeof(float);     // same plan as the launcher (an upper bound where it trims the grid further)
}

