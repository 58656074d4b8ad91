// patch_embed.hip -- PatchEmbed3D (visbackbone/video_swin.py:390-407) as ONE kernel: the (T,H,W) clip is read once, coalesced, straight into
// MFMA operands (no im2col buffer), the patch cover of the masking step and the appended zero frame (:398) are applied on the way in,
// Conv3d(3 -> E, kernel (2,4,4), stride (1,4,4)) runs on v_mfma_f32_16x16x32_bf16 and the LayerNorm (:405) is the epilogue.
// HBM-bound: 12 B/pixel-triple in, E * (4 + 2) B per token out.
#include "common.h"

namespace {

// Tokens are the MFMA COLUMNS (B operand = pixels), output channels the rows (A operand = weights): a token's E channels then live in
// the 4 lanes {r, r+16, r+32, r+48}, so the LayerNorm statistics are an in-lane sum + two shuffles.  With the weight flattened as
// k = c*32 + dt*16 + dy*4 + dx, MFMA k-step ks IS colour channel c, and a lane's 8 k-values (k = 32 ks + 8 g + e) are the two image rows
// dy = 2 (g & 1) + {0, 1} of frame t + (g >> 1): two 16-byte loads; the 16 lanes of a token tile read 256 contiguous bytes per row.
// The pixels enter as a bf16 hi + lo pair (two MFMAs per operand): the layer is 0.1 % of the flops and sets the precision downstream.
template <int NT>                                         // E / 16
__global__ __launch_bounds__(256) void patch_embed_fwd_kernel(const float* __restrict__ img, const uint8_t* __restrict__ cov, const u16* __restrict__ Wb,
                                                             const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, u16* __restrict__ xo, float* __restrict__ zo, float* __restrict__ mean_o,
                                                             float* __restrict__ rstd_o, int B, int T, int H, int W) {
  constexpr int E = NT * 16;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int Hp = H >> 2, Wp = W >> 2;
  const long ntok = (long)B * T * Hp * Wp;
  const long ntile = (ntok + 15) >> 4;
  const int nwave = (gridDim.x * blockDim.x) >> 6;
  // weight fragments: A[row = channel nt*16 + r][k = 32 ks + 8 g .. + 7]
  bf16x8 wf[NT][3];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) wf[nt][ks] = *reinterpret_cast<const bf16x8*>(Wb + (size_t)(nt * 16 + r) * 96 + ks * 32 + g * 8);
  float bz[NT][4], gm[NT][4], bt[NT][4];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = nt * 16 + g * 4 + j;
      bz[nt][j] = bias[ch]; gm[nt][j] = gamma[ch]; bt[nt][j] = beta[ch];
    }
  const int dt = g >> 1, dy0 = (g & 1) * 2;
  for (long tile = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < ntile; tile += nwave) {
    const long tok = tile * 16 + r;
    const bool tv = tok < ntok;
    const long tq = tv ? tok : 0;
    const int x = (int)(tq % Wp);
    const int y = (int)((tq / Wp) % Hp);
    const int t = (int)((tq / ((long)Wp * Hp)) % T);
    const int b = (int)(tq / ((long)Wp * Hp * T));
    bool valid = tv && (t + dt) < T;                      // frame T is the appended zero frame
    if (valid && cov) valid = cov[(((long)b * T + (t + dt)) * (H >> 5) + (y >> 3)) * (W >> 5) + (x >> 3)] == 0;
    const float* src = img + ((((long)b * T + (t + dt)) * 3) * H + 4 * y + dy0) * W + 4 * x;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 p0[3], p1[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      p0[ks] = make_float4(0.f, 0.f, 0.f, 0.f); p1[ks] = p0[ks];
      if (valid) {
        p0[ks] = *reinterpret_cast<const float4*>(src + (long)ks * H * W);
        p1[ks] = *reinterpret_cast<const float4*>(src + (long)ks * H * W + W);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const float v[8] = {p0[ks].x, p0[ks].y, p0[ks].z, p0[ks].w, p1[ks].x, p1[ks].y, p1[ks].z, p1[ks].w};
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        hi[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
        lo[e] = pack_bf2(v[2 * e] - __uint_as_float(hi[e] << 16), v[2 * e + 1] - __uint_as_float(hi[e] & 0xffff0000u));
      }
      const bf16x8 fh = __builtin_bit_cast(bf16x8, make_uint4(hi[0], hi[1], hi[2], hi[3]));
      const bf16x8 fl = __builtin_bit_cast(bf16x8, make_uint4(lo[0], lo[1], lo[2], lo[3]));
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], fh, acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], fl, acc[nt], 0, 0, 0);
      }
    }
    // lane (r, g): channels nt*16 + 4g + j of token r.  bias, f32 pre-norm output (the LayerNorm backward reads it), statistics, norm
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[nt][j] += bz[nt][j]; s += acc[nt][j]; }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mu = s * (1.0f / E);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = acc[nt][j] - mu; q += d * d; }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rs = __builtin_amdgcn_rsqf(q * (1.0f / E) + eps);
    if (tv) {
      float* zp = zo + tok * E + g * 4;
      u16* xp = xo + tok * E + g * 4;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        *reinterpret_cast<float4*>(zp + nt * 16) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
        float yv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) yv[j] = (acc[nt][j] - mu) * rs * gm[nt][j] + bt[nt][j];
        *reinterpret_cast<uint2*>(xp + nt * 16) = make_uint2(pack_bf2(yv[0], yv[1]), pack_bf2(yv[2], yv[3]));
      }
      if (g == 0) { mean_o[tok] = mu; rstd_o[tok] = rs; }
    }
  }
}

template <int NT>
int launch_pe(const float* img, const uint8_t* cov, const u16* Wb, const float* bias, const float* gamma, const float* beta, float eps, u16* xo, float* zo,
              float* mean_o, float* rstd_o, int B, int T, int H, int W, hipStream_t st) {
  const long ntile = ((long)B * T * (H / 4) * (W / 4) + 15) / 16;
  long blocks = (ntile + 3) / 4;
  if (blocks > 2048) blocks = 2048;                       // 8 resident workgroups per CU walk the tiles
  hipLaunchKernelGGL((patch_embed_fwd_kernel<NT>), dim3((unsigned)blocks), dim3(256), 0, st, img, cov, Wb, bias, gamma, beta, eps, xo, zo, mean_o, rstd_o, B, T, H, W);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

}  // namespace

extern "C" int vmvm_patch_embed_fwd(const float* img, const uint8_t* cov, const void* weight_bf16, const float* bias, const float* gamma,
                                    const float* beta, float eps, void* x_out, float* z_out, float* mean, float* rstd, int32_t B, int32_t T,
                                    int32_t H, int32_t W, int32_t E, void* stream) {
  if (!img || !weight_bf16 || !bias || !gamma || !beta || !x_out || !z_out || !mean || !rstd || B <= 0 || T <= 0 || (H & 3) || (W & 3)) return VMVM_EINVAL;
  if (cov && ((H & 31) || (W & 31))) return VMVM_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const u16* wb = reinterpret_cast<const u16*>(weight_bf16);
  u16* xo = reinterpret_cast<u16*>(x_out);
  switch (E) {
    case 96: return launch_pe<6>(img, cov, wb, bias, gamma, beta, eps, xo, z_out, mean, rstd, B, T, H, W, st);
    case 128: return launch_pe<8>(img, cov, wb, bias, gamma, beta, eps, xo, z_out, mean, rstd, B, T, H, W, st);
    case 192: return launch_pe<12>(img, cov, wb, bias, gamma, beta, eps, xo, z_out, mean, rstd, B, T, H, W, st);
    case 32: return launch_pe<2>(img, cov, wb, bias, gamma, beta, eps, xo, z_out, mean, rstd, B, T, H, W, st);
    case 64: return launch_pe<4>(img, cov, wb, bias, gamma, beta, eps, xo, z_out, mean, rstd, B, T, H, W, st);
    default: return VMVM_ENOSUPPORT;
  }
}
