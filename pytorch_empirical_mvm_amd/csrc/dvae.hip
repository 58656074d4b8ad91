// dvae.hip -- the non-GEMM passes of the frozen DALL-E dVAE tokenizer (MVM 'vq' target; visbackbone/dalle/encoder.py:41-93,
// __init__.py:38-54): stem im2col with the pixel pre-processing fused, 2x2 max-pool on NHWC fp16, reduction of the fused
// arg-max pairs.  All HBM-bound, one pass each; the convolutions themselves are GEMMs of gemm.hip (in_fp16 / conv_taps).
#include "common.h"

namespace {

inline int nblk(long n, int per) { return (int)((n + per - 1) / per); }

// ---- 7x7 stem (encoder.py:59: Conv2d(3 -> n_hid, 7), padding 3) as im2col rows for a k-major fp16 GEMM ---------------------------
// img: [n][3][H][W] f32, ImageNet-normalised (what the training step holds).  DalleModel.preprocess (__init__.py:38-42) un-normalises
// (x * std + mean) and map_pixels (utils.py:46-52) squeezes into [eps, 1 - eps]: v = 0.8 * (x * std + mean) + 0.1; the convolution pads
// the PRE-PROCESSED image with zeros.  Row m = pixel (n, y, x); column k = ky * 24 + kx * 3 + c for kx < 7 (columns 21..23 of a ky
// block and 168..191 are zero): 192 fp16 per row, K a multiple of the GEMM's 64-wide K tile.
// One workgroup = TW consecutive output pixels of one image row: the 7 x (TW + 6) x 3 input patch is loaded ONCE, coalesced along x,
// pre-processed and zero-padded on the way into LDS (every input pixel is used by up to 49 outputs); then item (pixel, 16-byte chunk)
// with the chunk index fastest, so 24 adjacent threads write one pixel's 384 contiguous bytes.
constexpr int STEM_TW = 32;
__global__ __launch_bounds__(256) void dvae_stem_im2col_kernel(const float* __restrict__ img, u16* __restrict__ cols, int n_img, int H, int W) {
  __shared__ float patch[3][7][STEM_TW + 6];
  const int segs = (W + STEM_TW - 1) / STEM_TW;
  const long wg = blockIdx.x;
  const int seg = (int)(wg % segs);
  const long t = wg / segs;
  const int y = (int)(t % H);
  const int n = (int)(t / H);
  const int x0 = seg * STEM_TW;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  for (int i = threadIdx.x; i < 3 * 7 * (STEM_TW + 6); i += 256) {
    const int xx = i % (STEM_TW + 6), r = i / (STEM_TW + 6), ky = r % 7, c = r / 7;
    const int gy = y + ky - 3, gx = x0 + xx - 3;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = 0.8f * (img[(((long)n * 3 + c) * H + gy) * W + gx] * stdv[c] + mean[c]) + 0.1f;
    patch[c][ky][xx] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < STEM_TW * 24; i += 256) {
    const int q = i % 24, px = i / 24;
    if (x0 + px >= W) break;
    const int ky = q / 3, part = q - ky * 3;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int idx = part * 8 + e;
      const int kx = idx / 3, c = idx - kx * 3;
      v[e] = (ky < 7 && idx < 21) ? patch[c][ky < 7 ? ky : 0][px + kx] : 0.f;
    }
    *reinterpret_cast<uint4*>(cols + ((((long)n * H + y) * W) + x0 + px) * 192 + q * 8) = pack_h8(v);
  }
}

// ---- MaxPool2d(2) (encoder.py:62,66,70) on an NHWC fp16 activation: one thread per 8 channels of an output pixel -------------------
__global__ void maxpool2x2_nhwc_f16_kernel(const u16* __restrict__ x, u16* __restrict__ y, int n_img, int H, int W, int C) {
  const int c8 = C >> 3, Ho = H >> 1, Wo = W >> 1;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n_img * Ho * Wo * c8) return;
  const int ch = (int)(i % c8);
  const long op = i / c8;
  const int xo = (int)(op % Wo);
  const long t = op / Wo;
  const int yo = (int)(t % Ho);
  const int n = (int)(t / Ho);
  const u16* p = x + (((long)n * H + 2 * yo) * W + 2 * xo) * C + ch * 8;
  const f16x8 a = *reinterpret_cast<const f16x8*>(p), b = *reinterpret_cast<const f16x8*>(p + C);
  const f16x8 c = *reinterpret_cast<const f16x8*>(p + (long)W * C), d = *reinterpret_cast<const f16x8*>(p + (long)W * C + C);
  *reinterpret_cast<f16x8*>(y + op * C + ch * 8) = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
}

// ---- token id of a row = column of the largest (value, column) pair over its 64-column groups (ties: smaller column) ---------------
__global__ void argmax_pairs_kernel(const float* __restrict__ pairs, int ld, int M, int groups, int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= M) return;
  float best = -__builtin_inff();
  int bi = 0;                                    // a row of NaNs (no comparison ever true) must still give an in-range id: it becomes a
  //                                                cross-entropy target and a gather index downstream
  for (int q = lane; q < groups; q += 64) {
    const float2 pr = *reinterpret_cast<const float2*>(pairs + (size_t)row * ld + 2 * q);
    const int idx = __float_as_int(pr.y);
    if (pr.x > best || (pr.x == best && idx < bi)) { best = pr.x; bi = idx; }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) out[row] = (int64_t)min(max(bi, 0), 64 * groups - 1);
}

}  // namespace

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int vmvm_dvae_stem_im2col(const float* img, void* cols, int32_t n_img, int32_t H, int32_t W, void* stream) {
  if (!img || !cols || n_img <= 0 || H <= 0 || W <= 0) return VMVM_EINVAL;
  const long wgs = (long)n_img * H * ((W + STEM_TW - 1) / STEM_TW);
  hipLaunchKernelGGL(dvae_stem_im2col_kernel, dim3((unsigned)wgs), dim3(256), 0, ST, img, reinterpret_cast<u16*>(cols), n_img, H, W);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

extern "C" int vmvm_maxpool2x2_nhwc_f16(const void* x, void* y, int32_t n_img, int32_t H, int32_t W, int32_t C, void* stream) {
  if (!x || !y || n_img <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 7)) return VMVM_EINVAL;
  const long n = (long)n_img * (H / 2) * (W / 2) * (C / 8);
  hipLaunchKernelGGL(maxpool2x2_nhwc_f16_kernel, dim3(nblk(n, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(x), reinterpret_cast<u16*>(y), n_img, H, W, C);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

extern "C" int vmvm_argmax_pairs(const float* pairs, int32_t ld, int32_t M, int32_t groups, int64_t* out, void* stream) {
  if (!pairs || !out || M <= 0 || groups <= 0 || ld < 2 * groups || (ld & 1)) return VMVM_EINVAL;
  hipLaunchKernelGGL(argmax_pairs_kernel, dim3(nblk(M, 4)), dim3(256), 0, ST, pairs, ld, M, groups, out);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
