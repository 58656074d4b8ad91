// misc.hip -- small fused HBM-bound kernels of the VIOLETv2 step (gfx950): im2col, token assembly,
// embeddings, fused cross-entropy / masked-L1 losses (forward + gradient in one pass), bias
// gradients, fused grad-norm + AdamW, and a hardware probe for ds_read_b64_tr_b16.
#include "common.h"

thread_local int g_vmvm_last_hip_error = 0;

extern "C" int vmvm_version(void) { return (0 << 16) | 1; }
extern "C" int vmvm_last_hip_error(void) { return g_vmvm_last_hip_error; }

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {   // 256-thread block reduction
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ---------------------------------------------------------------- PatchEmbed3D im2col
__global__ void im2col_kernel(const float* __restrict__ img, const uint8_t* __restrict__ cov, u16* __restrict__ cols, int B, int T, int H, int W) {
  const int Hp = H / 4, Wp = W / 4;
  const long ntok = (long)B * T * Hp * Wp;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntok * 6) return;
  const long tok = i / 6;                               // (colour, frame) fastest: six adjacent threads write one token's 192 contiguous bytes
  const int cd = (int)(i - tok * 6);
  const int c = cd >> 1, dt = cd & 1;
  const int x = (int)(tok % Wp);
  const int y = (int)((tok / Wp) % Hp);
  const int t = (int)((tok / ((long)Wp * Hp)) % T);
  const int b = (int)(tok / ((long)Wp * Hp * T));
  u16* dst = cols + tok * 192 + c * 32 + dt * 16;
  bool valid = (t + dt) < T;                            // frame T is the appended zero frame (video_swin.py:398)
  if (valid && cov) valid = cov[(((long)b * T + (t + dt)) * (H / 32) + (y >> 3)) * (W / 32) + (x >> 3)] == 0;
  const float* src = img + ((((long)b * T + (t + dt)) * 3 + c) * H + 4 * y) * W + 4 * x;
#pragma unroll
  for (int dy = 0; dy < 4; ++dy) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) v = *reinterpret_cast<const float4*>(src + (long)dy * W);
    const uint2 hi = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    *reinterpret_cast<uint2*>(dst + dy * 4) = hi;
    const float rx = v.x - __uint_as_float(hi.x << 16), ry = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float rz = v.z - __uint_as_float(hi.y << 16), rw = v.w - __uint_as_float(hi.y & 0xffff0000u);
    *reinterpret_cast<uint2*>(dst + 96 + dy * 4) = make_uint2(pack_bf2(rx, ry), pack_bf2(rz, rw));
  }
}

// ---------------------------------------------------------------- EncVideo token assembly
__global__ void encvideo_assemble_kernel(const u16* __restrict__ fc, const float* __restrict__ cls, const float* __restrict__ pos,
                                         const float* __restrict__ len, u16* __restrict__ out, int B, int T, int hw, int Hd) {
  const int nch = Hd / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * T * (1 + hw) * nch;
  if (i >= total) return;
  const int ch = (int)(i % nch);
  const long row = i / nch;
  const int pp = (int)(row % (1 + hw));
  const long bt = row / (1 + hw);
  const int t = (int)(bt % T);
  float v[8];
  if (pp == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = cls[ch * 8 + e];
  } else {
    unpack_bf8(*reinterpret_cast<const uint4*>(fc + (bt * hw + (pp - 1)) * Hd + ch * 8), v);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += pos[(long)pp * Hd + ch * 8 + e] + len[(long)t * Hd + ch * 8 + e];
  *reinterpret_cast<uint4*>(out + row * Hd + ch * 8) = pack_bf8(v);
}

// mode 0: d_fc copy ; mode 1: dpos[p] (p==0 also -> dcls) ; mode 2: dlen[t]
__global__ void encvideo_assemble_bwd_kernel(const u16* __restrict__ dpre, u16* __restrict__ dfc, float* __restrict__ dcls,
                                             float* __restrict__ dpos, float* __restrict__ dlen, int B, int T, int hw, int Hd, int mode) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int P = 1 + hw;
  if (mode == 0) {
    const int nch = Hd / 8;
    const long total = (long)B * T * hw * nch;
    if (i >= total) return;
    const int ch = (int)(i % nch);
    const long row = i / nch;
    const int pp = (int)(row % hw);
    const long bt = row / hw;
    *reinterpret_cast<uint4*>(dfc + row * Hd + ch * 8) = *reinterpret_cast<const uint4*>(dpre + (bt * P + pp + 1) * Hd + ch * 8);
  }
}
// Round 6 (run-to-run reproducibility; rounds 2-5: one f32 atomic per clip and output).  A workgroup owns 64 columns of ONE output row
// -- dpos[pp] (MODE 1: sum over the B * T (clip, frame) rows at position pp; pp == 0 also goes to dcls) or dlen[t] (MODE 2: sum over the
// B * P rows of frame t) -- its four row lanes each sum a quarter of the terms in index order and lane 0 adds the four in order.
template <int MODE>
__global__ __launch_bounds__(256) void encvideo_embed_grad_kernel(const u16* __restrict__ dpre, float* __restrict__ dcls, float* __restrict__ dpos,
                                                                  float* __restrict__ dlen, int B, int T, int hw, int Hd) {
  __shared__ float part[4][64];
  const int P = 1 + hw;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ncg = (Hd + 63) / 64;
  const int o = blockIdx.x / ncg, col = (blockIdx.x % ncg) * 64 + cl;      // output row (pp or t), column
  const int nterm = MODE == 1 ? B * T : B * P;
  float acc = 0.f;
  if (col < Hd) {
    for (int k = rl; k < nterm; k += 4) {
      long row;
      if (MODE == 1) row = (long)k * P + o;                                  // k = b * T + t
      else row = ((long)(k / P) * T + o) * P + (k % P);                      // k = b * P + pp
      acc += bf2f(dpre[row * Hd + col]);
    }
  }
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && col < Hd) {
    const float v = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
    if (MODE == 1) { dpos[(long)o * Hd + col] += v; if (o == 0) dcls[col] += v; }
    else dlen[(long)o * Hd + col] += v;
  }
}

// ---------------------------------------------------------------- BERT embeddings
__global__ void bert_embed_kernel(const int64_t* __restrict__ txt, const float* __restrict__ word, const float* __restrict__ pos,
                                  const float* __restrict__ type0, u16* __restrict__ out, int B, int X, int Hd) {
  const int nch = Hd / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * X * nch) return;
  const int ch = (int)(i % nch);
  const long row = i / nch;
  const int x = (int)(row % X);
  const long id = txt[row];
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = word[id * Hd + ch * 8 + e] + pos[(long)x * Hd + ch * 8 + e] + type0[ch * 8 + e];
  *reinterpret_cast<uint4*>(out + row * Hd + ch * 8) = pack_bf8(v);
}
// Round 6 (run-to-run reproducibility; rounds 1-5: three f32 atomics per (row, column)).  1 024-thread workgroups, three kinds:
//   blocks [0, n):                 text row i = (b, x), the token ids of the whole batch in LDS -- the row that holds the FIRST occurrence
//                                  of its token sums every row with that token in row order into dword[tok] (single writer);
//   blocks [n, n + X ncg):         dpos[x], 64 columns: 16 row lanes sum the B rows of position x (lane order), lane 0 adds the 16;
//   blocks [n + X ncg, .. + ncg):  dtype0, 64 columns: 16 row lanes over ALL n rows, eight independent loads in flight per lane.
__global__ __launch_bounds__(1024) void bert_embed_bwd_kernel(const int64_t* __restrict__ txt, const u16* __restrict__ dsum, float* __restrict__ dword,
                                                              float* __restrict__ dpos, float* __restrict__ dtype0, int B, int X, int Hd) {
  extern __shared__ int ids[];                             // [B * X]
  __shared__ int first;
  __shared__ float part[16][64];
  const int n = B * X, ncg = (Hd + 63) / 64;
  const int blk = blockIdx.x;
  if (blk < n) {
    const int i = blk;
    for (int k = threadIdx.x; k < n; k += 1024) ids[k] = (int)txt[k];
    if (threadIdx.x == 0) first = 1;
    __syncthreads();
    const int tok = ids[i];
    for (int k = threadIdx.x; k < i; k += 1024)
      if (ids[k] == tok) first = 0;                        // (benign race: every writer stores 0)
    __syncthreads();
    if (!first) return;
    for (int col = threadIdx.x; col < Hd; col += 1024) {
      float t = 0.f;
      for (int k = i; k < n; ++k)
        if (ids[k] == tok) t += bf2f(dsum[(long)k * Hd + col]);
      dword[(long)tok * Hd + col] += t;
    }
    return;
  }
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;  // 64 columns x 16 row lanes
  const bool is_pos = blk < n + X * ncg;
  const int j = is_pos ? blk - n : blk - n - X * ncg;
  const int x = is_pos ? j / ncg : 0, col = (is_pos ? j % ncg : j) * 64 + cl;
  float t = 0.f;
  if (col < Hd) {
    if (is_pos) {
      for (int bb = rl; bb < B; bb += 16) t += bf2f(dsum[((long)bb * X + x) * Hd + col]);
    } else {
      int k = rl;
      for (; k + 7 * 16 < n; k += 8 * 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = bf2f(dsum[(long)(k + u * 16) * Hd + col]);
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
      }
      for (; k < n; k += 16) t += bf2f(dsum[(long)k * Hd + col]);
    }
  }
  part[rl][cl] = t;
  __syncthreads();
  if (rl == 0 && col < Hd) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][cl];
    if (is_pos) dpos[(long)x * Hd + col] += v; else dtype0[col] += v;
  }
}

// ---------------------------------------------------------------- cross entropy (ignore_index = -1)
__global__ void count_valid_kernel(const int64_t* __restrict__ target, int M, float* __restrict__ n_valid) {
  __shared__ float sh[4];
  float c = 0.f;
  for (int i = threadIdx.x; i < M; i += 256) c += target[i] >= 0 ? 1.f : 0.f;
  c = block_sum(c, sh);
  if (threadIdx.x == 0) *n_valid = c;
}
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits, int ld, int V, const int64_t* __restrict__ target,
                                                            const float* __restrict__ n_valid, float* __restrict__ loss_sum,
                                                            u16* __restrict__ dlogits, int ld_d) {
  __shared__ float sh[4];
  const long m = blockIdx.x;
  const long tgt = target[m];
  u16* drow = dlogits ? dlogits + m * ld_d : nullptr;
  if (tgt < 0) {
    if (drow) for (int i = threadIdx.x; i < ld_d; i += 256) drow[i] = 0;
    return;
  }
  const float* row = logits + m * ld;
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < V; i += 256) mx = fmaxf(mx, row[i]);
  mx = block_max(mx, sh);
  float s = 0.f;
  for (int i = threadIdx.x; i < V; i += 256) s += __expf(row[i] - mx);
  s = block_sum(s, sh);
  const float lse = mx + __logf(s);
  const float inv_n = 1.0f / fmaxf(*n_valid, 1.0f);
  if (threadIdx.x == 0) atomicAdd(loss_sum, (lse - row[tgt]) * inv_n);
  if (drow) {
    for (int i = threadIdx.x; i < ld_d; i += 256) {
      float g = 0.f;
      if (i < V) g = (__expf(row[i] - lse) - (i == tgt ? 1.f : 0.f)) * inv_n;
      drow[i] = f2bf(g);
    }
  }
}

// ---------------------------------------------------------------- MVM pixel masked L1
__global__ __launch_bounds__(256) void pixel_l1_kernel(const u16* __restrict__ pred, const float* __restrict__ img, const uint8_t* __restrict__ cov,
                                                       const float* __restrict__ mask_sum, float* __restrict__ loss_sum, u16* __restrict__ dpred,
                                                       int B, int T, int h, int w, int ps, int nch, float inv_div) {
  __shared__ float sh[4];
  const long row = blockIdx.x;                 // (b*T+t)*h*w + i*w + j
  const int hw = h * w, C3 = nch * ps * ps;
  const int ij = (int)(row % hw);
  const long bt = row / hw;
  const int i = ij / w, j = ij % w;
  u16* drow = dpred + row * C3;
  if (!cov[row]) {
    for (int k = threadIdx.x * 8; k < C3; k += 256 * 8) *reinterpret_cast<uint4*>(drow + k) = make_uint4(0, 0, 0, 0);
    return;
  }
  const int H = h * ps, W = w * ps;
  const float coef = inv_div / (*mask_sum + 1e-5f);
  float acc = 0.f;
  for (int k = threadIdx.x * 4; k < C3; k += 256 * 4) {
    const int c = k / (ps * ps), rem = k - c * ps * ps, dy = rem / ps, dx = rem - dy * ps;
    const float4 t4 = *reinterpret_cast<const float4*>(img + ((bt * nch + c) * H + (i * ps + dy)) * (long)W + j * ps + dx);
    const uint2 p2 = *reinterpret_cast<const uint2*>(pred + row * C3 + k);
    const float pv[4] = {__uint_as_float(p2.x << 16), __uint_as_float(p2.x & 0xffff0000u), __uint_as_float(p2.y << 16), __uint_as_float(p2.y & 0xffff0000u)};
    const float tv[4] = {t4.x, t4.y, t4.z, t4.w};
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = pv[e] - tv[e];
      acc += fabsf(d);
      g[e] = (d > 0.f ? coef : (d < 0.f ? -coef : 0.f));
    }
    *reinterpret_cast<uint2*>(drow + k) = make_uint2(pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]));
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss_sum, acc * coef);
}

// ---------------------------------------------------------------- MVM feature-target masked L1 (2d_feature / 3d_feature)
// loss += sum_rows cov[row] * sum_c |pred - target| * coef,  dpred = cov[row] * sign(pred - target) * coef,
// coef = inv_div / (mask_sum + 1e-5)   (main_pretrain.py:523-525 / :542-544: the mask sum counts covered PATCHES, inv_div = 1/3)
__global__ __launch_bounds__(256) void feature_l1_kernel(const u16* __restrict__ pred, const u16* __restrict__ tgt, const uint8_t* __restrict__ cov,
                                                         const float* __restrict__ mask_sum, float inv_div, float* __restrict__ loss_sum,
                                                         u16* __restrict__ dpred, int C) {
  __shared__ float sh[4];
  const long row = blockIdx.x;
  u16* drow = dpred + row * C;
  if (!cov[row]) {
    for (int k = threadIdx.x * 8; k < C; k += 256 * 8) *reinterpret_cast<uint4*>(drow + k) = make_uint4(0, 0, 0, 0);
    return;
  }
  const float coef = inv_div / (*mask_sum + 1e-5f);
  float acc = 0.f;
  for (int k = threadIdx.x * 8; k < C; k += 256 * 8) {
    const uint4 p4 = *reinterpret_cast<const uint4*>(pred + row * C + k);
    const uint4 t4 = *reinterpret_cast<const uint4*>(tgt + row * C + k);
    const uint32_t pw[4] = {p4.x, p4.y, p4.z, p4.w}, tw[4] = {t4.x, t4.y, t4.z, t4.w};
    uint32_t gw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d0 = __uint_as_float(pw[e] << 16) - __uint_as_float(tw[e] << 16);
      const float d1 = __uint_as_float(pw[e] & 0xffff0000u) - __uint_as_float(tw[e] & 0xffff0000u);
      acc += fabsf(d0) + fabsf(d1);
      gw[e] = pack_bf2(d0 > 0.f ? coef : (d0 < 0.f ? -coef : 0.f), d1 > 0.f ? coef : (d1 < 0.f ? -coef : 0.f));
    }
    *reinterpret_cast<uint4*>(drow + k) = make_uint4(gw[0], gw[1], gw[2], gw[3]);
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss_sum, acc * coef);
}

// ---------------------------------------------------------------- VTM head tail
__global__ void rowdot_kernel(const u16* __restrict__ hid, int M, int K, const float* __restrict__ w, const float* __restrict__ b,
                              float inv_temp, float* __restrict__ out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= M) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += bf2f(hid[(long)wave * K + k]) * w[k];
  s = wave_sum(s);
  if (lane == 0) out[wave] = (s + b[0]) * inv_temp;
}
__global__ void rowdot_bwd_kernel(const u16* __restrict__ hid, int M, int K, const float* __restrict__ w, const float* __restrict__ dout,
                                  float inv_temp, u16* __restrict__ dhid, float* __restrict__ dw, float* __restrict__ db, int relu_mask) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f, sb = 0.f;
  const float wk = w[k];
  for (int m = 0; m < M; ++m) {
    const float g = dout[m] * inv_temp;
    const float hv = bf2f(hid[(long)m * K + k]);
    s += g * hv;
    sb += g;
    dhid[(long)m * K + k] = f2bf((relu_mask && !(hv > 0.f)) ? 0.f : g * wk);
  }
  dw[k] += s;
  if (k == 0) db[0] += sb;
}

// ---------------------------------------------------------------- generic helpers
__global__ void cast_kernel(const float* __restrict__ src, u16* __restrict__ dst, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    *reinterpret_cast<uint4*>(dst + i) = pack_bf8(f);
  } else {
    for (long k = i; k < n; ++k) dst[k] = f2bf(src[k]);
  }
}
__global__ void uncast_kernel(const u16* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float f[8];
    unpack_bf8(*reinterpret_cast<const uint4*>(src + i), f);
    *reinterpret_cast<float4*>(dst + i) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(dst + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
  } else {
    for (long k = i; k < n; ++k) dst[k] = __uint_as_float((uint32_t)src[k] << 16);
  }
}
__global__ void add_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i + 8 <= n) {
    float x[8], y[8];
    unpack_bf8(*reinterpret_cast<const uint4*>(a + i), x);
    unpack_bf8(*reinterpret_cast<const uint4*>(b + i), y);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] += y[e];
    *reinterpret_cast<uint4*>(out + i) = pack_bf8(x);
  } else {
    for (long k = i; k < n; ++k) out[k] = f2bf(bf2f(a[k]) + bf2f(b[k]));
  }
}
__global__ void gather_rows_kernel(const u16* __restrict__ src, int ld_src, const int32_t* __restrict__ idx, u16* __restrict__ dst, int ld_dst,
                                   long M, int C, int rows_out_per_batch, int rows_in_per_batch) {
  const int nch = C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * nch) return;
  const int ch = (int)(i % nch);
  const long m = i / nch;
  long s;
  if (rows_out_per_batch > 0) {
    const long b = m / rows_out_per_batch;
    const int sr = idx[m - b * rows_out_per_batch];
    s = sr < 0 ? -1 : (long)sr + b * rows_in_per_batch;
  } else {
    s = idx[m];
  }
  uint4 v = make_uint4(0, 0, 0, 0);
  if (s >= 0) v = *reinterpret_cast<const uint4*>(src + s * ld_src + ch * 8);
  *reinterpret_cast<uint4*>(dst + m * ld_dst + ch * 8) = v;
}

// ---- index plumbing of the DropPath dead-clip elimination (engine._swin_block): the attention branch of a Swin block runs on the clips
// whose stochastic-depth draw kept them (video_swin.py:46-54: a dropped clip's branch output is multiplied by 0).
// expand: out[j * len + t] = map[t] < 0 ? -1 : map[t] + list[j] * stride   (per-clip window map -> absolute row map of the kept clips)
// inverse of a gather map: out[src[i]] = i (out pre-set to -1 by the caller of the kernel)
__global__ void invert_map_kernel(const int32_t* __restrict__ src, int n, int32_t* __restrict__ out, int n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int s = src[i]; if (s >= 0 && s < n_out) out[s] = i; }
}
__global__ void expand_batch_map_kernel(const int32_t* __restrict__ map, int len, const int32_t* __restrict__ list, int n, int stride, int32_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * len) return;
  const int j = (int)(i / len), t = (int)(i - (long)j * len);
  const int v = map[t];
  const int b = list[j];                                   // (a negative entry is a padding clip: all its rows are -1)
  out[i] = (v < 0 || b < 0) ? -1 : v + b * stride;
}
// copy: dst rows of the listed clips = src rows of the same clips (identity path of the dropped clips); C % 8 == 0
__global__ void copy_batches_kernel(const u16* __restrict__ src, int ld_src, u16* __restrict__ dst, int ld_dst, const int32_t* __restrict__ list, int n,
                                    int rows, int C) {
  const int nch = C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * rows * nch) return;
  const int ch = (int)(i % nch);
  const long r = i / nch;
  const long row = (long)list[r / rows] * rows + (r % rows);
  *reinterpret_cast<uint4*>(dst + row * ld_dst + ch * 8) = *reinterpret_cast<const uint4*>(src + row * ld_src + ch * 8);
}

__global__ void scatter_add_rows_kernel(const u16* __restrict__ src, int ld_src, const int32_t* __restrict__ idx, float* __restrict__ dst,
                                        int ld_dst, long M, int C) {
  const int nch = C / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * nch) return;
  const int ch = (int)(i % nch);
  const long m = i / nch;
  const int d = idx[m];
  if (d < 0) return;
  float v[8];
  unpack_bf8(*reinterpret_cast<const uint4*>(src + m * ld_src + ch * 8), v);
  float* o = dst + (long)d * ld_dst + ch * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) atomicAdd(o + e, v[e]);
}
// Gradient of the token pool (backward of the sequence assembly main_pretrain.py:243-259): pass 1 holds sequence i = [video i ; text i],
// pass 2 holds the B*O sequences p = i*O + o = [video i ; text tj(p)].  A pool row's gradient is the sum of its copies: video rows have
// the fixed fan-in 1 + O, text row j the sequences of pass 2 listed in txt_list[txt_off[j] .. txt_off[j+1]).  One thread sums 8
// channels in f32 and writes bf16 -- no f32 scatter buffer, no atomics, no cast pass.
__global__ void pool_grad_kernel(const u16* __restrict__ g1, const u16* __restrict__ g2, const u16* __restrict__ g3, u16* __restrict__ out,
                                 int B, int O, int Lv, int X, int Hd, const int32_t* __restrict__ txt_off, const int32_t* __restrict__ txt_list) {
  const int c8 = Hd >> 3, Lq = Lv + X;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long rows = (long)B * (Lv + X);
  if (i >= rows * c8) return;
  const int ch = (int)(i % c8);
  const long row = i / c8;
  float acc[8], v[8];
  if (row < (long)B * Lv) {
    const int b = (int)(row / Lv), t = (int)(row - (long)b * Lv);
    unpack_bf8(*reinterpret_cast<const uint4*>(g1 + ((size_t)b * Lq + t) * Hd + ch * 8), acc);
    if (g3) {
      unpack_bf8(*reinterpret_cast<const uint4*>(g3 + ((size_t)b * Lq + t) * Hd + ch * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    for (int o = 0; o < O; ++o) {
      unpack_bf8(*reinterpret_cast<const uint4*>(g2 + (((size_t)b * O + o) * Lq + t) * Hd + ch * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  } else {
    const long tr = row - (long)B * Lv;
    const int j = (int)(tr / X), x = (int)(tr - (long)j * X);
    unpack_bf8(*reinterpret_cast<const uint4*>(g1 + ((size_t)j * Lq + Lv + x) * Hd + ch * 8), acc);
    if (g3) {
      unpack_bf8(*reinterpret_cast<const uint4*>(g3 + ((size_t)j * Lq + Lv + x) * Hd + ch * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
    for (int e_ = txt_off[j]; e_ < txt_off[j + 1]; ++e_) {
      unpack_bf8(*reinterpret_cast<const uint4*>(g2 + ((size_t)txt_list[e_] * Lq + Lv + x) * Hd + ch * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
  *reinterpret_cast<uint4*>(out + (size_t)row * Hd + ch * 8) = pack_bf8(acc);
}
__global__ void gelu_bwd_kernel(const u16* __restrict__ dy, const u16* __restrict__ u, u16* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f2bf(bf2f(dy[i]) * gelu_grad_f(bf2f(u[i])));
}
__global__ void dropout_kernel(const u16* __restrict__ x, u16* __restrict__ y, long n, float p, uint64_t seed, uint64_t offset) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i >= n) return;
  const uint32_t thr = dropout_threshold(p);
  const float ks = 1.f / (1.f - p);
  uint32_t bits[8];
  dropout_bits8(seed, offset, (uint64_t)(i >> 3), bits);
  for (int e = 0; e < 8 && i + e < n; ++e) y[i + e] = bits[e] < thr ? (u16)0 : f2bf(bf2f(x[i + e]) * ks);
}

// column sums: block = 8 column-chunks (64 cols) x 32 row lanes ; grid.y splits the rows
// `part` != NULL (round 6, run-to-run reproducibility): block row y leaves its sums in part[y][N] and colsum_fin_kernel adds the
// gridDim.y rows in order; NULL: one f32 atomic per (block row, column) into `out` as in rounds 1-5.
__global__ __launch_bounds__(256) void colsum_kernel(const u16* __restrict__ X, int M, int N, int ldx, const float* __restrict__ row_scale,
                                                     int rows_per_scale, float all_scale, float* __restrict__ out, float* __restrict__ part) {
  __shared__ float sh[32][65];
  const int cc = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int col = blockIdx.x * 64 + cc * 8;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < N) {
    // four rows per trip: the loads (and the row-scale lookups) of a trip are independent, so four requests per lane are in flight
    const long step = (long)gridDim.y * 32;
    long m = (long)blockIdx.y * 32 + rl;
    for (; m + 3 * step < M; m += 4 * step) {
      uint4 raw[4]; float sc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        raw[u] = *reinterpret_cast<const uint4*>(X + (m + u * step) * ldx + col);
        sc[u] = row_scale ? row_scale[(m + u * step) / rows_per_scale] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float v[8];
        unpack_bf8(raw[u], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e] * sc[u];
      }
    }
    for (; m < M; m += step) {
      float v[8];
      unpack_bf8(*reinterpret_cast<const uint4*>(X + m * ldx + col), v);
      const float s = row_scale ? row_scale[m / rows_per_scale] : 1.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e] * s;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[rl][cc * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) s += sh[k][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < N) {
      if (part) part[(size_t)blockIdx.y * N + c] = s * all_scale; else atomicAdd(out + c, s * all_scale);
    }
  }
}
__global__ __launch_bounds__(1024) void colsum_fin_kernel(const float* __restrict__ part, int gy, int N, float* __restrict__ out) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;  // 64 columns x 16 row lanes, lanes combined in order
  const int c = blockIdx.x * 64 + cl;
  float t = 0.f;
  if (c < N)
    for (int y = rl; y < gy; y += 16) t += part[(size_t)y * N + c];
  sh[rl][cl] = t;
  __syncthreads();
  if (rl == 0 && c < N) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += sh[k][cl];
    out[c] += v;
  }
}

// ---------------------------------------------------------------- batched bf16 transpose (W [N][K] -> W^T [K][N])
// One launch transposes every 2-D weight of the arena: table[tile] = {offset, N, K, tile_row*65536 + tile_col}.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const u16* __restrict__ src, u16* __restrict__ dst, const int4* __restrict__ table, const int ntiles) {
  // 64x64 tile through LDS as 32-bit words holding 2x2 sub-blocks: a thread transposes 2x2 in registers, so both the
  // LDS writes and reads are conflict-free 32-bit accesses and global accesses stay 16-byte.
  // Round 6: a RESIDENT grid walking the tile table (two LDS tiles: the next tile's loads are issued before this tile's stores).  One
  // workgroup per 8-KiB tile was 55 000 workgroups for the arena's 450 MB: 0.93 ms = 0.97 TB/s, bound by workgroup dispatch, not by HBM.
  __shared__ uint32_t t[2][64][33];                    // [buffer][row][col pair], +1 pad
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  auto load = [&](int tile, uint4 (&v)[2]) __attribute__((always_inline)) {
    const int4 e = table[tile];
    const long off = e.x; const int N = e.y, K = e.z, r0 = (e.w >> 16) * 64, c0 = (e.w & 0xffff) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (threadIdx.x >> 3) + 32 * i, ch = threadIdx.x & 7;
      v[i] = make_uint4(0, 0, 0, 0);
      if (r0 + row < N && c0 + ch * 8 < K) {
        const v4u x = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(src + off + (long)(r0 + row) * K + c0 + ch * 8));
        v[i] = make_uint4(x[0], x[1], x[2], x[3]);
      }
    }
  };
  uint4 v[2];
  int tile = blockIdx.x, buf = 0;
  if (tile < ntiles) load(tile, v);
  for (; tile < ntiles; tile += gridDim.x, buf ^= 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (threadIdx.x >> 3) + 32 * i, ch = threadIdx.x & 7;
      t[buf][row][ch * 4 + 0] = v[i].x; t[buf][row][ch * 4 + 1] = v[i].y; t[buf][row][ch * 4 + 2] = v[i].z; t[buf][row][ch * 4 + 3] = v[i].w;
    }
    __syncthreads();                                   // (one barrier per tile: the other buffer was last read two iterations ago, behind the previous barrier)
    const int4 e = table[tile];
    const long off = e.x; const int N = e.y, K = e.z, r0 = (e.w >> 16) * 64, c0 = (e.w & 0xffff) * 64;
    if (tile + (int)gridDim.x < ntiles) load(tile + gridDim.x, v);
    // output row = source column c (0..63), 8 consecutive source rows per 16-byte store
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int col = (threadIdx.x >> 3) + 32 * i, ch = threadIdx.x & 7;        // 8 lanes write one 128-byte row segment
      if (c0 + col < K && r0 + ch * 8 < N) {
        uint32_t w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = t[buf][ch * 8 + k][col >> 1];
        const int sh = (col & 1) * 16;
        v4u o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ((w[2 * k] >> sh) & 0xffffu) | (((w[2 * k + 1] >> sh) & 0xffffu) << 16);
        *reinterpret_cast<v4u*>(dst + off + (long)(c0 + col) * N + r0 + ch * 8) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- bf16 -> fp8 (OCP e4m3) per-tensor quantisation
__global__ __launch_bounds__(256) void cast_fp8_kernel(const u16* __restrict__ src, uint8_t* __restrict__ dst, long n8, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + i * 8);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float f[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = fminf(fmaxf(__uint_as_float(w[e] << 16) * scale, -448.f), 448.f);
      f[2 * e + 1] = fminf(fmaxf(__uint_as_float(w[e] & 0xffff0000u) * scale, -448.f), 448.f);
    }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    *reinterpret_cast<uint2*>(dst + i * 8) = make_uint2((uint32_t)lo, (uint32_t)hi);
  }
}


// ---------------------------------------------------------------- self-attention of ONE query position per sequence
// HF BertSelfAttention (call site model.py:213) restricted to a single query row: the VTM pass (main_pretrain.py:243-262) reads the
// fusion encoder's output at the text [CLS] position only, so in its LAST layer every other query row is dead code; K and V of all
// positions are still needed.  One workgroup per (sequence, head): scores of the L keys against the one query (f32), softmax, dropout
// on the probabilities (Philox stream of the GEMM epilogues: 8-element blocks, 16 bits per element), P V.  The probabilities before
// (p) and after (pd = mask * p / keep) dropout are saved for the backward (L floats each per (sequence, head): nothing is recomputed).
// HBM-bound by the K / V rows (read once, whole 128-byte lines for head_dim 64).
template <int HD>
__global__ __launch_bounds__(256) void attn_qrow_fwd_kernel(const u16* __restrict__ q, int ld_q, const u16* __restrict__ kv, int ld_kv, int k_off, int v_off,
                                                            const uint8_t* __restrict__ keymask, u16* __restrict__ out, int ld_out,
                                                            float* __restrict__ probs, float* __restrict__ probs_drop, int L, int heads, float scale,
                                                            float dropout_p, uint64_t seed, uint64_t offset) {
  extern __shared__ float qsm[];                         // [HD] query, [L] pd, [4][HD] partial outputs
  __shared__ float sh[4];
  float* pdsm = qsm + HD;
  float* part = pdsm + ((L + 3) & ~3);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads, tid = threadIdx.x;
  if (tid < HD) qsm[tid] = bf2f(q[(size_t)seq * ld_q + h * HD + tid]);
  __syncthreads();
  const u16* kb = kv + (size_t)seq * L * ld_kv + k_off + h * HD;
  const u16* vb = kv + (size_t)seq * L * ld_kv + v_off + h * HD;
  // scores: thread-per-key, kept in LDS (pdsm doubles as the score row until the probabilities replace it)
  float mx = -__builtin_inff();
  for (int j = tid; j < L; j += 256) {
    float a = 0.f;
    const uint4* kr = reinterpret_cast<const uint4*>(kb + (size_t)j * ld_kv);
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      float f[8];
      unpack_bf8(kr[c], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) a = fmaf(f[e], qsm[c * 8 + e], a);
    }
    const bool ok = !keymask || keymask[(size_t)seq * L + j] != 0;
    const float sj = ok ? a * scale : -__builtin_inff();
    pdsm[j] = sj;                                        // (each thread re-reads only its own entries below: no barrier needed)
    mx = fmaxf(mx, sj);
  }
  mx = block_max(mx, sh);
  float sum = 0.f;
  for (int j = tid; j < L; j += 256) { const float sj = pdsm[j]; const float e_ = (sj == -__builtin_inff()) ? 0.f : __expf(sj - mx); pdsm[j] = e_; sum += e_; }
  sum = block_sum(sum, sh);
  const float inv = sum > 0.f ? 1.0f / sum : 0.f;
  const bool has_drop = dropout_p > 0.f;
  const uint32_t thr = dropout_threshold(dropout_p);
  const float keep = has_drop ? 1.0f / (1.0f - dropout_p) : 1.0f;
  for (int j = tid; j < L; j += 256) {
    const float pj = pdsm[j] * inv;
    float pd = pj;
    if (has_drop) {
      const uint64_t el = ((uint64_t)seq * heads + h) * (uint64_t)L + j;
      uint32_t bits[8];
      dropout_bits8(seed, offset, el >> 3, bits);
      pd = bits[el & 7] < thr ? 0.f : pj * keep;
    }
    probs[((size_t)seq * heads + h) * L + j] = pj;
    probs_drop[((size_t)seq * heads + h) * L + j] = pd;
    pdsm[j] = pd;
  }
  __syncthreads();
  // out[d] = sum_j pd[j] * v[j][d]: 4 groups of 64 threads take interleaved keys, lane = d (coalesced 128-byte rows)
  {
    const int grp = tid >> 6, d = tid & 63;
    float acc = 0.f;
    if (d < HD)
      for (int j = grp; j < L; j += 4) acc = fmaf(pdsm[j], bf2f(vb[(size_t)j * ld_kv + d]), acc);
    if (d < HD) part[grp * HD + d] = acc;
  }
  __syncthreads();
  if (tid < HD) out[(size_t)seq * ld_out + h * HD + tid] = f2bf(part[tid] + part[HD + tid] + part[2 * HD + tid] + part[3 * HD + tid]);
}

// backward of the above: given d(out) of the one query row -> dq (that row), dK / dV of every position (written whole: masked keys get zeros)
template <int HD>
__global__ __launch_bounds__(256) void attn_qrow_bwd_kernel(const u16* __restrict__ dout, int ld_dout, const u16* __restrict__ q, int ld_q,
                                                            const u16* __restrict__ kv, int ld_kv, int k_off, int v_off,
                                                            const float* __restrict__ probs, const float* __restrict__ probs_drop,
                                                            u16* __restrict__ dq, int ld_dq, u16* __restrict__ dkv, int ld_dkv, int L, int heads, float scale) {
  extern __shared__ float sm_[];                         // [HD] dout, [HD] q, [L] ds, [4][HD] partial dq
  __shared__ float sh[4];
  float* dosm = sm_;
  float* qsm = sm_ + HD;
  float* dssm = qsm + HD;
  float* part = dssm + ((L + 3) & ~3);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads, tid = threadIdx.x;
  if (tid < HD) { dosm[tid] = bf2f(dout[(size_t)seq * ld_dout + h * HD + tid]); qsm[tid] = bf2f(q[(size_t)seq * ld_q + h * HD + tid]); }
  __syncthreads();
  const u16* kb = kv + (size_t)seq * L * ld_kv + k_off + h * HD;
  const u16* vb = kv + (size_t)seq * L * ld_kv + v_off + h * HD;
  u16* dkb = dkv + (size_t)seq * L * ld_dkv + k_off + h * HD;
  u16* dvb = dkv + (size_t)seq * L * ld_dkv + v_off + h * HD;
  const float* pr = probs + ((size_t)seq * heads + h) * L;
  const float* pdr = probs_drop + ((size_t)seq * heads + h) * L;
  // dpd[j] = dout . v[j] ; dp[j] = dpd[j] * (pd[j] / p[j]) ; delta = sum_j pd[j] * dpd[j] ; ds[j] = p[j] * (dp[j] - delta) ; dv[j] = pd[j] * dout
  float delta = 0.f;
  for (int j = tid; j < L; j += 256) {
    const uint4* vr = reinterpret_cast<const uint4*>(vb + (size_t)j * ld_kv);
    float a = 0.f;
    const float pdj = pdr[j];
    uint4* dvr = reinterpret_cast<uint4*>(dvb + (size_t)j * ld_dkv);
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      float f[8], o[8];
      unpack_bf8(vr[c], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { a = fmaf(f[e], dosm[c * 8 + e], a); o[e] = pdj * dosm[c * 8 + e]; }
      dvr[c] = pack_bf8(o);
    }
    dssm[j] = a;                                         // dpd[j] for now (own entries only)
    delta = fmaf(pdj, a, delta);
  }
  delta = block_sum(delta, sh);
  for (int j = tid; j < L; j += 256) {
    const float pj = pr[j], pdj = pdr[j];
    const float m = pj > 0.f ? pdj / pj : 0.f;           // dropout multiplier (0 or 1 / keep)
    const float ds = pj * (dssm[j] * m - delta) * scale;
    dssm[j] = ds;
    uint4* dkr = reinterpret_cast<uint4*>(dkb + (size_t)j * ld_dkv);
#pragma unroll
    for (int c = 0; c < HD / 8; ++c) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = ds * qsm[c * 8 + e];
      dkr[c] = pack_bf8(o);
    }
  }
  __syncthreads();
  {
    const int grp = tid >> 6, d = tid & 63;
    float acc = 0.f;
    if (d < HD)
      for (int j = grp; j < L; j += 4) acc = fmaf(dssm[j], bf2f(kb[(size_t)j * ld_kv + d]), acc);
    if (d < HD) part[grp * HD + d] = acc;
  }
  __syncthreads();
  if (tid < HD) dq[(size_t)seq * ld_dq + h * HD + tid] = f2bf(part[tid] + part[HD + tid] + part[2 * HD + tid] + part[3 * HD + tid]);
}

// ---------------------------------------------------------------- optimizer
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out, float* __restrict__ partial) {
  __shared__ float sh[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const float4 v = *reinterpret_cast<const float4*>(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long k = i; k < n; ++k) s += g[k] * g[k];
    }
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) {
    if (partial) partial[blockIdx.x] = s;            // deterministic path: fixed-order second pass (sumsq_final_kernel)
    else atomicAdd(out, s);
  }
}
// fixed-order sum of the per-block partials: every data-parallel rank gets bit-identical clip coefficients from identical gradients
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *out += s;
}
// U = float4 groups per thread and iteration (all of their loads are issued before the first use), NT = non-temporal loads / stores
// (every byte is touched once per step: nothing of it is worth a cache line)
template <int U, bool NT>
__global__ __launch_bounds__(256) void adamw_kernel(const vmvm_adamw_desc d) {
  float coef = d.grad_scale;
  if (d.max_grad_norm > 0.f && d.sumsq) {
    const float norm = sqrtf(*d.sumsq) * d.grad_scale;
    coef *= fminf(1.0f, d.max_grad_norm / (norm + 1e-6f));
  }
  const float decay = 1.0f - d.lr * d.weight_decay;
  const float step_size = d.lr / d.bias_corr1;
  const float inv_sqrt_bc2 = rsqrtf(d.bias_corr2);
  u16* pb = reinterpret_cast<u16*>(d.param_bf16);
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  auto ld4 = [](const float* q) __attribute__((always_inline)) {
    const v4f x = NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(q)) : *reinterpret_cast<const v4f*>(q);
    return make_float4(x[0], x[1], x[2], x[3]);
  };
  auto st4 = [](float* q, float4 x) __attribute__((always_inline)) {
    const v4f y = {x.x, x.y, x.z, x.w};
    if (NT) __builtin_nontemporal_store(y, reinterpret_cast<v4f*>(q)); else *reinterpret_cast<v4f*>(q) = y;
  };
  auto upd = [&](float& p, float g, float& m, float& v) __attribute__((always_inline)) {
    const float gg = g * coef;
    p *= decay;
    m = d.beta1 * m + (1.f - d.beta1) * gg;
    v = d.beta2 * v + (1.f - d.beta2) * gg * gg;
    const float denom = sqrtf(v) * inv_sqrt_bc2 + d.eps;
    p -= step_size * m / denom;
  };
  const long chunk = (long)256 * 4;                       // elements one workgroup covers per group: a wave's loads are 1 KiB contiguous
  const long stride = (long)gridDim.x * chunk * U;
  for (long base = (long)blockIdx.x * chunk * U + (long)threadIdx.x * 4; base < d.n; base += stride) {
    float4 p4[U], g4[U], m4[U], v4[U];
    bool full[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + u * chunk;
      full[u] = i + 4 <= d.n;
      if (full[u]) { p4[u] = ld4(d.param + i); g4[u] = ld4(d.grad + i); m4[u] = ld4(d.m + i); v4[u] = ld4(d.v + i); }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = base + u * chunk;
      if (full[u]) {
        upd(p4[u].x, g4[u].x, m4[u].x, v4[u].x); upd(p4[u].y, g4[u].y, m4[u].y, v4[u].y);
        upd(p4[u].z, g4[u].z, m4[u].z, v4[u].z); upd(p4[u].w, g4[u].w, m4[u].w, v4[u].w);
        st4(d.param + i, p4[u]); st4(d.m + i, m4[u]); st4(d.v + i, v4[u]);
        if (pb) {
          const v2u b2 = {pack_bf2(p4[u].x, p4[u].y), pack_bf2(p4[u].z, p4[u].w)};
          if (NT) __builtin_nontemporal_store(b2, reinterpret_cast<v2u*>(pb + i)); else *reinterpret_cast<v2u*>(pb + i) = b2;
        }
      } else if (i < d.n) {                                // the arena's last partial group (n % 4 != 0)
        for (long e = i; e < d.n; ++e) {
          float p = d.param[e], m = d.m[e], v = d.v[e];
          upd(p, d.grad[e], m, v);
          d.param[e] = p; d.m[e] = m; d.v[e] = v;
          if (pb) pb[e] = f2bf(p);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- probe: ds_read_b64_tr_b16 lane mapping
__global__ void probe_tr16_kernel(int32_t* out) {
  __shared__ __attribute__((aligned(16))) u16 lds[64 * 16];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 16; i += 64) lds[i] = (u16)i;           // lds[row*16+col] = row*16+col, 16 cols per row
  __syncthreads();
  // every 16-lane group g reads the 4x16 block of rows 4g..4g+3: lane i -> row 4g + i/4, cols (i%4)*4..+3
  const int g = lane >> 4, i = lane & 15;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + (4 * g + (i >> 2)) * 16 + (i & 3) * 4));
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (int32_t)(u16)v[e];
}

inline int nblk(long n, int per) { return (int)((n + per - 1) / per); }

}  // namespace


// ---------------------------------------------------------------------------------------------------------------------
// Device-side masking (Agent_Pretrain.masking main_pretrain.py:276-372, 'rm' / 'bm'): one workgroup per clip turns
// explicit uniform draws into the patch cover, the [MASK]-ed token ids and the MLM labels.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int randint_u(float u, int n) {          // numpy randint(0, n) from u in [0,1): floor(u*n), clamped
  int v = (int)(u * (float)n);
  return v > n - 1 ? n - 1 : v;
}
__global__ __launch_bounds__(256) void masking_kernel(int64_t* __restrict__ txt, int64_t* __restrict__ ans_mtm, uint8_t* __restrict__ cov,
                                                      const float* __restrict__ u_type, const float* __restrict__ u_txt,
                                                      const float* __restrict__ u_rm, const float* __restrict__ u_bm,
                                                      const int32_t* __restrict__ types, int n_types, int X, int T, int h, int w, float p,
                                                      int cls, int sep, int pad, int msk) {
  __shared__ int cub[64][6];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int type = types[randint_u(u_type[b], n_types)];             // random.choice(pretrain_masks) (:303)
  for (int x = tid; x < X; x += 256) {                               // MLM: non-special token with rand < p -> label = id, id <- [MASK] (:305,:346,:354)
    const int64_t id = txt[(size_t)b * X + x];
    const bool spc = id == cls || id == sep || id == pad || id == msk;
    const bool sel = !spc && u_txt[(size_t)b * X + x] < p;
    ans_mtm[(size_t)b * X + x] = sel ? id : (int64_t)-1;
    if (sel) txt[(size_t)b * X + x] = msk;
  }
  const int hw = h * w;
  if (type == 0) {                                                   // 'rm': rand((1+hw)*T) < p, the per-frame cls slot is never a target (:348-352)
    for (int i = tid; i < T * hw; i += 256) {
      const int t = i / hw, pp = i - t * hw;
      cov[((size_t)b * T + t) * hw + pp] = u_rm[((size_t)b * T + t) * (1 + hw) + 1 + pp] < p ? 1 : 0;
    }
  } else {                                                           // 'bm': T cuboids, numpy randint bounds of :308-313
    for (int k = tid; k < T; k += 256) {
      const float* u = u_bm + ((size_t)b * T + k) * 6;
      const int t = T > 1 ? 1 + randint_u(u[0], T - 1) : 1;
      const int hh = 1 + randint_u(u[1], h * 2 / 3 - 1), ww = 1 + randint_u(u[2], w * 2 / 3 - 1);
      cub[k][0] = randint_u(u[3], T - t + 1); cub[k][1] = cub[k][0] + t;
      cub[k][2] = randint_u(u[4], h - hh + 1); cub[k][3] = cub[k][2] + hh;
      cub[k][4] = randint_u(u[5], w - ww + 1); cub[k][5] = cub[k][4] + ww;
    }
    __syncthreads();
    for (int i = tid; i < T * hw; i += 256) {
      const int t = i / hw, pp = i - t * hw, ph = pp / w, pw = pp - ph * w;
      int c = 0;
      for (int k = 0; k < T; ++k)
        c |= (t >= cub[k][0] && t < cub[k][1] && ph >= cub[k][2] && ph < cub[k][3] && pw >= cub[k][4] && pw < cub[k][5]) ? 1 : 0;
      cov[((size_t)b * T + t) * hw + pp] = (uint8_t)c;
    }
  }
}

#define ST reinterpret_cast<hipStream_t>(stream)

extern "C" int vmvm_masking(int64_t* txt, int64_t* ans_mtm, uint8_t* cov, const float* u_type, const float* u_txt, const float* u_rm,
                            const float* u_bm, const int32_t* types, int32_t n_types, int32_t has_bm, int32_t B, int32_t X, int32_t T, int32_t h,
                            int32_t w, float p, int32_t cls, int32_t sep, int32_t pad, int32_t mask_id, void* stream) {
  if (!txt || !ans_mtm || !cov || !u_type || !u_txt || !u_rm || !u_bm || !types || n_types <= 0 || B <= 0 || X <= 0 || T <= 0 || h <= 0 || w <= 0)
    return VMVM_EINVAL;
  if (T > 64) return VMVM_ENOSUPPORT;
  if (has_bm && (h * 2 / 3 < 2 || w * 2 / 3 < 2)) return VMVM_EINVAL;        // numpy's randint(1, 1) raises in the reference as well
  hipLaunchKernelGGL(masking_kernel, dim3(B), dim3(256), 0, ST, txt, ans_mtm, cov, u_type, u_txt, u_rm, u_bm, types, n_types, X, T, h, w, p,
                     cls, sep, pad, mask_id);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

extern "C" int vmvm_patch_im2col(const float* img, const uint8_t* cov, void* cols, int32_t B, int32_t T, int32_t H, int32_t W, void* stream) {
  if (!img || !cols || B <= 0 || T <= 0 || (H & 3) || (W & 3)) return VMVM_EINVAL;
  if (cov && ((H & 31) || (W & 31))) return VMVM_EINVAL;
  const long n = (long)B * T * (H / 4) * (W / 4) * 6;
  hipLaunchKernelGGL(im2col_kernel, dim3(nblk(n, 256)), dim3(256), 0, ST, img, cov, reinterpret_cast<u16*>(cols), B, T, H, W);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_encvideo_assemble(const void* fc_out, const float* cls, const float* pos, const float* len, void* out,
                                      int32_t B, int32_t T, int32_t hw, int32_t Hd, void* stream) {
  if (!fc_out || !cls || !pos || !len || !out || (Hd & 7)) return VMVM_EINVAL;
  const long n = (long)B * T * (1 + hw) * (Hd / 8);
  hipLaunchKernelGGL(encvideo_assemble_kernel, dim3(nblk(n, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(fc_out), cls, pos, len,
                     reinterpret_cast<u16*>(out), B, T, hw, Hd);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_encvideo_assemble_bwd(const void* dpre, void* d_fc_out, float* dcls, float* dpos, float* dlen,
                                          int32_t B, int32_t T, int32_t hw, int32_t Hd, void* stream) {
  if (!dpre || !d_fc_out || !dcls || !dpos || !dlen || (Hd & 7)) return VMVM_EINVAL;
  const u16* dp = reinterpret_cast<const u16*>(dpre);
  u16* df = reinterpret_cast<u16*>(d_fc_out);
  hipLaunchKernelGGL(encvideo_assemble_bwd_kernel, dim3(nblk((long)B * T * hw * (Hd / 8), 256)), dim3(256), 0, ST, dp, df, dcls, dpos, dlen, B, T, hw, Hd, 0);
  const int ncg = (Hd + 63) / 64;
  hipLaunchKernelGGL((encvideo_embed_grad_kernel<1>), dim3((1 + hw) * ncg), dim3(256), 0, ST, dp, dcls, dpos, dlen, B, T, hw, Hd);
  hipLaunchKernelGGL((encvideo_embed_grad_kernel<2>), dim3(T * ncg), dim3(256), 0, ST, dp, dcls, dpos, dlen, B, T, hw, Hd);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_bert_embed(const int64_t* txt, const float* word, const float* pos, const float* type0, void* out,
                               int32_t B, int32_t X, int32_t Hd, void* stream) {
  if (!txt || !word || !pos || !type0 || !out || (Hd & 7)) return VMVM_EINVAL;
  hipLaunchKernelGGL(bert_embed_kernel, dim3(nblk((long)B * X * (Hd / 8), 256)), dim3(256), 0, ST, txt, word, pos, type0, reinterpret_cast<u16*>(out), B, X, Hd);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_bert_embed_bwd(const int64_t* txt, const void* dsum, float* dword, float* dpos, float* dtype0,
                                   int32_t B, int32_t X, int32_t Hd, void* stream) {
  if (!txt || !dsum || !dword || !dpos || !dtype0) return VMVM_EINVAL;
  if ((long)B * X * 4 > 48 * 1024) return VMVM_ENOSUPPORT;      // the batch's token ids live in LDS
  const int ncg = (Hd + 63) / 64;
  hipLaunchKernelGGL(bert_embed_bwd_kernel, dim3(B * X + X * ncg + ncg), dim3(1024), (size_t)B * X * sizeof(int), ST, txt, reinterpret_cast<const u16*>(dsum), dword, dpos, dtype0, B, X, Hd);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_count_valid(const int64_t* target, int32_t M, float* n_valid, void* stream) {
  if (!target || !n_valid || M <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(256), 0, ST, target, M, n_valid);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_cross_entropy(const float* logits, int32_t ld, int32_t M, int32_t V, const int64_t* target, const float* n_valid,
                                  float* loss_sum, void* dlogits, int32_t ld_d, void* stream) {
  if (!logits || !target || !n_valid || !loss_sum || M <= 0 || V <= 0 || ld < V) return VMVM_EINVAL;
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(M), dim3(256), 0, ST, logits, ld, V, target, n_valid, loss_sum, reinterpret_cast<u16*>(dlogits), ld_d);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
// VTM head: cross entropy of the (B, O) logit matrix against column 0 (main_pretrain.py:262-263, 566) with its gradient kept in f32 --
// the positive's and the negatives' terms of a clip nearly cancel, a bf16-rounded softmax adds noise of the size of the signal.
// One workgroup, fixed summation order (the loss is bit-reproducible).
__global__ __launch_bounds__(256) void vtm_ce_kernel(const float* __restrict__ lg, int B, int O, float* loss_sum, float* dlg) {
  __shared__ float red[256];
  float acc = 0.f;
  const float invB = 1.0f / (float)B;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float* l = lg + (size_t)i * O;
    float m = l[0];
    for (int j = 1; j < O; ++j) m = fmaxf(m, l[j]);
    float s = 0.f;
    for (int j = 0; j < O; ++j) s += __expf(l[j] - m);
    const float inv = 1.0f / s;
    for (int j = 0; j < O; ++j) dlg[(size_t)i * O + j] = (__expf(l[j] - m) * inv - (j == 0 ? 1.0f : 0.0f)) * invB;
    acc += (m + __logf(s) - l[0]) * invB;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss_sum += red[0];
}
extern "C" int vmvm_vtm_ce(const float* logits, int32_t B, int32_t O, float* loss_sum, float* dlogits, void* stream) {
  if (!logits || !loss_sum || !dlogits || B <= 0 || O <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(vtm_ce_kernel, dim3(1), dim3(256), 0, ST, logits, B, O, loss_sum, dlogits);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_pixel_l1(const void* pred, const float* img, const uint8_t* cov, const float* mask_sum, float* loss_sum, void* dpred,
                             int32_t B, int32_t T, int32_t h, int32_t w, int32_t ps, int32_t channels, float inv_div, void* stream) {
  if (!pred || !img || !cov || !mask_sum || !loss_sum || !dpred || (ps & 3) || channels <= 0 || ((channels * ps * ps) & 7)) return VMVM_EINVAL;
  hipLaunchKernelGGL(pixel_l1_kernel, dim3(B * T * h * w), dim3(256), 0, ST, reinterpret_cast<const u16*>(pred), img, cov, mask_sum, loss_sum,
                     reinterpret_cast<u16*>(dpred), B, T, h, w, ps, channels, inv_div);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_feature_l1(const void* pred, const void* target, const uint8_t* cov, const float* mask_sum, float inv_div, float* loss_sum,
                               void* dpred, int32_t M, int32_t C, void* stream) {
  if (!pred || !target || !cov || !mask_sum || !loss_sum || !dpred || M <= 0 || C <= 0 || (C & 7)) return VMVM_EINVAL;
  hipLaunchKernelGGL(feature_l1_kernel, dim3(M), dim3(256), 0, ST, reinterpret_cast<const u16*>(pred), reinterpret_cast<const u16*>(target), cov,
                     mask_sum, inv_div, loss_sum, reinterpret_cast<u16*>(dpred), C);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_rowdot(const void* hid, int32_t M, int32_t K, const float* w, const float* b, float inv_temp, float* out, void* stream) {
  if (!hid || !w || !b || !out || M <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(rowdot_kernel, dim3(nblk((long)M * 64, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(hid), M, K, w, b, inv_temp, out);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_rowdot_bwd(const void* hid, int32_t M, int32_t K, const float* w, const float* dout, float inv_temp, void* dhid,
                               float* dw, float* db, int32_t relu_mask, void* stream) {
  if (!hid || !w || !dout || !dhid || !dw || !db) return VMVM_EINVAL;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(nblk(K, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(hid), M, K, w, dout, inv_temp,
                     reinterpret_cast<u16*>(dhid), dw, db, relu_mask);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_cast_bf16_to_fp8(const void* src, void* dst, int64_t n, float scale, void* stream) {
  if (!src || !dst || n <= 0 || (n & 7)) return VMVM_EINVAL;
  int grid = nblk(n / 8, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cast_fp8_kernel, dim3(grid), dim3(256), 0, ST, reinterpret_cast<const u16*>(src), reinterpret_cast<uint8_t*>(dst), (long)(n / 8), scale);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(cast_kernel, dim3(nblk((n + 7) / 8, 256)), dim3(256), 0, ST, src, reinterpret_cast<u16*>(dst), (long)n);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return VMVM_EINVAL;
  if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return VMVM_EINVAL;
  hipLaunchKernelGGL(uncast_kernel, dim3(nblk((n + 7) / 8, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(src), dst, (long)n);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream) {
  if (!a || !b || !out || n <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(add_kernel, dim3(nblk((n + 7) / 8, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(a), reinterpret_cast<const u16*>(b),
                     reinterpret_cast<u16*>(out), (long)n);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_gather_rows_bf16(const void* src, int32_t ld_src, const int32_t* idx, void* dst, int32_t ld_dst, int32_t M, int32_t C,
                                     int32_t rows_out_per_batch, int32_t rows_in_per_batch, void* stream) {
  if (!src || !idx || !dst || M <= 0 || (C & 7) || (ld_src & 7) || (ld_dst & 7)) return VMVM_EINVAL;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk((long)M * (C / 8), 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(src), ld_src, idx,
                     reinterpret_cast<u16*>(dst), ld_dst, (long)M, C, rows_out_per_batch, rows_in_per_batch);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_scatter_add_rows_bf16(const void* src, int32_t ld_src, const int32_t* idx, float* dst, int32_t ld_dst, int32_t M, int32_t C,
                                          void* stream) {
  if (!src || !idx || !dst || M <= 0 || (C & 7) || (ld_src & 7)) return VMVM_EINVAL;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(nblk((long)M * (C / 8), 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(src), ld_src, idx,
                     dst, ld_dst, (long)M, C);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_pool_grad_bf16(const void* g1, const void* g2, const void* g3, void* out, int32_t B, int32_t O, int32_t Lv, int32_t X, int32_t Hd,
                                   const int32_t* txt_off, const int32_t* txt_list, void* stream) {
  if (!g1 || !g2 || !out || !txt_off || !txt_list || B <= 0 || O <= 0 || Lv <= 0 || X <= 0 || Hd <= 0 || (Hd & 7)) return VMVM_EINVAL;
  const long n = (long)B * (Lv + X) * (Hd / 8);
  hipLaunchKernelGGL(pool_grad_kernel, dim3(nblk(n, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(g1), reinterpret_cast<const u16*>(g2),
                     reinterpret_cast<const u16*>(g3), reinterpret_cast<u16*>(out), B, O, Lv, X, Hd, txt_off, txt_list);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_gelu_bwd_bf16(const void* dy, const void* u, void* out, int64_t n, void* stream) {
  if (!dy || !u || !out || n <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nblk(n, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(dy), reinterpret_cast<const u16*>(u),
                     reinterpret_cast<u16*>(out), (long)n);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return VMVM_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3(nblk((n + 7) / 8, 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(x), reinterpret_cast<u16*>(y),
                     (long)n, p, seed, offset);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
static int colsum_launch(const void* X, int32_t M, int32_t N, int32_t ldx, const float* row_scale, int32_t rows_per_scale, float all_scale, float* out,
                         int32_t accumulate, void* ws, int64_t ws_bytes, void* stream) {
  if (!X || !out || M <= 0 || N <= 0 || (N & 7) || (ldx & 7)) return VMVM_EINVAL;
  if (!accumulate && hipMemsetAsync(out, 0, (size_t)N * 4, ST) != hipSuccess) return VMVM_EHIP;
  int gy = (M + 32 * 16 - 1) / (32 * 16);
  if (gy > 256) gy = 256;
  if (gy < 1) gy = 1;
  float* part = (ws && ws_bytes >= (int64_t)gy * N * (int64_t)sizeof(float)) ? reinterpret_cast<float*>(ws) : nullptr;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, gy), dim3(256), 0, ST, reinterpret_cast<const u16*>(X), M, N, ldx, row_scale, rows_per_scale,
                     all_scale, out, part);
  VMVM_CHECK_LAUNCH();
  if (part) {
    hipLaunchKernelGGL(colsum_fin_kernel, dim3(nblk(N, 64)), dim3(1024), 0, ST, part, gy, N, out);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}
extern "C" int vmvm_colsum_bf16(const void* X, int32_t M, int32_t N, int32_t ldx, const float* row_scale, int32_t rows_per_scale, float* out,
                                int32_t accumulate, void* stream) {
  return colsum_launch(X, M, N, ldx, row_scale, rows_per_scale, 1.f, out, accumulate, nullptr, 0, stream);
}
extern "C" int vmvm_colsum_bf16_ws(const void* X, int32_t M, int32_t N, int32_t ldx, const float* row_scale, int32_t rows_per_scale, float* out,
                                   int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream) {
  return colsum_launch(X, M, N, ldx, row_scale, rows_per_scale, 1.f, out, accumulate, workspace, workspace_bytes, stream);
}
extern "C" int64_t vmvm_colsum_workspace_size(int32_t M, int32_t N) {
  if (M <= 0 || N <= 0) return VMVM_EINVAL;
  int64_t gy = ((int64_t)M + 32 * 16 - 1) / (32 * 16);
  if (gy > 256) gy = 256;
  return gy * N * (int64_t)sizeof(float);
}
// out[n] += scale * sum_m X[m][n]: the separate-pass form of vmvm_gemm_desc.colsum / colsum_scale (gemm.hip, shapes off the fused build;
// runs in front of that GEMM's own kernels, so it may use the GEMM's workspace for its partial rows)
int vmvm_colsum_scaled(const void* X, int32_t M, int32_t N, int32_t ldx, float scale, float* out, void* ws, int64_t ws_bytes, void* stream) {
  return colsum_launch(X, M, N, ldx, nullptr, 0, scale, out, 1, ws, ws_bytes, stream);
}
extern "C" int vmvm_transpose_batched_bf16(const void* src, void* dst, const int32_t* table, int32_t ntiles, void* stream) {
  if (!src || !dst || !table || ntiles <= 0) return VMVM_EINVAL;
  const int grid = ntiles < 256 * 8 ? ntiles : 256 * 8;       // 8 workgroups per CU (17 KiB of LDS, 256 threads each)
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(grid), dim3(256), 0, ST, reinterpret_cast<const u16*>(src), reinterpret_cast<u16*>(dst),
                     reinterpret_cast<const int4*>(table), ntiles);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_expand_batch_map(const int32_t* map, int32_t len, const int32_t* list, int32_t n, int32_t stride, int32_t* out, void* stream) {
  if (!map || !list || !out || len <= 0 || n <= 0) return VMVM_EINVAL;
  hipLaunchKernelGGL(expand_batch_map_kernel, dim3(nblk((long)n * len, 256)), dim3(256), 0, ST, map, len, list, n, stride, out);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_invert_map(const int32_t* src, int32_t n_src, int32_t* out, int32_t n_out, void* stream) {
  if (!src || !out || n_src <= 0 || n_out <= 0) return VMVM_EINVAL;
  if (hipMemsetAsync(out, 0xff, (size_t)n_out * sizeof(int32_t), ST) != hipSuccess) return VMVM_EHIP;
  hipLaunchKernelGGL(invert_map_kernel, dim3(nblk((long)n_src, 256)), dim3(256), 0, ST, src, n_src, out, n_out);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_copy_batches_bf16(const void* src, int32_t ld_src, void* dst, int32_t ld_dst, const int32_t* list, int32_t n, int32_t rows_per_batch,
                                      int32_t C, void* stream) {
  if (!src || !dst || !list || n <= 0 || rows_per_batch <= 0 || C <= 0 || (C & 7) || (ld_src & 7) || (ld_dst & 7)) return VMVM_EINVAL;
  hipLaunchKernelGGL(copy_batches_kernel, dim3(nblk((long)n * rows_per_batch * (C / 8), 256)), dim3(256), 0, ST, reinterpret_cast<const u16*>(src), ld_src,
                     reinterpret_cast<u16*>(dst), ld_dst, list, n, rows_per_batch, C);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_attn_query_row_fwd(const void* q, int32_t ld_q, const void* kv, int32_t ld_kv, int32_t k_off, int32_t v_off, const uint8_t* keymask,
                                       void* out, int32_t ld_out, float* probs, float* probs_drop, int32_t nseq, int32_t L, int32_t heads, int32_t head_dim,
                                       float scale, float dropout_p, uint64_t seed, uint64_t offset, void* stream) {
  if (!q || !kv || !out || !probs || !probs_drop || nseq <= 0 || L <= 0 || heads <= 0) return VMVM_EINVAL;
  if (head_dim != 64 && head_dim != 32) return VMVM_ENOSUPPORT;
  if (L > 8192 || (ld_kv & 7) || (k_off & 7) || (v_off & 7)) return VMVM_ENOSUPPORT;
  if (dropout_p < 0.f || dropout_p >= 1.f) return VMVM_EINVAL;
  const size_t sm = (size_t)(head_dim + ((L + 3) & ~3) + 4 * head_dim) * sizeof(float);
  if (head_dim == 64)
    hipLaunchKernelGGL(attn_qrow_fwd_kernel<64>, dim3(nseq * heads), dim3(256), sm, ST, reinterpret_cast<const u16*>(q), ld_q, reinterpret_cast<const u16*>(kv), ld_kv,
                       k_off, v_off, keymask, reinterpret_cast<u16*>(out), ld_out, probs, probs_drop, L, heads, scale, dropout_p, seed, offset);
  else
    hipLaunchKernelGGL(attn_qrow_fwd_kernel<32>, dim3(nseq * heads), dim3(256), sm, ST, reinterpret_cast<const u16*>(q), ld_q, reinterpret_cast<const u16*>(kv), ld_kv,
                       k_off, v_off, keymask, reinterpret_cast<u16*>(out), ld_out, probs, probs_drop, L, heads, scale, dropout_p, seed, offset);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_attn_query_row_bwd(const void* dout, int32_t ld_dout, const void* q, int32_t ld_q, const void* kv, int32_t ld_kv, int32_t k_off, int32_t v_off,
                                       const float* probs, const float* probs_drop, void* dq, int32_t ld_dq, void* dkv, int32_t ld_dkv, int32_t nseq, int32_t L,
                                       int32_t heads, int32_t head_dim, float scale, void* stream) {
  if (!dout || !q || !kv || !probs || !probs_drop || !dq || !dkv || nseq <= 0 || L <= 0 || heads <= 0) return VMVM_EINVAL;
  if (head_dim != 64 && head_dim != 32) return VMVM_ENOSUPPORT;
  if (L > 8192 || (ld_kv & 7) || (ld_dkv & 7) || (k_off & 7) || (v_off & 7)) return VMVM_ENOSUPPORT;
  const size_t sm = (size_t)(2 * head_dim + ((L + 3) & ~3) + 4 * head_dim) * sizeof(float);
  if (head_dim == 64)
    hipLaunchKernelGGL(attn_qrow_bwd_kernel<64>, dim3(nseq * heads), dim3(256), sm, ST, reinterpret_cast<const u16*>(dout), ld_dout, reinterpret_cast<const u16*>(q), ld_q,
                       reinterpret_cast<const u16*>(kv), ld_kv, k_off, v_off, probs, probs_drop, reinterpret_cast<u16*>(dq), ld_dq, reinterpret_cast<u16*>(dkv), ld_dkv,
                       L, heads, scale);
  else
    hipLaunchKernelGGL(attn_qrow_bwd_kernel<32>, dim3(nseq * heads), dim3(256), sm, ST, reinterpret_cast<const u16*>(dout), ld_dout, reinterpret_cast<const u16*>(q), ld_q,
                       reinterpret_cast<const u16*>(kv), ld_kv, k_off, v_off, probs, probs_drop, reinterpret_cast<u16*>(dq), ld_dq, reinterpret_cast<u16*>(dkv), ld_dkv,
                       L, heads, scale);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_sumsq_f32(const float* g, int64_t n, float* out_accum, void* workspace, uint64_t workspace_bytes, void* stream) {
  if (!g || !out_accum || n <= 0) return VMVM_EINVAL;
  int grid = nblk((n + 3) / 4, 256);
  if (grid > 2048) grid = 2048;
  float* partial = (workspace && workspace_bytes >= (uint64_t)grid * sizeof(float)) ? reinterpret_cast<float*>(workspace) : nullptr;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, ST, g, (long)n, out_accum, partial);
  VMVM_CHECK_LAUNCH();
  if (partial) {
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, ST, partial, grid, out_accum);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}
extern "C" int vmvm_adamw(const vmvm_adamw_desc* d, void* stream) {
  if (!d || !d->param || !d->grad || !d->m || !d->v || d->n <= 0) return VMVM_EINVAL;
  // Round 6: EIGHT float4 groups per thread and iteration (32 loads in flight per thread before the first use) and non-temporal loads /
  // stores -- every byte is touched once per step.  Measured on two boxes (tools/scratch/bench_adamw.py, 197 M elements, 30 B each):
  // U = 1 cached 4.65 TB/s (rounds 1-5), U = 1 nt 4.87, U = 2 nt 4.93, U = 4 nt 5.27-5.64, U = 8 nt 5.41-5.49, U = 6 nt 4.73 (an odd
  // group count leaves the workgroups' chunks mis-aligned to the 8-KiB interleave).
  constexpr int U = 8;
  int grid = nblk((d->n + 3) / 4, 256 * U);
  if (grid > 4096 / U) grid = 4096 / U;
  hipLaunchKernelGGL((adamw_kernel<U, true>), dim3(grid), dim3(256), 0, ST, *d);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}
extern "C" int vmvm_probe_tr16(int32_t* out, void* stream) {
  if (!out) return VMVM_EINVAL;
  hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, ST, out);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

// scratch of the fixed-order gradient-norm sum (one f32 partial per workgroup)
extern "C" int64_t vmvm_sumsq_workspace_size(int64_t n) {
  if (n <= 0) return VMVM_EINVAL;
  int64_t grid = (((n + 3) / 4) + 255) / 256;
  if (grid > 2048) grid = 2048;
  return grid * (int64_t)sizeof(float);
}

