// attention.hip -- fused short-sequence attention (forward + backward) for gfx950.
//
// Both attention flavours of the VIOLETv2 step have SHORT sequences whose whole K/V for one
// (sequence, head) fits in LDS: Video-Swin windows (N = 392 tokens, head_dim 32, relative-position
// bias + shift mask) and the BERT fusion encoder (L = 432, head_dim 64, key-padding mask, dropout).
// One workgroup = one (sequence, head).  A wave owns 16 query rows at a time and keeps the ENTIRE
// score row block S^T (keys x 16 queries) in MFMA accumulators (<= 28 tiles of 16 keys), so the
// softmax is exact two-pass in registers -- S/P never touch LDS or HBM.
//   S^T tile = mfma(K_tile[16 keys x hd], Q^T)          lane: query = lane&15, keys 4*(lane>>4)+j
//   P stays in registers and is fed straight back as the B operand of  O^T = mfma(V^T, P^T):
//   the MFMA k-slot order is free as long as both operands agree, so the accumulator layout of two
//   adjacent S^T tiles IS a valid 32-deep operand (no cross-lane movement, no LDS round trip).
//   K/V/Q/dO live in LDS once, row-major ([token][hd], 16-byte-chunk XOR swizzle): operands with k = hd are
//   ds_read_b128 fragments, operands with k = tokens (V^T, K^T, Q^T, dO^T) are produced by the gfx950
//   transposing read ds_read_b64_tr_b16 from the SAME image -- no transposed copies, no scatter writes.
// Relative-position bias is evaluated arithmetically: idx(i,j) = rc[i]-rc[j]+rc0 into the per-head
// table slice held in LDS; the shift mask is region[i] != region[j] ? -100 : 0.
// Backward is two kernels (no atomics on dQ/dK/dV): A) per query tile: dQ (+ the bias-table
// gradient, accumulated in an LDS copy of the table, one global atomic per entry per workgroup),
// B) per key tile: dK, dV.  Probabilities are recomputed from the saved log-sum-exp.
// Attention-prob dropout: Philox4x32-7 per 4x4 (query,key) block, 8 random bits per element
// (p_eff = round(256p)/256), the same block is addressed row-wise by fwd/A and column-wise by B.
#include "common.h"

namespace {

constexpr float NEG_INF = -__builtin_huge_valf();

struct Smem {
  int nt, nt2, lp16, lp32;
  int off_a, off_b, off_rc, off_reg, off_tab, off_dtab, off_lse, off_delta, total;
};

// which: 0 fwd (a=K, b=V) ; 1 bwdA (a=K, b=V, dtab) ; 2 bwdB (a=Q, b=dO, lse, delta)
__host__ __device__ inline Smem smem_layout(int L, int hd, int mode, int table_len, int which) {
  Smem s;
  s.nt = (L + 15) / 16; s.nt2 = (s.nt + 1) / 2; s.lp16 = s.nt * 16; s.lp32 = s.nt2 * 32;
  int o = 0;
  s.off_a = o; o += s.lp32 * hd * 2;
  s.off_b = o; o += s.lp32 * hd * 2;
  s.off_rc = o; o += s.lp32 * 4;
  s.off_reg = o; o += s.lp32;                                         // region (mode 0) or keymask (mode 1)
  s.off_tab = o; if (mode == 0) o += ((table_len + 3) & ~3) * 4;
  s.off_dtab = o; if (mode == 0 && which == 1) o += ((table_len + 3) & ~3) * 4;
  s.off_lse = o; if (which == 2) o += s.lp32 * 4;
  s.off_delta = o; if (which == 2) o += s.lp32 * 4;
  s.total = (o + 15) & ~15;
  return s;
}

// XCD-aware bijective block remap: XCD x (= blockIdx % 8) gets a contiguous run of logical ids
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
}

template <int HD>
__device__ __forceinline__ int k_off_swz(int row, int chunk) {   // row-major [row][HD] bf16, 16B chunk swizzle
  if (HD == 32) return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4);
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// fill row-major swizzled [rows][HD] from global rows (zero beyond L) with direct-to-LDS DMA (buffer_load ... lds): no VGPR
// round trip, all requests of a thread in flight at once.  The LDS image is lane-linear per wave instruction, so the chunk
// swizzle is applied to the SOURCE column; rows >= L fall beyond the descriptor's num_records and read as zero.
// Caller must `s_waitcnt vmcnt(0)` + barrier before reading.
template <int HD>
__device__ __forceinline__ void fill_rowmajor(unsigned char* dst, const u16* src, int ld, int L, int rows, int tid, int nthreads) {
  constexpr int CPR = HD / 8;
  const unsigned bytes = (unsigned)(((size_t)(L - 1) * ld + HD) * 2);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(src), 0, (int)bytes, 0x00020000);
  const int total = rows * CPR;
  const int wave_base = tid & ~63;
  typedef __attribute__((address_space(3))) void lds_void;
  for (int i0 = 0; i0 < total; i0 += nthreads) {
    const int u = i0 + tid;
    if (u < total) {
      const int row = u / CPR, chs = u - row * CPR;
      const int ch = (HD == 32) ? (chs ^ ((row >> 2) & 3)) : (chs ^ ((row >> 1) & 7));
      const unsigned goff = (unsigned)(((size_t)row * ld + ch * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + (size_t)(i0 + wave_base) * 16), 16, goff, 0, 0, 0);
    }
  }
}
__device__ __forceinline__ void fill_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bf16x8 load_frag_global(const u16* p, bool valid) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (valid) v = *reinterpret_cast<const uint4*>(p);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 frag_from_f32(const float* a, const float* b) {
  uint4 v = make_uint4(pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3]));
  return __builtin_bit_cast(bf16x8, v);
}
// operand with k = tokens from the row-major swizzled image: row/col index d = dt*16 + (lane&15); k-slots 0-3 = tokens
// tok_a + 0..3, k-slots 4-7 = tokens tok_b + 0..3 (tok_a/b = 16*tile + 4*(lane>>4)) -> two transposing 4x16 block reads.
template <int HD>
__device__ __forceinline__ bf16x8 frag_tokens(const unsigned char* img, int dt, int tok_a, int tok_b, int r) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const int chunk = dt * 2 + ((r & 3) >> 1), sub = (r & 1) * 8;
  const int ra = tok_a + (r >> 2), rb = tok_b + (r >> 2);
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + k_off_swz<HD>(ra, chunk) + sub));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + k_off_swz<HD>(rb, chunk) + sub));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
template <int HD>
__device__ __forceinline__ bf16x8 frag_hd(const unsigned char* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(img + k_off_swz<HD>(row, chunk));
}

__device__ __forceinline__ uint32_t drop_thr8(float p) { return (uint32_t)(p * 256.f + 0.5f); }
// 4x4 block of 8-bit randoms for (query block qb = q/4, key block kb = key/4) of (seq,head) stream `sh`
__device__ __forceinline__ uint4 drop_block(uint64_t seed, uint64_t offset, uint32_t sh, uint32_t qb, uint32_t kb) {
  const uint64_t c = offset + (((uint64_t)sh << 32) | ((uint64_t)qb << 12) | kb);
  return philox4x32_7(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0xa77eu, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ uint32_t u4_get(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// ================================================================================================
// forward
// ================================================================================================
// NX > 0: the number of 16-key tiles is a compile-time constant (the step's shapes: 392 -> 25, 196 -> 13, 432 -> 27, 232 -> 15),
// so the fully unrolled tile loops carry no runtime guards / exec-mask juggling and only the LAST tile checks key < L.
// MASK = false: un-shifted window block (no region compare).
#define TILE_ON(t) (NX ? ((t) < NX) : ((t) < nt))
#define PAIR_ON(c) (NX ? ((c) < (NX + 1) / 2) : ((c) < nt2))
#define KEY_OK(t, key) ((NX && (t) < NX - 1) ? true : ((key) < L))
template <int HD, int MODE, int NT_MAX, int NW, int NX, bool MASK>
__global__ __launch_bounds__(NW * 64) void attn_fwd_kernel(const vmvm_attn_fwd_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const Smem sm = smem_layout(L, HD, MODE, p.table_len, 0);
  const int logical = xcd_remap(blockIdx.x, p.nseq * heads);
  const int seq = logical / heads, h = logical - seq * heads;
  const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
  unsigned char* Ksm = smem + sm.off_a;
  unsigned char* Vsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);

  fill_rowmajor<HD>(Ksm, qkv + p.k_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
  fill_rowmajor<HD>(Vsm, qkv + p.v_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
  fill_wait();
  if (MODE == 0) {
    for (int i = tid; i < sm.lp32; i += NW * 64) {
      rc[i] = i < L ? p.rc[i] : 0;
      reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
    }
    for (int i = tid; i < p.table_len; i += NW * 64) tab[i] = p.bias_table[(size_t)i * heads + h];
  } else {
    for (int i = tid; i < sm.lp32; i += NW * 64) reg[i] = (i < L) ? (p.keymask ? p.keymask[(size_t)seq * L + i] : 1) : 0;
  }
  __syncthreads();

  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const uint32_t thr8 = drop_thr8(p.dropout_p);
  const float keep = has_drop ? 256.f / (256.f - (float)thr8) : 1.f;
  const int nt = sm.nt, nt2 = sm.nt2;

  for (int qt = wave; qt < nt; qt += NW) {
    const int q = qt * 16 + r;
    const bool qv = q < L;
    const u16* qp = qkv + (size_t)q * p.ld_qkv + p.q_off + h * HD + g * 8;
    bf16x8 qf[HD / 32];
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) qf[s] = load_frag_global(qp + s * 32, qv);

    f32x4 acc[NT_MAX];
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (TILE_ON(t)) {
        const int row = t * 16 + r;
#pragma unroll
        for (int s = 0; s < HD / 32; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ksm + k_off_swz<HD>(row, s * 4 + g));
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc[t], 0, 0, 0);
        }
      }
    }
    // ---- scores: bias / masks, row max
    const int rcq = (MODE == 0) ? rc[qv ? q : 0] : 0;
    const int regq = (MODE == 0) ? reg[qv ? q : 0] : 0;
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t) {
      if (TILE_ON(t)) {
        const int key0 = t * 16 + g * 4;
        if (MODE == 0) {
          const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
          const uchar4 gk = *reinterpret_cast<const uchar4*>(reg + key0);
          const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
          const int gks[4] = {gk.x, gk.y, gk.z, gk.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = acc[t][j] + tab[rcq - rks[j] + p.rc0];
            if (MASK) s += (regq != gks[j] ? -100.f : 0.f);
            s = KEY_OK(t, key0 + j) ? s : NEG_INF;
            acc[t][j] = s; mx = fmaxf(mx, s);
          }
        } else {
          const uchar4 mk = *reinterpret_cast<const uchar4*>(reg + key0);
          const int mks[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float s = mks[j] ? acc[t][j] * p.scale : NEG_INF;
            acc[t][j] = s; mx = fmaxf(mx, s);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = NEG_INF;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT_MAX; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float e = __expf(acc[t][j] - mx); acc[t][j] = e; sum += e; }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (g == 0 && qv) p.lse[((size_t)seq * heads + h) * L + q] = mx + __logf(sum);
    if (has_drop) {
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        if (TILE_ON(t)) {
          const uint4 blk = drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q >> 2), (uint32_t)(t * 4 + g));
          const uint32_t w = u4_get(blk, q & 3);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][j] = (((w >> (8 * j)) & 0xffu) < thr8) ? 0.f : acc[t][j] * keep;
        }
      }
    }
    // ---- O^T = V^T P^T
    f32x4 o[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NT_MAX / 2; ++c) {
      if (PAIR_ON(c)) {
        float a[4] = {acc[2 * c][0], acc[2 * c][1], acc[2 * c][2], acc[2 * c][3]};
        float b[4] = {acc[2 * c + 1][0], acc[2 * c + 1][1], acc[2 * c + 1][2], acc[2 * c + 1][3]};
        const bf16x8 pf = frag_from_f32(a, b);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          const bf16x8 vf = frag_tokens<HD>(Vsm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    }
    if (qv) {
      const float inv = seq_scale / sum;
      u16* op = reinterpret_cast<u16*>(p.out) + ((size_t)seq * L + q) * p.ld_out + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt)
        *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
    }
  }
}

// ================================================================================================
// backward A: dQ (+ delta, + relative-position-bias table gradient)
// ================================================================================================
template <int HD, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_kernel(const vmvm_attn_bwd_desc pb, const int nchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const Smem sm = smem_layout(L, HD, MODE, p.table_len, 1);
  // persistent over sequences: workgroup (chunk, head) walks seq = chunk, chunk+nchunks, ... so the bias-table gradient
  // is accumulated in LDS across many windows and flushed with ONE global atomic per entry per workgroup.
  const int logical = xcd_remap(blockIdx.x, nchunks * heads);
  const int chunk = logical / heads, h = logical - chunk * heads;
  unsigned char* Ksm = smem + sm.off_a;
  unsigned char* Vsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);
  float* dtab = reinterpret_cast<float*>(smem + sm.off_dtab);
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const bool want_dtab = pb.dbias_table != nullptr;
  const uint32_t thr8 = drop_thr8(p.dropout_p);
  const float keep = has_drop ? 256.f / (256.f - (float)thr8) : 1.f;
  const int nt = sm.nt, nt2 = sm.nt2;
  if (MODE == 0) {
    for (int i = tid; i < sm.lp32; i += NW * 64) rc[i] = i < L ? p.rc[i] : 0;
    for (int i = tid; i < p.table_len; i += NW * 64) { tab[i] = p.bias_table[(size_t)i * heads + h]; dtab[i] = 0.f; }
  }

  for (int seq = chunk; seq < p.nseq; seq += nchunks) {
  const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
  const u16* dO = reinterpret_cast<const u16*>(pb.dout) + (size_t)seq * L * pb.ld_dout + h * HD;
  const u16* O = reinterpret_cast<const u16*>(p.out) + (size_t)seq * L * p.ld_out + h * HD;
  __syncthreads();                                  // every wave is done with the previous sequence's LDS image
  fill_rowmajor<HD>(Ksm, qkv + p.k_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
  fill_rowmajor<HD>(Vsm, qkv + p.v_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
  fill_wait();
  if (MODE == 0) {
    for (int i = tid; i < sm.lp32; i += NW * 64) reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
  } else {
    for (int i = tid; i < sm.lp32; i += NW * 64) reg[i] = (i < L) ? (p.keymask ? p.keymask[(size_t)seq * L + i] : 1) : 0;
  }
  __syncthreads();

  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  const float* lse_g = p.lse + ((size_t)seq * heads + h) * L;
  float* delta_g = pb.delta + ((size_t)seq * heads + h) * L;

  for (int qt = wave; qt < nt; qt += NW) {
    const int q = qt * 16 + r;
    const bool qv = q < L;
    bf16x8 qf[HD / 32], dof[HD / 32];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) {
      qf[s] = load_frag_global(qkv + (size_t)q * p.ld_qkv + p.q_off + h * HD + g * 8 + s * 32, qv);
      dof[s] = load_frag_global(dO + (size_t)q * pb.ld_dout + g * 8 + s * 32, qv);
      const bf16x8 of = load_frag_global(O + (size_t)q * p.ld_out + g * 8 + s * 32, qv);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[s][e] * (float)of[e];
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);             // delta_q = sum_d dO'[q,d] * out'[q,d]
    if (g == 0 && qv) delta_g[q] = dl;
    const float lse = qv ? lse_g[q] : 0.f;
    const int rcq = (MODE == 0) ? rc[qv ? q : 0] : 0;
    const int regq = (MODE == 0) ? reg[qv ? q : 0] : 0;

    f32x4 dq[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nt2; ++c) {
      float ds[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * c + u;
#pragma unroll
        for (int j = 0; j < 4; ++j) ds[u][j] = 0.f;
        if (t < nt) {
          const int row = t * 16 + r;
          f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < HD / 32; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ksm + k_off_swz<HD>(row, s * 4 + g));
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], s4, 0, 0, 0);
            const bf16x8 vf = frag_hd<HD>(Vsm, row, s * 4 + g);
            dp4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[s], dp4, 0, 0, 0);
          }
          const int key0 = t * 16 + g * 4;
          uint32_t w = 0;
          if (has_drop) {
            const uint4 blk = drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q >> 2), (uint32_t)(t * 4 + g));
            w = u4_get(blk, q & 3);
          }
          if (MODE == 0) {
            const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
            const uchar4 gk = *reinterpret_cast<const uchar4*>(reg + key0);
            const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
            const int gks[4] = {gk.x, gk.y, gk.z, gk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int bi = rcq - rks[j] + p.rc0;
              const float s = s4[j] + tab[bi] + (regq != gks[j] ? -100.f : 0.f);
              const float pr = (key0 + j < L && qv) ? __expf(s - lse) : 0.f;
              const float d = pr * (dp4[j] * seq_scale - dl);
              ds[u][j] = d;
              if (want_dtab && pr != 0.f) atomicAdd(&dtab[bi], d);
            }
          } else {
            const uchar4 mk = *reinterpret_cast<const uchar4*>(reg + key0);
            const int mks[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float pr = (mks[j] && qv) ? __expf(s4[j] * p.scale - lse) : 0.f;
              float dpj = dp4[j] * seq_scale;
              if (has_drop) dpj = (((w >> (8 * j)) & 0xffu) < thr8) ? 0.f : dpj * keep;
              ds[u][j] = pr * (dpj - dl);
            }
          }
        }
      }
      const bf16x8 dsf = frag_from_f32(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        const bf16x8 kf = frag_tokens<HD>(Ksm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, dq[dt], 0, 0, 0);
      }
    }
    if (qv) {
      u16* dqp = reinterpret_cast<u16*>(pb.dqkv) + ((size_t)seq * L + q) * pb.ld_dqkv + p.q_off + h * HD + g * 4;
      const float sc = p.scale;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt)
        *reinterpret_cast<uint2*>(dqp + dt * 16) = make_uint2(pack_bf2(dq[dt][0] * sc, dq[dt][1] * sc), pack_bf2(dq[dt][2] * sc, dq[dt][3] * sc));
    }
  }
  }   // sequences
  if (MODE == 0 && want_dtab) {
    __syncthreads();
    for (int i = tid; i < p.table_len; i += NW * 64) {
      const float v = dtab[i];
      if (v != 0.f) atomicAdd(pb.dbias_table + (size_t)i * heads + h, v);
    }
  }
}

// ================================================================================================
// backward A' (window mode): dQ + bias-table gradient with the dS tiles accumulated IN REGISTERS across windows.
// A wave owns one fixed query tile (16 queries) for the whole launch and walks the windows of its chunk; the 16 x N
// block of dS it produces per window is summed into racc[] (same accumulator layout as the forward scores), so the
// scatter into the relative-position table (rc[i]-rc[j]+rc0) happens once per workgroup instead of once per window:
// per-element LDS atomics were 50% of the backward attention time.
// ================================================================================================
template <int NT_MAX, int NW, int NX, bool MASK>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dq_win_kernel(const vmvm_attn_bwd_desc pb, const int nchunks, const int nqg) {
  constexpr int HD = 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const Smem sm = smem_layout(L, HD, 0, p.table_len, 1);
  const int logical = xcd_remap(blockIdx.x, nchunks * heads * nqg);
  const int chunk = logical / (heads * nqg);
  const int rem = logical - chunk * heads * nqg;
  const int h = rem / nqg, qg = rem - h * nqg;
  unsigned char* Ksm = smem + sm.off_a;
  unsigned char* Vsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);
  float* dtab = reinterpret_cast<float*>(smem + sm.off_dtab);
  const bool want_dtab = pb.dbias_table != nullptr;
  const int nt = sm.nt, nt2 = sm.nt2;
  for (int i = tid; i < sm.lp32; i += NW * 64) rc[i] = i < L ? p.rc[i] : 0;
  for (int i = tid; i < p.table_len; i += NW * 64) { tab[i] = p.bias_table[(size_t)i * heads + h]; dtab[i] = 0.f; }

  const int qt = qg * NW + wave;
  const int q = qt * 16 + r;
  const bool qv = (qt < nt) && (q < L);
  f32x4 racc[NT_MAX];
#pragma unroll
  for (int t = 0; t < NT_MAX; ++t) racc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int seq = chunk; seq < p.nseq; seq += nchunks) {
    const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
    const u16* dO = reinterpret_cast<const u16*>(pb.dout) + (size_t)seq * L * pb.ld_dout + h * HD;
    const u16* O = reinterpret_cast<const u16*>(p.out) + (size_t)seq * L * p.ld_out + h * HD;
    __syncthreads();
    fill_rowmajor<HD>(Ksm, qkv + p.k_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
    fill_rowmajor<HD>(Vsm, qkv + p.v_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
  fill_wait();
    for (int i = tid; i < sm.lp32; i += NW * 64) reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
    __syncthreads();
    if (qt >= nt) continue;                                   // wave-uniform; barriers above are still reached by all waves
    const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
    const bf16x8 qf = load_frag_global(qkv + (size_t)q * p.ld_qkv + p.q_off + h * HD + g * 8, qv);
    const bf16x8 dof = load_frag_global(dO + (size_t)q * pb.ld_dout + g * 8, qv);
    const bf16x8 of = load_frag_global(O + (size_t)q * p.ld_out + g * 8, qv);
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)dof[e] * (float)of[e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (g == 0 && qv) pb.delta[((size_t)seq * heads + h) * L + q] = dl;
    const float lse = qv ? p.lse[((size_t)seq * heads + h) * L + q] : 0.f;
    const int rcq = rc[qv ? q : 0], regq = reg[qv ? q : 0];
    f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < NT_MAX / 2; ++c) {
      if (PAIR_ON(c)) {
        float ds[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * c + u;
#pragma unroll
          for (int j = 0; j < 4; ++j) ds[u][j] = 0.f;
          if (TILE_ON(t)) {
            const int row = t * 16 + r;
            f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp4 = f32x4{0.f, 0.f, 0.f, 0.f};
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Ksm, row, g), qf, s4, 0, 0, 0);
            dp4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_hd<HD>(Vsm, row, g), dof, dp4, 0, 0, 0);
            const int key0 = t * 16 + g * 4;
            const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
            const uchar4 gk = *reinterpret_cast<const uchar4*>(reg + key0);
            const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
            const int gks[4] = {gk.x, gk.y, gk.z, gk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float sv = s4[j] + tab[rcq - rks[j] + p.rc0];
              if (MASK) sv += (regq != gks[j] ? -100.f : 0.f);
              const float pr = (KEY_OK(t, key0 + j) && qv) ? __expf(sv - lse) : 0.f;
              const float d = pr * (dp4[j] * seq_scale - dl);
              ds[u][j] = d;
              racc[t][j] += d;
            }
          }
        }
        const bf16x8 dsf = frag_from_f32(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tokens<HD>(Ksm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r), dsf, dq[dt], 0, 0, 0);
      }
    }
    if (qv) {
      u16* dqp = reinterpret_cast<u16*>(pb.dqkv) + ((size_t)seq * L + q) * pb.ld_dqkv + p.q_off + h * HD + g * 4;
      const float sc = p.scale;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
        *reinterpret_cast<uint2*>(dqp + dt * 16) = make_uint2(pack_bf2(dq[dt][0] * sc, dq[dt][1] * sc), pack_bf2(dq[dt][2] * sc, dq[dt][3] * sc));
    }
  }
  if (want_dtab) {
    if (qv) {
      const int rcq = rc[q];
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        if (TILE_ON(t)) {
          const int key0 = t * 16 + g * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (KEY_OK(t, key0 + j)) atomicAdd(&dtab[rcq - rc[key0 + j] + p.rc0], racc[t][j]);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < p.table_len; i += NW * 64) {
      const float v = dtab[i];
      if (v != 0.f) atomicAdd(pb.dbias_table + (size_t)i * heads + h, v);
    }
  }
}

// ================================================================================================
// backward B: dK, dV (per key tile; probabilities recomputed from lse; delta from kernel A)
// ================================================================================================
template <int HD, int MODE, int NW>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dkv_kernel(const vmvm_attn_bwd_desc pb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const Smem sm = smem_layout(L, HD, MODE, p.table_len, 2);
  const int logical = xcd_remap(blockIdx.x, p.nseq * heads);
  const int seq = logical / heads, h = logical - seq * heads;
  const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
  const u16* dO = reinterpret_cast<const u16*>(pb.dout) + (size_t)seq * L * pb.ld_dout + h * HD;
  unsigned char* Qsm = smem + sm.off_a;
  unsigned char* dOsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);
  float* lse_s = reinterpret_cast<float*>(smem + sm.off_lse);
  float* delta_s = reinterpret_cast<float*>(smem + sm.off_delta);

  fill_rowmajor<HD>(Qsm, qkv + p.q_off + h * HD, p.ld_qkv, L, sm.lp32, tid, NW * 64);
  fill_rowmajor<HD>(dOsm, dO, pb.ld_dout, L, sm.lp32, tid, NW * 64);
  fill_wait();
  const float* lse_g = p.lse + ((size_t)seq * heads + h) * L;
  const float* delta_g = pb.delta + ((size_t)seq * heads + h) * L;
  for (int i = tid; i < sm.lp32; i += NW * 64) {
    lse_s[i] = i < L ? lse_g[i] : __builtin_huge_valf();       // +inf -> p = 0 for padded queries
    delta_s[i] = i < L ? delta_g[i] : 0.f;
    if (MODE == 0) {
      rc[i] = i < L ? p.rc[i] : 0;
      reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
    } else {
      reg[i] = (i < L) ? (p.keymask ? p.keymask[(size_t)seq * L + i] : 1) : 0;
    }
  }
  if (MODE == 0)
    for (int i = tid; i < p.table_len; i += NW * 64) tab[i] = p.bias_table[(size_t)i * heads + h];
  __syncthreads();

  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const uint32_t thr8 = drop_thr8(p.dropout_p);
  const float keep = has_drop ? 256.f / (256.f - (float)thr8) : 1.f;
  const int nt = sm.nt, nt2 = sm.nt2;

  // each wave owns TWO key tiles at a time: the Q / dO fragments (k = hd) and the transposed Q^T / dO^T fragments
  // (k = tokens) are read from LDS once and feed both tiles -> half the LDS traffic per MFMA.
  constexpr int KT = (HD == 32) ? 2 : 1;
  for (int kp = wave; kp * KT < nt; kp += NW) {
    int key[KT]; bool kv[KT];
    bf16x8 kf[KT][HD / 32], vf[KT][HD / 32];
    int rck[KT], regk[KT];
    f32x4 dk[KT][HD / 16], dv[KT][HD / 16];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      key[t] = (kp * KT + t) * 16 + r;
      kv[t] = key[t] < L;
#pragma unroll
      for (int s = 0; s < HD / 32; ++s) {
        kf[t][s] = load_frag_global(qkv + (size_t)key[t] * p.ld_qkv + p.k_off + h * HD + g * 8 + s * 32, kv[t]);
        vf[t][s] = load_frag_global(qkv + (size_t)key[t] * p.ld_qkv + p.v_off + h * HD + g * 8 + s * 32, kv[t]);
      }
      rck[t] = (MODE == 0) ? rc[kv[t] ? key[t] : 0] : 0;
      regk[t] = reg[kv[t] ? key[t] : 0];
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) { dk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }

    for (int c = 0; c < nt2; ++c) {
      float pt[KT][2][4], ds[KT][2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * c + u;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) { pt[t][u][j] = 0.f; ds[t][u][j] = 0.f; }
        if (qt < nt) {
          const int qrow = qt * 16 + r;            // A-operand row owned by this lane
          f32x4 s4[KT], dp4[KT];
#pragma unroll
          for (int t = 0; t < KT; ++t) { s4[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp4[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int s = 0; s < HD / 32; ++s) {
            const bf16x8 qf = frag_hd<HD>(Qsm, qrow, s * 4 + g);
            const bf16x8 dof = frag_hd<HD>(dOsm, qrow, s * 4 + g);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              s4[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[t][s], s4[t], 0, 0, 0);
              dp4[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[t][s], dp4[t], 0, 0, 0);
            }
          }
          // lane now holds (query = qt*16 + 4g + j, key[t])
          const int q0 = qt * 16 + g * 4;
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + q0);
          const float4 d4 = *reinterpret_cast<const float4*>(delta_s + q0);
          const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
          const float dls[4] = {d4.x, d4.y, d4.z, d4.w};
          if (MODE == 0) {
            const int4 rq = *reinterpret_cast<const int4*>(rc + q0);
            const uchar4 gq = *reinterpret_cast<const uchar4*>(reg + q0);
            const int rqs[4] = {rq.x, rq.y, rq.z, rq.w};
            const int gqs[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float sv = s4[t][j] + tab[rqs[j] - rck[t] + p.rc0] + (gqs[j] != regk[t] ? -100.f : 0.f);
                const float pr = kv[t] ? __expf(sv - ls[j]) : 0.f;
                pt[t][u][j] = pr;
                ds[t][u][j] = pr * (dp4[t][j] * seq_scale - dls[j]);
              }
          } else {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              uint4 blk = make_uint4(0, 0, 0, 0);
              if (has_drop) blk = drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q0 >> 2), (uint32_t)(key[t] >> 2));
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float pr = (kv[t] && regk[t]) ? __expf(s4[t][j] * p.scale - ls[j]) : 0.f;
                float dpj = dp4[t][j] * seq_scale, pj = pr;
                if (has_drop) {
                  const bool dropped = ((u4_get(blk, j) >> (8 * (key[t] & 3))) & 0xffu) < thr8;
                  dpj = dropped ? 0.f : dpj * keep;
                  pj = dropped ? 0.f : pr * keep;
                }
                pt[t][u][j] = pj;
                ds[t][u][j] = pr * (dpj - dls[j]);
              }
            }
          }
        }
      }
      bf16x8 pf[KT], dsf[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t) { pf[t] = frag_from_f32(pt[t][0], pt[t][1]); dsf[t] = frag_from_f32(ds[t][0], ds[t][1]); }
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        const bf16x8 dof = frag_tokens<HD>(dOsm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r);
        const bf16x8 qf = frag_tokens<HD>(Qsm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r);
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          dv[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf[t], dv[t][dt], 0, 0, 0);
          dk[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, dsf[t], dk[t][dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kv[t]) {
        u16* base = reinterpret_cast<u16*>(pb.dqkv) + ((size_t)seq * L + key[t]) * pb.ld_dqkv + h * HD + g * 4;
        const float ksc = (MODE == 1) ? p.scale : 1.0f;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          *reinterpret_cast<uint2*>(base + p.k_off + dt * 16) =
              make_uint2(pack_bf2(dk[t][dt][0] * ksc, dk[t][dt][1] * ksc), pack_bf2(dk[t][dt][2] * ksc, dk[t][dt][3] * ksc));
          *reinterpret_cast<uint2*>(base + p.v_off + dt * 16) =
              make_uint2(pack_bf2(dv[t][dt][0] * seq_scale, dv[t][dt][1] * seq_scale), pack_bf2(dv[t][dt][2] * seq_scale, dv[t][dt][3] * seq_scale));
        }
      }
    }
  }
}

template <typename K>
int set_smem(K kernel, int bytes) {
  if (bytes > 160 * 1024) return VMVM_ENOSUPPORT;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return VMVM_EHIP;
  }
  return VMVM_OK;
}

int check_desc(const vmvm_attn_fwd_desc* d) {
  if (!d || !d->qkv || !d->out || !d->lse) return VMVM_EINVAL;
  if (d->nseq <= 0 || d->L <= 0 || d->heads <= 0) return VMVM_EINVAL;
  if (d->mode == 0 && (d->head_dim != 32 || !d->bias_table || !d->rc || d->table_len <= 0)) return VMVM_EINVAL;
  if (d->mode == 1 && d->head_dim != 64) return VMVM_EINVAL;
  if (d->mode != 0 && d->mode != 1) return VMVM_EINVAL;
  if ((d->ld_qkv & 7) || (d->ld_out & 7) || (d->q_off & 7) || (d->k_off & 7) || (d->v_off & 7)) return VMVM_EINVAL;
  if (d->L > 448) return VMVM_ENOSUPPORT;          // full-row-in-registers design (C5 needs the streaming variant)
  if (d->region && d->n_win <= 0) return VMVM_EINVAL;
  if (d->seq_scale && d->seqs_per_scale <= 0) return VMVM_EINVAL;
  return VMVM_OK;
}

}  // namespace

#define LAUNCH_FWD(HD, MODE, NTM, NW, NX, MASK)                                              \
  do {                                                                                       \
    int rc_ = set_smem(attn_fwd_kernel<HD, MODE, NTM, NW, NX, MASK>, sm.total);              \
    if (rc_) return rc_;                                                                     \
    hipLaunchKernelGGL((attn_fwd_kernel<HD, MODE, NTM, NW, NX, MASK>), dim3(nb), dim3(NW * 64), sm.total, st, *d); \
  } while (0)

extern "C" int vmvm_attention_fwd(const vmvm_attn_fwd_desc* d, void* stream) {
  int rc = check_desc(d);
  if (rc) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const Smem sm = smem_layout(d->L, d->head_dim, d->mode, d->table_len, 0);
  const int nb = d->nseq * d->heads;
  const bool mask = d->region != nullptr;
  if (d->mode == 0) {
    if (sm.nt == 25) { if (mask) LAUNCH_FWD(32, 0, 26, 4, 25, true); else LAUNCH_FWD(32, 0, 26, 4, 25, false); }
    else if (sm.nt == 13) { if (mask) LAUNCH_FWD(32, 0, 14, 4, 13, true); else LAUNCH_FWD(32, 0, 14, 4, 13, false); }
    else if (sm.nt <= 16) LAUNCH_FWD(32, 0, 16, 4, 0, true);
    else LAUNCH_FWD(32, 0, 28, 4, 0, true);
  } else {
    // (exact-NT instantiations of the head_dim-64 kernel spill: the hoisted loads of 27 unguarded tiles exceed 256 VGPRs)
    if (sm.nt <= 16) LAUNCH_FWD(64, 1, 16, 4, 0, true);
    else LAUNCH_FWD(64, 1, 28, 8, 0, true);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

#define LAUNCH_BWD(KERN, HD, MODE, NW, WHICH)                                                \
  do {                                                                                       \
    const Smem s_ = smem_layout(d->f.L, d->f.head_dim, d->f.mode, d->f.table_len, WHICH);    \
    int rc_ = set_smem(KERN<HD, MODE, NW>, s_.total);                                        \
    if (rc_) return rc_;                                                                     \
    hipLaunchKernelGGL((KERN<HD, MODE, NW>), dim3(nb), dim3(NW * 64), s_.total, st, *d);     \
    VMVM_CHECK_LAUNCH();                                                                     \
  } while (0)
#define LAUNCH_BWD_DQ(HD, MODE, NW)                                                          \
  do {                                                                                       \
    const Smem s_ = smem_layout(d->f.L, d->f.head_dim, d->f.mode, d->f.table_len, 1);        \
    int rc_ = set_smem(attn_bwd_dq_kernel<HD, MODE, NW>, s_.total);                          \
    if (rc_) return rc_;                                                                     \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, MODE, NW>), dim3(nchunks * d->f.heads), dim3(NW * 64), s_.total, st, *d, nchunks); \
    VMVM_CHECK_LAUNCH();                                                                     \
  } while (0)

extern "C" int vmvm_attention_bwd(const vmvm_attn_bwd_desc* d, void* stream) {
  if (!d) return VMVM_EINVAL;
  int rc = check_desc(&d->f);
  if (rc) return rc;
  if (!d->dout || !d->dqkv || !d->delta || (d->ld_dout & 7) || (d->ld_dqkv & 7)) return VMVM_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nb = d->f.nseq * d->f.heads;
  // dq: persistent workgroups, ~3 per CU (mode 0) so the per-workgroup bias-table flush is amortised over many windows
  int nchunks = d->f.nseq;
  if (d->f.mode == 0) {
    const int want = (768 + d->f.heads - 1) / d->f.heads;
    if (nchunks > want) nchunks = want;
  }
  if (d->f.mode == 0) {
    const Smem s_ = smem_layout(d->f.L, 32, 0, d->f.table_len, 1);
    const int nqg = (s_.nt + 3) / 4;
    int nch = 768 / (d->f.heads * nqg);
    if (nch < 1) nch = 1;
    if (nch > d->f.nseq) nch = d->f.nseq;
#define LAUNCH_DQW(NTM, NX, MASK)                                                                                   \
    do {                                                                                                            \
      int rc_ = set_smem(attn_bwd_dq_win_kernel<NTM, 4, NX, MASK>, s_.total);                                        \
      if (rc_) return rc_;                                                                                          \
      hipLaunchKernelGGL((attn_bwd_dq_win_kernel<NTM, 4, NX, MASK>), dim3(nch * d->f.heads * nqg), dim3(256), s_.total, st, *d, nch, nqg); \
    } while (0)
    const bool mask = d->f.region != nullptr;
    if (s_.nt == 25) { if (mask) LAUNCH_DQW(26, 25, true); else LAUNCH_DQW(26, 25, false); }
    else if (s_.nt == 13) { if (mask) LAUNCH_DQW(14, 13, true); else LAUNCH_DQW(14, 13, false); }
    else if (s_.nt <= 16) LAUNCH_DQW(16, 0, true);
    else LAUNCH_DQW(28, 0, true);
    VMVM_CHECK_LAUNCH();
    LAUNCH_BWD(attn_bwd_dkv_kernel, 32, 0, 4, 2);
  } else {
    LAUNCH_BWD_DQ(64, 1, 8);
    LAUNCH_BWD(attn_bwd_dkv_kernel, 64, 1, 8, 2);
  }
  return VMVM_OK;
}
