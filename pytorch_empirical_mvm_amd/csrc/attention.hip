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
// Attention-prob dropout: Philox4x32-7 per 4x4 (query,key) block, 16 random bytes; an element compares a 16-bit field
// (p quantised to 1/65536, attn_common.h drop_field), the same block is addressed row-wise by fwd/A and column-wise by B.
#include "attn_common.h"
#include <vmvm_probe_hooks.h>
#include <cstdlib>

namespace {


// ================================================================================================
// forward
// ================================================================================================
// NX > 0: the number of 16-key tiles is a compile-time constant (the step's shapes: 392 -> 25, 196 -> 13, 432 -> 27, 232 -> 15),
// so the fully unrolled tile loops carry no runtime guards / exec-mask juggling and only the LAST tile checks key < L.
// MASK = false: un-shifted window block (no region compare).
#define TILE_ON(t) (NX ? ((t) < NX) : ((t) < nt))
#define PAIR_ON(c) (NX ? ((c) < (NX + 1) / 2) : ((c) < nt2))
#define KEY_OK(t, key) ((NX && (t) < NX - 1) ? true : ((key) < L))
// CAUSAL (mode 1, `causal_from` = cf > 0): the seq2seq mask of VIOLET_Base.get_attn_mask (model.py:191-199) -- keys below cf (the
// visual tokens) follow the key mask for every query; a key >= cf (text) is visible only to text queries at or after it.
// COLSUM (mode 1, `att_colsum` set): also accumulates att_colsum[seq][key] += att_scale * sum_q P[q][key] over the heads -- the
// layer- and head-averaged attention column sums of VIOLET_Pretrain.get_att (main_pretrain.py:211-215, attention-guided masking)
// without materialising any attention matrix.  P is what HF returns as `attentions`: the probabilities AFTER attention dropout.
template <int HD, int MODE, int NT_MAX, int NW, int NX, bool MASK, bool CAUSAL = false, bool COLSUM = false>
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
    // key mask as an ADDITIVE bias (0 / -inf; also -inf beyond L) in the unused rc slot: score = fma(acc, scale * log2e, kbias[key])
    // in the log2 domain is one packed FMA per two elements instead of byte compares and selects
    float* kb = reinterpret_cast<float*>(rc);
    for (int i = tid; i < sm.lp32; i += NW * 64) {
      const bool on = (i < L) && (p.keymask ? p.keymask[(size_t)seq * L + i] != 0 : true);
      reg[i] = on ? 1 : 0;
      kb[i] = on ? 0.f : NEG_INF;
    }
  }
  __syncthreads();

  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = has_drop ? 65536.f / (65536.f - (float)thr16) : 1.f;
  const int nt = sm.nt, nt2 = sm.nt2;
  constexpr bool FAST1 = (MODE == 1) && !CAUSAL;          // fusion encoder: log2-domain packed softmax math
  const float sc2 = p.scale * 1.4426950408889634f;

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
        if (NX && HD == 64 && (t & 3) == 3) asm volatile("" ::: "memory");   // exact-tile build: keep the K fragment loads from being hoisted en bloc
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
        } else if (FAST1) {
          const float4 k4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rc) + key0);
          const f32x2 s01 = __builtin_elementwise_fma(f32x2{acc[t][0], acc[t][1]}, f32x2{sc2, sc2}, f32x2{k4.x, k4.y});
          const f32x2 s23 = __builtin_elementwise_fma(f32x2{acc[t][2], acc[t][3]}, f32x2{sc2, sc2}, f32x2{k4.z, k4.w});
          acc[t] = f32x4{s01[0], s01[1], s23[0], s23[1]};
          mx = fmaxf(fmaxf(mx, s01[0]), s01[1]);
          mx = fmaxf(fmaxf(mx, s23[0]), s23[1]);
        } else {
          const uchar4 mk = *reinterpret_cast<const uchar4*>(reg + key0);
          const int mks[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bool ok = mks[j] != 0;
            if (CAUSAL) ok = (key0 + j < p.causal_from) ? ok : (q >= p.causal_from && key0 + j <= q && key0 + j < L);
            float s = ok ? acc[t][j] * p.scale : NEG_INF;
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
    if constexpr (FAST1 && NX > 0 && !COLSUM) {
      // exact-tile fusion build: exp2 / row sum / dropout mask / bf16 pack / P V of one PAIR of key tiles at a time, so the P V MFMAs of
      // a pair run while the VALU works on the next pair (as three separate loops over the row block, the matrix pipe idled through
      // the whole softmax pass and the VALU through the whole P V pass)
      f32x2 sum2 = f32x2{0.f, 0.f};
      const f32x2 nmx = f32x2{-mx, -mx};
      f32x4 o[HD / 16];
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto pairs = [&](auto drop_c) {
        constexpr int DROPM = decltype(drop_c)::value;       // 0 no dropout, 1 Philox decisions, 2 Philox decisions + stored for the backward (drop_mask)
        constexpr bool DROP = DROPM != 0;
        uint4 own = make_uint4(0, 0, 0, 0);
        const uint32_t* mrow = DROPM == 2 ? uniform_ptr(p.drop_mask + ((size_t)(seq * heads + h) * nt + qt) * nt * 8) : nullptr;
#pragma unroll
        for (int c = 0; c < NT_MAX / 2; ++c) {
          if (PAIR_ON(c)) {
            float e[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int t = 2 * c + u;
              const f32x2 d01 = f32x2{acc[t][0], acc[t][1]} + nmx, d23 = f32x2{acc[t][2], acc[t][3]} + nmx;
              const f32x2 e01 = f32x2{__builtin_amdgcn_exp2f(d01[0]), __builtin_amdgcn_exp2f(d01[1])};
              const f32x2 e23 = f32x2{__builtin_amdgcn_exp2f(d23[0]), __builtin_amdgcn_exp2f(d23[1])};
              sum2 += e01; sum2 += e23;                        // the row sum is of the UN-dropped probabilities
              e[u][0] = e01[0]; e[u][1] = e01[1]; e[u][2] = e23[0]; e[u][3] = e23[1];
              if (DROP && TILE_ON(t)) {
                if ((t & 3) == 0)
                  own = quad_transpose(drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q >> 2), (uint32_t)((t + (lane & 3)) * 4 + g)), lane & 1, lane & 2);
                const uint32_t w = u4_static(own, t & 3);
                bool dr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { dr[j] = drop_field(w, j) < thr16; e[u][j] = dr[j] ? 0.f : e[u][j]; }
                if (DROPM == 2) {                              // the four compares' lane masks ARE the tile's record: 8 dwords, lanes 0-7 store them
                  // ... and they are wave-uniform scalar pairs already: four SCALAR stores, no vector instruction (attn_common.h)
                  store_lane_masks(mrow, t * 32, __builtin_amdgcn_ballot_w64(dr[0]), __builtin_amdgcn_ballot_w64(dr[1]), __builtin_amdgcn_ballot_w64(dr[2]),
                                   __builtin_amdgcn_ballot_w64(dr[3]));
                }
              }
            }
            const bf16x8 pf = frag_from_f32(e[0], e[1]);
#pragma unroll
            for (int dt = 0; dt < HD / 16; ++dt) {
              const bf16x8 vf = frag_tokens<HD>(Vsm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r);
              o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
            }
          }
        }
      };
      if (has_drop) { if (p.drop_mask) { pairs(std::integral_constant<int, 2>{}); scalar_stores_done(); } else pairs(std::integral_constant<int, 1>{}); } else pairs(std::integral_constant<int, 0>{});
      float sum = sum2[0] + sum2[1];
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      if (g == 0 && qv) p.lse[((size_t)seq * heads + h) * L + q] = (mx + __log2f(sum)) * 0.6931471805599453f;
      if (qv) {
        const float inv = seq_scale * keep / sum;
        u16* op = reinterpret_cast<u16*>(p.out) + ((size_t)seq * L + q) * p.ld_out + h * HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt)
          *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
      }
    } else {
    float sum = 0.f;
    if (FAST1) {
      f32x2 sum2 = f32x2{0.f, 0.f};
      const f32x2 nmx = f32x2{-mx, -mx};
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        const f32x2 d01 = f32x2{acc[t][0], acc[t][1]} + nmx, d23 = f32x2{acc[t][2], acc[t][3]} + nmx;
        const f32x2 e01 = f32x2{__builtin_amdgcn_exp2f(d01[0]), __builtin_amdgcn_exp2f(d01[1])};
        const f32x2 e23 = f32x2{__builtin_amdgcn_exp2f(d23[0]), __builtin_amdgcn_exp2f(d23[1])};
        acc[t] = f32x4{e01[0], e01[1], e23[0], e23[1]};
        sum2 += e01; sum2 += e23;
      }
      sum = sum2[0] + sum2[1];
    } else {
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float e = __expf(acc[t][j] - mx); acc[t][j] = e; sum += e; }
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    // saved log-sum-exp stays in natural-log units of the scaled scores (the backward kernels convert)
    if (g == 0 && qv) p.lse[((size_t)seq * heads + h) * L + q] = FAST1 ? (mx + __log2f(sum)) * 0.6931471805599453f : mx + __logf(sum);
    if (has_drop) {
      uint4 own = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        if ((t & 3) == 0 && TILE_ON(t))                   // quad lane i: block of tile t+i  (block id = (sequence-head, q/4, key/4)), then transposed
          own = quad_transpose(drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q >> 2), (uint32_t)((t + (lane & 3)) * 4 + g)), lane & 1, lane & 2);
        if (TILE_ON(t)) {
          const uint32_t w = u4_static(own, t & 3);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[t][j] = (drop_field(w, j) < thr16) ? 0.f : acc[t][j];      // the 1/(1-p) keep scale rides on 1/sum below
        }
      }
    }
    if (COLSUM) {                                       // 16 queries of the tile live in one 16-lane row: xor-shuffles 1..8 stay inside it
      const float wq = qv ? p.att_scale * keep / sum : 0.f;
      float* cs = p.att_colsum + (size_t)seq * L;
#pragma unroll
      for (int t = 0; t < NT_MAX; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v = acc[t][j] * wq;
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            const int key = t * 16 + g * 4 + j;
            if (r == 0 && key < L) atomicAdd(cs + key, v);
          }
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
      const float inv = seq_scale * keep / sum;
      u16* op = reinterpret_cast<u16*>(p.out) + ((size_t)seq * L + q) * p.ld_out + h * HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt)
        *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
    }
    }
  }
}

// ================================================================================================
// backward A: dQ (+ delta, + relative-position-bias table gradient)
// ================================================================================================
// QT = query tiles a wave owns at a time (fusion build: 2 -- the K / V fragments, the transposed K fragments and the key mask a
// wave reads from LDS per key tile then serve two score tiles: 7 -> 3.5 KB of LDS reads per score tile)
template <int HD, int MODE, int NW, int NXB, bool CAUSAL = false, int QT = 1>
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
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = has_drop ? 65536.f / (65536.f - (float)thr16) : 1.f;
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
    float* kb = reinterpret_cast<float*>(rc);          // additive key mask (0 / -inf), see attn_fwd_kernel
    for (int i = tid; i < sm.lp32; i += NW * 64) {
      const bool on = (i < L) && (p.keymask ? p.keymask[(size_t)seq * L + i] != 0 : true);
      reg[i] = on ? 1 : 0;
      kb[i] = on ? 0.f : NEG_INF;
    }
  }
  __syncthreads();

  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  constexpr bool FAST1 = (MODE == 1) && !CAUSAL;
  const float sc2 = p.scale * 1.4426950408889634f, cdk = seq_scale * keep;
  const float* lse_g = p.lse + ((size_t)seq * heads + h) * L;
  float* delta_g = pb.delta + ((size_t)seq * heads + h) * L;

  // the dropout tests are compile-time inside the tile loops (two copies of the loop, chosen once per sequence): as run-time tests they
  // were ~60 scalar branches per query tile, each one a scheduling boundary between the tiles' MFMA and VALU work
  auto qloop = [&](auto drop_c) {
  constexpr int DROPM = decltype(drop_c)::value;          // 0 no dropout, 1 Philox decisions, 2 the forward's stored decisions (drop_mask; exact-tile fusion build)
  constexpr bool DROP = DROPM == 1;
  static_assert(DROPM != 2 || (QT == 1 && NXB > 0 && NXB * 8 <= 256 && FAST1), "stored-mask path: one query tile per wave, exact tile count");
  for (int qt0 = wave * QT; qt0 < nt; qt0 += NW * QT) {
    int q[QT]; bool qv[QT];
    bf16x8 qf[QT][HD / 32], dof[QT][HD / 32];
    float dl[QT], lse[QT], lse2[QT];
    int rcq[QT], regq[QT];
    f32x4 dq[QT][HD / 16];
    uint4 own[QT];                                    // quad-shared dropout block (see quad_bcast)
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      q[i] = (qt0 + i) * 16 + r;
      qv[i] = q[i] < L;
      dl[i] = 0.f;
#pragma unroll
      for (int s = 0; s < HD / 32; ++s) {
        qf[i][s] = load_frag_global(qkv + (size_t)q[i] * p.ld_qkv + p.q_off + h * HD + g * 8 + s * 32, qv[i]);
        dof[i][s] = load_frag_global(dO + (size_t)q[i] * pb.ld_dout + g * 8 + s * 32, qv[i]);
        const bf16x8 of = load_frag_global(O + (size_t)q[i] * p.ld_out + g * 8 + s * 32, qv[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl[i] += (float)dof[i][s][e] * (float)of[e];
      }
      dl[i] += __shfl_xor(dl[i], 16, 64);
      dl[i] += __shfl_xor(dl[i], 32, 64);             // delta_q = sum_d dO'[q,d] * out'[q,d]
      if (g == 0 && qv[i]) delta_g[q[i]] = dl[i];
      lse[i] = qv[i] ? lse_g[q[i]] : 0.f;
      lse2[i] = qv[i] ? lse[i] * 1.4426950408889634f : __builtin_huge_valf();
      rcq[i] = (MODE == 0) ? rc[qv[i] ? q[i] : 0] : 0;
      regq[i] = (MODE == 0) ? reg[qv[i] ? q[i] : 0] : 0;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) dq[i][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      own[i] = make_uint4(0, 0, 0, 0);
    }

    uint32_t mw[4] = {0u, 0u, 0u, 0u};                  // stored decisions of this query tile's row of tiles: nt x 8 dwords, dword i in lane i & 63 of mw[i >> 6]
    if (DROPM == 2) {
      const uint32_t* mrow = p.drop_mask + ((size_t)(seq * heads + h) * nt + qt0) * nt * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) mw[i] = (i * 64 + lane < NXB * 8) ? mrow[i * 64 + lane] : 0u;
    }
    // score / dP MFMAs of one pair of key tiles
    auto score_pair = [&](int c, f32x4 (&so)[2][QT], f32x4 (&dpo)[2][QT]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * c + u;
#pragma unroll
        for (int i = 0; i < QT; ++i) { so[u][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dpo[u][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (NXB ? (t < NXB) : (t < nt)) {
          const int row = t * 16 + r;
#pragma unroll
          for (int s = 0; s < HD / 32; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ksm + k_off_swz<HD>(row, s * 4 + g));
            const bf16x8 vf = frag_hd<HD>(Vsm, row, s * 4 + g);
#pragma unroll
            for (int i = 0; i < QT; ++i) {
              so[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[i][s], so[u][i], 0, 0, 0);
              dpo[u][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[i][s], dpo[u][i], 0, 0, 0);
            }
          }
        }
      }
    };
    // exact-tile fusion build: software-pipelined by one pair -- the score MFMAs of pair c + 1 are issued in front of the softmax-side
    // VALU work of pair c (they only depend on LDS reads), so their latency sits under it instead of in front of the next pair
    constexpr bool PIPE = FAST1 && NXB > 0 && QT == 1;
    f32x4 s4c[2][QT], dpc[2][QT];
    if (PIPE) score_pair(0, s4c, dpc);
    // NXB > 0: exact tile count -> fully unrolled, immediate LDS offsets, no per-tile guards
#pragma unroll
    for (int c = 0; c < (NXB ? (NXB + 1) / 2 : nt2); ++c) {
      f32x4 s4n[2][QT], dpn[2][QT];
      if (PIPE) { if (c + 1 < (NXB + 1) / 2) score_pair(c + 1, s4n, dpn); }
      else score_pair(c, s4c, dpc);
      float ds[QT][2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * c + u;
#pragma unroll
        for (int i = 0; i < QT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) ds[i][u][j] = 0.f;
        if (NXB ? (t < NXB) : (t < nt)) {
          const int key0 = t * 16 + g * 4;
          float4 k4 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (FAST1) k4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rc) + key0);
#pragma unroll
          for (int i = 0; i < QT; ++i) {
          const f32x4 s4 = s4c[u][i], dp4 = dpc[u][i];
          uint32_t w = 0;
          if (DROP) {
            if ((t & 3) == 0) own[i] = quad_transpose(drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q[i] >> 2), (uint32_t)((t + (lane & 3)) * 4 + g)), lane & 1, lane & 2);
            w = u4_static(own[i], t & 3);
          }
          if (MODE == 0) {
            const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
            const uchar4 gk = *reinterpret_cast<const uchar4*>(reg + key0);
            const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
            const int gks[4] = {gk.x, gk.y, gk.z, gk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int bi = rcq[i] - rks[j] + p.rc0;
              const float sv = s4[j] + tab[bi] + (regq[i] != gks[j] ? -100.f : 0.f);
              const float pr = (key0 + j < L && qv[i]) ? __expf(sv - lse[i]) : 0.f;
              const float dd = pr * (dp4[j] * seq_scale - dl[i]);
              ds[i][u][j] = dd;
              if (want_dtab && pr != 0.f) atomicAdd(&dtab[bi], dd);
            }
          } else if (FAST1) {
            // log2 domain, two elements per instruction: P = exp2(fma(S, scale*log2e, kbias[key]) - lse*log2e) (lse2 = +inf on padded
            // rows), dS = P * (keep-scaled dP under the dropout mask - delta)
            const f32x2 nl = f32x2{-lse2[i], -lse2[i]}, ndl = f32x2{-dl[i], -dl[i]}, c2 = f32x2{cdk, cdk};
            const f32x2 x01 = __builtin_elementwise_fma(f32x2{s4[0], s4[1]}, f32x2{sc2, sc2}, f32x2{k4.x, k4.y}) + nl;
            const f32x2 x23 = __builtin_elementwise_fma(f32x2{s4[2], s4[3]}, f32x2{sc2, sc2}, f32x2{k4.z, k4.w}) + nl;
            const float pr[4] = {__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1]), __builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
            const f32x2 d01 = f32x2{dp4[0], dp4[1]} * c2, d23 = f32x2{dp4[2], dp4[3]} * c2;
            float dp[4] = {d01[0], d01[1], d23[0], d23[1]};
            if (DROP) {
#pragma unroll
              for (int j = 0; j < 4; ++j) dp[j] = (drop_field(w, j) < thr16) ? 0.f : dp[j];
            }
            if (DROPM == 2) {                              // element j's lane mask back into a scalar pair, applied by one v_cndmask each
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int d0 = t * 8 + 2 * j;
                const uint32_t lo = __builtin_amdgcn_readlane(mw[d0 >> 6], d0 & 63), hi = __builtin_amdgcn_readlane(mw[(d0 + 1) >> 6], (d0 + 1) & 63);
                dp[j] = zero_where(dp[j], ((uint64_t)hi << 32) | lo);
              }
            }
            const f32x2 o01 = f32x2{pr[0], pr[1]} * (f32x2{dp[0], dp[1]} + ndl), o23 = f32x2{pr[2], pr[3]} * (f32x2{dp[2], dp[3]} + ndl);
            ds[i][u][0] = o01[0]; ds[i][u][1] = o01[1]; ds[i][u][2] = o23[0]; ds[i][u][3] = o23[1];
          } else {
            const uchar4 mk = *reinterpret_cast<const uchar4*>(reg + key0);
            const int mks[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              bool ok = mks[j] != 0;
              if (CAUSAL) ok = (key0 + j < p.causal_from) ? ok : (q[i] >= p.causal_from && key0 + j <= q[i] && key0 + j < L);
              const float pr = (ok && qv[i]) ? __expf(s4[j] * p.scale - lse[i]) : 0.f;
              float dpj = dp4[j] * seq_scale;
              if (DROP) dpj = (drop_field(w, j) < thr16) ? 0.f : dpj * keep;
              ds[i][u][j] = pr * (dpj - dl[i]);
            }
          }
          }
        }
      }
      bf16x8 dsf[QT];
#pragma unroll
      for (int i = 0; i < QT; ++i) dsf[i] = frag_from_f32(ds[i][0], ds[i][1]);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        const bf16x8 kf = frag_tokens<HD>(Ksm, dt, c * 32 + g * 4, c * 32 + 16 + g * 4, r);
#pragma unroll
        for (int i = 0; i < QT; ++i) dq[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf[i], dq[i][dt], 0, 0, 0);
      }
      if (PIPE) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < QT; ++i) { s4c[u][i] = s4n[u][i]; dpc[u][i] = dpn[u][i]; }
      }
      if (NXB) __builtin_amdgcn_sched_barrier(0);       // exact-tile build: straight-line code -- keep the scheduler from hoisting all pairs' LDS reads / Philox blocks (spills)
    }
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      if (qv[i]) {
        u16* dqp = reinterpret_cast<u16*>(pb.dqkv) + ((size_t)seq * L + q[i]) * pb.ld_dqkv + p.q_off + h * HD + g * 4;
        const float sc = p.scale;
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt)
          *reinterpret_cast<uint2*>(dqp + dt * 16) = make_uint2(pack_bf2(dq[i][dt][0] * sc, dq[i][dt][1] * sc), pack_bf2(dq[i][dt][2] * sc, dq[i][dt][3] * sc));
      }
    }
  }
  };
  if (has_drop) {
    if constexpr (QT == 1 && NXB > 0 && FAST1) { if (p.drop_mask) qloop(std::integral_constant<int, 2>{}); else qloop(std::integral_constant<int, 1>{}); }
    else qloop(std::integral_constant<int, 1>{});
  } else qloop(std::integral_constant<int, 0>{});
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
// forward (window mode, exact tile count): batch-persistent twin of the backward kernels above.  A workgroup = (head, query
// group, chunk of window-major sequences); each of its 7 waves owns one query tile, holds the whole 16 x L score block in
// accumulators (exact two-pass softmax) and keeps the bias + shift-mask block as packed-bf16 registers feeding the score MFMA's
// C operand.  K / V of the next sequence stream into the other LDS buffer (DMA requests spread over the tile loop).
// ================================================================================================
template <int NX, bool MASK>
__global__ __launch_bounds__(448) void attn_fwd_win2_kernel(const vmvm_attn_fwd_desc p, const int nqg, const int nch) {
  constexpr int HD = 32, NWV = 7, NPK = (NX + 1) / 2;                   // NPK key-tile pairs
  constexpr int LP32 = NPK * 32, KV = LP32 * HD * 2;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const s16x4_ ident = bias_ident_frag(lane);
  const int L = p.L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nqg);
  const int qg = logical % nqg;
  const int t1 = logical / nqg;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  const int tl4 = (p.table_len + 3) & ~3, lr4 = (L + 3) & ~3;
  float* tabs = reinterpret_cast<float*>(smem + 4 * KV);                // this head's bias-table column
  int* rcs = reinterpret_cast<int*>(tabs + tl4);
  unsigned char* regs = reinterpret_cast<unsigned char*>(rcs + lr4);    // region ids of the current window position
  const int qt = qg * NWV + wave;
  const int q = qt * 16 + r;
  const bool active = qt < NX;
  const bool qv = active && (q < L);

  for (int i = tid; i < p.table_len; i += NWV * 64) tabs[i] = p.bias_table[(size_t)i * heads + h];
  for (int i = tid; i < L; i += NWV * 64) rcs[i] = p.rc[i];
  uint32_t bm[NX * 2];
  auto build_bm = [&]() {                                 // bias + shift mask of this wave's score block, packed bf16 (from LDS)
    // four keys per LDS read (rc codes as int4, region ids as one dword) and selects instead of per-element branches: every tile but
    // the last holds only keys < L in the exact-tile build; the last one clamps its reads and selects the pad value
    const int rcq = rcs[qv ? q : 0] + p.rc0;
    const int regq = MASK ? regs[qv ? q : 0] : 0;
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const int key0 = t * 16 + g * 4;
      const bool whole = (t < NX - 1) || (key0 + 4 <= L);
      const int k0c = whole ? key0 : 0;
      const int4 rk = *reinterpret_cast<const int4*>(rcs + k0c);
      const uint32_t gk = MASK ? *reinterpret_cast<const uint32_t*>(regs + k0c) : 0u;
      const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
      float b4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float b = tabs[rcq - rks[j]];
        if (MASK) b += ((int)((gk >> (8 * j)) & 0xffu) != regq) ? -100.f : 0.f;
        b4[j] = whole ? b : PAD_BIAS;
      }
      if (!whole && t == NX - 1) {                         // ragged last tile: keys L-1 and below inside this lane's four
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = key0 + j;
          if (key < L) {
            float b = tabs[rcq - rcs[key]];
            if (MASK) b += (regs[key] != regq) ? -100.f : 0.f;
            b4[j] = b;
          }
        }
      }
      bm[2 * t] = pack_bf2(b4[0], b4[1]);
      bm[2 * t + 1] = pack_bf2(b4[2], b4[3]);
      if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);                // straight-line code: keep the tiles' reads from piling up (registers)
    }
  };

  const int total = nWin * B;
  const int per = (total + nch - 1) / nch;
  const int b0 = ch * per, b1 = (b0 + per < total) ? b0 + per : total;
  const int w0 = b0 / B;
  int w_nx = w0, c_nx = b0 - w0 * B;                    // (window position, clip) of the NEXT sequence to request
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == B) { c_nx = 0; ++w_nx; } };
  const uint32_t off_q = (uint32_t)q * p.ld_qkv + p.q_off + h * HD + g * 8;
  const uint32_t off_o = (uint32_t)q * p.ld_out + h * HD + g * 4;
  const uint32_t off_ls = (uint32_t)h * L + q;
  constexpr int NF = (LP32 * 4 + NWV * 64 - 1) / (NWV * 64);            // 16-byte DMA requests per thread per image
  static_assert(NF <= NX, "the fill requests are spread over the key tiles");
  uint32_t goff[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int u = i * NWV * 64 + tid, row = u >> 2, chs = u & 3;
    goff[i] = (u < LP32 * 4) ? (uint32_t)((row * p.ld_qkv + ((chs ^ swz_chunk<32>(row)) << 3)) * 2) : 0xffffffffu;
  }
  const unsigned fill_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2);
  bf16x8 qf;
  float ss_n = 1.0f;
  auto fetch = [&](size_t seq) {
    qf = load_frag_global(reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv + off_q, qv);
    ss_n = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  };
  if (b0 < b1) {
    const u16* kv0 = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    fill_pre<NF, NWV * 64 * 16>(smem + (tid & ~63) * 16, KV, kv0 + p.k_off, kv0 + p.v_off, fill_bytes, goff);
    fetch(seq_nx());
    advance();
  }

  int wprev = -1, w_cu = w0, c_cu = b0 - w0 * B;        // (window position, clip) of the sequence being processed
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    fill_wait();
    __syncthreads();                                      // sequence b landed for everyone; everyone left the other buffer
    const int wcur = w_cu;
    if (++c_cu == B) { c_cu = 0; ++w_cu; }
    if (wcur != wprev) {                                  // (workgroup-uniform) new window position: rebuild the bias + mask block
      wprev = wcur;
      if (MASK) {
        if (b > b0) __syncthreads();                      // everyone is done with the previous window's region row
        for (int i = tid; i < L; i += NWV * 64) regs[i] = p.region[(size_t)wcur * L + i];
        __syncthreads();
      }
      if (MASK || b == b0) build_bm();
    }
    const bf16x8 cqf = qf;
    const float seq_scale = ss_n;
    const bool has_next = b + 1 < b1;
    const u16* kv_nx = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    unsigned char* dst_nx = smem + (cur ^ 1) * 2 * KV + (tid & ~63) * 16;
    if (has_next) {
      fetch(seq_nx());
      advance();
      if (!active) fill_pre<NF, NWV * 64 * 16>(dst_nx, KV, kv_nx + p.k_off, kv_nx + p.v_off, fill_bytes, goff);
    }
    if (!active) continue;
    const unsigned char* Ksm = smem + cur * 2 * KV;
    const unsigned char* kb = Ksm + k_off_swz<HD>(r, g);
    const unsigned char* tv0 = Ksm + KV + k_off_swz<HD>(g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8;
    const unsigned char* tv1 = Ksm + KV + k_off_swz<HD>(g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8;
    const uint32_t tv0a = lds_addr(tv0), tv1a = lds_addr(tv1);
    // pass 1: scores (+ bias through the C operand) for the whole row block, running maximum
    const __amdgpu_buffer_rsrc_t rk_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(kv_nx + p.k_off)), 0, __builtin_amdgcn_readfirstlane((int)fill_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(kv_nx + p.v_off)), 0, __builtin_amdgcn_readfirstlane((int)fill_bytes), 0x00020000);
    f32x4 acc[NX];
    float mx = NEG_INF;
    // pass 0: the bias + mask blocks of all NX tiles through the matrix core (independent products, NX - 1 of them between any block
    // and the score MFMA that accumulates onto it -- see the hazard note at bias_block_mfma)
#pragma unroll
    for (int t = 0; t < NX; ++t) acc[t] = bias_block_mfma(ident, bm[2 * t], bm[2 * t + 1], f32x4{0.f, 0.f, 0.f, 0.f});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kb + t * 1024), cqf, acc[t], 0, 0, 0);
      if (t < NF && has_next) fill_one_r(dst_nx + t * NWV * 64 * 16, KV, rk_nx, rv_nx, goff[t], t == NF - 1);
    }
    // The maxima run on inline-asm v_max3_f32 and hipcc does not insert MFMA-result wait states in front of inline asm: the
    // barrier keeps every one of them BEHIND all score MFMAs (measured without it: a max3 scheduled one instruction after the MFMA
    // whose result it reads -- stale data, flaky tests).  The last tile's maximum is 2 (NX - 1) instructions after its MFMA's issue.
    __builtin_amdgcn_sched_barrier(0);
    float mx1 = NEG_INF;                                  // two chains of v_max3_f32: two new elements per instruction
#pragma unroll
    for (int t = 0; t < NX; ++t) { mx = max3_f32(mx, acc[t][0], acc[t][1]); mx1 = max3_f32(mx1, acc[t][2], acc[t][3]); }
    __builtin_amdgcn_sched_barrier(0);
    mx = fmaxf(mx, mx1);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    f32x2 nm2 = {-mx * LOG2E, -mx * LOG2E};
    asm volatile("" : "+v"(nm2));                         // real register pair (see attn_bwd_dq_win2_kernel)
    // pass 2: p = exp2((s - max) * log2 e), row sum, P V with P as the B operand (the MFMA k-slot order is free)
    // row sums through the matrix core as well: an all-ones A operand makes every row of the product the column sums of P over the
    // pair's 32 keys (of the bf16 P the output is built from), complete across the wave -- no per-element adds, no shuffles
    typedef __attribute__((ext_vector_type(8))) short s16x8o;
    const bf16x8 ones8 = __builtin_bit_cast(bf16x8, s16x8o{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});
    f32x4 osum = {0.f, 0.f, 0.f, 0.f};
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < NPK; ++c) {
      s16x4 a0, a1, c0, c1;
      tr_read4(a0, a1, c0, c1, tv0a, tv1a, c * 2048);
      uint32_t pw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * c + u;
        if (t < NX) {
#pragma unroll
          for (int hj = 0; hj < 2; ++hj) {
            const f32x2 e = __builtin_elementwise_fma(f32x2{acc[t][2 * hj], acc[t][2 * hj + 1]}, f32x2{LOG2E, LOG2E}, nm2);
            const f32x2 pr = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            pw[2 * u + hj] = pack_bf2v(pr);
          }
        }
      }
      const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pw[0], pw[1], pw[2], pw[3]));
      tr_wait4(a0, a1, c0, c1);
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const s16x8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const s16x8 v1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      o[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v0), pf, o[0], 0, 0, 0);
      o[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v1), pf, o[1], 0, 0, 0);
      osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, pf, osum, 0, 0, 0);
    }
    const float sum = osum[0];
    if (qv) {
      const float inv = seq_scale / sum;
      u16* op = reinterpret_cast<u16*>(p.out) + seq * L * p.ld_out + off_o;
      *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf2(o[0][0] * inv, o[0][1] * inv), pack_bf2(o[0][2] * inv, o[0][3] * inv));
      *reinterpret_cast<uint2*>(op + 16) = make_uint2(pack_bf2(o[1][0] * inv, o[1][1] * inv), pack_bf2(o[1][2] * inv, o[1][3] * inv));
      if (g == 0) (p.lse + seq * heads * L)[off_ls] = mx + __builtin_amdgcn_logf(sum) * LN2;
    }
  }
}

// ================================================================================================
// backward A'' (window mode, exact tile count): persistent over the BATCH for a fixed (window position, head, query tile).
// Relative-position bias + shift mask of a wave's score block are constant across clips, so they are built ONCE into registers
// (packed bf16) and the per-clip work per score element drops to: add, fma+exp2, fma+mul, accumulate.  K/V of the next clip
// stream into the other LDS buffer (direct-to-LDS DMA) while the current one is processed; the next clip's Q / dO / O
// fragments are prefetched into registers.  A query tile is shared by TWO waves (one per half of the key tiles) so the
// register-resident state (bias 2x13 + dS sums 4x13 VGPRs) leaves room for 3+ waves per SIMD; the two dQ partials are
// combined through LDS at the next loop-top barrier.  7 query tiles x 2 = 14 waves per workgroup.
// ================================================================================================
constexpr int win2_rows(int nx, int ns) {               // LDS rows per K/V image: the last split's last tile PAIR may reach past nx
  const int nh = (nx + ns - 1) / ns;
  return ((ns - 1) * nh + 2 * ((nh + 1) / 2)) * 16;
}
template <int NX, bool MASK, int NQ, int NS>
__global__ __launch_bounds__(NQ * NS * 64) void attn_bwd_dq_win2_kernel(const vmvm_attn_bwd_desc pb, const int nqg, const int nch) {
  constexpr int HD = 32, NWV = NQ * NS, NH = (NX + NS - 1) / NS;       // NH key tiles per split (the last split may have fewer)
  constexpr int NPH = (NH + 1) / 2;                                     // tile pairs per split
  constexpr int LP32 = win2_rows(NX, NS), KV = LP32 * HD * 2;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nqg);
  const int qg = logical % nqg;
  const int t1 = logical / nqg;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  const int tl4 = (p.table_len + 3) & ~3, lr4 = (L + 3) & ~3;
  float* dtab = reinterpret_cast<float*>(smem + 4 * KV);
  constexpr int XCH = (NS - 1) * NQ * 64 * 8;
  float* xch = dtab + tl4;                                              // [2][NS-1][NQ][64 lanes][8] f32 : dQ partials of splits 1.., double-buffered by sequence parity
  float* tabs = xch + 2 * XCH;                                          // this head's bias-table column
  int* rcs = reinterpret_cast<int*>(tabs + tl4);
  unsigned char* regs = reinterpret_cast<unsigned char*>(rcs + lr4);    // region ids of the current window position
  const bool want_dtab = pb.dbias_table != nullptr;
  const int ql = wave % NQ, kh = wave / NQ;                             // query tile slot, key split
  const int qt = qg * NQ + ql;
  const int q = qt * 16 + r;
  const bool active = qt < NX;
  const bool qv = active && (q < L);
  const int tbase = kh * NH;                                            // first key tile of this split
  const int ntl = (NX - tbase < NH) ? NX - tbase : NH;                  // tiles in this split

  for (int i = tid; i < p.table_len; i += NWV * 64) { dtab[i] = 0.f; tabs[i] = p.bias_table[(size_t)i * heads + h]; }
  for (int i = tid; i < L; i += NWV * 64) rcs[i] = p.rc[i];
  uint32_t bm[NH * 2];
  auto build_bm = [&]() {                                 // bias + shift mask of this wave's score block, packed bf16 (from LDS)
    // (vector reads + selects, as in attn_fwd_win2_kernel; only a lane whose four keys straddle L takes the per-element path)
    const int rcq = rcs[qv ? q : 0] + p.rc0;
    const int regq = MASK ? regs[qv ? q : 0] : 0;
#pragma unroll
    for (int tl = 0; tl < NH; ++tl) {
      const int key0 = (tbase + tl) * 16 + g * 4;
      const bool on = tl < ntl;
      const bool whole = on && key0 + 4 <= L;
      const int k0c = whole ? key0 : 0;
      const int4 rk = *reinterpret_cast<const int4*>(rcs + k0c);
      const uint32_t gk = MASK ? *reinterpret_cast<const uint32_t*>(regs + k0c) : 0u;
      const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
      float b4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float b = tabs[rcq - rks[j]];
        if (MASK) b += ((int)((gk >> (8 * j)) & 0xffu) != regq) ? -100.f : 0.f;
        b4[j] = whole ? b : NEG_INF;
      }
      if (on && !whole) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = key0 + j;
          if (key < L) {
            float b = tabs[rcq - rcs[key]];
            if (MASK) b += (regs[key] != regq) ? -100.f : 0.f;
            b4[j] = b;
          }
        }
      }
      bm[2 * tl] = pack_bf2(b4[0], b4[1]);
      bm[2 * tl + 1] = pack_bf2(b4[2], b4[3]);
      if ((tl & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
  };
  f32x2 racc[NH][2];
#pragma unroll
  for (int tl = 0; tl < NH; ++tl) { racc[tl][0] = f32x2{0.f, 0.f}; racc[tl][1] = f32x2{0.f, 0.f}; }

  // this workgroup's run of sequences, WINDOW-major (i = w * B + clip): the bias block only changes with the window position
  const int total = nWin * B;
  const int per = (total + nch - 1) / nch;
  const int b0 = ch * per, b1 = (b0 + per < total) ? b0 + per : total;
  // Address generation is kept out of the per-sequence path (it was ~35% of the loop): per-lane element offsets inside a
  // sequence and the DMA source offsets are computed once; per sequence only wave-uniform bases change.
  const int w0 = b0 / B;
  int w_nx = w0, c_nx = b0 - w0 * B;                    // (window position, clip) of the NEXT sequence to request
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == B) { c_nx = 0; ++w_nx; } };
  const uint32_t off_q = (uint32_t)q * p.ld_qkv + p.q_off + h * HD + g * 8;
  const uint32_t off_do = (uint32_t)q * pb.ld_dout + h * HD + g * 8;
  const uint32_t off_o = (uint32_t)q * p.ld_out + h * HD + g * 8;
  const uint32_t off_dq = (uint32_t)q * pb.ld_dqkv + p.q_off + h * HD + g * 4;
  const uint32_t off_ls = (uint32_t)h * L + q;
  constexpr int NF = (LP32 * 4 + NWV * 64 - 1) / (NWV * 64);            // 16-byte DMA requests per thread per image
  static_assert(NF <= NPH, "the fill requests are spread over the tile pairs");
  uint32_t goff[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int u = i * NWV * 64 + tid, row = u >> 2, chs = u & 3;
    goff[i] = (u < LP32 * 4) ? (uint32_t)((row * p.ld_qkv + ((chs ^ swz_chunk<32>(row)) << 3)) * 2) : 0xffffffffu;   // (beyond the image: out of range -> no-op)
  }
  const unsigned fill_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2);
  auto issue = [&](size_t seq, int buf) {
    const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv + h * HD;
    fill_pre<NF, NWV * 64 * 16>(smem + buf * 2 * KV + (tid & ~63) * 16, KV, qkv + p.k_off, qkv + p.v_off, fill_bytes, goff);
  };
  bf16x8 qf, dof, of;
  float lse_n = 0.f, ss_n = 1.0f;                       // raw prefetched values: NO arithmetic on them before the next iteration
  auto fetch = [&](size_t seq) {
    const u16* qb = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
    const u16* dob = reinterpret_cast<const u16*>(pb.dout) + seq * L * pb.ld_dout;
    const u16* ob = reinterpret_cast<const u16*>(p.out) + seq * L * p.ld_out;
    qf = load_frag_global(qb + off_q, qv);
    dof = load_frag_global(dob + off_do, qv);
    of = load_frag_global(ob + off_o, qv);
    lse_n = qv ? (p.lse + seq * heads * L)[off_ls] : __builtin_huge_valf();      // (a use here would drain the whole DMA queue)
    ss_n = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  };
  if (b0 < b1) { issue(seq_nx(), 0); fetch(seq_nx()); advance(); }

  f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};     // this half's partial of the PREVIOUS clip (kh == 0 keeps it)
  auto flush_prev = [&](size_t seq, int par) {           // split 0: add the other splits' partials (in LDS buffer `par`) and store dQ of clip bprev
    if (kh == 0 && qv) {
      float4 x0 = make_float4(dq[0][0], dq[0][1], dq[0][2], dq[0][3]), x1 = make_float4(dq[1][0], dq[1][1], dq[1][2], dq[1][3]);
#pragma unroll
      for (int o = 0; o < NS - 1; ++o) {
        const float* x = xch + par * XCH + ((o * NQ + ql) * 64 + lane) * 8;
        const float4 y0 = *reinterpret_cast<const float4*>(x), y1 = *reinterpret_cast<const float4*>(x + 4);
        x0.x += y0.x; x0.y += y0.y; x0.z += y0.z; x0.w += y0.w;
        x1.x += y1.x; x1.y += y1.y; x1.z += y1.z; x1.w += y1.w;
      }
      u16* dqp = reinterpret_cast<u16*>(pb.dqkv) + seq * L * pb.ld_dqkv + off_dq;
      const float sc = p.scale;
      *reinterpret_cast<uint2*>(dqp) = make_uint2(pack_bf2(x0.x * sc, x0.y * sc), pack_bf2(x0.z * sc, x0.w * sc));
      *reinterpret_cast<uint2*>(dqp + 16) = make_uint2(pack_bf2(x1.x * sc, x1.y * sc), pack_bf2(x1.z * sc, x1.w * sc));
    }
  };

  int wprev = -1, w_cu = w0, c_cu = b0 - w0 * B;        // (window position, clip) of the sequence being processed
  size_t seq_pv = 0;
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    fill_wait();
    __syncthreads();                                      // sequence b landed; other buffer free; partials of b-1 visible; tables staged
    if (b > b0) flush_prev(seq_pv, cur ^ 1);
    seq_pv = seq;
    const int wcur = w_cu;
    if (++c_cu == B) { c_cu = 0; ++w_cu; }
    if (wcur != wprev) {                                  // (workgroup-uniform) new window position: rebuild the bias + mask block
      wprev = wcur;
      if (MASK) {
        __syncthreads();                                  // everyone is done with the previous window's region row
        for (int i = tid; i < L; i += NWV * 64) regs[i] = p.region[(size_t)wcur * L + i];
        __syncthreads();
      }
      if (MASK || b == b0) build_bm();
    }
    const bf16x8 cqf = qf, cdof = dof, cof = of;
    const float clse2 = lse_n * LOG2E, seq_scale = ss_n;
    const bool has_next = b + 1 < b1;                      // next sequence: Q/dO/O fragments now, the K/V fill spread over the tile loop
    const u16* kv_nx = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    unsigned char* dst_nx = smem + (cur ^ 1) * 2 * KV + (tid & ~63) * 16;
    if (has_next) {
      fetch(seq_nx());
      advance();
      if (!active) fill_pre<NF, NWV * 64 * 16>(dst_nx, KV, kv_nx + p.k_off, kv_nx + p.v_off, fill_bytes, goff);
    }
    // (no second barrier per sequence: the partials of sequence b go to exchange buffer b & 1, split 0 reads them after the NEXT
    //  loop-top barrier, and that buffer is written again only two sequences later)
    if (!active) continue;
    const unsigned char* Ksm = smem + cur * 2 * KV;
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)cdof[e] * (float)cof[e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (kh == 0 && g == 0 && qv) (pb.delta + seq * heads * L)[off_ls] = dl;
    dq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    // broadcast operands of the packed math as REAL register pairs: with an op_sel broadcast the compiler pairs the scalar with
    // an arbitrary neighbour register, and when that neighbour is the landing register of a prefetch load the hazard pass
    // inserts s_waitcnt vmcnt(0) in the middle of the tile loop (draining the DMA queue every sequence)
    f32x2 nl2 = {-clse2, -clse2}, ss2 = {seq_scale, seq_scale}, ndl2 = {-dl, -dl};
    asm volatile("" : "+v"(nl2), "+v"(ss2), "+v"(ndl2));
    // lane bases: every tile of this half is a compile-time immediate away (the 16-byte chunk swizzle only depends on row % 16)
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const unsigned char* kb = Ksm + k_off_swz<HD>(tbase * 16 + r, g);
    const unsigned char* tk0 = Ksm + k_off_swz<HD>(tbase * 16 + g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8;
    const unsigned char* tk1 = Ksm + k_off_swz<HD>(tbase * 16 + g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8;
    const uint32_t tk0a = lds_addr(tk0), tk1a = lds_addr(tk1);
    // software pipeline over tile pairs: the K / V fragments of pair c+1 and the transposed K operand of pair c are requested
    // before the VALU work on pair c, so the LDS latency sits under the exp / multiply chain instead of in front of the MFMAs
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      kf[u] = *reinterpret_cast<const bf16x8*>(kb + u * 1024);
      vf[u] = *reinterpret_cast<const bf16x8*>(kb + KV + u * 1024);
    }
#pragma unroll
    for (int c = 0; c < NPH; ++c) {
      f32x4 s4[2], dp4[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tl = 2 * c + u;
        if (tl < NH) {                                    // (tl >= ntl in the last split: bias = -inf -> p = 0, rows are zero-filled)
          float b0f, b1f, b2f, b3f;                       // volatile: keeps the unpack inside the loop (else 4*NH VGPRs get hoisted)
          asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(b0f) : "v"(bm[2 * tl]));
          asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(b1f) : "v"(bm[2 * tl]));
          asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(b2f) : "v"(bm[2 * tl + 1]));
          asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(b3f) : "v"(bm[2 * tl + 1]));
          s4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], cqf, f32x4{b0f, b1f, b2f, b3f}, 0, 0, 0);
          dp4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], cdof, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
      }
      // transposed K operand of this pair.  Inline asm, not the builtin: the compiler treats the builtin as "may alias the
      // in-flight LDS DMA" and puts s_waitcnt vmcnt(0) in front of it, i.e. drains the next sequence's K/V fill every pair.
      s16x4 a0, a1, c0, c1;
      tr_read4(a0, a1, c0, c1, tk0a, tk1a, c * 2048);
      if (c < NF && has_next)
        fill_one(dst_nx + c * NWV * 64 * 16, KV, kv_nx + p.k_off, kv_nx + p.v_off, fill_bytes, goff[c], c == NF - 1);
      if (c + 1 < NPH) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + 2 + u < NH) {
            kf[u] = *reinterpret_cast<const bf16x8*>(kb + (2 * c + 2 + u) * 1024);
            vf[u] = *reinterpret_cast<const bf16x8*>(kb + KV + (2 * c + 2 + u) * 1024);
          }
        }
      }
      // explicit 2-wide f32 math (v_pk_fma / v_pk_mul / v_pk_add) on the accumulator register pairs; the running dS sums are
      // updated in place (volatile asm: the compiler otherwise sinks all 4*NH adds below the loop and keeps every dS alive)
      uint32_t dsw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int tl = 2 * c + u;
        if (tl < NH) {
#pragma unroll
          for (int hj = 0; hj < 2; ++hj) {
            const f32x2 sv = {s4[u][2 * hj], s4[u][2 * hj + 1]}, dpv = {dp4[u][2 * hj], dp4[u][2 * hj + 1]};
            const f32x2 e = __builtin_elementwise_fma(sv, f32x2{LOG2E, LOG2E}, nl2);
            const f32x2 pr = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            const f32x2 d = pr * __builtin_elementwise_fma(dpv, ss2, ndl2);
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(racc[tl][hj]) : "v"(d));
            dsw[2 * u + hj] = pack_bf2v(d);
          }
        }
      }
      const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(dsw[0], dsw[1], dsw[2], dsw[3]));
      tr_wait4(a0, a1, c0, c1);
      const s16x8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const s16x8 v1 = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      dq[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v0), dsf, dq[0], 0, 0, 0);
      dq[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v1), dsf, dq[1], 0, 0, 0);
    }
    if (kh > 0) {                                         // splits 1.. publish their partials; split 0 adds them after the next barrier
      float* x = xch + cur * XCH + (((kh - 1) * NQ + ql) * 64 + lane) * 8;
      *reinterpret_cast<float4*>(x) = make_float4(dq[0][0], dq[0][1], dq[0][2], dq[0][3]);
      *reinterpret_cast<float4*>(x + 4) = make_float4(dq[1][0], dq[1][1], dq[1][2], dq[1][3]);
    }
  }
  __syncthreads();
  if (b1 > b0 && active) flush_prev(seq_pv, (b1 - 1 - b0) & 1);
  if (want_dtab) {
    if (qv) {
      const int rcq = rcs[q] + p.rc0;
#pragma unroll
      for (int tl = 0; tl < NH; ++tl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = (tbase + tl) * 16 + g * 4 + j;
          if (tl < ntl && key < L) atomicAdd(&dtab[rcq - rcs[key]], racc[tl][j >> 1][j & 1]);
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < p.table_len; i += NWV * 64) {
      const float v = dtab[i];
      if (v != 0.f) atomicAdd(pb.dbias_table + (size_t)i * heads + h, v);
    }
  }
}

// ================================================================================================
// backward B: dK, dV (per key tile; probabilities recomputed from lse; delta from kernel A)
// ================================================================================================
template <int HD, int MODE, int NW, int NXB, bool CAUSAL = false, int KTP = 0, bool PIPEP = true>
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
    lse_s[i] = i < L ? lse_g[i] * ((MODE == 1 && !CAUSAL) ? 1.4426950408889634f : 1.0f) : __builtin_huge_valf();       // +inf -> p = 0 for padded queries (fusion build: log2 units)
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
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = has_drop ? 65536.f / (65536.f - (float)thr16) : 1.f;
  const int nt = sm.nt, nt2 = sm.nt2;
  constexpr bool FAST1 = (MODE == 1) && !CAUSAL;
  const float sc2 = p.scale * 1.4426950408889634f, cdk = seq_scale * keep;

  // each wave owns TWO key tiles at a time: the Q / dO fragments (k = hd) and the transposed Q^T / dO^T fragments
  // (k = tokens) are read from LDS once and feed both tiles -> half the LDS traffic per MFMA.
  constexpr int KT = KTP ? KTP : (HD == 32) ? 2 : 1;
  auto kloop = [&](auto drop_c) {                       // (compile-time dropout tests, as in attn_bwd_dq_kernel)
  constexpr int DROPM = decltype(drop_c)::value;          // 0 no dropout, 1 Philox decisions, 2 the forward's stored decisions (drop_mask; exact-tile fusion build)
  constexpr bool DROP = DROPM == 1;
  static_assert(DROPM != 2 || (KT == 1 && NXB > 0 && FAST1), "stored-mask path: one key tile per wave, exact tile count");
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

    uint4 own[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) own[t] = make_uint4(0, 0, 0, 0);
    // stored decisions: record (query tile qt, this key tile) holds element (query 16 qt + r', key 16 kt + 4 g' + j') as bit 16 g' + r' of
    // word j'; this lane (key r, queries 4g..4g+3) needs word r & 3, bits 16 (r >> 2) + 4 g + j: one dword per query tile, all NXB loaded up front
    uint32_t wm[NXB ? NXB : 1];
    const int mshift = 16 * ((r >> 2) & 1) + 4 * g;
    if (DROPM == 2) {
      const uint32_t* mcol = p.drop_mask + ((size_t)(seq * heads + h) * nt * nt + kp) * 8 + 2 * (r & 3) + (r >> 3);
#pragma unroll
      for (int qt = 0; qt < NXB; ++qt) wm[qt] = mcol[(size_t)qt * nt * 8];
    }
    // score / dP MFMAs of one pair of query tiles (software-pipelined by one pair in the exact-tile fusion build, as in the dQ kernel)
    auto score_pair = [&](int c, f32x4 (&so)[2][KT], f32x4 (&dpo)[2][KT]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * c + u;
#pragma unroll
        for (int t = 0; t < KT; ++t) { so[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dpo[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        if (NXB ? (qt < NXB) : (qt < nt)) {
          const int qrow = qt * 16 + r;            // A-operand row owned by this lane
#pragma unroll
          for (int s = 0; s < HD / 32; ++s) {
            const bf16x8 qf = frag_hd<HD>(Qsm, qrow, s * 4 + g);
            const bf16x8 dof = frag_hd<HD>(dOsm, qrow, s * 4 + g);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              so[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[t][s], so[u][t], 0, 0, 0);
              dpo[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[t][s], dpo[u][t], 0, 0, 0);
            }
          }
        }
      }
    };
    constexpr bool PIPE = FAST1 && NXB > 0 && PIPEP;
    f32x4 s4c[2][KT], dpc[2][KT];
    if (PIPE) score_pair(0, s4c, dpc);
#pragma unroll
    for (int c = 0; c < (NXB ? (NXB + 1) / 2 : nt2); ++c) {
      f32x4 s4n[2][KT], dpn[2][KT];
      if (PIPE) { if (c + 1 < (NXB + 1) / 2) score_pair(c + 1, s4n, dpn); }
      else score_pair(c, s4c, dpc);
      float pt[KT][2][4], ds[KT][2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * c + u;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) { pt[t][u][j] = 0.f; ds[t][u][j] = 0.f; }
        if (NXB ? (qt < NXB) : (qt < nt)) {
          f32x4 s4[KT], dp4[KT];
#pragma unroll
          for (int t = 0; t < KT; ++t) { s4[t] = s4c[u][t]; dp4[t] = dpc[u][t]; }
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
              if (DROP) {                               // quad lane i evaluates the block of query tile (qt & ~3) + i
                if ((qt & 3) == 0) own[t] = drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)((qt + (lane & 3)) * 4 + g), (uint32_t)(key[t] >> 2));
                blk = (c & 1) ? (u ? quad_bcast<3>(own[t]) : quad_bcast<2>(own[t])) : (u ? quad_bcast<1>(own[t]) : quad_bcast<0>(own[t]));
              }
              if (FAST1) {
                // log2 domain, packed: the key's additive mask and the queries' lse fold into one offset per query; the keep scale
                // of P rides on the final dV scale
                const float kbk = (kv[t] && regk[t]) ? 0.f : NEG_INF;
                const f32x2 off01 = f32x2{kbk, kbk} - f32x2{ls[0], ls[1]}, off23 = f32x2{kbk, kbk} - f32x2{ls[2], ls[3]};
                const f32x2 x01 = __builtin_elementwise_fma(f32x2{s4[t][0], s4[t][1]}, f32x2{sc2, sc2}, off01);
                const f32x2 x23 = __builtin_elementwise_fma(f32x2{s4[t][2], s4[t][3]}, f32x2{sc2, sc2}, off23);
                const float pr[4] = {__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1]), __builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
                const f32x2 c2 = f32x2{cdk, cdk};
                const f32x2 d01 = f32x2{dp4[t][0], dp4[t][1]} * c2, d23 = f32x2{dp4[t][2], dp4[t][3]} * c2;
                // dS = P * (mask * dP - delta) = (mask * P) * dP - P * delta: ONE select per element (the masked P, which dV needs anyway)
                float pj[4] = {pr[0], pr[1], pr[2], pr[3]};
                if (DROP) {
#pragma unroll
                  for (int j = 0; j < 4; ++j) {
                    const bool dropped = drop_field(u4_get(blk, j), key[t] & 3) < thr16;
                    pj[j] = dropped ? 0.f : pj[j];
                  }
                }
                if (DROPM == 2) {
                  const uint32_t kept = ~(wm[NXB ? qt : 0] >> mshift);                     // bit j: element j is kept
#pragma unroll
                  for (int j = 0; j < 4; ++j) pj[j] = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, pj[j]) & (uint32_t)__builtin_amdgcn_sbfe((int)kept, j, 1));
                }
                const f32x2 o01 = __builtin_elementwise_fma(f32x2{pj[0], pj[1]}, d01, -(f32x2{pr[0], pr[1]} * f32x2{dls[0], dls[1]}));
                const f32x2 o23 = __builtin_elementwise_fma(f32x2{pj[2], pj[3]}, d23, -(f32x2{pr[2], pr[3]} * f32x2{dls[2], dls[3]}));
                pt[t][u][0] = pj[0]; pt[t][u][1] = pj[1]; pt[t][u][2] = pj[2]; pt[t][u][3] = pj[3];
                ds[t][u][0] = o01[0]; ds[t][u][1] = o01[1]; ds[t][u][2] = o23[0]; ds[t][u][3] = o23[1];
              } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                bool ok = kv[t] && regk[t];
                if (CAUSAL) ok = (key[t] < p.causal_from) ? ok : (kv[t] && q0 + j >= p.causal_from && key[t] <= q0 + j);
                const float pr = ok ? __expf(s4[t][j] * p.scale - ls[j]) : 0.f;
                float dpj = dp4[t][j] * seq_scale, pj = pr;
                if (DROP) {
                  const bool dropped = drop_field(u4_get(blk, j), key[t] & 3) < thr16;
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
      if (PIPE) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int t = 0; t < KT; ++t) { s4c[u][t] = s4n[u][t]; dpc[u][t] = dpn[u][t]; }
        __builtin_amdgcn_sched_barrier(0);
      } else if (NXB) {
        asm volatile("" ::: "memory");             // exact-tile build: keep the unrolled pairs' LDS reads from being hoisted en bloc
      }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kv[t]) {
        u16* base = reinterpret_cast<u16*>(pb.dqkv) + ((size_t)seq * L + key[t]) * pb.ld_dqkv + h * HD + g * 4;
        const float ksc = (MODE == 1) ? p.scale : 1.0f;
        const float vsc = FAST1 ? cdk : seq_scale;           // fusion build: P carries no keep scale, dV does
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          *reinterpret_cast<uint2*>(base + p.k_off + dt * 16) =
              make_uint2(pack_bf2(dk[t][dt][0] * ksc, dk[t][dt][1] * ksc), pack_bf2(dk[t][dt][2] * ksc, dk[t][dt][3] * ksc));
          *reinterpret_cast<uint2*>(base + p.v_off + dt * 16) =
              make_uint2(pack_bf2(dv[t][dt][0] * vsc, dv[t][dt][1] * vsc), pack_bf2(dv[t][dt][2] * vsc, dv[t][dt][3] * vsc));
        }
      }
    }
  }
  };
  if (has_drop) {
    if constexpr (KT == 1 && NXB > 0 && FAST1) { if (p.drop_mask) kloop(std::integral_constant<int, 2>{}); else kloop(std::integral_constant<int, 1>{}); }
    else kloop(std::integral_constant<int, 1>{});
  } else kloop(std::integral_constant<int, 0>{});
}


// ================================================================================================
// backward B' (window mode, exact tile count): batch-persistent dK / dV, the transposed twin of attn_bwd_dq_win2_kernel.
// A workgroup = (head, key group, chunk of window-major sequences); each of its 7 waves owns TWO key tiles whose K / V
// fragments live in registers, and whose bias + shift-mask block against ALL query tiles is built once per window position
// into packed-bf16 registers (the score MFMA's C operand).  Per sequence the Q / dO images plus lse / delta stream into the
// other LDS buffer by direct-to-LDS DMA spread over the query-tile loop; the next sequence's K / V fragments are prefetched
// into registers.  Per score element: unpack, fma + exp2, packed fma / mul, two bf16 conversions.
// ================================================================================================
template <int NX, bool MASK, int KT, int NWV>
__global__ __launch_bounds__(NWV * 64) void attn_bwd_dkv_win2_kernel(const vmvm_attn_bwd_desc pb, const int nkg, const int nch) {
  constexpr int HD = 32, NPQ = (NX + 1) / 2;                            // NPQ query-tile pairs
  constexpr int LP32 = NPQ * 32, IMG = LP32 * HD * 2;                   // rows / bytes of one Q or dO image
  constexpr int LV = 512;                                               // floats reserved for lse / delta (L <= 512)
  constexpr int BUF = 2 * IMG + 2 * LV * 4;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nkg);
  const int kg = logical % nkg;
  const int t1 = logical / nkg;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  const int tl4 = (p.table_len + 3) & ~3, lr4 = (L + 3) & ~3;
  float* tabs = reinterpret_cast<float*>(smem + 2 * BUF);               // this head's bias-table column
  int* rcs = reinterpret_cast<int*>(tabs + tl4);
  unsigned char* regs = reinterpret_cast<unsigned char*>(rcs + lr4);    // region ids of the current window position
  const int kp = kg * NWV + wave;                                       // key-tile pair of this wave
  const bool active = kp * KT < NX;
  int key[KT]; bool kv[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) { key[t] = (kp * KT + t) * 16 + r; kv[t] = active && key[t] < L; }

  for (int i = tid; i < p.table_len; i += NWV * 64) tabs[i] = p.bias_table[(size_t)i * heads + h];
  for (int i = tid; i < L; i += NWV * 64) rcs[i] = p.rc[i];
  uint32_t bm[KT][NX][2];
  auto build_bm = [&]() {                                 // bias + mask of (query tiles x this wave's keys), packed bf16 (from LDS)
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const int rck = rcs[kv[t] ? key[t] : 0] - p.rc0;
      const int regk = MASK ? regs[kv[t] ? key[t] : 0] : 0;
#pragma unroll
      for (int qt = 0; qt < NX; ++qt) {                   // (vector reads + selects, as in attn_fwd_win2_kernel)
        if constexpr (KT > 1) {                           // two key tiles per wave (196-token windows): no registers to spare for the vector form
          float b4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = qt * 16 + g * 4 + j;
            float b = NEG_INF;
            if (kv[t] && q < L) {
              b = tabs[rcs[q] - rck];
              if (MASK) b += (regs[q] != regk) ? -100.f : 0.f;
            }
            b4[j] = b;
          }
          bm[t][qt][0] = pack_bf2(b4[0], b4[1]);
          bm[t][qt][1] = pack_bf2(b4[2], b4[3]);
          continue;
        }
        const int q0 = qt * 16 + g * 4;
        const bool whole = kv[t] && ((qt < NX - 1) || (q0 + 4 <= L));
        const int q0c = whole ? q0 : 0;
        const int4 rq = *reinterpret_cast<const int4*>(rcs + q0c);
        const uint32_t gq = MASK ? *reinterpret_cast<const uint32_t*>(regs + q0c) : 0u;
        const int rqs[4] = {rq.x, rq.y, rq.z, rq.w};
        float b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float b = tabs[rqs[j] - rck];
          if (MASK) b += ((int)((gq >> (8 * j)) & 0xffu) != regk) ? -100.f : 0.f;
          b4[j] = whole ? b : NEG_INF;
        }
        if (kv[t] && !whole && qt == NX - 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = q0 + j;
            if (q < L) {
              float b = tabs[rcs[q] - rck];
              if (MASK) b += (regs[q] != regk) ? -100.f : 0.f;
              b4[j] = b;
            }
          }
        }
        bm[t][qt][0] = pack_bf2(b4[0], b4[1]);
        bm[t][qt][1] = pack_bf2(b4[2], b4[3]);
        if ((qt & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  const int total = nWin * B;
  const int per = (total + nch - 1) / nch;
  const int b0 = ch * per, b1 = (b0 + per < total) ? b0 + per : total;
  const int w0 = b0 / B;
  int w_nx = w0, c_nx = b0 - w0 * B;                    // (window position, clip) of the NEXT sequence to request
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == B) { c_nx = 0; ++w_nx; } };
  // per-lane constant offsets (elements) inside one sequence; per sequence only wave-uniform bases change
  uint32_t off_k[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) off_k[t] = (uint32_t)key[t] * p.ld_qkv + h * HD + g * 8;
  uint32_t off_dk[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) off_dk[t] = (uint32_t)key[t] * pb.ld_dqkv + h * HD + g * 4;
  constexpr int NF = (LP32 * 4 + NWV * 64 - 1) / (NWV * 64);            // 16-byte DMA requests per thread per image
  static_assert(NF + 1 <= NPQ, "the fill requests are spread over the query-tile pairs");
  uint32_t goq[NF], god[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int u = i * NWV * 64 + tid, row = u >> 2, chs = u & 3, cs = (chs ^ swz_chunk<32>(row)) << 3;
    goq[i] = (u < LP32 * 4) ? (uint32_t)((row * p.ld_qkv + cs) * 2) : 0xffffffffu;
    god[i] = (u < LP32 * 4) ? (uint32_t)((row * pb.ld_dout + cs) * 2) : 0xffffffffu;
  }
  const unsigned q_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2), do_bytes = (unsigned)(((size_t)(L - 1) * pb.ld_dout + HD) * 2);
  // requests of the next sequence's images; step i in [0, NF) = image chunk i, step NF = lse (waves 0-1) / delta (waves 2-3)
  auto dma_step = [&](size_t seq, int buf, int i) {
    unsigned char* dst = smem + buf * BUF;
    if (i < NF) {
      const u16* qsrc = uniform_ptr(reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv + p.q_off + h * HD);
      const u16* dsrc = uniform_ptr(reinterpret_cast<const u16*>(pb.dout) + seq * L * pb.ld_dout + h * HD);
      dma16_pair(dst + (tid & ~63) * 16 + i * NWV * 64 * 16, IMG, qsrc, q_bytes, goq[i], dsrc, do_bytes, god[i], i == NF - 1);
    } else if (wave < 4) {
      const float* src = uniform_ptr(((wave < 2) ? p.lse : pb.delta) + (seq * heads + h) * L);
      dma16_one(dst + 2 * IMG + (wave >> 1) * LV * 4 + (wave & 1) * 1024, src, (unsigned)(L * 4), (uint32_t)(((wave & 1) * 64 + lane) * 16));
    }
  };
  bf16x8 kf[KT], vf[KT];                                 // prefetched K / V fragments of the next sequence
  float ss_n = 1.0f;
  auto fetch = [&](size_t seq) {
    const u16* qb = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      kf[t] = load_frag_global(qb + p.k_off + off_k[t], kv[t]);
      vf[t] = load_frag_global(qb + p.v_off + off_k[t], kv[t]);
    }
    ss_n = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  };
  if (b0 < b1) {
#pragma unroll
    for (int i = 0; i <= NF; ++i) dma_step(seq_nx(), 0, i);
    fetch(seq_nx());
    advance();
  }

  int wprev = -1, w_cu = w0, c_cu = b0 - w0 * B;        // (window position, clip) of the sequence being processed
  for (int b = b0; b < b1; ++b) {
    const int cur = (b - b0) & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    fill_wait();
    __syncthreads();                                      // sequence b landed for everyone; everyone left the other buffer
    const int wcur = w_cu;
    if (++c_cu == B) { c_cu = 0; ++w_cu; }
    if (wcur != wprev) {                                  // (workgroup-uniform) new window position: rebuild the bias + mask block
      wprev = wcur;
      if (MASK) {
        if (b > b0) __syncthreads();                      // everyone is done with the previous window's region row
        for (int i = tid; i < L; i += NWV * 64) regs[i] = p.region[(size_t)wcur * L + i];
        __syncthreads();
      }
      if (MASK || b == b0) build_bm();
    }
    bf16x8 ckf[KT], cvf[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) { ckf[t] = kf[t]; cvf[t] = vf[t]; }
    const float seq_scale = ss_n;
    const bool has_next = b + 1 < b1;
    const size_t sq_nx = seq_nx();
    if (has_next) {
      fetch(sq_nx);
      advance();
      if (!active) {
#pragma unroll
        for (int i = 0; i <= NF; ++i) dma_step(sq_nx, cur ^ 1, i);
      }
    }
    if (!active) continue;
    const unsigned char* Qs = smem + cur * BUF;
    const unsigned char* dOs = Qs + IMG;
    const float* lse_s = reinterpret_cast<const float*>(Qs + 2 * IMG);
    const float* delta_s = lse_s + LV;
    const uint32_t lse_a = lds_addr(reinterpret_cast<const unsigned char*>(lse_s + g * 4)), delta_a = lds_addr(reinterpret_cast<const unsigned char*>(delta_s + g * 4));
    f32x4 dk[KT][2], dv[KT][2];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) { dk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x2 ss2 = {seq_scale, seq_scale};
    asm volatile("" : "+v"(ss2));                         // real register pair (see attn_bwd_dq_win2_kernel)
    // lane bases: every query tile is a compile-time immediate away (the chunk swizzle only depends on row % 16)
    const unsigned char* qb_ = Qs + k_off_swz<HD>(r, g);
    const unsigned char* tq0 = Qs + k_off_swz<HD>(g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8;
    const unsigned char* tq1 = Qs + k_off_swz<HD>(g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8;
    const uint32_t tq0a = lds_addr(tq0), tq1a = lds_addr(tq1);
#pragma unroll
    for (int c = 0; c < NPQ; ++c) {
      uint32_t pw[KT][4], dw[KT][4];
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int x = 0; x < 4; ++x) { pw[t][x] = 0u; dw[t][x] = 0u; }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int qt = 2 * c + u;
        if (qt < NX) {
          const bf16x8 qf = *reinterpret_cast<const bf16x8*>(qb_ + qt * 1024);
          const bf16x8 dof = *reinterpret_cast<const bf16x8*>(qb_ + IMG + qt * 1024);
          f32x4 l4, d4;
          lds_read2_b128(l4, d4, lse_a, delta_a, qt * 64);
          lds_wait2(l4, d4);
          f32x2 nl[2] = {f32x2{-l4[0] * LOG2E, -l4[1] * LOG2E}, f32x2{-l4[2] * LOG2E, -l4[3] * LOG2E}};
          f32x2 nd[2] = {f32x2{-d4[0], -d4[1]}, f32x2{-d4[2], -d4[3]}};
#pragma unroll
          for (int t = 0; t < KT; ++t) {
            float b0f, b1f, b2f, b3f;                     // volatile: keeps the unpack inside the loop (else 4x the registers get hoisted)
            asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(b0f) : "v"(bm[t][qt][0]));
            asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(b1f) : "v"(bm[t][qt][0]));
            asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(b2f) : "v"(bm[t][qt][1]));
            asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(b3f) : "v"(bm[t][qt][1]));
            const f32x4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, ckf[t], f32x4{b0f, b1f, b2f, b3f}, 0, 0, 0);
            const f32x4 dp4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, cvf[t], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
              const f32x2 sv = {s4[2 * hj], s4[2 * hj + 1]}, dpv = {dp4[2 * hj], dp4[2 * hj + 1]};
              const f32x2 e = __builtin_elementwise_fma(sv, f32x2{LOG2E, LOG2E}, nl[hj]);
              const f32x2 pr = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
              const f32x2 d = pr * __builtin_elementwise_fma(dpv, ss2, nd[hj]);
              pw[t][2 * u + hj] = pack_bf2v(pr);
              dw[t][2 * u + hj] = pack_bf2v(d);
            }
          }
        }
      }
      // transposed Q / dO operands of this query-tile pair (asm reads: see tr_read4)
      s16x4 a0, a1, c0, c1, e0, e1, f0, f1;
      tr_read4(a0, a1, c0, c1, tq0a, tq1a, c * 2048);
      tr_read4(e0, e1, f0, f1, tq0a, tq1a, IMG + c * 2048);
      if (c <= NF && has_next) dma_step(sq_nx, cur ^ 1, c);
      tr_wait4(a0, a1, c0, c1);
      tr_wait4(e0, e1, f0, f1);
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      const s16x8 q0v = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const s16x8 q1v = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
      const s16x8 d0v = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
      const s16x8 d1v = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pw[t][0], pw[t][1], pw[t][2], pw[t][3]));
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(dw[t][0], dw[t][1], dw[t][2], dw[t][3]));
        dv[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, d0v), pf, dv[t][0], 0, 0, 0);
        dv[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, d1v), pf, dv[t][1], 0, 0, 0);
        dk[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q0v), dsf, dk[t][0], 0, 0, 0);
        dk[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q1v), dsf, dk[t][1], 0, 0, 0);
      }
    }
    u16* dbase = reinterpret_cast<u16*>(pb.dqkv) + seq * L * pb.ld_dqkv;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kv[t]) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          *reinterpret_cast<uint2*>(dbase + off_dk[t] + p.k_off + dt * 16) = make_uint2(pack_bf2(dk[t][dt][0], dk[t][dt][1]), pack_bf2(dk[t][dt][2], dk[t][dt][3]));
          *reinterpret_cast<uint2*>(dbase + off_dk[t] + p.v_off + dt * 16) =
              make_uint2(pack_bf2(dv[t][dt][0] * seq_scale, dv[t][dt][1] * seq_scale), pack_bf2(dv[t][dt][2] * seq_scale, dv[t][dt][3] * seq_scale));
        }
      }
    }
  }
}

// ================================================================================================
// streaming variants (any L; used above 448 tokens: Swin-L-384 windows of 8x12x12 = 1152 tokens and the 2352-token fusion
// sequences of 16 x 384^2 clips).  Same lane layout and math as the kernels above, but K / V (forward, dQ) or Q / dO (dK/dV) pass
// through LDS in chunks of KC tokens and the softmax is the online (running max / running sum) form.  A workgroup owns NW query
// tiles (forward, dQ) or NW*KT key tiles (dK/dV) of one (sequence, head); blocks of the same (sequence, head) are adjacent in the
// XCD remap so the chunks they re-read stay in that XCD's L2.
// ================================================================================================
struct SmemS { int lpk, off_b, off_rc, off_reg, off_tab, off_lse, off_delta, total; };
// which: 0 fwd / dq / dbias, 2 dkv (+lse/delta chunk)
__host__ __device__ inline SmemS smem_stream(int L, int hd, int mode, int table_len, int which, int KC) {
  SmemS s;
  s.lpk = (L + KC - 1) / KC * KC;
  int o = KC * hd * 2;
  s.off_b = o; o += KC * hd * 2;
  s.off_rc = o; if (mode == 0) o += s.lpk * 4;
  s.off_reg = o; o += s.lpk;
  s.off_tab = o; if (mode == 0) o += ((table_len + 3) & ~3) * 4;
  s.off_lse = o; if (which == 2) o += KC * 4;
  s.off_delta = o; if (which == 2) o += KC * 4;
  s.total = (o + 15) & ~15;
  return s;
}

// Scores live in the log2 domain (s2 = s * log2 e: the table is staged pre-multiplied, the q k^T term enters through one fma), so
// the softmax needs exp2 only; MASK = false: no region map (un-shifted windows), the compare/select/add per element is compiled out;
// the key < L guard runs only in the last chunk.
template <int HD, int MODE, int NW, int KC, bool MASK>
__global__ __launch_bounds__(NW * 64, 4) void attn_fwd_stream_kernel(const vmvm_attn_fwd_desc p, const int nqb) {
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NTC = KC / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const SmemS sm = smem_stream(L, HD, MODE, p.table_len, 0, KC);
  const int logical = xcd_remap(blockIdx.x, p.nseq * heads * nqb);
  const int sh = logical / nqb, qb = logical - sh * nqb;
  const int seq = sh / heads, h = sh - seq * heads;
  const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
  unsigned char* Ksm = smem;
  unsigned char* Vsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);
  if (MODE == 0) {
    for (int i = tid; i < sm.lpk; i += NW * 64) {
      rc[i] = i < L ? p.rc[i] : 0;
      reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
    }
    for (int i = tid; i < p.table_len; i += NW * 64) tab[i] = p.bias_table[(size_t)i * heads + h] * LOG2E;
  } else {
    for (int i = tid; i < sm.lpk; i += NW * 64) reg[i] = (i < L) ? (p.keymask ? p.keymask[(size_t)seq * L + i] : 1) : 0;
  }
  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = has_drop ? 65536.f / (65536.f - (float)thr16) : 1.f;

  const int qt = qb * NW + wave;
  const int q = qt * 16 + r;
  const bool qv = q < L;
  const u16* qp = qkv + (size_t)q * p.ld_qkv + p.q_off + h * HD + g * 8;
  bf16x8 qf[HD / 32];
#pragma unroll
  for (int s = 0; s < HD / 32; ++s) qf[s] = load_frag_global(qp + s * 32, qv);
  f32x4 o[HD / 16];
#pragma unroll
  for (int dt = 0; dt < HD / 16; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -3.0e38f, lsum = 0.f;                // running max (finite start: exp(-inf - m) = 0, never inf - inf) and this lane's share of the sum

  for (int k0 = 0; k0 < L; k0 += KC) {
    __syncthreads();                             // every wave is done with the previous chunk (and, first time, the tables are staged)
    fill_rowmajor<HD>(Ksm, qkv + (size_t)k0 * p.ld_qkv + p.k_off + h * HD, p.ld_qkv, L - k0, KC, tid, NW * 64);
    fill_rowmajor<HD>(Vsm, qkv + (size_t)k0 * p.ld_qkv + p.v_off + h * HD, p.ld_qkv, L - k0, KC, tid, NW * 64);
    fill_wait();
    __syncthreads();
    const int rcq = (MODE == 0) ? rc[qv ? q : 0] + p.rc0 : 0;
    const int regq = (MODE == 0 && MASK) ? reg[qv ? q : 0] : 0;
    const bool tail = k0 + KC > L;
    const float sc2 = p.scale * LOG2E;
    f32x4 acc[NTC];
    float cmx = NEG_INF;
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int row = t * 16 + r;
#pragma unroll
      for (int s = 0; s < HD / 32; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ksm + k_off_swz<HD>(row, s * 4 + g));
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc[t], 0, 0, 0);
      }
      const int key0 = k0 + t * 16 + g * 4;
      if (MODE == 0) {
        const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
        const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
        int gks[4] = {0, 0, 0, 0};
        if (MASK) { const uchar4 gk = *reinterpret_cast<const uchar4*>(reg + key0); gks[0] = gk.x; gks[1] = gk.y; gks[2] = gk.z; gks[3] = gk.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float s = __builtin_fmaf(acc[t][j], LOG2E, tab[rcq - rks[j]]);
          if (MASK) s += (regq != gks[j] ? -100.f * LOG2E : 0.f);
          acc[t][j] = s;
        }
      } else {
        const uchar4 mk = *reinterpret_cast<const uchar4*>(reg + key0);
        const int mks[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = mks[j] ? acc[t][j] * sc2 : NEG_INF;
      }
      if (MODE == 0 && tail) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (key0 + j < L) ? acc[t][j] : NEG_INF;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) cmx = fmaxf(cmx, acc[t][j]);
    }
    cmx = fmaxf(cmx, __shfl_xor(cmx, 16, 64));
    cmx = fmaxf(cmx, __shfl_xor(cmx, 32, 64));
    const float mnew = fmaxf(m, cmx);
    const float alpha = __builtin_amdgcn_exp2f(m - mnew);
    m = mnew;
    float csum = 0.f;
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float e = __builtin_amdgcn_exp2f(acc[t][j] - mnew); acc[t][j] = e; csum += e; }
    }
    lsum = lsum * alpha + csum;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) o[dt] *= alpha;
    if (has_drop) {
      uint4 own = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
        if ((t & 3) == 0)                           // quad lane i: block of (absolute) tile t+i; chunks are whole groups of four tiles
          own = quad_transpose(drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q >> 2), (uint32_t)((k0 / 16 + t + (lane & 3)) * 4 + g)), lane & 1, lane & 2);
        const uint32_t w = u4_static(own, t & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = (drop_field(w, j) < thr16) ? 0.f : acc[t][j] * keep;
      }
    }
#pragma unroll
    for (int c = 0; c < NTC / 2; ++c) {
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
  float sum = lsum;
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  if (g == 0 && qv) p.lse[((size_t)seq * heads + h) * L + q] = (m + __builtin_amdgcn_logf(sum)) * LN2;      // v_log_f32 = log2
  if (qv) {
    const float inv = seq_scale / sum;
    u16* op = reinterpret_cast<u16*>(p.out) + ((size_t)seq * L + q) * p.ld_out + h * HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt)
      *reinterpret_cast<uint2*>(op + dt * 16) = make_uint2(pack_bf2(o[dt][0] * inv, o[dt][1] * inv), pack_bf2(o[dt][2] * inv, o[dt][3] * inv));
  }
}

// dQ (+ delta): workgroup (sequence chunk, head, query block) walks its sequences (the per-head table is staged once); K / V
// stream through LDS in KC-token chunks.  The bias-table gradient has its own kernel (attn_bwd_dbias_stream_kernel).
template <int HD, int MODE, int NW, int KC, bool MASK>
__global__ __launch_bounds__(NW * 64, 4) void attn_bwd_dq_stream_kernel(const vmvm_attn_bwd_desc pb, const int nchunks, const int nqb) {
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const SmemS sm = smem_stream(L, HD, MODE, p.table_len, 0, KC);
  const int logical = xcd_remap(blockIdx.x, nchunks * heads * nqb);
  const int ch = logical / nqb, qb = logical - ch * nqb;
  const int chunk = ch / heads, h = ch - chunk * heads;
  unsigned char* Ksm = smem;
  unsigned char* Vsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = has_drop ? 65536.f / (65536.f - (float)thr16) : 1.f;
  if (MODE == 0) {
    for (int i = tid; i < sm.lpk; i += NW * 64) rc[i] = i < L ? p.rc[i] : 0;
    for (int i = tid; i < p.table_len; i += NW * 64) tab[i] = p.bias_table[(size_t)i * heads + h] * LOG2E;     // log2-domain scores
  }
  const int qt = qb * NW + wave;
  const int q = qt * 16 + r;
  const bool qv = q < L;

  for (int seq = chunk; seq < p.nseq; seq += nchunks) {
    const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
    const u16* dO = reinterpret_cast<const u16*>(pb.dout) + (size_t)seq * L * pb.ld_dout + h * HD;
    const u16* O = reinterpret_cast<const u16*>(p.out) + (size_t)seq * L * p.ld_out + h * HD;
    __syncthreads();                                // previous sequence's last chunk is consumed before its mask image changes
    if (MODE == 0) {
      for (int i = tid; i < sm.lpk; i += NW * 64) reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
    } else {
      for (int i = tid; i < sm.lpk; i += NW * 64) reg[i] = (i < L) ? (p.keymask ? p.keymask[(size_t)seq * L + i] : 1) : 0;
    }
    const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
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
    dl += __shfl_xor(dl, 32, 64);
    if (g == 0 && qv) pb.delta[((size_t)seq * heads + h) * L + q] = dl;
    const float lse2 = qv ? p.lse[((size_t)seq * heads + h) * L + q] * LOG2E : __builtin_huge_valf();   // padded query: p = exp2(-inf) = 0
    const float sc2 = p.scale * LOG2E;
    f32x4 dq[HD / 16];
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < L; k0 += KC) {
      __syncthreads();
      fill_rowmajor<HD>(Ksm, qkv + (size_t)k0 * p.ld_qkv + p.k_off + h * HD, p.ld_qkv, L - k0, KC, tid, NW * 64);
      fill_rowmajor<HD>(Vsm, qkv + (size_t)k0 * p.ld_qkv + p.v_off + h * HD, p.ld_qkv, L - k0, KC, tid, NW * 64);
      fill_wait();
      __syncthreads();
      const int rcq = (MODE == 0) ? rc[qv ? q : 0] + p.rc0 : 0;
      const int regq = (MODE == 0 && MASK) ? reg[qv ? q : 0] : 0;
      const bool tail = k0 + KC > L;
      uint4 own = make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int c = 0; c < KC / 32; ++c) {
        float ds[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = 2 * c + u;
          const int row = t * 16 + r;
          f32x4 s4 = f32x4{0.f, 0.f, 0.f, 0.f}, dp4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < HD / 32; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ksm + k_off_swz<HD>(row, s * 4 + g));
            s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], s4, 0, 0, 0);
            const bf16x8 vf = frag_hd<HD>(Vsm, row, s * 4 + g);
            dp4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[s], dp4, 0, 0, 0);
          }
          const int key0 = k0 + t * 16 + g * 4;
          uint32_t w = 0;
          if (has_drop) {
            if ((t & 3) == 0) own = quad_transpose(drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)(q >> 2), (uint32_t)((k0 / 16 + t + (lane & 3)) * 4 + g)), lane & 1, lane & 2);
            w = u4_static(own, t & 3);
          }
          if (MODE == 0) {
            const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
            const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
            int gks[4] = {0, 0, 0, 0};
            if (MASK) { const uchar4 gk = *reinterpret_cast<const uchar4*>(reg + key0); gks[0] = gk.x; gks[1] = gk.y; gks[2] = gk.z; gks[3] = gk.w; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float s = __builtin_fmaf(s4[j], LOG2E, tab[rcq - rks[j]]) - lse2;
              if (MASK) s += (regq != gks[j] ? -100.f * LOG2E : 0.f);
              float pr = __builtin_amdgcn_exp2f(s);
              if (tail) pr = (key0 + j < L) ? pr : 0.f;
              ds[u][j] = pr * __builtin_fmaf(dp4[j], seq_scale, -dl);
            }
          } else {
            const uchar4 mk = *reinterpret_cast<const uchar4*>(reg + key0);
            const int mks[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float pr = mks[j] ? __builtin_amdgcn_exp2f(__builtin_fmaf(s4[j], sc2, -lse2)) : 0.f;
              float dpj = dp4[j] * seq_scale;
              if (has_drop) dpj = (drop_field(w, j) < thr16) ? 0.f : dpj * keep;
              ds[u][j] = pr * (dpj - dl);
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
    }
    if (qv) {
      u16* dqp = reinterpret_cast<u16*>(pb.dqkv) + ((size_t)seq * L + q) * pb.ld_dqkv + p.q_off + h * HD + g * 4;
      const float sc = p.scale;
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt)
        *reinterpret_cast<uint2*>(dqp + dt * 16) = make_uint2(pack_bf2(dq[dt][0] * sc, dq[dt][1] * sc), pack_bf2(dq[dt][2] * sc, dq[dt][3] * sc));
    }
  }
}

// Relative-position-bias table gradient of the streaming window path.  Scattering every dS element into an LDS table copy from
// the dQ kernel costs one conflicting LDS atomic per score element (47% of the config-5 step); instead a workgroup of THIS kernel
// owns a fixed (head, query block, key chunk), recomputes S / dP for its block over the sequences and sums dS in registers (the
// accumulator layout of the score tiles), so the scatter through rc[i]-rc[j]+rc0 happens once per workgroup.  Sequences are
// walked window-major: the bias + shift-mask block of a window position is built once (it is the score MFMA's C operand) and
// serves that position's sequence of every clip -- per score element only exp2 and three fmas remain.  Runs after the dQ kernel
// (reads delta).
template <int NW, int KC>
__global__ __launch_bounds__(NW * 64) void attn_bwd_dbias_stream_kernel(const vmvm_attn_bwd_desc pb, const int nchunks, const int nqb, const int nwin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HD = 32, NTC = KC / 16;
  constexpr float LOG2E = 1.4426950408889634f;
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const SmemS sm = smem_stream(L, HD, 0, p.table_len, 0, KC);
  const int nkc = (L + KC - 1) / KC;
  const int logical = xcd_remap(blockIdx.x, nchunks * heads * nqb * nkc);
  const int kc = logical % nkc, rest = logical / nkc;
  const int qb = rest % nqb, ch = rest / nqb;
  const int chunk = ch / heads, h = ch - chunk * heads;
  const int k0 = kc * KC;
  constexpr int BUF = 2 * KC * HD * 2;               // one (K chunk, V chunk) pair; TWO of them sit in front of the stream layout's tables
  unsigned char* Ksm = smem;
  unsigned char* Vsm = smem + KC * HD * 2;
  int* rc = reinterpret_cast<int*>(smem + BUF + sm.off_rc);
  float* tab = reinterpret_cast<float*>(smem + BUF + sm.off_tab);
  float* dtab = tab;                                 // the gradient copy reuses the table's LDS once the sequence loop is done
  for (int i = tid; i < sm.lpk; i += NW * 64) rc[i] = i < L ? p.rc[i] : 0;
  for (int i = tid; i < p.table_len; i += NW * 64) tab[i] = p.bias_table[(size_t)i * heads + h];
  __syncthreads();
  const int qt = qb * NW + wave;
  const int q = qt * 16 + r;
  const bool qv = q < L;
  const int rcq = rc[qv ? q : 0];
  const int nclip = p.nseq / nwin;                   // host guarantees nwin divides nseq (nwin = nseq otherwise)
  f32x4 racc[NTC];
#pragma unroll
  for (int t = 0; t < NTC; ++t) racc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int w = chunk; w < nwin; w += nchunks) {
    // bias + mask block of this window position against the key chunk (-inf: padded key / padded query -> p = 0)
    f32x4 bm[NTC];
    {
      const unsigned char* regw = p.region ? p.region + (size_t)(w % p.n_win) * L : nullptr;
      const int regq = (regw && qv) ? regw[q] : 0;
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
        const int key0 = k0 + t * 16 + g * 4;
        const int4 rk = *reinterpret_cast<const int4*>(rc + key0);
        const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = qv && key0 + j < L;
          float v = tab[rcq - rks[j] + p.rc0];
          if (regw && ok && regq != regw[key0 + j]) v -= 100.f;
          bm[t][j] = ok ? v : NEG_INF;
        }
      }
    }
    // The clip loop is latency-bound (16 MFMAs per wave and clip against a DMA fill, four global loads and a barrier), so it is
    // software-pipelined: clip b+1's K / V chunk streams into the other LDS buffer and its Q / dO / lse / delta rows into registers
    // while clip b is computed; the fragments are read with inline asm (a plain load would make the compiler drain that prefetch).
    auto seq_of = [&](int b) { return (size_t)b * nwin + w; };
    auto issue = [&](int b, int buf) {
      const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + seq_of(b) * L * p.ld_qkv;
      fill_rowmajor<HD>(Ksm + buf * BUF, qkv + (size_t)k0 * p.ld_qkv + p.k_off + h * HD, p.ld_qkv, L - k0, KC, tid, NW * 64);
      fill_rowmajor<HD>(Vsm + buf * BUF, qkv + (size_t)k0 * p.ld_qkv + p.v_off + h * HD, p.ld_qkv, L - k0, KC, tid, NW * 64);
    };
    bf16x8 qf_n, dof_n;
    float nl_n = 0.f, ndl_n = 0.f, ss_n = 1.0f;
    auto fetch = [&](int b) {
      const size_t seq = seq_of(b);
      const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
      const u16* dO = reinterpret_cast<const u16*>(pb.dout) + seq * L * pb.ld_dout + h * HD;
      ss_n = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
      qf_n = load_frag_global(qkv + (size_t)q * p.ld_qkv + p.q_off + h * HD + g * 8, qv);
      dof_n = load_frag_global(dO + (size_t)q * pb.ld_dout + g * 8, qv);
      nl_n = qv ? p.lse[(seq * heads + h) * L + q] : 0.f;
      ndl_n = qv ? pb.delta[(seq * heads + h) * L + q] : 0.f;
    };
    __syncthreads();                                   // everyone is done with both buffers (previous window position)
    issue(0, 0);
    fetch(0);
    for (int b = 0; b < nclip; ++b) {
      const int cur = b & 1;
      fill_wait();
      __syncthreads();                                 // clip b's chunk landed for everyone; everyone left the other buffer
      const bf16x8 qf = qf_n, dof = dof_n;
      const float nl = -nl_n * LOG2E, ndl = -ndl_n, ss = ss_n;
      if (b + 1 < nclip) { issue(b + 1, cur ^ 1); fetch(b + 1); }
      const uint32_t ka = lds_addr(Ksm + cur * BUF + k_off_swz<HD>(r, g)), va = lds_addr(Vsm + cur * BUF + k_off_swz<HD>(r, g));
      f32x4 kraw[NTC], vraw[NTC];
#pragma unroll
      for (int t = 0; t < NTC; ++t) lds_read2_b128(kraw[t], vraw[t], ka, va, t * 1024);
#pragma unroll
      for (int t = 0; t < NTC; ++t) {
        lds_wait2(kraw[t], vraw[t]);                   // ties THIS tile's registers to the wait (all 2 * NTC reads are in flight together: only the first one waits)
        const f32x4 s4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kraw[t]), qf, bm[t], 0, 0, 0);
        const f32x4 dp4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vraw[t]), dof, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(s4[j], LOG2E, nl));
          racc[t][j] = __builtin_fmaf(pr, __builtin_fmaf(dp4[j], ss, ndl), racc[t][j]);
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < p.table_len; i += NW * 64) dtab[i] = 0.f;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NTC; ++t) {
    const int key0 = k0 + t * 16 + g * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (qv && key0 + j < L && racc[t][j] != 0.f) atomicAdd(&dtab[rcq - rc[key0 + j] + p.rc0], racc[t][j]);
  }
  __syncthreads();
  for (int i = tid; i < p.table_len; i += NW * 64) {
    const float v = dtab[i];
    if (v != 0.f) atomicAdd(pb.dbias_table + (size_t)i * heads + h, v);
  }
}

// dK / dV: workgroup = (sequence, head, block of NW*KT key tiles); Q / dO (+ lse, delta) stream through LDS in KC-query chunks.
template <int HD, int MODE, int NW, int KC, bool MASK>
__global__ __launch_bounds__(NW * 64, 4) void attn_bwd_dkv_stream_kernel(const vmvm_attn_bwd_desc pb, const int nkb) {
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, g = lane >> 4;
  const int L = p.L, heads = p.heads;
  const SmemS sm = smem_stream(L, HD, MODE, p.table_len, 2, KC);
  const int logical = xcd_remap(blockIdx.x, p.nseq * heads * nkb);
  const int sh = logical / nkb, kb = logical - sh * nkb;
  const int seq = sh / heads, h = sh - seq * heads;
  const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv;
  const u16* dO = reinterpret_cast<const u16*>(pb.dout) + (size_t)seq * L * pb.ld_dout + h * HD;
  unsigned char* Qsm = smem;
  unsigned char* dOsm = smem + sm.off_b;
  int* rc = reinterpret_cast<int*>(smem + sm.off_rc);
  unsigned char* reg = smem + sm.off_reg;
  float* tab = reinterpret_cast<float*>(smem + sm.off_tab);
  float* lse_s = reinterpret_cast<float*>(smem + sm.off_lse);
  float* delta_s = reinterpret_cast<float*>(smem + sm.off_delta);
  const float* lse_g = p.lse + ((size_t)seq * heads + h) * L;
  const float* delta_g = pb.delta + ((size_t)seq * heads + h) * L;
  for (int i = tid; i < sm.lpk; i += NW * 64) {
    if (MODE == 0) {
      rc[i] = i < L ? p.rc[i] : 0;
      reg[i] = (p.region && i < L) ? p.region[(size_t)(seq % p.n_win) * L + i] : 0;
    } else {
      reg[i] = (i < L) ? (p.keymask ? p.keymask[(size_t)seq * L + i] : 1) : 0;
    }
  }
  if (MODE == 0)
    for (int i = tid; i < p.table_len; i += NW * 64) tab[i] = p.bias_table[(size_t)i * heads + h] * LOG2E;      // log2-domain scores
  __syncthreads();

  const float seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  const bool has_drop = (MODE == 1) && p.dropout_p > 0.f;
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = has_drop ? 65536.f / (65536.f - (float)thr16) : 1.f;
  const int nt = (L + 15) / 16;
  const float sc2 = p.scale * LOG2E;

  constexpr int KT = (HD == 32) ? 2 : 1;
  int key[KT]; bool kv[KT];
  bf16x8 kf[KT][HD / 32], vf[KT][HD / 32];
  int rck[KT], regk[KT];
  f32x4 dk[KT][HD / 16], dv[KT][HD / 16];
  uint4 own[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    key[t] = ((kb * NW + wave) * KT + t) * 16 + r;
    kv[t] = key[t] < L;
#pragma unroll
    for (int s = 0; s < HD / 32; ++s) {
      kf[t][s] = load_frag_global(qkv + (size_t)key[t] * p.ld_qkv + p.k_off + h * HD + g * 8 + s * 32, kv[t]);
      vf[t][s] = load_frag_global(qkv + (size_t)key[t] * p.ld_qkv + p.v_off + h * HD + g * 8 + s * 32, kv[t]);
    }
    rck[t] = (MODE == 0) ? rc[kv[t] ? key[t] : 0] - p.rc0 : 0;
    regk[t] = reg[kv[t] ? key[t] : 0];
    own[t] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) { dk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  for (int q0c = 0; q0c < L; q0c += KC) {
    __syncthreads();
    fill_rowmajor<HD>(Qsm, qkv + (size_t)q0c * p.ld_qkv + p.q_off + h * HD, p.ld_qkv, L - q0c, KC, tid, NW * 64);
    fill_rowmajor<HD>(dOsm, dO + (size_t)q0c * pb.ld_dout, pb.ld_dout, L - q0c, KC, tid, NW * 64);
    for (int i = tid; i < KC; i += NW * 64) {
      lse_s[i] = (q0c + i < L) ? lse_g[q0c + i] * LOG2E : __builtin_huge_valf();     // log2 domain; +inf -> p = 0 for padded queries
      delta_s[i] = (q0c + i < L) ? delta_g[q0c + i] : 0.f;
    }
    fill_wait();
    __syncthreads();
#pragma unroll
    for (int c = 0; c < KC / 32; ++c) {
      float pt[KT][2][4], ds[KT][2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ql = 2 * c + u;                    // query tile inside the chunk
        const int qt = q0c / 16 + ql;                // absolute query tile
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) { pt[t][u][j] = 0.f; ds[t][u][j] = 0.f; }
        if (qt < nt) {
          const int qrow = ql * 16 + r;
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
          const int ql0 = ql * 16 + g * 4, q0 = q0c + ql0;
          const float4 l4 = *reinterpret_cast<const float4*>(lse_s + ql0);
          const float4 d4 = *reinterpret_cast<const float4*>(delta_s + ql0);
          const float ls[4] = {l4.x, l4.y, l4.z, l4.w};
          const float dls[4] = {d4.x, d4.y, d4.z, d4.w};
          if (MODE == 0) {
            const int4 rq = *reinterpret_cast<const int4*>(rc + q0);
            const int rqs[4] = {rq.x, rq.y, rq.z, rq.w};
            int gqs[4] = {0, 0, 0, 0};
            if (MASK) { const uchar4 gq = *reinterpret_cast<const uchar4*>(reg + q0); gqs[0] = gq.x; gqs[1] = gq.y; gqs[2] = gq.z; gqs[3] = gq.w; }
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float sv = __builtin_fmaf(s4[t][j], LOG2E, tab[rqs[j] - rck[t]]) - ls[j];
                if (MASK) sv += (gqs[j] != regk[t] ? -100.f * LOG2E : 0.f);
                const float pr = kv[t] ? __builtin_amdgcn_exp2f(sv) : 0.f;
                pt[t][u][j] = pr;
                ds[t][u][j] = pr * __builtin_fmaf(dp4[t][j], seq_scale, -dls[j]);
              }
          } else {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              uint4 blk = make_uint4(0, 0, 0, 0);
              if (has_drop) {                               // quad lane i evaluates the block of query tile (qt & ~3) + i
                if ((qt & 3) == 0) own[t] = drop_block(p.seed, p.offset, (uint32_t)(seq * heads + h), (uint32_t)((qt + (lane & 3)) * 4 + g), (uint32_t)(key[t] >> 2));
                blk = (c & 1) ? (u ? quad_bcast<3>(own[t]) : quad_bcast<2>(own[t])) : (u ? quad_bcast<1>(own[t]) : quad_bcast<0>(own[t]));
              }
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float pr = (kv[t] && regk[t]) ? __builtin_amdgcn_exp2f(__builtin_fmaf(s4[t][j], sc2, -ls[j])) : 0.f;
                float dpj = dp4[t][j] * seq_scale, pj = pr;
                if (has_drop) {
                  const bool dropped = drop_field(u4_get(blk, j), key[t] & 3) < thr16;
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
  if (d->L > 16384) return VMVM_ENOSUPPORT;        // dropout block ids carry key/4 in 12 bits; above 448 tokens the streaming kernels run
  if (d->region && d->n_win <= 0) return VMVM_EINVAL;
  if (d->seq_scale && d->seqs_per_scale <= 0) return VMVM_EINVAL;
  if (d->causal_from < 0 || (d->causal_from > 0 && d->mode != 1)) return VMVM_EINVAL;
  if (d->att_colsum && (d->mode != 1 || d->causal_from > 0)) return VMVM_EINVAL;
  if (d->att_colsum && (d->L > 448 || (d->stream_min_len > 0 && d->L >= d->stream_min_len))) return VMVM_ENOSUPPORT;
  if (d->causal_from > 0 && (d->L > 448 || (d->stream_min_len > 0 && d->L >= d->stream_min_len))) return VMVM_ENOSUPPORT;   // no streaming seq2seq build
  return VMVM_OK;
}

inline bool use_stream(const vmvm_attn_fwd_desc* d) { return d->L > 448 || (d->stream_min_len > 0 && d->L >= d->stream_min_len); }
// stored dropout decisions (vmvm_attn_fwd_desc.drop_mask): the exact-tile fusion kernels only
inline bool drop_mask_ok(const vmvm_attn_fwd_desc* d) {
  return d->mode == 1 && d->head_dim == 64 && d->dropout_p > 0.f && d->causal_from <= 0 && !d->att_colsum && !use_stream(d) && d->L == 432;      // exact tiles only: the stored-decision kernels have no per-tile guards (a ragged 27th tile would record rows / keys >= L)
}

}  // namespace

// win_layout = 1 kernels (attention_win3.hip)
namespace vmvm_w3 {
bool applicable(const vmvm_attn_fwd_desc* d);
int launch_dkv(const vmvm_attn_bwd_desc* d, hipStream_t st);
int launch_dq(const vmvm_attn_bwd_desc* d, hipStream_t st);
int launch_fwd(const vmvm_attn_fwd_desc* d, hipStream_t st);
int64_t dbias_ws_size(const vmvm_attn_bwd_desc* d);
}  // namespace vmvm_w3
// key-blocked win_layout = 1 kernels (attention_win4.hip, round 5): the same problems as vmvm_w3::applicable
namespace vmvm_w4 {
int launch_fwd(const vmvm_attn_fwd_desc* d, hipStream_t st);
int launch_dkv(const vmvm_attn_bwd_desc* d, hipStream_t st);
}  // namespace vmvm_w4

// A/B switches of the win_layout = 1 kernels, read once: VMVM_NO_WIN3 (all), VMVM_NO_WIN3_FWD (1), VMVM_NO_WIN3_DQ (2) fall back to the
// order-agnostic win2 kernels on the same layout (tools/scratch/ab_win3*.sh)
static bool w3_off(int which) {
  static const int bits = (getenv("VMVM_NO_WIN3") ? 7 : 0) | (getenv("VMVM_NO_WIN3_FWD") ? 2 : 0) | (getenv("VMVM_NO_WIN3_DQ") ? 4 : 0);
  return (bits >> which) & 1;
}

// A/B switches of the key-blocked kernels, read once: VMVM_NO_WIN4 (all), VMVM_NO_WIN4_FWD / _DQ / _DKV fall back to the round-4 win3 kernels
static bool w4_off(int which) {
  static const int bits = (getenv("VMVM_NO_WIN4") ? 7 : 0) | (getenv("VMVM_NO_WIN4_FWD") ? 1 : 0) | (getenv("VMVM_NO_WIN4_DQ") ? 2 : 0) | (getenv("VMVM_NO_WIN4_DKV") ? 4 : 0);
  return (bits >> which) & 1;
}

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
  if (!vmvm_hook::W4_TIMELINE_BUILD && d->drop_mask && !drop_mask_ok(d)) return VMVM_ENOSUPPORT;      // (timeline probe builds pass their stamp buffer as drop_mask, tools/scratch/w4_timeline.py)
  if (use_stream(d)) {                               // streaming kernels: K / V chunks of 128 tokens, 8 query tiles per workgroup
    constexpr int NWS = 8, KCS = 128;
    const SmemS ss = smem_stream(d->L, d->head_dim, d->mode, d->table_len, 0, KCS);
    const int nqb = ((d->L + 15) / 16 + NWS - 1) / NWS;
    const int grid = d->nseq * d->heads * nqb;
#define LAUNCH_FWD_S(HD, MODE, MASK)                                                                                 \
    do {                                                                                                             \
      int rc_ = set_smem(attn_fwd_stream_kernel<HD, MODE, NWS, KCS, MASK>, ss.total);                                 \
      if (rc_) return rc_;                                                                                           \
      hipLaunchKernelGGL((attn_fwd_stream_kernel<HD, MODE, NWS, KCS, MASK>), dim3(grid), dim3(NWS * 64), ss.total, st, *d, nqb); \
    } while (0)
    if (d->mode == 0) { if (d->region) LAUNCH_FWD_S(32, 0, true); else LAUNCH_FWD_S(32, 0, false); }
    else LAUNCH_FWD_S(64, 1, false);
    VMVM_CHECK_LAUNCH();
    return VMVM_OK;
  }
  const Smem sm = smem_layout(d->L, d->head_dim, d->mode, d->table_len, 0);
  const int nb = d->nseq * d->heads;
  const bool mask = d->region != nullptr;
  if (d->mode == 0) {
    // batch-persistent kernel for the step's exact window shapes (392 / 196 tokens); per-(sequence, head) kernel otherwise
    const int nwin = d->n_win > 0 ? d->n_win : 1;
    const int npk = (sm.nt + 1) / 2, tl4 = (d->table_len + 3) & ~3, lr4 = (d->L + 3) & ~3;
    const int smem2 = 4 * npk * 32 * 64 + tl4 * 4 + lr4 * 4 + ((lr4 + 15) & ~15);
    const bool pers_ok = (d->nseq % nwin == 0) && (sm.nt == 25 || sm.nt == 13) && d->dropout_p == 0.f && smem2 <= 160 * 1024;
    if (vmvm_w3::applicable(d) && !d->att_colsum && !w3_off(1) && !w4_off(0)) {
      int rc_ = vmvm_w4::launch_fwd(d, st);
      if (rc_) return rc_;
    } else if (vmvm_w3::applicable(d) && !d->att_colsum && !w3_off(1)) {
      int rc_ = vmvm_w3::launch_fwd(d, st);
      if (rc_) return rc_;
    } else if (pers_ok) {
      const int nqg = (sm.nt + 6) / 7;
      const int base = d->heads * nqg;
      int nch = 1; float best = 1e30f;
      for (int c = 1; c <= 64 && c <= d->nseq; ++c) {
        const float cost = (float)((base * c + 255) / 256) * ((float)((d->nseq + c - 1) / c) + 3.f);
        if (cost < best - 1e-6f) { best = cost; nch = c; }
      }
#define LAUNCH_FWD2(NX, MASK)                                                                                        \
      do {                                                                                                           \
        int rc_ = set_smem(attn_fwd_win2_kernel<NX, MASK>, smem2);                                                    \
        if (rc_) return rc_;                                                                                         \
        hipLaunchKernelGGL((attn_fwd_win2_kernel<NX, MASK>), dim3(base * nch), dim3(448), smem2, st, *d, nqg, nch);    \
      } while (0)
      if (sm.nt == 25) { if (mask) LAUNCH_FWD2(25, true); else LAUNCH_FWD2(25, false); }
      else { if (mask) LAUNCH_FWD2(13, true); else LAUNCH_FWD2(13, false); }
    }
    else if (sm.nt == 25) { if (mask) LAUNCH_FWD(32, 0, 26, 4, 25, true); else LAUNCH_FWD(32, 0, 26, 4, 25, false); }
    else if (sm.nt == 13) { if (mask) LAUNCH_FWD(32, 0, 14, 4, 13, true); else LAUNCH_FWD(32, 0, 14, 4, 13, false); }
    else if (sm.nt <= 16) LAUNCH_FWD(32, 0, 16, 4, 0, true);
    else LAUNCH_FWD(32, 0, 28, 4, 0, true);
  } else {
    // (exact-NT instantiations of the head_dim-64 kernel need the compiler barrier in the score loop: hoisted en bloc, the K
    //  fragment loads of 27 unguarded tiles exceed 256 VGPRs)
    if (d->att_colsum) {                                            // get_att pass: + attention column sums
      int rc_ = set_smem(attn_fwd_kernel<64, 1, 28, 8, 0, true, false, true>, sm.total);
      if (rc_) return rc_;
      hipLaunchKernelGGL((attn_fwd_kernel<64, 1, 28, 8, 0, true, false, true>), dim3(nb), dim3(8 * 64), sm.total, st, *d);
    }
    else if (d->causal_from > 0) {                                  // seq2seq mask (smtm pass): generic tile loops
      int rc_ = set_smem(attn_fwd_kernel<64, 1, 28, 8, 0, true, true>, sm.total);
      if (rc_) return rc_;
      hipLaunchKernelGGL((attn_fwd_kernel<64, 1, 28, 8, 0, true, true>), dim3(nb), dim3(8 * 64), sm.total, st, *d);
    }
    else if (sm.nt <= 16) LAUNCH_FWD(64, 1, 16, 4, 0, true);
    else if (sm.nt == 27) LAUNCH_FWD(64, 1, 28, 8, 27, true);       // L = 432 (fusion encoder): exact tile count, no per-tile guards       // L = 432 (fusion encoder): exact tile count, no per-tile guards
    else LAUNCH_FWD(64, 1, 28, 8, 0, true);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

#define LAUNCH_BWD(KERN, HD, MODE, NW, WHICH, NXB)                                           \
  do {                                                                                       \
    const Smem s_ = smem_layout(d->f.L, d->f.head_dim, d->f.mode, d->f.table_len, WHICH);    \
    int rc_ = set_smem(KERN<HD, MODE, NW, NXB>, s_.total);                                   \
    if (rc_) return rc_;                                                                     \
    hipLaunchKernelGGL((KERN<HD, MODE, NW, NXB>), dim3(nb), dim3(NW * 64), s_.total, st, *d); \
    VMVM_CHECK_LAUNCH();                                                                     \
  } while (0)
#define LAUNCH_BWD_DQ(HD, MODE, NW, NXB)                                                     \
  do {                                                                                       \
    const Smem s_ = smem_layout(d->f.L, d->f.head_dim, d->f.mode, d->f.table_len, 1);        \
    int rc_ = set_smem(attn_bwd_dq_kernel<HD, MODE, NW, NXB>, s_.total);                     \
    if (rc_) return rc_;                                                                     \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, MODE, NW, NXB>), dim3(nchunks * d->f.heads), dim3(NW * 64), s_.total, st, *d, nchunks); \
    VMVM_CHECK_LAUNCH();                                                                     \
  } while (0)

extern "C" int vmvm_attention_bwd_table_is_separate(const vmvm_attn_bwd_desc* d) {
  return (d && d->f.mode == 0 && d->dbias_table && d->f.L > 0 && use_stream(&d->f)) ? 1 : 0;
}

extern "C" int vmvm_attention_bwd(const vmvm_attn_bwd_desc* d, void* stream) {
  if (!d) return VMVM_EINVAL;
  int rc = check_desc(&d->f);
  if (rc) return rc;
  if (d->table_phase < 0 || d->table_phase > 2) return VMVM_EINVAL;
  if (!d->dout || !d->dqkv || !d->delta || (d->ld_dout & 7) || (d->ld_dqkv & 7)) return VMVM_EINVAL;
  if (!vmvm_hook::W4_TIMELINE_BUILD && d->f.drop_mask && !drop_mask_ok(&d->f)) return VMVM_ENOSUPPORT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (use_stream(&d->f)) {                           // streaming kernels (see vmvm_attention_fwd)
    constexpr int NWS = 8, KCS = 128;
    const int nt_ = (d->f.L + 15) / 16;
    const SmemS sa = smem_stream(d->f.L, d->f.head_dim, d->f.mode, d->f.table_len, 0, KCS);
    const SmemS sb = smem_stream(d->f.L, d->f.head_dim, d->f.mode, d->f.table_len, 2, KCS);
    const int nqb = (nt_ + NWS - 1) / NWS;
    int nchs = 2048 / (d->f.heads * nqb);             // ~8 workgroups per CU; a workgroup stages the per-head table once for its sequences
    if (nchs < 1) nchs = 1;
    if (nchs > d->f.nseq || d->f.mode == 1) nchs = d->f.nseq;
    const int kt_ = d->f.mode == 0 ? 2 : 1;
    const int nkb = (nt_ + NWS * kt_ - 1) / (NWS * kt_);
#define LAUNCH_BWD_S(HD, MODE, MASK)                                                                                 \
    do {                                                                                                             \
      int rc_ = set_smem(attn_bwd_dq_stream_kernel<HD, MODE, NWS, KCS, MASK>, sa.total);                              \
      if (rc_) return rc_;                                                                                           \
      hipLaunchKernelGGL((attn_bwd_dq_stream_kernel<HD, MODE, NWS, KCS, MASK>), dim3(nchs * d->f.heads * nqb), dim3(NWS * 64), sa.total, st, *d, nchs, nqb); \
      VMVM_CHECK_LAUNCH();                                                                                           \
      rc_ = set_smem(attn_bwd_dkv_stream_kernel<HD, MODE, NWS, KCS, MASK>, sb.total);                                 \
      if (rc_) return rc_;                                                                                           \
      hipLaunchKernelGGL((attn_bwd_dkv_stream_kernel<HD, MODE, NWS, KCS, MASK>), dim3(d->f.nseq * d->f.heads * nkb), dim3(NWS * 64), sb.total, st, *d, nkb); \
      VMVM_CHECK_LAUNCH();                                                                                           \
    } while (0)
    if (d->table_phase != 2) {
      if (d->f.mode == 0) { if (d->f.region) LAUNCH_BWD_S(32, 0, true); else LAUNCH_BWD_S(32, 0, false); }
      else LAUNCH_BWD_S(64, 1, false);
    }
    if (d->f.mode == 0 && d->dbias_table && d->table_phase != 1) {
      // workgroup block of this kernel: NWB waves x 16 queries against KCB keys (two 16 x KCB register blocks per wave: the running dS
      // sums and the bias + mask block)
      constexpr int NWB = 4, KCB = 128;
      const int nkc = (d->f.L + KCB - 1) / KCB;
      const int nqbb = (nt_ + NWB - 1) / NWB;
      // window positions (the bias + mask block is per position); without a region map every sequence shares one block
      int nwin = d->f.region ? ((d->f.n_win > 0 && d->f.nseq % d->f.n_win == 0) ? d->f.n_win : d->f.nseq) : 1;
      int ncb = (8192 / NWB) / (d->f.heads * nqbb * nkc);
      if (ncb < 1) ncb = 1;
      if (!d->f.region && ncb > 1) {                  // no mask: split the sequences themselves to fill the chip
        nwin = ncb < d->f.nseq ? ncb : d->f.nseq;
        while (d->f.nseq % nwin) --nwin;
      }
      if (ncb > nwin) ncb = nwin;
      const SmemS sd = smem_stream(d->f.L, 32, 0, d->f.table_len, 0, KCB);
      const int sm_db = sd.total + 2 * KCB * 32 * 2;                    // second (K chunk, V chunk) buffer of the pipelined clip loop
      int rc_ = set_smem(attn_bwd_dbias_stream_kernel<NWB, KCB>, sm_db);
      if (rc_) return rc_;
      hipLaunchKernelGGL((attn_bwd_dbias_stream_kernel<NWB, KCB>), dim3(ncb * d->f.heads * nqbb * nkc), dim3(NWB * 64), sm_db, st, *d, ncb, nqbb, nwin);
      VMVM_CHECK_LAUNCH();
    }
    return VMVM_OK;
  }
  if (d->table_phase == 2) return VMVM_OK;           // no separate table launch for this problem: phase 1 did everything
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
    const int nwin = d->f.n_win > 0 ? d->f.n_win : 1;
    // batch-persistent kernel: exact tile counts of the step's windows (392 / 196 tokens) and everything fits the 160 KB LDS
    const int nq = s_.nt == 25 ? 4 : 7, ns = s_.nt == 25 ? 2 : 1;
    const int lp32 = win2_rows(s_.nt, ns);
    const int tl4 = (d->f.table_len + 3) & ~3, lr4 = (d->f.L + 3) & ~3;
    const int smem2 = 4 * lp32 * 64 + 2 * tl4 * 4 + 2 * (ns - 1) * nq * 64 * 8 * 4 + lr4 * 4 + ((lr4 + 15) & ~15);
    const bool pers_ok = (d->f.nseq % nwin == 0) && (s_.nt == 25 || s_.nt == 13) && smem2 <= 160 * 1024;
    const bool w3 = vmvm_w3::applicable(&d->f) && !w3_off(0);
    if (w3 && !w3_off(2)) {
      int rc_ = vmvm_w3::launch_dq(d, st);
      if (rc_) return rc_;
    } else if (pers_ok) {
      // nt = 25 (392-token window): 4 query tiles x 2 key splits (8 waves, ~220 VGPRs, no spill); nt = 13: 7 tiles x 1 split.
      // Measured alternatives on MI355X (stage-3 shape, B=32): (5,2)/(6,2) spill, (4,3) is 15% slower.
      const int nqg2 = (s_.nt + nq - 1) / nq;
      // chunks of the (window position x clip) sequence range per (head, query group): whole rounds of the 256 CUs (one
      // workgroup per CU: the LDS images fill it) x sequences per round, plus ~10 sequence-times of per-workgroup set-up
      // (table staging, bias block, table-gradient flush through LDS atomics)
      const int base = d->f.heads * nqg2;
      int nbc = 1; float best = 1e30f;
      for (int c = 1; c <= 64 && c <= d->f.nseq; ++c) {
        const float cost = (float)((base * c + 255) / 256) * ((float)((d->f.nseq + c - 1) / c) + 10.f);
        if (cost < best - 1e-6f) { best = cost; nbc = c; }
      }
      const int grid2 = base * nbc;
#define LAUNCH_DQW2(NX, MASK, NQ, NS)                                                                               \
      do {                                                                                                          \
        int rc_ = set_smem(attn_bwd_dq_win2_kernel<NX, MASK, NQ, NS>, smem2);                                        \
        if (rc_) return rc_;                                                                                        \
        hipLaunchKernelGGL((attn_bwd_dq_win2_kernel<NX, MASK, NQ, NS>), dim3(grid2), dim3(NQ * NS * 64), smem2, st, *d, nqg2, nbc); \
      } while (0)
#define LAUNCH_DQW2_M(NX, NQ, NS) do { if (mask) LAUNCH_DQW2(NX, true, NQ, NS); else LAUNCH_DQW2(NX, false, NQ, NS); } while (0)
      if (s_.nt == 25) LAUNCH_DQW2_M(25, 4, 2); else LAUNCH_DQW2_M(13, 7, 1);
    }
    else if (s_.nt == 25) { if (mask) LAUNCH_DQW(26, 25, true); else LAUNCH_DQW(26, 25, false); }
    else if (s_.nt == 13) { if (mask) LAUNCH_DQW(14, 13, true); else LAUNCH_DQW(14, 13, false); }
    else if (s_.nt <= 16) LAUNCH_DQW(16, 0, true);
    else LAUNCH_DQW(28, 0, true);
    VMVM_CHECK_LAUNCH();
    {
      // dK / dV: batch-persistent kernel for the step's exact window shapes, generic per-(sequence, head) kernel otherwise
      const int npq = (s_.nt + 1) / 2, img = npq * 32 * 64;
      const int smem3 = 2 * (2 * img + 2 * 512 * 4) + tl4 * 4 + lr4 * 4 + ((lr4 + 15) & ~15);
      const bool dkv_ok = (d->f.nseq % nwin == 0) && (s_.nt == 25 || s_.nt == 13) && d->f.L <= 512 && smem3 <= 160 * 1024;
      // key-blocked dK / dV for the UN-shifted blocks only (-14 % there; its masked build -- 9 walk variants at 128 registers -- spills
      // and measured 5-9 % slower than the win3 kernel: profiles/r05_window_attention_win4_ab.txt); VMVM_WIN4_DKV_MASKED=1 forces it
      static const bool dkv_masked = getenv("VMVM_WIN4_DKV_MASKED") != nullptr;
      if (vmvm_w3::applicable(&d->f) && !w3_off(0) && !w4_off(2) && (!d->f.region || dkv_masked)) {
        int rc_ = vmvm_w4::launch_dkv(d, st);
        if (rc_) return rc_;
      } else if (vmvm_w3::applicable(&d->f) && !w3_off(0)) {
        int rc_ = vmvm_w3::launch_dkv(d, st);
        if (rc_) return rc_;
      } else if (dkv_ok) {
        // key tiles per wave: 2 for the 196-token window (K / V register fragments serve twice the MFMAs per LDS read); 1 for
        // the 392-token window, where two tiles' bias blocks (100 VGPRs) spill -- measured 12% slower than one tile per wave
        const int kt_ = s_.nt == 25 ? 1 : 2, nw_ = 7;
        const int nkg = (s_.nt + kt_ * nw_ - 1) / (kt_ * nw_);
        const int base = d->f.heads * nkg;
        int nch = 1; float best = 1e30f;
        for (int c = 1; c <= 64 && c <= d->f.nseq; ++c) {
          const float cost = (float)((base * c + 255) / 256) * ((float)((d->f.nseq + c - 1) / c) + 3.f);
          if (cost < best - 1e-6f) { best = cost; nch = c; }
        }
#define LAUNCH_DKV2_(NX, MASK, KT, NWV)                                                                              \
        do {                                                                                                        \
          int rc_ = set_smem(attn_bwd_dkv_win2_kernel<NX, MASK, KT, NWV>, smem3);                                    \
          if (rc_) return rc_;                                                                                      \
          hipLaunchKernelGGL((attn_bwd_dkv_win2_kernel<NX, MASK, KT, NWV>), dim3(base * nch), dim3(NWV * 64), smem3, st, *d, nkg, nch); \
        } while (0)
        if (s_.nt == 25) { if (mask) LAUNCH_DKV2_(25, true, 1, 7); else LAUNCH_DKV2_(25, false, 1, 7); }
        else { if (mask) LAUNCH_DKV2_(13, true, 2, 7); else LAUNCH_DKV2_(13, false, 2, 7); }
        VMVM_CHECK_LAUNCH();
      } else {
        LAUNCH_BWD(attn_bwd_dkv_kernel, 32, 0, 4, 2, 0);
      }
    }
  } else {
    const Smem sb_ = smem_layout(d->f.L, 64, 1, d->f.table_len, 1);
    if (d->f.causal_from > 0) {                                     // seq2seq mask (smtm pass)
      const Smem s1_ = smem_layout(d->f.L, 64, 1, d->f.table_len, 1), s2_ = smem_layout(d->f.L, 64, 1, d->f.table_len, 2);
      int rc_ = set_smem(attn_bwd_dq_kernel<64, 1, 8, 0, true>, s1_.total);
      if (rc_) return rc_;
      hipLaunchKernelGGL((attn_bwd_dq_kernel<64, 1, 8, 0, true>), dim3(nchunks * d->f.heads), dim3(8 * 64), s1_.total, st, *d, nchunks);
      VMVM_CHECK_LAUNCH();
      rc_ = set_smem(attn_bwd_dkv_kernel<64, 1, 8, 0, true>, s2_.total);
      if (rc_) return rc_;
      hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, 1, 8, 0, true>), dim3(nb), dim3(8 * 64), s2_.total, st, *d);
      VMVM_CHECK_LAUNCH();
    } else if (sb_.nt == 27) {        // L = 432 (fusion encoder): exact tile count, fully unrolled tile loops
      // Measured and not kept (round 4, profiles/r04_ab_fusion_attention_variants.txt): two key tiles per wave in dK/dV (KTP = 2), two query
      // tiles per wave in dQ (QT = 2) -- half the LDS bytes per score tile, same time: these kernels are not LDS-bound --, and 9 waves
      // (3 full passes over the 27 owner tiles instead of 3 + a 3-wave pass): slower, the 8-wave form already loads the SIMDs 7/7/7/6.
      // (round 5: a key-blocked, persistent dQ kernel -- tools/scratch/attention_fus4_experiment.hip -- is 17 % faster alone and changes
      //  nothing in the step: DESIGN 8 round 5)
      LAUNCH_BWD_DQ(64, 1, 8, 27);
      LAUNCH_BWD(attn_bwd_dkv_kernel, 64, 1, 8, 2, 27);
    } else {
      LAUNCH_BWD_DQ(64, 1, 8, 0);
      LAUNCH_BWD(attn_bwd_dkv_kernel, 64, 1, 8, 2, 0);
    }
  }
  return VMVM_OK;
}

extern "C" int64_t vmvm_attention_drop_mask_size(const vmvm_attn_fwd_desc* d) {
  if (!d || !drop_mask_ok(d)) return 0;
  return (int64_t)d->nseq * d->heads * 27 * 27 * 8 * 4;
}

// scratch of the reproducible table gradient (vmvm_attn_bwd_desc.dbias_ws): one partial table per workgroup of the win_layout = 1 dQ kernel
extern "C" int64_t vmvm_attention_bwd_dbias_ws_size(const vmvm_attn_bwd_desc* d) {
  if (!d || d->f.nseq <= 0 || d->f.heads <= 0 || d->f.L <= 0) return VMVM_EINVAL;
  if (d->f.mode != 0 || !d->f.win_layout || !vmvm_w3::applicable(&d->f) || w3_off(0)) return 0;
  return vmvm_w3::dbias_ws_size(d);
}

// `delta` scratch of the backward (f32 [nseq][heads][L])
extern "C" int64_t vmvm_attention_bwd_workspace_size(const vmvm_attn_bwd_desc* d) {
  if (!d || d->f.nseq <= 0 || d->f.heads <= 0 || d->f.L <= 0) return VMVM_EINVAL;
  return (int64_t)d->f.nseq * d->f.heads * d->f.L * (int64_t)sizeof(float);
}

