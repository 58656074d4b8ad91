// attention_win3.hip -- Video-Swin window attention (WindowAttention3D.forward video_swin.py:147-172 with the shift mask of
// compute_mask :292-307) for windows of (8,7,7) = 392 tokens, head_dim 32, tokens laid out in the win_layout = 1 order
// (include/vmvm.h, attn_win3.h, swin_index.win3_perm): backward kernels.
//
// What the layout buys over the win2 kernels of attention.hip (measured there, profiles/r04_*: at two waves per SIMD the LDS, VALU and
// matrix pipes of a wave's tile loop add up instead of overlapping, and neither the next sequence's DMA nor the exponentials nor the
// order of the MFMAs is what the time goes to -- so the lever is fewer LDS and VALU instructions per score element):
//  * the relative-position bias of a (query tile, key tile) pair is ONE 16-byte LDS read per lane from a windowed copy of the head's
//    table column (four 8 x 8 Toeplitz blocks per tile pair; row = lane constant + per-tile immediate): no register-resident bias
//    block per wave (26-50 VGPRs), no unpack instructions, no rebuild per window position, f32 bias instead of bf16;
//  * the registers that frees hold a second key tile per wave (dK/dV) -- every Q / dO / lse / delta fragment read from LDS serves two
//    score tiles;
//  * the shift mask never appears: region-major tiles are fully live or fully masked per window type, and the tile loops exist once per
//    live-class set with the masked tiles absent at compile time (49 % of the tile pairs of an edge window, 74 % of a corner window).
#include "attn_common.h"
#include "attn_win3.h"
#include "attn_win3_dev.h"
#include <vmvm_probe_hooks.h>
#include <cstdlib>

namespace {

constexpr int W3_DTAB = 2560;                        // floats per workgroup slot of vmvm_attn_bwd_desc.dbias_ws (15 * 169 = 2 535, padded)
template <int V> struct IC { static constexpr int value = V; };

// ================================================================================================
// dK / dV.  Each wave owns TWO key tiles of one class (K / V fragments in registers) and walks the LIVE query tiles of the window
// type in pairs (a pair = one 32-deep dK / dV MFMA); Q / dO images + lse / delta of the next sequence stream into the other LDS buffer
// by DMA spread over the walk.  Lane (r, g): key r of the tile, queries 4g..4g+3 of the query tile (S = Q K^T, rows = queries).
// NWV = 8: a workgroup = (head, key group, chunk of clips), key group 0 = classes A + B (tiles 0-13), 1 = C + D (tiles 14-24).
// NWV = 12: a workgroup = (head, chunk of clips), three waves per SIMD (<= 168 registers).
//
// What bounds it (tools/probe/valu_probe.hip, profiles/r04_probe_valu_issue.txt): the softmax-side VALU chain.  One wave issues a
// packed-f32 / convert / exp instruction every 6-10 cycles whatever its SIMD partner does, two waves together reach 3.3-4 ns per
// instruction per SIMD and the VALU saturates at ~2.5 ns only with three or more; MFMAs of the same wave do NOT run under its own
// packed-f32 instructions.  So with two waves per SIMD a sequence costs the SUM of its VALU and matrix time (measured: 9 us), a
// barrier-separated ping-pong of the two waves (one multiplying, one in the chain) is no faster than letting them run free (built,
// measured, removed: the lone chain wave is issue-limited), and what helps is fewer chain instructions per score element and a third
// wave per SIMD.
// ================================================================================================
template <bool MASK, int NWV>
__global__ __launch_bounds__(NWV * 64) void attn_bwd_dkv_win3_kernel(const vmvm_attn_bwd_desc pb, const int nch) {
  constexpr int HD = 32, KT = 2;
  constexpr int NKG = NWV == 12 ? 1 : 2;                  // key groups per (head, chunk)
  constexpr int ROWS = w3::NT * 16, IMG = ROWS * HD * 2;               // 400 rows, 25 600 bytes per Q or dO image
  constexpr int LV = 512;                                               // floats reserved for lse / delta
  constexpr int BUF = 2 * IMG + 2 * LV * 4;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int L = w3::L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * NKG);
  const int kg = NKG == 2 ? (logical & 1) : 0;
  const int t1 = NKG == 2 ? (logical >> 1) : logical;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  unsigned char* TL = smem + 2 * BUF;
  // this wave's key tiles
  const int kt0 = (kg == 0 ? 0 : w3::CB[2]) + 2 * wave;
  const bool active = NKG == 1 ? true : kt0 < (kg == 0 ? w3::CB[2] : w3::NT);
  const int kc = kt0 < w3::CB[1] ? 0 : kt0 < w3::CB[2] ? 1 : kt0 < w3::CB[3] ? 2 : 3;      // class of both tiles
  int key[KT]; bool kv[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) { key[t] = (kt0 + t) * 16 + r; kv[t] = active && (kt0 + t) < w3::NT && key[t] < L; }

  // windowed table of this head (staged through the second buffer, which the first sequence does not use)
  {
    float* stage = reinterpret_cast<float*>(smem + BUF);
    for (int i = tid; i < 15 * 169; i += NWV * 64) stage[i] = p.bias_table[(size_t)i * heads + h];
    __syncthreads();
    w3_build_table<1>(TL, stage, tid, NWV * 64);
    __syncthreads();
  }
  // lane bases into the table: row = A(q) - A(k) + 84, window s = 4 (g & 1) - (r & 7) + 7; A(q) = tile immediate + lq * step
  const int lk = r >> 3, lq = g >> 1, sw = 4 * (g & 1) - (r & 7) + 7;
  const unsigned char* tb1[KT]; const unsigned char* tb13[KT]; const unsigned char* tb24[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int ktc = (kt0 + t) < w3::NT ? kt0 + t : w3::NT - 1;
    const int ak = w3_posA_rt(2 * ktc + lk);
    const unsigned char* b = TL + (84 - ak) * w3::ROWB + sw * 16;
    tb1[t] = b + lq * w3::ROWB;
    tb13[t] = b + lq * 13 * w3::ROWB;
    tb24[t] = lq ? TL + 169 * w3::ROWB + sw * 16 : b + w3::tileA0(w3::NT - 1) * w3::ROWB;      // query tile 24: its second position is padding
  }

  // this workgroup's sequences: the clips [c0, c1) of EVERY window position, window-major (a workgroup sees every window type, so the
  // masked tiles the edge / corner windows skip shorten every workgroup alike)
  const int cper = (B + nch - 1) / nch;
  const int c0 = ch * cper, c1 = (c0 + cper < B) ? c0 + cper : B;
  const int ncl = c1 > c0 ? c1 - c0 : 0, total = ncl * nWin;
  int w_nx = 0, c_nx = c0;                               // (window position, clip) of the NEXT sequence to request
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == c1) { c_nx = c0; ++w_nx; } };
  uint32_t off_k[KT], off_dk[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) { off_k[t] = (uint32_t)key[t] * p.ld_qkv + h * HD + g * 8; off_dk[t] = (uint32_t)key[t] * pb.ld_dqkv + h * HD + g * 4; }
  constexpr int NF = (ROWS * 4 + NWV * 64 - 1) / (NWV * 64);           // 16-byte DMA requests per thread per image
  uint32_t goq[NF], god[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int u = i * NWV * 64 + tid, row = u >> 2, chs = u & 3, cs = (chs ^ swz_chunk<32>(row)) << 3;
    goq[i] = (u < ROWS * 4) ? (uint32_t)((row * p.ld_qkv + cs) * 2) : 0xffffffffu;
    god[i] = (u < ROWS * 4) ? (uint32_t)((row * pb.ld_dout + cs) * 2) : 0xffffffffu;
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
  // K / V fragments.  The next sequence's are loaded by inline asm into the SAME registers once the walk has multiplied with them for
  // the last time (no second set), and waited for by the counted s_waitcnt at the loop top: as compiler-visible loads their consumer
  // at the loop header made the compiler wait with vmcnt(0) -- i.e. for the previous sequence's dK / dV stores as well (it cannot count
  // memory instructions across the back edge).  Lanes of padding keys never load and keep their zeros.
  f32x4 kf[KT], vf[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) { kf[t] = f32x4{0.f, 0.f, 0.f, 0.f}; vf[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  float ss_n = 1.0f;
  auto fetch = [&](size_t seq) {
    const u16* qb = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kv[t]) {
        const u16* pk = qb + p.k_off + off_k[t];
        const u16* pv = qb + p.v_off + off_k[t];
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(kf[t]) : "v"(pk) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(vf[t]) : "v"(pv) : "memory");
      }
    }
    if (p.seq_scale) {                                    // scalar load by hand: as a vector load its consumer is one more vmcnt(0) at the loop header
      const float* sp = p.seq_scale + seq / p.seqs_per_scale;
      asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ss_n) : "s"(sp) : "memory");
    }
  };
  if (total > 0) {
#pragma unroll
    for (int i = 0; i <= NF; ++i) dma_step(seq_nx(), 0, i);
    fetch(seq_nx());
    advance();
  }
  // memory instructions a wave issues AFTER its last DMA request of a sequence: the next fragment loads (when the walk is long enough
  // to hold them) and the dK / dV stores -- what the counted wait at the loop top may leave in flight
  const bool full = active && kt0 + 1 < w3::NT - 1;     // both tiles hold 16 real keys: every load / store below is issued by the wave

  int wprev = -1, w_cu = 0, c_cu = c0, m4 = 15;
  for (int b = 0; b < total; ++b) {
    const int cur = b & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    // This sequence's images and fragments must have landed -- but NOT the dK / dV stores of the previous sequence, the youngest 8
    // memory instructions of a wave that owns two full key tiles (vmcnt counts loads and stores in order).  Raw barrier: LDS is
    // read-only inside a sequence, the barrier only orders "my DMA landed" / "everyone left the other buffer".
    // (ONE operand-carrying wait: with one asm per branch the compiler copied the fragment registers in front of one of them)
    if (!(b > 0 && full)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(kf[0]), "+v"(vf[0]), "+v"(kf[1]), "+v"(vf[1]) : : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int wcur = w_cu;
    if (++c_cu == c1) { c_cu = c0; ++w_cu; }
    if (MASK && wcur != wprev) {                          // (workgroup-uniform) new window position: which query classes this wave's keys see
      wprev = wcur;
      m4 = w3_live_rt(kc, __builtin_amdgcn_readfirstlane(w3_window_type(p.region, wcur)));
    }
    const float seq_scale = ss_n;
    const bool has_next = b + 1 < total;
    const size_t sq_nx = seq_nx();
    if (has_next) advance();
    if (!active) {                                        // idle wave: its share of the DMA requests
      if (has_next) {
#pragma unroll
        for (int i = 0; i <= NF; ++i) dma_step(sq_nx, cur ^ 1, i);
      }
      continue;
    }
    const unsigned char* Qs = smem + cur * BUF;
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
    const unsigned char* qb_ = Qs + k_off_swz<HD>(r, g);
    const unsigned char* tq0 = Qs + k_off_swz<HD>(g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8;
    const unsigned char* tq1 = Qs + k_off_swz<HD>(g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8;
    const uint32_t tq0a = lds_addr(tq0), tq1a = lds_addr(tq1);

    vmvm_hook::W3Timeline tl(b, lane, wave);                // (probe builds: cycle stamps of one wave; nothing here)
    // the walk over the live query tiles of class set M4 (compile-time list)
    auto walk = [&](auto m4c) {
      constexpr int M4 = decltype(m4c)::value;
      constexpr w3::TileList QL = w3::list_all(M4);
      constexpr int NP = (QL.n + 1) / 2;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      bf16x8 qf[2], dof[2];                               // fragments of the next pair
      f32x4 l4[2], d4[2], bias[2][KT];
      f32x4 s4[2][KT], dp4[2][KT];                        // scores / dP of the current pair
      f32x2 nl[2][2], nd[2][2];
      uint32_t pw[KT][4], dw[KT][4];
      s16x4 a0, a1, c0_, c1_, e0, e1, f0, f1;
      auto reads = [&](const int c) {                     // LDS reads for pair c
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < QL.n) {
            const int qt = QL.t[2 * c + u];
            qf[u] = *reinterpret_cast<const bf16x8*>(qb_ + qt * 1024);
            dof[u] = *reinterpret_cast<const bf16x8*>(qb_ + IMG + qt * 1024);
            lds_read2_b128(l4[u], d4[u], lse_a, delta_a, qt * 64);
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              const unsigned char* bp = qt == w3::NT - 1 ? tb24[t] : (w3::tileStep(qt) == 13 ? tb13[t] : tb1[t]) + w3::tileA0(qt) * w3::ROWB;
              bias[u][t] = *reinterpret_cast<const f32x4*>(bp);
            }
          }
        }
      };
      auto mma1 = [&](const int c) {                      // S = Q K^T + bias, dP = dO V^T of pair c; then -lse log2 e, -delta
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < QL.n) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              s4[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[u], __builtin_bit_cast(bf16x8, kf[t]), bias[u][t], 0, 0, 0);
              dp4[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[u], __builtin_bit_cast(bf16x8, vf[t]), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < QL.n) {
            lds_wait2(l4[u], d4[u]);
            nl[u][0] = f32x2{-l4[u][0] * LOG2E, -l4[u][1] * LOG2E}; nl[u][1] = f32x2{-l4[u][2] * LOG2E, -l4[u][3] * LOG2E};
            nd[u][0] = f32x2{-d4[u][0], -d4[u][1]}; nd[u][1] = f32x2{-d4[u][2], -d4[u][3]};
          }
        }
      };
      auto chain = [&](const int c) {                     // P = exp2(S log2 e - lse), dS = P (dP scale - delta), both to bf16 operands
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int x = 0; x < 4; ++x) { pw[t][x] = 0u; dw[t][x] = 0u; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < QL.n) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
#pragma unroll
              for (int hj = 0; hj < 2; ++hj) {
                const f32x2 sv = {s4[u][t][2 * hj], s4[u][t][2 * hj + 1]}, dpv = {dp4[u][t][2 * hj], dp4[u][t][2 * hj + 1]};
                const f32x2 e = __builtin_elementwise_fma(sv, f32x2{LOG2E, LOG2E}, nl[u][hj]);
                const f32x2 pr = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                const f32x2 d = pr * __builtin_elementwise_fma(dpv, ss2, nd[u][hj]);
                pw[t][2 * u + hj] = pack_bf2v(pr);
                dw[t][2 * u + hj] = pack_bf2v(d);
              }
            }
          }
        }
      };
      auto trreads = [&](const int c) {                   // transposed Q / dO operands of pair c
        const int qa = QL.t[2 * c], qb2 = (2 * c + 1 < QL.n) ? QL.t[2 * c + 1] : QL.t[2 * c];
        tr_read4_2(a0, a1, c0_, c1_, tq0a, tq1a, qa * 1024, qb2 * 1024);
        tr_read4_2(e0, e1, f0, f1, tq0a, tq1a, IMG + qa * 1024, IMG + qb2 * 1024);
      };
      auto mma2 = [&]() {                                 // dV += dO^T P, dK += Q^T dS over the pair's 32 queries
        tr_wait4(a0, a1, c0_, c1_);
        tr_wait4(e0, e1, f0, f1);
        const s16x8 q0v = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const s16x8 q1v = {c0_[0], c0_[1], c0_[2], c0_[3], c1_[0], c1_[1], c1_[2], c1_[3]};
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
      };
      tl.stamp(1);
      reads(0);
      mma1(0);
      tl.stamp(2);
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        trreads(c);
        if (c <= NF && has_next) dma_step(sq_nx, cur ^ 1, c);
        tl.stamp(10);
        chain(c);
        __builtin_amdgcn_sched_barrier(0);                // (the next pair's fragments once the chain has released its registers: asking
        tl.stamp(11);                                         //  for them before the chain costs 40 registers and measured the same -- the
        if (c + 1 < NP) reads(c + 1);                     //  LDS pipe is a co-bound of this loop, the requests queue either way)
        tl.stamp(12);
        mma2();
        tl.stamp(13);
        if (c + 1 < NP) mma1(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        tl.stamp(14);
      }
      if (has_next) {                                     // (short walks: the DMA steps the pairs did not cover)
#pragma unroll
        for (int c = NP; c <= NF; ++c) dma_step(sq_nx, cur ^ 1, c);
      }
    };
    if constexpr (!MASK) {
      walk(IC<15>{});
    } else {
      switch (m4) {
        case 15: walk(IC<15>{}); break;
        case 3: walk(IC<3>{}); break;
        case 12: walk(IC<12>{}); break;
        case 5: walk(IC<5>{}); break;
        case 10: walk(IC<10>{}); break;
        case 1: walk(IC<1>{}); break;
        case 2: walk(IC<2>{}); break;
        case 4: walk(IC<4>{}); break;
        default: walk(IC<8>{}); break;
      }
    }
    tl.stamp(20);
    tl.flush(0);
    const float ssv = seq_scale;
    if (has_next) fetch(sq_nx);                           // the fragment registers are free: the next sequence's K / V (4 loads, before the 8 stores)
    u16* dbase = reinterpret_cast<u16*>(pb.dqkv) + seq * L * pb.ld_dqkv;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kv[t]) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          *reinterpret_cast<uint2*>(dbase + off_dk[t] + p.k_off + dt * 16) = make_uint2(pack_bf2(dk[t][dt][0], dk[t][dt][1]), pack_bf2(dk[t][dt][2], dk[t][dt][3]));
          *reinterpret_cast<uint2*>(dbase + off_dk[t] + p.v_off + dt * 16) =
              make_uint2(pack_bf2(dv[t][dt][0] * ssv, dv[t][dt][1] * ssv), pack_bf2(dv[t][dt][2] * ssv, dv[t][dt][3] * ssv));
        }
      }
    }
  }
}

// ================================================================================================
// dQ (+ the bias-table gradient).  A workgroup = (head, group of 4 query tiles, chunk of clips); wave (ql, kh) owns query tile ql of the
// group against one HALF of the key tiles -- the halves interleave by class (w3::list_part: 13 / 12 tiles), so whatever classes a
// window type leaves live are split evenly between the two waves of a query tile; the two dQ partials are combined through LDS at the
// next loop-top barrier.  K / V of the next sequence stream into the other LDS buffer; the next sequence's Q / dO / O fragments are
// prefetched into registers.  Lane (r, g): query r of the tile, keys 4g..4g+3 of the key tile (S^T = K Q^T, rows = keys).
// The running sums of dS (bias-table gradient) stay in registers over the workgroup's whole run, one slot per key tile of the half, and
// are scattered once at the end: entry (dq - dk + 7) * 169 + A(q) - A(k) + 84 of the head's column.
// ================================================================================================
template <bool MASK, int NS>
__global__ __launch_bounds__(4 * NS * 64) void attn_bwd_dq_win3_kernel(const vmvm_attn_bwd_desc pb, const int nqg, const int nch) {
  constexpr int HD = 32, NQ = 4, NWV = NQ * NS;
  constexpr int NH = w3::max_part(NS);                                  // running-sum slots of a wave: tiles of the longest key part (13 of 2, 9 of 3)
  constexpr bool XF32 = NS == 2;                                        // dQ partials cross the workgroup as f32 (two parts) or packed bf16 (three: LDS)
  constexpr int ROWS = w3::NT * 16, KV = ROWS * HD * 2;                 // 400 rows, 25 600 bytes per K or V image
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int L = w3::L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nqg);
  const int qg = logical % nqg;
  const int t1 = logical / nqg;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  constexpr int XCH = (NS - 1) * NQ * 64 * 8;                           // values per parity
  constexpr int XB = XF32 ? 4 : 2;                                      // bytes per value
  unsigned char* xch = smem + 4 * KV;                                   // [2][NS-1][NQ][64 lanes][8]: dQ partials of parts 1.., double-buffered by sequence parity
  unsigned char* TL = smem + 4 * KV + 2 * XCH * XB;
  const bool want_dtab = pb.dbias_table != nullptr;
  const int ql = wave % NQ, kh = wave / NQ;                             // query tile slot, key half
  // query tiles of this group: the 25 tiles are dealt evenly over the nqg groups (nqg = 7: 4,4,4,4,4,4,1 as in win2; nqg = 8: runs of
  // 3 / 4 tiles -- heads x 8 base workgroups divide the 256 CUs for every stage's head count, and a CU carries 6 waves instead of 8)
  const int qt0 = nqg == 7 ? qg * NQ : (w3::NT * qg) / nqg;
  const int qcnt = nqg == 7 ? (w3::NT - qt0 < NQ ? w3::NT - qt0 : NQ) : (w3::NT * (qg + 1)) / nqg - qt0;
  const int qt = qt0 + ql;
  const int q = qt * 16 + r;
  const bool active = ql < qcnt;
  const bool qv = active && (q < L);
  const int qc = qt < w3::CB[1] ? 0 : qt < w3::CB[2] ? 1 : qt < w3::CB[3] ? 2 : 3;

  {                                                       // windowed table of this head (staged through the second K / V buffer)
    float* stage = reinterpret_cast<float*>(smem + 2 * KV);
    for (int i = tid; i < 15 * 169; i += NWV * 64) stage[i] = p.bias_table[(size_t)i * heads + h];
    __syncthreads();
    w3_build_table<0>(TL, stage, tid, NWV * 64);
    __syncthreads();
  }
  // lane bases into the table: row = A(q) - A(k) + 84, window s = 7 - (r & 7) + 4 (g & 1); A(k) = tile immediate + lk * step
  const int lq = r >> 3, lk = g >> 1, sw = 7 - (r & 7) + 4 * (g & 1);
  const int aq = w3_posA_rt(2 * (active ? qt : 0) + lq);
  const unsigned char* tb1 = TL + (aq - lk) * w3::ROWB + sw * 16;
  const unsigned char* tb13 = TL + (aq - 13 * lk) * w3::ROWB + sw * 16;
  const unsigned char* tb24 = lk ? TL + 169 * w3::ROWB + sw * 16 : TL + (aq + 84 - w3::tileA0(w3::NT - 1)) * w3::ROWB + sw * 16;      // key tile 24: second position = padding

  f32x2 racc[NH][2];
#pragma unroll
  for (int tl = 0; tl < NH; ++tl) { racc[tl][0] = f32x2{0.f, 0.f}; racc[tl][1] = f32x2{0.f, 0.f}; }

  // this workgroup's sequences: the clips [c0, c1) of every window position, window-major
  const int cper = (B + nch - 1) / nch;
  const int c0 = ch * cper, c1 = (c0 + cper < B) ? c0 + cper : B;
  const int ncl = c1 > c0 ? c1 - c0 : 0, total = ncl * nWin;
  int w_nx = 0, c_nx = c0;
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == c1) { c_nx = c0; ++w_nx; } };
  const uint32_t off_q = (uint32_t)q * p.ld_qkv + p.q_off + h * HD + g * 8;
  const uint32_t off_do = (uint32_t)q * pb.ld_dout + h * HD + g * 8;
  const uint32_t off_o = (uint32_t)q * p.ld_out + h * HD + g * 8;
  const uint32_t off_dq = (uint32_t)q * pb.ld_dqkv + p.q_off + h * HD + g * 4;
  const uint32_t off_ls = (uint32_t)h * L + q;
  constexpr int NF = (ROWS * 4 + NWV * 64 - 1) / (NWV * 64);            // 16-byte DMA requests per thread per image
  uint32_t goff[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const int u = i * NWV * 64 + tid, row = u >> 2, chs = u & 3;
    goff[i] = (u < ROWS * 4) ? (uint32_t)((row * p.ld_qkv + ((chs ^ swz_chunk<32>(row)) << 3)) * 2) : 0xffffffffu;
  }
  const unsigned fill_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2);
  bf16x8 qf, dof, of;
  float lse_n = 0.f, ss_n = 1.0f;
  auto fetch = [&](size_t seq) {
    const u16* qb = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
    const u16* dob = reinterpret_cast<const u16*>(pb.dout) + seq * L * pb.ld_dout;
    const u16* ob = reinterpret_cast<const u16*>(p.out) + seq * L * p.ld_out;
    qf = load_frag_global(qb + off_q, qv);
    dof = load_frag_global(dob + off_do, qv);
    of = load_frag_global(ob + off_o, qv);
    lse_n = qv ? (p.lse + seq * heads * L)[off_ls] : __builtin_huge_valf();
    ss_n = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
  };
  if (total > 0) {
    const u16* qkv = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    fill_pre<NF, NWV * 64 * 16>(smem + (tid & ~63) * 16, KV, qkv + p.k_off, qkv + p.v_off, fill_bytes, goff);
    fetch(seq_nx());
    advance();
  }

  f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  auto flush_prev = [&](size_t seq, int par) {           // split 0: add the other split's partial (in LDS buffer `par`) and store dQ
    if (kh == 0 && qv) {
      float4 x0 = make_float4(dq[0][0], dq[0][1], dq[0][2], dq[0][3]), x1 = make_float4(dq[1][0], dq[1][1], dq[1][2], dq[1][3]);
#pragma unroll
      for (int s_ = 0; s_ < NS - 1; ++s_) {
        const unsigned char* x = xch + ((size_t)par * XCH + ((s_ * NQ + ql) * 64 + lane) * 8) * XB;
        float4 y0, y1;
        if (XF32) { y0 = *reinterpret_cast<const float4*>(x); y1 = *reinterpret_cast<const float4*>(x + 16); }
        else {
          const uint4 w = *reinterpret_cast<const uint4*>(x);
          y0 = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
          y1 = make_float4(__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u));
        }
        x0.x += y0.x; x0.y += y0.y; x0.z += y0.z; x0.w += y0.w;
        x1.x += y1.x; x1.y += y1.y; x1.z += y1.z; x1.w += y1.w;
      }
      u16* dqp = reinterpret_cast<u16*>(pb.dqkv) + seq * L * pb.ld_dqkv + off_dq;
      const float sc = p.scale;
      *reinterpret_cast<uint2*>(dqp) = make_uint2(pack_bf2(x0.x * sc, x0.y * sc), pack_bf2(x0.z * sc, x0.w * sc));
      *reinterpret_cast<uint2*>(dqp + 16) = make_uint2(pack_bf2(x1.x * sc, x1.y * sc), pack_bf2(x1.z * sc, x1.w * sc));
    }
  };

  int wprev = -1, w_cu = 0, c_cu = c0, m4 = 15;
  size_t seq_pv = 0;
  for (int b = 0; b < total; ++b) {
    const int cur = b & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    fill_wait();
    __syncthreads();                                      // sequence b landed; other buffer free; partials of b-1 visible
    if (b > 0) flush_prev(seq_pv, cur ^ 1);
    seq_pv = seq;
    const int wcur = w_cu;
    if (++c_cu == c1) { c_cu = c0; ++w_cu; }
    if (MASK && wcur != wprev) {
      wprev = wcur;
      m4 = w3_live_rt(qc, __builtin_amdgcn_readfirstlane(w3_window_type(p.region, wcur)));
    }
    const bf16x8 cqf = qf, cdof = dof, cof = of;
    const float clse2 = lse_n * LOG2E, seq_scale = ss_n;
    const bool has_next = b + 1 < total;
    const u16* kv_nx = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    unsigned char* dst_nx = smem + (cur ^ 1) * 2 * KV + (tid & ~63) * 16;
    if (has_next) {
      fetch(seq_nx());
      advance();
      if (!active) fill_pre<NF, NWV * 64 * 16>(dst_nx, KV, kv_nx + p.k_off, kv_nx + p.v_off, fill_bytes, goff);
    }
    if (!active) continue;
    const unsigned char* Ksm = smem + cur * 2 * KV;
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)cdof[e] * (float)cof[e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (kh == 0 && g == 0 && qv) (pb.delta + seq * heads * L)[off_ls] = dl;
    dq[0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 nl2 = {-clse2, -clse2}, ss2 = {seq_scale, seq_scale}, ndl2 = {-dl, -dl};
    asm volatile("" : "+v"(nl2), "+v"(ss2), "+v"(ndl2));   // real register pairs (see attn_bwd_dq_win2_kernel)
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const unsigned char* kb = Ksm + k_off_swz<HD>(r, g);                 // tile 0; every tile is a compile-time immediate away
    const unsigned char* tk0 = Ksm + k_off_swz<HD>(g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8;
    const unsigned char* tk1 = Ksm + k_off_swz<HD>(g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8;
    const uint32_t tk0a = lds_addr(tk0), tk1a = lds_addr(tk1);

    auto walk = [&](auto khc, auto m4c) {
      constexpr int KH = decltype(khc)::value, M4 = decltype(m4c)::value;
      constexpr w3::TileList KL = w3::list_part(M4, KH, NS);
      constexpr int NP = (KL.n + 1) / 2;
      bf16x8 kf[2], vf[2];
      f32x4 bias[2];
      auto reads = [&](const int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < KL.n) {
            const int kt = KL.t[2 * c + u];
            kf[u] = *reinterpret_cast<const bf16x8*>(kb + kt * 1024);
            vf[u] = *reinterpret_cast<const bf16x8*>(kb + KV + kt * 1024);
            const unsigned char* bp = kt == w3::NT - 1 ? tb24 : (w3::tileStep(kt) == 13 ? tb13 : tb1) + (84 - w3::tileA0(kt)) * w3::ROWB;
            bias[u] = *reinterpret_cast<const f32x4*>(bp);
          }
        }
      };
      constexpr bool AHEAD = NS == 2;                       // the next pair's fragments are requested before the chain (three parts: no registers for that -- the third wave of the SIMD covers the wait)
      if (AHEAD && NP > 0) reads(0);
#pragma unroll
      for (int c = 0; c < NP; ++c) {
        f32x4 s4[2], dp4[2];
        if (!AHEAD) reads(c);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < KL.n) {
            s4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[u], cqf, bias[u], 0, 0, 0);
            dp4[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[u], cdof, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
        }
        const int ta = KL.t[2 * c], tb = (2 * c + 1 < KL.n) ? KL.t[2 * c + 1] : KL.t[2 * c];
        s16x4 a0, a1, c0_, c1_;
        tr_read4_2(a0, a1, c0_, c1_, tk0a, tk1a, ta * 1024, tb * 1024);
        if (c < NF && has_next) fill_one(dst_nx + c * NWV * 64 * 16, KV, kv_nx + p.k_off, kv_nx + p.v_off, fill_bytes, goff[c], c == NF - 1);
        if (AHEAD && c + 1 < NP) reads(c + 1);
        uint32_t dsw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < KL.n) {
            constexpr int dummy = 0; (void)dummy;
            const int slot = w3::slot_in_part(KL.t[2 * c + u], KH, NS);
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
              const f32x2 sv = {s4[u][2 * hj], s4[u][2 * hj + 1]}, dpv = {dp4[u][2 * hj], dp4[u][2 * hj + 1]};
              const f32x2 e = __builtin_elementwise_fma(sv, f32x2{LOG2E, LOG2E}, nl2);
              const f32x2 pr = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
              const f32x2 d = pr * __builtin_elementwise_fma(dpv, ss2, ndl2);
              asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(racc[slot][hj]) : "v"(d));
              dsw[2 * u + hj] = pack_bf2v(d);
            }
          }
        }
        const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(dsw[0], dsw[1], dsw[2], dsw[3]));
        tr_wait4(a0, a1, c0_, c1_);
        const s16x8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const s16x8 v1 = {c0_[0], c0_[1], c0_[2], c0_[3], c1_[0], c1_[1], c1_[2], c1_[3]};
        dq[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v0), dsf, dq[0], 0, 0, 0);
        dq[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v1), dsf, dq[1], 0, 0, 0);
      }
      if (has_next) {                                     // (short walks: the DMA requests the pairs did not cover)
#pragma unroll
        for (int c = NP; c < NF; ++c) fill_one(dst_nx + c * NWV * 64 * 16, KV, kv_nx + p.k_off, kv_nx + p.v_off, fill_bytes, goff[c], c == NF - 1);
      }
    };
    auto walk_m = [&](auto khc) {
      if constexpr (!MASK) {
        walk(khc, IC<15>{});
      } else {
        switch (m4) {
          case 15: walk(khc, IC<15>{}); break;
          case 3: walk(khc, IC<3>{}); break;
          case 12: walk(khc, IC<12>{}); break;
          case 5: walk(khc, IC<5>{}); break;
          case 10: walk(khc, IC<10>{}); break;
          case 1: walk(khc, IC<1>{}); break;
          case 2: walk(khc, IC<2>{}); break;
          case 4: walk(khc, IC<4>{}); break;
          default: walk(khc, IC<8>{}); break;
        }
      }
    };
    if (kh == 0) walk_m(IC<0>{}); else if (NS == 2 || kh == 1) walk_m(IC<1>{}); else walk_m(IC<NS - 1>{});
    if (kh > 0) {                                         // parts 1.. publish their partials; part 0 adds them after the next barrier
      unsigned char* x = xch + ((size_t)cur * XCH + (((kh - 1) * NQ + ql) * 64 + lane) * 8) * XB;
      if (XF32) {
        *reinterpret_cast<float4*>(x) = make_float4(dq[0][0], dq[0][1], dq[0][2], dq[0][3]);
        *reinterpret_cast<float4*>(x + 16) = make_float4(dq[1][0], dq[1][1], dq[1][2], dq[1][3]);
      } else {
        *reinterpret_cast<uint4*>(x) = make_uint4(pack_bf2(dq[0][0], dq[0][1]), pack_bf2(dq[0][2], dq[0][3]), pack_bf2(dq[1][0], dq[1][1]), pack_bf2(dq[1][2], dq[1][3]));
      }
    }
  }
  __syncthreads();
  if (total > 0 && active) flush_prev(seq_pv, (total - 1) & 1);
  if (want_dtab) {
    float* dtab = reinterpret_cast<float*>(smem);         // (the K / V buffers are free now)
    __syncthreads();
    for (int i = tid; i < 15 * 169; i += NWV * 64) dtab[i] = 0.f;
    __syncthreads();
    // Reproducible form (pb.dbias_ws): the waves scatter ONE AFTER THE OTHER (NWV barriers, once per workgroup: the order of the f32
    // adds into an LDS entry is then fixed -- inside one ds_add instruction colliding lanes are served in lane order) and the table goes
    // to this workgroup's slot of the scratch instead of global atomics; w3_dbias_reduce_kernel sums a head's slots in order.
    const bool det = pb.dbias_ws != nullptr;
    for (int turn = 0; turn < (det ? NWV : 1); ++turn) {
    if (qv && (!det || wave == turn)) {
      auto scatter = [&](auto khc) {
        constexpr int KH = decltype(khc)::value;
        constexpr w3::TileList KL = w3::list_part(15, KH, NS);
#pragma unroll
        for (int i = 0; i < KL.n; ++i) {
          const int kt = KL.t[i];
          if (!(kt == w3::NT - 1 && lk == 1)) {
            const int rho = aq - (w3::tileA0(kt) + lk * w3::tileStep(kt)) + 84;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int dlt = (r & 7) - (4 * (g & 1) + j) + 7;
              atomicAdd(&dtab[dlt * 169 + rho], racc[i][j >> 1][j & 1]);
            }
          }
        }
      };
      if (kh == 0) scatter(IC<0>{}); else if (NS == 2 || kh == 1) scatter(IC<1>{}); else scatter(IC<NS - 1>{});
    }
    __syncthreads();
    }
    if (det) {
      float* slot = reinterpret_cast<float*>(pb.dbias_ws) + (size_t)logical * W3_DTAB;          // logical = h * (nch * nqg) + (ch * nqg + qg)
      for (int i = tid; i < 15 * 169; i += NWV * 64) slot[i] = dtab[i];
    } else {
      for (int i = tid; i < 15 * 169; i += NWV * 64) {
        const float v = dtab[i];
        if (v != 0.f) atomicAdd(pb.dbias_table + (size_t)i * heads + h, v);
      }
    }
  }
}

// dbias_table[i][h] += sum over the per workgroups of head h of their partial tables, in slot order (the reproducible form above)
__global__ __launch_bounds__(256) void w3_dbias_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dbias_table, const int heads, const int per) {
  const int i = blockIdx.x * 256 + threadIdx.x, h = blockIdx.y;
  if (i >= 15 * 169) return;
  const float* src = ws + (size_t)h * per * W3_DTAB + i;
  float t = 0.f;
  for (int s = 0; s < per; ++s) t += src[(size_t)s * W3_DTAB];
  dbias_table[(size_t)i * heads + h] += t;
}

// ================================================================================================
// forward.  The batch-persistent structure of attn_fwd_win2_kernel (attention.hip: a workgroup = (head, group of 7 query tiles, chunk
// of clips), a wave = one query tile with the whole 16 x L score block in accumulators, exact two-pass softmax, the bias block of the
// wave as packed-bf16 registers entering through the score MFMA's C operand, K / V of the next sequence streamed into the other LDS
// buffer) on the region-major layout: the bias block no longer carries the shift mask, so it is built ONCE per workgroup instead of
// once per window position, and the score / exp2 / P V work of the key tiles a window type masks for the wave's query class is absent
// (the tile loops exist once per live-class set, chosen per window position).
// ================================================================================================
template <bool MASK>
__global__ __launch_bounds__(448) void attn_fwd_win3_kernel(const vmvm_attn_fwd_desc p, const int nqg, const int nch) {
  constexpr int HD = 32, NWV = 7, NX = w3::NT, NPK = (NX + 1) / 2;
  constexpr int LP32 = NPK * 32, KV = LP32 * HD * 2;                    // 416 rows per image: the last tile PAIR reaches past the window
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const s16x4_ ident = bias_ident_frag(lane);
  const int L = w3::L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nqg);
  const int qg = logical % nqg;
  const int t1 = logical / nqg;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  float* tabs = reinterpret_cast<float*>(smem + 4 * KV);                // this head's bias-table column
  int* rcs = reinterpret_cast<int*>(tabs + 15 * 169 + 1);
  const int qt = qg * NWV + wave;
  const int q = qt * 16 + r;
  const bool active = qt < NX;
  const bool qv = active && (q < L);
  const int qc = qt < w3::CB[1] ? 0 : qt < w3::CB[2] ? 1 : qt < w3::CB[3] ? 2 : 3;

  for (int i = tid; i < 15 * 169; i += NWV * 64) tabs[i] = p.bias_table[(size_t)i * heads + h];
  for (int i = tid; i < NX * 16; i += NWV * 64) rcs[i] = i < L ? p.rc[i] : 0;
  __syncthreads();
  uint32_t bm[NX * 2];                                    // bias of this wave's score block, packed bf16 (padding keys: PAD_BIAS)
  {
    const int rcq = rcs[qv ? q : 0] + p.rc0;
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const int key0 = t * 16 + g * 4;
      const int4 rk = *reinterpret_cast<const int4*>(rcs + key0);
      const int rks[4] = {rk.x, rk.y, rk.z, rk.w};
      float b4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b4[j] = (t < NX - 1 || key0 + j < L) ? tabs[rcq - rks[j]] : PAD_BIAS;
      bm[2 * t] = pack_bf2(b4[0], b4[1]);
      bm[2 * t + 1] = pack_bf2(b4[2], b4[3]);
      if ((t & 1) == 1) __builtin_amdgcn_sched_barrier(0);                // straight-line code: keep the tiles' reads from piling up (registers)
    }
  }

  // this workgroup's sequences: the clips [c0, c1) of every window position, window-major
  const int cper = (B + nch - 1) / nch;
  const int c0 = ch * cper, c1 = (c0 + cper < B) ? c0 + cper : B;
  const int ncl = c1 > c0 ? c1 - c0 : 0, total = ncl * nWin;
  int w_nx = 0, c_nx = c0;
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == c1) { c_nx = c0; ++w_nx; } };
  const uint32_t off_q = (uint32_t)q * p.ld_qkv + p.q_off + h * HD + g * 8;
  const uint32_t off_o = (uint32_t)q * p.ld_out + h * HD + g * 4;
  const uint32_t off_ls = (uint32_t)h * L + q;
  constexpr int NF = (LP32 * 4 + NWV * 64 - 1) / (NWV * 64);            // 16-byte DMA requests per thread per image (4)
  static_assert(NF <= 5, "the fill requests are spread over the live key tiles (the smallest live set is class D: 5 tiles)");
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
  if (total > 0) {
    const u16* kv0 = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    fill_pre<NF, NWV * 64 * 16>(smem + (tid & ~63) * 16, KV, kv0 + p.k_off, kv0 + p.v_off, fill_bytes, goff);
    fetch(seq_nx());
    advance();
  }

  int wprev = -1, w_cu = 0, c_cu = c0, m4 = 15;
  for (int b = 0; b < total; ++b) {
    const int cur = b & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    fill_wait();
    __syncthreads();                                      // sequence b landed for everyone; everyone left the other buffer
    const int wcur = w_cu;
    if (++c_cu == c1) { c_cu = c0; ++w_cu; }
    if (MASK && wcur != wprev) {
      wprev = wcur;
      m4 = w3_live_rt(qc, __builtin_amdgcn_readfirstlane(w3_window_type(p.region, wcur)));
    }
    const bf16x8 cqf = qf;
    const float seq_scale = ss_n;
    const bool has_next = b + 1 < total;
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
    const __amdgpu_buffer_rsrc_t rk_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(kv_nx + p.k_off)), 0, __builtin_amdgcn_readfirstlane((int)fill_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(kv_nx + p.v_off)), 0, __builtin_amdgcn_readfirstlane((int)fill_bytes), 0x00020000);
    f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float mx = NEG_INF, sum = 1.f;

    auto walk = [&](auto mc) {
      constexpr int M4 = decltype(mc)::value;
      constexpr w3::TileList LST = w3::list_all(M4);
      constexpr int NL = LST.n;
      f32x4 acc[NX];
      // pass 0: the bias blocks of the live tiles through the matrix core (independent products, NL - 1 >= 4 of them between any block
      // and the score MFMA that accumulates onto it -- see the hazard note at bias_block_mfma)
#pragma unroll
      for (int i = 0; i < NL; ++i) acc[LST.t[i]] = bias_block_mfma(ident, bm[2 * LST.t[i]], bm[2 * LST.t[i] + 1], f32x4{0.f, 0.f, 0.f, 0.f});
      __builtin_amdgcn_sched_barrier(0);
      // pass 1: scores for the live part of the row block
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int t = LST.t[i];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(kb + t * 1024), cqf, acc[t], 0, 0, 0);
        if (i < NF && has_next) fill_one_r(dst_nx + i * NWV * 64 * 16, KV, rk_nx, rv_nx, goff[i], i == NF - 1);
      }
      // (inline-asm maxima: hipcc inserts no MFMA-result wait states in front of inline asm -- the barrier keeps them BEHIND all score
      // MFMAs, see attn_fwd_win2_kernel)
      __builtin_amdgcn_sched_barrier(0);
      float mx1 = NEG_INF;
#pragma unroll
      for (int i = 0; i < NL; ++i) { const int t = LST.t[i]; mx = max3_f32(mx, acc[t][0], acc[t][1]); mx1 = max3_f32(mx1, acc[t][2], acc[t][3]); }
      __builtin_amdgcn_sched_barrier(0);
      mx = fmaxf(mx, mx1);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      f32x2 nm2 = {-mx * LOG2E, -mx * LOG2E};
      asm volatile("" : "+v"(nm2));                         // real register pair
      // pass 2: p = exp2((s - max) * log2 e); P V and the row sums (all-ones A operand) through the matrix core, per live tile pair
      typedef __attribute__((ext_vector_type(8))) short s16x8o;
      const bf16x8 ones8 = __builtin_bit_cast(bf16x8, s16x8o{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});
      f32x4 osum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < NL; i += 2) {                      // classes start on even tiles: list entries (i, i + 1) are a tile pair of the image
        const int c = LST.t[i] >> 1;
        s16x4 a0, a1, c0_, c1_;
        tr_read4(a0, a1, c0_, c1_, tv0a, tv1a, c * 2048);
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
        tr_wait4(a0, a1, c0_, c1_);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const s16x8 v1 = {c0_[0], c0_[1], c0_[2], c0_[3], c1_[0], c1_[1], c1_[2], c1_[3]};
        o[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v0), pf, o[0], 0, 0, 0);
        o[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v1), pf, o[1], 0, 0, 0);
        osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, pf, osum, 0, 0, 0);
      }
      sum = osum[0];
    };
    if (MASK) {
      switch (m4) {
        case 15: walk(IC<15>{}); break;
        case 3: walk(IC<3>{}); break;
        case 12: walk(IC<12>{}); break;
        case 5: walk(IC<5>{}); break;
        case 10: walk(IC<10>{}); break;
        case 1: walk(IC<1>{}); break;
        case 2: walk(IC<2>{}); break;
        case 4: walk(IC<4>{}); break;
        default: walk(IC<8>{}); break;
      }
    } else {
      walk(IC<15>{});
    }
    if (qv) {
      const float inv = seq_scale / sum;
      u16* op = reinterpret_cast<u16*>(p.out) + seq * L * p.ld_out + off_o;
      *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf2(o[0][0] * inv, o[0][1] * inv), pack_bf2(o[0][2] * inv, o[0][3] * inv));
      *reinterpret_cast<uint2*>(op + 16) = make_uint2(pack_bf2(o[1][0] * inv, o[1][1] * inv), pack_bf2(o[1][2] * inv, o[1][3] * inv));
      if (g == 0) (p.lse + seq * heads * L)[off_ls] = mx + __builtin_amdgcn_logf(sum) * LN2;
    }
  }
}

template <typename K>
int w3_set_smem(K kernel, int bytes) {
  if (bytes > 160 * 1024) return VMVM_ENOSUPPORT;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return VMVM_EHIP;
  }
  return VMVM_OK;
}

// chunks of the CLIP range per base workgroup set (a workgroup takes its clips of every window position): whole rounds of the 256 CUs x
// sequences per round + per-workgroup set-up worth `setup` sequences
int w3_chunks(int base, int nclip, int nwin, float setup) {
  int nch = 1; float best = 1e30f;
  for (int c = 1; c <= 64 && c <= nclip; ++c) {
    const float cost = (float)((base * c + 255) / 256) * ((float)(((nclip + c - 1) / c) * nwin) + setup);
    if (cost < best - 1e-6f) { best = cost; nch = c; }
  }
  return nch;
}

}  // namespace

namespace vmvm_w3 {

__attribute__((visibility("hidden"))) bool applicable(const vmvm_attn_fwd_desc* d) {
  const int nwin = d->n_win > 0 ? d->n_win : 1;
  return d->win_layout == 1 && d->mode == 0 && d->L == w3::L && d->head_dim == 32 && d->table_len == 15 * 169 && d->rc0 == 7 * 169 + 6 * 13 + 6 &&
         d->dropout_p == 0.f && d->nseq % nwin == 0;
}

__attribute__((visibility("hidden"))) int launch_fwd(const vmvm_attn_fwd_desc* d, hipStream_t st) {
  constexpr int KV = 416 * 64;
  const int smem = 4 * KV + (15 * 169 + 1) * 4 + 400 * 4;
  const int nqg = 4, base = d->heads * nqg;
  const int nwin = d->n_win > 0 ? d->n_win : 1;
  const int nch = w3_chunks(base, d->nseq / nwin, nwin, 3.f);
  if (d->region) {
    int rc_ = w3_set_smem(attn_fwd_win3_kernel<true>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_fwd_win3_kernel<true>), dim3(base * nch), dim3(448), smem, st, *d, nqg, nch);
  } else {
    int rc_ = w3_set_smem(attn_fwd_win3_kernel<false>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_fwd_win3_kernel<false>), dim3(base * nch), dim3(448), smem, st, *d, nqg, nch);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

__attribute__((visibility("hidden"))) int launch_dq(const vmvm_attn_bwd_desc* d, hipStream_t st) {
  constexpr int KV = 400 * 64;
  // Key parts per query tile: 2.  Three (12 waves = 3 per SIMD, 9 running-sum slots per wave, bf16 partials, no fragment prefetch: 168
  // registers with 3-20 spilled) measured 15 % SLOWER at every stage (profiles/r04_ab_dq_win3_three_parts.txt); the kernel stays generic in NS.
  constexpr int ns = 2;
  const int smem = 4 * KV + 2 * (ns - 1) * 4 * 64 * 8 * (ns == 2 ? 4 : 2) + w3::TAB_BYTES;
  const int nqg = 7;                      // query groups of 4 tiles (4,4,4,4,4,4,1); 8 groups of 3-4 tiles measured the same (round 4), the kernel takes either
  const int base = d->f.heads * nqg;
  const int nwin = d->f.n_win > 0 ? d->f.n_win : 1;
  const int nch = w3_chunks(base, d->f.nseq / nwin, nwin, 10.f);
  const bool mask = d->f.region != nullptr;
#define W3_LAUNCH_DQ(MASK, NS_)                                                                                       \
  do {                                                                                                               \
    int rc_ = w3_set_smem(attn_bwd_dq_win3_kernel<MASK, NS_>, smem);                                                  \
    if (rc_) return rc_;                                                                                             \
    hipLaunchKernelGGL((attn_bwd_dq_win3_kernel<MASK, NS_>), dim3(base * nch), dim3(4 * NS_ * 64), smem, st, *d, nqg, nch); \
  } while (0)
  vmvm_attn_bwd_desc dd = *d;
  const int64_t need = (int64_t)base * nch * W3_DTAB * (int64_t)sizeof(float);
  if (!dd.dbias_table || !dd.dbias_ws || dd.dbias_ws_bytes < need) dd.dbias_ws = nullptr;      // (atomics)
  d = &dd;
  if (mask) W3_LAUNCH_DQ(true, 2); else W3_LAUNCH_DQ(false, 2);
  VMVM_CHECK_LAUNCH();
  if (dd.dbias_ws) {
    hipLaunchKernelGGL(w3_dbias_reduce_kernel, dim3((15 * 169 + 255) / 256, dd.f.heads), dim3(256), 0, st, reinterpret_cast<const float*>(dd.dbias_ws), dd.dbias_table,
                       dd.f.heads, nqg * nch);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}

// bytes of vmvm_attn_bwd_desc.dbias_ws for this problem (same plan as launch_dq)
__attribute__((visibility("hidden"))) int64_t dbias_ws_size(const vmvm_attn_bwd_desc* d) {
  const int nqg = 7, base = d->f.heads * nqg;
  const int nwin = d->f.n_win > 0 ? d->f.n_win : 1;
  const int nch = w3_chunks(base, d->f.nseq / nwin, nwin, 10.f);
  return (int64_t)base * nch * W3_DTAB * (int64_t)sizeof(float);
}

__attribute__((visibility("hidden"))) int launch_dkv(const vmvm_attn_bwd_desc* d, hipStream_t st) {
  constexpr int IMG = 400 * 64, BUF = 2 * IMG + 2 * 512 * 4;
  const int smem = 2 * BUF + w3::TAB_BYTES;
  // 8 waves x two key tiles, two key groups per head.  (12 waves on ONE group -- three waves per SIMD, 24 of the 25 key tiles -- measured
  // -15 % before the cost of the odd tile, profiles/r04_window_attention_dkv_win3_anatomy.txt section 5; the kernel is generic in NWV.)
  const int base = d->f.heads * 2;
  const int nwin = d->f.n_win > 0 ? d->f.n_win : 1;
  const int nch = w3_chunks(base, d->f.nseq / nwin, nwin, 4.f);
  const bool mask = d->f.region != nullptr;
#define W3_LAUNCH_DKV(MASK, NWV)                                                                                     \
  do {                                                                                                               \
    int rc_ = w3_set_smem(attn_bwd_dkv_win3_kernel<MASK, NWV>, smem);                                                 \
    if (rc_) return rc_;                                                                                             \
    hipLaunchKernelGGL((attn_bwd_dkv_win3_kernel<MASK, NWV>), dim3(base * nch), dim3(NWV * 64), smem, st, *d, nch);   \
  } while (0)
  if (mask) W3_LAUNCH_DKV(true, 8); else W3_LAUNCH_DKV(false, 8);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

}  // namespace vmvm_w3
