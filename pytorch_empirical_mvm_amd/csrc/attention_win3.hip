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

namespace {

template <int V> struct IC { static constexpr int value = V; };
#ifndef W3_ABL
#define W3_ABL 0
#endif

// windowed copy of one head's table column (stage: the 2535 entries [delta * 169 + rho] in LDS): row rho holds the 12 windows of 4
// consecutive entries a lane can need.  DIR 0 (a lane's 4 values are consecutive KEYS, delta falls): entry j of window s is
// delta = 14 - s - j; DIR 1 (consecutive QUERIES, delta rises): delta = s + j.  Row 169 = -inf (padding tokens).
template <int DIR>
__device__ __forceinline__ void w3_build_table(unsigned char* tl, const float* stage, int tid, int nthreads) {
  for (int i = tid; i < w3::NROW * 12; i += nthreads) {
    const int rho = i / 12, s = i - rho * 12;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = DIR ? s + j : 14 - s - j;
      v[j] = rho < 169 ? stage[e * 169 + rho] : NEG_INF;
    }
    *reinterpret_cast<f32x4*>(tl + i * 16) = v;
  }
}
__device__ __forceinline__ int w3_posA_rt(int pos) {      // A = 13 h + w of slot position `pos` (run-time index, set-up code only)
  static constexpr int PA[50] = {w3::posA(0),  w3::posA(1),  w3::posA(2),  w3::posA(3),  w3::posA(4),  w3::posA(5),  w3::posA(6),  w3::posA(7),  w3::posA(8),  w3::posA(9),
                                 w3::posA(10), w3::posA(11), w3::posA(12), w3::posA(13), w3::posA(14), w3::posA(15), w3::posA(16), w3::posA(17), w3::posA(18), w3::posA(19),
                                 w3::posA(20), w3::posA(21), w3::posA(22), w3::posA(23), w3::posA(24), w3::posA(25), w3::posA(26), w3::posA(27), w3::posA(28), w3::posA(29),
                                 w3::posA(30), w3::posA(31), w3::posA(32), w3::posA(33), w3::posA(34), w3::posA(35), w3::posA(36), w3::posA(37), w3::posA(38), w3::posA(39),
                                 w3::posA(40), w3::posA(41), w3::posA(42), w3::posA(43), w3::posA(44), w3::posA(45), w3::posA(46), w3::posA(47), w3::posA(48), w3::posA(49)};
  return PA[pos];
}
// window type of window position w from its (tile-uniform) region row: bit 0 = split along h (A | C differ), bit 1 = along w (A | B)
__device__ __forceinline__ int w3_window_type(const uint8_t* region, int w) {
  const uint8_t* rw = region + (size_t)w * w3::L;
  const int ra = rw[0], rb = rw[w3::CB[1] * 16], rcl = rw[w3::CB[2] * 16];
  return (ra != rcl ? 1 : 0) | (ra != rb ? 2 : 0);
}
__device__ __forceinline__ int w3_live_rt(int c, int wt) {
  int m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const bool ok = (!(wt & 1) || ((c >> 1) == (k >> 1))) && (!(wt & 2) || ((c & 1) == (k & 1)));
    m |= ok ? (1 << k) : 0;
  }
  return m;
}
// transposing reads of TWO tiles (byte offsets offa / offb from the lane bases pa: hd 0-15, pc: hd 16-31), asm as tr_read4
__device__ __forceinline__ void tr_read4_2(s16x4& a0, s16x4& a1, s16x4& c0, s16x4& c1, uint32_t pa, uint32_t pc, const int offa, const int offb) {
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
               "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
               : "=&v"(a0), "=&v"(a1), "=&v"(c0), "=&v"(c1) : "v"(pa), "v"(pc), "i"(offa), "i"(offb) : "memory");
}

// ================================================================================================
// dK / dV.  A workgroup = (head, key group, chunk of clips): key group 0 = classes A + B (tiles 0-13), 1 = C + D (tiles 14-24); each
// of its 8 waves owns TWO key tiles of one class (K / V fragments in registers, prefetched per sequence) and walks the LIVE query
// tiles of the window type in pairs; Q / dO images + lse / delta of the next sequence stream into the other LDS buffer by DMA spread
// over the walk.  Lane (r, g): key r of the tile, queries 4g..4g+3 of the query tile (S = Q K^T, rows = queries).
//
// PING-PONG.  Measured on the first version of this kernel (and on the win2 kernels): a sequence took exactly the SUM of its LDS, VALU
// and matrix-pipe times -- the two waves of a SIMD run the same instruction stream in lockstep from the sequence barrier on, so both
// sit in the softmax VALU chain together and then both queue on the matrix pipe.  Here the waves form two groups (waves 0-3 / 4-7: one
// wave of each per SIMD) that run HALF A STEP APART, separated by raw s_barrier: while one group is in its MFMA-only phase
// M(c) = [dV / dK products of pair c, score / dP products of pair c + 1] its SIMD partners are in the VALU + LDS phase
// V(c) = [softmax-side chain of pair c, fragment / bias / lse reads of pair c + 1, transposing reads of pair c, DMA requests].
// Every wave of the workgroup executes the same number of barriers per sequence (2 NPAD + 2): shorter live lists are padded with
// empty phases, idle waves only keep the count.  LDS is read-only inside a sequence, so the raw barriers carry no memory ordering.
// ================================================================================================
constexpr int w3_npad(int m4) {                          // pairs of the longest live list among the classes that share a workgroup
  return m4 == 15 ? 13 : (m4 == 3 || m4 == 5 || m4 == 10) ? 7 : m4 == 12 ? 6 : (m4 == 1 || m4 == 2) ? 4 : 3;
}
__device__ __forceinline__ void w3_bar() {
  __builtin_amdgcn_sched_barrier(0);
#ifdef W3_PINGPONG
  __builtin_amdgcn_s_barrier();
#endif
  __builtin_amdgcn_sched_barrier(0);
}
template <bool MASK>
__global__ __launch_bounds__(512) void attn_bwd_dkv_win3_kernel(const vmvm_attn_bwd_desc pb, const int nch) {
  constexpr int HD = 32, NWV = 8, KT = 2;
  constexpr int ROWS = w3::NT * 16, IMG = ROWS * HD * 2;               // 400 rows, 25 600 bytes per Q or dO image
  constexpr int LV = 512;                                               // floats reserved for lse / delta
  constexpr int BUF = 2 * IMG + 2 * LV * 4;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                             // ping-pong group
  const int r = lane & 15, g = lane >> 4;
  const int L = w3::L, heads = p.heads, nWin = p.n_win, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * 2);
  const int kg = logical & 1;
  const int t1 = logical >> 1;
  const int ch = t1 % nch;
  const int h = t1 / nch;
  unsigned char* TL = smem + 2 * BUF;
  // this wave's key tiles
  const int kt0 = (kg == 0 ? 0 : w3::CB[2]) + 2 * wave;
  const bool active = kt0 < (kg == 0 ? w3::CB[2] : w3::NT);
  const int kc = kt0 < w3::CB[1] ? 0 : kt0 < w3::CB[2] ? 1 : kt0 < w3::CB[3] ? 2 : 3;      // class of both tiles
  int key[KT]; bool kv[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) { key[t] = (kt0 + t) * 16 + r; kv[t] = active && (kt0 + t) < w3::NT && key[t] < L; }

  // windowed table of this head (staged through the second buffer, which the first sequence does not use)
  {
    float* stage = reinterpret_cast<float*>(smem + BUF);
    if (!(W3_ABL & 64)) for (int i = tid; i < 15 * 169; i += NWV * 64) stage[i] = p.bias_table[(size_t)i * heads + h];
    __syncthreads();
    if (!(W3_ABL & 64)) w3_build_table<1>(TL, stage, tid, NWV * 64);
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
  // prefetched K / V fragments of the next sequence.  Loaded by inline asm and waited for by the counted s_waitcnt at the loop top:
  // as compiler-visible loads their consumer at the loop header made the compiler wait with vmcnt(0) -- i.e. for the previous
  // sequence's dK / dV stores as well (it cannot count memory instructions across the back edge).  Lanes of padding keys never load
  // and keep their zeros.
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
  if (grp == 1) __builtin_amdgcn_s_setprio(1);           // the later-dispatched half loses every arbitration otherwise (static form)

  const int nstores = (active && kt0 + 1 < w3::NT - 1) ? 8 : 0;        // both tiles hold 16 real keys: every store below is issued
  int wprev = -1, w_cu = 0, c_cu = c0, m4 = 15, npad = 13;
  for (int b = 0; b < total; ++b) {
    const int cur = b & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    // This sequence's images must have landed -- but NOT the dK / dV stores of the previous sequence, the youngest 8 memory
    // instructions of a wave that owns two key tiles (vmcnt counts loads and stores in order; measured: waiting for the stores' write
    // acknowledgements cost 2.8 us per sequence, a quarter of this kernel).  Raw barrier: LDS is read-only inside a sequence, the
    // barrier only orders "my DMA landed" / "everyone left the other buffer".
    // (ONE operand-carrying wait: with one asm per branch the compiler copied the fragment registers in front of one of them)
    if (!(b > 0 && nstores == 8)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(8)" : "+v"(kf[0]), "+v"(vf[0]), "+v"(kf[1]), "+v"(vf[1]) : : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int wcur = w_cu;
    if (++c_cu == c1) { c_cu = c0; ++w_cu; }
    if (MASK && wcur != wprev) {                          // (workgroup-uniform) new window position: which query classes this wave's keys see
      wprev = wcur;
      const int wt = __builtin_amdgcn_readfirstlane(w3_window_type(p.region, wcur));
      m4 = w3_live_rt(kc, wt);
      npad = kg == 0 ? (wt == 0 ? 13 : wt == 3 ? 4 : 7) : (wt == 0 ? 13 : wt == 1 ? 6 : wt == 2 ? 7 : 3);
    }
    bf16x8 ckf[KT], cvf[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) { ckf[t] = __builtin_bit_cast(bf16x8, kf[t]); cvf[t] = __builtin_bit_cast(bf16x8, vf[t]); }
    const float seq_scale = ss_n;
    const bool has_next = b + 1 < total;
    const size_t sq_nx = seq_nx();
    if (has_next) {
      if (!(W3_ABL & 512)) fetch(sq_nx);
      advance();
    }
    if (!active) {                                        // idle wave: its share of the DMA requests, and the barrier count
      if (has_next && !(W3_ABL & 256)) {
#pragma unroll
        for (int i = 0; i <= NF; ++i) dma_step(sq_nx, cur ^ 1, i);
      }
      for (int i = 0; i < 2 * npad + 1; ++i) w3_bar();
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

    // the walk over the live query tiles of class set M4 (compile-time list, pairs of tiles = one 32-deep dK / dV MFMA)
    auto walk = [&](auto m4c) {
      constexpr int M4 = decltype(m4c)::value;
      constexpr w3::TileList QL = w3::list_all(M4);
      constexpr int NP = (QL.n + 1) / 2, NPAD = w3_npad(M4);
      static_assert(NP <= NPAD, "padding covers the list");
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      bf16x8 qf[2], dof[2];                               // fragments of the NEXT pair (read in V, multiplied in M)
      f32x4 l4[2], d4[2], bias[2][KT];
      f32x4 s4[2][KT], dp4[2][KT];                        // scores / dP of the CURRENT pair (written in M, consumed in V)
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
      auto mma1 = [&](const int c) {                      // S = Q K^T + bias, dP = dO V^T of pair c
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (2 * c + u < QL.n) {
#pragma unroll
            for (int t = 0; t < KT; ++t) {
              s4[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[u], ckf[t], bias[u][t], 0, 0, 0);
              dp4[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[u], cvf[t], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            }
          }
        }
      };
      f32x2 nl[2][2], nd[2][2];
      auto prep = [&](const int c) {                      // -lse log2 e, -delta of pair c (their reads were issued a phase ago)
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
      if (grp == 1) w3_bar();                             // half a step behind group 0
      reads(0);
      mma1(0);
      prep(0);
      if (NP > 1) reads(1);
#pragma unroll
      for (int c = 0; c < NPAD; ++c) {
        // ---- V(c): VALU + LDS phase
        if (c < NP) {
          if (!(W3_ABL & 16)) trreads(c);
          if (!(W3_ABL & 1)) chain(c);
        }
        if (c <= NF && has_next && !(W3_ABL & 256)) dma_step(sq_nx, cur ^ 1, c);
        w3_bar();
        // ---- M(c): matrix phase
        if (c < NP) {
          if (!(W3_ABL & 2)) mma2();
          if (c + 1 < NP) {
            if (!(W3_ABL & 4)) mma1(c + 1);
            prep(c + 1);
            if (c + 2 < NP && !(W3_ABL & 8)) reads(c + 2);
          }
        }
        w3_bar();
      }
      if (has_next && !(W3_ABL & 256)) {                                     // (short walks: the DMA steps the phases did not cover)
#pragma unroll
        for (int c = NPAD; c <= NF; ++c) dma_step(sq_nx, cur ^ 1, c);
      }
      if (grp == 0) w3_bar();
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
    u16* dbase = reinterpret_cast<u16*>(pb.dqkv) + seq * L * pb.ld_dqkv;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      if (kv[t] && (!(W3_ABL & 128) || dk[t][0][0] == 1234.5f)) {
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

__attribute__((visibility("hidden"))) int launch_dkv(const vmvm_attn_bwd_desc* d, hipStream_t st) {
  constexpr int IMG = 400 * 64, BUF = 2 * IMG + 2 * 512 * 4;
  const int smem = 2 * BUF + w3::TAB_BYTES;
  const int base = d->f.heads * 2;
  const int nwin = d->f.n_win > 0 ? d->f.n_win : 1;
  const int nch = w3_chunks(base, d->f.nseq / nwin, nwin, 4.f);
  const bool mask = d->f.region != nullptr;
  if (mask) {
    int rc_ = w3_set_smem(attn_bwd_dkv_win3_kernel<true>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_bwd_dkv_win3_kernel<true>), dim3(base * nch), dim3(512), smem, st, *d, nch);
  } else {
    int rc_ = w3_set_smem(attn_bwd_dkv_win3_kernel<false>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_bwd_dkv_win3_kernel<false>), dim3(base * nch), dim3(512), smem, st, *d, nch);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

}  // namespace vmvm_w3
