// attention_win4.hip -- Video-Swin window attention (WindowAttention3D.forward video_swin.py:147-172 with the shift mask of
// compute_mask :292-307) for windows of (8,7,7) = 392 tokens, head_dim 32, tokens in the win_layout = 1 order (include/vmvm.h,
// attn_win3.h, swin_index.win3_perm): the KEY-BLOCKED kernels of round 5.
//
// Why another family.  The round-4 kernels (attention_win3.hip) keep a wave's whole 16 x 392 score block in ~100
// accumulator registers (exact two-pass softmax), i.e. 1.75-2 waves per SIMD -- and the round-4 probes showed that ONE wave issues a
// softmax-side VALU instruction only every 6-10 cycles, that the VALU saturates from three waves per SIMD on, and that at two waves per
// SIMD the matrix, VALU and LDS times of a wave ADD.  These kernels keep the per-wave state small instead (<= 128 registers: four
// waves per SIMD, 13 waves = one 32-query job each per workgroup):
//  * a wave = TWO 16-query tiles against one 32-key block at a time: every K / V fragment read from LDS serves two score tiles (the
//    LDS bytes per score element of a 32 x 32 MFMA tile at the 16-token granularity of the region-major layout);
//  * ONE pass over the keys with a FIXED softmax reference m = the row maximum over the first live key block: p = 2^((s - m) log2 e) is
//    exact for any m (softmax is shift-invariant; bf16 / f32 keep their relative precision at any exponent), so the per-element chain is
//    one packed fma, one exp2 and half a convert -- no running maximum, no rescale, no per-element compare -- and the row sums come from
//    the matrix core (all-ones A operand).  The only way it can fail is overflow (a later score more than ~88 above the first block's
//    maximum): then l or O is not finite, the wave notices at the end of the sequence and repeats it with the exact row maximum
//    (a run-time loop over the live blocks; tested by a spiked input, tools/gpu_check.py check_attn_window_spike);
//  * bias = one 16-byte LDS read per lane and score tile from the windowed table copy, as the score MFMA's C operand (attn_win3_dev.h);
//    masked (query class, key class) pairs absent at compile time (one tile walk per live-class set).
#include "attn_common.h"
#include "attn_win3.h"
#include "attn_win3_dev.h"
#include <vmvm_probe_hooks.h>
#include <cstdlib>

namespace {

// Per-wave timeline stamps (tools/scratch/w4_timeline.py) and the "no odd tile" ablation are HOOKS (vmvm_probe_hooks.h): no-ops /
// constants in the library build, instrumented in tools/probe/hooks (switches -DW4_TIMELINE, -DW4_NO_ODD live there, not here).
#define W4_STAMP(idx) vmvm_hook::w4_stamp(p.drop_mask, (int)blockIdx.x, b, lane, wave, (idx))

template <int V> struct IC4 { static constexpr int value = V; };

constexpr int W4_ROWS = 416;                       // rows per K / V (Q / dO) image: 13 blocks of 32 tokens, rows >= 392 are zero
constexpr int W4_IMG = W4_ROWS * 64;               // 26 624 bytes
constexpr int W4_NB = 13;                          // 32-token blocks per window (block 12 = tile 24 + an all-padding tile 25)

// first 32-token block of each class (classes start on even tiles: 0, 8, 14, 20)
__device__ __forceinline__ int w4_block_class(int c) { return c < 4 ? 0 : c < 7 ? 1 : c < 10 ? 2 : 3; }

// ---- LDS reads through inline asm (immediate offsets; completion by the COUNTED waits below -- LDS operations of a wave return in
// order, so lgkmcnt(N) = "all but the youngest N have landed"; scalar loads the compiler may add only make a counted wait stricter).
// Plain C++ reads of an LDS region that a direct-to-LDS DMA writes would make hipcc drain vmcnt in front of them (attn_common.h).
__device__ __forceinline__ void w4_read_b128(f32x4& d, uint32_t addr, const int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(off) : "memory");
}
__device__ __forceinline__ void w4_read_frag(bf16x8& d, uint32_t addr, const int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(off) : "memory");
}
template <int N>
__device__ __forceinline__ void w4_wait_kb(bf16x8& k0, bf16x8& k1, f32x4& s00, f32x4& s01, f32x4& s10, f32x4& s11) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(k0), "+v"(k1), "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11) : "i"(N));
}
template <int N>
__device__ __forceinline__ void w4_wait_kb1(bf16x8& k0, bf16x8& k1, f32x4& s00, f32x4& s01) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(k0), "+v"(k1), "+v"(s00), "+v"(s01) : "i"(N));
}
template <int N>
__device__ __forceinline__ void w4_wait_v(s16x4& a0, s16x4& a1, s16x4& c0, s16x4& c1) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1) : "i"(N));
}

constexpr int W4_ROWB = w3::ROWB;                  // 256-byte table rows (attn_win3.h: bank-conflict-free b128 reads)
constexpr int W4_TAB_BYTES = w3::TAB_BYTES;
// How the next sequence's register fragments are fetched (measured, tools/scratch/w4_timeline.py): they are requested by PLAIN loads in
// the epilogue, in front of the output stores, and an empty asm statement behind the stores makes the compiler wait for them there --
// inside one basic block it counts the stores and emits s_waitcnt vmcnt(#stores), so the stores stay in flight (their acknowledgement
// takes ~1 500 cycles).  Two earlier forms lost: (i) asm loads inside the walk's last key block + a counted wait at the loop top -- the
// destinations were live across two separately allocated walk variants and the back edge, and hipcc merged those paths with register
// copies that read the destinations before the data had landed (wrong Q on some launches: an asm load's destination must not stay live
// across code the compiler allocates freely, cdna_hip_programming 5.7 item 1); (ii) loads + vmcnt(0) in one asm statement behind the
// stores: correct, and 2 000 cycles per sequence waiting for the store acknowledgements.
template <typename T>
__device__ __forceinline__ void w4_need(T& a, T& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// ================================================================================================
// forward.  Workgroup = (head, chunk of clips, residue class of window positions): 13 waves, wave w = query tiles 2w, 2w + 1 (wave 12:
// tile 24 alone = 8 real queries, the NQ = 1 walks); batch-persistent, K / V of the next sequence streamed into the other LDS buffer
// by DMA requests issued inside the first key blocks of the walk.  Lane (r, g): query r of each tile, keys 4g..4g+3 of each key tile
// (S^T = K Q^T: the P rows feed the P V product as its B operand straight from the accumulator layout).
// LDS reads of key block i, in issue order: [K_i, bias_i: 6, requested during block i - 1's softmax] V_i: 4 [K_(i+1), bias_(i+1): 6].
// ================================================================================================
template <bool MASK>
__global__ __launch_bounds__(832) void attn_fwd_win4_kernel(const vmvm_attn_fwd_desc p, const int nch, const int nwg) {
  constexpr int HD = 32, NWV = 13, KV = W4_IMG;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int L = w3::L, heads = p.heads, nWin = p.n_win > 0 ? p.n_win : 1, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nwg);
  const int wg = logical % nwg;                           // window positions wg, wg + nwg, ... (interleaved: every workgroup sees every window type)
  const int ch = (logical / nwg) % nch;
  const int h = logical / (nwg * nch);
  unsigned char* TL = smem + 4 * KV;
  {                                                       // windowed table of this head (staged through the second K / V buffer)
    float* stage = reinterpret_cast<float*>(smem + 2 * KV);
    for (int i = tid; i < 15 * 169; i += NWV * 64) stage[i] = p.bias_table[(size_t)i * heads + h];
    __syncthreads();
    w3_build_table<0>(TL, stage, tid, NWV * 64);
    __syncthreads();
  }
  // this wave's query tiles (wave 12: tile 24 only)
  const bool odd = !vmvm_hook::W4_SKIP_ODD && wave == NWV - 1;
  const int qt[2] = {2 * wave, odd ? 2 * wave : (2 * wave + 1 < w3::NT ? 2 * wave + 1 : w3::NT - 1)};     // (vmvm_hook::W4_SKIP_ODD probe builds: wave 12's second tile repeats tile 24, not stored)
  const int qc = w3::cls_of(qt[0]);                       // both tiles lie in one class (classes start on even tiles)

  // this workgroup's sequences: the clips [c0, c1) of its window positions, window-major
  const int cper = (B + nch - 1) / nch;
  const int c0 = ch * cper, c1 = (c0 + cper < B) ? c0 + cper : B;
  const int ncl = c1 > c0 ? c1 - c0 : 0, total = ncl * ((nWin - wg + nwg - 1) / nwg);
  int w_nx = wg, c_nx = c0;
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == c1) { c_nx = c0; w_nx += nwg; } };
  constexpr int NF = (W4_ROWS * 4) / (NWV * 64);                        // 16-byte DMA requests per thread per image: exactly 2
  static_assert(NF * NWV * 64 == W4_ROWS * 4, "the image is an exact number of workgroup-wide requests");
  const unsigned fill_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2);
  // Per-lane addresses are REBUILT from the lane id at the top of every sequence (behind an opaque copy of it): as loop invariants of the
  // whole kernel they are ~20 registers the compiler spills, and every scratch reload is an s_waitcnt vmcnt(0) -- i.e. a wait for the
  // previous sequence's output stores (measured: 1-2 k cycles per sequence, tools/scratch/w4_timeline.py)
  struct LaneAddr { uint32_t off_q[2], goff[NF]; };
  auto lane_addr = [&](int ln) __attribute__((always_inline)) {
    LaneAddr a;
    const int r_ = ln & 15, g_ = ln >> 4, tid_ = wave * 64 + ln;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int qq = qt[x] * 16 + r_;
      a.off_q[x] = (uint32_t)(qq < L ? qq : L - 1) * p.ld_qkv + p.q_off + h * HD + g_ * 8;      // padding queries read row L - 1 (finite, never stored)
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int u = i * NWV * 64 + tid_, row = u >> 2, chs = u & 3;
      a.goff[i] = (uint32_t)((row * p.ld_qkv + ((chs ^ swz_chunk<32>(row)) << 3)) * 2);
    }
    return a;
  };
  bf16x8 qf[2];
  auto fetch = [&](size_t seq, const LaneAddr& a, bf16x8 (&dst)[2]) __attribute__((always_inline)) {
    const u16* qb = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
#pragma unroll
    for (int x = 0; x < 2; ++x) dst[x] = *reinterpret_cast<const bf16x8*>(qb + a.off_q[x]);
  };
  if (total > 0) {
    const LaneAddr a0 = lane_addr(lane);
    const u16* kv0 = reinterpret_cast<const u16*>(p.qkv) + seq_nx() * L * p.ld_qkv + h * HD;
    fill_pre<NF, NWV * 64 * 16>(smem + wave * 1024, KV, kv0 + p.k_off, kv0 + p.v_off, fill_bytes, a0.goff);
    fetch(seq_nx(), a0, qf);
    advance();
    w4_need(qf[0], qf[1]);
    fill_wait();
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const bf16x8 ones8 = __builtin_bit_cast(bf16x8, s16x8{0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80});

  int wprev = -1, w_cu = wg, c_cu = c0, m4 = 15;
  for (int b = 0; b < total; ++b) {
    const int cur = b & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    W4_STAMP(0);
    __syncthreads();                                      // (every wave's DMA requests for sequence b landed before it arrived: the epilogue's wait)
    W4_STAMP(1);                                      // sequence b landed for everyone; everyone left the other buffer
    const int wcur = w_cu;
    if (++c_cu == c1) { c_cu = c0; w_cu += nwg; }
    if (MASK && wcur != wprev) {
      wprev = wcur;
      m4 = w3_live_rt(qc, __builtin_amdgcn_readfirstlane(w3_window_type(p.region, wcur)));
    }
    const float seq_scale = p.seq_scale ? p.seq_scale[(uint32_t)seq / (uint32_t)p.seqs_per_scale] : 1.0f;
    const bool has_next = b + 1 < total;
    const size_t seq_n = seq_nx();
    const u16* kv_nx = reinterpret_cast<const u16*>(p.qkv) + seq_n * L * p.ld_qkv + h * HD;
    unsigned char* dst_nx = smem + (cur ^ 1) * 2 * KV + wave * 1024;
    if (has_next) advance();
    int ln = lane;
    asm volatile("" : "+v"(ln));                            // opaque per sequence: nothing below is a kernel-wide loop invariant
    const int r = ln & 15, g = ln >> 4;
    const LaneAddr la = lane_addr(ln);
    // lane bases into the table: row = A(q) - A(k) + 84, window s = 7 - (r & 7) + 4 (g & 1); A(k) = tile immediate + lk * step
    const int lq = r >> 3, lk = g >> 1, sw = 7 - (r & 7) + 4 * (g & 1);
    const uint32_t tl0 = lds_addr(TL) + sw * 16;
    uint32_t tb1[2], tb13[2], tb24[2];
    const uint32_t tbpad = tl0 + 169 * W4_ROWB;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int aq = w3_posA_rt(2 * qt[x] + lq);
      tb1[x] = tl0 + (aq - lk) * W4_ROWB;
      tb13[x] = tl0 + (aq - 13 * lk) * W4_ROWB;
      tb24[x] = lk ? tbpad : tl0 + (aq + 84 - w3::tileA0(w3::NT - 1)) * W4_ROWB;      // key tile 24: second position = padding
    }
    const unsigned char* Ksm = smem + cur * 2 * KV;
    const uint32_t kba = lds_addr(Ksm + k_off_swz<HD>(r, g));
    const uint32_t tv0a = lds_addr(Ksm + KV + k_off_swz<HD>(g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8);
    const uint32_t tv1a = lds_addr(Ksm + KV + k_off_swz<HD>(g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8);
    const __amdgpu_buffer_rsrc_t rk_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(kv_nx + p.k_off)), 0, __builtin_amdgcn_readfirstlane((int)fill_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv_nx = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(kv_nx + p.v_off)), 0, __builtin_amdgcn_readfirstlane((int)fill_bytes), 0x00020000);

    f32x4 o[2][2], osum[2];
    float mref[2] = {0.f, 0.f};
    bf16x8 qn[2];

    // K fragments of key tiles (t, t + 1): 2 reads; their bias blocks for the wave's query tiles: 2 NQ reads
    auto issue_k = [&](const int t, bf16x8 (&kf)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) w4_read_frag(kf[u], kba, (t + u) * 1024);
    };
    auto issue_b = [&](auto nqc, const int t, f32x4 (&s)[2][2]) __attribute__((always_inline)) {
      constexpr int NQ = decltype(nqc)::value;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int kt = t + u;
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
          if (kt >= w3::NT) w4_read_b128(s[x][u], tbpad, 0);
          else if (kt == w3::NT - 1) w4_read_b128(s[x][u], tb24[x], 0);
          else if (w3::tileStep(kt < w3::NT ? kt : 0) == 13) w4_read_b128(s[x][u], tb13[x], (84 - w3::tileA0(kt < w3::NT ? kt : 0)) * W4_ROWB);
          else w4_read_b128(s[x][u], tb1[x], (84 - w3::tileA0(kt < w3::NT ? kt : 0)) * W4_ROWB);
        }
      }
    };

    // one walk over the live key blocks; given == true: the softmax reference is mref (exact row maxima), else the first block's maxima
    auto walk = [&](auto mc, auto nqc, const bool given, const bool first) __attribute__((always_inline)) {
      constexpr int M4 = decltype(mc)::value, NQ = decltype(nqc)::value;
      constexpr w3::TileList LST = w3::list_all(M4);
      constexpr int NP = (LST.n + 1) / 2;
      constexpr int NKB = 2 + 2 * NQ;                       // reads of one (K, bias) request group
      const bf16x8 cqf[2] = {qf[0], qf[1]};
      bf16x8 kf[2][2];
      f32x4 s[2][2][2];
      f32x2 nm2[2];
      issue_k(LST.t[0], kf[0]);
      issue_b(nqc, LST.t[0], s[0]);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int cb = i & 1;
        const int t = LST.t[2 * i];
        s16x4 a0, a1, c0_, c1_;
        tr_read4(a0, a1, c0_, c1_, tv0a, tv1a, (t >> 1) * 2048);
        if (NQ == 2) w4_wait_kb<4>(kf[cb][0], kf[cb][1], s[cb][0][0], s[cb][0][1], s[cb][1][0], s[cb][1][1]);
        else w4_wait_kb1<4>(kf[cb][0], kf[cb][1], s[cb][0][0], s[cb][0][1]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int x = 0; x < NQ; ++x) s[cb][x][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cb][u], cqf[x], s[cb][x][u], 0, 0, 0);
        if (i + 1 < NP) {
          issue_k(LST.t[2 * i + 2], kf[cb ^ 1]);
          issue_b(nqc, LST.t[2 * i + 2], s[cb ^ 1]);
        }
        if (first && i < NF && has_next) fill_one_r(dst_nx + i * NWV * 64 * 16, KV, rk_nx, rv_nx, la.goff[i], false);
        if (first && i == NP - 1 && has_next) fetch(seq_n, la, qn);       // plain loads: they land under this block's softmax / P V and the epilogue
        if (i == 0) {
#pragma unroll
          for (int x = 0; x < NQ; ++x) {
            float mx = mref[x];
            if (!given) {
              mx = fmaxf(fmaxf(fmaxf(s[cb][x][0][0], s[cb][x][0][1]), fmaxf(s[cb][x][0][2], s[cb][x][0][3])),
                         fmaxf(fmaxf(s[cb][x][1][0], s[cb][x][1][1]), fmaxf(s[cb][x][1][2], s[cb][x][1][3])));
              mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
              mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
              mref[x] = mx;
            }
            nm2[x] = f32x2{-mx * LOG2E, -mx * LOG2E};
            asm volatile("" : "+v"(nm2[x]));               // a real register pair
          }
        }
        bf16x8 pf[2];
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
          uint32_t pw[4];
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
              const f32x2 e = __builtin_elementwise_fma(f32x2{s[cb][x][u][2 * hj], s[cb][x][u][2 * hj + 1]}, f32x2{LOG2E, LOG2E}, nm2[x]);
              pw[2 * u + hj] = pack_bf2v(f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])});
            }
          pf[x] = __builtin_bit_cast(bf16x8, make_uint4(pw[0], pw[1], pw[2], pw[3]));
        }
        if (i + 1 < NP) w4_wait_v<NKB>(a0, a1, c0_, c1_); else w4_wait_v<0>(a0, a1, c0_, c1_);
        const s16x8 v0 = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const s16x8 v1 = {c0_[0], c0_[1], c0_[2], c0_[3], c1_[0], c1_[1], c1_[2], c1_[3]};
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
          o[x][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v0), pf[x], o[x][0], 0, 0, 0);
          o[x][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v1), pf[x], o[x][1], 0, 0, 0);
          osum[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, pf[x], osum[x], 0, 0, 0);
        }
        W4_STAMP(3 + i);
      }
    };
    auto walk_n = [&](auto nqc, const bool given, const bool first) __attribute__((always_inline)) {
#pragma unroll
      for (int x = 0; x < 2; ++x) { o[x][0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[x][1] = f32x4{0.f, 0.f, 0.f, 0.f}; osum[x] = f32x4{1.f, 1.f, 1.f, 1.f}; }
#pragma unroll
      for (int x = 0; x < decltype(nqc)::value; ++x) osum[x] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (MASK) {
        switch (m4) {
          case 15: walk(IC4<15>{}, nqc, given, first); break;
          case 3: walk(IC4<3>{}, nqc, given, first); break;
          case 12: walk(IC4<12>{}, nqc, given, first); break;
          case 5: walk(IC4<5>{}, nqc, given, first); break;
          case 10: walk(IC4<10>{}, nqc, given, first); break;
          case 1: walk(IC4<1>{}, nqc, given, first); break;
          case 2: walk(IC4<2>{}, nqc, given, first); break;
          case 4: walk(IC4<4>{}, nqc, given, first); break;
          default: walk(IC4<8>{}, nqc, given, first); break;
        }
      } else {
        walk(IC4<15>{}, nqc, given, first);
      }
    };
    // wave 12 holds query class D only: its walks exist for the live sets of class D (15, 12, 10, 8) -- the switch above is shared
    auto walk_m = [&](const bool given, const bool first) __attribute__((always_inline)) { if (odd) walk_n(IC4<1>{}, given, first); else walk_n(IC4<2>{}, given, first); };
    // l in (0, 2^125) and every |O| < inf (NaN fails both compares).  The bound on l is NOT "finite": the epilogue multiplies by
    // v_rcp_f32(l), which flushes a denormal result to zero -- for l in (2^126, 2^128) the row came out as ZEROS with a correct lse
    // (a row whose maximum sits 87.3 .. 88.7 above the first block's: found in round 6 by the widened spike test, a random query
    // against a spiked key).  Below 2^125 the reciprocal is a normal number with a factor 2 to spare for seq_scale.
    auto finite = [&]() __attribute__((always_inline)) {
      bool ok = true;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        float m0 = fmaxf(fmaxf(fabsf(o[x][0][0]), fabsf(o[x][0][1])), fmaxf(fabsf(o[x][0][2]), fabsf(o[x][0][3])));
        float m1 = fmaxf(fmaxf(fabsf(o[x][1][0]), fabsf(o[x][1][1])), fmaxf(fabsf(o[x][1][2]), fabsf(o[x][1][3])));
        const float sm = o[x][0][0] + o[x][0][1] + o[x][0][2] + o[x][0][3] + o[x][1][0] + o[x][1][1] + o[x][1][2] + o[x][1][3];     // NaN anywhere -> NaN
        ok = ok && (osum[x][0] < 4.0e37f) && (osum[x][0] > 0.f) && (fmaxf(m0, m1) < 3.0e38f) && (sm == sm);
      }
      return ok;
    };
    W4_STAMP(2);
    for (bool given = false;;) {                            // (one call site: the walks are inlined once)
      walk_m(given, !given);
      if (given || __builtin_expect(!__any(!finite()), 1)) break;
      // overflow against the first block's maxima: exact row maxima over the live blocks (run-time loop, cold), then the walk again.
      float mx[2] = {NEG_INF, NEG_INF};
      const bf16x8 cqf[2] = {qf[0], qf[1]};
      for (int c = 0; c < W4_NB; ++c) {
        if (!((m4 >> w4_block_class(c)) & 1)) continue;
        for (int u = 0; u < 2; ++u) {
          const int kt = 2 * c + u;
          const bf16x8 kfr = *reinterpret_cast<const bf16x8*>(Ksm + k_off_swz<HD>(r, g) + kt * 1024);
          for (int x = 0; x < 2; ++x) {
            const int ktc = kt < w3::NT ? kt : 0;
            const int a0k = w3_posA_rt(2 * ktc), stp = kt >= w3::NT - 1 ? 1 : w3_posA_rt(2 * ktc + 1) - a0k;
            const uint32_t bp = kt >= w3::NT ? tbpad : kt == w3::NT - 1 ? tb24[x] : (stp == 13 ? tb13[x] : tb1[x]) + (84 - a0k) * W4_ROWB;
            typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
            const f32x4 bias = *reinterpret_cast<lds_f32x4*>(bp);
            const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, cqf[x], bias, 0, 0, 0);
            mx[x] = fmaxf(mx[x], fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
          }
        }
      }
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        mx[x] = fmaxf(mx[x], __shfl_xor(mx[x], 16, 64));
        mx[x] = fmaxf(mx[x], __shfl_xor(mx[x], 32, 64));
        mref[x] = mx[x];
      }
      given = true;
    }
    // epilogue: request the next sequence's Q fragments, store, then wait for the fragments only (w4_need above).  Addresses are
    // rebuilt from the lane id here -- as loop invariants they get spilled, and every scratch reload is a vmcnt(0) in front of a store
    {
      // straight-line buffer stores (lanes of padding queries / the odd wave's second tile carry an out-of-range offset and are dropped
      // by the bounds check): with the stores in branches the compiler cannot count them and waits with vmcnt(0)
      const int r_ = r, g_ = g;
      u16* ob = reinterpret_cast<u16*>(p.out) + seq * L * p.ld_out + h * HD;
      float* lb = p.lse + (seq * heads + h) * L;
      const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(ob)), 0, __builtin_amdgcn_readfirstlane((int)(((L - 1) * p.ld_out + HD) * 2)), 0x00020000);
      const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(lb)), 0, L * 4, 0x00020000);
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int qq = (2 * wave + x) * 16 + r_;
        const bool ok = qq < L && (x == 0 || !odd);
        const float sum = osum[x][0];
        const float inv = seq_scale * __builtin_amdgcn_rcpf(sum);
        const uint32_t vo = ok ? (uint32_t)(qq * p.ld_out + g_ * 4) * 2u : 0x80000000u;       // (not 0xffffffff: + 32 below must not wrap into range)
        const uint32_t vl = (ok && g_ == 0) ? (uint32_t)qq * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_{pack_bf2(o[x][0][0] * inv, o[x][0][1] * inv), pack_bf2(o[x][0][2] * inv, o[x][0][3] * inv)}, ro, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_{pack_bf2(o[x][1][0] * inv, o[x][1][1] * inv), pack_bf2(o[x][1][2] * inv, o[x][1][3] * inv)}, ro, vo + 32u, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mref[x] + __builtin_amdgcn_logf(sum) * LN2), rl, vl, 0, 0);
      }
    }
    W4_STAMP(20);
    if (has_next) { w4_need(qn[0], qn[1]); qf[0] = qn[0]; qf[1] = qn[1]; }
    W4_STAMP(21);
  }
}

// ================================================================================================
// dK / dV.  The mirror of the forward: workgroup = (head, chunk of clips, residue class of window positions), 13 waves, wave w = key
// tiles 2w, 2w + 1 (wave 12: tile 24 alone, the NK = 1 walks) with their K / V fragments and the dK^T / dV^T accumulators in registers;
// it walks the LIVE query blocks (two query tiles = one 32-deep dK / dV MFMA) of the window type.  Q / dO images + lse / delta of the next
// sequence stream into the other LDS buffer by DMA.  Lane (r, g): key r of each tile, queries 4g..4g+3 of each query tile (S = Q K^T,
// rows = queries: P and dS feed the dV^T = dO^T P and dK^T = Q^T dS products as B operands straight from the accumulator layout).
// Per block: 12 b128 reads (Q / dO fragments, lse, delta, bias) -> 8 MFMAs (S, dP) -> chain (P = 2^(S log2 e - lse log2 e),
// dS = P (dP scale - delta)) -> 8 transposing reads (Q^T, dO^T) -> 8 MFMAs (dV^T, dK^T).  delta comes from the dQ kernel (launched first).
// ================================================================================================
constexpr int W4_LV = 2048;                                // bytes reserved per lse / delta image (two 1-KiB DMA requests)
constexpr int W4_BBUF = 2 * W4_IMG + 2 * W4_LV;            // one backward LDS buffer: Q image, dO image, lse, delta = 57 344 bytes

template <int N>
__device__ __forceinline__ void w4_wait6(bf16x8& a, bf16x8& b, f32x4& c, f32x4& d, f32x4& e0, f32x4& e1) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e0), "+v"(e1) : "i"(N));
}
template <int N>
__device__ __forceinline__ void w4_wait5(bf16x8& a, bf16x8& b, f32x4& c, f32x4& d, f32x4& e0) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e0) : "i"(N));
}
template <int N>
__device__ __forceinline__ void w4_wait12(bf16x8& a0, bf16x8& a1, bf16x8& b0, bf16x8& b1, f32x4& c0, f32x4& c1, f32x4& d0, f32x4& d1, f32x4& e0, f32x4& e1, f32x4& e2, f32x4& e3) {
  asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3) : "i"(N));
}
template <int N>
__device__ __forceinline__ void w4_wait10(bf16x8& a0, bf16x8& a1, bf16x8& b0, bf16x8& b1, f32x4& c0, f32x4& c1, f32x4& d0, f32x4& d1, f32x4& e0, f32x4& e1) {
  asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1), "+v"(e0), "+v"(e1) : "i"(N));
}
__device__ __forceinline__ void w4_wait_tr8(s16x4& a0, s16x4& a1, s16x4& c0, s16x4& c1, s16x4& e0, s16x4& e1, s16x4& f0, s16x4& f1) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1), "+v"(e0), "+v"(e1), "+v"(f0), "+v"(f1));
}

template <bool MASK>
__global__ __launch_bounds__(832) void attn_bwd_dkv_win4_kernel(const vmvm_attn_bwd_desc pb, const int nch, const int nwg) {
  constexpr int HD = 32, NWV = 13, IMG = W4_IMG, BUF = W4_BBUF;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = w3::L, heads = p.heads, nWin = p.n_win > 0 ? p.n_win : 1, B = p.nseq / nWin;
  const int logical = xcd_remap(blockIdx.x, heads * nch * nwg);
  const int wg = logical % nwg;
  const int ch = (logical / nwg) % nch;
  const int h = logical / (nwg * nch);
  unsigned char* TL = smem + 2 * BUF;
  {                                                       // windowed table of this head, consecutive QUERIES per read (staged through the second buffer)
    float* stage = reinterpret_cast<float*>(smem + BUF);
    for (int i = tid; i < 15 * 169; i += NWV * 64) stage[i] = p.bias_table[(size_t)i * heads + h];
    __syncthreads();
    w3_build_table<1>(TL, stage, tid, NWV * 64);
    __syncthreads();
  }
  const bool odd = !MASK && wave == NWV - 1;              // (the masked build runs wave 12 as a two-tile job with the all-padding tile 25: 18 walk variants spill where 9 do not)
  const int kt[2] = {2 * wave, odd ? 2 * wave : (2 * wave + 1 < w3::NT ? 2 * wave + 1 : w3::NT - 1)};
  const int kc = w3::cls_of(kt[0]);
  const int cper = (B + nch - 1) / nch;
  const int c0 = ch * cper, c1 = (c0 + cper < B) ? c0 + cper : B;
  const int ncl = c1 > c0 ? c1 - c0 : 0, total = ncl * ((nWin - wg + nwg - 1) / nwg);
  int w_nx = wg, c_nx = c0;
  auto seq_nx = [&]() { return (size_t)c_nx * nWin + w_nx; };
  auto advance = [&]() { if (++c_nx == c1) { c_nx = c0; w_nx += nwg; } };
  constexpr int NF = (W4_ROWS * 4) / (NWV * 64);          // 2 requests per thread per image
  const unsigned q_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2), do_bytes = (unsigned)(((size_t)(L - 1) * pb.ld_dout + HD) * 2);
  struct LaneAddr { uint32_t off_k[2], goq[NF], god[NF]; };
  auto lane_addr = [&](int ln) __attribute__((always_inline)) {
    LaneAddr a;
    const int r_ = ln & 15, g_ = ln >> 4, tid_ = wave * 64 + ln;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int key = kt[u] * 16 + r_;
      a.off_k[u] = (uint32_t)(key < L ? key : L - 1) * p.ld_qkv + h * HD + g_ * 8;     // padding keys read row L - 1 (finite, never stored)
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int u = i * NWV * 64 + tid_, row = u >> 2, chs = u & 3, cs = (chs ^ swz_chunk<32>(row)) << 3;
      a.goq[i] = (uint32_t)((row * p.ld_qkv + cs) * 2);
      a.god[i] = (uint32_t)((row * pb.ld_dout + cs) * 2);
    }
    return a;
  };
  // DMA requests of one sequence's images: step i < NF = image chunk i of Q and dO, step NF = lse (waves 0-1) / delta (waves 2-3)
  auto dma_step = [&](size_t seq, int buf, int i, const LaneAddr& a) __attribute__((always_inline)) {
    unsigned char* dst = smem + buf * BUF;
    if (i < NF) {
      const u16* qsrc = uniform_ptr(reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv + p.q_off + h * HD);
      const u16* dsrc = uniform_ptr(reinterpret_cast<const u16*>(pb.dout) + seq * L * pb.ld_dout + h * HD);
      dma16_pair(dst + wave * 1024 + i * NWV * 64 * 16, IMG, qsrc, q_bytes, a.goq[i], dsrc, do_bytes, a.god[i], false);
    } else if (wave < 4) {
      const float* src = uniform_ptr(((wave < 2) ? p.lse : pb.delta) + (seq * heads + h) * L);
      dma16_one(dst + 2 * IMG + (wave >> 1) * W4_LV + (wave & 1) * 1024, src, (unsigned)(L * 4), (uint32_t)(((wave & 1) * 64 + lane) * 16));
    }
  };
  bf16x8 kf[2], vf[2];
  auto fetch = [&](size_t seq, const LaneAddr& a, bf16x8 (&kd)[2], bf16x8 (&vd)[2]) __attribute__((always_inline)) {
    const u16* qb = reinterpret_cast<const u16*>(p.qkv) + seq * L * p.ld_qkv;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      kd[u] = *reinterpret_cast<const bf16x8*>(qb + p.k_off + a.off_k[u]);
      vd[u] = *reinterpret_cast<const bf16x8*>(qb + p.v_off + a.off_k[u]);
    }
  };
  if (total > 0) {
    const LaneAddr a0 = lane_addr(lane);
#pragma unroll
    for (int i = 0; i <= NF; ++i) dma_step(seq_nx(), 0, i, a0);
    fetch(seq_nx(), a0, kf, vf);
    advance();
    w4_need(kf[0], kf[1]);
    w4_need(vf[0], vf[1]);
    fill_wait();
  }

  int wprev = -1, w_cu = wg, c_cu = c0, m4 = 15;
  for (int b = 0; b < total; ++b) {
    const int cur = b & 1;
    const size_t seq = (size_t)c_cu * nWin + w_cu;
    W4_STAMP(0);
    __syncthreads();                                      // sequence b landed for everyone (each wave waited for its requests in the epilogue); the other buffer is free
    W4_STAMP(1);
    const int wcur = w_cu;
    if (++c_cu == c1) { c_cu = c0; w_cu += nwg; }
    if (MASK && wcur != wprev) {
      wprev = wcur;
      m4 = w3_live_rt(kc, __builtin_amdgcn_readfirstlane(w3_window_type(p.region, wcur)));
    }
    const float seq_scale = p.seq_scale ? p.seq_scale[(uint32_t)seq / (uint32_t)p.seqs_per_scale] : 1.0f;
    const bool has_next = b + 1 < total;
    const size_t seq_n = seq_nx();
    if (has_next) advance();
    int ln = lane;
    asm volatile("" : "+v"(ln));                            // opaque per sequence (see the forward)
    const int r = ln & 15, g = ln >> 4;
    const unsigned char* Qs = smem + cur * BUF;
    f32x4 dk[2][2], dv[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int d = 0; d < 2; ++d) { dk[u][d] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u][d] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    f32x2 ss2 = {seq_scale, seq_scale};
    asm volatile("" : "+v"(ss2));

    auto walk = [&](auto mc, auto nkc) __attribute__((always_inline)) {
      constexpr int M4 = decltype(mc)::value, NK = decltype(nkc)::value;
      constexpr w3::TileList QL = w3::list_all(M4);
      constexpr int NP = (QL.n + 1) / 2;
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      // (per-lane bases rebuilt here from an opaque lane id: as values live across the switch over the walks they get spilled)
      int lw = lane;
      asm volatile("" : "+v"(lw));
      const int r = lw & 15, g = lw >> 4;
      // lane bases into the table: row = A(q) - A(k) + 84, window s = 4 (g & 1) - (r & 7) + 7; A(q) = tile immediate + lq * step
      const int lk = r >> 3, lq = g >> 1, sw = 4 * (g & 1) - (r & 7) + 7;
      const uint32_t tl0 = lds_addr(TL) + sw * 16;
      const uint32_t tbpad = tl0 + 169 * W4_ROWB;
      uint32_t tb1[2], tb13[2], tb24[2];
  #pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ak = w3_posA_rt(2 * kt[u] + lk);
        const uint32_t bb = tl0 + (84 - ak) * W4_ROWB;
        tb1[u] = bb + lq * W4_ROWB;
        tb13[u] = bb + 13 * lq * W4_ROWB;
        tb24[u] = lq ? tbpad : bb + w3::tileA0(w3::NT - 1) * W4_ROWB;          // query tile 24: its second position is padding
      }
      const uint32_t qba = lds_addr(Qs + k_off_swz<HD>(r, g));                    // Q fragment of tile 0 (dO: + IMG); every tile an immediate away
      const uint32_t tq0a = lds_addr(Qs + k_off_swz<HD>(g * 4 + (r >> 2), (r & 3) >> 1) + (r & 1) * 8);
      const uint32_t tq1a = lds_addr(Qs + k_off_swz<HD>(g * 4 + (r >> 2), 2 + ((r & 3) >> 1)) + (r & 1) * 8);
      const uint32_t lsa = lds_addr(Qs + 2 * IMG) + g * 16;                       // lse of queries 4g..4g+3 of tile 0 (delta: + W4_LV)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int t = QL.t[2 * i];                           // query tiles t, t + 1 (t + 1 = 25: the all-padding tile)
        if (i < NF + 1 && has_next) { int l2 = lane; asm volatile("" : "+v"(l2)); dma_step(seq_n, cur ^ 1, i, lane_addr(l2)); }     // (addresses built here: not live across the walk)
        uint32_t pw[2][4], dw[2][4];
        // one query tile at a time (its 4 + NK reads, 2 NK MFMAs and chain): both tiles' fragments, scores and dP at once are 64 registers
        // next to the 48 of the accumulators and K / V fragments -- a spilling loop
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          const int qt = t + x;
          bf16x8 qf, dof;
          f32x4 l4, d4, s[2], dp[2];
          w4_read_frag(qf, qba, qt * 1024);
          w4_read_frag(dof, qba, IMG + qt * 1024);
          w4_read_b128(l4, lsa, qt * 64);
          w4_read_b128(d4, lsa, W4_LV + qt * 64);
#pragma unroll
          for (int u = 0; u < NK; ++u) {
            if (qt >= w3::NT) w4_read_b128(s[u], tbpad, 0);
            else if (qt == w3::NT - 1) w4_read_b128(s[u], tb24[u], 0);
            else if (w3::tileStep(qt < w3::NT ? qt : 0) == 13) w4_read_b128(s[u], tb13[u], w3::tileA0(qt < w3::NT ? qt : 0) * W4_ROWB);
            else w4_read_b128(s[u], tb1[u], w3::tileA0(qt < w3::NT ? qt : 0) * W4_ROWB);
          }
          if (NK == 2) w4_wait6<0>(qf, dof, l4, d4, s[0], s[1]); else w4_wait5<0>(qf, dof, l4, d4, s[0]);
#pragma unroll
          for (int u = 0; u < NK; ++u) {
            s[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, kf[u], s[u], 0, 0, 0);
            dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, vf[u], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          }
          const f32x2 nl[2] = {f32x2{l4[0], l4[1]} * f32x2{-LOG2E, -LOG2E}, f32x2{l4[2], l4[3]} * f32x2{-LOG2E, -LOG2E}};
#pragma unroll
          for (int u = 0; u < NK; ++u)
#pragma unroll
            for (int hj = 0; hj < 2; ++hj) {
              const f32x2 e = __builtin_elementwise_fma(f32x2{s[u][2 * hj], s[u][2 * hj + 1]}, f32x2{LOG2E, LOG2E}, nl[hj]);
              const f32x2 pr = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
              const f32x2 dd = pr * __builtin_elementwise_fma(f32x2{dp[u][2 * hj], dp[u][2 * hj + 1]}, ss2, -f32x2{d4[2 * hj], d4[2 * hj + 1]});
              pw[u][2 * x + hj] = pack_bf2v(pr);
              dw[u][2 * x + hj] = pack_bf2v(dd);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
        // (the transposed operands are requested BEHIND the chain: in front of it their 16 registers are the difference between 128 and a
        // spilling loop; the other waves of the SIMD cover the wait)
        s16x4 a0, a1, c0_, c1_, e0, e1, f0, f1;
        tr_read4(a0, a1, c0_, c1_, tq0a, tq1a, (t >> 1) * 2048);
        tr_read4(e0, e1, f0, f1, tq0a, tq1a, IMG + (t >> 1) * 2048);
        // the next sequence's K / V fragments go into the SAME registers, behind their last use (plain loads: they land under this
        // block's products and the epilogue; a second register set was spilled -- with a wait for the loads in front of the spill)
        if (i == NP - 1 && has_next) { int l2 = lane; asm volatile("" : "+v"(l2)); fetch(seq_n, lane_addr(l2), kf, vf); }
        w4_wait_tr8(a0, a1, c0_, c1_, e0, e1, f0, f1);
        const s16x8 q0v = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const s16x8 q1v = {c0_[0], c0_[1], c0_[2], c0_[3], c1_[0], c1_[1], c1_[2], c1_[3]};
        const s16x8 d0v = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
        const s16x8 d1v = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
#pragma unroll
        for (int u = 0; u < NK; ++u) {
          const bf16x8 pf = __builtin_bit_cast(bf16x8, make_uint4(pw[u][0], pw[u][1], pw[u][2], pw[u][3]));
          const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(dw[u][0], dw[u][1], dw[u][2], dw[u][3]));
          dv[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, d0v), pf, dv[u][0], 0, 0, 0);
          dv[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, d1v), pf, dv[u][1], 0, 0, 0);
          dk[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q0v), dsf, dk[u][0], 0, 0, 0);
          dk[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, q1v), dsf, dk[u][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);                  // (keep these products in front of the next block's reads: sunk behind them they hold 32 registers too many)
        W4_STAMP(3 + i);
      }
      if (has_next) {                                       // (short walks: the DMA steps the blocks did not cover)
#pragma unroll
        for (int i = NP; i <= NF; ++i) dma_step(seq_n, cur ^ 1, i, lane_addr(ln));
      }
    };
    auto walk_n = [&](auto nkc) __attribute__((always_inline)) {
      if (MASK) {
        switch (m4) {
          case 15: walk(IC4<15>{}, nkc); break;
          case 3: walk(IC4<3>{}, nkc); break;
          case 12: walk(IC4<12>{}, nkc); break;
          case 5: walk(IC4<5>{}, nkc); break;
          case 10: walk(IC4<10>{}, nkc); break;
          case 1: walk(IC4<1>{}, nkc); break;
          case 2: walk(IC4<2>{}, nkc); break;
          case 4: walk(IC4<4>{}, nkc); break;
          default: walk(IC4<8>{}, nkc); break;
        }
      } else {
        walk(IC4<15>{}, nkc);
      }
    };
    W4_STAMP(2);
    if (!MASK && odd) walk_n(IC4<1>{}); else walk_n(IC4<2>{});
    {
      // epilogue: straight-line buffer stores (padding keys / the odd wave's second tile: out-of-range offsets), then the wait for the
      // next sequence's fragments and DMA requests only (the compiler counts the stores)
      u16* db = reinterpret_cast<u16*>(pb.dqkv) + seq * L * pb.ld_dqkv + h * HD;
      const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(db)), 0, __builtin_amdgcn_readfirstlane((int)(((L - 1) * pb.ld_dqkv + 3 * heads * HD) * 2)), 0x00020000);
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int key = (2 * wave + u) * 16 + r;
        const bool ok = key < L && (u == 0 || !odd);
        const uint32_t vo = ok ? (uint32_t)(key * pb.ld_dqkv + g * 4) * 2u : 0x80000000u;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_{pack_bf2(dk[u][d][0], dk[u][d][1]), pack_bf2(dk[u][d][2], dk[u][d][3])}, rd, vo + (uint32_t)(p.k_off + d * 16) * 2u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(u32x2_{pack_bf2(dv[u][d][0] * seq_scale, dv[u][d][1] * seq_scale), pack_bf2(dv[u][d][2] * seq_scale, dv[u][d][3] * seq_scale)}, rd,
                                                vo + (uint32_t)(p.v_off + d * 16) * 2u, 0, 0);
        }
      }
    }
    W4_STAMP(20);
    if (has_next) {
      w4_need(kf[0], kf[1]);
      w4_need(vf[0], vf[1]);
    }
    W4_STAMP(21);
  }
}

template <typename K>
int w4_set_smem(K kernel, int bytes) {
  if (bytes > 160 * 1024) return VMVM_ENOSUPPORT;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return VMVM_EHIP;
  }
  return VMVM_OK;
}

// A workgroup takes a chunk of the CLIP range x a residue class of the window positions (w = wg mod nwg: interleaved, so it sees every
// window type and the masked tiles of edge / corner windows shorten all workgroups alike).  (nch, nwg) minimise whole rounds of
// `slots` concurrently resident workgroups x (sequences per workgroup + per-workgroup set-up worth `setup` sequences).
void w4_chunks(int base, int nclip, int nwin, float setup, int slots, int* nch_out, int* nwg_out) {
  int nch = 1, nwg = 1; float best = 1e30f;
  for (int g = 1; g <= nwin && g <= 16; g *= 2)
    for (int c = 1; c <= 64 && c <= nclip; ++c) {
      const float cost = (float)((base * c * g + slots - 1) / slots) * ((float)(((nclip + c - 1) / c) * ((nwin + g - 1) / g)) + setup);
      if (cost < best - 1e-6f) { best = cost; nch = c; nwg = g; }
    }
  *nch_out = nch; *nwg_out = nwg;
}

}  // namespace

namespace vmvm_w4 {

__attribute__((visibility("hidden"))) int launch_fwd(const vmvm_attn_fwd_desc* d, hipStream_t st) {
  const int smem = 4 * W4_IMG + W4_TAB_BYTES;
  const int nwin = d->n_win > 0 ? d->n_win : 1;
  int nch, nwg;
  w4_chunks(d->heads, d->nseq / nwin, nwin, 2.f, 256, &nch, &nwg);
  if (d->region) {
    int rc_ = w4_set_smem(attn_fwd_win4_kernel<true>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_fwd_win4_kernel<true>), dim3(d->heads * nch * nwg), dim3(832), smem, st, *d, nch, nwg);
  } else {
    int rc_ = w4_set_smem(attn_fwd_win4_kernel<false>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_fwd_win4_kernel<false>), dim3(d->heads * nch * nwg), dim3(832), smem, st, *d, nch, nwg);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

__attribute__((visibility("hidden"))) int launch_dkv(const vmvm_attn_bwd_desc* d, hipStream_t st) {
  const int smem = 2 * W4_BBUF + W4_TAB_BYTES;
  const int nwin = d->f.n_win > 0 ? d->f.n_win : 1;
  int nch, nwg;
  w4_chunks(d->f.heads, d->f.nseq / nwin, nwin, 2.f, 256, &nch, &nwg);
  if (d->f.region) {
    int rc_ = w4_set_smem(attn_bwd_dkv_win4_kernel<true>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_bwd_dkv_win4_kernel<true>), dim3(d->f.heads * nch * nwg), dim3(832), smem, st, *d, nch, nwg);
  } else {
    int rc_ = w4_set_smem(attn_bwd_dkv_win4_kernel<false>, smem);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_bwd_dkv_win4_kernel<false>), dim3(d->f.heads * nch * nwg), dim3(832), smem, st, *d, nch, nwg);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

}  // namespace vmvm_w4
