// NOT COMPILED INTO libvmvm (round 5 experiment, kept for the record: profiles/r05_fusion_dq_keyblocked_experiment.txt).  The kernel below is
// parity-green and 17 % faster than attn_bwd_dq_kernel alone, and changes nothing in the step (its scalar record loads are HBM misses
// there).  To run it: copy to pytorch_empirical_mvm_amd/csrc/attention_fus4.hip, add it to build.py SOURCES and restore the
// vmvm_f4::launch_dq dispatch in attention.hip (git show 718f769^:pytorch_empirical_mvm_amd/csrc/attention.hip | grep -n vmvm_f4).
// attention_fus4.hip -- fusion-encoder self-attention backward (BertSelfAttention of the cross-modal encoder, VIOLET_Base.go_cross
// model.py:183-189: L = 432 tokens, head_dim 64, additive key mask, attention-probability dropout) in the KEY-BLOCKED form of
// attention_win4.hip: small per-wave state, three waves per SIMD, two query tiles per wave against one key tile at a
// time, a lean per-element chain.  The round-2..4 kernels (attention.hip: attn_bwd_dq_kernel / attn_bwd_dkv_kernel) stay as the
// generic path (other L, seq2seq mask, Philox decisions without a record).
//
//  * dQ kernel: workgroup = one (sequence, head), 12 waves; wave w = query tiles 2w, 2w + 1 (24 tiles), waves 0..2 then tiles 24..26
//    alone -- SIMD loads 7 / 7 / 7 / 6 tiles.  K and V images (448 rows x 128 B, swizzled, DMA) in LDS; a wave keeps Q / dO fragments,
//    -lse log2 e and -delta of its tiles in registers and walks the 27 key tiles: S^T = K Q^T and dP^T = V dO^T (8 MFMAs per step for two
//    query tiles), chain P = 2^(S scale log2 e - lse log2 e), dS = P (keep-scaled dP under the stored dropout decision - delta) --
//    4 VALU instructions per element: one packed fma, one exp2, one v_cndmask (the forward's lane masks arrive as SCALAR pairs through
//    s_load: no vector instruction fetches them), half a packed fma, half a packed multiply, half a convert -- and after every second key
//    tile dQ^T += K^T dS^T with K^T read by transposing LDS reads (8 MFMAs).  Fragments of key tile t + 1 are requested before the MFMAs
//    of tile t (counted lgkmcnt waits, as in attention_win4.hip).  Key tiles that contain masked keys (a 27-bit word per sequence,
//    computed while the images land) add the 0 / -inf key bias from LDS on the way into the exponential; the others skip it.
//    delta = rowsum(dO * O) is computed in the prologue and stored for the dK / dV kernel, as in attn_bwd_dq_kernel.
#include "attn_common.h"
#include <cstdlib>

namespace {

template <int V> struct ICF { static constexpr int value = V; };

// Probe builds (-DF4_TIMELINE, tools/scratch/f4_timeline.py): s_memtime stamps of every wave of workgroup F4_TL_WG into the buffer passed as
// vmvm_attn_bwd_desc.dbias_table ([12 waves][64 stamps] u64; unused by the mode-1 kernels).  Production builds: nothing.
#ifdef F4_TIMELINE
#ifndef F4_TL_WG
#define F4_TL_WG 100
#endif
#define F4_TL_UNIT 3
#define F4_STAMP(idx) do { if (pb.dbias_table && blockIdx.x == F4_TL_WG && lane == 0) reinterpret_cast<unsigned long long*>(pb.dbias_table)[wave * 64 + (idx)] = __builtin_readcyclecounter(); } while (0)
#else
#define F4_TL_UNIT 0
#define F4_STAMP(idx) do { } while (0)
#endif

constexpr int F4_L = 432, F4_NT = 27;
constexpr int F4_NW = 8;
#ifndef F4_DB
#define F4_DB 1
#endif
#ifndef F4_TOUCH
#define F4_TOUCH 3
#endif
#ifndef F4_PRIO
#define F4_PRIO 0
#endif
#ifndef F4_ABL
#define F4_ABL 0                                    // probe builds: 1 no chain, 2 no score MFMAs, 4 no dQ MFMAs, 8 fragments read once (results wrong)
#endif
// LDS map of the backward kernels (bytes).  K and V images: 27 tiles x 2 KiB each, no padding tile (the walk's last 32-key block
// re-reads tile 26 for its missing half, whose dS half is zero).  Region A = tiles 0..13 of both images, region B = tiles 14..26:
// the two halves of the rolling refill (a region of the NEXT unit is requested as soon as every wave has left it).
constexpr int F4_IMG = F4_NT * 2048;               // 55 296
constexpr int F4_TA = 14;                          // key tiles of region A
constexpr int F4_KB = 2 * F4_IMG;                  // float kb[2][448]: additive key mask (0 / -inf) of the current / next unit
constexpr int F4_FLG = F4_KB + 2 * 448 * 4;        // int flags[2][8]: masked-key nibbles per 64 keys
constexpr int F4_DUMP = F4_FLG + 64;               // 256 bytes nobody reads: destination of the L2 touch requests
constexpr int F4_SLOT = F4_DUMP + 256;             // 6 f32 partial-dQ slots of the three tail tiles (4 KiB each)
constexpr int F4_SMEM_DQ = F4_SLOT + 6 * 4096;     // 139 072
static_assert(F4_SMEM_DQ <= 160 * 1024 && (F4_SLOT & 15) == 0, "LDS map");

typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(8))) short s16x8f;

__device__ __forceinline__ void f4_read_frag(bf16x8& d, uint32_t addr, const int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(off) : "memory");
}
__device__ __forceinline__ void f4_read_tr(s16x4& d, uint32_t addr, const int off) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(off) : "memory");
}
// counted waits (LDS operations of a wave return in order): "all but the youngest N have landed", naming the registers they make valid
template <int N>
__device__ __forceinline__ void f4_wait4(bf16x8& a, bf16x8& b, bf16x8& c, bf16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}
template <int N>
__device__ __forceinline__ void f4_wait_tr8(s16x4& a0, s16x4& a1, s16x4& b0, s16x4& b1, s16x4& c0, s16x4& c1, s16x4& d0, s16x4& d1) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(d0), "+v"(d1) : "i"(N));
}
// the eight dwords of one (query tile, key tile) dropout record as scalars (the pointer is wave-uniform)
__device__ __forceinline__ u32x8 f4_load_masks(const uint32_t* p) {
  typedef const __attribute__((address_space(4))) u32x8 c_u32x8;
  return *reinterpret_cast<c_u32x8*>(reinterpret_cast<uintptr_t>(p));
}
__device__ __forceinline__ void f4_use_masks(u32x8& m) { asm volatile("" : "+s"(m)); }       // the compiler's wait for the s_load lands here

// a * {s, s} + c with the wave-uniform factor in a SCALAR pair (as a VGPR pair it is two registers per factor the walks cannot spare)
__device__ __forceinline__ f32x2 f4_pk_fma_s(const f32x2 a, const uint64_t s, const f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(s), "v"(c));
  return d;
}
__device__ __forceinline__ uint64_t f4_pair(float x) {
  const uint32_t b = __builtin_amdgcn_readfirstlane(__float_as_uint(x));
  return ((uint64_t)b << 32) | b;
}
__device__ __forceinline__ void f4_slot_write(uint32_t addr, const f32x4& v, const int off) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "i"(off) : "memory");
}
__device__ __forceinline__ void f4_slot_read(f32x4& v, uint32_t addr, const int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(off) : "memory");
}
__device__ __forceinline__ void f4_lds_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void f4_wait_f4(f32x4& a, f32x4& b, f32x4& c, f32x4& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
// every vector-memory request of this wave has completed (DMA landed, loads returned), then the workgroup barrier
__device__ __forceinline__ void f4_sync_all() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ================================================================================================
// dQ (+ delta).  Persistent: workgroup b walks the units b, b + G, ... (unit = (sequence, head)); per unit two phases (key tiles 0..13,
// 14..26), each = main walk of the wave's two query tiles + its piece of the three tail tiles; barriers: P1 (region A landed, slots
// free) / X (everyone left region A, region B landed) / Y (tail partials complete).  DMA requests: region A of the next unit after X,
// region B of the current unit after P1 -- each has a whole phase to land.
// ================================================================================================
// what a wave holds of its NQ query tiles
template <int NQ>
struct F4Q {
  bf16x8 qf[NQ][2], dof[NQ][2];
  f32x2 nl2[NQ], ndl[NQ];
  const uint32_t* mrow[NQ];
};
struct F4U {                                          // one unit's pointers (wave-uniform)
  const u16 *qkv, *dO, *O;
  const float* lse;
  float* delta;
  u16* dq;
  const uint32_t* mrec;
  const uint8_t* km;
  float seq_scale;
};

template <int DROPM>                                  // 0: no dropout, 2: the forward's stored decisions (vmvm_attn_fwd_desc.drop_mask)
__global__ __launch_bounds__(F4_NW * 64) void attn_bwd_dq_fus4_kernel(const vmvm_attn_bwd_desc pb, const int G) {
  constexpr int HD = 64, L = F4_L, NT = F4_NT, TA = F4_TA;
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const vmvm_attn_fwd_desc& p = pb.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int heads = p.heads, total = p.nseq * heads;
  const int first = xcd_remap(blockIdx.x, G);
  if (first >= total) return;

  // tail (waves 0..5): query tile 24 + wave % 3, piece wave / 3 of each phase's key tiles: A [0,8) [8,14), B [14,20) [20,27)
  const int ti = wave % 3, pj = wave / 3;
  const bool has_tail = wave < 6;

  auto unit_of = [&](const int u) __attribute__((always_inline)) {
    const int seq = u / heads, h = u - seq * heads;
    F4U o;
    o.qkv = uniform_ptr(reinterpret_cast<const u16*>(p.qkv) + (size_t)seq * L * p.ld_qkv + h * HD);
    o.dO = uniform_ptr(reinterpret_cast<const u16*>(pb.dout) + (size_t)seq * L * pb.ld_dout + h * HD);
    o.O = uniform_ptr(reinterpret_cast<const u16*>(p.out) + (size_t)seq * L * p.ld_out + h * HD);
    o.lse = uniform_ptr(p.lse + (size_t)u * L);
    o.delta = const_cast<float*>(uniform_ptr(pb.delta + (size_t)u * L));
    o.dq = const_cast<u16*>(uniform_ptr(reinterpret_cast<u16*>(pb.dqkv) + (size_t)seq * L * pb.ld_dqkv + p.q_off + h * HD));
    o.mrec = DROPM == 2 ? uniform_ptr(p.drop_mask + (size_t)u * NT * NT * 8) : nullptr;
    o.km = p.keymask ? uniform_ptr(p.keymask + (size_t)seq * L) : nullptr;
    o.seq_scale = p.seq_scale ? p.seq_scale[seq / p.seqs_per_scale] : 1.0f;
    return o;
  };
  // DMA of one region of a unit's K and V images: 1-KiB wave requests (8 rows x 128 B; the chunk swizzle (row & 7) = lane >> 3 is a lane
  // constant), request j of the region's 2 n to wave j mod 12
  const unsigned img_bytes = (unsigned)(((size_t)(L - 1) * p.ld_qkv + HD) * 2);
  auto dma_region = [&](const F4U& un, const int region) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) void lds_void;
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(un.qkv + p.k_off)), 0, __builtin_amdgcn_readfirstlane((int)img_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(un.qkv + p.v_off)), 0, __builtin_amdgcn_readfirstlane((int)img_bytes), 0x00020000);
    int ln = lane;
    asm volatile("" : "+v"(ln));                         // (per-lane constants are rebuilt at their use: held through the walks they spill)
    const uint32_t lane_goff = (uint32_t)(((ln >> 3) * p.ld_qkv + (((ln & 7) ^ (ln >> 3)) << 3)) * 2);
    const int n = region ? (NT - TA) * 2 : TA * 2, i0 = region ? TA * 2 : 0;
    for (int j = wave; j < 2 * n; j += F4_NW) {
      const bool isv = j >= n;
      const int i = i0 + (isv ? j - n : j);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isv ? rv : rk, (lds_void*)(smem + (isv ? F4_IMG : 0) + i * 1024), 16, lane_goff, i * 16 * p.ld_qkv, 0, 0);
    }
  };
  // L2 touches: one dword of every 128-byte line a later load of this wave will want, requested as a DMA into a 256-byte LDS area nobody
  // reads (no register is held, nothing waits for it).  A global load issued where its value is needed costs 6-8 k cycles under load
  // (tools/scratch/f4_timeline.py); behind a touch half a phase earlier it finds its line in L2.
  // touch_rows: the Q / dO / O rows of the wave's three main tiles and its tail tile of unit `un`: 64 rows = 64 lanes per tensor.
  auto touch_rows = [&](const F4U& t_un) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) void lds_void;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int q = ln < 48 ? 48 * wave + ln : (24 + ti) * 16 + (ln - 48);
    lds_void* dump = (lds_void*)(smem + F4_DUMP);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(t_un.qkv + p.q_off)), 0, __builtin_amdgcn_readfirstlane((int)img_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(t_un.dO)), 0, __builtin_amdgcn_readfirstlane((int)(((L - 1) * pb.ld_dout + HD) * 2)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(t_un.O)), 0, __builtin_amdgcn_readfirstlane((int)(((L - 1) * p.ld_out + HD) * 2)), 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, dump, 4, (uint32_t)(q * p.ld_qkv * 2), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, dump, 4, (uint32_t)(q * pb.ld_dout * 2), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ro, dump, 4, (uint32_t)(q * p.ld_out * 2), 0, 0, 0);
  };
  // touch_masks: the dropout records of those four query tiles for the key tiles of one phase (448 of a tile row's 864 bytes: 4 lines)
  auto touch_masks = [&](const F4U& t_un, const int phase) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) void lds_void;
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int row = ln >> 2, qt = row < 3 ? 3 * wave + row : 24 + ti;
    const uint32_t off = ln < 16 ? (uint32_t)(qt * (NT * 32) + phase * (TA * 32) + (ln & 3) * 128) : 0x80000000u;      // (lanes >= 16: out of range, no access)
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(uniform_ptr(t_un.mrec)), 0, NT * NT * 32, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (lds_void*)(smem + F4_DUMP), 4, off, 0, 0, 0);
  };
  // key-mask bias + "tile has masked keys" nibbles into copy `cp` from the lane's key-mask byte (tid = key; plain LDS stores: call with no
  // DMA of this wave in flight)
  auto kb_write = [&](const uint8_t km, const int cp) __attribute__((always_inline)) {
    float* kb = reinterpret_cast<float*>(smem + F4_KB) + cp * 448;
    int* flg = reinterpret_cast<int*>(smem + F4_FLG) + cp * 8;
    if (tid < 448) {
      const bool on = km != 0;
      kb[tid] = on ? 0.f : NEG_INF;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(!on);
      int nib = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) nib |= (((b >> (16 * q)) & 0xffffull) != 0ull) ? (1 << q) : 0;
      if (lane == 0) flg[wave] = nib;
    }
  };
  auto read_flags = [&](const int cp) __attribute__((always_inline)) {
    const int* flg = reinterpret_cast<const int*>(smem + F4_FLG) + cp * 8;
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) w |= (uint32_t)flg[i] << (4 * i);
    return (uint32_t)__builtin_amdgcn_readfirstlane(w);
  };
  // Q / dO / O fragments, delta, lse of a wave's tiles.  `store`: this wave writes delta (a tail tile is shared by four waves)
  auto prologue = [&](auto nqc, const F4U& un, const int qt0, const bool store) __attribute__((always_inline)) {
    constexpr int NQ = decltype(nqc)::value;
    F4Q<NQ> w;
#pragma unroll
    for (int x = 0; x < NQ; ++x) {
      const int q = (qt0 + x) * 16 + r;
      w.mrow[x] = DROPM == 2 ? un.mrec + (size_t)(qt0 + x) * NT * 8 : nullptr;
      float dl = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        w.qf[x][s] = *reinterpret_cast<const bf16x8*>(un.qkv + (size_t)q * p.ld_qkv + p.q_off + g * 8 + s * 32);
        w.dof[x][s] = *reinterpret_cast<const bf16x8*>(un.dO + (size_t)q * pb.ld_dout + g * 8 + s * 32);
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(un.O + (size_t)q * p.ld_out + g * 8 + s * 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)w.dof[x][s][e] * (float)of[e];
      }
      dl += __shfl_xor(dl, 16, 64);
      dl += __shfl_xor(dl, 32, 64);                     // delta_q = sum_d dO[q, d] * out[q, d]
      if (g == 0 && store) un.delta[q] = dl;
      const float l2 = un.lse[q] * LOG2E;
      w.nl2[x] = f32x2{-l2, -l2};
      w.ndl[x] = f32x2{-dl, -dl};
    }
    return w;
  };

  int k = 0;                                          // index of the unit this workgroup is on
  const uint32_t thr16 = drop_thr16(p.dropout_p);
  const float keep = DROPM ? 65536.f / (65536.f - (float)thr16) : 1.f;
  const uint64_t sc2s = f4_pair(p.scale * LOG2E);

  // lane bases into the images.  Plain fragments: row r of a tile, 16-byte chunk (4 s + g) ^ (r & 7) for head-dim half s (the XOR with 4 s
  // is not an add: one base per half).  Transposed fragments (frag_tokens, attn_common.h): row 4 g + r / 4 of a 32-key block (+ 16 for
  // its second tile), chunk (2 dt + (r & 3) / 2) ^ (row & 7), byte 8 (r & 1): one base per 16-wide head-dim tile dt.
  const uint32_t k0 = lds_addr(smem);
  uint32_t ka[2], va[2], kt[4];
#pragma unroll
  for (int s = 0; s < 2; ++s) { ka[s] = k0 + k_off_swz<HD>(r, s * 4 + g); va[s] = ka[s] + F4_IMG; }
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) kt[dt] = k0 + k_off_swz<HD>(g * 4 + (r >> 2), dt * 2 + ((r & 3) >> 1)) + (r & 1) * 8;

  // the walk of NQ query tiles over the key tiles [T0, T1) (T0 even): dq += their share of dS K
  auto walk = [&](auto nqc, auto t0c, auto t1c, const auto& w, auto& dq, const uint32_t flags, const int kcp, const uint64_t cdks) __attribute__((always_inline)) {
    constexpr int NQ = decltype(nqc)::value, T0 = decltype(t0c)::value, T1 = decltype(t1c)::value;
    static_assert(!(T0 & 1), "a walk starts on a 32-key block");
    // the tail pieces (NQ = 1) run while the main job's accumulators and fragments are live: ONE fragment buffer (the next tile's
    // fragments are requested behind the score MFMAs instead of in front of them) keeps them out of scratch
    constexpr bool DB = F4_DB && NQ > 1;
    bf16x8 kf[2][2], vf[2][2];
    s16x4 tr[4][2];
    u32x8 mk[2][NQ];                                   // dropout records of key tile t / t + 1 (requested a whole step ahead: an s_load that misses the scalar cache takes about one)
    uint32_t dsh[NQ][2][2];
    auto issue_frags = [&](const int t, bf16x8 (&kd)[2], bf16x8 (&vd)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < 2; ++s) f4_read_frag(kd[s], ka[s], t * 2048);
#pragma unroll
      for (int s = 0; s < 2; ++s) f4_read_frag(vd[s], va[s], t * 2048);
    };
    issue_frags(T0, kf[0], vf[0]);
    if (DROPM == 2) {
#pragma unroll
      for (int x = 0; x < NQ; ++x) mk[0][x] = f4_load_masks(w.mrow[x] + T0 * 8);
    }
#pragma unroll
    for (int t = T0; t < T1; ++t) {
      const int cb = DB ? (t & 1) : 0, hb = t & 1, c = t >> 1;
      if (DB && (t & 1)) f4_wait4<8>(kf[cb][0], kf[cb][1], vf[cb][0], vf[cb][1]);          // (the block's 8 transposing reads are younger; one buffer: older)
      else f4_wait4<0>(kf[cb][0], kf[cb][1], vf[cb][0], vf[cb][1]);
      if (DROPM == 2) {
#pragma unroll
        for (int x = 0; x < NQ; ++x) f4_use_masks(mk[(t - T0) & 1][x]);
        if (t + 1 < T1) {
#pragma unroll
          for (int x = 0; x < NQ; ++x) mk[(t - T0 + 1) & 1][x] = f4_load_masks(w.mrow[x] + (t + 1) * 8);
        }
      }
      if (DB && t + 1 < T1 && !((F4_ABL & 8) && t > T0)) issue_frags(t + 1, kf[cb ^ 1], vf[cb ^ 1]);
      if (!(t & 1)) {
        // (the last block's second tile does not exist: its dS half is zero, its K^T half re-reads tile 26)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { f4_read_tr(tr[dt][0], kt[dt], c * 4096); f4_read_tr(tr[dt][1], kt[dt], c * 4096 + (t == NT - 1 ? 0 : 2048)); }
      }
      f32x4 s4[NQ], dp4[NQ];
      if (F4_PRIO) __builtin_amdgcn_s_setprio(1);          // (MFMA bursts first: the matrix pipe then works under the other wave's chain)
      if (F4_ABL & 2) {
#pragma unroll
        for (int x = 0; x < NQ; ++x) { s4[x] = __builtin_bit_cast(f32x4, kf[cb][0]); dp4[x] = __builtin_bit_cast(f32x4, vf[cb][1]); }
      } else
#pragma unroll
      for (int x = 0; x < NQ; ++x) {
        s4[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cb][0], w.qf[x][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        s4[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[cb][1], w.qf[x][1], s4[x], 0, 0, 0);
        dp4[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[cb][0], w.dof[x][0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dp4[x] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[cb][1], w.dof[x][1], dp4[x], 0, 0, 0);
      }
      if (F4_PRIO) __builtin_amdgcn_s_setprio(0);
      if (!DB && t + 1 < T1) {
        asm volatile("" :: "v"(s4[0]), "v"(dp4[0]));       // (behind the MFMAs that read the buffer)
        issue_frags(t + 1, kf[0], vf[0]);
      }
      auto chain = [&](const f32x2 b01, const f32x2 b23, const bool biased) __attribute__((always_inline)) {
#pragma unroll
        for (int x = 0; x < NQ; ++x) {
          const f32x2 a01 = biased ? w.nl2[x] + b01 : w.nl2[x], a23 = biased ? w.nl2[x] + b23 : w.nl2[x];
          const f32x2 x01 = f4_pk_fma_s(f32x2{s4[x][0], s4[x][1]}, sc2s, a01);
          const f32x2 x23 = f4_pk_fma_s(f32x2{s4[x][2], s4[x][3]}, sc2s, a23);
          const f32x2 p01 = f32x2{__builtin_amdgcn_exp2f(x01[0]), __builtin_amdgcn_exp2f(x01[1])};
          const f32x2 p23 = f32x2{__builtin_amdgcn_exp2f(x23[0]), __builtin_amdgcn_exp2f(x23[1])};
          float dp[4] = {dp4[x][0], dp4[x][1], dp4[x][2], dp4[x][3]};
          if (DROPM == 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dp[j] = zero_where(dp[j], ((uint64_t)mk[(t - T0) & 1][x][2 * j + 1] << 32) | mk[(t - T0) & 1][x][2 * j]);
          }
          const f32x2 d01 = p01 * f4_pk_fma_s(f32x2{dp[0], dp[1]}, cdks, w.ndl[x]);
          const f32x2 d23 = p23 * f4_pk_fma_s(f32x2{dp[2], dp[3]}, cdks, w.ndl[x]);
          dsh[x][hb][0] = pack_bf2v(d01);
          dsh[x][hb][1] = pack_bf2v(d23);
        }
      };
      if (F4_ABL & 1) {
#pragma unroll
        for (int x = 0; x < NQ; ++x) { dsh[x][hb][0] = __float_as_uint(s4[x][0]) ^ __float_as_uint(dp4[x][1]); dsh[x][hb][1] = __float_as_uint(s4[x][2]) ^ __float_as_uint(dp4[x][3]); }
      } else
      if ((flags >> t) & 1) {                            // (cold: a key tile with masked keys)
        typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
        int ln = lane;
        asm volatile("" : "+v"(ln));                     // (the address is built here: not a register held through the walk)
        const f32x4 b4 = *reinterpret_cast<lds_f32x4*>(k0 + F4_KB + kcp * 448 * 4 + (ln >> 4) * 16 + t * 64);
        chain(f32x2{b4[0], b4[1]}, f32x2{b4[2], b4[3]}, true);
      } else {
        chain(f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, false);
      }
      if ((t & 1) || t == T1 - 1) {
        if (!(t & 1)) {
#pragma unroll
          for (int x = 0; x < NQ; ++x) { dsh[x][1][0] = 0u; dsh[x][1][1] = 0u; }
        }
        if (t + 1 < T1) f4_wait_tr8<4>(tr[0][0], tr[0][1], tr[1][0], tr[1][1], tr[2][0], tr[2][1], tr[3][0], tr[3][1]);
        else f4_wait_tr8<0>(tr[0][0], tr[0][1], tr[1][0], tr[1][1], tr[2][0], tr[2][1], tr[3][0], tr[3][1]);
        if (F4_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x8f kv = {tr[dt][0][0], tr[dt][0][1], tr[dt][0][2], tr[dt][0][3], tr[dt][1][0], tr[dt][1][1], tr[dt][1][2], tr[dt][1][3]};
#pragma unroll
          for (int x = 0; x < NQ; ++x) {
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, make_uint4(dsh[x][0][0], dsh[x][0][1], dsh[x][1][0], dsh[x][1][1]));
            if (F4_ABL & 4) { dq[x][dt][0] += __uint_as_float(dsh[x][0][0] ^ (uint32_t)kv[dt & 3]); continue; }
            dq[x][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kv), dsf, dq[x][dt], 0, 0, 0);
          }
        }
        if (F4_PRIO) __builtin_amdgcn_s_setprio(0);
      }
      if (k == F4_TL_UNIT) F4_STAMP((NQ > 1 ? 3 : 33) + t);
    }
  };
  auto store_dq = [&](const F4U& un, const int qt, const f32x4 (&d)[4]) __attribute__((always_inline)) {
    u16* dqp = un.dq + (size_t)(qt * 16 + r) * pb.ld_dqkv + g * 4;
    const float sc = p.scale;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<uint2*>(dqp + dt * 16) = make_uint2(pack_bf2(d[dt][0] * sc, d[dt][1] * sc), pack_bf2(d[dt][2] * sc, d[dt][3] * sc));
  };


  auto slot_addr = [&]() __attribute__((always_inline)) {      // this wave's partial; its tile's two: + 0, + 3 * 4096 from piece 0's
    int ln = lane;
    asm volatile("" : "+v"(ln));
    return (uint32_t)(k0 + F4_SLOT + ((pj * 3 + ti) * 4) * 1024 + ln * 16);
  };

  // Order inside a unit (measured, tools/scratch/f4_timeline.py: with the tail pieces BEHIND the main walks their fragment loads cost 6-13 k
  // cycles per phase, nothing to hide them behind): every global load is issued in front of a barrier it has to cross anyway --
  // main + tail fragments of the next unit before P1, the tail fragments again (registers) and the next unit's key mask before X --
  // and each phase starts with its tail piece.
  F4U un = unit_of(first);
  kb_write(un.km ? (tid < L ? un.km[tid] : (uint8_t)1) : (uint8_t)1, 0);
  dma_region(un, 0);
  dma_region(un, 1);

  F4Q<3> wq = prologue(ICF<3>{}, un, 3 * wave, true);
  F4Q<1> wt = prologue(ICF<1>{}, un, 24 + ti, has_tail && pj == 0);
  for (;; ++k) {
    const int u_next = first + (k + 1) * G;
    const bool has_next = u_next < total;
    if (k == F4_TL_UNIT) F4_STAMP(0);
    f4_sync_all();                                    // P1: region A of this unit landed everywhere, the tail slots are free
    if (k == F4_TL_UNIT) F4_STAMP(1);
    const uint32_t flags = read_flags(k & 1);
    const int kcp = k & 1;
    const uint64_t cdks = f4_pair(un.seq_scale * keep);
    if (has_tail) {
      f32x4 d1[1][4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) d1[0][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (pj == 0) walk(ICF<1>{}, ICF<0>{}, ICF<8>{}, wt, d1, flags, kcp, cdks);
      else walk(ICF<1>{}, ICF<8>{}, ICF<TA>{}, wt, d1, flags, kcp, cdks);
      const uint32_t slot = slot_addr();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) f4_slot_write(slot, d1[0][dt], dt * 1024);
    }
    // DMA requests are issued BEHIND the tail pieces: those run beside live state of the main job and the compiler reloads some of it
    // from scratch there -- a scratch load behind DMA requests waits for them to land (vmcnt is in order; measured: 2-5 k cycles)
    if (k > 0) dma_region(un, 1);                     // (region B: free since the previous unit's last barrier)
    if (DROPM == 2 && (F4_TOUCH & 2)) touch_masks(un, 1);
    f32x4 dq[3][4];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[x][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k == F4_TL_UNIT) F4_STAMP(2);
    walk(ICF<3>{}, ICF<0>{}, ICF<TA>{}, wq, dq, flags, kcp, cdks);
    if (k == F4_TL_UNIT) F4_STAMP(30);
    F4U nx = un;
    uint8_t km_nx = 1;
    if (has_next) {
      nx = unit_of(u_next);
      if (nx.km && tid < L) km_nx = nx.km[tid];
    }
    wt = prologue(ICF<1>{}, un, 24 + ti, false);
    f4_sync_all();                                    // X: every wave has left region A; region B landed everywhere
    if (k == F4_TL_UNIT) F4_STAMP(31);
    if (has_next) kb_write(km_nx, (k + 1) & 1);
    if (has_tail) {
      f32x4 d1[1][4];
      const uint32_t slot = slot_addr();
      f4_slot_read(d1[0][0], slot, 0); f4_slot_read(d1[0][1], slot, 1024); f4_slot_read(d1[0][2], slot, 2048); f4_slot_read(d1[0][3], slot, 3072);
      f4_wait_f4(d1[0][0], d1[0][1], d1[0][2], d1[0][3]);
      if (pj == 0) walk(ICF<1>{}, ICF<TA>{}, ICF<20>{}, wt, d1, flags, kcp, cdks);
      else walk(ICF<1>{}, ICF<20>{}, ICF<NT>{}, wt, d1, flags, kcp, cdks);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) f4_slot_write(slot, d1[0][dt], dt * 1024);
    }
    if (has_next) {
      dma_region(nx, 0);
      if (F4_TOUCH & 1) touch_rows(nx);
      if (DROPM == 2 && (F4_TOUCH & 2)) touch_masks(nx, 0);
    }
    if (k == F4_TL_UNIT) F4_STAMP(32);
    walk(ICF<3>{}, ICF<TA>{}, ICF<NT>{}, wq, dq, flags, kcp, cdks);
    store_dq(un, 3 * wave, dq[0]);
    store_dq(un, 3 * wave + 1, dq[1]);
    store_dq(un, 3 * wave + 2, dq[2]);
    if (k == F4_TL_UNIT) F4_STAMP(62);
    f4_lds_done();
    __builtin_amdgcn_s_barrier();                     // Y: the tail partials are complete
    if (has_tail && pj == 0) {                        // fixed order: deterministic
      f32x4 acc[4], v[4];
      const uint32_t slot = slot_addr();
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) f4_slot_read(acc[dt], slot, dt * 1024);
      f4_wait_f4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) f4_slot_read(v[dt], slot, 3 * 4096 + dt * 1024);
      f4_wait_f4(v[0], v[1], v[2], v[3]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[dt] += v[dt];
      store_dq(un, 24 + ti, acc);
    }
    if (k == F4_TL_UNIT) F4_STAMP(63);
    if (!has_next) break;
    un = nx;
    wq = prologue(ICF<3>{}, un, 3 * wave, true);
    wt = prologue(ICF<1>{}, un, 24 + ti, has_tail && pj == 0);
  }
}

template <typename K>
int f4_set_smem(K kernel, int bytes) {
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return VMVM_EHIP;
  }
  return VMVM_OK;
}

}  // namespace

namespace vmvm_f4 {

// the problems of the key-blocked fusion kernels: L = 432, head_dim 64, key mask only, dropout off or recorded (attention.hip: drop_mask_ok)
__attribute__((visibility("hidden"))) bool applicable_bwd(const vmvm_attn_bwd_desc* d) {
  const vmvm_attn_fwd_desc& f = d->f;
  if (f.mode != 1 || f.head_dim != 64 || f.L != F4_L || f.causal_from > 0 || f.att_colsum) return false;
  if (f.stream_min_len > 0 && f.L >= f.stream_min_len) return false;
  if (f.dropout_p > 0.f && !f.drop_mask) return false;
  return true;
}

__attribute__((visibility("hidden"))) int launch_dq(const vmvm_attn_bwd_desc* d, hipStream_t st) {
  const int total = d->f.nseq * d->f.heads;
  const int nb = total < 256 ? ((total + 7) & ~7) : 256;      // persistent: one workgroup per CU (the LDS images fill it); a multiple of 8 for the XCD map
  if (d->f.dropout_p > 0.f) {
    int rc_ = f4_set_smem(attn_bwd_dq_fus4_kernel<2>, F4_SMEM_DQ);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_bwd_dq_fus4_kernel<2>), dim3(nb), dim3(F4_NW * 64), F4_SMEM_DQ, st, *d, nb);
  } else {
    int rc_ = f4_set_smem(attn_bwd_dq_fus4_kernel<0>, F4_SMEM_DQ);
    if (rc_) return rc_;
    hipLaunchKernelGGL((attn_bwd_dq_fus4_kernel<0>), dim3(nb), dim3(F4_NW * 64), F4_SMEM_DQ, st, *d, nb);
  }
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

}  // namespace vmvm_f4
