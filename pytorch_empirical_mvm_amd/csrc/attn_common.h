// attn_common.h -- device helpers shared by the attention translation units (attention.hip, attention_win3.hip): LDS image swizzles,
// direct-to-LDS DMA fills, fragment reads (plain and transposing), the bias-through-MFMA block, Philox dropout blocks.
#pragma once
#include "common.h"
#include <type_traits>

namespace {


constexpr float NEG_INF = -__builtin_huge_valf();
// Bias blocks that enter the score through an MFMA (bias_block_mfma): out-of-range keys carry a large FINITE negative value -- the
// identity product multiplies every entry by 0 or 1, and 0 * -inf would be NaN.  exp2 of it is exactly 0 all the same.
constexpr float PAD_BIAS = -1.0e30f;
// One (16 x 16) bias + mask block, kept as packed bf16 in two registers per lane, added to a score accumulator by the matrix core:
// D = I(16x16) * Bias + C on v_mfma_f32_16x16x16_bf16.  The two registers ARE the B operand of that instruction (lane (col r, group g)
// holds rows 4g..4g+3 of column r -- the accumulator layout of the score MFMA), the A operand is the identity (1.0 where the lane's
// row equals one of its four k slots).  It replaces four VALU unpack instructions per block in kernels whose VALU pipe is the bound
// while the matrix pipe idles; exact (1.0 * b accumulates in f32).
// HAZARD (measured, tools/probe/bias_mfma_probe.hip): hipcc 7.2 emits NO wait states between v_mfma_f32_16x16x16_bf16 and a
// v_mfma_f32_16x16x32_bf16 that reads its result as SrcC (or the reverse), and the hardware does not forward between the two
// instruction types -- the consumer reads two stale registers.  With one independent MFMA between producer and consumer the result
// is correct.  Every use below therefore issues the bias products EARLY (a whole pass / tile pair ahead of their consumers) and pins
// that order with sched_barrier.
typedef __attribute__((ext_vector_type(4))) short s16x4_;
__device__ __forceinline__ s16x4_ bias_ident_frag(int lane) {
  const int r = lane & 15, g = lane >> 4;
  s16x4_ a;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = (4 * g + j == r) ? (short)0x3f80 : (short)0;      // bf16 1.0
  return a;
}
__device__ __forceinline__ f32x4 bias_block_mfma(const s16x4_& ident, uint32_t w0, uint32_t w1, const f32x4& c) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_;
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ident, __builtin_bit_cast(s16x4_, u32x2_{w0, w1}), c, 0, 0, 0);
}
__device__ __forceinline__ float max3_f32(float a, float b, float c) {        // v_max3_f32 without fmaxf's canonicalisation of each input
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

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

// 16-byte chunk swizzle of the row-major [row][HD] bf16 LDS images.  Measured (tools/probe/lds_conflict_probe.hip, 16 waves hammering
// the two access patterns of these kernels -- ds_read_b128 fragments: lane (r, g) -> row r, chunk g; ds_read_b64_tr_b16: lane (r, g) ->
// row 4g + r/4, chunk (r & 3) / 2, byte 8 (r & 1)), cycles per 4 reads x 16 waves:
//   64-byte rows (head_dim 32):  (row >> 1) & 3 : 208 / 128     the round-1 choice (row >> 2) & 3 : 321 / 192 (= no swizzle at all)
//   128-byte rows (head_dim 64): row & 7        : 224 / 128     the round-1 choice (row >> 1) & 7 : 224 / 195
// Periodic in 8 rows, so per-tile immediate offsets (16 rows) still work.
template <int HD>
__device__ __forceinline__ int swz_chunk(int row) { return HD == 32 ? ((row >> 1) & 3) : (row & 7); }
template <int HD>
__device__ __forceinline__ int k_off_swz(int row, int chunk) {   // row-major [row][HD] bf16, 16B chunk swizzle
  if (HD == 32) return row * 64 + ((chunk ^ swz_chunk<32>(row)) << 4);
  return row * 128 + ((chunk ^ swz_chunk<64>(row)) << 4);
}

// a pointer the compiler must treat as wave-uniform (it is: derived from blockIdx and loop counters, but after SGPR spilling the
// compiler loses that and wraps every buffer instruction built on it in a waterfall loop)
template <typename T>
__device__ __forceinline__ const T* uniform_ptr(const T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const T*>(((uint64_t)hi << 32) | lo);
}
// fill row-major swizzled [rows][HD] from global rows (zero beyond L) with direct-to-LDS DMA (buffer_load ... lds): no VGPR
// round trip, all requests of a thread in flight at once.  The LDS image is lane-linear per wave instruction, so the chunk
// swizzle is applied to the SOURCE column; rows >= L fall beyond the descriptor's num_records and read as zero.
// Caller must `s_waitcnt vmcnt(0)` + barrier before reading.
template <int HD>
__device__ __forceinline__ void fill_rowmajor(unsigned char* dst, const u16* src, int ld, int L, int rows, int tid, int nthreads) {
  constexpr int CPR = HD / 8;
  const unsigned bytes = (unsigned)(((size_t)(L - 1) * ld + HD) * 2);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(src)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  const int total = rows * CPR;
  const int wave_base = tid & ~63;
  typedef __attribute__((address_space(3))) void lds_void;
  for (int i0 = 0; i0 < total; i0 += nthreads) {
    const int u = i0 + tid;
    if (u < total) {
      const int row = u / CPR, chs = u - row * CPR;
      const int ch = chs ^ swz_chunk<HD>(row);
      const unsigned goff = (unsigned)(((size_t)row * ld + ch * 8) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(dst + (size_t)(i0 + wave_base) * 16), 16, goff, 0, 0, 0);
    }
  }
}
// K and V images of one (sequence, head) with PRECOMPUTED per-thread source offsets (goff[i] = 0xffffffff: beyond the image)
template <int NF, int STEP>
__device__ __forceinline__ void fill_pre(unsigned char* dst, int kv_bytes, const u16* ksrc, const u16* vsrc, unsigned bytes, const uint32_t* goff) {
  typedef __attribute__((address_space(3))) void lds_void;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(ksrc)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(vsrc)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    if (i < NF - 1 || goff[i] != 0xffffffffu) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)(dst + i * STEP), 16, goff[i], 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + kv_bytes + i * STEP), 16, goff[i], 0, 0, 0);
    }
  }
}
// one 16-byte request of each image (the caller spreads the NF requests over its compute loop: the texture addresser takes
// ~16 cycles per 1-KiB wave request, so 8 waves issuing a whole fill back to back serialise for ~1500 cycles)
__device__ __forceinline__ void fill_one(unsigned char* dst, int kv_bytes, const u16* ksrc, const u16* vsrc, unsigned bytes, uint32_t goff, bool guard) {
  typedef __attribute__((address_space(3))) void lds_void;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(ksrc)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(vsrc)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  if (!guard || goff != 0xffffffffu) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)dst, 16, goff, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + kv_bytes), 16, goff, 0, 0, 0);
  }
}
// the same with the two buffer resources built by the caller ONCE per sequence (building them per request costs 8 v_readfirstlane each)
__device__ __forceinline__ void fill_one_r(unsigned char* dst, int kv_bytes, const __amdgpu_buffer_rsrc_t rk, const __amdgpu_buffer_rsrc_t rv, uint32_t goff, bool guard) {
  typedef __attribute__((address_space(3))) void lds_void;
  if (!guard || goff != 0xffffffffu) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void*)dst, 16, goff, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_void*)(dst + kv_bytes), 16, goff, 0, 0, 0);
  }
}
// DMA helpers for kernels that precompute per-thread source offsets (0xffffffff: beyond the image -> skipped when `guard`)
__device__ __forceinline__ void dma16_pair(unsigned char* dst, int img_bytes, const u16* src0, unsigned bytes0, uint32_t off0,
                                           const u16* src1, unsigned bytes1, uint32_t off1, bool guard) {
  typedef __attribute__((address_space(3))) void lds_void;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(src0)), 0, __builtin_amdgcn_readfirstlane((int)bytes0), 0x00020000);
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(uniform_ptr(src1)), 0, __builtin_amdgcn_readfirstlane((int)bytes1), 0x00020000);
  if (!guard || off0 != 0xffffffffu) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_void*)dst, 16, off0, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_void*)(dst + img_bytes), 16, off1, 0, 0, 0);
  }
}
__device__ __forceinline__ void dma16_one(unsigned char* dst, const float* src, unsigned bytes, uint32_t off) {
  typedef __attribute__((address_space(3))) void lds_void;
  const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_ptr(src)), 0, __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_void*)dst, 16, off, 0, 0, 0);
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


// ---- transposing LDS reads through inline asm (asynchronous: results are valid only after tr_wait*) ----
// The ds_read_tr builtin carries no memory operand, so the compiler's wait-count pass assumes it may alias any in-flight
// direct-to-LDS load and drains vmcnt before it; kernels that overlap the next fill with compute must issue it this way.
__device__ __forceinline__ uint32_t lds_addr(const unsigned char* p) {
  typedef __attribute__((address_space(3))) const unsigned char lds_u8;
  return (uint32_t)(size_t)(lds_u8*)p;
}
__device__ __forceinline__ void tr_read4(s16x4& a0, s16x4& a1, s16x4& c0, s16x4& c1, uint32_t pa, uint32_t pc, const int off) {
  // `off` must fold to a constant after inlining / unrolling: it becomes the instruction's immediate offset (one base VGPR
  // per operand instead of one address VGPR per tile)
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%6\n\tds_read_b64_tr_b16 %1, %4 offset:%7\n\t"
               "ds_read_b64_tr_b16 %2, %5 offset:%6\n\tds_read_b64_tr_b16 %3, %5 offset:%7"
               : "=&v"(a0), "=&v"(a1), "=&v"(c0), "=&v"(c1) : "v"(pa), "v"(pc), "i"(off), "i"(off + 1024) : "memory");
}
// Two 16-byte LDS reads through inline asm (immediate offset, as tr_read4): a plain load of an LDS region that a direct-to-LDS DMA
// also writes makes the compiler put s_waitcnt vmcnt(0) in front of it -- i.e. the wave drains the NEXT sequence's prefetch it has
// just issued (measured in attn_bwd_dkv_win2_kernel: 5 drains per sequence, ~half of its wave cycles)
__device__ __forceinline__ void lds_read2_b128(f32x4& a, f32x4& b, uint32_t pa, uint32_t pb, const int off) {
  asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4" : "=&v"(a), "=&v"(b) : "v"(pa), "v"(pb), "i"(off) : "memory");
}
__device__ __forceinline__ void lds_wait2(f32x4& a, f32x4& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void tr_wait4(s16x4& a0, s16x4& a1, s16x4& c0, s16x4& c1) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1));
}

// Attention-probability dropout threshold: p is quantised to 1/65536 (0.1 -> 6554/65536 = 0.10001, the rate of the GEMM-epilogue
// and query-row streams) although a 4 x 4 block still draws 16 random BYTES from one Philox evaluation: element j of a word compares the
// 16-bit field (byte j+1 mod 4 : byte j), i.e. its NEIGHBOUR's byte j+1 is the high byte and decides; its own byte j is the low byte
// and only breaks the tie when the high byte equals the threshold's high byte (1 element in 256).  The high bytes of the four elements
// are a permutation of the word's bytes, so every element's marginal drop probability is exactly thr16 / 65536 and neighbouring
// decisions are coupled only through that 1-in-256 tie-break.  (Rounds 1-3: an 8-bit compare, p_eff = 26/256 = 0.1016.)
__device__ __forceinline__ uint32_t drop_thr16(float p) { return (uint32_t)(p * 65536.f + 0.5f); }
__device__ __forceinline__ uint32_t drop_field(uint32_t w, int j) {      // j in 0..3 (compile- or run-time): v_alignbit + mask
  return __builtin_amdgcn_alignbit(w, w, (uint32_t)(8 * j)) & 0xffffu;
}
// 4x4 block of 8-bit randoms for (query block qb = q/4, key block kb = key/4) of (seq,head) stream `sh`
__device__ __forceinline__ uint4 drop_block(uint64_t seed, uint64_t offset, uint32_t sh, uint32_t qb, uint32_t kb) {
  const uint64_t c = offset + (((uint64_t)sh << 32) | ((uint64_t)qb << 12) | kb);
  return philox4x32_7(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0xa77eu, 0u), make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}
// a where the lane's bit of the (wave-uniform) mask m is clear, 0 where it is set: v_cndmask_b32 with the mask as a scalar pair
__device__ __forceinline__ float zero_where(float a, uint64_t m) {
  float d;
  asm("v_cndmask_b32_e64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "s"(m));
  return d;
}
// The four 64-bit lane masks m0..m3 of a score tile's dropout compares to base[off .. off + 32) -- eight dwords (m0 low, m0 high, m1 low,
// ...) -- by SCALAR stores: the masks are wave-uniform SGPR pairs (v_cmp destinations), so the record costs no vector instruction.
// (A first version moved them into lanes 0-7 of a VGPR with eight v_writelane_b32 + one vector store: +17 % on the forward kernel, and
// it ran into a hardware hazard -- v_writelane_b32 reading an SGPR that the VALU instruction in front of it wrote gets the register's
// PREVIOUS value, tools/probe/writelane_probe.hip / profiles/r04_probe_writelane_sgpr_hazard.txt.)  s_nop 4: the wait states between a
// VALU write of an SGPR and a memory instruction that reads it (the compiler does not see inside the asm).  The scalar data cache is
// written back by scalar_stores_done() before the wave ends; the consumers are later kernels.
__device__ __forceinline__ void store_lane_masks(const uint32_t* base, const int off, uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3) {
  asm volatile("s_nop 4\n\ts_store_dwordx2 %0, %4, %5\n\ts_store_dwordx2 %1, %4, %6\n\ts_store_dwordx2 %2, %4, %7\n\ts_store_dwordx2 %3, %4, %8"
               :: "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(base), "i"(off), "i"(off + 8), "i"(off + 16), "i"(off + 24) : "memory");
}
__device__ __forceinline__ void scalar_stores_done() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t u4_get(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
// The four lanes of a quad (4 consecutive query rows in the forward / dQ layout, 4 consecutive keys in the dK/dV layout) need the
// SAME 4x4 block of random bytes for a tile, so one Philox evaluation per tile wastes 3/4 of the wave's work.  Instead quad lane i
// evaluates the block of tile t0+i (one Philox call per FOUR tiles) and the quad reads it back with a DPP quad broadcast.
template <int I>
__device__ __forceinline__ uint4 quad_bcast(const uint4& b) {
  constexpr int ctrl = I | (I << 2) | (I << 4) | (I << 6);             // quad_perm:[I,I,I,I]
  return make_uint4((uint32_t)__builtin_amdgcn_mov_dpp((int)b.x, ctrl, 0xf, 0xf, false), (uint32_t)__builtin_amdgcn_mov_dpp((int)b.y, ctrl, 0xf, 0xf, false),
                    (uint32_t)__builtin_amdgcn_mov_dpp((int)b.z, ctrl, 0xf, 0xf, false), (uint32_t)__builtin_amdgcn_mov_dpp((int)b.w, ctrl, 0xf, 0xf, false));
}
__device__ __forceinline__ uint4 quad_bcast_i(const uint4& b, int i) {   // i is a constant after unrolling
  return i == 0 ? quad_bcast<0>(b) : i == 1 ? quad_bcast<1>(b) : i == 2 ? quad_bcast<2>(b) : quad_bcast<3>(b);
}
// Forward / dQ layout: quad lane c (= q & 3) needs dword c of the block of every tile of the group, i.e. the 4 x 4 TRANSPOSE of
// (owner lane, dword) across the quad: y[u] = block of tile t0 + u, dword c  ==  u4_get(quad_bcast<u>(own), q & 3).  Two butterfly
// stages (lane ^ 1, lane ^ 2), 4 DPP moves + 12 selects per FOUR tiles, instead of 4 DPP moves + a lane-indexed dword select per
// tile (which the compiler turned into exec-mask branches).  b0 / b1 = bit 0 / bit 1 of the lane index.
__device__ __forceinline__ uint4 quad_transpose(const uint4& x, bool b0, bool b1) {
  constexpr int X1 = 1 | (0 << 2) | (3 << 4) | (2 << 6), X2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);       // quad_perm:[1,0,3,2] / [2,3,0,1]
  const uint32_t s0 = b0 ? x.x : x.y, s1 = b0 ? x.z : x.w;
  const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)s0, X1, 0xf, 0xf, false), r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)s1, X1, 0xf, 0xf, false);
  const uint32_t z0 = b0 ? r0 : x.x, z1 = b0 ? x.y : r0, z2 = b0 ? r1 : x.z, z3 = b0 ? x.w : r1;
  const uint32_t t0 = b1 ? z0 : z2, t1 = b1 ? z1 : z3;
  const uint32_t q0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t0, X2, 0xf, 0xf, false), q1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)t1, X2, 0xf, 0xf, false);
  return make_uint4(b1 ? q0 : z0, b1 ? q1 : z1, b1 ? z2 : q0, b1 ? z3 : q1);
}
__device__ __forceinline__ uint32_t u4_static(const uint4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }      // i constant after unrolling

}  // namespace
