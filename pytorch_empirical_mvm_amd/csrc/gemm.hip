// gemm.hip -- bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
// C[M,N] = epilogue( sum_k A(m,k) B(n,k) ), f32 accumulate on v_mfma_f32_16x16x32_bf16.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA tiles.
// Operands are staged global -> registers -> LDS (double buffered, one barrier per K tile; the
// global loads of tile t+1 are in flight while tile t is multiplied).  k-major operand tiles are
// stored [128][64] with a 16-byte-chunk XOR swizzle ((row>>1)&7) so the ds_read_b128 fragment
// reads are conflict free; m/n-major operand tiles ([64 k][128]) are read with the gfx950
// transposing LDS read ds_read_b64_tr_b16 (32-byte slot swizzle), so wgrad/dgrad need no
// transposed copies of activations or weights in HBM.
// The MFMA is issued with swapped operands (D' = B_frag x A_frag) so each lane owns 4 CONSECUTIVE
// output columns of one row -> 8-byte bf16 / 16-byte f32 stores and vector bias/residual loads.
// Block -> tile mapping is XCD-aware: the 8 XCDs get contiguous runs of the tile list, and all
// N-tiles of one M-panel are neighbours, so an activation panel is fetched into one L2 only.
#include "common.h"
#include <vmvm_probe_hooks.h>
#include "gemm_epi.h"

int vmvm_colsum_scaled(const void* X, int32_t M, int32_t N, int32_t ldx, float scale, float* out, void* ws, int64_t ws_bytes, void* stream);      // misc.hip
int vmvm_gemm_pp(const vmvm_gemm_desc& d, int need, hipStream_t st);
int vmvm_gemm_pp_fp8(const vmvm_gemm_desc& d, int need, hipStream_t st);      // gemm_pp.hip: 256x256 ping-pong main loop

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KiB per operand tile
constexpr int SMEM_BYTES = 4 * TILE_BYTES;       // A,B x double buffer = 64 KiB
constexpr int PERS_SMEM_BYTES = SMEM_BYTES + 4 * 4096;   // + the wave-private store tiles of gemm_pers_kernel: 80 KiB, two workgroups per CU fill the 160 KiB

__device__ __forceinline__ int swz_m(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }

// ---- global -> register staging of one operand tile (4 x 16B chunks per thread) ----------------
template <bool KMAJOR>
__device__ __forceinline__ void load_tile(const u16* __restrict__ P, int ld, int rows_total, int K,
                                          int row0, int k0, int tid, uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (KMAJOR) {
      const int row = row0 + (c >> 3), k = k0 + (c & 7) * 8;
      if (row < rows_total && k < K) v = *reinterpret_cast<const uint4*>(P + (size_t)row * ld + k);
    } else {
      const int k = k0 + (c >> 4), x = row0 + (c & 15) * 8;
      if (k < K && x < rows_total) v = *reinterpret_cast<const uint4*>(P + (size_t)k * ld + x);
    }
    r[i] = v;
  }
}

template <bool KMAJOR>
__device__ __forceinline__ void store_tile(unsigned char* lds, int tid, const uint4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    int off;
    if (KMAJOR) {
      const int row = c >> 3, kc = c & 7;
      off = row * 128 + ((kc ^ ((row >> 1) & 7)) << 4);
    } else {
      const int krow = c >> 4, xc = c & 15;
      off = krow * 256 + ((((xc >> 1) ^ swz_m(krow)) & 7) << 5) + ((xc & 1) << 4);
    }
    *reinterpret_cast<uint4*>(lds + off) = r[i];
  }
}

// ---- direct global -> LDS staging (buffer_load_dwordx4 ... lds): no VGPR round trip, no ds_write.  The LDS image is
// lane-linear per wave instruction (wave-uniform base + lane*16), so the XOR swizzle is applied to the per-lane SOURCE
// offset (same involution as the read side).  Out-of-range rows fall beyond num_records and read as zero.
// k-major chunk swizzle: rows of one fragment read must map to distinct 16-byte slots.  Plain fragments read 16 consecutive
// rows (swizzle by (row>>1)&7); the N-permuted B fragments read rows {8a + 4hf + b} (swizzle by b/2 | a<<1).
template <bool PERM>
__device__ __forceinline__ int kswz(int row) { return PERM ? (((row >> 1) & 1) | (((row >> 3) & 3) << 1)) : ((row >> 1) & 7); }

template <bool KMAJOR, bool PERM = false>
__device__ __forceinline__ void issue_tile(__amdgpu_buffer_rsrc_t rsrc, int ld, int row0, int k0, unsigned char* lds, int tid) {
  const int wave_base = (tid & ~63);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + i * 256;
    unsigned goff;
    if (KMAJOR) {
      const int row = u >> 3, cs = u & 7;
      const int c = cs ^ kswz<PERM>(row);
      goff = (unsigned)(((size_t)(row0 + row) * ld + k0 + c * 8) * 2);
    } else {
      const int krow = u >> 4, unit = u & 15;
      const int slot = ((unit >> 1) ^ swz_m(krow)) & 7;
      goff = (unsigned)(((size_t)(k0 + krow) * ld + row0 + (slot * 2 + (unit & 1)) * 8) * 2);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + (i * 256 + wave_base) * 16), 16, goff, 0, 0, 0);
  }
}

// ---- LDS -> MFMA fragment (8 bf16 along k for row/col `x` of the tile), k-step s (32 wide) ------
template <bool KMAJOR, bool TR>
__device__ __forceinline__ bf16x8 read_frag(const unsigned char* lds, int x16, int s, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if (KMAJOR) {
    const int row = x16 * 16 + r, kc = s * 4 + g;
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else if (TR) {
    // two transposing reads: block rows k = 32s+8g+{0..3} and +{4..7}, 16 columns of slot x16
    const int krow = s * 32 + g * 8 + (r >> 2);
    const int o1 = krow * 256 + (((x16 ^ swz_m(krow)) & 7) << 5) + (r & 3) * 8;
    const int krow2 = krow + 4;
    const int o2 = krow2 * 256 + (((x16 ^ swz_m(krow2)) & 7) << 5) + (r & 3) * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o1));
    s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o2));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
  } else {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int krow = s * 32 + g * 8 + e;
      const int off = krow * 256 + (((x16 ^ swz_m(krow)) & 7) << 5) + r * 2;
      v[e] = *reinterpret_cast<const short*>(lds + off);
    }
    return __builtin_bit_cast(bf16x8, v);
  }
}


// B-operand fragment with the permuted N index: MFMA tile (jb, hf) of a 32-column block covers columns
// 32*jb + 8*(q/4) + 4*hf + q%4 for operand lane q, so D' row 4g+e of the tile pair is column 32*jb + 8g + 4*hf + e.
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag_bperm(const unsigned char* lds, int col_base, int hf, int s, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if (KMAJOR) {
    const int row = col_base + 8 * (r >> 2) + 4 * hf + (r & 3), kc = s * 4 + g;
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ kswz<true>(row)) << 4));
  } else {
    // transposing read: the 4 lanes p = r&3 of a k-row supply the 4 x 8-byte pieces; piece p now comes from columns
    // col_base + 8p + 4hf .. +3 (instead of 4p..4p+3), which realises the same column permutation
    const int col = col_base + 8 * (r & 3) + 4 * hf;
    const int slot = col >> 4, sub = (col & 15) * 2;
    const int krow = s * 32 + g * 8 + (r >> 2), krow2 = krow + 4;
    const int o1 = krow * 256 + (((slot ^ swz_m(krow)) & 7) << 5) + sub;
    const int o2 = krow2 * 256 + (((slot ^ swz_m(krow2)) & 7) << 5) + sub;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o1));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o2));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}


// ---- asynchronous transposing reads (inline asm) --------------------------------------------------------------------------
// The ds_read_tr builtin carries no memory operand, so the compiler's wait-count pass assumes it may alias an in-flight
// direct-to-LDS load and emits s_waitcnt vmcnt(0) in front of it -- which drains the NEXT K tile's DMA right after it was
// issued (no load/compute overlap inside a workgroup).  Issued through asm the reads are invisible to that pass; the halves
// are only valid after tr_wait8 (s_waitcnt lgkmcnt(0)), which also ties the registers so no use can be scheduled early.
__device__ __forceinline__ uint32_t lds_addr32(const unsigned char* p) {
  typedef __attribute__((address_space(3))) const unsigned char lds_u8;
  return (uint32_t)(size_t)(lds_u8*)p;
}
__device__ __forceinline__ void tr_issue2(s16x4& lo, s16x4& hi, uint32_t a1, uint32_t a2) {
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(a1), "v"(a2) : "memory");
}
__device__ __forceinline__ void tr_wait8(s16x4* lo, s16x4* hi) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]));
}
__device__ __forceinline__ bf16x8 tr_cat(const s16x4& a, const s16x4& b) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// m-major operand, slot x16 (same addressing as read_frag<false, true>)
__device__ __forceinline__ void tr_issue_frag(const unsigned char* lds, int x16, int s, int lane, s16x4& lo, s16x4& hi) {
  const int r = lane & 15, g = lane >> 4;
  const int krow = s * 32 + g * 8 + (r >> 2), krow2 = krow + 4;
  const int o1 = krow * 256 + (((x16 ^ swz_m(krow)) & 7) << 5) + (r & 3) * 8;
  const int o2 = krow2 * 256 + (((x16 ^ swz_m(krow2)) & 7) << 5) + (r & 3) * 8;
  tr_issue2(lo, hi, lds_addr32(lds + o1), lds_addr32(lds + o2));
}
// n-major B operand with the column permutation of read_frag_bperm<false>
__device__ __forceinline__ void tr_issue_bperm(const unsigned char* lds, int col_base, int hf, int s, int lane, s16x4& lo, s16x4& hi) {
  const int r = lane & 15, g = lane >> 4;
  const int col = col_base + 8 * (r & 3) + 4 * hf;
  const int slot = col >> 4, sub = (col & 15) * 2;
  const int krow = s * 32 + g * 8 + (r >> 2), krow2 = krow + 4;
  const int o1 = krow * 256 + (((slot ^ swz_m(krow)) & 7) << 5) + sub;
  const int o2 = krow2 * 256 + (((slot ^ swz_m(krow2)) & 7) << 5) + sub;
  tr_issue2(lo, hi, lds_addr32(lds + o1), lds_addr32(lds + o2));
}

template <bool AK, bool BKM, bool TR, bool DIRECT>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const vmvm_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.M, N = p.N, K = p.K;
  const int nbn = (N + BN - 1) / BN, nbm = (M + BM - 1) / BM;
  const int nb = nbm * nbn;
  // XCD-aware bijective remap (block b runs on XCD b%8)
  const int bid = blockIdx.x;
  const int q = nb >> 3, rr = nb & 7, xcd = bid & 7, idx = bid >> 3;
  // XCD remap over the whole grid (tiles x K-slices); consecutive logical ids = neighbouring tiles of one K-slice
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nbt = nb * S;
  const int q2 = nbt >> 3, rr2 = nbt & 7;
  const int logical = (xcd < rr2 ? xcd * (q2 + 1) : rr2 * (q2 + 1) + (xcd - rr2) * q2) + idx;
  const int slice = logical / nb, tile = logical - slice * nb;
  int tm, tn;
  raster(tile, nbm, nbn, 8, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  (void)q; (void)rr;

  const u16* A = reinterpret_cast<const u16*>(p.A);
  const u16* B = reinterpret_cast<const u16*>(p.B);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (K + BK - 1) / BK;
  const int per = (nk_all + S - 1) / S;
  const int kt0 = slice * per;
  const int nk = (kt0 + per < nk_all) ? kt0 + per : nk_all;        // this block multiplies K-tiles [kt0, nk)
  if (kt0 >= nk) return;

  auto mma_tile = [&](const unsigned char* la, const unsigned char* lb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag<AK, TR>(la, wm * 4 + i, s, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<BKM, TR>(lb, wn * 4 + j, s, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  };

  if (DIRECT) {
    const size_t bytesA = (size_t)(AK ? M : K) * p.lda * 2, bytesB = (size_t)(BKM ? N : K) * p.ldb * 2;
    const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(A), 0, (int)bytesA, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(B), 0, (int)bytesB, 0x00020000);
    issue_tile<AK>(ra_, p.lda, m0, kt0 * BK, smem, tid);
    issue_tile<BKM>(rb_, p.ldb, n0, kt0 * BK, smem + TILE_BYTES, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      const unsigned char* la = smem + cur * 2 * TILE_BYTES;
      if (kt + 1 < nk) {
        unsigned char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
        issue_tile<AK>(ra_, p.lda, m0, (kt + 1) * BK, na, tid);
        issue_tile<BKM>(rb_, p.ldb, n0, (kt + 1) * BK, na + TILE_BYTES, tid);
      }
      mma_tile(la, la + TILE_BYTES);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  } else {
    uint4 ra[4], rb[4];
    load_tile<AK>(A, p.lda, M, K, m0, kt0 * BK, tid, ra);
    load_tile<BKM>(B, p.ldb, N, K, n0, kt0 * BK, tid, rb);
    store_tile<AK>(smem, tid, ra);
    store_tile<BKM>(smem + TILE_BYTES, tid, rb);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
      const int cur = (kt - kt0) & 1;
      const unsigned char* la = smem + cur * 2 * TILE_BYTES;
      if (kt + 1 < nk) {
        load_tile<AK>(A, p.lda, M, K, m0, (kt + 1) * BK, tid, ra);
        load_tile<BKM>(B, p.ldb, N, K, n0, (kt + 1) * BK, tid, rb);
      }
      mma_tile(la, la + TILE_BYTES);
      if (kt + 1 < nk) {
        unsigned char* na = smem + (cur ^ 1) * 2 * TILE_BYTES;
        store_tile<AK>(na, tid, ra);
        store_tile<BKM>(na + TILE_BYTES, tid, rb);
      }
      __syncthreads();
    }
  }

  // ------------------------------- epilogue -------------------------------
  const int r = lane & 15, g = lane >> 4;
  EpiCtx ec;
  ec.has_drop = p.dropout_p > 0.f; ec.thr = dropout_threshold(p.dropout_p);
  ec.keep_scale = ec.has_drop ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  ec.S = S; ec.slice = slice; ec.M = M; ec.N = N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + r;
    if (m >= M) continue;
    long dst = m;
    if (p.row_map) {
      const int within = m % p.map_len;
      const int mapped = p.row_map[within];
      if (mapped < 0) continue;
      dst = (long)mapped + (long)(m / p.map_len) * p.map_stride;
    }
    const float rs = p.row_scale ? p.row_scale[(m + p.scale_row0) / p.rows_per_scale] : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epi_store(p, ec, v, m, dst, n, rs);
    }
  }
}

// C[m][n] += sum_s ws[s][m][n]   (float4 lanes; every slab element was written by exactly one block), slabs summed in index order.
// Small outputs with hundreds of slices (the 128 x 128 weight gradient of Swin stage 1: 512 slabs, 16 workgroups' worth of output)
// would sum their slabs one after the other on a handful of CUs (122 us): the slabs are split over blockIdx.y.  Rounds 2-5 combined the
// gridDim.y partial sums with f32 atomics; round 6 (run-to-run reproducibility): TWO PASSES -- pass 1 (PARTIAL = true) leaves the sum of
// the slabs [y per, (y + 1) per) in slab y per (every thread reads all its slabs before it writes, and no other y touches them), pass 2
// sums those `ny` slabs in order into C.  No atomics, one more ~3 us launch for the few small-output shapes that split over y.
// Fused bias gradient (vmvm_gemm_desc.colsum) of a split problem: the GEMM units leave parts[slice][m] behind the slabs
// (colsum_parts()), the first M threads of the final pass add them in slice order.
template <bool PARTIAL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(float* __restrict__ ws, float* __restrict__ C, int M, int N, int ldc, int S, int sstep, int per,
                                                            float* __restrict__ colsum, const float* __restrict__ parts, int S_parts) {
  const long n4 = N >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (!PARTIAL && colsum && i < M) {
    float t = 0.f;
    for (int s = 0; s < S_parts; ++s) t += parts[(size_t)s * M + i];
    colsum[i] += t;
  }
  if (i >= (long)M * n4) return;
  const long m = i / n4, c = (i - m * n4) * 4;
  const size_t slab = (size_t)M * N;
  const int s0 = PARTIAL ? blockIdx.y * per : 0, s1 = PARTIAL ? ((s0 + per < S) ? s0 + per : S) : S;
  // the slabs are read ONCE (non-temporal) and four at a time -- the loop body used to be one dependent load-add per slab, so a
  // thread had 16 bytes in flight; the sum order is the slab order
  typedef float v4f __attribute__((ext_vector_type(4)));
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* src = ws + m * N + c;
  const size_t st_ = slab * (size_t)sstep;
  int s = s0;
  for (; s + 4 <= s1; s += 4) {
    v4f v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + (size_t)(s + u) * st_));
#pragma unroll
    for (int u = 0; u < 4; ++u) { a.x += v[u][0]; a.y += v[u][1]; a.z += v[u][2]; a.w += v[u][3]; }
  }
  for (; s < s1; ++s) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + (size_t)s * st_));
    a.x += v[0]; a.y += v[1]; a.z += v[2]; a.w += v[3];
  }
  if (PARTIAL) {
    if (s1 > s0) *reinterpret_cast<float4*>(ws + (size_t)s0 * slab + m * N + c) = a;
  } else {
    float* o = C + m * ldc + c;
    const float4 c0 = *reinterpret_cast<const float4*>(o);
    *reinterpret_cast<float4*>(o) = make_float4(c0.x + a.x, c0.y + a.y, c0.z + a.z, c0.w + a.w);
  }
}
static inline int launch_splitk_reduce(const vmvm_gemm_desc& d, int S, hipStream_t st) {
  float* ws = reinterpret_cast<float*>(d.workspace);
  float* C = reinterpret_cast<float*>(d.C);
  const int M = d.M, N = d.N;
  const long n = (long)M * (N >> 2);
  const unsigned nbx = (unsigned)((n + 255) / 256);
  unsigned ny = 1;
  if (nbx <= 64 && S >= 16) {                            // a quarter of the CUs or fewer: spread the slabs too (from 256 workgroups of output on,
    ny = (1024 + nbx - 1) / nbx;                         //  the second pass costs more than the sequential sum: 512 x 512, 32 slabs: 47 -> 63 us)
    if (ny > (unsigned)S / 4) ny = (unsigned)S / 4;
    if (ny < 1) ny = 1;
  }
  float* parts = colsum_parts(d, S);
  float* cs = parts ? d.colsum : nullptr;
  if (ny > 1) {
    const int per = (S + (int)ny - 1) / (int)ny;
    const int ny_eff = (S + per - 1) / per;
    hipLaunchKernelGGL((splitk_reduce_kernel<true>), dim3(nbx, ny_eff), dim3(256), 0, st, ws, C, M, N, d.ldc, S, 1, per, nullptr, nullptr, 0);
    hipLaunchKernelGGL((splitk_reduce_kernel<false>), dim3(nbx, 1), dim3(256), 0, st, ws, C, M, N, d.ldc, ny_eff, per, 0, cs, parts, S);
  } else {
    hipLaunchKernelGGL((splitk_reduce_kernel<false>), dim3(nbx, 1), dim3(256), 0, st, ws, C, M, N, d.ldc, S, 1, 0, cs, parts, S);
  }
  return 0;
}

template <bool AK, bool BKM, bool TR, bool DIRECT>
int launch(const vmvm_gemm_desc& d, hipStream_t st) {
  const int nb = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN) * (d.splitk > 1 ? d.splitk : 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<AK, BKM, TR, DIRECT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_kernel<AK, BKM, TR, DIRECT>), dim3(nb), dim3(256), SMEM_BYTES, st, d);
  VMVM_CHECK_LAUNCH();
  if (d.splitk > 1 && d.workspace) {
    launch_splitk_reduce(d, d.splitk, st);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}

// =====================================================================================================================
// 256x256x64 tile, 512 threads = 8 waves (2 x 4), wave tile 128x64 = 8x4 MFMA tiles (128 accumulator VGPRs).
// A 128^2 tile needs 64 B/clk/CU of L2->LDS traffic at MFMA peak; the 256^2 tile halves that (128 flop per staged byte),
// which is what lifts the large fusion-encoder / Swin stage-3 GEMMs above the L2-bandwidth ceiling of the small tile.
// Direct-to-LDS staging only (buffer_load ... lds, double buffered, one barrier per K tile), transposing reads for
// m/n-major operands; 128 KiB LDS -> one workgroup per CU, 2 waves per SIMD.
// =====================================================================================================================
constexpr int GB = 256;                              // big tile edge
constexpr int BIG_TILE_BYTES = GB * BK * 2;          // 32 KiB per operand tile
constexpr int BIG_SMEM_BYTES = 4 * BIG_TILE_BYTES;   // 128 KiB

template <bool KMAJOR>
__device__ __forceinline__ void issue_tile_big(__amdgpu_buffer_rsrc_t rsrc, int ld, int row0, int k0, unsigned char* lds, int tid) {
  const int wave_base = (tid & ~63);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int u = tid + i * 512;                     // 2048 16-byte units per tile
    unsigned goff;
    if (KMAJOR) {
      const int row = u >> 3, cs = u & 7;
      const int c = cs ^ ((row >> 1) & 7);
      goff = (unsigned)(((size_t)(row0 + row) * ld + k0 + c * 8) * 2);
    } else {
      const int krow = u >> 5, unit = u & 31;        // [64 k][256 x]: 32 units per k-row
      const int sl = unit >> 1;
      const int slot = (sl & ~7) | ((sl ^ swz_m(krow)) & 7);
      goff = (unsigned)(((size_t)(k0 + krow) * ld + row0 + (slot * 2 + (unit & 1)) * 8) * 2);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + (i * 512 + wave_base) * 16), 16, goff, 0, 0, 0);
  }
}

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag_big(const unsigned char* lds, int x16, int s, int lane) {
  const int r = lane & 15, g = lane >> 4;
  if (KMAJOR) {
    const int row = x16 * 16 + r, kc = s * 4 + g;
    return *reinterpret_cast<const bf16x8*>(lds + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4));
  } else {
    const int krow = s * 32 + g * 8 + (r >> 2), krow2 = krow + 4;
    const int o1 = krow * 512 + (((x16 & ~7) | ((x16 ^ swz_m(krow)) & 7)) << 5) + (r & 3) * 8;
    const int o2 = krow2 * 512 + (((x16 & ~7) | ((x16 ^ swz_m(krow2)) & 7)) << 5) + (r & 3) * 8;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o1));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + o2));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(const vmvm_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int M = p.M, N = p.N, K = p.K;
  const int nbn = (N + GB - 1) / GB, nbm = (M + GB - 1) / GB;
  const int nb = nbm * nbn;
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nbt = nb * S;
  const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
  const int q2 = nbt >> 3, rr2 = nbt & 7;
  const int logical = (xcd < rr2 ? xcd * (q2 + 1) : rr2 * (q2 + 1) + (xcd - rr2) * q2) + idx;
  const int slice = logical / nb, tile = logical - slice * nb;
  int tm, tn;
  raster(tile, nbm, nbn, 4, tm, tn);
  const int m0 = tm * GB, n0 = tn * GB;
  const u16* A = reinterpret_cast<const u16*>(p.A);
  const u16* B = reinterpret_cast<const u16*>(p.B);

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (K + BK - 1) / BK;
  const int per = (nk_all + S - 1) / S;
  const int kt0 = slice * per;
  const int nk = (kt0 + per < nk_all) ? kt0 + per : nk_all;
  if (kt0 >= nk) return;
  const size_t bytesA = (size_t)(AK ? M : K) * p.lda * 2, bytesB = (size_t)(BKM ? N : K) * p.ldb * 2;
  const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(A), 0, (int)bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(B), 0, (int)bytesB, 0x00020000);
  issue_tile_big<AK>(ra_, p.lda, m0, kt0 * BK, smem, tid);
  issue_tile_big<BKM>(rb_, p.ldb, n0, kt0 * BK, smem + BIG_TILE_BYTES, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = (kt - kt0) & 1;
    const unsigned char* la = smem + cur * 2 * BIG_TILE_BYTES;
    const unsigned char* lb = la + BIG_TILE_BYTES;
    if (kt + 1 < nk) {
      unsigned char* na = smem + (cur ^ 1) * 2 * BIG_TILE_BYTES;
      issue_tile_big<AK>(ra_, p.lda, m0, (kt + 1) * BK, na, tid);
      issue_tile_big<BKM>(rb_, p.ldb, n0, (kt + 1) * BK, na + BIG_TILE_BYTES, tid);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[8], fb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag_big<BKM>(lb, wn * 4 + j, s, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] = read_frag_big<AK>(la, wm * 8 + i, s, lane);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int r = lane & 15, g = lane >> 4;
  EpiCtx ec;
  ec.has_drop = p.dropout_p > 0.f; ec.thr = dropout_threshold(p.dropout_p);
  ec.keep_scale = ec.has_drop ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  ec.S = S; ec.slice = slice; ec.M = M; ec.N = N;
#pragma clang loop unroll(full)
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + r;
    bool valid = m < M;
    long dst = m;
    if (valid && p.row_map) {
      const int mapped = p.row_map[m % p.map_len];
      valid = mapped >= 0;
      dst = (long)mapped + (long)(m / p.map_len) * p.map_stride;
    }
    const float rs = (valid && p.row_scale) ? p.row_scale[(m + p.scale_row0) / p.rows_per_scale] : 1.0f;
#pragma clang loop unroll(full)
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (valid && n < N) epi_store(p, ec, v, m, dst, n, rs);
    }
  }
}

template <bool AK, bool BKM>
int launch_big(const vmvm_gemm_desc& d, hipStream_t st) {
  const int nb = ((d.M + GB - 1) / GB) * ((d.N + GB - 1) / GB) * (d.splitk > 1 ? d.splitk : 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<AK, BKM>), hipFuncAttributeMaxDynamicSharedMemorySize, BIG_SMEM_BYTES);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_big_kernel<AK, BKM>), dim3(nb), dim3(512), BIG_SMEM_BYTES, st, d);
  VMVM_CHECK_LAUNCH();
  if (d.splitk > 1 && d.workspace) {
    launch_splitk_reduce(d, d.splitk, st);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}

// =====================================================================================================================
// 256x128x64 tile, 512 threads = 8 waves (4 x 2, wave tile 64x64), THREE LDS stages (3 x 48 KiB = 144 KiB, one workgroup
// per CU) with counted vmcnt: the DMA loads of two K tiles stay in flight across the (raw) barrier while a third is
// multiplied.  The 2-stage kernels above are latency bound (PMC: 44% of wave cycles in s_waitcnt/barrier, MFMA 25-33%
// busy, L2 30% busy): one tile of prefetch cannot cover the L2/HBM latency of the next.  One barrier per K tile.
// =====================================================================================================================
constexpr int P3_BM = 256, P3_BN = 128;
constexpr int P3_A_BYTES = P3_BM * BK * 2, P3_B_BYTES = P3_BN * BK * 2;      // 32 KiB + 16 KiB
constexpr int P3_STAGE = P3_A_BYTES + P3_B_BYTES;
constexpr int P3_SMEM = 3 * P3_STAGE;

template <bool KMAJOR>
__device__ __forceinline__ void issue_b128_512(__amdgpu_buffer_rsrc_t rsrc, int ld, int row0, int k0, unsigned char* lds, int tid) {
  const int wave_base = (tid & ~63);                  // [128 rows][64 k] or [64 k][128 x] : 1024 units, 2 per thread
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = tid + i * 512;
    unsigned goff;
    if (KMAJOR) {
      const int row = u >> 3, cs = u & 7;
      const int c = cs ^ ((row >> 1) & 7);
      goff = (unsigned)(((size_t)(row0 + row) * ld + k0 + c * 8) * 2);
    } else {
      const int krow = u >> 4, unit = u & 15;
      const int slot = ((unit >> 1) ^ swz_m(krow)) & 7;
      goff = (unsigned)(((size_t)(k0 + krow) * ld + row0 + (slot * 2 + (unit & 1)) * 8) * 2);
    }
    typedef __attribute__((address_space(3))) void lds_void;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void*)(lds + (i * 512 + wave_base) * 16), 16, goff, 0, 0, 0);
  }
}

template <bool AK, bool BKM>
__global__ __launch_bounds__(512, 2) void gemm_p3_kernel(const vmvm_gemm_desc p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.M, N = p.N, K = p.K;
  const int nbn = (N + P3_BN - 1) / P3_BN, nbm = (M + P3_BM - 1) / P3_BM;
  const int nb = nbm * nbn;
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nbt = nb * S;
  const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
  const int q2 = nbt >> 3, rr2 = nbt & 7;
  const int logical = (xcd < rr2 ? xcd * (q2 + 1) : rr2 * (q2 + 1) + (xcd - rr2) * q2) + idx;
  const int slice = logical / nb, tile = logical - slice * nb;
  int tm, tn;
  raster(tile, nbm, nbn, 4, tm, tn);
  const int m0 = tm * P3_BM, n0 = tn * P3_BN;
  const u16* A = reinterpret_cast<const u16*>(p.A);
  const u16* B = reinterpret_cast<const u16*>(p.B);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk_all = (K + BK - 1) / BK;
  const int per = (nk_all + S - 1) / S;
  const int kt0 = slice * per;
  const int nk = (kt0 + per < nk_all) ? kt0 + per : nk_all;
  if (kt0 >= nk) return;
  const int ntile = nk - kt0;
  const size_t bytesA = (size_t)(AK ? M : K) * p.lda * 2, bytesB = (size_t)(BKM ? N : K) * p.ldb * 2;
  const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(A), 0, (int)bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(B), 0, (int)bytesB, 0x00020000);

  auto issue = [&](int t) {                            // 6 DMA instructions per thread per tile
    unsigned char* st = smem + (t % 3) * P3_STAGE;
    issue_tile_big<AK>(ra_, p.lda, m0, (kt0 + t) * BK, st, tid);
    issue_b128_512<BKM>(rb_, p.ldb, n0, (kt0 + t) * BK, st + P3_A_BYTES, tid);
  };
  issue(0);
  if (ntile > 1) issue(1);
  for (int t = 0; t < ntile; ++t) {
    // tile t landed (this thread's share): at most the 6 loads of tile t+1 may still be outstanding
    if (t + 1 < ntile) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // every wave's share landed AND every wave finished tile t-1
    asm volatile("" ::: "memory");
    if (t + 2 < ntile) issue(t + 2);                  // overwrites the stage read in iteration t-1
    const unsigned char* la = smem + (t % 3) * P3_STAGE;
    const unsigned char* lb = la + P3_A_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag_big<AK>(la, wm * 4 + i, s, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag<BKM, true>(lb, wn * 4 + j, s, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
  }

  const int r = lane & 15, g = lane >> 4;
  EpiCtx ec;
  ec.has_drop = p.dropout_p > 0.f; ec.thr = dropout_threshold(p.dropout_p);
  ec.keep_scale = ec.has_drop ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  ec.S = S; ec.slice = slice; ec.M = M; ec.N = N;
#pragma clang loop unroll(full)
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + r;
    bool valid = m < M;
    long dst = m;
    if (valid && p.row_map) {
      const int mapped = p.row_map[m % p.map_len];
      valid = mapped >= 0;
      dst = (long)mapped + (long)(m / p.map_len) * p.map_stride;
    }
    const float rs = (valid && p.row_scale) ? p.row_scale[(m + p.scale_row0) / p.rows_per_scale] : 1.0f;
#pragma clang loop unroll(full)
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + g * 4;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (valid && n < N) epi_store(p, ec, v, m, dst, n, rs);
    }
  }
}

inline bool tiles_p3_ok(int M, int N) { return ((M + P3_BM - 1) / P3_BM) * ((N + P3_BN - 1) / P3_BN) >= 224; }

template <bool AK, bool BKM>
int launch_p3(const vmvm_gemm_desc& d, hipStream_t st) {
  const int nb = ((d.M + P3_BM - 1) / P3_BM) * ((d.N + P3_BN - 1) / P3_BN) * (d.splitk > 1 ? d.splitk : 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p3_kernel<AK, BKM>), hipFuncAttributeMaxDynamicSharedMemorySize, P3_SMEM);
    attr_done = true;
  }
  hipLaunchKernelGGL((gemm_p3_kernel<AK, BKM>), dim3(nb), dim3(512), P3_SMEM, st, d);
  VMVM_CHECK_LAUNCH();
  if (d.splitk > 1 && d.workspace) {
    launch_splitk_reduce(d, d.splitk, st);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}

// =====================================================================================================================
// Persistent 128x128x64 kernel (direct staging): 2 workgroups per CU walk the (tile, K-slice) list of their XCD in order and
// issue the DMA loads of the NEXT tile's first K tile before the epilogue of the current one.  Short-K problems (Swin
// stage 1/2: K = 128..512, i.e. 2..8 K tiles) are otherwise dominated by the load latency of the prologue and by store
// phases during which nothing is in flight (measured 1.8-2.2 TB/s on HBM-bound shapes = half of what the copy rate allows).
// =====================================================================================================================
// F16: fp16 operands / outputs (v_mfma_f32_16x16x32_f16).  CONV: A is an NHWC activation read as an implicit 3x3 convolution
// (K = 9 * C_in, one tap per C_in/64 consecutive K tiles; rows whose tap leaves the image get an out-of-range DMA offset = zeros).
// FP8: A and B are OCP e4m3 bytes (k-major both); the descriptor reaches the kernel with K / lda / ldb counted in 2-byte units, so
// the staging is unchanged: a 64-"element" K tile is 128 fp8 = one v_mfma_scale_f32_16x16x128_f8f6f4 per fragment pair (a lane's
// operand = the two 16-byte chunks 2g, 2g+1 of its row; block scales fixed at 1.0, the per-tensor scale product is `alpha`).
// TM = 2: tile 256 (M) x 64 (N), the four waves stacked along M (64 x 64 each as in the square tile) -- the dVAE's group-1
// convolutions have 64 output channels, on a 128-wide tile half of every MFMA multiplies zero padding.  ARELU: max(A, 0) on the A
// fragments after the LDS read (fp16 builds: the block-input ReLU of the dVAE's residual path, encoder.py:27-28, without a pass
// over the activation); one v_pk_max_f16 per MFMA, in its shadow.  F & EF_ARGMAX: instead of storing, each 64-column group of a
// row leaves its (maximum, column) pair in C (f32 [M][ldc], pairs at 2 * (n / 64)) -- the 8192-wide logits are never written.
template <bool AK, bool BKM, int F, bool F16 = false, bool CONV = false, bool FP8 = false, int TM = 1, bool ARELU = false>
__global__ __launch_bounds__(256, 2) void gemm_pers_kernel(const vmvm_gemm_desc p) {
  static_assert(TM == 1 || (TM == 2 && AK && BKM && !FP8 && !(F & EF_COLSUM)), "the 256x64 tile serves k-major x k-major operands");
  static_assert(!ARELU || (F16 && AK), "A-operand ReLU: fp16 k-major build");
  constexpr int BM = TM == 2 ? 256 : 128, BN = TM == 2 ? 64 : 128;        // (shadow the file-scope square tile)
  constexpr int NA = BM / 32, NB = BN / 32;                                 // 16-byte DMA requests per thread per operand tile
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = TM == 2 ? wave : (wave >> 1), wn = TM == 2 ? 0 : (wave & 1);
  const int M = p.M, N = p.N, K = p.K;
  const int nbn = (N + BN - 1) / BN, nbm = (M + BM - 1) / BM;
  const int nb = nbm * nbn;
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nbt = nb * S;
  const int nk_all = (K + BK - 1) / BK;
  const int per = (nk_all + S - 1) / S;
  // this XCD's contiguous run of logical work items, shared round-robin by its resident workgroups
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;       // gridDim.x is a multiple of 8
  const int q2 = nbt >> 3, rr2 = nbt & 7;
  const int x_start = (xcd < rr2) ? xcd * (q2 + 1) : rr2 * (q2 + 1) + (xcd - rr2) * q2;
  const int x_cnt = q2 + (xcd < rr2 ? 1 : 0);
  const u16* A = reinterpret_cast<const u16*>(p.A);
  const u16* B = reinterpret_cast<const u16*>(p.B);
  const size_t bytesA = (size_t)(AK ? M : K) * p.lda * 2, bytesB = (size_t)(BKM ? N : K) * p.ldb * 2;
  // conv mode: the descriptor base sits (W+1) pixels BEFORE the activation so every tap offset is non-negative; the rows that
  // would read in front of / behind the tensor are exactly the ones masked out below
  const int cC = CONV ? K / p.conv_taps : 0;                               // input channels
  const unsigned conv_shift = CONV ? (unsigned)((p.conv_w + 1) * cC * 2) : 0u;
  const unsigned a_records = (unsigned)bytesA + 2u * conv_shift;
  const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(A) - (conv_shift >> 1), 0, (int)a_records, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(B), 0, (int)bytesB, 0x00020000);
  const int r = lane & 15, g = lane >> 4;
  EpiCtx ec;
  ec.has_drop = p.dropout_p > 0.f; ec.thr = dropout_threshold(p.dropout_p);
  ec.keep_scale = ec.has_drop ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  ec.S = S; ec.M = M; ec.N = N;

  auto decode = [&](int w, int& slice, int& m0, int& n0, int& kt0, int& nk) {
    const int logical = x_start + w;
    slice = logical / nb;
    int tm, tn;
    raster(logical - slice * nb, nbm, nbn, vmvm_hook::GM, tm, tn);
    m0 = tm * BM; n0 = tn * BN;
    kt0 = slice * per;
    nk = (kt0 + per < nk_all) ? kt0 + per : nk_all;
  };
  // DMA requests of a K tile: the per-lane source offsets only depend on the tile origin, so they are computed once per tile
  // (vo*) and the K position travels in the instruction's SCALAR offset; the LDS destination is a scalar too.  (Per-request
  // vector address arithmetic was ~30 of the ~100 non-MFMA issue slots of a K step.)
  const int wave_base_s = __builtin_amdgcn_readfirstlane(tid & ~63);
  const unsigned kstepA = (unsigned)((AK ? BK : BK * p.lda) * 2), kstepB = (unsigned)((BKM ? BK : BK * p.ldb) * 2);
  auto tile_offsets = [&](int m0, int n0, unsigned (&vA)[NA], unsigned (&vB)[NB], int (&py)[NA], int (&px)[NA]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int u = tid + i * 256;
      py[i] = 0; px[i] = 0;
      if (CONV) {                                           // pixel (y, x) of this request's row; rows >= M are parked outside the image
        const int pix = m0 + (u >> 3);
        const int q = pix / p.conv_w;
        px[i] = pix - q * p.conv_w; py[i] = (pix < M) ? q % p.conv_h : -4;
      }
      if (AK) { const int row = u >> 3, cs = u & 7; vA[i] = (unsigned)(((size_t)(m0 + row) * p.lda + (cs ^ kswz<false>(row)) * 8) * 2); }
      else { const int krow = u >> 4, unit = u & 15; const int slot = ((unit >> 1) ^ swz_m(krow)) & 7;
             vA[i] = (unsigned)(((size_t)krow * p.lda + m0 + (slot * 2 + (unit & 1)) * 8) * 2); }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int u = tid + i * 256;
      if (BKM) { const int row = u >> 3, cs = u & 7; vB[i] = (unsigned)(((size_t)(n0 + row) * p.ldb + (cs ^ kswz<true>(row)) * 8) * 2); }
      else { const int krow = u >> 4, unit = u & 15; const int slot = ((unit >> 1) ^ swz_m(krow)) & 7;
             vB[i] = (unsigned)(((size_t)krow * p.ldb + n0 + (slot * 2 + (unit & 1)) * 8) * 2); }
    }
  };
  auto issue = [&](const unsigned* vA, const unsigned* vB, const int* py, const int* px, int kt, int buf) {
    typedef __attribute__((address_space(3))) void lds_void;
    unsigned char* st = smem + buf * STAGE_BYTES + wave_base_s * 16;
    unsigned sa = (unsigned)kt * kstepA;
    const unsigned sb = (unsigned)kt * kstepB;
    int dy = 0, dx = 0;
    if (CONV) {                                             // K tile kt = tap * (C/64) + channel block
      const int cpt = cC >> 6, tap = kt / cpt, cb = kt - tap * cpt;
      dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1;
      sa = (unsigned)(((dy + 1) * p.conv_w + (dx + 1)) * cC + cb * 64) * 2u;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      unsigned vo = vA[i];
      if (CONV) vo = ((unsigned)(py[i] + dy) < (unsigned)p.conv_h && (unsigned)(px[i] + dx) < (unsigned)p.conv_w) ? vo : a_records;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_void*)(st + i * 4096), 16, vo, sa, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const unsigned vo = vB[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_void*)(st + A_BYTES + i * 4096), 16, vo, sb, 0, 0);
    }
  };

  int w = li;
  if (w >= x_cnt) return;
  int slice, m0, n0, kt0, nk;
  decode(w, slice, m0, n0, kt0, nk);
  vmvm_hook::GemmTimeline tl;                           // (probe builds: per-CU stamps / start-up staggers; nothing here)
  tl.init(smem, tid);
  vmvm_hook::gemm_stagger(smem, tid, li, nk - kt0);
  unsigned it = 0;                                      // running K-tile counter -> LDS buffer parity
  unsigned voA[NA], voB[NB], nvoA[NA], nvoB[NB];
  int pyA[NA], pxA[NA], npyA[NA], npxA[NA];
  tile_offsets(m0, n0, voA, voB, pyA, pxA);
  issue(voA, voB, pyA, pxA, kt0, it & 1);
  while (true) {
    tl.stamp(0, tid);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fused column sum (bias gradient): on the first N tile the wn == 0 waves also multiply their A fragments with an all-ones
    // operand; every row of that product is sum_k A(m,k) for the lane's column m
    f32x4 cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = (F & EF_COLSUM) && p.colsum && n0 == 0 && wn == 0;
    const int wn_ = w + per_xcd;
    const bool more = wn_ < x_cnt;
    int nslice = 0, nm0 = 0, nn0 = 0, nkt0 = 0, nnk = 0;
    if (more) { decode(wn_, nslice, nm0, nn0, nkt0, nnk); tile_offsets(nm0, nn0, nvoA, nvoB, npyA, npxA); }
    // Epilogue operands first: row bookkeeping and the per-element loads of the tile (saved activation / code, residual) go out BEFORE the K
    // loop -- they only depend on the tile origin, and the first K step's vmcnt(0) (which waits for the operand DMA anyway) covers
    // their latency; issued after the loop they queued behind the next tile's prefetch and stalled every tile's first store.
    bool rvalid[4]; long rdst[4]; float rrs[4];
#pragma clang loop unroll(full)
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + r;
      bool valid = m < M;
      long dst = m;
      if ((F & EF_MAP) && valid && p.row_map) {
        const int mapped = p.row_map[m % p.map_len];
        valid = mapped >= 0;
        dst = (long)mapped + (long)(m / p.map_len) * p.map_stride;
      }
      rvalid[i] = valid; rdst[i] = dst;
      rrs[i] = ((F & EF_RS) && valid && p.row_scale) ? p.row_scale[(m + p.scale_row0) / p.rows_per_scale] : 1.0f;
    }
    uint4 auxv[4][2], resv[4][2];
#pragma clang loop unroll(full)
    for (int jb = 0; jb < 2; ++jb) {
      const int n = n0 + wn * 64 + jb * 32 + g * 8;
      const bool full = n + 8 <= N;
#pragma clang loop unroll(full)
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r;
        auxv[i][jb] = make_uint4(0, 0, 0, 0); resv[i][jb] = make_uint4(0, 0, 0, 0);
        if ((F & (EF_ACT3 | EF_ACT24)) && p.act >= 3 && rvalid[i] && full) {
          if (!F16 && p.aux_code8) {
            const uint2 c8 = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(p.aux) + (size_t)m * p.ldaux + n);
            auxv[i][jb] = make_uint4(c8.x, c8.y, 0, 0);
          } else {
            auxv[i][jb] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p.aux) + (size_t)m * p.ldaux + n);
          }
        }
        if ((F & EF_RESID) && p.resid && rvalid[i] && full)
          resv[i][jb] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p.resid) + (size_t)rdst[i] * p.ldr + n);
      }
    }
    for (int kt = kt0; kt < nk; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                   // K tile `kt` landed for everyone; everyone left the other buffer
      const int cur = it & 1;
      const unsigned char* la = smem + cur * STAGE_BYTES;
      const unsigned char* lb = la + A_BYTES;
      if (FP8) {
        typedef __attribute__((ext_vector_type(8))) int i32x8;
        i32x8 fa8[4], fb8[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = (wm * 4 + i) * 16 + r, sw = kswz<false>(row);
          const uint4 lo = *reinterpret_cast<const uint4*>(la + row * 128 + (((2 * g) ^ sw) << 4));
          const uint4 hi = *reinterpret_cast<const uint4*>(la + row * 128 + (((2 * g + 1) ^ sw) << 4));
          fa8[i] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wn * 64 + (j >> 1) * 32 + 8 * (r >> 2) + 4 * (j & 1) + (r & 3), sw = kswz<true>(row);
          const uint4 lo = *reinterpret_cast<const uint4*>(lb + row * 128 + (((2 * g) ^ sw) << 4));
          const uint4 hi = *reinterpret_cast<const uint4*>(lb + row * 128 + (((2 * g + 1) ^ sw) << 4));
          fb8[j] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
        if (kt + 1 < nk) issue(voA, voB, pyA, pxA, kt + 1, cur ^ 1);
        else if (more) issue(nvoA, nvoB, npyA, npxA, nkt0, cur ^ 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb8[j], fa8[i], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        ++it;
        continue;
      }
      // software pipeline over the two 32-wide k-slices: slice 0's fragment reads go out right after the barrier (ahead of the
      // next tile's DMA requests), slice 1's reads are issued before slice 0's MFMAs so their LDS latency sits under the MFMAs
      bf16x8 fa[2][4], fb[2][4];
      s16x4 alo[2][4], ahi[2][4], blo[2][4], bhi[2][4];
      auto load_set = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (AK) {
            fa[s][i] = read_frag<true, true>(la, wm * 4 + i, s, lane);
            if (ARELU) fa[s][i] = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(f16x8, fa[s][i]), f16x8{0, 0, 0, 0, 0, 0, 0, 0}));
          }
          else tr_issue_frag(la, wm * 4 + i, s, lane, alo[s][i], ahi[s][i]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BKM) fb[s][j] = read_frag_bperm<true>(lb, wn * 64 + (j >> 1) * 32, j & 1, s, lane);
          else tr_issue_bperm(lb, wn * 64 + (j >> 1) * 32, j & 1, s, lane, blo[s][j], bhi[s][j]);
        }
      };
      auto finish_set = [&](int s) {                      // transposing (asm) reads: wait + assemble; k-major reads: compiler-managed
        if (!AK) {
          tr_wait8(alo[s], ahi[s]);
#pragma unroll
          for (int i = 0; i < 4; ++i) fa[s][i] = tr_cat(alo[s][i], ahi[s][i]);
        }
        if (!BKM) {
          tr_wait8(blo[s], bhi[s]);
#pragma unroll
          for (int j = 0; j < 4; ++j) fb[s][j] = tr_cat(blo[s][j], bhi[s][j]);
        }
      };
      auto mfma_set = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = F16 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fb[s][j]), __builtin_bit_cast(f16x8, fa[s][i]), acc[i][j], 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);
        if ((F & EF_COLSUM) && do_cs) {
          typedef __attribute__((ext_vector_type(8))) short s16x8;
          const s16x8 o8 = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};      // bf16 1.0
#pragma unroll
          for (int i = 0; i < 4; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, o8), fa[s][i], cs[i], 0, 0, 0);
        }
      };
      load_set(0);
      if (kt + 1 < nk) issue(voA, voB, pyA, pxA, kt + 1, cur ^ 1);
      else if (more) issue(nvoA, nvoB, npyA, npxA, nkt0, cur ^ 1);     // cross-tile prefetch: overlaps this tile's last MFMAs + epilogue
      finish_set(0);
      load_set(1);
      mfma_set(0);
      finish_set(1);
      mfma_set(1);
      ++it;
    }
    ec.slice = slice;
    tl.stamp(1, tid);
    // (An LDS-staged, 16-byte-per-lane coalesced epilogue was measured here: correct but 1.4-1.7x SLOWER on every shape --
    //  two extra barriers and an LDS round trip per tile cost more than the 32-byte store fragments; kept direct.)
    if ((F & EF_COLSUM) && do_cs && g == 0) {            // lanes 0-15: column m = m0 + wm*64 + i*16 + r (all 16 rows of the product are equal)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r;
        if (m < M) {
          const float t = p.colsum_scale != 0.f ? cs[i][0] * p.colsum_scale : cs[i][0];
          float* parts = colsum_parts(p, p.splitk);          // split problem with room behind the slabs: fixed-order sum in the reduce pass
          if (parts) parts[(size_t)slice * M + m] = t; else atomicAdd(p.colsum + m, t);
        }
      }
    }
    if constexpr ((F & EF_ARGMAX) != 0) {
      // (maximum, column) of every row over this wave's 64 columns; ties keep the smaller column (torch.argmax's answer)
      float bzm[2][8];
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        const int n = n0 + wn * 64 + jb * 32 + g * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bzm[jb][e] = (p.bias && n + 8 <= N) ? p.bias[n + e] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + r;
        float best = -__builtin_inff();
        int bi = 0;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int n = n0 + wn * 64 + jb * 32 + g * 8 + e;
            const float v = acc[i][2 * jb + (e >> 2)][e & 3] + bzm[jb][e];
            if (n < N && v > best) { best = v; bi = n; }
          }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
          const float ov = __shfl_xor(best, off, 64);
          const int oi = __shfl_xor(bi, off, 64);
          if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (g == 0 && m < M) {
          float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + 2 * ((n0 + wn * 64) >> 6);
          *reinterpret_cast<float2*>(c) = make_float2(best, __int_as_float(bi));
        }
      }
    } else {
    // epilogue: math + stores (the row bookkeeping and the big loads of the tile were issued in front of the K loop; the bias is a
    // cached 256-byte read that is not worth 16 registers across the main loop)
    float bz[2][8];
#pragma clang loop unroll(full)
    for (int jb = 0; jb < 2; ++jb) {
      const int n = n0 + wn * 64 + jb * 32 + g * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) bz[jb][e] = 0.f;
      if ((F & EF_BIAS) && p.bias && n + 8 <= N) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
        bz[jb][0] = b0.x; bz[jb][1] = b0.y; bz[jb][2] = b0.z; bz[jb][3] = b0.w; bz[jb][4] = b1.x; bz[jb][5] = b1.y; bz[jb][6] = b1.z; bz[jb][7] = b1.w;
      }
    }
    // 16-bit outputs of the bf16 build leave through a wave-private 4 KiB LDS tile (no barrier: the tile sits behind the operand
    // ring and belongs to one wave): in the MFMA layout a store instruction writes 16 rows x 64 bytes -- 16 half lines, each paid
    // for on the address path -- re-tiled it writes 8 rows x 128 bytes, whole lines.  Two passes of 32 rows per wave tile.
    // Measured per class inside a step (tools/probe/ab_shapes.sh, profiles/r02_gemm_pers_retile_ab.txt): the bias / GELU / GELU'
    // classes gain 5-15 % (two outputs, or epilogue-bound short-K shapes); the plain and the residual classes LOSE 2-10 % (their tiles
    // are main-loop bound and the LDS round trip only adds latency), so they keep the direct stores.
    constexpr bool RT = TM == 1 && !F16 && !FP8 &&
                        (F == (EF_BIAS | EF_COLSCALE | EF_RS) || F == (EF_BIAS | EF_ACT1 | EF_RS) || F == (EF_ACT3 | EF_RS));
    if constexpr (RT) {
      unsigned char* stg = smem + 2 * STAGE_BYTES + wave * 4096;
      const int sr = lane >> 3, sg = lane & 7;             // store layout: row inside an 8-row group, 16-byte column group
      const bool has_pre = (F & EF_ACT1) && p.act == 1 && p.C2;
      const bool code8 = (F & (EF_ACT1 | EF_ACT3)) && p.aux_code8 != 0;     // 8-bit GELU' codes in C2 (act 1) / aux (act 3)
      const bool c8 = has_pre && code8;
      const int n_s = n0 + wn * 64 + sg * 8;
#pragma clang loop unroll(full)
      for (int hb = 0; hb < 2; ++hb) {
        uint4 o[2][2], pr[2][2];
#pragma clang loop unroll(full)
        for (int ii = 0; ii < 2; ++ii) {
          const int i = hb * 2 + ii;
          const int m = m0 + wm * 64 + i * 16 + r;
#pragma clang loop unroll(full)
          for (int jb = 0; jb < 2; ++jb) {
            const int n = n0 + wn * 64 + jb * 32 + g * 8;
            float v[8] = {acc[i][2 * jb][0], acc[i][2 * jb][1], acc[i][2 * jb][2], acc[i][2 * jb][3],
                          acc[i][2 * jb + 1][0], acc[i][2 * jb + 1][1], acc[i][2 * jb + 1][2], acc[i][2 * jb + 1][3]};
            o[ii][jb] = make_uint4(0, 0, 0, 0); pr[ii][jb] = make_uint4(0, 0, 0, 0);
            if (rvalid[i] && n < N) {
              if constexpr (vmvm_hook::EPI & 2) {
                o[ii][jb] = pack8<F16>(v);
                pr[ii][jb] = make_uint4(gelu_code4(v[0], v[1], v[2], v[3]), gelu_code4(v[4], v[5], v[6], v[7]), 0, 0);
              } else {
                if (code8) epi_math8<(F | EF_CODE8), F16>(p, ec, v, m, n, rrs[i], bz[jb], auxv[i][jb], resv[i][jb], o[ii][jb], pr[ii][jb]);
                else epi_math8<F, F16>(p, ec, v, m, n, rrs[i], bz[jb], auxv[i][jb], resv[i][jb], o[ii][jb], pr[ii][jb]);
              }
            }
          }
        }
        // store-side rows of this half: 8 * s4 + sr
        bool svalid[4]; long sdst[4]; int srow[4];
#pragma clang loop unroll(full)
        for (int s4 = 0; s4 < 4; ++s4) {
          const int ms = m0 + wm * 64 + hb * 32 + s4 * 8 + sr;
          bool valid = ms < M && n_s < N;
          long dst = ms;
          if ((F & EF_MAP) && valid && p.row_map) {
            const int mapped = p.row_map[ms % p.map_len];
            valid = mapped >= 0;
            dst = (long)mapped + (long)(ms / p.map_len) * p.map_stride;
          }
          svalid[s4] = valid; sdst[s4] = dst; srow[s4] = ms;
        }
#pragma clang loop unroll(full)
        for (int pass = 0; pass < 2; ++pass) {
          if (pass == 1 && !has_pre) break;
          if (pass == 1 && c8) {                          // 8-bit GELU' codes: 64-byte rows, 16 rows per store instruction (as in gemm_pp.h)
            const int c8r = lane >> 2, c8g = lane & 3;
#pragma clang loop unroll(full)
            for (int ii = 0; ii < 2; ++ii)
#pragma clang loop unroll(full)
              for (int jb = 0; jb < 2; ++jb) {
                const int row = ii * 16 + r;
                *reinterpret_cast<uint2*>(stg + row * 64 + (((jb * 4 + g) ^ (((row >> 2) & 3) << 1)) << 3)) = make_uint2(pr[ii][jb].x, pr[ii][jb].y);
              }
            uint4 t2[2];
#pragma clang loop unroll(full)
            for (int s2 = 0; s2 < 2; ++s2) {
              const int row = s2 * 16 + c8r;
              t2[s2] = *reinterpret_cast<const uint4*>(stg + row * 64 + (((c8g * 2) ^ (((row >> 2) & 3) << 1)) << 3));
            }
#pragma clang loop unroll(full)
            for (int s2 = 0; s2 < 2; ++s2) {
              const int ms = m0 + wm * 64 + hb * 32 + s2 * 16 + c8r, n_c = n0 + wn * 64 + c8g * 16;
              if constexpr (vmvm_hook::EPI & 1) { asm volatile("" ::"v"(t2[s2].x), "v"(t2[s2].y), "v"(t2[s2].z), "v"(t2[s2].w)); continue; }
              if (ms < M && n_c + 16 <= N)
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.C2) + (size_t)ms * p.ldc2 + n_c) = t2[s2];
              else if (ms < M && n_c + 8 <= N)
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.C2) + (size_t)ms * p.ldc2 + n_c) = make_uint2(t2[s2].x, t2[s2].y);
            }
            break;
          }
#pragma clang loop unroll(full)
          for (int ii = 0; ii < 2; ++ii)
#pragma clang loop unroll(full)
            for (int jb = 0; jb < 2; ++jb) {
              const int row = ii * 16 + r;
              *reinterpret_cast<uint4*>(stg + row * 128 + (((jb * 4 + g) ^ (row & 7)) << 4)) = pass ? pr[ii][jb] : o[ii][jb];
            }
          uint4 t[4];
#pragma clang loop unroll(full)
          for (int s4 = 0; s4 < 4; ++s4) t[s4] = *reinterpret_cast<const uint4*>(stg + (s4 * 8 + sr) * 128 + ((sg ^ sr) << 4));
#pragma clang loop unroll(full)
          for (int s4 = 0; s4 < 4; ++s4) {
            if (!svalid[s4]) continue;
            if constexpr (vmvm_hook::EPI & 1) { asm volatile("" ::"v"(t[s4].x), "v"(t[s4].y), "v"(t[s4].z), "v"(t[s4].w)); continue; }
            if (pass == 0) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.C) + (size_t)sdst[s4] * p.ldc + n_s) = t[s4];
            else *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.C2) + (size_t)srow[s4] * p.ldc2 + n_s) = t[s4];
          }
        }
      }
    } else {
#pragma clang loop unroll(full)
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + r;
#pragma clang loop unroll(full)
      for (int jb = 0; jb < 2; ++jb) {
        const int n = n0 + wn * 64 + jb * 32 + g * 8;
        float v[8] = {acc[i][2 * jb][0], acc[i][2 * jb][1], acc[i][2 * jb][2], acc[i][2 * jb][3],
                      acc[i][2 * jb + 1][0], acc[i][2 * jb + 1][1], acc[i][2 * jb + 1][2], acc[i][2 * jb + 1][3]};
        if (FP8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
        }
        if (rvalid[i] && n < N) epi_store8<F, F16>(p, ec, v, m, rdst[i], n, rrs[i], N - n, bz[jb], auxv[i][jb], resv[i][jb]);
      }
    }
    }
    }
    tl.stamp(2, tid);
    tl.next_tile();
    if (!more) break;
    w = wn_; slice = nslice; m0 = nm0; n0 = nn0; kt0 = nkt0; nk = nnk;
#pragma unroll
    for (int i = 0; i < NA; ++i) { voA[i] = nvoA[i]; pyA[i] = npyA[i]; pxA[i] = npxA[i]; }
#pragma unroll
    for (int i = 0; i < NB; ++i) voB[i] = nvoB[i];
  }
}


template <bool AK, bool BKM, int F>
int launch_pers_f(const vmvm_gemm_desc& d, hipStream_t st) {
  const int items = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN) * (d.splitk > 1 ? d.splitk : 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pers_kernel<AK, BKM, F>), hipFuncAttributeMaxDynamicSharedMemorySize, PERS_SMEM_BYTES);
    attr_done = true;
  }
  int grid = 2 * vmvm_usable_cus(d.reserve_cus);        // 2 workgroups per CU (64 KiB LDS each), multiple of 8
  if (vmvm_hook::ONE_WG) {     // (probe builds: ONE workgroup per CU through an LDS request no second workgroup fits beside; false here)
    grid = vmvm_usable_cus(d.reserve_cus);
    if (items < grid) grid = ((items + 7) / 8) * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pers_kernel<AK, BKM, F>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipLaunchKernelGGL((gemm_pers_kernel<AK, BKM, F>), dim3(grid), dim3(256), 100 << 10, st, d);
    VMVM_CHECK_LAUNCH();
    return VMVM_OK;
  }
  if (items < grid) grid = ((items + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_pers_kernel<AK, BKM, F>), dim3(grid), dim3(256), PERS_SMEM_BYTES, st, d);
  VMVM_CHECK_LAUNCH();
  if (d.splitk > 1 && d.workspace) {
    launch_splitk_reduce(d, d.splitk, st);
    VMVM_CHECK_LAUNCH();
  }
  return VMVM_OK;
}

// epilogue features this descriptor needs (see EF_*)
static int epi_need(const vmvm_gemm_desc& d) {
  int f = 0;
  if (d.bias) f |= EF_BIAS;
  if (d.col_scale_n > 0) f |= EF_COLSCALE;
  if (d.act == 1) f |= EF_ACT1;
  if (d.act == 2 || d.act == 4) f |= EF_ACT24;
  if (d.act == 3) f |= EF_ACT3;
  if (d.row_scale) f |= EF_RS;
  if (d.dropout_p > 0.f) f |= EF_DROP;
  if (d.resid) f |= EF_RESID;
  if (d.splitk > 1) f |= EF_SPLIT;
  if (d.out_fp32) f |= EF_F32;
  if (d.row_map) f |= EF_MAP;
  if (d.N & 7) f |= EF_EDGE4;
  return f;
}

// The instantiations cover the epilogue classes of the training step (plain; qkv = bias + q scale; fc1 = bias + GELU + saved
// pre-activation; fc2 dgrad = GELU' x saved; proj / fc2 = bias + dropout + residual (+ window un-gather); wgrad = f32 split-K)
// and fall back to the all-features build for anything else.
// fp16 / implicit-convolution builds (frozen dVAE tokenizer): one epilogue mask covers its four GEMM forms
// (3x3: bias + ReLU;  1x1: bias, post_gain column scale, residual, bf16 / f32 output;  last 1x1: bias + fused arg-max)
constexpr int EF_TEACHER = EF_BIAS | EF_COLSCALE | EF_ACT24 | EF_RESID | EF_F32;
constexpr int EF_TEACHER_CONV = EF_BIAS | EF_ACT24;
constexpr int EF_TEACHER_ARGMAX = EF_BIAS | EF_ARGMAX;
constexpr int EF_TEACHER_RES = EF_BIAS | EF_COLSCALE | EF_RESID;      // id_path / conv_4: the two 1x1 forms inside a block (16-bit out, no activation)
template <bool CONV, int F, int TM, bool ARELU>
int launch_pers_teacher(const vmvm_gemm_desc& d, hipStream_t st) {
  constexpr int BM_ = TM == 2 ? 256 : 128, BN_ = TM == 2 ? 64 : 128;
  constexpr int SMEM_ = 2 * (BM_ + BN_) * BK * 2;
  const int items = ((d.M + BM_ - 1) / BM_) * ((d.N + BN_ - 1) / BN_);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pers_kernel<true, true, F, true, CONV, false, TM, ARELU>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_);
    attr_done = true;
  }
  int grid = 2 * vmvm_usable_cus(d.reserve_cus);
  if (items < grid) grid = ((items + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_pers_kernel<true, true, F, true, CONV, false, TM, ARELU>), dim3(grid), dim3(256), SMEM_, st, d);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

// fp8 build (forward Linear layers of config 5): bias, GELU (+ saved pre-activation), ReLU, residual, bf16 / f32 output
constexpr int EF_FP8 = EF_BIAS | EF_ACT1 | EF_ACT24 | EF_RESID | EF_F32;
int launch_pers_fp8(const vmvm_gemm_desc& d, hipStream_t st) {
  const int items = ((d.M + BM - 1) / BM) * ((d.N + BN - 1) / BN);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pers_kernel<true, true, EF_FP8, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    attr_done = true;
  }
  int grid = 2 * vmvm_usable_cus(d.reserve_cus);
  if (items < grid) grid = ((items + 7) / 8) * 8;
  hipLaunchKernelGGL((gemm_pers_kernel<true, true, EF_FP8, false, false, true>), dim3(grid), dim3(256), SMEM_BYTES, st, d);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

template <bool AK, bool BKM>
int launch_pers(const vmvm_gemm_desc& d, hipStream_t st) {
  const int need = epi_need(d);
#define TRY_EPI(MASK) if ((need & ~(MASK)) == 0) return launch_pers_f<AK, BKM, (MASK)>(d, st)
  if constexpr (AK && BKM) {
    TRY_EPI(0);
    TRY_EPI(EF_BIAS | EF_COLSCALE | EF_RS);
    TRY_EPI(EF_BIAS | EF_ACT1 | EF_RS);
    TRY_EPI(EF_ACT3 | EF_RS);
    TRY_EPI(EF_BIAS | EF_RESID | EF_DROP | EF_RS | EF_MAP);
  } else if constexpr (!AK && !BKM) {
    if (d.colsum) { if ((need & ~(EF_SPLIT | EF_F32)) == 0) return launch_pers_f<AK, BKM, (EF_SPLIT | EF_F32 | EF_COLSUM)>(d, st); }
    else TRY_EPI(EF_SPLIT | EF_F32);
  }
#undef TRY_EPI
  return launch_pers_f<AK, BKM, EF_ALL>(d, st);
}

}  // namespace

// Tile counts, grids and split-K plans below are 32-bit: a problem with more than 2^24 output tiles of 128 x 128 (M * N > 2.7e11) is
// refused up front (found by the host-side sanitizer sweep, tools/cabi_validation.py: 2^30 x 2^30 overflowed the tile count to 0 and
// the split plan divided by it).
static bool gemm_tiles_ok(int M, int N) {
  return (int64_t)((M + BM - 1) / BM) * (int64_t)((N + BN - 1) / BN) <= (int64_t)1 << 24;
}

extern "C" int vmvm_gemm_bf16(const vmvm_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->B || !d->C) return VMVM_EINVAL;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0) return VMVM_EINVAL;
  if ((d->N & 3) || (d->lda & 7) || (d->ldb & 7) || (d->ldc & 3)) return VMVM_EINVAL;
  if (d->a_relu && !d->in_fp16) return VMVM_ENOSUPPORT;
  if (d->aux_code8) {                                   // 8-bit GELU' code: an epilogue form of the 128x128 persistent and the 256x256 ping-pong kernels
    if ((d->in_fp8 && d->act != 1) || d->in_fp16 || d->conv_taps || d->out_fp32 || !d->a_kmajor || !d->b_kmajor || (d->variant != 0 && d->variant != 6 && d->variant != 7)) return VMVM_ENOSUPPORT;
    if ((d->K % (d->in_fp8 ? 2 * BK : BK)) || (d->N & 7) || (d->act != 1 && d->act != 3)) return VMVM_EINVAL;
    if (d->act == 1 && d->C2 && (d->ldc2 & 7)) return VMVM_EINVAL;
    if (d->act == 3 && (d->ldaux & 7)) return VMVM_EINVAL;
  }
  if (d->in_fp8) {
    // fp8 (OCP e4m3) operands: k-major both, whole 128-element K tiles, epilogue features within EF_FP8
    if (!d->a_kmajor || !d->b_kmajor || d->in_fp16 || d->conv_taps) return VMVM_ENOSUPPORT;
    if ((d->K % 128) || (d->N & 7) || (d->lda & 15) || (d->ldb & 15) || d->lda < d->K || d->ldb < d->K) return VMVM_EINVAL;
    if (d->accumulate || d->colsum || d->splitk > 1 || (d->row_scale && d->rows_per_scale <= 0)) return VMVM_ENOSUPPORT;
    if ((size_t)d->M * d->lda >= 0x7fffffffull || (size_t)d->N * d->ldb >= 0x7fffffffull) return VMVM_ENOSUPPORT;
    vmvm_gemm_desc d8 = *d;
    d8.K = d->K / 2; d8.lda = d->lda / 2; d8.ldb = d->ldb / 2;      // the kernels count 2-byte units
    d8.splitk = 1;
    // long reductions on the 256x256 ping-pong main loop (v_mfma_scale_f32_32x32x64_f8f6f4): its epilogue classes include the row
    // scale / dropout / q-scale forms; measured against the 128x128 fp8 build in tools/gpu_check.py check_gemm_fp8
    {
      const int t256 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
      const bool pays = d->K >= 2048 && d->N >= 512 && t256 >= 128;
      if (!d->out_fp32 && !d->row_map && d->act != 2 && (d->variant == 7 || (d->variant == 0 && pays))) {
        const int rc_ = vmvm_gemm_pp_fp8(d8, epi_need(d8), reinterpret_cast<hipStream_t>(stream));
        if (rc_ != VMVM_ENOSUPPORT) return rc_;
      }
    }
    if (d->row_scale || d->dropout_p > 0.f || d->row_map || d->col_scale_n || (d->act != 0 && d->act != 1 && d->act != 2)) return VMVM_ENOSUPPORT;
    return launch_pers_fp8(d8, reinterpret_cast<hipStream_t>(stream));
  }
  if (d->in_fp16 || d->conv_taps) {
    // fp16 / implicit 3x3 convolution builds of the persistent kernel (frozen dVAE tokenizer): k-major operands, whole K tiles,
    // epilogue features within EF_TEACHER
    const int taps = d->conv_taps;
    if (!d->in_fp16 || !d->a_kmajor || !d->b_kmajor || (taps != 0 && taps != 9)) return VMVM_ENOSUPPORT;
    if ((d->K % BK) || (d->N & 7) || d->row_scale || d->dropout_p > 0.f || d->row_map || d->C2 || d->accumulate || d->colsum ||
        (d->act != 0 && d->act != 2 && d->act != 5) || d->splitk > 1) return VMVM_ENOSUPPORT;
    if (d->act == 5 && (taps || !d->out_fp32 || (d->N & 63) || d->ldc < 2 * (d->N >> 6) || d->resid || d->col_scale_n)) return VMVM_EINVAL;
    if (taps && (d->resid || d->col_scale_n || d->out_fp32)) return VMVM_ENOSUPPORT;
    const int Cin = taps ? d->K / taps : d->K;
    if (taps && ((d->K % taps) || (Cin % BK) || d->conv_h <= 0 || d->conv_w <= 0 || (d->M % (d->conv_h * d->conv_w)))) return VMVM_EINVAL;
    if (d->lda < Cin || d->ldb < d->K) return VMVM_EINVAL;
    const size_t bA = (size_t)d->M * d->lda * 2 + (taps ? (size_t)4 * (d->conv_w + 1) * Cin : 0), bB = (size_t)d->N * d->ldb * 2;
    if (bA >= 0x7fffffffull || bB >= 0x7fffffffull) return VMVM_ENOSUPPORT;       // 32-bit buffer offsets: the caller chunks the frames
    vmvm_gemm_desc dt = *d;
    dt.splitk = 1;
    hipStream_t st_ = reinterpret_cast<hipStream_t>(stream);
    const bool ar = dt.a_relu != 0;
    if (taps) {
      if (dt.N <= 64) return ar ? launch_pers_teacher<true, EF_TEACHER_CONV, 2, true>(dt, st_) : launch_pers_teacher<true, EF_TEACHER_CONV, 2, false>(dt, st_);
      return ar ? launch_pers_teacher<true, EF_TEACHER_CONV, 1, true>(dt, st_) : launch_pers_teacher<true, EF_TEACHER_CONV, 1, false>(dt, st_);
    }
    if (dt.act == 5) return ar ? launch_pers_teacher<false, EF_TEACHER_ARGMAX, 1, true>(dt, st_) : launch_pers_teacher<false, EF_TEACHER_ARGMAX, 1, false>(dt, st_);
    if (!ar && !dt.out_fp32 && dt.act == 0) return launch_pers_teacher<false, EF_TEACHER_RES, 1, false>(dt, st_);
    return ar ? launch_pers_teacher<false, EF_TEACHER, 1, true>(dt, st_) : launch_pers_teacher<false, EF_TEACHER, 1, false>(dt, st_);
  }
  if (!gemm_tiles_ok(d->M, d->N)) return VMVM_ENOSUPPORT;
  if (d->act < 0 || d->act > 4) return VMVM_EINVAL;             // (act = 5, the fused arg-max, exists in the fp16 teacher builds above only)
  // 16-byte chunks may straddle the logical extent as long as the row stride covers the round-up
  const int K8 = (d->K + 7) & ~7, M8 = (d->M + 7) & ~7, N8 = (d->N + 7) & ~7;
  if (d->a_kmajor ? (d->lda < K8) : (d->lda < M8)) return VMVM_EINVAL;
  if (d->b_kmajor ? (d->ldb < K8) : (d->ldb < N8)) return VMVM_EINVAL;
  if (d->accumulate && !d->out_fp32) return VMVM_EINVAL;
  if ((d->act == 3 || d->act == 4) && !d->aux) return VMVM_EINVAL;
  if (d->row_map && d->map_len <= 0) return VMVM_EINVAL;
  if (d->row_scale && d->rows_per_scale <= 0) return VMVM_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool tr = d->variant != 1;
  const vmvm_gemm_desc* d_in = d;                      // (the caller's workspace fields survive in here: the plan below may null dd.workspace)
  vmvm_gemm_desc dd = *d;
  d = &dd;
  // split-K: weight gradients have few output tiles and a very long reduction (tokens); spread the reduction over
  // ~4 workgroups per CU and combine with f32 atomics into the gradient accumulator.
  const bool plain_acc = dd.out_fp32 && dd.accumulate && !dd.bias && !dd.row_scale && !dd.act && !dd.resid && !dd.row_map &&
                         dd.dropout_p <= 0.f && dd.col_scale_n == 0;
  // tile choice: the 256^2 kernel (direct staging only) when the problem is large enough to fill the chip with big tiles
  const size_t bytesA = (size_t)(dd.a_kmajor ? dd.M : dd.K) * dd.lda * 2, bytesB = (size_t)(dd.b_kmajor ? dd.N : dd.K) * dd.ldb * 2;
  const bool direct = (dd.variant == 0 || dd.variant >= 3) && (dd.K % BK == 0) && bytesA < 0x7fffffffull && bytesB < 0x7fffffffull;
  const int nk_all_ = (dd.K + BK - 1) / BK;
  const int tiles_big = ((dd.M + GB - 1) / GB) * ((dd.N + GB - 1) / GB);
  const int tiles_small = ((dd.M + BM - 1) / BM) * ((dd.N + BN - 1) / BN);
  // Measured on MI355X (profiles/): with one 8-wave workgroup per CU the 256^2 kernel only wins for long reductions
  // (8192^3: 1074 vs 862 TF); at the step's K = 128..3072 two co-resident 128^2 workgroups hide each other's
  // prologue/epilogue better, so auto-dispatch takes the big tile only for K >= 4096 non-split problems.
  const bool pp_shape = dd.a_kmajor && dd.b_kmajor && !dd.out_fp32 && !dd.row_map && !(dd.N & 7) && dd.splitk <= 1;      // gemm_pp.h candidates (below)
  bool big = direct && !pp_shape && dd.M >= 1024 && dd.N >= 1024 && dd.K >= 4096 && !(plain_acc && dd.splitk != 1) && tiles_big >= 224;
  if (dd.variant == 3) big = false;
  if (dd.variant == 4) big = direct;
  const int tiles_p3 = ((dd.M + P3_BM - 1) / P3_BM) * ((dd.N + P3_BN - 1) / P3_BN);
  // The 3-stage 256x128 kernel (variant 5) and the 256^2 kernel (variant 4) are kept as tested alternatives; after the
  // persistent kernel got cross-tile prefetch and 16-byte stores it is faster than both on every shape of the step.
  bool p3 = direct && !big && dd.variant == 5;
  const int tiles = big ? tiles_big : (p3 ? tiles_p3 : tiles_small);
  // weight gradients with a long reduction on the 256x256 ping-pong main loop (gemm_pp.h, m/n-major operands through transposing
  // LDS reads): one round of (tile, K-slice) items over the 256 CUs, f32 slabs + reduce
  if (direct && plain_acc && !dd.a_kmajor && !dd.b_kmajor && (dd.variant == 0 || dd.variant == 7) && dd.splitk == 0 && dd.workspace &&
      dd.K >= 4096 && dd.M >= 256 && dd.N >= 256 && !(dd.M & 7) && !(dd.N & 7) &&
      ((long)dd.M * dd.N >= (1 << 20) || dd.variant == 7)) {       // measured: +7..25 % from 2048x512 up, a loss below (the slabs of a one-round plan are 67 MB whatever the shape)
    const int t256 = ((dd.M + 255) / 256) * ((dd.N + 255) / 256);
    // (round 6, measured and left alone: 192 or 128 target units -- the weight gradient on three quarters / half of the CUs for
    //  proportionally longer, 4/7 of the slab bytes -- give the same step, 105.2-105.7 ms at all three: the second stream has slack,
    //  the step is the main stream's chain)
    int s = 256 / t256;
    if (s > nk_all_ / 4) s = nk_all_ / 4;
    if (s < 1) s = 1;
    const int per = (nk_all_ + s - 1) / s;
    s = (nk_all_ + per - 1) / per;
    const size_t need_ws = (size_t)s * dd.M * dd.N * sizeof(float);
    if (s == 1 || need_ws <= (size_t)dd.workspace_bytes) {
      vmvm_gemm_desc dp = dd;
      dp.splitk = s;
      if (s == 1) dp.workspace = nullptr;
      const int rc_ = vmvm_gemm_pp(dp, epi_need(dp) | (s > 1 ? EF_SPLIT : 0), st);
      if (rc_ == VMVM_OK && s > 1) {
        launch_splitk_reduce(dp, s, st);
        VMVM_CHECK_LAUNCH();
      }
      if (rc_ != VMVM_ENOSUPPORT) return rc_;
    }
  }
  if (dd.splitk == 0) {
    dd.splitk = 1;
    if (plain_acc) {
      int s = (big ? 512 : (p3 ? 512 : (direct ? 512 : 1024))) / tiles;      // persistent kernel: 512 resident workgroups
      if (s > nk_all_ / 4) s = nk_all_ / 4;
      if (s > 1) dd.splitk = s;
    }
  } else if (dd.splitk > 1 && !plain_acc) {
    return VMVM_EINVAL;
  }
  if (dd.splitk > 1 && dd.workspace) {
    // every (slice, tile) must be written: slices whose K range is empty would leave garbage -> shrink S to the useful count
    const int nk_all = (dd.K + BK - 1) / BK;
    const int per = (nk_all + dd.splitk - 1) / dd.splitk;
    dd.splitk = (nk_all + per - 1) / per;
    const size_t need = (size_t)dd.splitk * dd.M * dd.N * sizeof(float);
    if ((size_t)dd.workspace_bytes < need || dd.splitk < 2) dd.workspace = nullptr;     // fall back to atomics
  }
  // direct-to-LDS staging needs whole 64-wide K tiles for k-major operands (an out-of-extent k chunk would read the next
  // columns, not zeros) and 32-bit byte offsets; everything else takes the register-staged path (variant 2 forces it).
  const bool pers = direct && !big && !p3 && (dd.variant == 6 || dd.variant == 0 || dd.variant == 7);
  if (dd.colsum) {
    if (dd.a_kmajor) return VMVM_ENOSUPPORT;
    const bool plain_wgrad = !dd.b_kmajor && dd.out_fp32 && !dd.bias && !dd.col_scale_n && !dd.act && !dd.row_scale && dd.dropout_p <= 0.f &&
                             !dd.resid && !dd.row_map && !(dd.N & 7);
    if (!(pers && plain_wgrad)) {                        // not the fused build: one separate pass over A (= X of the column sum)
      const int rc_ = vmvm_colsum_scaled(dd.A, dd.K, dd.M, dd.lda, dd.colsum_scale != 0.f ? dd.colsum_scale : 1.f, dd.colsum, d_in->workspace, d_in->workspace_bytes, stream);
      if (rc_) return rc_;
      dd.colsum = nullptr;
    }
  }
  // 256x256 ping-pong kernel (gemm_pp.h): large k-major x k-major problems with a long enough reduction.  Measured against the
  // 128x128 persistent kernel on MI355X (tools/probe/gemm_probe, profiles/r02_gemm_probe.txt): +8-10 % at (69120, 3072|2304, 768)
  // with plain / bias / GELU / GELU' epilogues, +3-13 % at K >= 2048, +6 % at 8192^3; slower for K <= 512 (the store tail of a
  // 256x256 tile is not hidden by a second workgroup) and when M*N gives fewer than ~3 rounds of 256 tiles at K < 2048.
  if (pers && pp_shape && (dd.variant == 7 || dd.variant == 0)) {
    const long tiles = (long)((dd.M + 255) / 256) * ((dd.N + 255) / 256);
    bool pays = dd.K >= 768 && dd.N >= 512 && (dd.K >= 2048 ? tiles >= 128 : tiles >= 1024);
    // without an epilogue the 256x256 tile's store tail is one bf16 output: it already wins from ~1.5 rounds of tiles (in-step A/B,
    // tools/scratch/force_pp_compare.sh: 50176 x 512 x 1536 108 -> 97 us, 200704 x 256 x 768 125 -> 114; 12544-row shapes lose)
    if (!pays && epi_need(dd) == 0 && dd.K >= 768 && dd.N >= 256 && tiles >= 384) pays = true;
    // GELU forward with its second output (saved pre-activation / 8-bit code): the 128x128 kernel's epilogue of one workgroup runs under
    // the main loop of the other one on the CU, and since its stores are re-tiled it beats the ping-pong kernel's exposed store tail at
    // K < 2048 (55296 x 3072 x 768: 367 -> 348 us, tools/scratch/no_pp_compare.sh).  Re-measured after the round-4 epilogue diet
    // (tools/scratch/fc1_pp_vs_pers.py): the two kernels are within 1-3 % on this class (69120 x 3072 x 768: 429-454 vs 426-440 us) and the
    // step does not tell them apart (108.8 ms either way) -- the rule stays.
    if ((epi_need(dd) & EF_ACT1) && dd.C2 && dd.K < 2048) pays = false;
    // Round quantisation of the one-workgroup-per-CU grid: 69120 x 768 is 810 tiles = 3.16 rounds of 256, i.e. a fourth round for 54
    // tiles.  For long reductions the rows that fill WHOLE rounds go to the ping-pong kernel and the remaining rows (a short second
    // launch) to the 128x128 kernel, whose 512 slots take them in one partial round of quarter-size tiles.  Measured
    // (tools/scratch/msplit_probe.py): 69120 x 768 x 3072 404 -> 361 us, x 2304 280 -> 265 us; a loss at K = 768 and when the
    // remainder is more than ~a quarter of a round (50176 x 512: 1.53 rounds).  Row-indexed epilogue operands move with the rows;
    // the dropout stream is indexed by the absolute element (8-element blocks), so its offset moves by m_split * N / 8.
    const int cus_ = vmvm_usable_cus(dd.reserve_cus);      // the ping-pong grid: one workgroup per usable CU (fewer while a collective is pending)
    // (explicit: the split below offsets A / C / resid by ROWS of 2-byte elements -- k-major operands, 16-bit output, no row map /
    //  column sums / accumulation.  pp_shape implies all of this today; the split must not depend on that staying true.)
    const bool split_ok = dd.a_kmajor && dd.b_kmajor && !dd.out_fp32 && !dd.row_map && !dd.colsum && !dd.accumulate && !dd.in_fp8 && !dd.conv_taps;
    // (round 5: also at 1024 <= K < 2048 and where the ping-pong kernel alone does not "pay" -- 47040 x 512 x 1536 is 368 tiles -- when the
    // remainder fits ONE round of quarter tiles: whole ping-pong rounds + one 128x128 round beat 2.9 rounds of the 128x128 kernel)
    const long rem_q_all = tiles > cus_ ? (long)((dd.M - (int)(((tiles / cus_) * cus_) / ((dd.N + 255) / 256)) * 256 + 127) / 128) * ((dd.N + 127) / 128) : (1L << 40);
    const bool split_fits = dd.K >= 1024 && dd.N >= 512 && rem_q_all <= 2 * cus_;
    if ((pays || split_fits) && split_ok && dd.variant == 0 && (dd.K >= 2048 || split_fits) && !dd.aux && !dd.C2 && tiles > cus_ && !(dd.N & 7)) {      // (round 5: row_scale moves with the rows through scale_row0)
      const int nbn_ = (dd.N + 255) / 256, nbm_ = (dd.M + 255) / 256;
      const int tm_split = (int)(((tiles / cus_) * cus_) / nbn_);
      const long rem_tiles = (long)(nbm_ - tm_split) * nbn_;
      // Round 5: the remainder may be up to ONE round of the 128x128 kernel's 2 x CUs slots (quarter-size tiles): 47040 x 512 x 2048 is
      // 368 tiles = 1.44 rounds, i.e. two ping-pong rounds (126 us, 784 TF against hipBLASLt's 1204); one whole round (32768 rows) + the
      // other 14272 rows as 448 quarter tiles in one 128x128 round is 48 + 36 us by the tile times of the two kernels.  (Round 3 allowed
      // a quarter of a round only: its counter-example, 50176 x 512, leaves 544 quarter tiles -- a second 128x128 round.)
      const long rem_q = (long)((dd.M - tm_split * 256 + 127) / 128) * ((dd.N + 127) / 128);
      if (tm_split > 0 && tm_split < nbm_ && (rem_tiles <= cus_ / 4 || rem_q <= 2 * cus_)) {
        const int m_split = tm_split * 256;
        vmvm_gemm_desc d1 = dd;
        d1.M = m_split;
        const int rc1 = vmvm_gemm_pp(d1, epi_need(d1), st);
        if (rc1 == VMVM_OK) {
          vmvm_gemm_desc d2 = dd;
          d2.M = dd.M - m_split;
          d2.A = reinterpret_cast<const char*>(dd.A) + (size_t)m_split * dd.lda * 2;
          d2.C = reinterpret_cast<char*>(dd.C) + (size_t)m_split * dd.ldc * 2;
          if (dd.resid) d2.resid = reinterpret_cast<const char*>(dd.resid) + (size_t)m_split * dd.ldr * 2;
          d2.offset = dd.offset + (uint64_t)m_split * (uint64_t)dd.N / 8;
          d2.scale_row0 = dd.scale_row0 + m_split;
          d2.variant = 6;
          return vmvm_gemm_bf16(&d2, stream);
        }
        if (rc1 != VMVM_ENOSUPPORT) return rc1;
      }
    }
    if (pays || dd.variant == 7) {
      const int rc_ = vmvm_gemm_pp(dd, epi_need(dd), st);
      if (rc_ != VMVM_ENOSUPPORT) return rc_;
      if (dd.variant == 7) return rc_;
    }
  }
  if (dd.aux_code8 && !pers) return VMVM_ENOSUPPORT;      // operands beyond the 32-bit DMA offsets
  if (pers) {
    if (d->a_kmajor && d->b_kmajor) return launch_pers<true, true>(*d, st);
    if (d->a_kmajor && !d->b_kmajor) return launch_pers<true, false>(*d, st);
    if (!d->a_kmajor && !d->b_kmajor) return launch_pers<false, false>(*d, st);
    return launch_pers<false, true>(*d, st);
  }
  if (p3) {
    if (d->a_kmajor && d->b_kmajor) return launch_p3<true, true>(*d, st);
    if (d->a_kmajor && !d->b_kmajor) return launch_p3<true, false>(*d, st);
    if (!d->a_kmajor && !d->b_kmajor) return launch_p3<false, false>(*d, st);
    return launch_p3<false, true>(*d, st);
  }
  if (big) {
    if (d->a_kmajor && d->b_kmajor) return launch_big<true, true>(*d, st);
    if (d->a_kmajor && !d->b_kmajor) return launch_big<true, false>(*d, st);
    if (!d->a_kmajor && !d->b_kmajor) return launch_big<false, false>(*d, st);
    return launch_big<false, true>(*d, st);
  }
  if (d->a_kmajor && d->b_kmajor) return direct ? launch<true, true, true, true>(*d, st) : launch<true, true, true, false>(*d, st);
  if (d->a_kmajor && !d->b_kmajor)
    return direct ? launch<true, false, true, true>(*d, st) : (tr ? launch<true, false, true, false>(*d, st) : launch<true, false, false, false>(*d, st));
  if (!d->a_kmajor && !d->b_kmajor)
    return direct ? launch<false, false, true, true>(*d, st) : (tr ? launch<false, false, true, false>(*d, st) : launch<false, false, false, false>(*d, st));
  return direct ? launch<false, true, true, true>(*d, st) : (tr ? launch<false, true, true, false>(*d, st) : launch<false, true, false, false>(*d, st));
}

// Scratch the split-K slabs of this problem take when `workspace` is supplied (a workspace-size query per op that
// takes one).  Mirrors the plan of vmvm_gemm_bf16 for variant 0 / splitk 0|>1; 0 = this problem never splits.
extern "C" int64_t vmvm_gemm_workspace_size(const vmvm_gemm_desc* d) {
  if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0) return VMVM_EINVAL;
  if (!gemm_tiles_ok(d->M, d->N)) return VMVM_ENOSUPPORT;
  const bool plain_acc = d->out_fp32 && d->accumulate && !d->bias && !d->row_scale && !d->act && !d->resid && !d->row_map &&
                         d->dropout_p <= 0.f && d->col_scale_n == 0;
  if (!plain_acc || d->in_fp8 || d->in_fp16 || d->conv_taps || d->splitk == 1) return 0;
  const int nk_all = (d->K + BK - 1) / BK;
  int s = d->splitk;
  if (s == 0) {
    const int tiles = ((d->M + BM - 1) / BM) * ((d->N + BN - 1) / BN);
    s = ((d->K % BK == 0) ? 512 : 1024) / tiles;
    if (s > nk_all / 4) s = nk_all / 4;
  }
  int64_t need = 0;
  if (s >= 2) {
    const int per = (nk_all + s - 1) / s;
    s = (nk_all + per - 1) / per;
    if (s >= 2) need = (int64_t)s * d->M * d->N * (int64_t)sizeof(float) + (d->colsum ? (int64_t)s * d->M * (int64_t)sizeof(float) : 0);     // (+ the bias-gradient partials, colsum_parts)
  }
  if (d->splitk == 0 && !d->a_kmajor && !d->b_kmajor && d->K % BK == 0 && d->K >= 4096 && d->M >= 256 && d->N >= 256 &&
      (long)d->M * d->N >= (1 << 20)) {                                                                                       // 256x256 plan
    int s2 = 256 / (((d->M + 255) / 256) * ((d->N + 255) / 256));
    if (s2 > nk_all / 4) s2 = nk_all / 4;
    if (s2 >= 2) {
      const int per = (nk_all + s2 - 1) / s2;
      s2 = (nk_all + per - 1) / per;
      const int64_t n2 = s2 >= 2 ? (int64_t)s2 * d->M * d->N * (int64_t)sizeof(float) + (d->colsum ? (int64_t)s2 * d->M * (int64_t)sizeof(float) : 0) : 0;
      if (n2 > need) need = n2;
    }
  }
  return need;
}

