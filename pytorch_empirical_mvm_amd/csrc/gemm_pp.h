// gemm_pp.h -- 256x256 "ping-pong" bf16 MFMA GEMM for gfx950: the main loop of the large NT / TN GEMMs of the step.
//
// C[M,N] = epilogue( sum_k A(m,k) B(n,k) ), same descriptor / epilogue as gemm.hip (gemm_epi.h).
//
// Why a second main loop.  The 128x128 persistent kernel stages 64 B/clk/CU from L2 at MFMA peak (more than the L2 delivers),
// issues ~100 non-MFMA instructions per 32 MFMAs and drains its DMA queue (vmcnt(0)) in front of every K step.  Here:
//  * tile 256x256, 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128x64 = 4x2 v_mfma_f32_32x32x16_bf16 tiles (128 accumulator
//    registers): half the L2->LDS bytes, half the LDS fragment bytes and half the MFMA issue slots per flop;
//  * a ring of NS LDS stages of BK=32 k-columns (A 16 KiB + B 16 KiB each), filled by direct-to-LDS DMA that runs NS-1 stages
//    ahead of the multiply and is retired with COUNTED s_waitcnt vmcnt(N) -- the queue is never drained inside the loop, the
//    ring runs across tile boundaries (the next tile's first stages are in flight during the epilogue);
//  * the two wave groups (M halves; waves w and w+4 share a SIMD) run half a step apart: each step is
//        [barrier] LOAD: 12 ds_read_b128 + DMA requests  [barrier] MULTIPLY: 16 MFMAs (s_setprio 1)
//    and group 1 starts one barrier late, so on every SIMD one wave multiplies while its partner loads.
// Synchronisation (global barrier count b0, b1, ...; group 0 loads stage i between b[2i] and b[2i+1], group 1 between b[2i+1]
// and b[2i+2]):  every wave retires ITS DMA pieces of stage i with a counted vmcnt right before it arrives at b[2i] (group 0:
// first thing of LOAD(i); group 1: last thing of LOAD(i-1)), so after b[2i] the stage is complete for every reader.  The DMA of
// stage i+NS-1 overwrites the slot of stage i-1; it is requested after b[2i] (group 0) / b[2i+1] (group 1), when both groups'
// reads of stage i-1 have been waited for (group 1 waits lgkmcnt(0) before it arrives at b[2i]).
// Operand swizzles: LDS images are lane-linear per DMA instruction, so the 16-byte-chunk XOR swizzle sits on the per-lane
// SOURCE offset and on the read address (same involution).  The B fragment rows are permuted inside each 32-column block so a
// lane's 16 accumulators of a tile are two runs of 8 consecutive output columns (16-byte stores, 32 contiguous bytes per row
// and instruction), which is what epi_store8 consumes.
#pragma once
#include <vmvm_probe_hooks.h>
#include "gemm_epi.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int PP_T = 256;                              // tile edge (M and N)

template <int BK> __device__ __forceinline__ int pp_fsw(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// transposing LDS reads through inline asm (see gemm.hip: the builtin makes the compiler drain the DMA queue in front of every read);
// the halves are valid only after pp_tr_wait (s_waitcnt lgkmcnt(0) that also ties the registers)
__device__ __forceinline__ uint32_t pp_lds_addr(const unsigned char* p) {
  typedef __attribute__((address_space(3))) const unsigned char lds_u8;
  return (uint32_t)(size_t)(lds_u8*)p;
}
template <int O1, int O2>
__device__ __forceinline__ void pp_tr_issue2(s16x4& lo, s16x4& hi, uint32_t a) {      // two reads off one address register (16-bit immediates)
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4" : "=&v"(lo), "=&v"(hi) : "v"(a), "n"(O1), "n"(O2) : "memory");
}
__device__ __forceinline__ bf16x8 pp_tr_cat(const s16x4& a, const s16x4& b) {
  typedef __attribute__((ext_vector_type(8))) short s16x8_;
  const s16x8_ v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int pp_smem_bytes(int WM, int BK, int NS) { return NS * (WM * 128 + PP_T) * BK * 2 + WM * 4 * 4096; }   // ring + 4 KiB per wave (epilogue re-tiling)

// WM = wave groups along M: 2 -> 512 threads, tile 256x256, one workgroup per CU, the two groups ping-pong (see above);
//                           1 -> 256 threads, tile 128x256, TWO independent workgroups per CU (each wave still owns 128x64):
//      one workgroup's epilogue (VALU + the HBM write burst) runs under the other one's main loop.  One barrier per step.
// DBG (probe builds only): 1 = no epilogue (accumulators kept live), 2 = s_memtime phase timers of block 0 -> p.workspace
// FP8: A [M][K] and B [N][K] are OCP e4m3 bytes (k-major); the descriptor reaches the kernel with K / lda / ldb counted in 2-byte
//      units (as the 128x128 fp8 build), so the DMA ring and the LDS images are byte-for-byte those of the bf16 kernel: a 64-"element"
//      stage row holds 128 fp8.  A 16-MFMA sub-step becomes 8 x v_mfma_scale_f32_32x32x64_f8f6f4 (same 512 matrix-pipe cycles, twice
//      the K): a lane's operand is the 32 consecutive K bytes = two adjacent 16-byte chunks of its row.  C = epilogue(alpha * acc).
template <bool AK, bool BKM, int F, int WM, int BK, int NS, bool F16 = false, int DBG = 0, bool FP8 = false>
__global__ __launch_bounds__(WM * 256, 2) void gemm_pp_kernel(const vmvm_gemm_desc p) {
  static_assert(AK == BKM, "both operands k-major (forward / dgrad) or both m/n-major (weight gradient dW = dY^T X)");
  static_assert(!FP8 || (AK && !F16), "fp8 operands: k-major x k-major");
  constexpr bool TR = !AK;
  static_assert(BK == 32 || BK == 64, "BK");
  static_assert(WM == 1 || WM == 2, "WM");
  static_assert(NS >= 2, "ring depth");
  static_assert((F & EF_EDGE4) == 0, "N % 8 == 0 (the bias is folded into the accumulators; the 4-wide edge path would add it again)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT = WM * 256;                  // threads
  constexpr int TM = WM * 128;                  // tile rows
  constexpr int ROWB = BK * 2;                  // bytes per LDS row (k-major image)
  constexpr int OPA = TM * ROWB;                // bytes of the A image of a stage
  constexpr int STB = (TM + PP_T) * ROWB;       // bytes per stage
  constexpr int CPR = BK / 8;                   // 16-byte chunks per row
  constexpr int PDA = BK / 16;                  // DMA instructions per thread and stage: A
  constexpr int PDB = BK / (8 * WM);            //                                         B
  constexpr int P = PDA + PDB;
  constexpr int SUB = BK / 32;                  // 16-MFMA sub-steps per stage
  typedef __attribute__((address_space(3))) void lds_void;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int M = p.M, N = p.N, K = p.K;
  const int nbn = (N + PP_T - 1) / PP_T, nbm = (M + TM - 1) / TM;
  const int nb = nbm * nbn;
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int nbt = nb * S;
  const int nk_all = K / BK;
  const int per = (nk_all + S - 1) / S;
  const int xcd = blockIdx.x & 7, li = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;      // gridDim.x is a multiple of 8
  const int q2 = nbt >> 3, rr2 = nbt & 7;
  const int x_start = (xcd < rr2) ? xcd * (q2 + 1) : rr2 * (q2 + 1) + (xcd - rr2) * q2;
  const int x_cnt = q2 + (xcd < rr2 ? 1 : 0);
  if (li >= x_cnt) return;                      // whole workgroup: nothing to do
  const unsigned bytesA = (unsigned)((size_t)(TR ? K : M) * p.lda * 2), bytesB = (unsigned)((size_t)(TR ? K : N) * p.ldb * 2);
  const __amdgpu_buffer_rsrc_t ra_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, (int)bytesB, 0x00020000);

  auto decode = [&](int w, int& slice, int& m0, int& n0, int& kt0, int& nk) {
    const int logical = x_start + w;
    slice = logical / nb;
    int tm, tn;
    raster(logical - slice * nb, nbm, nbn, vmvm_hook::GM_PP<WM>, tm, tn);
    m0 = tm * TM; n0 = tn * PP_T;
    kt0 = slice * per;
    nk = (kt0 + per < nk_all) ? kt0 + per : nk_all;
  };

  // ---- DMA side: a cursor that runs NS-1 stages ahead of the multiply, across tile boundaries -------------------------------
  const int wave_base = wave * 64;
  unsigned ivA[PDA], ivB[PDB];                  // per-lane source offsets of the cursor's tile (K position goes in the scalar offset)
  int iw = li, ikt = 0, ink = 0, islot = 0;
  auto cursor_tile = [&]() {
    int sl = 0, m0 = 0, n0 = 0;
    while (iw < x_cnt) {                        // skip empty K slices (the multiply side runs zero steps for them)
      decode(iw, sl, m0, n0, ikt, ink);
      if (ikt < ink) break;
      iw += per_xcd;
    }
    if (iw < x_cnt) {
      if (TR) {
        // m/n-major operand images: [BK k-rows][TM or 256 columns], 32-byte slots XOR-swizzled by ((krow & 3) << 1) so that the
        // 2 slots x 4 k-rows a half-wave's transposing read touches fall on 8 different bank octets
#pragma unroll
        for (int i = 0; i < PDA; ++i) {
          const int u = tid + i * NT, krow = u / (TM / 8), x = u % (TM / 8);
          const int src = (((x >> 1) ^ ((krow & 3) << 1)) * 2 + (x & 1)) * 8;
          ivA[i] = (unsigned)(((size_t)krow * p.lda + m0 + src) * 2);
        }
#pragma unroll
        for (int i = 0; i < PDB; ++i) {
          const int u = tid + i * NT, krow = u / (PP_T / 8), x = u % (PP_T / 8);
          const int src = (((x >> 1) ^ ((krow & 3) << 1)) * 2 + (x & 1)) * 8;
          ivB[i] = (unsigned)(((size_t)krow * p.ldb + n0 + src) * 2);
        }
      } else {
#pragma unroll
        for (int i = 0; i < PDA; ++i) {
          const int u = tid + i * NT, row = u / CPR, cs = u % CPR;
          ivA[i] = (unsigned)(((size_t)(m0 + row) * p.lda + ((cs ^ pp_fsw<BK>(row)) * 8)) * 2);
        }
#pragma unroll
        for (int i = 0; i < PDB; ++i) {
          const int u = tid + i * NT, row = u / CPR, cs = u % CPR;
          ivB[i] = (unsigned)(((size_t)(n0 + row) * p.ldb + ((cs ^ pp_fsw<BK>(row)) * 8)) * 2);
        }
      }
    } else {                                    // past the end: phantom stages (out-of-range requests) keep the vmcnt arithmetic uniform
#pragma unroll
      for (int i = 0; i < PDA; ++i) ivA[i] = bytesA;
#pragma unroll
      for (int i = 0; i < PDB; ++i) ivB[i] = bytesB;
      ikt = 0; ink = 0x7fffffff;
    }
  };
  // (the per-lane offsets reach the DMA builtin through pointer parameters: with a local array element as its argument
  //  hipcc 7.2 silently drops the HOST stub of the kernel)
  auto issue_piece_ = [&](const unsigned* vA, const unsigned* vB, int j) {      // j in [0, P): A pieces first
    unsigned char* st = smem + islot * STB + wave_base * 16;
    const unsigned soA = (unsigned)ikt * (unsigned)(TR ? BK * p.lda * 2 : BK * 2), soB = (unsigned)ikt * (unsigned)(TR ? BK * p.ldb * 2 : BK * 2);
    if (j < PDA) { const unsigned vo = vA[j]; __builtin_amdgcn_raw_ptr_buffer_load_lds(ra_, (lds_void*)(st + j * (NT * 16)), 16, vo, soA, 0, 0); }
    else { const unsigned vo = vB[j - PDA]; __builtin_amdgcn_raw_ptr_buffer_load_lds(rb_, (lds_void*)(st + OPA + (j - PDA) * (NT * 16)), 16, vo, soB, 0, 0); }
  };
  auto issue_stage = [&]() {
#pragma unroll
    for (int j = 0; j < P; ++j) issue_piece_(ivA, ivB, j);
    islot = (islot + 1 == NS) ? 0 : islot + 1;
    if (++ikt >= ink) { iw += per_xcd; cursor_tile(); }
  };

  // ---- fragment read addresses (byte offsets inside a stage; the k-chunk index is XORed with the row swizzle) --------------------
  const int l31 = lane & 31, hh = lane >> 5;
  const int rowA = wr * 128 + l31;                                                     // + 32 * mb
  const int rowB = wc * 64 + 16 * (l31 >> 4) + 8 * ((l31 >> 2) & 1) + 4 * ((l31 >> 3) & 1) + (l31 & 3);   // + 32 * nb (column permutation)
  int aoff[SUB][2], boff[SUB][2];
#pragma unroll
  for (int kk = 0; kk < SUB; ++kk)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = FP8 ? kk * 4 + hh * 2 + ks : kk * 4 + ks * 2 + hh;      // fp8: (ks = 0, 1) = low / high 16 bytes of the lane's 32 K bytes
      aoff[kk][ks] = rowA * ROWB + ((c ^ pp_fsw<BK>(rowA)) << 4);
      boff[kk][ks] = OPA + rowB * ROWB + ((c ^ pp_fsw<BK>(rowB)) << 4);
    }

  // transposing reads (m/n-major images): 16-lane group g = (column block g & 1, k block g >> 1) of a 32x16 fragment; lane r of the
  // group addresses piece r & 3 (4 columns, 8 bytes) of k-row r >> 2 and receives column r of the 4 x 16 block.  The B pieces sit at
  // columns 8 * (p & 1) + 4 * (p >> 1): the same column permutation as rowB above.
  const int tg = lane >> 4, tr_ = lane & 15;
  const int tkrow = (tg >> 1) * 8 + (tr_ >> 2), tf = ((tr_ >> 2) & 3) << 1;
  int toffA[4], toffB[2];
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) toffA[mb] = tkrow * (TM * 2) + (((wr * 8 + mb * 2 + (tg & 1)) ^ tf) << 5) + (tr_ & 3) * 8;
#pragma unroll
  for (int nb_ = 0; nb_ < 2; ++nb_)
    toffB[nb_] = OPA + tkrow * (PP_T * 2) + (((wc * 4 + nb_ * 2 + (tg & 1)) ^ tf) << 5) + (((tr_ & 1) * 8 + ((tr_ >> 1) & 1) * 4) * 2);

  EpiCtx ec;
  ec.has_drop = p.dropout_p > 0.f; ec.thr = dropout_threshold(p.dropout_p);
  ec.keep_scale = ec.has_drop ? 1.0f / (1.0f - p.dropout_p) : 1.0f;
  ec.S = S; ec.M = M; ec.N = N;

  if (DBG & 4) {                                // start stagger: the write bursts of a round's epilogues are spread over a tile time
    int sl, m0_, n0_, k0_, k1_;
    decode(li, sl, m0_, n0_, k0_, k1_);
    const int wait64 = ((li & 3) * (k1_ - k0_) * SUB * 1365) >> 8;        // (li & 3) / 4 of an estimated tile time, in 64-cycle units
    for (int t = 0; t < wait64; t += 100) __builtin_amdgcn_s_sleep(100);
  }
  // ---- prologue: NS-1 stages in flight; group 1 starts one barrier late ------------------------------------------------------------
  cursor_tile();
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) issue_stage();
  if (WM == 2 && wr == 1) {
    pp_wait_vm<(NS - 2) * P>();                 // stage 0 (my pieces)
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_sched_barrier(0);

  int w = li, cslot = 0;
  unsigned tacc[5] = {0, 0, 0, 0, 0};
  unsigned long long tprev = 0;
  auto tick = [&](int slot) {
    if (DBG & 2) {
      unsigned long long t;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      if (slot >= 0) tacc[slot] += (unsigned)(t - tprev);
      tprev = t;
    }
  };
  while (true) {
    int slice, m0, n0, kt0, nk;
    decode(w, slice, m0, n0, kt0, nk);
    // accumulators start at the bias (times the DropPath row scale in the 'producer' form): the epilogue then carries neither the
    // 32 bias registers nor the adds
    f32x16 acc[4][2];
    const float inv_alpha = FP8 ? 1.0f / p.alpha : 1.0f;          // fp8: the epilogue multiplies by alpha, the folded bias is pre-divided
    {
      float bz[4][8];
#pragma clang loop unroll(full)
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wc * 64 + q * 16 + hh * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) bz[q][e] = 0.f;
        if ((F & EF_BIAS) && p.bias && n + 8 <= N) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
          bz[q][0] = b0.x; bz[q][1] = b0.y; bz[q][2] = b0.z; bz[q][3] = b0.w; bz[q][4] = b1.x; bz[q][5] = b1.y; bz[q][6] = b1.z; bz[q][7] = b1.w;
        }
      }
#pragma clang loop unroll(full)
      for (int mb = 0; mb < 4; ++mb) {
        float bs = 1.0f;
        if ((F & EF_BIAS) && (F & EF_RS) && p.row_scale && p.scale_bias_only) {
          const int m = m0 + wr * 128 + mb * 32 + l31;
          bs = m < M ? p.row_scale[(m + p.scale_row0) / p.rows_per_scale] : 0.f;
        }
#pragma clang loop unroll(full)
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[mb][q >> 1][(q & 1) * 8 + e] = FP8 ? bz[q][e] * bs * inv_alpha : bz[q][e] * bs;
      }
    }

    float csum[4] = {0.f, 0.f, 0.f, 0.f};
    const bool do_cs = (F & EF_COLSUM) && p.colsum && n0 == 0 && wc == 0;
    tick(-1);
    for (int kt = kt0; kt < nk; ++kt) {
      const unsigned char* sb = smem + cslot * STB;
#pragma unroll
      for (int kk = 0; kk < SUB; ++kk) {
        // ------------------------------- LOAD -------------------------------
        tick(4);                                                       // (multiply + loop overhead of the previous step)
        if (kk == 0 && wr == 0) pp_wait_vm<(NS - 2) * P>();           // stage `it` landed (my pieces)
        tick(0);
        if (WM == 2 || kk == 0) {
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        tick(1);
        typedef __attribute__((ext_vector_type(8))) int i32x8_;
        typedef __attribute__((ext_vector_type(4))) int i32x4_;
        bf16x8 fa[2][4], fb[2][2];
        i32x8_ fa8[4], fb8[2];
        s16x4 alo[2][4], ahi[2][4], blo[2][2], bhi[2][2];
        if (TR) {
          const uint32_t sbase = pp_lds_addr(sb);
          uint32_t aA[4], aB[2];                 // one address register per fragment column block; k position in the immediates
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) aA[mb] = sbase + toffA[mb];
#pragma unroll
          for (int nb_ = 0; nb_ < 2; ++nb_) aB[nb_] = sbase + toffB[nb_];
#define PP_TR_SET(KK, KS)                                                                                                          \
          do {                                                                                                                     \
            constexpr int koB = ((KK) * 32 + (KS) * 16) * (PP_T * 2), koA = ((KK) * 32 + (KS) * 16) * (TM * 2);                    \
            pp_tr_issue2<koB, koB + 4 * (PP_T * 2)>(blo[KS][0], bhi[KS][0], aB[0]);                                               \
            pp_tr_issue2<koB, koB + 4 * (PP_T * 2)>(blo[KS][1], bhi[KS][1], aB[1]);                                               \
            pp_tr_issue2<koA, koA + 4 * (TM * 2)>(alo[KS][0], ahi[KS][0], aA[0]);                                                 \
            pp_tr_issue2<koA, koA + 4 * (TM * 2)>(alo[KS][1], ahi[KS][1], aA[1]);                                                 \
            pp_tr_issue2<koA, koA + 4 * (TM * 2)>(alo[KS][2], ahi[KS][2], aA[2]);                                                 \
            pp_tr_issue2<koA, koA + 4 * (TM * 2)>(alo[KS][3], ahi[KS][3], aA[3]);                                                 \
          } while (0)
          if (kk == 0) { PP_TR_SET(0, 0); PP_TR_SET(0, 1); }
          else { PP_TR_SET(1, 0); PP_TR_SET(1, 1); }
#undef PP_TR_SET
        } else if constexpr (FP8) {
          // a lane's 32 K bytes = two 16-byte chunks, loaded straight into the halves of the 8-register MFMA operand
#pragma unroll
          for (int nb_ = 0; nb_ < 2; ++nb_) {
            const i32x4_ lo = *reinterpret_cast<const i32x4_*>(sb + boff[kk][0] + nb_ * 32 * ROWB), hi = *reinterpret_cast<const i32x4_*>(sb + boff[kk][1] + nb_ * 32 * ROWB);
            fb8[nb_] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int mb = 0; mb < 4; ++mb) {
            const i32x4_ lo = *reinterpret_cast<const i32x4_*>(sb + aoff[kk][0] + mb * 32 * ROWB), hi = *reinterpret_cast<const i32x4_*>(sb + aoff[kk][1] + mb * 32 * ROWB);
            fa8[mb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int nb_ = 0; nb_ < 2; ++nb_) fb[ks][nb_] = *reinterpret_cast<const bf16x8*>(sb + boff[kk][ks] + nb_ * 32 * ROWB);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) fa[ks][mb] = *reinterpret_cast<const bf16x8*>(sb + aoff[kk][ks] + mb * 32 * ROWB);
          }
        }
        if (kk == 0) issue_stage();                                    // stage it+NS-1 -> the slot of stage it-1
        if (WM == 2) {
          if (wr == 1) {
            if (kk == SUB - 1) pp_wait_vm<(NS - 2) * P>();             // stage it+1 (my pieces), needed by group 0 after the next barrier
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // my reads of this stage are done before anyone may overwrite it
          }
          __builtin_amdgcn_sched_barrier(0);
          tick(2);
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          tick(3);
        }
        // ----------------------------- MULTIPLY -----------------------------
        if (TR) {
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(alo[0][0]), "+v"(alo[0][1]), "+v"(alo[0][2]), "+v"(alo[0][3]), "+v"(ahi[0][0]), "+v"(ahi[0][1]), "+v"(ahi[0][2]), "+v"(ahi[0][3]),
                         "+v"(alo[1][0]), "+v"(alo[1][1]), "+v"(alo[1][2]), "+v"(alo[1][3]), "+v"(ahi[1][0]), "+v"(ahi[1][1]), "+v"(ahi[1][2]), "+v"(ahi[1][3]),
                         "+v"(blo[0][0]), "+v"(blo[0][1]), "+v"(bhi[0][0]), "+v"(bhi[0][1]), "+v"(blo[1][0]), "+v"(blo[1][1]), "+v"(bhi[1][0]), "+v"(bhi[1][1]));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) fa[ks][mb] = pp_tr_cat(alo[ks][mb], ahi[ks][mb]);
#pragma unroll
            for (int nb_ = 0; nb_ < 2; ++nb_) fb[ks][nb_] = pp_tr_cat(blo[ks][nb_], bhi[ks][nb_]);
          }
          if ((F & EF_COLSUM) && do_cs) {        // bias gradient: column sums of the m-major A operand (dY), two k per v_dot2c
            const bf16x2 one2 = __builtin_bit_cast(bf16x2, 0x3f803f80u);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
              for (int mb = 0; mb < 4; ++mb) {
                // (pairs through shufflevector: with bit_cast(u32x4)[e] -> bit_cast(bf16x2) hipcc 7.2 feeds the FIRST dword to all four)
                const bf16x8 f8 = fa[ks][mb];
                csum[mb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f8, f8, 0, 1), one2, csum[mb], false);
                csum[mb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f8, f8, 2, 3), one2, csum[mb], false);
                csum[mb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f8, f8, 4, 5), one2, csum[mb], false);
                csum[mb] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f8, f8, 6, 7), one2, csum[mb], false);
              }
          }
        }
        __builtin_amdgcn_s_setprio(1);
        if constexpr (FP8) {
#pragma unroll
          for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int nb_ = 0; nb_ < 2; ++nb_)
              acc[mb][nb_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb8[nb_], fa8[mb], acc[mb][nb_], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
              for (int nb_ = 0; nb_ < 2; ++nb_)
                acc[mb][nb_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][nb_], fa[ks][mb], acc[mb][nb_], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (WM == 2) __builtin_amdgcn_sched_barrier(0);
      }
      cslot = (cslot + 1 == NS) ? 0 : cslot + 1;
    }

    // ------------------------------- epilogue (the DMA of the next tile's first stages is in flight) -------------------------------
    if (DBG & 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
      w += per_xcd;
      if (w >= x_cnt) break;
      continue;
    }
    ec.slice = slice;
    if ((F & EF_COLSUM) && do_cs) {              // lanes l and l + 32 hold the two k-halves of column m = ... + (l & 31)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) {
        float t = csum[mb] + __shfl_xor(csum[mb], 32, 64);
        if (p.colsum_scale != 0.f) t *= p.colsum_scale;
        const int m = m0 + wr * 128 + mb * 32 + l31;
        if (hh == 0 && m < M) {
          float* parts = colsum_parts(p, S);                 // (gemm.hip: [S][M] partials behind the slabs, summed in slice order by the reduce pass)
          if (parts) parts[(size_t)slice * M + m] = t; else atomicAdd(p.colsum + m, t);
        }
      }
    }
    bool rvalid[4]; long rdst[4]; float rrs[4];
#pragma clang loop unroll(full)
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + wr * 128 + mb * 32 + l31;
      bool valid = m < M;
      long dst = m;
      if ((F & EF_MAP) && valid && p.row_map) {
        const int mapped = p.row_map[m % p.map_len];
        valid = mapped >= 0;
        dst = (long)mapped + (long)(m / p.map_len) * p.map_stride;
      }
      rvalid[mb] = valid; rdst[mb] = dst;
      rrs[mb] = ((F & EF_RS) && valid && p.row_scale) ? p.row_scale[(m + p.scale_row0) / p.rows_per_scale] : 1.0f;
    }
    const float bz0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};       // (the bias is already in the accumulators)
    // 16-bit outputs leave through a per-wave 4 KiB LDS tile: the MFMA layout gives a lane ONE row (a store instruction would
    // touch 64 different cache lines, ~75 cycles each on the address path: the store tail of a tile cost more than a third of its
    // main loop); re-tiled, a store instruction writes 8 rows x 128 contiguous bytes.
    constexpr bool TS = (F & (EF_F32 | EF_SPLIT | EF_EDGE4)) == 0;
    unsigned char* stg = smem + NS * STB + wave * 4096;
    const int sr = lane >> 3, sg = lane & 7;             // store layout: row inside an 8-row group, 16-byte column group
    const bool has_pre = (F & EF_ACT1) && p.act == 1 && p.C2;
    // The saved activation (act 3 / 4) or the residual comes in the same way, backwards: requested up front in the store layout
    // (whole lines per instruction, all 16 requests of the tile in flight together), re-tiled to the MFMA layout row block by
    // row block.  (A mask that carries BOTH operands keeps the direct one-row-per-lane loads.)
    constexpr bool HAS_AUX = (F & (EF_ACT3 | EF_ACT24)) != 0, HAS_RES = (F & EF_RESID) != 0;
    constexpr bool TL = TS && (HAS_AUX != HAS_RES) && (F & (EF_MAP | EF_DROP)) == 0;      // (dropout: no registers left for it)
    const bool use_aux = HAS_AUX && p.act >= 3, use_res = HAS_RES && p.resid != nullptr;
    constexpr int XB = HAS_AUX ? 2 : 1;          // row blocks of requests in flight (register budget: the residual classes also hold the bias)
    // 8-bit GELU' codes (EF_CODE8): a row block of the wave's tile is 32 rows x 64 BYTES; global side = lane (row c8r + 16 * s2, 16-byte
    // group c8g), MFMA side = 8 bytes per (row l31, chunk 2q + hh); the 8-byte chunk index is XORed with an even row code so a 16-byte
    // group stays whole and the 8-byte reads of rows l31 and l31 + 16 are the only ones that share banks (512 bytes = two passes anyway)
    constexpr bool C8 = (F & EF_CODE8) != 0;
    const int c8r = lane >> 2, c8g = lane & 3;
    auto c8swz = [](int row) { return ((row >> 2) & 3) << 1; };
    uint4 xin[XB][4];
    auto request_block = [&](int mb, uint4 (&x)[4]) {
      const int n_s = n0 + wc * 64 + sg * 8;
      if constexpr (C8) {
#pragma clang loop unroll(full)
        for (int s2 = 0; s2 < 2; ++s2) {
          const int ms = m0 + wr * 128 + mb * 32 + s2 * 16 + c8r, n_c = n0 + wc * 64 + c8g * 16;
          x[s2] = make_uint4(0, 0, 0, 0);
          if (ms < M && n_c + 16 <= N && use_aux)
            x[s2] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(p.aux) + (size_t)ms * p.ldaux + n_c);
        }
        return;
      }
#pragma clang loop unroll(full)
      for (int s4 = 0; s4 < 4; ++s4) {
        const int ms = m0 + wr * 128 + mb * 32 + s4 * 8 + sr;
        x[s4] = make_uint4(0, 0, 0, 0);
        bool valid = ms < M && n_s + 8 <= N;
        if (HAS_AUX) {
          if (valid && use_aux) x[s4] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p.aux) + (size_t)ms * p.ldaux + n_s);
        } else {
          long dst = ms;
          if ((F & EF_MAP) && valid && p.row_map) {
            const int mapped = p.row_map[ms % p.map_len];
            valid = mapped >= 0;
            dst = (long)mapped + (long)(ms / p.map_len) * p.map_stride;
          }
          if (valid && use_res) x[s4] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p.resid) + (size_t)dst * p.ldr + n_s);
        }
      }
    };
    if constexpr (TL) {
      request_block(0, xin[0]);
      if (XB == 2) request_block(1, xin[XB - 1]);
    }
#pragma clang loop unroll(full)
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + wr * 128 + mb * 32 + l31;
      uint4 auxv[4], resv[4];
#pragma clang loop unroll(full)
      for (int q = 0; q < 4; ++q) { auxv[q] = make_uint4(0, 0, 0, 0); resv[q] = make_uint4(0, 0, 0, 0); }
      if constexpr (TL) {
        if constexpr (C8) {
          if (use_aux) {
#pragma clang loop unroll(full)
            for (int s2 = 0; s2 < 2; ++s2) {
              const int row = s2 * 16 + c8r;
              *reinterpret_cast<uint4*>(stg + row * 64 + (((c8g * 2) ^ c8swz(row)) << 3)) = xin[mb % XB][s2];
            }
            if (mb + XB < 4) request_block(mb + XB, xin[mb % XB]);
#pragma clang loop unroll(full)
            for (int q = 0; q < 4; ++q) {
              const uint2 t = *reinterpret_cast<const uint2*>(stg + l31 * 64 + (((2 * q + hh) ^ c8swz(l31)) << 3));
              auxv[q] = make_uint4(t.x, t.y, 0, 0);
            }
          }
        } else if (use_aux || use_res) {
#pragma clang loop unroll(full)
          for (int s4 = 0; s4 < 4; ++s4) *reinterpret_cast<uint4*>(stg + (s4 * 8 + sr) * 128 + ((sg ^ sr) << 4)) = xin[mb % XB][s4];
          if (mb + XB < 4) request_block(mb + XB, xin[mb % XB]);
#pragma clang loop unroll(full)
          for (int q = 0; q < 4; ++q) {
            const uint4 t = *reinterpret_cast<const uint4*>(stg + l31 * 128 + (((2 * q + hh) ^ (l31 & 7)) << 4));
            if (HAS_AUX) auxv[q] = t; else resv[q] = t;
          }
        }
      } else {
#pragma clang loop unroll(full)
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wc * 64 + q * 16 + hh * 8;
          const bool full = n + 8 <= N;
          if (HAS_AUX && p.act >= 3 && rvalid[mb] && full)
            auxv[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p.aux) + (size_t)m * p.ldaux + n);
          if (HAS_RES && p.resid && rvalid[mb] && full)
            resv[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(p.resid) + (size_t)rdst[mb] * p.ldr + n);
        }
      }
      if constexpr (TS) {
        uint4 o[4], pr[4];
#pragma clang loop unroll(full)
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wc * 64 + q * 16 + hh * 8;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = FP8 ? acc[mb][q >> 1][(q & 1) * 8 + e] * p.alpha : acc[mb][q >> 1][(q & 1) * 8 + e];
          o[q] = make_uint4(0, 0, 0, 0); pr[q] = make_uint4(0, 0, 0, 0);
          if (rvalid[mb] && n < N) epi_math8<(F & ~EF_BIAS), F16>(p, ec, v, m, n, rrs[mb], bz0, auxv[q], resv[q], o[q], pr[q]);
        }
        // store-side rows of this row block: 8 * s4 + sr
        bool svalid[4]; long sdst[4]; int srow[4];
        const int n_s = n0 + wc * 64 + sg * 8;
#pragma clang loop unroll(full)
        for (int s4 = 0; s4 < 4; ++s4) {
          const int ms = m0 + wr * 128 + mb * 32 + s4 * 8 + sr;
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
          if (C8 && pass == 1) {                        // the codes: 64-byte rows, 16 rows per store instruction
#pragma clang loop unroll(full)
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<uint2*>(stg + l31 * 64 + (((2 * q + hh) ^ c8swz(l31)) << 3)) = make_uint2(pr[q].x, pr[q].y);
            uint4 t2[2];
#pragma clang loop unroll(full)
            for (int s2 = 0; s2 < 2; ++s2) {
              const int row = s2 * 16 + c8r;
              t2[s2] = *reinterpret_cast<const uint4*>(stg + row * 64 + (((c8g * 2) ^ c8swz(row)) << 3));
            }
#pragma clang loop unroll(full)
            for (int s2 = 0; s2 < 2; ++s2) {
              const int ms = m0 + wr * 128 + mb * 32 + s2 * 16 + c8r, n_c = n0 + wc * 64 + c8g * 16;
              if (ms < M && n_c + 16 <= N)
                *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.C2) + (size_t)ms * p.ldc2 + n_c) = t2[s2];
            }
            break;
          }
#pragma clang loop unroll(full)
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint4*>(stg + l31 * 128 + (((2 * q + hh) ^ (l31 & 7)) << 4)) = pass ? pr[q] : o[q];
          uint4 t[4];
#pragma clang loop unroll(full)
          for (int s4 = 0; s4 < 4; ++s4) t[s4] = *reinterpret_cast<const uint4*>(stg + (s4 * 8 + sr) * 128 + ((sg ^ sr) << 4));
#pragma clang loop unroll(full)
          for (int s4 = 0; s4 < 4; ++s4) {
            if (!svalid[s4]) continue;
            if (pass == 0) *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.C) + (size_t)sdst[s4] * p.ldc + n_s) = t[s4];
            else *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(p.C2) + (size_t)srow[s4] * p.ldc2 + n_s) = t[s4];
          }
        }
      } else {
#pragma clang loop unroll(full)
        for (int q = 0; q < 4; ++q) {
          const int n = n0 + wc * 64 + q * 16 + hh * 8;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = FP8 ? acc[mb][q >> 1][(q & 1) * 8 + e] * p.alpha : acc[mb][q >> 1][(q & 1) * 8 + e];
          if (rvalid[mb] && n < N) epi_store8<(F & ~EF_BIAS), F16>(p, ec, v, m, rdst[mb], n, rrs[mb], N - n, bz0, auxv[q], resv[q]);
        }
      }
    }
    w += per_xcd;
    if (w >= x_cnt) break;
  }
  if (WM == 2 && wr == 0) __builtin_amdgcn_s_barrier();      // matches group 1's extra first barrier
  if ((DBG & 2) && blockIdx.x == 0 && lane == 0 && p.workspace) {
    unsigned* o = reinterpret_cast<unsigned*>(p.workspace) + wave * 8;
#pragma unroll
    for (int i = 0; i < 5; ++i) o[i] = tacc[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // phantom requests of the ring's tail
}

template <bool AK, bool BKM, int F, int WM, int BK, int NS, int DBG = 0, bool FP8 = false>
int launch_pp_f(const vmvm_gemm_desc& d, hipStream_t st) {
  constexpr int smem = pp_smem_bytes(WM, BK, NS);
  static_assert(smem <= (WM == 2 ? 160 : 80) * 1024, "LDS budget");
  const int items = ((d.M + WM * 128 - 1) / (WM * 128)) * ((d.N + PP_T - 1) / PP_T) * (d.splitk > 1 ? d.splitk : 1);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<AK, BKM, F, WM, BK, NS, false, DBG, FP8>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_done = true;
  }
  int grid = 2 * vmvm_usable_cus(d.reserve_cus) / WM;   // one (WM = 2) or two workgroups per CU, multiple of 8
  if (items < grid) grid = ((items + 7) / 8) * 8;
  if (DBG & 8) grid = 64;                               // probe: a quarter of the CUs (is the store tail a per-CU or a chip-wide limit?)
  hipLaunchKernelGGL((gemm_pp_kernel<AK, BKM, F, WM, BK, NS, false, DBG, FP8>), dim3(grid), dim3(WM * 256), smem, st, d);
  VMVM_CHECK_LAUNCH();
  return VMVM_OK;
}

}  // namespace
