// gemm_pp.hip -- instantiations + dispatch of the 256x256 ping-pong GEMM (gemm_pp.h); its own translation unit so that it
// compiles beside gemm.hip.  Entered from vmvm_gemm_bf16 (gemm.hip) only: vmvm_gemm_pp() is an internal symbol.
#include "gemm_pp.h"

namespace {

// epilogue classes of the step's large NT GEMMs (same grouping as launch_pers in gemm.hip; row_map users stay on the 128x128
// kernel: their K is a Swin channel count, below the range where this kernel pays)
template <int F>
int launch_nt(const vmvm_gemm_desc& d, hipStream_t st) { return launch_pp_f<true, true, F, 2, 64, 2>(d, st); }

template <int F>
int launch_tn(const vmvm_gemm_desc& d, hipStream_t st) { return launch_pp_f<false, false, F, 2, 64, 2>(d, st); }

template <int F>
int launch_nt_fp8(const vmvm_gemm_desc& d, hipStream_t st) { return launch_pp_f<true, true, F, 2, 64, 2, 0, true>(d, st); }

}  // namespace

// need = epilogue feature mask of the descriptor (epi_need in gemm.hip).  Returns VMVM_ENOSUPPORT when no instantiation covers it.
int vmvm_gemm_pp(const vmvm_gemm_desc& d, int need, hipStream_t st) {
  if (!d.a_kmajor && !d.b_kmajor) {             // weight gradient dW (+)= dY^T X: f32 slabs / accumulate, optional fused bias gradient
    if ((need & ~(EF_SPLIT | EF_F32)) != 0) return VMVM_ENOSUPPORT;
    return d.colsum ? launch_tn<(EF_SPLIT | EF_F32 | EF_COLSUM)>(d, st) : launch_tn<(EF_SPLIT | EF_F32)>(d, st);
  }
  if (!(d.a_kmajor && d.b_kmajor)) return VMVM_ENOSUPPORT;
#define TRY_EPI(MASK) if ((need & ~(MASK)) == 0) return launch_nt<(MASK)>(d, st)
  if (d.aux_code8) {                                    // 8-bit GELU' codes: the forward-with-code and the decode-and-multiply classes
    if ((d.N & 15) || (d.act == 1 && (d.ldc2 & 15)) || (d.act == 3 && (d.ldaux & 15))) return VMVM_ENOSUPPORT;
    if (d.act == 1 && d.C2) TRY_EPI(EF_BIAS | EF_ACT1 | EF_RS | EF_CODE8);
    if (d.act == 3) TRY_EPI(EF_ACT3 | EF_RS | EF_CODE8);
    return VMVM_ENOSUPPORT;
  }
  TRY_EPI(0);
  TRY_EPI(EF_BIAS | EF_COLSCALE | EF_RS);
  TRY_EPI(EF_BIAS | EF_ACT1 | EF_RS);
  TRY_EPI(EF_ACT3 | EF_RS);
  TRY_EPI(EF_BIAS | EF_RESID | EF_RS);
  TRY_EPI(EF_BIAS | EF_RESID | EF_DROP);
#undef TRY_EPI
  return VMVM_ENOSUPPORT;
}

// fp8 (e4m3) operands on the same kernel (vmvm_gemm_desc.in_fp8; K / lda / ldb already in 2-byte units): the forward epilogue classes
int vmvm_gemm_pp_fp8(const vmvm_gemm_desc& d, int need, hipStream_t st) {
  if (!(d.a_kmajor && d.b_kmajor) || d.aux_code8) return VMVM_ENOSUPPORT;       // (8-bit GELU' codes: the 128x128 fp8 build writes them)
#define TRY_EPI(MASK) if ((need & ~(MASK)) == 0) return launch_nt_fp8<(MASK)>(d, st)
  TRY_EPI(0);
  TRY_EPI(EF_BIAS | EF_COLSCALE | EF_RS);
  TRY_EPI(EF_BIAS | EF_ACT1 | EF_RS);
  TRY_EPI(EF_BIAS | EF_RESID | EF_DROP);
#undef TRY_EPI
  return VMVM_ENOSUPPORT;
}
