// blocks.hip -- block-level entry points of libvmvm (round 6): one foreign call = every launch of one fusion-encoder layer, forward or
// backward (include/vmvm.h, vmvm_bert_layer).  HOST code only: it fills the per-kernel descriptors exactly as the Python wrappers of
// pytorch_empirical_mvm_amd/kernels.py do (engine_fusion._bert_layer is the statement this file follows, line by line) and calls the
// per-kernel entry points of this library -- so the two forms launch the same kernels with the same arguments and are pinned against
// each other bit for bit (tests/test_round6_gpu.py).  Reference: transformers BertLayer via model.py:204-214.
#include "common.h"
#include <cmath>
#include <cstring>

namespace {

struct Ctx { const vmvm_bert_layer* l; hipStream_t st; int M, H, F; };

vmvm_gemm_desc gemm0(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int reserve) {
  vmvm_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.A = A; d.B = B; d.C = C;
  d.M = M; d.N = N; d.K = K; d.lda = lda; d.ldb = ldb; d.ldc = ldc;
  d.a_kmajor = 1; d.b_kmajor = 1;
  d.col_scale = 1.0f; d.alpha = 1.0f;
  d.reserve_cus = reserve;
  return d;
}

vmvm_attn_fwd_desc attn_desc(const vmvm_bert_layer* l) {
  vmvm_attn_fwd_desc a;
  memset(&a, 0, sizeof(a));
  const int H = l->hidden, hd = H / l->heads;
  a.qkv = l->qkv; a.ld_qkv = 3 * H; a.q_off = 0; a.k_off = H; a.v_off = 2 * H;
  a.out = l->ctx; a.ld_out = H; a.lse = l->lse;
  a.nseq = l->nseq; a.L = l->L; a.heads = l->heads; a.head_dim = hd; a.mode = 1;
  a.scale = 1.0f / sqrtf((float)hd);
  a.n_win = 1;
  a.keymask = l->keymask;
  a.dropout_p = l->p_attn; a.seed = l->seed; a.offset = l->off_attn;
  a.causal_from = l->causal_from;
  a.att_colsum = l->att_colsum; a.att_scale = 1.0f / (float)l->heads;
  a.drop_mask = reinterpret_cast<uint32_t*>(l->drop_mask);
  return a;
}

int check(const vmvm_bert_layer* l) {
  if (!l || l->nseq <= 0 || l->L <= 0 || l->hidden <= 0 || l->heads <= 0 || l->ffn <= 0 || (l->hidden % l->heads) || (l->hidden & 7) || (l->ffn & 7)) return VMVM_EINVAL;
  if ((int64_t)l->nseq * l->L > 0x7fffffff / 4) return VMVM_ENOSUPPORT;
  if (!l->Wqkv || !l->Wo || !l->W1 || !l->W2 || !l->bqkv || !l->bo || !l->b1 || !l->b2 || !l->ln1_g || !l->ln1_b || !l->ln2_g || !l->ln2_b) return VMVM_EINVAL;
  if (!l->x || !l->qkv || !l->ctx || !l->lse || !l->a || !l->x1 || !l->mean1 || !l->rstd1 || !l->u || !l->h || !l->f || !l->x2 || !l->mean2 || !l->rstd2) return VMVM_EINVAL;
  if (l->p_hidden < 0.f || l->p_hidden >= 1.f || l->p_attn < 0.f || l->p_attn >= 1.f) return VMVM_EINVAL;
  if (l->in_fp8 && (!l->Wqkv8 || !l->W18 || !l->x8 || !l->x18 || l->a8_scale <= 0.f || l->w8_scale <= 0.f)) return VMVM_EINVAL;
  return VMVM_OK;
}

int ln_fwd(const vmvm_bert_layer* l, const void* X, void* Y, const float* g, const float* b, float* mean, float* rstd, int M, int C, void* st) {
  vmvm_ln_fwd_desc d;
  memset(&d, 0, sizeof(d));
  d.X = X; d.ldx = C; d.Y = Y; d.ldy = C; d.gamma = g; d.beta = b; d.eps = l->ln_eps;
  d.M = M; d.C = C; d.nseg = 1;
  d.mean = mean; d.rstd = rstd;
  return vmvm_layernorm_fwd(&d, st);
}

// LayerNorm backward of the post-LN blocks: dX = LNbwd(dY) and, with hidden dropout on, dX2 = dropout_mask * dX / (1 - p)
int ln_bwd(const vmvm_bert_layer* l, const void* dY, const void* X, const float* g, const float* mean, const float* rstd, void* dX, void* dX2,
           float* dgamma, float* dbeta, uint64_t offset, int M, int C, void* st) {
  vmvm_ln_bwd_desc d;
  memset(&d, 0, sizeof(d));
  d.dY = dY; d.lddy = C; d.X = X; d.ldx = C; d.gamma = g; d.mean = mean; d.rstd = rstd;
  d.dX = dX; d.lddx = C; d.dgamma = dgamma; d.dbeta = dbeta;
  d.M = M; d.C = C; d.nseg = 1;
  d.dX2 = l->p_hidden > 0.f ? dX2 : nullptr; d.lddx2 = C;
  d.dropout_p = l->p_hidden; d.seed = l->seed; d.offset = offset;
  d.workspace = l->ws_main; d.workspace_bytes = (uint64_t)l->ws_main_bytes;
  d.reserve_cus = l->reserve_cus;
  return vmvm_layernorm_bwd(&d, st);
}

// dW += dy^T x ; db += colsum(dy)  on the side stream behind an event on the main stream (engine._linear_bwd / _wgrad_launch)
int wgrad(const vmvm_bert_layer* l, const void* dy, int ld_dy, const void* x, int ld_x, float* gW, float* gb, int N_out, int K_in, int M, hipStream_t st,
          hipStream_t side, hipEvent_t ev) {
  hipStream_t ws = st;
  if (side) {
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess) return VMVM_EHIP;
    ws = side;
  }
  vmvm_gemm_desc d = gemm0(dy, ld_dy, x, ld_x, gW, K_in, N_out, K_in, M, l->reserve_cus);
  d.a_kmajor = 0; d.b_kmajor = 0;
  d.out_fp32 = 1; d.accumulate = 1;
  d.colsum = gb;
  d.workspace = side ? l->ws_side : l->ws_main;
  d.workspace_bytes = side ? l->ws_side_bytes : l->ws_main_bytes;
  return vmvm_gemm_bf16(&d, ws);
}

// dx = dy W  : k-major x k-major on the transposed copy when there is one, else the m/n-major form on W
vmvm_gemm_desc dgrad(const vmvm_bert_layer* l, const void* dy, int ld_dy, const void* W, const void* WT, void* dx, int M, int N_out, int K_in) {
  // N_out = rows of W (the reduction of this GEMM), K_in = columns of W (its output width)
  if (WT) return gemm0(dy, ld_dy, WT, N_out, dx, K_in, M, K_in, N_out, l->reserve_cus);
  vmvm_gemm_desc d = gemm0(dy, ld_dy, W, K_in, dx, K_in, M, K_in, N_out, l->reserve_cus);
  d.b_kmajor = 0;
  return d;
}

}  // namespace

#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

extern "C" int vmvm_bert_layer_fwd(const vmvm_bert_layer* l, void* stream) {
  RC(check(l));
  const int M = l->nseq * l->L, H = l->hidden, F = l->ffn;
  // self-attention: fused QKV projection (query | key | value rows adjacent in the arena), flash kernel with key mask + dropout
  if (l->in_fp8) {
    RC(vmvm_cast_bf16_to_fp8(l->x, l->x8, (int64_t)M * H, l->a8_scale, stream));
    vmvm_gemm_desc d = gemm0(l->x8, H, l->Wqkv8, H, l->qkv, 3 * H, M, 3 * H, H, l->reserve_cus);
    d.bias = l->bqkv; d.in_fp8 = 1; d.alpha = 1.0f / (l->a8_scale * l->w8_scale);
    RC(vmvm_gemm_bf16(&d, stream));
  } else {
    vmvm_gemm_desc d = gemm0(l->x, H, l->Wqkv, H, l->qkv, 3 * H, M, 3 * H, H, l->reserve_cus);
    d.bias = l->bqkv;
    RC(vmvm_gemm_bf16(&d, stream));
  }
  {
    vmvm_attn_fwd_desc a = attn_desc(l);
    RC(vmvm_attention_fwd(&a, stream));
  }
  {                                                     // BertSelfOutput: dense + dropout + residual, LayerNorm
    vmvm_gemm_desc d = gemm0(l->ctx, H, l->Wo, H, l->a, H, M, H, H, l->reserve_cus);
    d.bias = l->bo; d.resid = l->x; d.ldr = H;
    d.dropout_p = l->p_hidden; d.seed = l->seed; d.offset = l->off_1;
    RC(vmvm_gemm_bf16(&d, stream));
    RC(ln_fwd(l, l->a, l->x1, l->ln1_g, l->ln1_b, l->mean1, l->rstd1, M, H, stream));
  }
  if (l->in_fp8) {                                      // BertIntermediate: dense + GELU (+ what the GELU backward needs)
    RC(vmvm_cast_bf16_to_fp8(l->x1, l->x18, (int64_t)M * H, l->a8_scale, stream));
    vmvm_gemm_desc d = gemm0(l->x18, H, l->W18, H, l->h, F, M, F, H, l->reserve_cus);
    d.bias = l->b1; d.act = 1; d.C2 = l->u; d.ldc2 = F; d.aux_code8 = l->code8; d.in_fp8 = 1; d.alpha = 1.0f / (l->a8_scale * l->w8_scale);
    RC(vmvm_gemm_bf16(&d, stream));
  } else {
    vmvm_gemm_desc d = gemm0(l->x1, H, l->W1, H, l->h, F, M, F, H, l->reserve_cus);
    d.bias = l->b1; d.act = 1; d.C2 = l->u; d.ldc2 = F; d.aux_code8 = l->code8;
    RC(vmvm_gemm_bf16(&d, stream));
  }
  {                                                     // BertOutput: dense + dropout + residual, LayerNorm
    vmvm_gemm_desc d = gemm0(l->h, F, l->W2, F, l->f, H, M, H, F, l->reserve_cus);
    d.bias = l->b2; d.resid = l->x1; d.ldr = H;
    d.dropout_p = l->p_hidden; d.seed = l->seed; d.offset = l->off_2;
    RC(vmvm_gemm_bf16(&d, stream));
    RC(ln_fwd(l, l->f, l->x2, l->ln2_g, l->ln2_b, l->mean2, l->rstd2, M, H, stream));
  }
  return VMVM_OK;
}

extern "C" int vmvm_bert_layer_bwd(const vmvm_bert_layer* l, void* stream, void* side_stream, void* fork_event) {
  RC(check(l));
  if (!l->d_out || !l->d_x || !l->df || !l->du || !l->dx1 || !l->da || !l->dctx || !l->dqkv || !l->delta) return VMVM_EINVAL;
  if (l->p_hidden > 0.f && (!l->dfm || !l->dam)) return VMVM_EINVAL;
  if (!l->gWqkv || !l->gWo || !l->gW1 || !l->gW2 || !l->gbqkv || !l->gbo || !l->gb1 || !l->gb2 || !l->gln1_g || !l->gln1_b || !l->gln2_g || !l->gln2_b) return VMVM_EINVAL;
  if (side_stream && !fork_event) return VMVM_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream), side = reinterpret_cast<hipStream_t>(side_stream);
  hipEvent_t ev = reinterpret_cast<hipEvent_t>(fork_event);
  const int M = l->nseq * l->L, H = l->hidden, F = l->ffn;
  const bool drop = l->p_hidden > 0.f;
  // output LayerNorm: df = d(f) (the residual gradient that travels on), dfm = its dropout-masked copy (what the dense layer saw)
  RC(ln_bwd(l, l->d_out, l->f, l->ln2_g, l->mean2, l->rstd2, l->df, l->dfm, l->gln2_g, l->gln2_b, l->off_2, M, H, stream));
  const void* dfm = drop ? l->dfm : l->df;
  RC(wgrad(l, dfm, H, l->h, F, l->gW2, l->gb2, H, F, M, st, side, ev));
  {                                                     // du = (dfm W2) * GELU'(.)
    vmvm_gemm_desc d = dgrad(l, dfm, H, l->W2, l->W2T, l->du, M, H, F);
    d.act = 3; d.aux = l->u; d.ldaux = F; d.aux_code8 = l->code8;
    RC(vmvm_gemm_bf16(&d, stream));
  }
  RC(wgrad(l, l->du, F, l->x1, H, l->gW1, l->gb1, F, H, M, st, side, ev));
  {                                                     // dx1 = du W1 + df
    vmvm_gemm_desc d = dgrad(l, l->du, F, l->W1, l->W1T, l->dx1, M, F, H);
    d.resid = l->df; d.ldr = H;
    RC(vmvm_gemm_bf16(&d, stream));
  }
  RC(ln_bwd(l, l->dx1, l->a, l->ln1_g, l->mean1, l->rstd1, l->da, l->dam, l->gln1_g, l->gln1_b, l->off_1, M, H, stream));
  const void* dam = drop ? l->dam : l->da;
  RC(wgrad(l, dam, H, l->ctx, H, l->gWo, l->gbo, H, H, M, st, side, ev));
  {
    vmvm_gemm_desc d = dgrad(l, dam, H, l->Wo, l->WoT, l->dctx, M, H, H);
    RC(vmvm_gemm_bf16(&d, stream));
  }
  {
    vmvm_attn_bwd_desc b;
    memset(&b, 0, sizeof(b));
    b.f = attn_desc(l);
    b.f.att_colsum = nullptr;
    b.dout = l->dctx; b.ld_dout = H; b.dqkv = l->dqkv; b.ld_dqkv = 3 * H;
    b.delta = l->delta;
    RC(vmvm_attention_bwd(&b, stream));
  }
  RC(wgrad(l, l->dqkv, 3 * H, l->x, H, l->gWqkv, l->gbqkv, 3 * H, H, M, st, side, ev));
  {                                                     // d(x) = dqkv Wqkv + da (the residual of the attention block)
    vmvm_gemm_desc d = dgrad(l, l->dqkv, 3 * H, l->Wqkv, l->WqkvT, l->d_x, M, 3 * H, H);
    d.resid = l->da; d.ldr = H;
    RC(vmvm_gemm_bf16(&d, stream));
  }
  return VMVM_OK;
}
