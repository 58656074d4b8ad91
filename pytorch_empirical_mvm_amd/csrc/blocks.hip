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

// =====================================================================================================================================
// One Video-Swin block (include/vmvm.h vmvm_swin_block).  engine_swin._swin_block_calls is the statement this follows, line by line:
// SwinTransformerBlock3D.forward / forward_part1 / forward_part2 (video_swin.py:206-263) with the window shift / partition / reverse as
// gather maps, DropPath as per-clip scales or as a compaction onto the kept clips.
// =====================================================================================================================================
namespace {

int check_swin(const vmvm_swin_block* b) {
  if (!b || b->B <= 0 || b->L <= 0 || b->Lp <= 0 || b->N <= 0 || b->nW <= 0 || b->C <= 0 || b->heads <= 0 || (b->C % b->heads) || (b->C & 7)) return VMVM_EINVAL;
  if ((int64_t)b->B * b->Lp > 0x7fffffff / 8 || b->Lp != b->nW * b->N) return VMVM_EINVAL;
  if (!b->x || !b->x1 || !b->x2 || !b->src || !b->rc) return VMVM_EINVAL;
  if (b->has_attn && (!b->xw || !b->mean1 || !b->rstd1 || !b->qkv || !b->ao || !b->lse || !b->Wqkv || !b->Wproj || !b->bqkv || !b->bproj || !b->n1_g || !b->n1_b || !b->table)) return VMVM_EINVAL;
  if (b->has_attn && b->compact_a && (!b->kept_a || !b->drop_a || !b->src_k || !b->scale_a || b->Bk <= 0 || b->nd_a <= 0)) return VMVM_EINVAL;
  if (b->has_mlp && (!b->y2 || !b->mean2 || !b->rstd2 || !b->h || !b->W1 || !b->W2 || !b->b1 || !b->b2 || !b->n2_g || !b->n2_b)) return VMVM_EINVAL;
  if (b->has_mlp && b->compact_m && (!b->kept_m || !b->drop_m || !b->map_m || !b->idm || !b->scale_m || b->Bm <= 0 || b->nd_m <= 0)) return VMVM_EINVAL;
  return VMVM_OK;
}

struct SwinPlan {                 // rows and maps of the two branches
  int Ma, rob_a, rib_a, map_len_a, map_stride_a, nseq; const int32_t* map_a;
  int Mm; const int32_t* map_m;
};
SwinPlan swin_plan(const vmvm_swin_block* b) {
  SwinPlan p;
  if (b->compact_a) { p.Ma = b->Bk * b->Lp; p.map_a = b->src_k; p.rob_a = b->Bk * b->Lp; p.rib_a = b->B * b->L; p.map_len_a = b->Bk * b->Lp; p.map_stride_a = 0; p.nseq = b->Bk * b->nW; }
  else { p.Ma = b->B * b->Lp; p.map_a = b->src; p.rob_a = b->Lp; p.rib_a = b->L; p.map_len_a = b->Lp; p.map_stride_a = b->L; p.nseq = b->B * b->nW; }
  p.Mm = (b->compact_m ? b->Bm : b->B) * b->L;
  p.map_m = b->compact_m ? b->map_m : nullptr;
  return p;
}

vmvm_attn_fwd_desc swin_attn_desc(const vmvm_swin_block* b, const SwinPlan& p) {
  vmvm_attn_fwd_desc a;
  memset(&a, 0, sizeof(a));
  const int C = b->C;
  a.qkv = b->qkv; a.ld_qkv = 3 * C; a.q_off = 0; a.k_off = C; a.v_off = 2 * C;
  a.out = b->ao; a.ld_out = C; a.lse = b->lse;
  a.nseq = p.nseq; a.L = b->N; a.heads = b->heads; a.head_dim = C / b->heads; a.mode = 0;
  a.scale = b->qscale;
  a.bias_table = b->table; a.table_len = b->table_len;
  a.rc = b->rc; a.rc0 = b->rc0;
  a.region = b->region; a.n_win = b->nW;
  a.seq_scale = b->scale_a; a.seqs_per_scale = b->nW;
  a.att_scale = 1.0f / (float)b->heads;
  a.win_layout = b->win_layout;
  return a;
}

int swin_ln_fwd(const void* X, void* Y, const float* g, const float* bta, float* mean, float* rstd, int M, int C, const int32_t* src, int rob, int rib, void* st) {
  vmvm_ln_fwd_desc d;
  memset(&d, 0, sizeof(d));
  d.X = X; d.ldx = C; d.Y = Y; d.ldy = C; d.gamma = g; d.beta = bta; d.eps = 1e-5f;
  d.M = M; d.C = C; d.nseg = 1;
  d.src = src; d.rows_out_per_batch = rob; d.rows_in_per_batch = rib; d.pad_mode = 0;
  d.mean = mean; d.rstd = rstd;
  return vmvm_layernorm_fwd(&d, st);
}

vmvm_ln_bwd_desc swin_ln_bwd0(const vmvm_swin_block* b, const void* dY, const void* X, const float* g, const float* mean, const float* rstd, void* dX,
                              float* dgamma, float* dbeta, int M, const void* dX_add) {
  vmvm_ln_bwd_desc d;
  memset(&d, 0, sizeof(d));
  const int C = b->C;
  d.dY = dY; d.lddy = C; d.X = X; d.ldx = C; d.gamma = g; d.mean = mean; d.rstd = rstd;
  d.dX = dX; d.lddx = C; d.dgamma = dgamma; d.dbeta = dbeta;
  d.M = M; d.C = C; d.nseg = 1;
  d.dX_add = dX_add; d.ldadd = dX_add ? C : 0;
  d.lddx2 = C;
  d.workspace = b->ws_main; d.workspace_bytes = (uint64_t)b->ws_main_bytes;
  d.reserve_cus = b->reserve_cus;
  return d;
}

// engine._linear_bwd's weight-gradient half: [weighted column-sum pass +] dW += dy^T x [with the fused, possibly scaled, bias gradient]
int swin_wgrad(const vmvm_swin_block* b, const void* dy, int ld_dy, const void* x, int ld_x, float* gW, float* gb, int N_out, int K_in, int M, int cs_mode, float cs_scale,
               const float* row_scale, int rows_per_scale, hipStream_t st, hipStream_t side, hipEvent_t ev) {
  hipStream_t ws = st;
  if (side) {
    if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess) return VMVM_EHIP;
    ws = side;
  }
  void* wsp = side ? b->ws_side : b->ws_main;
  const int64_t wsb = side ? b->ws_side_bytes : b->ws_main_bytes;
  if (cs_mode == 2) {                                   // per-clip DropPath weights: separate pass (the fused form has ONE scale)
    int rc_ = vmvm_colsum_bf16_ws(dy, M, N_out, ld_dy, row_scale, rows_per_scale, gb, 1, wsp, wsb, ws);
    if (rc_) return rc_;
  }
  vmvm_gemm_desc d = gemm0(dy, ld_dy, x, ld_x, gW, K_in, N_out, K_in, M, b->reserve_cus);
  d.a_kmajor = 0; d.b_kmajor = 0;
  d.out_fp32 = 1; d.accumulate = 1;
  d.colsum = cs_mode == 2 ? nullptr : gb;
  d.colsum_scale = cs_mode == 1 ? cs_scale : 0.0f;
  d.workspace = wsp; d.workspace_bytes = wsb;
  return vmvm_gemm_bf16(&d, ws);
}

vmvm_gemm_desc swin_dgrad(const vmvm_swin_block* b, const void* dy, int ld_dy, const void* W, const void* WT, void* dx, int M, int N_out, int K_in) {
  if (WT) return gemm0(dy, ld_dy, WT, N_out, dx, K_in, M, K_in, N_out, b->reserve_cus);
  vmvm_gemm_desc d = gemm0(dy, ld_dy, W, K_in, dx, K_in, M, K_in, N_out, b->reserve_cus);
  d.b_kmajor = 0;
  return d;
}

}  // namespace

extern "C" int vmvm_swin_block_fwd(const vmvm_swin_block* b, void* stream) {
  RC(check_swin(b));
  const SwinPlan p = swin_plan(b);
  const int C = b->C, B = b->B, L = b->L, Lp = b->Lp;
  if (b->has_attn) {
    if (b->compact_a) RC(vmvm_expand_batch_map(b->src, Lp, b->kept_a, b->Bk, L, b->src_k, stream));          // absolute rows of the kept clips (pads stay -1)
    RC(swin_ln_fwd(b->x, b->xw, b->n1_g, b->n1_b, b->mean1, b->rstd1, p.Ma, C, p.map_a, p.rob_a, p.rib_a, stream));
    {
      vmvm_gemm_desc d = gemm0(b->xw, C, b->Wqkv, C, b->qkv, 3 * C, p.Ma, 3 * C, C, b->reserve_cus);
      d.bias = b->bqkv; d.col_scale = b->qscale; d.col_scale_n = C;
      RC(vmvm_gemm_bf16(&d, stream));
    }
    {
      vmvm_attn_fwd_desc a = swin_attn_desc(b, p);
      RC(vmvm_attention_fwd(&a, stream));
    }
    {                                                   // projection + bias * DropPath scale + residual, un-gathered through the window map
      vmvm_gemm_desc d = gemm0(b->ao, C, b->Wproj, C, b->x1, C, p.Ma, C, C, b->reserve_cus);
      d.bias = b->bproj; d.row_scale = b->scale_a; d.rows_per_scale = Lp; d.scale_bias_only = 1;
      d.resid = b->x; d.ldr = C;
      d.row_map = p.map_a; d.map_len = p.map_len_a; d.map_stride = p.map_stride_a;
      RC(vmvm_gemm_bf16(&d, stream));
    }
    if (b->compact_a) RC(vmvm_copy_batches_bf16(b->x, C, b->x1, C, b->drop_a, b->nd_a, L, C, stream));      // identity path of the dropped clips
  }
  if (b->has_mlp) {
    if (b->compact_m) {
      RC(vmvm_expand_batch_map(b->idm, L, b->kept_m, b->Bm, L, b->map_m, stream));
      RC(swin_ln_fwd(b->x1, b->y2, b->n2_g, b->n2_b, b->mean2, b->rstd2, p.Mm, C, b->map_m, b->Bm * L, B * L, stream));
    } else {
      RC(swin_ln_fwd(b->x1, b->y2, b->n2_g, b->n2_b, b->mean2, b->rstd2, p.Mm, C, nullptr, 0, 0, stream));
    }
    {
      vmvm_gemm_desc d = gemm0(b->y2, C, b->W1, C, b->h, 4 * C, p.Mm, 4 * C, C, b->reserve_cus);
      d.bias = b->b1; d.act = 1; d.C2 = b->u; d.ldc2 = b->u ? 4 * C : 0; d.aux_code8 = b->code8;
      d.row_scale = b->scale_m; d.rows_per_scale = L;
      RC(vmvm_gemm_bf16(&d, stream));
    }
    {
      vmvm_gemm_desc d = gemm0(b->h, 4 * C, b->W2, 4 * C, b->x2, C, p.Mm, C, 4 * C, b->reserve_cus);
      d.bias = b->b2; d.row_scale = b->scale_m; d.rows_per_scale = L; d.scale_bias_only = 1;
      d.resid = b->x1; d.ldr = C;
      if (b->compact_m) { d.row_map = b->map_m; d.map_len = b->Bm * L; d.map_stride = 0; }
      RC(vmvm_gemm_bf16(&d, stream));
    }
    if (b->compact_m) RC(vmvm_copy_batches_bf16(b->x1, C, b->x2, C, b->drop_m, b->nd_m, L, C, stream));
  }
  return VMVM_OK;
}

extern "C" int vmvm_swin_block_bwd(const vmvm_swin_block* b, void* stream, void* side_stream, void* fork_event) {
  RC(check_swin(b));
  if (!b->d_out || !b->d_x) return VMVM_EINVAL;
  if (side_stream && !fork_event) return VMVM_EINVAL;
  if (b->has_mlp && (!b->du || !b->dy2 || (!b->dx1_window && !b->dx1) || !b->gW1 || !b->gW2 || !b->gb1 || !b->gb2 || !b->gn2_g || !b->gn2_b || !b->u || (b->compact_m && !b->dx2c))) return VMVM_EINVAL;
  if (b->has_attn && (!b->dao || !b->dqkv || !b->dxw || !b->delta || !b->gWqkv || !b->gWproj || !b->gbqkv || !b->gbproj || !b->gn1_g || !b->gn1_b || !b->gtable)) return VMVM_EINVAL;
  if (b->has_attn && !b->dx1w) return VMVM_EINVAL;
  if (b->dx1_window && (!b->has_attn || !b->has_mlp || b->compact_a || b->compact_m || b->Lp != b->L || !b->inv)) return VMVM_EINVAL;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream), side = reinterpret_cast<hipStream_t>(side_stream);
  hipEvent_t ev = reinterpret_cast<hipEvent_t>(fork_event);
  const SwinPlan p = swin_plan(b);
  const int C = b->C, B = b->B, L = b->L, Lp = b->Lp;
  const void* dx2 = b->d_out;
  const void* dx1 = dx2;                                // d(x1): = d(x2) when the MLP branch did not run
  // ---- MLP branch: fc2 (weight gradient; input gradient x GELU' x DropPath scale), fc1, norm2
  if (b->has_mlp) {
    const void* dx2c = dx2;
    if (b->compact_m) {
      RC(vmvm_gather_rows_bf16(dx2, C, b->map_m, b->dx2c, C, p.Mm, C, 0, 0, stream));
      dx2c = b->dx2c;
    }
    RC(swin_wgrad(b, dx2c, C, b->h, 4 * C, b->gW2, b->gb2, C, 4 * C, p.Mm, b->cs_mode_m, b->cs_scale_m, b->scale_m, L, st, side, ev));
    {
      vmvm_gemm_desc d = swin_dgrad(b, dx2c, C, b->W2, b->W2T, b->du, p.Mm, C, 4 * C);
      d.act = 3; d.aux = b->u; d.ldaux = 4 * C; d.aux_code8 = b->code8;
      d.row_scale = b->scale_m; d.rows_per_scale = L;
      RC(vmvm_gemm_bf16(&d, stream));
    }
    RC(swin_wgrad(b, b->du, 4 * C, b->y2, C, b->gW1, b->gb1, 4 * C, C, p.Mm, 0, 0.f, nullptr, 0, st, side, ev));
    {
      vmvm_gemm_desc d = swin_dgrad(b, b->du, 4 * C, b->W1, b->W1T, b->dy2, p.Mm, 4 * C, C);
      RC(vmvm_gemm_bf16(&d, stream));
    }
    if (b->dx1_window) {                                // d(x1) straight in window order (dX rows permuted per clip by the inverse window map)
      vmvm_ln_bwd_desc d = swin_ln_bwd0(b, b->dy2, b->x1, b->n2_g, b->mean2, b->rstd2, b->dx1w, b->gn2_g, b->gn2_b, B * L, dx2);
      d.dx_map = b->inv; d.dx_map_len = L;
      RC(vmvm_layernorm_bwd(&d, stream));
    } else if (b->compact_m) {
      vmvm_ln_bwd_desc d = swin_ln_bwd0(b, b->dy2, b->x1, b->n2_g, b->mean2, b->rstd2, b->dx1, b->gn2_g, b->gn2_b, p.Mm, dx2);
      d.src = b->map_m; d.rows_out_per_batch = b->Bm * L; d.rows_in_per_batch = B * L; d.pad_mode = 0;
      RC(vmvm_layernorm_bwd(&d, stream));
      RC(vmvm_copy_batches_bf16(dx2, C, b->dx1, C, b->drop_m, b->nd_m, L, C, stream));
      dx1 = b->dx1;
    } else {
      vmvm_ln_bwd_desc d = swin_ln_bwd0(b, b->dy2, b->x1, b->n2_g, b->mean2, b->rstd2, b->dx1, b->gn2_g, b->gn2_b, B * L, dx2);
      RC(vmvm_layernorm_bwd(&d, stream));
      dx1 = b->dx1;
    }
  }
  if (!b->has_attn) {                                   // d(x) = d(x1): the caller passes d_x = the buffer d(x1) was written to (or d_out)
    return VMVM_OK;
  }
  // ---- attention branch: projection, window attention (+ the relative-position-table gradient), qkv, norm1
  const void* dx1w = b->dx1w;
  if (!b->dx1_window) {
    if (b->compact_a) RC(vmvm_gather_rows_bf16(dx1, C, b->src_k, b->dx1w, C, p.Ma, C, 0, 0, stream));
    else RC(vmvm_gather_rows_bf16(dx1, C, b->src, b->dx1w, C, p.Ma, C, Lp, L, stream));
  }
  RC(swin_wgrad(b, dx1w, C, b->ao, C, b->gWproj, b->gbproj, C, C, p.Ma, b->cs_mode_a, b->cs_scale_a, b->scale_a, Lp, st, side, ev));
  {
    vmvm_gemm_desc d = swin_dgrad(b, dx1w, C, b->Wproj, b->WprojT, b->dao, p.Ma, C, C);
    RC(vmvm_gemm_bf16(&d, stream));
  }
  {
    vmvm_attn_bwd_desc a;
    memset(&a, 0, sizeof(a));
    a.f = swin_attn_desc(b, p);
    a.dout = b->dao; a.ld_dout = C; a.dqkv = b->dqkv; a.ld_dqkv = 3 * C;
    a.dbias_table = b->gtable; a.delta = b->delta;
    a.dbias_ws = b->ws_main; a.dbias_ws_bytes = b->ws_main_bytes;
    // streaming windows: the table gradient is a launch of its own that no input gradient waits for -> beside the GEMMs that follow
    a.table_phase = (side && b->table_side && vmvm_attention_bwd_table_is_separate(&a)) ? 1 : 0;
    RC(vmvm_attention_bwd(&a, stream));
    if (a.table_phase == 1) {
      if (hipEventRecord(ev, st) != hipSuccess || hipStreamWaitEvent(side, ev, 0) != hipSuccess) return VMVM_EHIP;
      a.table_phase = 2;
      RC(vmvm_attention_bwd(&a, side_stream));
    }
  }
  RC(swin_wgrad(b, b->dqkv, 3 * C, b->xw, C, b->gWqkv, b->gbqkv, 3 * C, C, p.Ma, 0, 0.f, nullptr, 0, st, side, ev));
  {
    vmvm_gemm_desc d = swin_dgrad(b, b->dqkv, 3 * C, b->Wqkv, b->WqkvT, b->dxw, p.Ma, 3 * C, C);
    RC(vmvm_gemm_bf16(&d, stream));
  }
  {
    vmvm_ln_bwd_desc d = swin_ln_bwd0(b, b->dxw, b->x, b->n1_g, b->mean1, b->rstd1, b->d_x, b->gn1_g, b->gn1_b, p.Ma, b->dx1_window ? dx1w : dx1);
    d.src = p.map_a; d.rows_out_per_batch = p.rob_a; d.rows_in_per_batch = p.rib_a; d.pad_mode = 0;
    if (b->dx1_window) {
      d.add_by_out = 1;
    } else if (b->src_major) {                          // x / d(x1) / d(x) in order, only dY looked up through the (inverse) map
      if (b->compact_a) {
        if (!b->inv_k) return VMVM_EINVAL;
        RC(vmvm_invert_map(b->src_k, p.Ma, b->inv_k, B * L, stream));
        d.inv = b->inv_k;
      } else {
        if (!b->inv) return VMVM_EINVAL;
        d.inv = b->inv;
      }
      d.rows_in_total = B * L;
    }
    RC(vmvm_layernorm_bwd(&d, stream));
  }
  if (b->compact_a) RC(vmvm_copy_batches_bf16(dx1, C, b->d_x, C, b->drop_a, b->nd_a, L, C, stream));         // d(x) of the dropped clips = d(x1)
  return VMVM_OK;
}
