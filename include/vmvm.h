/* vmvm.h -- C ABI of libvmvm.so: the MI355X (gfx950) kernels of the VIOLETv2 pretraining step.
 *
 * The reference (tsujuifu/pytorch_empirical-mvm) is 100% Python/ATen and has no FFI; the
 * boundary below is the one SURVEY.md section 8(b).4 defines.  Each entry point names the reference
 * code whose arithmetic it replaces (file:line relative to the reference root).
 *
 * Conventions (all entry points):
 *   - plain pointers / sizes only, no torch types; all pointers are DEVICE pointers unless noted;
 *   - returns 0 on success or a negative VMVM_E* code; never throws;
 *   - never allocates or frees device memory; never synchronises; enqueues on `stream`
 *     (a hipStream_t passed as void*) and returns;
 *   - re-entrant, no mutable globals; RNG = Philox4x32-7 keyed by (seed, offset) arguments;
 *   - bf16 = raw uint16 storage ("bf16" in comments), f32 = float.
 */
#ifndef VMVM_H
#define VMVM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VMVM_OK 0
#define VMVM_EINVAL (-1)
#define VMVM_ENOSUPPORT (-2)
#define VMVM_EHIP (-3)

/* library version, (major<<16)|minor */
int vmvm_version(void);
/* last hipError_t seen by this thread's failing launch (diagnostics only) */
int vmvm_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * GEMM   C[M,N] = epilogue( sum_k A(m,k) * B(n,k) )   bf16 x bf16 -> f32 accumulate (MFMA).
 * Replaces every nn.Linear / 1x1 conv forward, dgrad and wgrad on the path:
 *   video_swin.py:75-81 (Mlp), :147-172 (qkv/proj), :285-287 (PatchMerging.reduction),
 *   :401 (PatchEmbed3D.proj as im2col GEMM), model.py:39 (EncVideo.fc), HF BertLayer dense layers
 *   (call site model.py:213), main_pretrain.py:146-147 (fc), :178 (decoder_pixel 1x1 conv),
 *   HF BertOnlyMLMHead (call site main_pretrain.py:236).
 * Operand layouts:  a_kmajor=1: A is [M][lda] (k contiguous);  a_kmajor=0: A is [K][lda] (m contiguous)
 *                   b_kmajor=1: B is [N][ldb] (k contiguous);  b_kmajor=0: B is [K][ldb] (n contiguous)
 *   forward  Y=X W^T : (1,1)      dgrad dX=dY W : (1,0)      wgrad dW=dY^T X : (0,0)
 * ld* must be multiples of 8 and cover the contiguous extent rounded up to 8 (operands are read in 16-byte chunks;
 * a chunk straddling the logical extent must still lie inside the row); N%4==0.
 * Epilogue order: v = acc; v += bias[n]; v *= col_scale (n < col_scale_n);
 *   act (0 none, 1 GELU-erf [C2 receives the pre-activation], 2 ReLU, 3 multiply by GELU'(aux[m,n]),
 *        4 multiply by (aux[m,n] > 0), 5 fused arg-max [fp16 builds, see a_relu below]);
 *   v *= row_scale[m/rows_per_scale] ; dropout(p, Philox(seed, offset + m*N+n)) ; + resid[dst,n] ; store at row dst where
 *   dst = row_map ? row_map[m % map_len] + (m / map_len) * map_stride : m   (dst < 0 -> row skipped).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* A; const void* B; void* C;
  int32_t M, N, K, lda, ldb, ldc;
  int32_t a_kmajor, b_kmajor;
  const float* bias;
  const float* row_scale; int32_t rows_per_scale;
  int32_t scale_bias_only;  /* 1: row_scale multiplies the bias only (DropPath 'producer' form: A rows are pre-scaled) */
  int32_t act;
  const void* aux; int32_t ldaux;
  void* C2; int32_t ldc2;
  const void* resid; int32_t ldr;
  const int32_t* row_map; int32_t map_len, map_stride;
  int32_t out_fp32;       /* C is f32 (else bf16) */
  int32_t accumulate;     /* C += (f32 output only) */
  float col_scale; int32_t col_scale_n;
  float dropout_p; uint64_t seed, offset;
  int32_t variant;        /* 0 = auto; 1 = scalar (non-transposing) LDS reads; 2 = register-staged 128^2; 3 = direct 128^2; 4 = 256^2 (2-stage);
                             5 = 256x128 3-stage; 6 = persistent 128^2; 7 = persistent 256^2 ping-pong (k-major x k-major, bf16 out) */
  int32_t splitk;         /* 0 = auto (split the reduction when C is a plain f32 accumulator), 1 = off, >1 = K slices */
  void* workspace; int64_t workspace_bytes;   /* optional caller-owned scratch for split-K slabs (splitk*M*N f32); without it
                                                 the slices combine with f32 atomics */
  /* fp16 element type + implicit-GEMM convolution (frozen dVAE tokenizer of the MVM 'vq' target, visbackbone/dalle/encoder.py):
   * in_fp16 = 1: A, B, C (bf16 slots), aux and resid are IEEE fp16.  conv_taps = 9 (3x3, padding 1): A is an NHWC activation
   * [n_img*conv_h*conv_w][lda] with C_in = K / 9 channels (multiple of 64), B is [C_out][9*C_in] with k = tap*C_in + c and
   * tap = (dy+1)*3 + (dx+1); rows whose tap falls outside the image read zeros.  Both need k-major operands. */
  int32_t in_fp16, conv_taps, conv_h, conv_w;
  float* colsum;          /* optional, a_kmajor=0 only: colsum[m] += sum_k A(m,k)  (f32 [M]).  The bias gradient of a Linear is the
                             column sum of dY, and dY is the A operand of its weight-gradient GEMM dW = dY^T X: fused, the extra
                             pass over dY (vmvm_colsum_bf16) disappears into one more MFMA per fragment on the first N tile. */
  /* fp8 operands (BASELINE config 5's "fp8 MFMA path"; forward Linear layers): in_fp8 = 1 -> A [M][lda] and B [N][ldb] are OCP
   * e4m3 bytes, k-major both, K a multiple of 128, lda / ldb (in bytes = elements) multiples of 16; C = epilogue(alpha * sum_k
   * A(m,k) B(n,k)) with alpha = 1 / (scale_A * scale_B) of the per-tensor quantisation (vmvm_cast_bf16_to_fp8); bias, act 1 (GELU,
   * + C2 pre-activation) / 2 (ReLU), resid, bf16 or f32 output.  MFMA: v_mfma_scale_f32_16x16x128_f8f6f4, block scales 1.0. */
  int32_t in_fp8; float alpha;
  /* fp16 builds only.  a_relu = 1: the A operand is read through max(., 0) (the ReLU in front of conv_1 / the output conv of the
   * dVAE encoder, visbackbone/dalle/encoder.py:27,70, applied to the fragments instead of in a pass over the activation).
   * act = 5 (1x1 only, out_fp32 = 1, N % 64 == 0): fused arg-max -- nothing is stored but, per row m and 64-column group q, the
   * pair C[m*ldc + 2q] = max_n (acc + bias)[m, 64q .. 64q+63], C[m*ldc + 2q + 1] = that column's index (int32 bit pattern; ties:
   * the smaller column); ldc >= 2 * N / 64.  vmvm_argmax_pairs() reduces the pairs to token ids (visbackbone/dalle/__init__.py:53). */
  int32_t a_relu;
  /* aux_code8 = 1 (bf16 builds, k-major x k-major, K % 64 == 0, N % 8 == 0; 128x128 persistent kernel, or the 256x256 ping-pong kernel
   * when N and the row stride of the code tensor are multiples of 16; with in_fp8 the act = 1 form on the 128x128 fp8 build): the tensor saved
   * for the GELU backward is an 8-bit code of GELU'(pre-activation) instead of the bf16 pre-activation --
   *   act = 1: C2 is uint8 [M][ldc2] and receives round((GELU'(v) + 0.13) * 255 / 1.26)   (GELU' lies in [-0.129, 1.129]);
   *   act = 3: aux is that uint8 tensor [M][ldaux], v *= -0.13 + code * 1.26 / 255.
   * A quarter of the fc1 forward's stores, half of the fc2-dgrad's operand bytes and its whole erf / exp evaluation go away; the
   * multiplier is quantised to 0.0025 absolute (the bf16 product it feeds carries 0.4 % relative). */
  int32_t aux_code8;
  /* data-parallel runs: leave `reserve_cus` compute units (rounded up to a multiple of 8 = one per XCD, at most 128) out of the
   * PERSISTENT kernels' grids, so that a collective in flight on another stream (RCCL's channel workgroups) keeps its CUs and the
   * persistent workgroups -- which split the tiles statically -- are all resident at once.  0: the whole chip.  Changes grid
   * sizes only, never results of the bf16-output classes; f32 split-K plans do not depend on it either. */
  int32_t reserve_cus;
  /* with `colsum`: the column sums are accumulated as colsum[m] += colsum_scale * sum_k A(m,k); 0 is read as 1.  (A Swin branch that runs on
   * its kept clips only has ONE DropPath scale for all its rows -- video_swin.py:46-54: 1 / keep_prob -- so its bias gradient, the
   * scale-weighted column sum of dY, stays on the weight-gradient GEMM instead of a vmvm_colsum_bf16 pass with per-clip weights.) */
  float colsum_scale;
  /* row_scale is indexed by (m + scale_row0) / rows_per_scale: a launch over the rows [scale_row0, scale_row0 + M) of a larger problem
   * (the library's own split of a GEMM into whole ping-pong rounds + a remainder launch uses it; 0 for a whole problem) */
  int32_t scale_row0;
} vmvm_gemm_desc;
int vmvm_gemm_bf16(const vmvm_gemm_desc* d, void* stream);
/* bytes of `workspace` the split-K slabs of this descriptor take (0: the problem does not split; <0: VMVM_E*).  The library never
 * allocates: the caller (PyTorch's caching allocator in the reference loop) owns every buffer, including scratch. */
int64_t vmvm_gemm_workspace_size(const vmvm_gemm_desc* d);
/* dst[i] (f32) = src[i] (bf16): the reduced bf16 gradient payload of a data-parallel step back into the f32 gradient arena (utils/deepspeed.py:11-30
 * reduces 16-bit gradients too) */
int vmvm_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* dst[i] (OCP e4m3, saturating at +-448) = src[i] (bf16) * scale ; n a multiple of 8 */
int vmvm_cast_bf16_to_fp8(const void* src, void* dst, int64_t n, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Frozen DALL-E dVAE tokenizer (MVM 'vq' target): the passes around its convolution GEMMs.
 *   vmvm_dvae_stem_im2col: DalleModel.preprocess (visbackbone/dalle/__init__.py:38-42: un-normalise, map_pixels utils.py:46-52) fused
 *     with the im2col of the 7x7 stem (encoder.py:59).  img [n][3][H][W] f32 ImageNet-normalised -> cols fp16 [n*H*W][192] with
 *     k = ky*24 + kx*3 + c (kx < 7; the other columns zero); the stem is then vmvm_gemm_bf16(in_fp16) against a [n_hid][192] weight.
 *   vmvm_maxpool2x2_nhwc_f16: nn.MaxPool2d(2) (encoder.py:62,66,70) on an NHWC fp16 activation [n][H][W][C] -> [n][H/2][W/2][C]; C % 8 == 0.
 *   vmvm_argmax_pairs: torch.argmax(z_logits, 1) (__init__.py:53) from the (maximum, column) pairs the act = 5 GEMM epilogue leaves
 *     per 64-column group: pairs f32 [M][ld], `groups` pairs per row -> out int64 [M].
 * ------------------------------------------------------------------------------------------ */
int vmvm_dvae_stem_im2col(const float* img, void* cols, int32_t n_img, int32_t H, int32_t W, void* stream);
int vmvm_maxpool2x2_nhwc_f16(const void* x, void* y, int32_t n_img, int32_t H, int32_t W, int32_t C, void* stream);
int vmvm_argmax_pairs(const float* pairs, int32_t ld, int32_t M, int32_t groups, int64_t* out, void* stream);

/* column sums  out[n] (+)= sum_m scale[m/rows_per_scale] * X[m,n]   (bias gradients)  X bf16 [M][ldx], out f32 */
int vmvm_colsum_bf16(const void* X, int32_t M, int32_t N, int32_t ldx, const float* row_scale,
                     int32_t rows_per_scale, float* out, int32_t accumulate, void* stream);
/* the same with caller-owned scratch (vmvm_colsum_workspace_size bytes): the row-block partial sums are stored and added in a fixed
 * order by a second small kernel instead of one f32 atomic per (row block, column) -- run-to-run reproducible (round 6) */
int vmvm_colsum_bf16_ws(const void* X, int32_t M, int32_t N, int32_t ldx, const float* row_scale, int32_t rows_per_scale, float* out,
                        int32_t accumulate, void* workspace, int64_t workspace_bytes, void* stream);
int64_t vmvm_colsum_workspace_size(int32_t M, int32_t N);

/* ------------------------------------------------------------------------------------------
 * Gather-LayerNorm.  Output row m (width C = nseg*Cseg) = LN( concat_s  X[src[m*nseg+s], 0:Cseg] ).
 *   src == NULL -> identity (plain LayerNorm: video_swin.py:404,478; HF BertLayer LayerNorms; model.py:71)
 *   nseg=1, src = window map  -> norm1 + pad + roll + window_partition   (video_swin.py:206-229,84-88);
 *                                 src<0 rows are written as ZEROS (F.pad happens after norm1, :216)
 *   nseg=4, src = 2x2 map     -> PatchMerging gather+cat+norm (video_swin.py:273-286);
 *                                 src<0 segments enter the LN as zeros (F.pad before norm, :277)
 * `rows_in_per_batch`/`rows_out_per_batch`: src is given for one clip and re-based per clip.
 * Saves mean/rstd (f32 [M]) for the backward.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* X; int32_t ldx;            /* bf16 input rows */
  void* Y; int32_t ldy;                  /* bf16 output [M][C] */
  const float* gamma; const float* beta; float eps;
  int32_t M, C, nseg;
  const int32_t* src; int32_t rows_out_per_batch, rows_in_per_batch;
  int32_t pad_mode;                      /* 0: src<0 -> zero OUTPUT row ; 1: src<0 -> zero INPUT segment */
  float* mean; float* rstd;
  int32_t x_fp32;                        /* X is f32 (identity map, C <= 512): PatchEmbed3D keeps its conv output in f32 */
} vmvm_ln_fwd_desc;
int vmvm_layernorm_fwd(const vmvm_ln_fwd_desc* d, void* stream);

typedef struct {
  const void* dY; int32_t lddy;          /* bf16 [M][C] */
  const void* X; int32_t ldx;            /* the forward input rows (gathered the same way) */
  const float* gamma; const float* mean; const float* rstd;
  void* dX; int32_t lddx;                /* bf16, scattered to the source rows (each written once) */
  float* dgamma; float* dbeta;           /* f32 [C], ACCUMULATED (atomics) */
  int32_t M, C, nseg;
  const int32_t* src; int32_t rows_out_per_batch, rows_in_per_batch;
  int32_t pad_mode;
  const void* dX_add; int32_t ldadd;     /* optional bf16, indexed like dX (source rows): dX = LNbwd + dX_add (residual gradient) */
  /* optional second output dX2 = dropout_mask(seed, offset + m*C+c) * dX / (1-p)  (HF hidden dropout backward) */
  void* dX2; int32_t lddx2; float dropout_p; uint64_t seed, offset;
  int32_t x_fp32;
  /* optional scratch for the dgamma/dbeta partials (>= 1280 * 2C floats): per-workgroup partials are stored
   * and summed by a second small kernel instead of 2C global atomics per workgroup.  NULL: atomics. */
  void* workspace; uint64_t workspace_bytes;
  int32_t reserve_cus;                   /* as vmvm_gemm_desc.reserve_cus: CUs left out of the resident grid (data-parallel overlap) */
  /* optional (with src, nseg = 1, C <= 256), NULL = off: the INVERSE of src -- inv[n] = output row (within its batch) whose source is
   * source row n (within its batch), or -1 -- and the number of source rows.  The kernel then walks the SOURCE rows in order (x, dX_add
   * and dX stream; only dY / mean / rstd are looked up through the map) instead of the output rows (three scattered streams): the window
   * maps of Video-Swin stage 1-2 scatter 256-512-byte rows.  Source rows with inv < 0 are not written (the caller owns them).
   * Requires pad_mode == 0 (pad slots are constant zero: the walk never visits output rows without a source) and an INJECTIVE src
   * (every source row gathered by at most one output row: vmvm_invert_map keeps one of several and the others' dY would be lost);
   * anything else is refused / undefined. */
  const int32_t* inv; int32_t rows_in_total;
  /* optional (identity walk only: src == NULL), NULL = off: dX row m is written to row dx_map[m % dx_map_len] + (m / dx_map_len) *
   * dx_map_len instead of row m -- a per-batch PERMUTATION (every entry in [0, dx_map_len), each once; M a multiple of dx_map_len).
   * Video-Swin block backward: the norm2 backward writes d(x1) straight in the block's WINDOW order (dx_map = the inverse window map),
   * which is where the projection's backward GEMMs and the norm1 backward (add_by_out) want it -- no gather pass, no natural-order copy.
   * dx_map entries are NOT range-checked (the caller builds them from its own window map); dX must not alias dX_add (rows are written
   * out of order: VMVM_EINVAL); dX2 stays at row m, so dx_map together with dX2 is refused (VMVM_ENOSUPPORT). */
  const int32_t* dx_map; int32_t dx_map_len;
  /* with src (nseg == 1): dX_add is indexed by the OUTPUT row m (like dY) instead of the source row.  Not with inv (the source-major
   * walk indexes dX_add by source row: VMVM_ENOSUPPORT). */
  int32_t add_by_out;
} vmvm_ln_bwd_desc;
int vmvm_layernorm_bwd(const vmvm_ln_bwd_desc* d, void* stream);
/* out[0..n_out) = -1, then out[src[i]] = i for every src[i] >= 0 (i < n_src): the inverse of a gather map (vmvm_ln_bwd_desc.inv) */
int vmvm_invert_map(const int32_t* src, int32_t n_src, int32_t* out, int32_t n_out, void* stream);
int64_t vmvm_layernorm_bwd_workspace_size(const vmvm_ln_bwd_desc* d);   /* bytes of `workspace` for the dgamma / dbeta partial rows */

/* ------------------------------------------------------------------------------------------
 * Fused attention (whole K/V of one (sequence, head) resident in LDS up to 448 tokens, streamed in chunks above).
 *   mode 0: Video-Swin window attention  (WindowAttention3D.forward video_swin.py:147-172):
 *           S = q k^T (q pre-scaled by the qkv GEMM epilogue) + table[rc[i]-rc[j]+rc0][h]
 *               + (region[w][i] != region[w][j] ? -100 : 0) ; softmax ; P v.   head_dim 32.
 *   mode 1: BERT self-attention (HF BertSelfAttention, call site model.py:213):
 *           S = q k^T * scale + (keymask[b][j] ? 0 : -inf) ; softmax ; dropout(p) ; P v.   head_dim 64.
 * qkv: bf16 [nseq*L][ld_qkv], q at column q_off + h*hd, k at k_off + h*hd, v at v_off + h*hd.
 * out: bf16 [nseq*L][ld_out] column h*hd.   lse: f32 [nseq][heads][L] (log-sum-exp per query row).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* qkv; int32_t ld_qkv, q_off, k_off, v_off;
  void* out; int32_t ld_out;
  float* lse;
  int32_t nseq, L, heads, head_dim, mode;
  float scale;
  /* mode 0 */
  const float* bias_table; int32_t table_len;   /* [table_len][heads] f32 */
  const int32_t* rc; int32_t rc0;               /* [L] coordinate code, idx(i,j) = rc[i]-rc[j]+rc0 */
  const uint8_t* region; int32_t n_win;         /* [n_win][L] region id or NULL ; window = seq % n_win */
  /* mode 1 */
  const uint8_t* keymask;                       /* [nseq][L] 1 = attend, or NULL */
  float dropout_p; uint64_t seed, offset;
  /* DropPath 'producer' form: out rows of sequence s are multiplied by seq_scale[s / seqs_per_scale] (or NULL) */
  const float* seq_scale; int32_t seqs_per_scale;
  /* Sequences longer than 448 tokens (Swin-L-384 windows: 1152; 16-frame 384^2 fusion sequences: 2352) run the streaming
   * kernels: K/V pass through LDS in 128-token chunks with an online softmax, same masks / bias / dropout stream as the
   * resident kernels.  stream_min_len > 0 lowers that threshold (L >= stream_min_len streams; parity tests); 0 = default. */
  int32_t stream_min_len;
  /* mode 1 only, 0 = off: the seq2seq mask of VIOLET_Base.get_attn_mask(attn_mask_type="seq2seq") (model.py:191-199; the smtm pass
   * main_pretrain.py:217-224): keys < causal_from (visual tokens) follow `keymask` for every query; a key >= causal_from (text) is
   * visible only to queries q >= causal_from with key <= q (lower triangle; the text padding mask is NOT applied there, as in the
   * reference).  Resident kernels only (L <= 448). */
  int32_t causal_from;
  /* mode 1 forward only, NULL = off: att_colsum f32 [nseq][L] ACCUMULATED += att_scale * sum_heads sum_queries P[q][key] (P after
   * attention dropout = HF's `attentions`).  With att_scale = 1/heads and the same buffer passed to every layer this is
   * VIOLET_Pretrain.get_att's `cat([a.mean(dim=1, keepdim=True) ...]).sum(dim=(1, 2))` (main_pretrain.py:211-215), the sampling
   * weights of the attention-guided 'am' masking, without materialising attentions.  Resident kernels only (L <= 448). */
  float* att_colsum; float att_scale;
  /* mode 0, L = 392 (window (8,7,7), head_dim 32), 0 = off: the caller lays the tokens of a window out in the order of
   * swin_index.win3_perm -- slot = tile * 16 + l * 8 + d, two (h, w) positions x 8 temporal slices per 16-token tile, the (h, w)
   * positions region-major for the (0,3,3) shift -- and `rc` / `region` are given in that order.  The relative-position bias of a
   * score tile is then four 8 x 8 Toeplitz blocks (one 16-byte LDS read per lane and tile from a windowed copy of the head's table
   * column instead of a register-resident bias block per wave), every tile lies in one mask region of every window, and the
   * (query tile, key tile) pairs the shift mask (video_swin.py:292-307) zeroes are skipped.  `region` must be tile-uniform (checked on
   * the host side by the caller); kernels without a win_layout build ignore the flag (they are order-agnostic).
   * MASK SEMANTICS.  The reference ADDS -100 to the logit of a cross-region pair (video_swin.py:304-306); these kernels give such a
   * pair probability exactly 0.  The two are the same function while exp(s_masked - 100 - max_live) is below the resolution of the
   * row's f32 sum -- i.e. as long as no masked RAW logit exceeds the row's largest live logit by more than ~83 (100 - ln 2^24).  Beyond
   * that the reference itself leaks attention across the shift mask (an artefact of its finite constant); this library does not
   * reproduce the leak.  tools/gpu_check.py check_attn_window_mask_boundary pins the equality with masked logits 34 above the live
   * maximum; a trained network's cross-region logits sit within a few units of its live ones.
   * SOFTMAX REFERENCE.  The forward walks the keys once against a fixed reference (the row maximum of the first live key block) and
   * repeats a sequence with the exact row maxima when the row sum comes out above 2^125 or anything is non-finite (a later logit more
   * than ~86 above the first block's): results are those of the exact softmax at any logit scale; a NaN / inf input poisons its own
   * rows only and the retry runs once (check_attn_window_spike, check_attn_window_nonfinite). */
  int32_t win_layout;
  /* mode 1, dropout_p > 0, L = 432 (the fusion encoder's exact-tile kernels; no causal_from, att_colsum, streaming), NULL = off: the
   * forward WRITES its keep / drop decisions here and the backward READS them instead of evaluating the Philox stream twice more.
   * u32 [nseq * heads][27][27][8]: record (query tile qt, key tile t) = four 64-bit lane masks of the forward's compares -- bit
   * 16 g + r of word j set = attention probability (query 16 qt + r, key 16 t + 4 g + j) is DROPPED.  The decisions are those of
   * (seed, offset) either way, so a backward WITHOUT drop_mask after a forward with it (or the reverse) gives the same gradients.
   * Any other problem with drop_mask set is refused (VMVM_ENOSUPPORT).  vmvm_attention_drop_mask_size() gives the bytes. */
  uint32_t* drop_mask;
} vmvm_attn_fwd_desc;
int vmvm_attention_fwd(const vmvm_attn_fwd_desc* d, void* stream);
/* bytes of drop_mask for this problem, 0 when the problem has no stored-decision build */
int64_t vmvm_attention_drop_mask_size(const vmvm_attn_fwd_desc* d);

typedef struct {
  vmvm_attn_fwd_desc f;        /* same problem description (out = forward output O, lse = saved) */
  const void* dout; int32_t ld_dout;   /* bf16 [nseq*L][ld_dout] */
  void* dqkv; int32_t ld_dqkv;         /* bf16, same column layout as qkv */
  float* dbias_table;                  /* mode 0: f32 [table_len][heads], ACCUMULATED (atomics unless dbias_ws is given, below) */
  float* delta;                        /* workspace f32 [nseq][heads][L] */
  /* optional scratch for a REPRODUCIBLE table gradient (round 6; win_layout = 1 kernels, i.e. the (8,7,7) windows of C2-C4): every
   * workgroup of the dQ kernel leaves its partial table here (no global atomics; inside the workgroup the waves add their running sums
   * one after the other), and a second small kernel sums the partial tables of a head in a fixed order.  Size:
   * vmvm_attention_bwd_dbias_ws_size().  NULL / too small / another kernel family: f32 atomics, run-to-run differences in the last bits. */
  void* dbias_ws; int64_t dbias_ws_bytes;
  /* Which launches this call makes (round 6).  The streaming window kernels (L > 448: config 5's (8,12,12) windows) compute the table
   * gradient in a launch of its own that nothing on the input-gradient chain waits for, so a caller with a second stream runs it
   * there, beside the GEMMs that follow: 0 = everything (default); 1 = everything EXCEPT that launch; 2 = ONLY that launch (needs the
   * delta a phase-1 call wrote).  vmvm_attention_bwd_table_is_separate() says whether the problem has such a launch -- when it has
   * not, phase 1 is the whole backward and phase 2 does nothing. */
  int32_t table_phase;
} vmvm_attn_bwd_desc;
int vmvm_attention_bwd(const vmvm_attn_bwd_desc* d, void* stream);
int64_t vmvm_attention_bwd_workspace_size(const vmvm_attn_bwd_desc* d); /* bytes of the `delta` scratch */
int64_t vmvm_attention_bwd_dbias_ws_size(const vmvm_attn_bwd_desc* d);  /* bytes of `dbias_ws` (0: this problem has no reproducible build) */
int vmvm_attention_bwd_table_is_separate(const vmvm_attn_bwd_desc* d);  /* 1: the table gradient is a launch of its own (table_phase above), 0: it is not / no table */

/* ------------------------------------------------------------------------------------------
 * Small fused kernels
 * ------------------------------------------------------------------------------------------ */
/* PatchEmbed3D im2col (video_swin.py:390-401): img f32 (B,T,3,H,W) [the reference transposes to (B,3,T,H,W)
 * first, model.py:39] -> cols bf16 [B*T*(H/4)*(W/4)][192]: columns k = c*32 + dt*16 + dy*4 + dx hold bf16(x) and columns 96+k
 * hold bf16(x - bf16(x)), so the bf16 MFMA GEMM against [W | W] sees ~16 mantissa bits of every pixel ; frame T is the zero pad. */
/* cov (optional, u8 (B,T,H/32,W/32)): covered 32x32 pixel blocks read as zeros == `img *= 1-cov` of
 * Agent_Pretrain.masking (main_pretrain.py:362-364) without writing a masked copy of the clip. */
int vmvm_patch_im2col(const float* img, const uint8_t* cov, void* cols, int32_t B, int32_t T, int32_t H, int32_t W, void* stream);
/* PatchEmbed3D forward in one kernel (visbackbone/video_swin.py:390-407): zero frame appended (:398), Conv3d(3 -> E, (2,4,4), stride
 * (1,4,4)) on MFMA with the pixels as a bf16 hi + lo pair, bias, LayerNorm(eps) as the epilogue.  img f32 (B,T,3,H,W); cov as above;
 * weight_bf16 [E][96] with k = c*32 + dt*16 + dy*4 + dx (the Conv3d weight flattened); x_out bf16 [B*T*(H/4)*(W/4)][E] = the normalised
 * tokens, z_out f32 (same shape) = the conv output the LayerNorm backward reads, mean / rstd f32 per token.  E in {32, 64, 96, 128, 192}.
 * No im2col buffer exists in the forward; the weight gradient re-derives its operand with vmvm_patch_im2col in the backward. */
int vmvm_patch_embed_fwd(const float* img, const uint8_t* cov, const void* weight_bf16, const float* bias, const float* gamma,
                         const float* beta, float eps, void* x_out, float* z_out, float* mean, float* rstd, int32_t B, int32_t T,
                         int32_t H, int32_t W, int32_t E, void* stream);

/* Device-side masking (Agent_Pretrain.masking main_pretrain.py:276-372, mask types 'rm' and 'bm'; 'am' is not built) from
 * EXPLICIT uniform draws in [0,1) (f32, e.g. torch.rand on the device), so the same draws give the same batch on the CPU oracle:
 *   u_type [B]            mask type of clip b = types[floor(u * n_types)]   (types[i]: 0 = 'rm', 1 = 'bm'; random.choice :303)
 *   u_txt  [B][X]         MLM: a non-special token (cls/sep/pad/mask ids) with u < p gets label = id and id <- mask_id (:305,:346,:354)
 *   u_rm   [B][T][1+h*w]  'rm': patch (t, i) covered iff u[t][1+i] < p   (slot 0 is the frame's cls position, never a target :348-352)
 *   u_bm   [B][T][6]      'bm': cuboid k of T: t = 1+floor(u0*(T-1)) (T>1 else 1), hh = 1+floor(u1*(2h/3-1)), ww likewise,
 *                          t1 = floor(u3*(T-t+1)), h1 = floor(u4*(h-hh+1)), w1 = floor(u5*(w-ww+1))   (numpy randint bounds :308-313)
 * txt i64 [B][X] is updated in place; ans_mtm i64 [B][X] (-1 = not a target); cov u8 [B][T][h][w] is the patch cover that
 * vmvm_patch_im2col / vmvm_pixel_l1 consume (the reference's x32 expansion :362 and `img *= 1-cov` are applied there).
 * has_bm: 'bm' is among `types` (validates h, w >= 3 as numpy's randint does in the reference). */
int vmvm_masking(int64_t* txt, int64_t* ans_mtm, uint8_t* cov, const float* u_type, const float* u_txt, const float* u_rm,
                 const float* u_bm, const int32_t* types, int32_t n_types, int32_t has_bm, int32_t B, int32_t X, int32_t T, int32_t h,
                 int32_t w, float p, int32_t cls, int32_t sep, int32_t pad, int32_t mask_id, void* stream);

/* EncVideo token assembly (model.py:58-71): pre[b,t,0,:]=cls, pre[b,t,1+p,:]=fc_out[b,t,p,:]; + pos[p] + len[t].
 * out bf16 [B*T*(1+hw)][Hd] (the following LayerNorm is vmvm_layernorm_fwd). */
int vmvm_encvideo_assemble(const void* fc_out, const float* cls, const float* pos, const float* len,
                           void* out, int32_t B, int32_t T, int32_t hw, int32_t Hd, void* stream);
/* backward: d_fc_out (bf16) = dpre rows 1.. ; dcls/dpos/dlen f32 accumulated */
int vmvm_encvideo_assemble_bwd(const void* dpre, void* d_fc_out, float* dcls, float* dpos, float* dlen,
                               int32_t B, int32_t T, int32_t hw, int32_t Hd, void* stream);

/* BERT embeddings sum (HF BertEmbeddings.forward, call site model.py:107): out = word[txt]+pos[x]+type[0] (bf16) */
int vmvm_bert_embed(const int64_t* txt, const float* word, const float* pos, const float* type0,
                    void* out, int32_t B, int32_t X, int32_t Hd, void* stream);
int vmvm_bert_embed_bwd(const int64_t* txt, const void* dsum, float* dword, float* dpos, float* dtype0,
                        int32_t B, int32_t X, int32_t Hd, void* stream);

/* Cross entropy with ignore_index=-1 (agent.py:57; main_pretrain.py:560-561), fused forward + dlogits.
 * logits f32 [M][ld] (first V columns valid); target i64 [M]; loss_sum/count f32 scalars ACCUMULATED;
 * dlogits bf16 [M][ld_d] = (softmax - onehot) / max(*n_valid,1) (columns >= V get 0; ignored rows get 0);
 * loss_sum += sum_rows (lse - logit[target]) / max(*n_valid,1); n_valid = device f32 from vmvm_count_valid. */
int vmvm_count_valid(const int64_t* target, int32_t M, float* n_valid, void* stream);
int vmvm_cross_entropy(const float* logits, int32_t ld, int32_t M, int32_t V, const int64_t* target,
                       const float* n_valid, float* loss_sum, void* dlogits, int32_t ld_d, void* stream);
/* VTM head (main_pretrain.py:262-263, 566): loss_sum += mean_i CE(logits[i, :], class 0) of the f32 (B, O) matrix of pair scores, and
 * dlogits f32 [B][O] = (softmax - onehot_0) / B -- kept in f32 (the positive's and the negatives' terms of a clip nearly cancel). */
int vmvm_vtm_ce(const float* logits, int32_t B, int32_t O, float* loss_sum, float* dlogits, void* stream);

/* MVM pixel / HOG map loss (main_pretrain.py:420-432 pixel, :453-468 hog): pred bf16 [B*T*hw][channels*ps*ps] (1x1-conv output,
 * channel = c*ps*ps+dy*ps+dx, PixelShuffle(ps) video order); target f32 (B,T,channels,H,W) = the un-masked normalised frames
 * (channels 3) or the data loader's HOG maps (channels 1); mask = cov u8 (B,T,h,w) patch cover expanded x ps.
 * coef = inv_div / (mask_sum + 1e-5), mask_sum = device f32 (pixel: 3*ps*ps*sum(cov), inv_div 1/3; hog: ps*ps*sum(cov), inv_div 1);
 * loss_sum (ACCUMULATED) += coef * sum |pred-target|*mask ; dpred bf16 = sign(pred-target)*mask*coef. */
int vmvm_pixel_l1(const void* pred, const float* img, const uint8_t* cov, const float* mask_sum,
                  float* loss_sum, void* dpred, int32_t B, int32_t T, int32_t h, int32_t w, int32_t ps,
                  int32_t channels, float inv_div, void* stream);

/* MVM feature targets (calc_mvm_loss '3d_feature' main_pretrain.py:508-526 / '2d_feature' :527-545): masked L1 between the
 * fc_mvm prediction and the frozen Swin teacher's features, bf16 [M][C] each, row = (b, t, patch); cov u8 [M] is the patch cover
 * (= max_pool2d(mvm_mask, 32).sum(1)/3 of the reference); mask_sum = device f32 = sum(cov) (covered PATCHES, as the reference's
 * mask.sum()); loss_sum (ACCUMULATED) += sum |pred-target| cov * inv_div / (mask_sum + 1e-5) with inv_div = 1/3;
 * dpred bf16 [M][C] = cov * sign(pred - target) * inv_div / (mask_sum + 1e-5). */
int vmvm_feature_l1(const void* pred, const void* target, const uint8_t* cov, const float* mask_sum, float inv_div, float* loss_sum,
                    void* dpred, int32_t M, int32_t C, void* stream);

/* VTM head tail (main_pretrain.py:147,260): logit[m] = (dot(hid[m,:], w) + b) / temp ; hid bf16 [M][K] */
int vmvm_rowdot(const void* hid, int32_t M, int32_t K, const float* w, const float* b, float inv_temp,
                float* out, void* stream);
/* dhid = dout*inv_temp*w, masked by (hid > 0) when relu_mask (hid is then the ReLU output feeding the dot) */
int vmvm_rowdot_bwd(const void* hid, int32_t M, int32_t K, const float* w, const float* dout, float inv_temp,
                    void* dhid, float* dw, float* db, int32_t relu_mask, void* stream);

/* generic helpers */
int vmvm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst[m,:] = src[idx'(m),:] (zeros when idx < 0); with rows_out_per_batch > 0 the map is per clip:
 * idx'(m) = idx[m % rows_out_per_batch] + (m / rows_out_per_batch) * rows_in_per_batch  (window_partition of a gradient) */
int vmvm_gather_rows_bf16(const void* src, int32_t ld_src, const int32_t* idx, void* dst, int32_t ld_dst,
                          int32_t M, int32_t C, int32_t rows_out_per_batch, int32_t rows_in_per_batch, void* stream);
int vmvm_add_bf16(const void* a, const void* b, void* out, int64_t n, void* stream);
/* dst_f32[idx[m],:] += src[m,:] (atomics; idx < 0 skipped) -- gradient of a row gather with repeated sources */
int vmvm_scatter_add_rows_bf16(const void* src, int32_t ld_src, const int32_t* idx, float* dst, int32_t ld_dst,
                               int32_t M, int32_t C, void* stream);
/* Gradient of the token pool = backward of the sequence assembly of VIOLET_Pretrain.forward (main_pretrain.py:243-259).  g1: gradient of
 * the B pass-1 sequences [video i ; text i] (bf16 [B*(Lv+X)][Hd]); g2: of the B*O pass-2 sequences p = i*O + o = [video i ; text tj(p)];
 * g3: optional third pass laid out as g1 (smtm).  out (bf16 [B*Lv + B*X][Hd]): video row (i, t) = g1 + g3 + sum_o g2[(i*O+o)], text row
 * (j, x) = g1 + g3 + the pass-2 sequences txt_list[txt_off[j] .. txt_off[j+1]) (CSR over the text index of each pair, int32 on the
 * device).  f32 sums in registers, no atomics. */
int vmvm_pool_grad_bf16(const void* g1, const void* g2, const void* g3, void* out, int32_t B, int32_t O, int32_t Lv, int32_t X, int32_t Hd,
                        const int32_t* txt_off, const int32_t* txt_list, void* stream);
/* out = dy * GELU'(u)  (backward of the MLM head transform activation) */
int vmvm_gelu_bwd_bf16(const void* dy, const void* u, void* out, int64_t n, void* stream);
/* elementwise dropout y = keep(seed, offset+i) ? x/(1-p) : 0  (HF embedding dropout; main_pretrain.py:146 fc[0]).
 * The same call with dy as x is the backward. */
int vmvm_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer (agent.py:84-113,181-193): fused global grad-norm + clip + AdamW over a flat f32 arena,
 * also refreshing the bf16 compute copy.  seg_* describe contiguous segments (4 param groups).
 * ------------------------------------------------------------------------------------------ */
/* batched bf16 transpose over a flat arena: for each tile entry {offset, N, K, (tile_row<<16)|tile_col} (64x64 tiles, N and K
 * multiples of 8) writes dst[offset + k*N + n] = src[offset + n*K + k] -- keeps W^T copies of all weights so dgrad (dX = dY W) runs
 * on the k-major fast path */
int vmvm_transpose_batched_bf16(const void* src, void* dst, const int32_t* table, int32_t ntiles, void* stream);
/* *out_accum += sum g[i]^2 (clip_grad_norm_, agent.py:188).  With a scratch buffer (>= 8 KiB) the per-workgroup partials are summed
 * in a fixed order, so data-parallel ranks holding identical (all-reduced) gradients get bit-identical clip coefficients and their
 * replicas stay bit-identical; workspace NULL: f32 atomics (order-dependent rounding). */
int vmvm_sumsq_f32(const float* g, int64_t n, float* out_accum, void* workspace, uint64_t workspace_bytes, void* stream);
int64_t vmvm_sumsq_workspace_size(int64_t n);                            /* bytes of `workspace` for the fixed-order partial sums */
typedef struct {
  float* param; const float* grad; float* m; float* v; void* param_bf16;
  int64_t n;
  float lr, weight_decay, beta1, beta2, eps, bias_corr1, bias_corr2;
  const float* sumsq; float max_grad_norm;     /* clip coef = min(1, max_norm/(sqrt(*sumsq)+1e-6)) ; <=0 disables */
  float grad_scale;                            /* multiply grads (e.g. 1/world_size) before everything */
} vmvm_adamw_desc;
int vmvm_adamw(const vmvm_adamw_desc* d, void* stream);

/* DropPath dead-clip elimination (video_swin.py:46-63,250-263: a clip whose stochastic-depth draw is 0 contributes nothing through that
 * branch, forward or backward).  vmvm_expand_batch_map: out[j*len + t] = map[t] < 0 ? -1 : map[t] + list[j]*stride -- the per-clip window map
 * of vmvm_layernorm_fwd / the proj GEMM's row_map turned into the absolute row map of the KEPT clips `list`, so the branch runs on them
 * only.  vmvm_copy_batches_bf16: dst rows of the listed (dropped) clips = src rows (their identity path); rows_per_batch rows of C each. */
int vmvm_expand_batch_map(const int32_t* map, int32_t len, const int32_t* list, int32_t n, int32_t stride, int32_t* out, void* stream);
int vmvm_copy_batches_bf16(const void* src, int32_t ld_src, void* dst, int32_t ld_dst, const int32_t* list, int32_t n, int32_t rows_per_batch,
                           int32_t C, void* stream);

/* Self-attention of ONE query position per sequence (HF BertSelfAttention, call site model.py:213, restricted to a single query row).
 * The VTM pass reads the fusion encoder's output at the text [CLS] position only (main_pretrain.py:260), so in its last layer every other
 * query row is dead code while K / V of all positions are still needed.  q: bf16 [nseq][ld_q] (head h at column h * head_dim); kv: bf16
 * [nseq * L][ld_kv] with K of head h at column k_off + h * head_dim and V at v_off + h * head_dim; keymask u8 [nseq][L] or NULL;
 * out: bf16 [nseq][ld_out]; probs / probs_drop: f32 [nseq][heads][L], the softmax before and after dropout (saved for the backward).
 * Dropout: the Philox stream of the GEMM epilogues (element = ((seq * heads + h) * L + key), 8-element blocks from `offset`).
 * bwd: dq bf16 [nseq][ld_dq]; dkv bf16 [nseq * L][ld_dkv], same column layout as kv, EVERY K / V element written (masked keys: zeros).
 * head_dim 32 or 64, L <= 8192. */
int vmvm_attn_query_row_fwd(const void* q, int32_t ld_q, const void* kv, int32_t ld_kv, int32_t k_off, int32_t v_off, const uint8_t* keymask,
                            void* out, int32_t ld_out, float* probs, float* probs_drop, int32_t nseq, int32_t L, int32_t heads, int32_t head_dim,
                            float scale, float dropout_p, uint64_t seed, uint64_t offset, void* stream);
int vmvm_attn_query_row_bwd(const void* dout, int32_t ld_dout, const void* q, int32_t ld_q, const void* kv, int32_t ld_kv, int32_t k_off, int32_t v_off,
                            const float* probs, const float* probs_drop, void* dq, int32_t ld_dq, void* dkv, int32_t ld_dkv, int32_t nseq, int32_t L,
                            int32_t heads, int32_t head_dim, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Block-level entry points (round 6).  One call = every launch of one HF BertLayer of the fusion encoder
 * (model.py:204-214 via transformers BertLayer: self-attention with key mask + attention dropout, BertSelfOutput, BertIntermediate
 * (GELU), BertOutput; post-LayerNorm), forward or backward -- the same kernels through the same descriptors the per-kernel entry points
 * above take (which stay: the tests pin the two forms against each other bit for bit), but ONE descriptor fill and ONE foreign call
 * per layer and direction instead of ~10 / ~22.  Every buffer is the caller's (activations that the backward re-reads are OUTPUTS of
 * the forward call); nothing is allocated, nothing synchronises the host.
 *
 * Backward and the second stream: the four weight-gradient GEMMs (+ their fused bias column sums and split-K reduces) feed nothing
 * downstream, so vmvm_bert_layer_bwd enqueues them on `side_stream` behind `fork_event` records on `stream` (one event handle, re-
 * recorded in front of each of the four: a record / wait pair is consumed in stream order) while the input-gradient chain stays on
 * `stream` (DESIGN 5 "Two streams").  side_stream == NULL: everything on `stream`.  The caller joins the streams before it reads the
 * gradients, and keeps every buffer alive until the side stream has passed.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t nseq, L, hidden, heads, ffn;           /* M = nseq * L rows; head_dim = hidden / heads (64) */
  /* parameters: bf16 compute copies, k-major [out][in]; optional transposed copies [in][out] for the input-gradient GEMMs (NULL: the
   * m/n-major form on the un-transposed weight); f32 biases and LayerNorm parameters */
  const void *Wqkv, *Wo, *W1, *W2;               /* [3H][H], [H][H], [F][H], [H][F] */
  const void *WqkvT, *WoT, *W1T, *W2T;
  const float *bqkv, *bo, *b1, *b2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  float ln_eps;
  /* f32 gradient accumulators (backward) */
  float *gWqkv, *gWo, *gW1, *gW2, *gbqkv, *gbo, *gb1, *gb2, *gln1_g, *gln1_b, *gln2_g, *gln2_b;
  /* forward: x in; everything else out (bf16 unless noted) */
  const void* x;                                 /* [M][H] */
  void* qkv;                                     /* [M][3H] */
  void* ctx; float* lse;                         /* [M][H], f32 [nseq][heads][L] */
  void* a; void* x1; float *mean1, *rstd1;       /* attention block: dense + dropout + residual, its LayerNorm */
  void* u; int32_t code8;                        /* saved for the GELU backward: uint8 GELU' codes (code8 = 1) or the bf16 pre-activation, [M][F] */
  void* h; void* f; void* x2; float *mean2, *rstd2;   /* FFN: GELU(fc1) [M][F], fc2 + dropout + residual, output LayerNorm = the layer's output */
  const uint8_t* keymask; int32_t causal_from; float* att_colsum; void* drop_mask;      /* as vmvm_attn_fwd_desc */
  float p_hidden, p_attn; uint64_t seed, off_attn, off_1, off_2;                        /* dropout (0 = off) and Philox offsets */
  int32_t in_fp8; const void *Wqkv8, *W18; void *x8, *x18; float a8_scale, w8_scale;    /* opt-in e4m3 forward of the qkv / fc1 GEMMs (x8 / x18: e4m3 scratch [M][H]) */
  /* backward: d_out in; d_x out (the FIRST gradient of x: written, not accumulated); the rest is scratch */
  const void* d_out; void* d_x;
  void *df, *dfm, *du, *dx1, *da, *dam, *dctx, *dqkv;   /* dfm / dam only with p_hidden > 0 (else unused) */
  float* delta;                                  /* f32 [nseq][heads][L] */
  void* ws_main; int64_t ws_main_bytes;          /* scratch of `stream` (LayerNorm partial rows) */
  void* ws_side; int64_t ws_side_bytes;          /* scratch of `side_stream` (split-K slabs + bias-gradient partials) */
  int32_t reserve_cus;
} vmvm_bert_layer;
int vmvm_bert_layer_fwd(const vmvm_bert_layer* l, void* stream);
int vmvm_bert_layer_bwd(const vmvm_bert_layer* l, void* stream, void* side_stream, void* fork_event);

/* One Video-Swin block (SwinTransformerBlock3D.forward video_swin.py:206-263: norm1 -> shift / pad / window partition -> WindowAttention3D
 * -> reverse -> DropPath residual; norm2 -> Mlp -> DropPath residual), forward or backward, per call.  The caller decides the schedule
 * (which clips each branch runs on: DropPath draws are host-side, DESIGN 5 "Dead clips of DropPath") and owns every buffer; this call
 * issues the launches -- the same kernels through the same descriptors as the per-kernel entry points (pinned bit for bit).
 *   attention branch:  has_attn = 0: every clip dropped, x1 = x.  compact_a = 1: the branch runs on the Bk clips of `kept_a` (padding
 *     entries -1) through the absolute row map src_k = expand(src, kept_a) (written by the forward, re-read by the backward); the nd_a
 *     clips of `drop_a` are copied.  compact_a = 0: all B clips, per-clip scales `scale_a` (NULL: no DropPath).
 *   MLP branch: the same with has_mlp / compact_m / Bm / kept_m / drop_m / nd_m / scale_m and the identity map `idm` -> map_m.
 *   bias gradients: cs_mode = 0: plain fused column sum; 1: fused, times cs_scale (one DropPath scale for every row); 2: separate pass
 *     weighted by the per-clip scales.
 *   backward form: dx1_window = 1 (both branches on every clip, un-padded windows): norm2's backward writes d(x1) in window order through
 *     `inv` (vmvm_ln_bwd_desc.dx_map) and norm1's backward takes it by output row; else the gathered form, with the source-major
 *     LayerNorm walk when src_major = 1 (`inv`, or inv_k = invert(src_k) in the compact case). */
typedef struct {
  int32_t B, L, Lp, N, nW, C, heads; float qscale; int32_t win_layout, rc0, table_len, code8;
  int32_t has_attn, compact_a, Bk, nd_a, cs_mode_a; float cs_scale_a;
  const float* scale_a; const int32_t* kept_a; const int32_t* drop_a;      /* scale_a: [Bk] kept-clip scales (compact) or [B] per-clip scales */
  int32_t has_mlp, compact_m, Bm, nd_m, cs_mode_m; float cs_scale_m;
  const float* scale_m; const int32_t* kept_m; const int32_t* drop_m;
  int32_t dx1_window, src_major;
  const int32_t *src, *inv, *idm, *rc; const uint8_t* region;              /* [Lp], [L], [L], [N], [nW][N] or NULL */
  const void *Wqkv, *Wproj, *W1, *W2, *WqkvT, *WprojT, *W1T, *W2T;          /* bf16 [3C][C], [C][C], [4C][C], [C][4C]; transposed copies or NULL */
  const float *bqkv, *bproj, *b1, *b2, *n1_g, *n1_b, *n2_g, *n2_b, *table;  /* table: f32 [table_len][heads] */
  float *gWqkv, *gWproj, *gW1, *gW2, *gbqkv, *gbproj, *gb1, *gb2, *gn1_g, *gn1_b, *gn2_g, *gn2_b, *gtable;
  const void* x;                                                            /* bf16 [B*L][C] */
  void* xw; float *mean1, *rstd1; void* qkv; void* ao; float* lse; int32_t* src_k;   /* attention branch, rows = (compact_a ? Bk : B) * Lp */
  void* x1;                                                                 /* [B*L][C] (= x when has_attn = 0: pass the same pointer) */
  void* y2; float *mean2, *rstd2; void* u; void* h; int32_t* map_m;         /* MLP branch, rows = (compact_m ? Bm : B) * L; u: uint8 codes or bf16, NULL = not saved */
  void* x2;                                                                 /* [B*L][C] (= x1 when has_mlp = 0) */
  const void* d_out; void* d_x;                                             /* backward: d(x2) in, d(x) out (written) */
  void *dx2c, *du, *dy2, *dx1, *dx1w, *dao, *dqkv, *dxw; float* delta; int32_t* inv_k;
  void* ws_main; int64_t ws_main_bytes; void* ws_side; int64_t ws_side_bytes; int32_t reserve_cus;
  int32_t table_side;          /* 1: a table-gradient launch of its own (vmvm_attn_bwd_desc.table_phase) goes to the side stream; the caller keeps dao / qkv / lse / delta alive for it */
} vmvm_swin_block;
int vmvm_swin_block_fwd(const vmvm_swin_block* b, void* stream);
int vmvm_swin_block_bwd(const vmvm_swin_block* b, void* stream, void* side_stream, void* fork_event);

/* hardware probe used by tests: dumps the lane mapping of ds_read_b64_tr_b16 (out: 64*4 int32) */
int vmvm_probe_tr16(int32_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
