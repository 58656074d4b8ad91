"""Thin tensor-level wrappers over the C ABI (one Python function per libvmvm entry point).

torch is used only for device memory and the current stream; every call enqueues hand-written HIP
kernels.  All tensors must be contiguous-row CUDA tensors; bf16 tensors are torch.bfloat16."""
import ctypes as C

import torch

from . import lib as L

BF16, F32 = torch.bfloat16, torch.float32


_TYPES_CACHE = {}        # (mask-type list, device) -> int32 device tensor (no per-step host-to-device copy in masking())
_WORKSPACE = {}          # device -> caller-owned split-K scratch (set by the engine; the C ABI never allocates)
RESERVE_CUS = 0          # CUs the persistent grids leave free (vmvm_gemm_desc.reserve_cus); dist.GradReducer raises it while a collective is in flight
RESERVE_EVENT = None     # recorded on the reducer's side stream behind its last collective: once it has completed the grids take the whole chip again
RESERVE_RELEASES = 0     # how often the event (not the end of the backward) gave the CUs back (tests / profiling)
_RESERVE_POLL = 0


def reserve_cus():
    """CUs to leave free at this launch.  The reservation ends when the collectives that asked for it are DONE, not when the backward
    ends: every 8th launch polls the side stream's event (hipEventQuery, ~1 us) -- a 275 MB all-reduce is over in a few ms, the
    Video-Swin backward that follows it takes ~35."""
    global RESERVE_CUS, RESERVE_EVENT, RESERVE_RELEASES, _RESERVE_POLL
    if RESERVE_CUS and RESERVE_EVENT is not None:
        _RESERVE_POLL += 1
        if (_RESERVE_POLL & 7) == 0 and RESERVE_EVENT.query():
            RESERVE_CUS, RESERVE_EVENT = 0, None
            RESERVE_RELEASES += 1
    return RESERVE_CUS


def set_workspace(t):
    _WORKSPACE[t.device] = t


def _ld(t):
    assert t.stride(-1) == 1, "rows must be contiguous"
    return t.stride(0) if t.dim() > 1 else t.numel()


def gemm(A, B, *, a_kmajor=True, b_kmajor=True, M=None, N=None, K=None, bias=None, row_scale=None, rows_per_scale=0,
         scale_bias_only=False, act=0, aux=None, out_preact=None, resid=None, row_map=None, map_len=0, map_stride=0,
         out=None, out_dtype=BF16, accumulate=False, col_scale=1.0, col_scale_n=0, dropout_p=0.0, seed=0, offset=0,
         variant=0, out_rows=None, splitk=0, workspace=None, colsum=None, fp16=False, conv=None, fp8=False, alpha=1.0, a_relu=False, code8=False, colsum_scale=0.0):
    """C[M,N] = epilogue(sum_k A(m,k) B(n,k)); see include/vmvm.h:vmvm_gemm_desc."""
    if M is None:
        M = A.shape[0] if a_kmajor else A.shape[1]
    if K is None:
        K = B.shape[1] if conv is not None else (A.shape[1] if a_kmajor else A.shape[0])
    if N is None:
        N = B.shape[0] if b_kmajor else B.shape[1]
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty((rows, N), device=A.device, dtype=(torch.float16 if (fp16 and out_dtype == BF16) else out_dtype))
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = _ld(A), _ld(B), _ld(out)
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.bias = L.ptr(bias)
    d.row_scale, d.rows_per_scale = L.ptr(row_scale), rows_per_scale
    d.scale_bias_only = int(scale_bias_only)
    d.act = act
    d.aux, d.ldaux = L.ptr(aux), (_ld(aux) if aux is not None else 0)
    d.C2, d.ldc2 = L.ptr(out_preact), (_ld(out_preact) if out_preact is not None else 0)
    d.resid, d.ldr = L.ptr(resid), (_ld(resid) if resid is not None else 0)
    d.row_map, d.map_len, d.map_stride = L.ptr(row_map), map_len, map_stride
    d.out_fp32 = int(out.dtype == F32)
    d.accumulate = int(accumulate)
    d.col_scale, d.col_scale_n = col_scale, col_scale_n
    d.dropout_p, d.seed, d.offset = dropout_p, seed, offset
    d.variant = variant
    d.splitk = splitk
    if workspace is None and accumulate:
        workspace = _WORKSPACE.get(A.device)
    d.workspace, d.workspace_bytes = L.ptr(workspace), (workspace.numel() * workspace.element_size() if workspace is not None else 0)
    d.colsum = L.ptr(colsum)
    d.colsum_scale = float(colsum_scale)
    d.in_fp16 = int(fp16)
    d.conv_taps, d.conv_h, d.conv_w = conv if conv is not None else (0, 0, 0)
    d.in_fp8, d.alpha = int(fp8), float(alpha)
    d.a_relu = int(a_relu)
    d.aux_code8 = int(code8)
    d.reserve_cus = reserve_cus()
    if code8:
        t = aux if act == 3 else out_preact
        assert t is None or t.dtype == torch.uint8, "code8: the saved tensor is uint8"
    else:
        assert (aux is None or aux.dtype != torch.uint8) and (out_preact is None or out_preact.dtype != torch.uint8)
    L.check(L.load().vmvm_gemm_bf16(C.byref(d), L.stream()), "gemm")
    return out


def cast_fp8(x, scale=1.0, out=None):
    """bf16 -> OCP e4m3 bytes (uint8 tensor of the same shape), y = sat(x * scale)"""
    y = torch.empty(x.shape, device=x.device, dtype=torch.uint8) if out is None else out
    L.check(L.load().vmvm_cast_bf16_to_fp8(x.data_ptr(), y.data_ptr(), x.numel(), float(scale), L.stream()), "cast_fp8")
    return y


def dvae_stem_im2col(img):
    """img (n,3,H,W) f32 ImageNet-normalised -> fp16 [n*H*W, 192] rows of the pre-processed 7x7 stem (include/vmvm.h)"""
    n, _, H, W = img.shape
    cols = torch.empty((n * H * W, 192), device=img.device, dtype=torch.float16)
    L.check(L.load().vmvm_dvae_stem_im2col(img.data_ptr(), cols.data_ptr(), n, H, W, L.stream()), "dvae_stem_im2col")
    return cols


def maxpool2x2_nhwc(x, n, H, W):
    """x fp16 [n*H*W, C] (NHWC rows) -> [n*(H/2)*(W/2), C]"""
    Cc = x.shape[1]
    y = torch.empty((n * (H // 2) * (W // 2), Cc), device=x.device, dtype=x.dtype)
    L.check(L.load().vmvm_maxpool2x2_nhwc_f16(x.data_ptr(), y.data_ptr(), n, H, W, Cc, L.stream()), "maxpool2x2_nhwc")
    return y


def argmax_pairs(pairs, groups):
    """pairs f32 [M, ld] (max, column-bits) per 64-column group (gemm act=5) -> int64 [M]"""
    M = pairs.shape[0]
    out = torch.empty(M, device=pairs.device, dtype=torch.int64)
    L.check(L.load().vmvm_argmax_pairs(pairs.data_ptr(), _ld(pairs), M, groups, out.data_ptr(), L.stream()), "argmax_pairs")
    return out


def colsum(X, out, row_scale=None, rows_per_scale=0, accumulate=True, M=None, N=None, workspace=None):
    M = X.shape[0] if M is None else M
    N = X.shape[1] if N is None else N
    # row-block partials summed in a fixed order (reproducible; without scratch: f32 atomics).  `workspace`: the scratch of the STREAM
    # this call is issued on -- the engine's weight-gradient closures run on its second stream and pass that stream's own buffer
    ws = workspace if workspace is not None else _WORKSPACE.get(X.device)
    L.check(L.load().vmvm_colsum_bf16_ws(X.data_ptr(), M, N, _ld(X), L.ptr(row_scale), rows_per_scale, out.data_ptr(),
                                         int(accumulate), L.ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0, L.stream()), "colsum")
    return out


def layernorm_fwd(X, gamma, beta, eps, *, M=None, C_=None, nseg=1, src=None, rows_out_per_batch=0, rows_in_per_batch=0, pad_mode=0):
    M = X.shape[0] if M is None else M
    Cc = X.shape[1] * nseg if C_ is None else C_
    Y = torch.empty((M, Cc), device=X.device, dtype=BF16)
    mean = torch.empty(M, device=X.device, dtype=F32)
    rstd = torch.empty(M, device=X.device, dtype=F32)
    d = L.LnFwdDesc()
    d.X, d.ldx, d.Y, d.ldy = X.data_ptr(), _ld(X), Y.data_ptr(), Cc
    d.gamma, d.beta, d.eps = gamma.data_ptr(), beta.data_ptr(), eps
    d.M, d.C, d.nseg = M, Cc, nseg
    d.src, d.rows_out_per_batch, d.rows_in_per_batch, d.pad_mode = L.ptr(src), rows_out_per_batch, rows_in_per_batch, pad_mode
    d.mean, d.rstd = mean.data_ptr(), rstd.data_ptr()
    d.x_fp32 = int(X.dtype == F32)
    L.check(L.load().vmvm_layernorm_fwd(C.byref(d), L.stream()), "layernorm_fwd")
    return Y, mean, rstd


def layernorm_bwd(dY, X, gamma, mean, rstd, dgamma, dbeta, *, dX=None, rows_in=None, nseg=1, src=None, rows_out_per_batch=0,
                  rows_in_per_batch=0, pad_mode=0, dX_add=None, want_dX2=False, dropout_p=0.0, seed=0, offset=0, inv=None, dx_map=None,
                  add_by_out=False):
    M, Cc = dY.shape
    if dX is None:
        dX = torch.empty((rows_in if rows_in is not None else M, Cc // nseg), device=dY.device, dtype=BF16)
    dX2 = torch.empty((M, Cc), device=dY.device, dtype=BF16) if want_dX2 else None
    d = L.LnBwdDesc()
    d.dY, d.lddy, d.X, d.ldx = dY.data_ptr(), _ld(dY), X.data_ptr(), _ld(X)
    d.gamma, d.mean, d.rstd = gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    d.dX, d.lddx, d.dgamma, d.dbeta = dX.data_ptr(), _ld(dX), dgamma.data_ptr(), dbeta.data_ptr()
    d.M, d.C, d.nseg = M, Cc, nseg
    d.src, d.rows_out_per_batch, d.rows_in_per_batch, d.pad_mode = L.ptr(src), rows_out_per_batch, rows_in_per_batch, pad_mode
    d.dX_add, d.ldadd = L.ptr(dX_add), (_ld(dX_add) if dX_add is not None else 0)
    d.dX2, d.lddx2 = L.ptr(dX2), Cc
    d.dropout_p, d.seed, d.offset = dropout_p, seed, offset
    d.x_fp32 = int(X.dtype == F32)
    ws = _WORKSPACE.get(dY.device)
    d.workspace, d.workspace_bytes = L.ptr(ws), (ws.numel() * ws.element_size() if ws is not None else 0)
    d.reserve_cus = reserve_cus()
    d.inv, d.rows_in_total = L.ptr(inv), (dX.shape[0] if inv is not None else 0)       # source-major walk (see include/vmvm.h)
    d.dx_map, d.dx_map_len = L.ptr(dx_map), (dx_map.numel() if dx_map is not None else 0)       # identity walk: dX rows permuted per batch
    d.add_by_out = int(add_by_out)
    L.check(L.load().vmvm_layernorm_bwd(C.byref(d), L.stream()), "layernorm_bwd")
    return dX, dX2


def _attn_desc(qkv, out, lse, nseq, Lq, heads, hd, mode, scale, q_off, k_off, v_off, bias_table, rc, rc0, region, n_win,
               keymask, dropout_p, seed, offset, seq_scale, seqs_per_scale, stream_min_len=0, causal_from=0, att_colsum=None, win_layout=0, drop_mask=None):
    d = L.AttnFwdDesc()
    d.qkv, d.ld_qkv, d.q_off, d.k_off, d.v_off = qkv.data_ptr(), _ld(qkv), q_off, k_off, v_off
    d.out, d.ld_out, d.lse = out.data_ptr(), _ld(out), lse.data_ptr()
    d.nseq, d.L, d.heads, d.head_dim, d.mode, d.scale = nseq, Lq, heads, hd, mode, scale
    d.bias_table, d.table_len = L.ptr(bias_table), (bias_table.shape[0] if bias_table is not None else 0)
    d.rc, d.rc0 = L.ptr(rc), rc0
    d.region, d.n_win = L.ptr(region), n_win
    d.keymask = L.ptr(keymask)
    d.dropout_p, d.seed, d.offset = dropout_p, seed, offset
    d.seq_scale, d.seqs_per_scale = L.ptr(seq_scale), seqs_per_scale
    d.stream_min_len = stream_min_len
    d.causal_from = causal_from
    d.att_colsum, d.att_scale = L.ptr(att_colsum), 1.0 / heads
    d.win_layout = win_layout
    d.drop_mask = L.ptr(drop_mask)
    return d


_DROP_MASK_BYTES = {}


def attention_drop_mask(nseq, Lq, heads, hd, mode, dropout_p, device, stream_min_len=0, causal_from=0, att_colsum=None):
    """buffer for the forward's dropout decisions (vmvm_attn_fwd_desc.drop_mask) when this problem has a stored-decision build, else None"""
    key = (nseq, Lq, heads, hd, mode, dropout_p > 0, stream_min_len, causal_from, att_colsum is not None)
    n = _DROP_MASK_BYTES.get(key)
    if n is None:                                       # (pure host arithmetic in the library: asked once per problem shape)
        d = L.AttnFwdDesc()
        d.nseq, d.L, d.heads, d.head_dim, d.mode, d.dropout_p = nseq, Lq, heads, hd, mode, dropout_p
        d.stream_min_len, d.causal_from, d.att_colsum = stream_min_len, causal_from, L.ptr(att_colsum)
        n = _DROP_MASK_BYTES[key] = L.load().vmvm_attention_drop_mask_size(C.byref(d))
    return torch.empty((n // 4,), device=device, dtype=torch.int32) if n > 0 else None


def attention_fwd(qkv, nseq, Lq, heads, hd, mode, scale, *, q_off, k_off, v_off, bias_table=None, rc=None, rc0=0, region=None,
                  n_win=1, keymask=None, dropout_p=0.0, seed=0, offset=0, seq_scale=None, seqs_per_scale=0, stream_min_len=0, causal_from=0, att_colsum=None, win_layout=0, drop_mask=None):
    out = torch.empty((nseq * Lq, heads * hd), device=qkv.device, dtype=BF16)
    lse = torch.empty((nseq, heads, Lq), device=qkv.device, dtype=F32)
    d = _attn_desc(qkv, out, lse, nseq, Lq, heads, hd, mode, scale, q_off, k_off, v_off, bias_table, rc, rc0, region, n_win,
                   keymask, dropout_p, seed, offset, seq_scale, seqs_per_scale, stream_min_len, causal_from, att_colsum, win_layout, drop_mask)
    L.check(L.load().vmvm_attention_fwd(C.byref(d), L.stream()), "attention_fwd")
    return out, lse


def attention_bwd(dout, qkv, out, lse, nseq, Lq, heads, hd, mode, scale, *, q_off, k_off, v_off, bias_table=None, rc=None, rc0=0,
                  region=None, n_win=1, keymask=None, dropout_p=0.0, seed=0, offset=0, seq_scale=None, seqs_per_scale=0,
                  dbias_table=None, stream_min_len=0, causal_from=0, win_layout=0, drop_mask=None, table_phase=0, dqkv=None, delta=None):
    """table_phase (vmvm_attn_bwd_desc.table_phase): 1 = all but a separate table-gradient launch, returns (dqkv, delta) for the
    phase-2 call (`dqkv=`, `delta=`: the same buffers), which makes only that launch -- on whichever stream is current."""
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    delta = torch.empty((nseq, heads, Lq), device=qkv.device, dtype=F32) if delta is None else delta
    b = L.AttnBwdDesc()
    b.f = _attn_desc(qkv, out, lse, nseq, Lq, heads, hd, mode, scale, q_off, k_off, v_off, bias_table, rc, rc0, region, n_win,
                     keymask, dropout_p, seed, offset, seq_scale, seqs_per_scale, stream_min_len, causal_from, None, win_layout, drop_mask)
    b.dout, b.ld_dout, b.dqkv, b.ld_dqkv = dout.data_ptr(), _ld(dout), dqkv.data_ptr(), _ld(dqkv)
    b.dbias_table, b.delta = L.ptr(dbias_table), delta.data_ptr()
    ws = _WORKSPACE.get(qkv.device) if dbias_table is not None else None      # reproducible table gradient: per-workgroup partial tables in the caller's scratch
    b.dbias_ws, b.dbias_ws_bytes = L.ptr(ws), (ws.numel() * ws.element_size() if ws is not None else 0)
    b.table_phase = table_phase
    L.check(L.load().vmvm_attention_bwd(C.byref(b), L.stream()), "attention_bwd")
    return (dqkv, delta) if table_phase == 1 else dqkv


def attention_table_separate(Lq, mode, stream_min_len=0):
    """does the backward of this problem compute its bias-table gradient in a launch of its own (vmvm_attention_bwd_table_is_separate)?"""
    b = L.AttnBwdDesc()
    b.f.L, b.f.mode, b.f.stream_min_len = Lq, mode, stream_min_len
    b.dbias_table = 1                                   # (only tested for NULL)
    return bool(L.load().vmvm_attention_bwd_table_is_separate(C.byref(b)))


def invert_map(src, n_out):
    """inverse of a gather map: out[src[i]] = i, -1 where no i maps (int32 [n_out])"""
    out = torch.empty((n_out,), device=src.device, dtype=torch.int32)
    L.check(L.load().vmvm_invert_map(src.data_ptr(), src.numel(), out.data_ptr(), n_out, L.stream()), "invert_map")
    return out


def expand_batch_map(map_, list_, n, stride):
    """absolute row map of the clips in `list_` (first n entries) from a per-clip map: out[j*len + t] = map[t] < 0 ? -1 : map[t] + list[j]*stride"""
    out = torch.empty((n * map_.numel(),), device=map_.device, dtype=torch.int32)
    L.check(L.load().vmvm_expand_batch_map(map_.data_ptr(), map_.numel(), list_.data_ptr(), n, stride, out.data_ptr(), L.stream()), "expand_batch_map")
    return out


def copy_batches(src, dst, list_, n, rows_per_batch):
    """dst rows of the clips in `list_` (first n entries) = src rows of the same clips"""
    L.check(L.load().vmvm_copy_batches_bf16(src.data_ptr(), _ld(src), dst.data_ptr(), _ld(dst), list_.data_ptr(), n, rows_per_batch, src.shape[1], L.stream()),
            "copy_batches")
    return dst


def attn_query_row_fwd(q, kv, nseq, Lq, heads, hd, scale, *, k_off, v_off, keymask=None, dropout_p=0.0, seed=0, offset=0):
    """self-attention of ONE query position per sequence (see include/vmvm.h): q [nseq, heads*hd], kv [nseq*Lq, ld] -> (out [nseq, heads*hd],
    probs, probs_drop f32 [nseq, heads, Lq])"""
    out = torch.empty((nseq, heads * hd), device=q.device, dtype=BF16)
    probs = torch.empty((nseq, heads, Lq), device=q.device, dtype=F32)
    pdrop = torch.empty((nseq, heads, Lq), device=q.device, dtype=F32)
    L.check(L.load().vmvm_attn_query_row_fwd(q.data_ptr(), _ld(q), kv.data_ptr(), _ld(kv), k_off, v_off, L.ptr(keymask), out.data_ptr(), _ld(out),
                                             probs.data_ptr(), pdrop.data_ptr(), nseq, Lq, heads, hd, float(scale), float(dropout_p), seed, offset, L.stream()),
            "attn_query_row_fwd")
    return out, probs, pdrop


def attn_query_row_bwd(dout, q, kv, probs, pdrop, nseq, Lq, heads, hd, scale, *, k_off, v_off):
    """-> (dq [nseq, heads*hd], dkv [nseq*Lq, kv.shape[1]] with every K / V column of the layout written)"""
    dq = torch.empty((nseq, heads * hd), device=q.device, dtype=BF16)
    dkv = torch.empty((nseq * Lq, kv.shape[1]), device=q.device, dtype=BF16)
    L.check(L.load().vmvm_attn_query_row_bwd(dout.data_ptr(), _ld(dout), q.data_ptr(), _ld(q), kv.data_ptr(), _ld(kv), k_off, v_off, probs.data_ptr(),
                                             pdrop.data_ptr(), dq.data_ptr(), _ld(dq), dkv.data_ptr(), _ld(dkv), nseq, Lq, heads, hd, float(scale), L.stream()),
            "attn_query_row_bwd")
    return dq, dkv


def masking(txt, u_type, u_txt, u_rm, u_bm, types, T, h, w, p, tokens):
    """vmvm_masking: txt (B,X) i64 on the device is updated in place; returns (ans_mtm i64 (B,X), cov u8 (B,T,h,w))."""
    B, X = txt.shape
    ans = torch.empty_like(txt)
    cov = torch.empty((B, T, h, w), device=txt.device, dtype=torch.uint8)
    has_bm = int(bool((types == 1).any().item())) if types.device.type == "cpu" else 1
    key = (tuple(types.tolist()), txt.device) if types.device.type == "cpu" else None
    if key is not None and key in _TYPES_CACHE:
        tdev = _TYPES_CACHE[key]
    else:
        tdev = types.to(txt.device)
        if key is not None:
            _TYPES_CACHE[key] = tdev
    L.check(L.load().vmvm_masking(txt.data_ptr(), ans.data_ptr(), cov.data_ptr(), u_type.data_ptr(), u_txt.data_ptr(), u_rm.data_ptr(),
                                  u_bm.data_ptr(), tdev.data_ptr(), tdev.numel(), has_bm, B, X, T, h, w, float(p),
                                  tokens["cls"], tokens["sep"], tokens["pad"], tokens["mask"], L.stream()), "masking")
    return ans, cov


def patch_im2col(img, cov=None):
    B, T, _, H, W = img.shape
    cols = torch.empty((B * T * (H // 4) * (W // 4), 192), device=img.device, dtype=BF16)
    L.check(L.load().vmvm_patch_im2col(img.data_ptr(), L.ptr(cov), cols.data_ptr(), B, T, H, W, L.stream()), "im2col")
    return cols


def patch_embed_fwd(img, cov, w_bf16, bias, gamma, beta, eps):
    """PatchEmbed3D forward in one launch -> (x bf16 [M,E], z f32 [M,E], mean, rstd)   (include/vmvm.h)"""
    B, T, _, H, W = img.shape
    E = w_bf16.shape[0]
    M = B * T * (H // 4) * (W // 4)
    x = torch.empty((M, E), device=img.device, dtype=BF16)
    z = torch.empty((M, E), device=img.device, dtype=F32)
    mean = torch.empty(M, device=img.device, dtype=F32)
    rstd = torch.empty(M, device=img.device, dtype=F32)
    L.check(L.load().vmvm_patch_embed_fwd(img.data_ptr(), L.ptr(cov), w_bf16.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
                                          x.data_ptr(), z.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, T, H, W, E, L.stream()), "patch_embed_fwd")
    return x, z, mean, rstd


def encvideo_assemble(fc_out, cls, pos, len_, B, T, hw, Hd, out=None):
    out = torch.empty((B * T * (1 + hw), Hd), device=fc_out.device, dtype=BF16) if out is None else out
    L.check(L.load().vmvm_encvideo_assemble(fc_out.data_ptr(), cls.data_ptr(), pos.data_ptr(), len_.data_ptr(), out.data_ptr(),
                                            B, T, hw, Hd, L.stream()), "encvideo_assemble")
    return out


def encvideo_assemble_bwd(dpre, dcls, dpos, dlen, B, T, hw, Hd):
    dfc = torch.empty((B * T * hw, Hd), device=dpre.device, dtype=BF16)
    L.check(L.load().vmvm_encvideo_assemble_bwd(dpre.data_ptr(), dfc.data_ptr(), dcls.data_ptr(), dpos.data_ptr(), dlen.data_ptr(),
                                                B, T, hw, Hd, L.stream()), "encvideo_assemble_bwd")
    return dfc


def bert_embed(txt, word, pos, type0):
    B, X = txt.shape
    Hd = word.shape[1]
    out = torch.empty((B * X, Hd), device=txt.device, dtype=BF16)
    L.check(L.load().vmvm_bert_embed(txt.data_ptr(), word.data_ptr(), pos.data_ptr(), type0.data_ptr(), out.data_ptr(), B, X, Hd,
                                     L.stream()), "bert_embed")
    return out


def bert_embed_bwd(txt, dsum, dword, dpos, dtype0):
    B, X = txt.shape
    L.check(L.load().vmvm_bert_embed_bwd(txt.data_ptr(), dsum.data_ptr(), dword.data_ptr(), dpos.data_ptr(), dtype0.data_ptr(),
                                         B, X, dword.shape[1], L.stream()), "bert_embed_bwd")


def cross_entropy(logits, V, target, loss_sum, want_grad=True, ld_d=None):
    """loss_sum (f32[1], device) += mean CE over target != -1 ; returns dlogits (bf16) or None."""
    M = logits.shape[0]
    n_valid = torch.empty(1, device=logits.device, dtype=F32)
    L.check(L.load().vmvm_count_valid(target.data_ptr(), M, n_valid.data_ptr(), L.stream()), "count_valid")
    ld_d = ld_d or logits.shape[1]
    dlog = torch.empty((M, ld_d), device=logits.device, dtype=BF16) if want_grad else None
    L.check(L.load().vmvm_cross_entropy(logits.data_ptr(), _ld(logits), M, V, target.data_ptr(), n_valid.data_ptr(),
                                        loss_sum.data_ptr(), L.ptr(dlog), ld_d, L.stream()), "cross_entropy")
    return dlog


def vtm_ce(logits, B, O, loss_sum):
    """loss_sum (f32[1]) += mean CE of the f32 (B, O) pair-score matrix against column 0; returns its gradient, f32 [B * O]"""
    dlg = torch.empty((B * O,), device=logits.device, dtype=F32)
    L.check(L.load().vmvm_vtm_ce(logits.data_ptr(), B, O, loss_sum.data_ptr(), dlg.data_ptr(), L.stream()), "vtm_ce")
    return dlg


def pixel_l1(pred, img, cov, mask_sum, loss_sum, B, T, h, w, ps, channels=3, inv_div=1.0 / 3.0):
    dpred = torch.empty_like(pred)
    L.check(L.load().vmvm_pixel_l1(pred.data_ptr(), img.data_ptr(), cov.data_ptr(), mask_sum.data_ptr(), loss_sum.data_ptr(),
                                   dpred.data_ptr(), B, T, h, w, ps, channels, inv_div, L.stream()), "pixel_l1")
    return dpred


def feature_l1(pred, target, cov, mask_sum, loss_sum, inv_div=1.0 / 3.0):
    dpred = torch.empty_like(pred)
    M, C_ = pred.shape
    L.check(L.load().vmvm_feature_l1(pred.data_ptr(), target.data_ptr(), cov.data_ptr(), mask_sum.data_ptr(), inv_div, loss_sum.data_ptr(),
                                     dpred.data_ptr(), M, C_, L.stream()), "feature_l1")
    return dpred


def rowdot(hid, w, b, inv_temp):
    M, K = hid.shape
    out = torch.empty(M, device=hid.device, dtype=F32)
    L.check(L.load().vmvm_rowdot(hid.data_ptr(), M, K, w.data_ptr(), b.data_ptr(), inv_temp, out.data_ptr(), L.stream()), "rowdot")
    return out


def rowdot_bwd(hid, w, dout, inv_temp, dw, db, relu_mask=False):
    M, K = hid.shape
    dhid = torch.empty_like(hid)
    L.check(L.load().vmvm_rowdot_bwd(hid.data_ptr(), M, K, w.data_ptr(), dout.data_ptr(), inv_temp, dhid.data_ptr(), dw.data_ptr(),
                                     db.data_ptr(), int(relu_mask), L.stream()), "rowdot_bwd")
    return dhid


def cast_bf16(src, dst=None):
    if dst is None:
        dst = torch.empty(src.shape, device=src.device, dtype=BF16)
    L.check(L.load().vmvm_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), L.stream()), "cast")
    return dst


def cast_f32(src, dst):
    """dst f32 <- src bf16 (the reduced 16-bit gradient payload back into the f32 gradient arena)"""
    L.check(L.load().vmvm_cast_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), L.stream()), "cast_f32")
    return dst


def add_bf16(a, b, out=None):
    if out is None:
        out = torch.empty_like(a)
    L.check(L.load().vmvm_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), L.stream()), "add")
    return out


def gather_rows(src, idx, M, rows_out_per_batch=0, rows_in_per_batch=0, out=None):
    Cc = src.shape[1]
    dst = torch.empty((M, Cc), device=src.device, dtype=BF16) if out is None else out
    L.check(L.load().vmvm_gather_rows_bf16(src.data_ptr(), _ld(src), idx.data_ptr(), dst.data_ptr(), Cc, M, Cc, rows_out_per_batch,
                                           rows_in_per_batch, L.stream()), "gather_rows")
    return dst


def pool_grad(g1, g2, g3, B, O, Lv, X, txt_off, txt_list):
    """gradient of the token pool from the gradients of the gathered pass-1 / pass-2 (/ smtm) sequences (include/vmvm.h)"""
    Hd = g1.shape[1]
    out = torch.empty((B * (Lv + X), Hd), device=g1.device, dtype=BF16)
    L.check(L.load().vmvm_pool_grad_bf16(g1.data_ptr(), g2.data_ptr(), L.ptr(g3), out.data_ptr(), B, O, Lv, X, Hd, txt_off.data_ptr(),
                                         txt_list.data_ptr(), L.stream()), "pool_grad")
    return out


def scatter_add_rows(src, idx, dst_f32):
    M, Cc = src.shape
    L.check(L.load().vmvm_scatter_add_rows_bf16(src.data_ptr(), _ld(src), idx.data_ptr(), dst_f32.data_ptr(), _ld(dst_f32), M, Cc,
                                                L.stream()), "scatter_add_rows")
    return dst_f32


def gelu_bwd(dy, u):
    out = torch.empty_like(dy)
    L.check(L.load().vmvm_gelu_bwd_bf16(dy.data_ptr(), u.data_ptr(), out.data_ptr(), dy.numel(), L.stream()), "gelu_bwd")
    return out


def dropout(x, p, seed, offset):
    y = torch.empty_like(x)
    L.check(L.load().vmvm_dropout_bf16(x.data_ptr(), y.data_ptr(), x.numel(), p, seed, offset, L.stream()), "dropout")
    return y


def transpose_batched(src, dst, table):
    L.check(L.load().vmvm_transpose_batched_bf16(src.data_ptr(), dst.data_ptr(), table.data_ptr(), table.shape[0], L.stream()), "transpose_batched")


def sumsq(g, out):
    ws = _WORKSPACE.get(g.device)
    L.check(L.load().vmvm_sumsq_f32(g.data_ptr(), g.numel(), out.data_ptr(), L.ptr(ws), ws.numel() * ws.element_size() if ws is not None else 0,
                                    L.stream()), "sumsq")
    return out


def adamw(param, grad, m, v, param_bf16, *, lr, weight_decay, beta1, beta2, eps, step, sumsq_t=None, max_grad_norm=0.0, grad_scale=1.0):
    d = L.AdamWDesc()
    d.param, d.grad, d.m, d.v, d.param_bf16 = param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), L.ptr(param_bf16)
    d.n = param.numel()
    d.lr, d.weight_decay, d.beta1, d.beta2, d.eps = lr, weight_decay, beta1, beta2, eps
    d.bias_corr1, d.bias_corr2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    d.sumsq, d.max_grad_norm, d.grad_scale = L.ptr(sumsq_t), max_grad_norm, grad_scale
    L.check(L.load().vmvm_adamw(C.byref(d), L.stream()), "adamw")


def probe_tr16():
    out = torch.empty(256, device="cuda", dtype=torch.int32)
    L.check(L.load().vmvm_probe_tr16(out.data_ptr(), L.stream()), "probe")
    return out.cpu().view(64, 4)
