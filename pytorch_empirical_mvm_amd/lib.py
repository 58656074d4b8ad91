"""ctypes binding of libvmvm.so (include/vmvm.h).  No CPU fallback: if the HIP library is missing or a
call fails this raises -- the product path never routes through the oracle or eager PyTorch math."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VMVM_LIB") or os.path.join(_HERE, "libvmvm.so")      # VMVM_LIB: experiment builds only

c_void_p, c_int, c_float, c_u64, c_i64 = C.c_void_p, C.c_int32, C.c_float, C.c_uint64, C.c_int64


class GemmDesc(C.Structure):
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
                ("M", c_int), ("N", c_int), ("K", c_int), ("lda", c_int), ("ldb", c_int), ("ldc", c_int),
                ("a_kmajor", c_int), ("b_kmajor", c_int),
                ("bias", c_void_p),
                ("row_scale", c_void_p), ("rows_per_scale", c_int),
                ("scale_bias_only", c_int),
                ("act", c_int),
                ("aux", c_void_p), ("ldaux", c_int),
                ("C2", c_void_p), ("ldc2", c_int),
                ("resid", c_void_p), ("ldr", c_int),
                ("row_map", c_void_p), ("map_len", c_int), ("map_stride", c_int),
                ("out_fp32", c_int), ("accumulate", c_int),
                ("col_scale", c_float), ("col_scale_n", c_int),
                ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64),
                ("variant", c_int), ("splitk", c_int), ("workspace", c_void_p), ("workspace_bytes", c_i64),
                ("in_fp16", c_int), ("conv_taps", c_int), ("conv_h", c_int), ("conv_w", c_int), ("colsum", c_void_p),
                ("in_fp8", c_int), ("alpha", c_float), ("a_relu", c_int), ("aux_code8", c_int), ("reserve_cus", c_int), ("colsum_scale", c_float), ("scale_row0", c_int)]


class LnFwdDesc(C.Structure):
    _fields_ = [("X", c_void_p), ("ldx", c_int), ("Y", c_void_p), ("ldy", c_int),
                ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
                ("M", c_int), ("C", c_int), ("nseg", c_int),
                ("src", c_void_p), ("rows_out_per_batch", c_int), ("rows_in_per_batch", c_int),
                ("pad_mode", c_int), ("mean", c_void_p), ("rstd", c_void_p), ("x_fp32", c_int)]


class LnBwdDesc(C.Structure):
    _fields_ = [("dY", c_void_p), ("lddy", c_int), ("X", c_void_p), ("ldx", c_int),
                ("gamma", c_void_p), ("mean", c_void_p), ("rstd", c_void_p),
                ("dX", c_void_p), ("lddx", c_int), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("M", c_int), ("C", c_int), ("nseg", c_int),
                ("src", c_void_p), ("rows_out_per_batch", c_int), ("rows_in_per_batch", c_int),
                ("pad_mode", c_int),
                ("dX_add", c_void_p), ("ldadd", c_int),
                ("dX2", c_void_p), ("lddx2", c_int), ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64), ("x_fp32", c_int),
                ("workspace", c_void_p), ("workspace_bytes", c_u64), ("reserve_cus", c_int), ("inv", c_void_p), ("rows_in_total", c_int),
                ("dx_map", c_void_p), ("dx_map_len", c_int), ("add_by_out", c_int)]


class AttnFwdDesc(C.Structure):
    _fields_ = [("qkv", c_void_p), ("ld_qkv", c_int), ("q_off", c_int), ("k_off", c_int), ("v_off", c_int),
                ("out", c_void_p), ("ld_out", c_int), ("lse", c_void_p),
                ("nseq", c_int), ("L", c_int), ("heads", c_int), ("head_dim", c_int), ("mode", c_int),
                ("scale", c_float),
                ("bias_table", c_void_p), ("table_len", c_int),
                ("rc", c_void_p), ("rc0", c_int),
                ("region", c_void_p), ("n_win", c_int),
                ("keymask", c_void_p),
                ("dropout_p", c_float), ("seed", c_u64), ("offset", c_u64),
                ("seq_scale", c_void_p), ("seqs_per_scale", c_int), ("stream_min_len", c_int), ("causal_from", c_int), ("att_colsum", c_void_p), ("att_scale", c_float), ("win_layout", c_int), ("drop_mask", c_void_p)]


class AttnBwdDesc(C.Structure):
    _fields_ = [("f", AttnFwdDesc), ("dout", c_void_p), ("ld_dout", c_int), ("dqkv", c_void_p), ("ld_dqkv", c_int),
                ("dbias_table", c_void_p), ("delta", c_void_p), ("dbias_ws", c_void_p), ("dbias_ws_bytes", c_i64), ("table_phase", c_int)]


class BertLayer(C.Structure):
    """vmvm_bert_layer (include/vmvm.h): one fusion-encoder layer, forward or backward, per foreign call"""
    _fields_ = [("nseq", c_int), ("L", c_int), ("hidden", c_int), ("heads", c_int), ("ffn", c_int), ("Wqkv", c_void_p), ("Wo", c_void_p),
                ("W1", c_void_p), ("W2", c_void_p), ("WqkvT", c_void_p), ("WoT", c_void_p), ("W1T", c_void_p), ("W2T", c_void_p), ("bqkv", c_void_p),
                ("bo", c_void_p), ("b1", c_void_p), ("b2", c_void_p), ("ln1_g", c_void_p), ("ln1_b", c_void_p), ("ln2_g", c_void_p),
                ("ln2_b", c_void_p), ("ln_eps", c_float), ("gWqkv", c_void_p), ("gWo", c_void_p), ("gW1", c_void_p), ("gW2", c_void_p),
                ("gbqkv", c_void_p), ("gbo", c_void_p), ("gb1", c_void_p), ("gb2", c_void_p), ("gln1_g", c_void_p), ("gln1_b", c_void_p),
                ("gln2_g", c_void_p), ("gln2_b", c_void_p), ("x", c_void_p), ("qkv", c_void_p), ("ctx", c_void_p), ("lse", c_void_p),
                ("a", c_void_p), ("x1", c_void_p), ("mean1", c_void_p), ("rstd1", c_void_p), ("u", c_void_p), ("code8", c_int), ("h", c_void_p),
                ("f", c_void_p), ("x2", c_void_p), ("mean2", c_void_p), ("rstd2", c_void_p), ("keymask", c_void_p), ("causal_from", c_int),
                ("att_colsum", c_void_p), ("drop_mask", c_void_p), ("p_hidden", c_float), ("p_attn", c_float), ("seed", c_u64), ("off_attn", c_u64),
                ("off_1", c_u64), ("off_2", c_u64), ("in_fp8", c_int), ("Wqkv8", c_void_p), ("W18", c_void_p), ("x8", c_void_p), ("x18", c_void_p),
                ("a8_scale", c_float), ("w8_scale", c_float), ("d_out", c_void_p), ("d_x", c_void_p), ("df", c_void_p), ("dfm", c_void_p),
                ("du", c_void_p), ("dx1", c_void_p), ("da", c_void_p), ("dam", c_void_p), ("dctx", c_void_p), ("dqkv", c_void_p),
                ("delta", c_void_p), ("ws_main", c_void_p), ("ws_main_bytes", c_i64), ("ws_side", c_void_p), ("ws_side_bytes", c_i64),
                ("reserve_cus", c_int)]


class SwinBlock(C.Structure):
    """vmvm_swin_block (include/vmvm.h): one Video-Swin block, forward or backward, per foreign call"""
    _fields_ = [("B", c_int), ("L", c_int), ("Lp", c_int), ("N", c_int), ("nW", c_int), ("C", c_int), ("heads", c_int), ("qscale", c_float),
                ("win_layout", c_int), ("rc0", c_int), ("table_len", c_int), ("code8", c_int), ("has_attn", c_int), ("compact_a", c_int),
                ("Bk", c_int), ("nd_a", c_int), ("cs_mode_a", c_int), ("cs_scale_a", c_float), ("scale_a", c_void_p), ("kept_a", c_void_p),
                ("drop_a", c_void_p), ("has_mlp", c_int), ("compact_m", c_int), ("Bm", c_int), ("nd_m", c_int), ("cs_mode_m", c_int),
                ("cs_scale_m", c_float), ("scale_m", c_void_p), ("kept_m", c_void_p), ("drop_m", c_void_p), ("dx1_window", c_int),
                ("src_major", c_int), ("src", c_void_p), ("inv", c_void_p), ("idm", c_void_p), ("rc", c_void_p), ("region", c_void_p),
                ("Wqkv", c_void_p), ("Wproj", c_void_p), ("W1", c_void_p), ("W2", c_void_p), ("WqkvT", c_void_p), ("WprojT", c_void_p),
                ("W1T", c_void_p), ("W2T", c_void_p), ("bqkv", c_void_p), ("bproj", c_void_p), ("b1", c_void_p), ("b2", c_void_p),
                ("n1_g", c_void_p), ("n1_b", c_void_p), ("n2_g", c_void_p), ("n2_b", c_void_p), ("table", c_void_p), ("gWqkv", c_void_p),
                ("gWproj", c_void_p), ("gW1", c_void_p), ("gW2", c_void_p), ("gbqkv", c_void_p), ("gbproj", c_void_p), ("gb1", c_void_p),
                ("gb2", c_void_p), ("gn1_g", c_void_p), ("gn1_b", c_void_p), ("gn2_g", c_void_p), ("gn2_b", c_void_p), ("gtable", c_void_p),
                ("x", c_void_p), ("xw", c_void_p), ("mean1", c_void_p), ("rstd1", c_void_p), ("qkv", c_void_p), ("ao", c_void_p), ("lse", c_void_p),
                ("src_k", c_void_p), ("x1", c_void_p), ("y2", c_void_p), ("mean2", c_void_p), ("rstd2", c_void_p), ("u", c_void_p), ("h", c_void_p),
                ("map_m", c_void_p), ("x2", c_void_p), ("d_out", c_void_p), ("d_x", c_void_p), ("dx2c", c_void_p), ("du", c_void_p),
                ("dy2", c_void_p), ("dx1", c_void_p), ("dx1w", c_void_p), ("dao", c_void_p), ("dqkv", c_void_p), ("dxw", c_void_p),
                ("delta", c_void_p), ("inv_k", c_void_p), ("ws_main", c_void_p), ("ws_main_bytes", c_i64), ("ws_side", c_void_p),
                ("ws_side_bytes", c_i64), ("reserve_cus", c_int), ("table_side", c_int)]


class AdamWDesc(C.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("m", c_void_p), ("v", c_void_p), ("param_bf16", c_void_p),
                ("n", c_i64),
                ("lr", c_float), ("weight_decay", c_float), ("beta1", c_float), ("beta2", c_float), ("eps", c_float),
                ("bias_corr1", c_float), ("bias_corr2", c_float),
                ("sumsq", c_void_p), ("max_grad_norm", c_float), ("grad_scale", c_float)]


_PROTOS = {
    "vmvm_version": ([], c_int),
    "vmvm_last_hip_error": ([], c_int),
    "vmvm_gemm_bf16": ([C.POINTER(GemmDesc), c_void_p], c_int),
    "vmvm_colsum_bf16": ([c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p], c_int),
    "vmvm_colsum_bf16_ws": ([c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_i64, c_void_p], c_int),
    "vmvm_colsum_workspace_size": ([c_int, c_int], c_i64),
    "vmvm_layernorm_fwd": ([C.POINTER(LnFwdDesc), c_void_p], c_int),
    "vmvm_layernorm_bwd": ([C.POINTER(LnBwdDesc), c_void_p], c_int),
    "vmvm_attention_fwd": ([C.POINTER(AttnFwdDesc), c_void_p], c_int),
    "vmvm_attention_bwd": ([C.POINTER(AttnBwdDesc), c_void_p], c_int),
    "vmvm_patch_embed_fwd": ([c_void_p] * 6 + [c_float] + [c_void_p] * 4 + [c_int] * 5 + [c_void_p], c_int),
    "vmvm_patch_im2col": ([c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_masking": ([c_void_p] * 8 + [c_int] * 7 + [c_float] + [c_int] * 4 + [c_void_p], c_int),
    "vmvm_encvideo_assemble": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_encvideo_assemble_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_bert_embed": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_bert_embed_bwd": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_count_valid": ([c_void_p, c_int, c_void_p, c_void_p], c_int),
    "vmvm_cross_entropy": ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "vmvm_vtm_ce": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "vmvm_pixel_l1": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p], c_int),
    "vmvm_feature_l1": ([c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p], c_int),
    "vmvm_rowdot": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p], c_int),
    "vmvm_rowdot_bwd": ([c_void_p, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "vmvm_cast_bf16_to_fp8": ([c_void_p, c_void_p, c_i64, c_float, c_void_p], c_int),
    "vmvm_dvae_stem_im2col": ([c_void_p, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_maxpool2x2_nhwc_f16": ([c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_argmax_pairs": ([c_void_p, c_int, c_int, c_int, c_void_p, c_void_p], c_int),
    "vmvm_cast_f32_to_bf16": ([c_void_p, c_void_p, c_i64, c_void_p], c_int),
    "vmvm_gather_rows_bf16": ([c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_cast_bf16_to_f32": ([c_void_p, c_void_p, c_i64, c_void_p], c_int),
    "vmvm_expand_batch_map": ([c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p], c_int),
    "vmvm_invert_map": ([c_void_p, c_int, c_void_p, c_int, c_void_p], c_int),
    "vmvm_copy_batches_bf16": ([c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_attn_query_row_fwd": ([c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_float, c_float, c_u64, c_u64, c_void_p], c_int),
    "vmvm_attn_query_row_bwd": ([c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int,
                                c_int, c_int, c_float, c_void_p], c_int),
    "vmvm_add_bf16": ([c_void_p, c_void_p, c_void_p, c_i64, c_void_p], c_int),
    "vmvm_pool_grad_bf16": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p], c_int),
    "vmvm_scatter_add_rows_bf16": ([c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p], c_int),
    "vmvm_gelu_bwd_bf16": ([c_void_p, c_void_p, c_void_p, c_i64, c_void_p], c_int),
    "vmvm_dropout_bf16": ([c_void_p, c_void_p, c_i64, c_float, c_u64, c_u64, c_void_p], c_int),
    "vmvm_transpose_batched_bf16": ([c_void_p, c_void_p, c_void_p, c_int, c_void_p], c_int),
    "vmvm_sumsq_f32": ([c_void_p, c_i64, c_void_p, c_void_p, c_u64, c_void_p], c_int),
    "vmvm_adamw": ([C.POINTER(AdamWDesc), c_void_p], c_int),
    "vmvm_swin_block_fwd": ([C.POINTER(SwinBlock), c_void_p], c_int),
    "vmvm_swin_block_bwd": ([C.POINTER(SwinBlock), c_void_p, c_void_p, c_void_p], c_int),
    "vmvm_bert_layer_fwd": ([C.POINTER(BertLayer), c_void_p], c_int),
    "vmvm_bert_layer_bwd": ([C.POINTER(BertLayer), c_void_p, c_void_p, c_void_p], c_int),
    "vmvm_probe_tr16": ([c_void_p, c_void_p], c_int),
    # workspace-size queries: the caller owns every buffer, scratch included
    "vmvm_gemm_workspace_size": ([C.POINTER(GemmDesc)], c_i64),
    "vmvm_layernorm_bwd_workspace_size": ([C.POINTER(LnBwdDesc)], c_i64),
    "vmvm_attention_bwd_workspace_size": ([C.POINTER(AttnBwdDesc)], c_i64),
    "vmvm_attention_drop_mask_size": ([C.POINTER(AttnFwdDesc)], c_i64),
    "vmvm_attention_bwd_dbias_ws_size": ([C.POINTER(AttnBwdDesc)], c_i64),
    "vmvm_attention_bwd_table_is_separate": ([C.POINTER(AttnBwdDesc)], c_int),
    "vmvm_sumsq_workspace_size": ([c_i64], c_i64),
}

_lib = None


def load():
    """Load libvmvm.so; raises RuntimeError (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built -- run `python -m pytorch_empirical_mvm_amd.build` "
                               "(there is NO CPU / eager fallback for the VIOLETv2 step)")
        lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _PROTOS.items():
            fn = getattr(lib, name)        # AttributeError if the .so does not export a declared symbol
            fn.argtypes, fn.restype = args, res
        _lib = lib
    return _lib


def exported_symbols():
    return sorted(_PROTOS)


_PINNED = None        # raw handle of the stream the step is being issued on (None: ask torch)


def stream():
    """hipStream_t every wrapper in kernels.py launches on.  `torch.cuda.current_stream()` costs ~8 us per call (device-index and
    lazy-init checks, an environment look-up): at ~1 400 launches per step that was 8-10 ms of host time (tools/scratch/host_profile.py),
    so the step bodies PIN the handle (`pin_current` / `on_stream`) and this returns it; outside a pinned region torch is asked."""
    return _PINNED if _PINNED is not None else torch.cuda.current_stream().cuda_stream


class _Pin:
    __slots__ = ("handle", "saved", "ctx")

    def __init__(self, handle, ctx=None):
        self.handle, self.ctx = handle, ctx

    def __enter__(self):
        global _PINNED
        if self.ctx is not None:
            self.ctx.__enter__()
        self.saved, _PINNED = _PINNED, self.handle
        return self

    def __exit__(self, *exc):
        global _PINNED
        _PINNED = self.saved
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def pin_current():
    """pin torch's current stream for the body (the step functions wrap themselves in this)"""
    return _Pin(torch.cuda.current_stream().cuda_stream)


def on_stream(s):
    """`with torch.cuda.stream(s)` + the pin: EVERY stream switch inside the package goes through here, so a pinned region never
    launches on a stream torch does not consider current"""
    return _Pin(s.cuda_stream, torch.cuda.stream(s))


def ptr(t):
    return 0 if t is None else t.data_ptr()


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"libvmvm {what} failed: rc={rc} (hip error {load().vmvm_last_hip_error()})")
