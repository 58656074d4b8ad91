#!/usr/bin/env python3
"""Host-side argument validation of EVERY libvmvm entry point, without a GPU: each call below must be refused by the dispatch layer
(VMVM_EINVAL / VMVM_ENOSUPPORT) or answered by pure host arithmetic (the workspace-size queries) BEFORE any HIP call is made.

Used twice (SURVEY 5 "sanitizers", VERDICT r5 missing #4):
* `tests/test_cabi_cpu.py` runs it against the production library;
* `tests/test_sanitizers_cpu.py` runs it in a child process against the ASAN + UBSAN host build (`tools/build_asan.py`,
  `VMVM_LIB=build/libvmvm_asan.so`, the sanitizer runtime pre-loaded): a descriptor field read out of bounds, a signed overflow in a
  grid / workspace computation or a misaligned access in the dispatch code aborts the child.  Never on the GPU box.

    python tools/cabi_validation.py            # prints one line per call, exit code 0 when every call answered as expected
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EINVAL, ENOSUPPORT = -1, -2


def main(verbose=True):
    from pytorch_empirical_mvm_amd import lib
    l = lib.load()
    buf = (C.c_uint8 * 4096)()                      # a host buffer: pointers below only have to be non-null and 16-byte aligned, nothing is launched
    p = C.addressof(buf)
    p = (p + 15) & ~15
    bad = []

    def expect(name, rc, want):
        ok = rc in want if isinstance(want, (tuple, list, set)) else rc == want
        if verbose:
            print(f"{'ok  ' if ok else 'FAIL'} {name:58s} -> {rc}")
        if not ok:
            bad.append((name, rc, want))

    assert l.vmvm_version() >= 1
    # ---- GEMM
    d = lib.GemmDesc()
    expect("gemm: null descriptor", l.vmvm_gemm_bf16(None, None), EINVAL)
    expect("gemm: null operands", l.vmvm_gemm_bf16(C.byref(d), None), EINVAL)
    d.A = d.B = d.C = p
    expect("gemm: zero sizes", l.vmvm_gemm_bf16(C.byref(d), None), EINVAL)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.a_kmajor, d.b_kmajor = 128, 128, 64, 64, 64, 128, 1, 1
    d.M = -5
    expect("gemm: negative M", l.vmvm_gemm_bf16(C.byref(d), None), EINVAL)
    d.M, d.lda = 128, 63
    expect("gemm: leading dimension not a multiple of 8", l.vmvm_gemm_bf16(C.byref(d), None), (EINVAL, ENOSUPPORT))
    d.lda, d.act = 64, 99
    expect("gemm: unknown activation", l.vmvm_gemm_bf16(C.byref(d), None), (EINVAL, ENOSUPPORT))
    d.act, d.row_map, d.map_len = 0, p, 0
    expect("gemm: row_map without map_len", l.vmvm_gemm_bf16(C.byref(d), None), (EINVAL, ENOSUPPORT))
    expect("gemm workspace: null", l.vmvm_gemm_workspace_size(None), EINVAL)
    w = lib.GemmDesc()
    for (M, N, K_) in [(3072, 768, 69120), (768, 3072, 69120), (128, 128, 802816), (30522, 768, 1024), (1, 1, 1)]:
        w.M, w.N, w.K, w.out_fp32, w.accumulate = M, N, K_, 1, 1
        n = l.vmvm_gemm_workspace_size(C.byref(w))
        expect(f"gemm workspace: {M} x {N} x {K_}", 0 if n >= 0 else n, 0)
    # (round 6: this query DIVIDED BY ZERO -- the 32-bit tile count of 2^30 x 2^30 wrapped to 0 -- until the sweep found it)
    w.M, w.N, w.K = 2 ** 30, 2 ** 30, 2 ** 30
    expect("gemm workspace: 2^30 x 2^30 x 2^30 (tile count beyond 32 bits)", l.vmvm_gemm_workspace_size(C.byref(w)), ENOSUPPORT)
    d2 = lib.GemmDesc()
    d2.A = d2.B = d2.C = p
    d2.M, d2.N, d2.K, d2.lda, d2.ldb, d2.ldc, d2.a_kmajor, d2.b_kmajor = 2 ** 30, 2 ** 30, 64, 64, 64, 2 ** 30, 1, 1
    expect("gemm: 2^30 x 2^30 output", l.vmvm_gemm_bf16(C.byref(d2), None), ENOSUPPORT)
    # ---- LayerNorm
    f = lib.LnFwdDesc()
    expect("ln fwd: null", l.vmvm_layernorm_fwd(None, None), EINVAL)
    expect("ln fwd: empty", l.vmvm_layernorm_fwd(C.byref(f), None), EINVAL)
    f.X = f.Y = f.gamma = f.beta = f.mean = f.rstd = p
    f.M, f.C, f.nseg, f.ldx, f.ldy = 16, 100, 1, 104, 104
    expect("ln fwd: C not a multiple of 8", l.vmvm_layernorm_fwd(C.byref(f), None), EINVAL)
    f.C, f.ldx, f.ldy = 4096, 4096, 4096
    expect("ln fwd: C > 3072", l.vmvm_layernorm_fwd(C.byref(f), None), ENOSUPPORT)
    f.C, f.ldx, f.ldy, f.src, f.rows_out_per_batch = 768, 768, 768, p, 0
    expect("ln fwd: map without batch sizes", l.vmvm_layernorm_fwd(C.byref(f), None), EINVAL)
    b = lib.LnBwdDesc()
    expect("ln bwd: null", l.vmvm_layernorm_bwd(None, None), EINVAL)
    expect("ln bwd: empty", l.vmvm_layernorm_bwd(C.byref(b), None), EINVAL)
    b.dY = b.X = b.gamma = b.mean = b.rstd = b.dX = b.dgamma = b.dbeta = p
    b.M, b.C, b.nseg, b.lddy, b.ldx, b.lddx = 1568, 128, 1, 128, 128, 128
    b.dx_map, b.dx_map_len, b.dX2 = p, 1568, p
    expect("ln bwd: dx_map together with dX2", l.vmvm_layernorm_bwd(C.byref(b), None), ENOSUPPORT)
    b.dX2, b.dX_add = None, p
    expect("ln bwd: dx_map with dX aliasing dX_add", l.vmvm_layernorm_bwd(C.byref(b), None), EINVAL)
    b.dx_map, b.dx_map_len, b.src, b.rows_out_per_batch, b.rows_in_per_batch = None, 0, p, 1568, 1568
    b.dX_add, b.add_by_out, b.inv, b.rows_in_total = p + 2048, 1, p, 1568
    expect("ln bwd: add_by_out together with inv", l.vmvm_layernorm_bwd(C.byref(b), None), ENOSUPPORT)
    b.add_by_out, b.inv, b.src, b.dx_map, b.dx_map_len = 0, None, None, p, 1000
    expect("ln bwd: M not a multiple of dx_map_len", l.vmvm_layernorm_bwd(C.byref(b), None), EINVAL)
    expect("ln bwd workspace: null", l.vmvm_layernorm_bwd_workspace_size(None), EINVAL)
    # ---- attention
    a = lib.AttnFwdDesc()
    expect("attention fwd: null", l.vmvm_attention_fwd(None, None), EINVAL)
    expect("attention fwd: empty", l.vmvm_attention_fwd(C.byref(a), None), EINVAL)
    a.qkv = a.out = a.lse = p
    a.nseq, a.L, a.heads, a.head_dim, a.mode, a.ld_qkv, a.ld_out = 4, 392, 4, 48, 0, 384, 128
    expect("attention fwd: head_dim 48", l.vmvm_attention_fwd(C.byref(a), None), (EINVAL, ENOSUPPORT))
    a.head_dim, a.mode = 32, 7
    expect("attention fwd: unknown mode", l.vmvm_attention_fwd(C.byref(a), None), (EINVAL, ENOSUPPORT))
    ab = lib.AttnBwdDesc()
    expect("attention bwd: null", l.vmvm_attention_bwd(None, None), EINVAL)
    expect("attention bwd: empty", l.vmvm_attention_bwd(C.byref(ab), None), EINVAL)
    expect("attention drop-mask size: empty", 0 if l.vmvm_attention_drop_mask_size(C.byref(a)) <= 0 else 1, 0)
    ab.f.nseq, ab.f.heads, ab.f.L = 160, 12, 432
    expect("attention bwd workspace", 0 if l.vmvm_attention_bwd_workspace_size(C.byref(ab)) == 160 * 12 * 432 * 4 else 1, 0)
    # ---- optimizer / reductions
    ad = lib.AdamWDesc()
    expect("adamw: null", l.vmvm_adamw(None, None), EINVAL)
    expect("adamw: empty", l.vmvm_adamw(C.byref(ad), None), EINVAL)
    expect("sumsq: null", l.vmvm_sumsq_f32(None, 10, None, None, 0, None), EINVAL)
    expect("sumsq workspace", 0 if l.vmvm_sumsq_workspace_size(225_000_000) == 2048 * 4 else 1, 0)
    # ---- everything that takes raw pointers: null pointers / non-positive sizes are refused
    z = None
    expect("colsum: null", l.vmvm_colsum_bf16(z, 8, 8, 8, z, 0, z, 0, z), EINVAL)
    expect("patch_embed: null", l.vmvm_patch_embed_fwd(z, z, z, z, z, z, 1e-5, z, z, z, z, 1, 1, 32, 32, 128, z), EINVAL)
    expect("patch_im2col: null", l.vmvm_patch_im2col(z, z, z, 1, 1, 32, 32, z), EINVAL)
    expect("masking: null", l.vmvm_masking(z, z, z, z, z, z, z, z, 0, 0, 1, 32, 4, 3, 3, 0.15, 101, 102, 0, 103, z), EINVAL)
    expect("encvideo_assemble: null", l.vmvm_encvideo_assemble(z, z, z, z, z, 1, 1, 9, 768, z), EINVAL)
    expect("encvideo_assemble_bwd: null", l.vmvm_encvideo_assemble_bwd(z, z, z, z, z, 1, 1, 9, 768, z), EINVAL)
    expect("bert_embed: null", l.vmvm_bert_embed(z, z, z, z, z, 1, 32, 768, z), EINVAL)
    expect("bert_embed_bwd: null", l.vmvm_bert_embed_bwd(z, z, z, z, z, 1, 32, 768, z), EINVAL)
    expect("count_valid: null", l.vmvm_count_valid(z, 10, z, z), EINVAL)
    expect("cross_entropy: null", l.vmvm_cross_entropy(z, 8, 8, 8, z, z, z, z, 8, z), EINVAL)
    expect("vtm_ce: null", l.vmvm_vtm_ce(z, 2, 2, z, z, z), EINVAL)
    expect("pixel_l1: null", l.vmvm_pixel_l1(z, z, z, z, z, z, 1, 1, 3, 3, 32, 3, 1.0 / 3, z), EINVAL)
    expect("feature_l1: null", l.vmvm_feature_l1(z, z, z, z, 1.0, z, z, 8, 8, z), EINVAL)
    expect("rowdot: null", l.vmvm_rowdot(z, 8, 8, z, z, 1.0, z, z), EINVAL)
    expect("rowdot_bwd: null", l.vmvm_rowdot_bwd(z, 8, 8, z, z, 1.0, z, z, z, 0, z), EINVAL)
    expect("cast_bf16_to_fp8: null", l.vmvm_cast_bf16_to_fp8(z, z, 8, 1.0, z), EINVAL)
    expect("dvae_stem_im2col: null", l.vmvm_dvae_stem_im2col(z, z, 1, 8, 8, z), EINVAL)
    expect("maxpool2x2: null", l.vmvm_maxpool2x2_nhwc_f16(z, z, 1, 8, 8, 64, z), EINVAL)
    expect("argmax_pairs: null", l.vmvm_argmax_pairs(z, 8, 8, 1, z, z), EINVAL)
    expect("cast_f32_to_bf16: null", l.vmvm_cast_f32_to_bf16(z, z, 8, z), EINVAL)
    expect("cast_bf16_to_f32: null", l.vmvm_cast_bf16_to_f32(z, z, 8, z), EINVAL)
    expect("gather_rows: null", l.vmvm_gather_rows_bf16(z, 8, z, z, 8, 8, 8, 0, 0, z), EINVAL)
    expect("expand_batch_map: null", l.vmvm_expand_batch_map(z, 8, z, 1, 8, z, z), EINVAL)
    expect("invert_map: null", l.vmvm_invert_map(z, 8, z, 8, z), EINVAL)
    expect("copy_batches: null", l.vmvm_copy_batches_bf16(z, 8, z, 8, z, 1, 8, 8, z), EINVAL)
    expect("attn_query_row_fwd: null", l.vmvm_attn_query_row_fwd(z, 8, z, 8, 0, 8, z, z, 8, z, z, 1, 8, 1, 64, 1.0, 0.0, 0, 0, z), EINVAL)
    expect("attn_query_row_bwd: null", l.vmvm_attn_query_row_bwd(z, 8, z, 8, z, 8, 0, 8, z, z, z, 8, z, 8, 1, 8, 1, 64, 1.0, z), EINVAL)
    expect("add_bf16: null", l.vmvm_add_bf16(z, z, z, 8, z), EINVAL)
    expect("pool_grad: null", l.vmvm_pool_grad_bf16(z, z, z, z, 1, 1, 8, 8, 768, z, z, z), EINVAL)
    expect("scatter_add_rows: null", l.vmvm_scatter_add_rows_bf16(z, 8, z, z, 8, 8, 8, z), EINVAL)
    expect("gelu_bwd: null", l.vmvm_gelu_bwd_bf16(z, z, z, 8, z), EINVAL)
    expect("dropout: null", l.vmvm_dropout_bf16(z, z, 8, 0.1, 0, 0, z), EINVAL)
    expect("transpose_batched: null", l.vmvm_transpose_batched_bf16(z, z, z, 1, z), EINVAL)
    # ---- block-level calls (csrc/blocks.hip): refused before the first per-kernel call is made
    bl = lib.BertLayer()
    for nm in ("vmvm_bert_layer_fwd", "vmvm_bert_layer_bwd"):
        fn = getattr(l, nm)
        args = (None,) if nm.endswith("fwd") else (None, None, None)
        expect(f"{nm}: null descriptor", fn(None, *args), EINVAL)
        expect(f"{nm}: zeroed descriptor", fn(C.byref(bl), *args), EINVAL)
    bl.nseq, bl.L, bl.hidden, bl.heads, bl.ffn = 2, 64, 768, 12, 3072
    expect("bert_layer_fwd: sizes but no buffers", l.vmvm_bert_layer_fwd(C.byref(bl), None), EINVAL)
    expect("bert_layer_bwd: sizes but no buffers", l.vmvm_bert_layer_bwd(C.byref(bl), None, None, None), EINVAL)
    bl.hidden = 770
    expect("bert_layer_fwd: hidden not a multiple of the head count", l.vmvm_bert_layer_fwd(C.byref(bl), None), EINVAL)
    sb = lib.SwinBlock()
    for nm in ("vmvm_swin_block_fwd", "vmvm_swin_block_bwd"):
        fn = getattr(l, nm)
        args = (None,) if nm.endswith("fwd") else (None, None, None)
        expect(f"{nm}: null descriptor", fn(None, *args), EINVAL)
        expect(f"{nm}: zeroed descriptor", fn(C.byref(sb), *args), EINVAL)
    sb.B, sb.L, sb.Lp, sb.N, sb.nW, sb.C, sb.heads = 2, 6272, 6272, 392, 16, 128, 4
    sb.has_attn = sb.has_mlp = 1
    expect("swin_block_fwd: sizes but no buffers", l.vmvm_swin_block_fwd(C.byref(sb), None), EINVAL)
    for f_ in ("x", "x1", "x2", "src", "rc", "xw", "mean1", "rstd1", "qkv", "ao", "lse", "Wqkv", "Wproj", "bqkv", "bproj", "n1_g", "n1_b", "table",
               "y2", "mean2", "rstd2", "h", "W1", "W2", "b1", "b2", "n2_g", "n2_b"):
        setattr(sb, f_, p)
    sb.Lp = 6000
    expect("swin_block_fwd: Lp != nW * N", l.vmvm_swin_block_fwd(C.byref(sb), None), EINVAL)
    sb.Lp, sb.compact_a = 6272, 1
    expect("swin_block_fwd: compacted branch without its clip lists", l.vmvm_swin_block_fwd(C.byref(sb), None), EINVAL)
    sb.compact_a = 0
    expect("swin_block_bwd: no incoming gradient", l.vmvm_swin_block_bwd(C.byref(sb), None, None, None), EINVAL)
    sb.d_out = sb.d_x = p
    expect("swin_block_bwd: side stream without a fork event", l.vmvm_swin_block_bwd(C.byref(sb), None, p, None), EINVAL)
    expect("swin_block_bwd: no backward buffers", l.vmvm_swin_block_bwd(C.byref(sb), None, None, None), EINVAL)
    sb.B = 1 << 20
    expect("swin_block_fwd: row count beyond 32-bit indexing", l.vmvm_swin_block_fwd(C.byref(sb), None), EINVAL)
    expect("colsum_ws: null", l.vmvm_colsum_bf16_ws(z, 8, 8, 8, z, 0, z, 0, z, 0, z), EINVAL)
    expect("colsum workspace: 69120 x 3072", 0 if l.vmvm_colsum_workspace_size(69120, 3072) > 0 else -1, 0)
    expect("colsum workspace: negative", 0 if l.vmvm_colsum_workspace_size(-1, 8) <= 0 else -1, 0)
    expect("attention_bwd_dbias_ws_size: null", 0 if l.vmvm_attention_bwd_dbias_ws_size(None) <= 0 else -1, 0)
    ab = lib.AttnBwdDesc()
    ab.f.nseq, ab.f.heads, ab.f.L, ab.f.head_dim, ab.f.win_layout = 512, 4, 392, 32, 1
    expect("attention_bwd_dbias_ws_size: a window problem", 0 if l.vmvm_attention_bwd_dbias_ws_size(C.byref(ab)) >= 0 else -1, 0)
    ab.dbias_table, ab.f.mode, ab.f.L = p, 0, 1152
    expect("attention_bwd_table_is_separate: streaming windows", l.vmvm_attention_bwd_table_is_separate(C.byref(ab)), 1)
    ab.f.L = 392
    expect("attention_bwd_table_is_separate: (8,7,7) windows", l.vmvm_attention_bwd_table_is_separate(C.byref(ab)), 0)
    expect("attention_bwd_table_is_separate: null", l.vmvm_attention_bwd_table_is_separate(None), 0)
    ab.table_phase = 7
    expect("attention_bwd: unknown table_phase", l.vmvm_attention_bwd(C.byref(ab), None), EINVAL)
    expect("probe_tr16: null", l.vmvm_probe_tr16(z, z), EINVAL)
    expect("last_hip_error: no HIP call was made", l.vmvm_last_hip_error(), 0)
    # ---- the sweep names every exported entry point (a new one without a case fails here)
    src = open(os.path.abspath(__file__)).read()
    for nm in lib.exported_symbols():
        expect(f"sweep covers {nm}", 0 if (nm in src or nm.replace("vmvm_", "") in src) else -99, 0)
    return bad


if __name__ == "__main__":
    bad = main()
    print(f"{'FAILED' if bad else 'all refused / answered on the host'}: {len(bad)} unexpected results {bad[:5]}")
    sys.exit(1 if bad else 0)
