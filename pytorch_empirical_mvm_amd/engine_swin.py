"""Video-Swin half of `VioletEngine` (engine.py): patch embedding, Swin blocks, patch merging, the backbone forward.
Reference: SwinTransformer3D / BasicLayer / SwinTransformerBlock3D (video_swin.py:176-482).  Methods of the engine class; state lives there."""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import config as CFG
from . import kernels as K
from . import swin_index as SI
from .store import BF16, F32, V, DropScale, _acc, _gout, _h2d, _dev_i32


class SwinMixin:
    # -------------------------------------------------------------- Video-Swin
    def _patch_embed(self, img, cov):
        """PatchEmbed3D (video_swin.py:390-407) in one kernel (`vmvm_patch_embed_fwd`: clip read once, cover + zero frame applied on
        the way into the MFMA operands, LayerNorm as the epilogue).  The layer is 0.1% of the FLOPs but sets the precision of
        everything downstream, so the pixels enter as a bf16 hi/lo pair and the conv output stays f32 into the LayerNorm (the
        reference runs this conv in fp16 = 3 more mantissa bits than bf16).  No im2col buffer is kept: the weight gradient
        re-derives its [M,192] = [hi | lo] operand in the backward, where it lives for one GEMM."""
        S, pre = self.store, "enc_img.swin.patch_embed."
        E = self.cfg["embed_dim"]
        wb = S.b(pre + "proj.weight", (E, 96))
        x, z, mean, rstd = K.patch_embed_fwd(img, cov, wb, S.p(pre + "proj.bias"), S.p(pre + "norm.weight"), S.p(pre + "norm.bias"), 1e-5)
        out = V(x)

        def bwd():
            dz, _ = K.layernorm_bwd(out.g, z, S.p(pre + "norm.weight"), mean, rstd, S.g(pre + "norm.weight"), S.g(pre + "norm.bias"))
            K.colsum(dz, S.g(pre + "proj.bias"), accumulate=True)
            cols = K.patch_im2col(img, cov)                                   # [M,192] = [hi | lo]
            K.gemm(dz, cols, a_kmajor=False, b_kmajor=False, M=E, N=96, K=dz.shape[0], out=S.g(pre + "proj.weight", (E, 96)), accumulate=True)
        self.tape.append(bwd)
        return out

    def _window_attention_bwd(self, dao, qkv, ao, lse, nseq, N, nh, hd, scale, gtable, akw):
        """window-attention backward; a table gradient that is a launch of its own (streaming windows: vmvm_attn_bwd_desc.table_phase)
        goes to the second stream, beside the GEMMs that follow -- nothing on the input-gradient chain reads it"""
        if self.wstream is not None and self.sw.table_side and self._cached(("tabsep", N), lambda: K.attention_table_separate(N, 0)):
            dqkv, delta = K.attention_bwd(dao, qkv, ao, lse, nseq, N, nh, hd, 0, scale, dbias_table=gtable, table_phase=1, **akw)
            self._wgrad_launch(lambda ws: K.attention_bwd(dao, qkv, ao, lse, nseq, N, nh, hd, 0, scale, dbias_table=gtable, table_phase=2,
                                                          dqkv=dqkv, delta=delta, **akw), (dao, qkv, ao, lse, dqkv, delta, akw.get("seq_scale")))
            return dqkv
        return K.attention_bwd(dao, qkv, ao, lse, nseq, N, nh, hd, 0, scale, dbias_table=gtable, **akw)

    def _swin_block(self, xv, B, dims, C, nh, pre, shifted, dp):
        if self.sw.block_abi and self.device.type == "cuda" and not self.store.frozen:
            return self._swin_block_block(xv, B, dims, C, nh, pre, shifted, dp)
        return self._swin_block_calls(xv, B, dims, C, nh, pre, shifted, dp)

    def _swin_block_block(self, xv, B, dims, C, nh, pre, shifted, dp):
        """SwinTransformerBlock3D through the block-level C ABI (include/vmvm.h vmvm_swin_block; csrc/blocks.hip issues the launches of
        `_swin_block_calls` below through the same per-kernel descriptors -- that function stays as the statement of the block and as the
        other side of the bit-for-bit test).  Here: the schedule (which clips each branch runs on), the index tables, the buffers."""
        from . import lib as L
        import ctypes as Ct
        S, cfg = self.store, self.cfg
        D, H, W = dims
        L_ = D * H * W
        win = tuple(cfg["window"])
        ws, ss = SI.get_window_size(dims, win, tuple(i // 2 for i in win) if shifted else (0, 0, 0))
        wm, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
        N = ws[0] * ws[1] * ws[2]
        Lp = wm.size
        nW = Lp // N
        dev = self.device
        reg_np = SI.region_ids(Dp, Hp, Wp, ws, ss)
        rc_np, rc0 = SI.rc_codes(N, win)
        w3 = 1 if (SI.win3_ok(ws, ss) and self.sw.win_layout) else 0
        pm = SI.win3_perm() if w3 else None
        src = self._cached(("wm", dims, ws, ss, w3), lambda: _dev_i32(wm.reshape(nW, N)[:, pm].reshape(-1) if w3 else wm, dev))

        def _inv_host():
            sm = wm.reshape(nW, N)[:, pm].reshape(-1) if w3 else wm
            inv = np.full(L_, -1, dtype=np.int32)
            ok = np.flatnonzero(sm >= 0)
            inv[sm[ok]] = ok
            return _dev_i32(inv, dev)
        src_major = C <= 256 and self.sw.ln_src_major
        reg = None if reg_np is None else self._cached(("reg", Dp, Hp, Wp, ws, ss, w3), lambda: torch.from_numpy(np.ascontiguousarray(reg_np[:, pm]) if w3 else reg_np).to(dev))
        rc = self._cached(("rc", N, win, w3), lambda: _dev_i32(rc_np[pm] if w3 else rc_np, dev))
        scale = 32 ** -0.5 if C // nh == 32 else (C // nh) ** -0.5
        dp, dp2 = (dp if isinstance(dp, (tuple, list)) else (dp, dp))
        ds = dp if (isinstance(dp, DropScale) or dp is None) else DropScale(dp)
        ds2 = dp2 if (isinstance(dp2, DropScale) or dp2 is None) else DropScale(dp2)
        # ---- the schedule, exactly as `_swin_block_calls` decides it
        Bk, compact = B, False
        if ds is not None and self.sw.droppath_dce != "0":
            Bk, kept_a, dpk_a = ds.take(math.gcd(L_, Lp), B)
            compact = Bk < B and ds.scale is not None
            if not compact:
                Bk = B
        has_attn = not (compact and Bk == 0)
        Bm, compact2 = B, False
        if ds2 is not None and self.sw.droppath_dce not in ("0", "attn"):
            Bm, kept_m, dpk_m = ds2.take(L_, B)
            compact2 = Bm < B and ds2.scale is not None
            if not compact2:
                Bm = B
        has_mlp = not (compact2 and Bm == 0)
        c8 = getattr(self, "gelu_code8", True) and C % 64 == 0
        win_dx1 = self.sw.dx1_window and not compact and not compact2 and Lp == L_

        def cs(d_, comp):                              # bias-gradient form of a branch (engine._linear_bwd): (mode, scale)
            if d_ is None:
                return 0, 0.0                          # no DropPath scales at all: plain fused column sum
            if comp or d_.n_kept == B:
                return (1, float(d_.scale)) if d_.scale is not None else (2, 0.0)
            return 2, 0.0
        b = L.SwinBlock()
        b.B, b.L, b.Lp, b.N, b.nW, b.C, b.heads, b.qscale = B, L_, Lp, N, nW, C, nh, scale
        b.win_layout, b.rc0, b.code8 = w3, rc0, int(c8)
        table = S.p(pre + "attn.relative_position_bias_table")
        b.table_len = table.shape[0]
        b.has_attn, b.compact_a, b.Bk = int(has_attn), int(compact), (Bk if compact else B)
        b.has_mlp, b.compact_m, b.Bm = int(has_mlp), int(compact2), (Bm if compact2 else B)
        b.cs_mode_a, b.cs_scale_a = cs(ds, compact)
        b.cs_mode_m, b.cs_scale_m = cs(ds2, compact2)
        sa = (dpk_a if compact else (None if ds is None else ds.dev))
        sm_ = (dpk_m if compact2 else (None if ds2 is None else ds2.dev))
        b.scale_a, b.scale_m = L.ptr(sa), L.ptr(sm_)
        if compact:
            b.kept_a, b.drop_a, b.nd_a = kept_a.data_ptr(), ds.dropped.data_ptr(), B - ds.n_kept
        if compact2:
            b.kept_m, b.drop_m, b.nd_m = kept_m.data_ptr(), ds2.dropped.data_ptr(), B - ds2.n_kept
        b.dx1_window, b.src_major = int(win_dx1), int(src_major)
        inv = self._cached(("wminv", dims, ws, ss, w3), _inv_host) if (win_dx1 or (src_major and not compact)) else None
        idm = self._cached(("idmap", L_), lambda: _dev_i32(np.arange(L_), dev)) if compact2 else None
        b.src, b.inv, b.idm, b.rc, b.region = src.data_ptr(), L.ptr(inv), L.ptr(idm), rc.data_ptr(), L.ptr(reg)
        names = dict(Wqkv="attn.qkv.weight", Wproj="attn.proj.weight", W1="mlp.fc1.weight", W2="mlp.fc2.weight")
        for k_, n_ in names.items():
            w_, wt = S.b(pre + n_), S.bt(pre + n_)
            setattr(b, k_, w_.data_ptr())
            setattr(b, k_ + "T", L.ptr(wt if (wt is not None and wt.shape[1] == w_.shape[0]) else None))      # (engine._linear_bwd's choice)
            setattr(b, "g" + k_, S.g(pre + n_).data_ptr())
        for k_, n_ in dict(bqkv="attn.qkv.bias", bproj="attn.proj.bias", b1="mlp.fc1.bias", b2="mlp.fc2.bias", n1_g="norm1.weight", n1_b="norm1.bias",
                           n2_g="norm2.weight", n2_b="norm2.bias", table="attn.relative_position_bias_table").items():
            setattr(b, k_, S.p(pre + n_).data_ptr())
            setattr(b, "g" + k_, S.g(pre + n_).data_ptr())
        e = torch.empty
        x = xv.t
        Ma, Mm = (Bk if compact else B) * Lp, (Bm if compact2 else B) * L_
        keep = [x, src, rc, reg, inv, idm, sa, sm_, ds, ds2]
        b.x = x.data_ptr()
        if has_attn:
            xw, mean1, rstd1 = e((Ma, C), device=dev, dtype=BF16), e(Ma, device=dev, dtype=F32), e(Ma, device=dev, dtype=F32)
            qkv, ao, lse = e((Ma, 3 * C), device=dev, dtype=BF16), e((Ma, C), device=dev, dtype=BF16), e((Ma // N, nh, N), device=dev, dtype=F32)
            x1 = e((B * L_, C), device=dev, dtype=BF16)
            src_k = e((Ma,), device=dev, dtype=torch.int32) if compact else None
            b.xw, b.mean1, b.rstd1, b.qkv, b.ao, b.lse, b.src_k = (L.ptr(t_) for t_ in (xw, mean1, rstd1, qkv, ao, lse, src_k))
            keep += [xw, mean1, rstd1, qkv, ao, lse, src_k]
        else:
            x1 = x
        b.x1 = x1.data_ptr()
        if has_mlp:
            y2, mean2, rstd2 = e((Mm, C), device=dev, dtype=BF16), e(Mm, device=dev, dtype=F32), e(Mm, device=dev, dtype=F32)
            u, h = e((Mm, 4 * C), device=dev, dtype=torch.uint8 if c8 else BF16), e((Mm, 4 * C), device=dev, dtype=BF16)
            x2 = e((B * L_, C), device=dev, dtype=BF16)
            map_m = e((Mm,), device=dev, dtype=torch.int32) if compact2 else None
            b.y2, b.mean2, b.rstd2, b.u, b.h, b.map_m = (L.ptr(t_) for t_ in (y2, mean2, rstd2, u, h, map_m))
            keep += [x1, y2, mean2, rstd2, u, h, map_m]
        else:
            x2 = x1
        b.x2 = x2.data_ptr()
        b.reserve_cus = K.reserve_cus()
        lib = L.load()
        L.check(lib.vmvm_swin_block_fwd(Ct.byref(b), L.stream()), "swin_block_fwd")
        out = V(x2)

        def bwd():
            assert keep is not None                    # (the descriptor holds raw addresses: the forward buffers live until this has run)
            dx2 = out.g
            held = []
            b.d_out = dx2.data_ptr()
            dx1_t = dx2
            if has_mlp:
                dx2c = e((Mm, C), device=dev, dtype=BF16) if compact2 else None
                du, dy2 = e((Mm, 4 * C), device=dev, dtype=BF16), e((Mm, C), device=dev, dtype=BF16)
                b.dx2c, b.du, b.dy2 = L.ptr(dx2c), du.data_ptr(), dy2.data_ptr()
                if not win_dx1:
                    dx1_t = e((B * L_, C), device=dev, dtype=BF16)
                    b.dx1 = dx1_t.data_ptr()
                held += [dx2c if compact2 else dx2, h, du, y2]
            if has_attn:
                dx1w = e((Ma, C), device=dev, dtype=BF16)
                dao, dqkv, dxw = e((Ma, C), device=dev, dtype=BF16), e((Ma, 3 * C), device=dev, dtype=BF16), e((Ma, C), device=dev, dtype=BF16)
                delta = e((Ma // N, nh, N), device=dev, dtype=F32)
                dx = e((B * L_, C), device=dev, dtype=BF16)
                inv_k = e((B * L_,), device=dev, dtype=torch.int32) if (compact and src_major) else None
                b.dx1w, b.dao, b.dqkv, b.dxw, b.delta, b.d_x, b.inv_k = (L.ptr(t_) for t_ in (dx1w, dao, dqkv, dxw, delta, dx, inv_k))
                held += [dx1w, ao, dqkv, xw]
            else:
                dx = dx1_t                             # every clip of the attention branch dropped: d(x) = d(x1)
                b.d_x = dx.data_ptr()
            wsm = K._WORKSPACE.get(x.device)
            b.ws_main, b.ws_main_bytes = L.ptr(wsm), (wsm.numel() if wsm is not None else 0)
            side = self.wstream
            b.ws_side, b.ws_side_bytes = (self.workspace_w.data_ptr(), self.workspace_w.numel()) if side is not None else (0, 0)
            b.reserve_cus = K.reserve_cus()
            b.table_side = int(side is not None and self.sw.table_side)
            if has_attn and b.table_side and self._cached(("tabsep", N), lambda: K.attention_table_separate(N, 0)):
                held += [dao, qkv, lse, delta]         # (its table-gradient launch reads these on the side stream)
            L.check(lib.vmvm_swin_block_bwd(Ct.byref(b), L.stream(), side.cuda_stream if side is not None else None,
                                            self._fork_event() if side is not None else None), "swin_block_bwd")
            self._whold(tuple(held) + (sa, sm_))
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def _swin_block_calls(self, xv, B, dims, C, nh, pre, shifted, dp):
        S, cfg = self.store, self.cfg
        D, H, W = dims
        L = D * H * W
        win = tuple(cfg["window"])
        ws, ss = SI.get_window_size(dims, win, tuple(i // 2 for i in win) if shifted else (0, 0, 0))
        wm, (Dp, Hp, Wp) = SI.window_map(D, H, W, ws, ss)
        N = ws[0] * ws[1] * ws[2]
        Lp = wm.size
        nW = Lp // N
        dev = self.device
        reg_np = SI.region_ids(Dp, Hp, Wp, ws, ss)
        rc_np, rc0 = SI.rc_codes(N, win)
        # win_layout = 1 (include/vmvm.h, swin_index.win3_perm): the order of the tokens INSIDE a window is free -- one gather map serves
        # the LayerNorm, the projection's un-gather epilogue and the backward -- so (8,7,7) windows are laid out d-fastest and region-major,
        # which is what the win3 attention kernels assume (Toeplitz bias reads, masked score tiles skipped); rc / region follow the slots
        w3 = 1 if (SI.win3_ok(ws, ss) and self.sw.win_layout) else 0
        pm = SI.win3_perm() if w3 else None               # (applied inside the cached builders: host work once per shape, not per block call)
        src = self._cached(("wm", dims, ws, ss, w3), lambda: _dev_i32(wm.reshape(nW, N)[:, pm].reshape(-1) if w3 else wm, dev))

        def _inv_host():                                       # natural row -> its window slot (the LayerNorm backward's source-major walk, C <= 256)
            sm = wm.reshape(nW, N)[:, pm].reshape(-1) if w3 else wm
            inv = np.full(L, -1, dtype=np.int32)
            ok = np.flatnonzero(sm >= 0)
            inv[sm[ok]] = ok
            return _dev_i32(inv, dev)
        src_major = C <= 256 and self.sw.ln_src_major
        reg = None if reg_np is None else self._cached(("reg", Dp, Hp, Wp, ws, ss, w3), lambda: torch.from_numpy(np.ascontiguousarray(reg_np[:, pm]) if w3 else reg_np).to(dev))
        rc = self._cached(("rc", N, win, w3), lambda: _dev_i32(rc_np[pm] if w3 else rc_np, dev))
        scale = 32 ** -0.5 if C // nh == 32 else (C // nh) ** -0.5
        # DropPath (video_swin.py:46-63): the block calls it TWICE -- on the attention branch (:256) and on the MLP branch (:248) -- with
        # independent per-sample draws; `dp` = (scale vector of the attention branch, scale vector of the MLP branch), or one vector for both
        dp, dp2 = (dp if isinstance(dp, (tuple, list)) else (dp, dp))
        # Dead clips of the attention branch: a clip whose DropPath draw is 0 gets x1 = x -- its LayerNorm, qkv, window attention and
        # projection contribute nothing, forward or backward.  The draw is known on the host, so the branch runs on the KEPT clips only:
        # the per-clip window map becomes the absolute row map of the kept clips (vmvm_expand_batch_map) for the gather-LayerNorm, the
        # projection's un-gather epilogue and the backward's gather; the dropped clips' rows are copied (vmvm_copy_batches_bf16).
        ds = dp if (isinstance(dp, DropScale) or dp is None) else DropScale(dp)
        dpv = None if ds is None else ds.dev
        ds2 = dp2 if (isinstance(dp2, DropScale) or dp2 is None) else DropScale(dp2)
        dp2 = None if ds2 is None else ds2.dev
        Bk, compact = B, False                                 # clips the attention branch runs on
        if ds is not None and self.sw.droppath_dce != "0":
            Bk, kept_a, dpk_a = ds.take(math.gcd(L, Lp), B)                   # (both row counts, Bk * L and Bk * Lp, in whole K tiles)
            drop_a, nd_a = ds.dropped, B - ds.n_kept
            compact = Bk < B and ds.scale is not None
            if not compact:
                Bk = B                                         # (nothing to eliminate / VMVM_DROPPATH_DCE=0: dropped clips are scaled by 0)
        x = xv.t
        g1, b1 = S.p(pre + "norm1.weight"), S.p(pre + "norm1.bias")
        table = S.p(pre + "attn.relative_position_bias_table")
        if compact and Bk == 0:                                # every clip dropped: the branch is the identity
            x1, src_k = x, None
        else:
            if compact:
                src_k = K.expand_batch_map(src, kept_a, Bk, L)         # [Bk * Lp] absolute rows (pads stay -1)
                lnkw = dict(M=Bk * Lp, C_=C, nseg=1, src=src_k, rows_out_per_batch=Bk * Lp, rows_in_per_batch=B * L, pad_mode=0)
                mapkw = dict(row_map=src_k, map_len=Bk * Lp, map_stride=0)
                dpk = dpk_a
            else:
                src_k = None
                lnkw = dict(M=B * Lp, C_=C, nseg=1, src=src, rows_out_per_batch=Lp, rows_in_per_batch=L, pad_mode=0)
                mapkw = dict(row_map=src, map_len=Lp, map_stride=L)
                dpk = dpv
            xw, mean1, rstd1 = K.layernorm_fwd(x, g1, b1, 1e-5, **lnkw)
            qkv = K.gemm(xw, S.b(pre + "attn.qkv.weight"), bias=S.p(pre + "attn.qkv.bias"), col_scale=scale, col_scale_n=C)
            akw = dict(q_off=0, k_off=C, v_off=2 * C, bias_table=table, rc=rc, rc0=rc0, region=reg, n_win=nW, seq_scale=dpk, seqs_per_scale=nW, win_layout=w3)
            ao, lse = K.attention_fwd(qkv, Bk * nW, N, nh, C // nh, 0, scale, **akw)
            x1 = K.gemm(ao, S.b(pre + "attn.proj.weight"), bias=S.p(pre + "attn.proj.bias"), row_scale=dpk, rows_per_scale=Lp,
                        scale_bias_only=True, resid=x, out_rows=B * L, **mapkw)
            if compact:
                K.copy_batches(x, x1, drop_a, nd_a, L)                # identity path of the dropped clips
        g2, b2 = S.p(pre + "norm2.weight"), S.p(pre + "norm2.bias")
        # saved for the GELU backward: an 8-bit code of GELU'(fc1 output) (vmvm_gemm_desc.aux_code8) where the persistent kernel's
        # whole-K-tile staging applies, the bf16 pre-activation otherwise
        c8 = getattr(self, "gelu_code8", True) and C % 64 == 0 and not S.frozen
        # the MLP branch on ITS kept clips (the second, independent draw): LayerNorm through an absolute identity map of the kept clips'
        # rows, fc1 compact, fc2 scattering back through the same map (+ residual); one extra gather of d(x2) in the backward
        Bm, compact2 = B, False
        if ds2 is not None and self.sw.droppath_dce not in ("0", "attn"):
            Bm, kept_m, dpk_m = ds2.take(L, B)
            drop_m, nd_m = ds2.dropped, B - ds2.n_kept
            compact2 = Bm < B and ds2.scale is not None
            if not compact2:
                Bm = B
        if compact2 and Bm == 0:
            x2 = x1
        else:
            if compact2:
                idm = self._cached(("idmap", L), lambda: _dev_i32(np.arange(L), dev))
                map_m = K.expand_batch_map(idm, kept_m, Bm, L)
                dpm = dpk_m
                y2, mean2, rstd2 = K.layernorm_fwd(x1, g2, b2, 1e-5, M=Bm * L, C_=C, nseg=1, src=map_m, rows_out_per_batch=Bm * L, rows_in_per_batch=B * L,
                                                   pad_mode=0)
                mkw = dict(row_map=map_m, map_len=Bm * L, map_stride=0, out_rows=B * L)
            else:
                map_m, dpm, mkw = None, dp2, {}
                y2, mean2, rstd2 = K.layernorm_fwd(x1, g2, b2, 1e-5)
            u = None if S.frozen else torch.empty((Bm * L, 4 * C), device=dev, dtype=torch.uint8 if c8 else BF16)     # frozen teacher: no backward, nothing saved
            h = K.gemm(y2, S.b(pre + "mlp.fc1.weight"), bias=S.p(pre + "mlp.fc1.bias"), act=1, out_preact=u, row_scale=dpm, rows_per_scale=L, code8=c8)
            x2 = K.gemm(h, S.b(pre + "mlp.fc2.weight"), bias=S.p(pre + "mlp.fc2.bias"), row_scale=dpm, rows_per_scale=L,
                        scale_bias_only=True, resid=x1, **mkw)
            if compact2:
                K.copy_batches(x1, x2, drop_m, nd_m, L)
        out = V(x2)

        # d(x1) straight in WINDOW order (blocks that run every clip, un-padded windows): the norm2 backward scatters its rows through the
        # inverse window map (vmvm_ln_bwd_desc.dx_map), the projection's backward GEMMs read them in order and the norm1 backward takes
        # them as its residual gradient by output row (add_by_out) -- no gather pass (2 x the activation bytes) and no natural-order d(x1)
        win_dx1 = self.sw.dx1_window and not compact and not compact2 and Lp == L

        def bwd():
            dx2 = out.g
            if win_dx1:
                du = self._linear_bwd(dx2, h, pre + "mlp.fc2.weight", pre + "mlp.fc2.bias", row_scale=dpm, rows_per_scale=L, cs_scale=ds2.scale if (ds2 is not None and ds2.n_kept == B) else None,
                                      dx_kw=dict(act=3, aux=u, row_scale=dpm, rows_per_scale=L, code8=c8))
                dy2 = self._linear_bwd(du, y2, pre + "mlp.fc1.weight", pre + "mlp.fc1.bias")
                inv_map = self._cached(("wminv", dims, ws, ss, w3), _inv_host)
                dx1w, _ = K.layernorm_bwd(dy2, x1, g2, mean2, rstd2, S.g(pre + "norm2.weight"), S.g(pre + "norm2.bias"), dX_add=dx2, dx_map=inv_map)
                dao = self._linear_bwd(dx1w, ao, pre + "attn.proj.weight", pre + "attn.proj.bias", row_scale=dpk, rows_per_scale=Lp,
                                       cs_scale=ds.scale if (ds is not None and ds.n_kept == B) else None)
                dqkv = self._window_attention_bwd(dao, qkv, ao, lse, Bk * nW, N, nh, C // nh, scale, S.g(pre + "attn.relative_position_bias_table"), akw)
                dxw = self._linear_bwd(dqkv, xw, pre + "attn.qkv.weight", pre + "attn.qkv.bias")
                dx, _ = K.layernorm_bwd(dxw, x, g1, mean1, rstd1, S.g(pre + "norm1.weight"), S.g(pre + "norm1.bias"), rows_in=B * L, nseg=1,
                                        pad_mode=0, dX_add=dx1w, add_by_out=True, src=src, rows_out_per_batch=Lp, rows_in_per_batch=L)
                _acc(xv, dx)
                return
            if compact2 and Bm == 0:
                dx1 = dx2
            else:
                dx2c = K.gather_rows(dx2, map_m, Bm * L) if compact2 else dx2
                # (compact: the padding clips' rows of dx2c are zeros and the kept clips share one scale -> the bias gradient stays fused)
                du = self._linear_bwd(dx2c, h, pre + "mlp.fc2.weight", pre + "mlp.fc2.bias", row_scale=dpm, rows_per_scale=L, cs_scale=ds2.scale if (compact2 or (ds2 is not None and ds2.n_kept == B)) else None,
                                      dx_kw=dict(act=3, aux=u, row_scale=dpm, rows_per_scale=L, code8=c8))
                dy2 = self._linear_bwd(du, y2, pre + "mlp.fc1.weight", pre + "mlp.fc1.bias")
                if compact2:
                    dx1, _ = K.layernorm_bwd(dy2, x1, g2, mean2, rstd2, S.g(pre + "norm2.weight"), S.g(pre + "norm2.bias"), rows_in=B * L, nseg=1,
                                             src=map_m, rows_out_per_batch=Bm * L, rows_in_per_batch=B * L, pad_mode=0, dX_add=dx2)
                    K.copy_batches(dx2, dx1, drop_m, nd_m, L)
                else:
                    dx1, _ = K.layernorm_bwd(dy2, x1, g2, mean2, rstd2, S.g(pre + "norm2.weight"), S.g(pre + "norm2.bias"), dX_add=dx2)
            if compact and Bk == 0:
                _acc(xv, dx1)
                return
            dx1w = K.gather_rows(dx1, src_k, Bk * Lp) if compact else K.gather_rows(dx1, src, B * Lp, Lp, L)
            dao = self._linear_bwd(dx1w, ao, pre + "attn.proj.weight", pre + "attn.proj.bias", row_scale=dpk, rows_per_scale=Lp,
                                   cs_scale=ds.scale if (compact or (ds is not None and ds.n_kept == B)) else None)      # (every clip kept: one scale as well)
            dqkv = self._window_attention_bwd(dao, qkv, ao, lse, Bk * nW, N, nh, C // nh, scale, S.g(pre + "attn.relative_position_bias_table"), akw)
            dxw = self._linear_bwd(dqkv, xw, pre + "attn.qkv.weight", pre + "attn.qkv.bias")
            bkw = dict(src=src_k, rows_out_per_batch=Bk * Lp, rows_in_per_batch=B * L) if compact else dict(src=src, rows_out_per_batch=Lp, rows_in_per_batch=L)
            if src_major:                                     # x / d(x1) / d(x) in order, only dY looked up through the (inverse) map
                bkw["inv"] = K.invert_map(src_k, B * L) if compact else self._cached(("wminv", dims, ws, ss, w3), _inv_host)
            dx, _ = K.layernorm_bwd(dxw, x, g1, mean1, rstd1, S.g(pre + "norm1.weight"), S.g(pre + "norm1.bias"), rows_in=B * L, nseg=1,
                                    pad_mode=0, dX_add=dx1, **bkw)
            if compact:
                K.copy_batches(dx1, dx, drop_a, nd_a, L)              # d(x) of the dropped clips = d(x1)
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def _patch_merge(self, xv, B, dims, C, pre):
        S = self.store
        D, H, W = dims
        mm, (D2, H2, W2) = SI.merge_map(D, H, W)
        src = self._cached(("mm", dims), lambda: _dev_i32(mm, self.device))
        L, Lo = D * H * W, D2 * H2 * W2
        x = xv.t
        gam, bet = S.p(pre + "norm.weight"), S.p(pre + "norm.bias")
        y, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-5, M=B * Lo, C_=4 * C, nseg=4, src=src, rows_out_per_batch=Lo, rows_in_per_batch=L, pad_mode=1)
        o = K.gemm(y, S.b(pre + "reduction.weight"))
        out = V(o)

        def bwd():
            dy = self._linear_bwd(out.g, y, pre + "reduction.weight", None)
            dx, _ = K.layernorm_bwd(dy, x, gam, mean, rstd, S.g(pre + "norm.weight"), S.g(pre + "norm.bias"), rows_in=B * L, nseg=4, src=src,
                                    rows_out_per_batch=Lo, rows_in_per_batch=L, pad_mode=1)
            _acc(xv, dx)
        self.tape.append(bwd)
        return out, (D2, H2, W2)

    def swin_forward(self, img, cov, dp_all, final_norm=True):
        """img f32 (B,T,3,H,W) -> V([B*T*h*w, 8E]) channels-last tokens (after the final norm; `final_norm=False`: the last
        stage's output, what HF SwinModel reports as hidden_states[-1])."""
        cfg, S = self.cfg, self.store
        B, T, _, H, W = img.shape
        xv = self._patch_embed(img, cov)
        dims = (T, H // 4, W // 4)
        C = cfg["embed_dim"]
        blk = 0
        n_st = len(cfg["depths"])
        for i, (d, nh) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
            if i == n_st - 2 and n_st >= 3:
                # runs in the backward right after stage n-2's last block: the gradients of stages >= n-2 are final
                self.tape.append(lambda: self.on_swin_tail_ready() if self.on_swin_tail_ready is not None else None)
            for b in range(d):
                dp = None if dp_all is None else dp_all[blk]
                xv = self._swin_block(xv, B, dims, C, nh, f"enc_img.swin.layers.{i}.blocks.{b}.", b % 2 == 1, dp)
                blk += 1
            if i < len(cfg["depths"]) - 1:
                xv, dims = self._patch_merge(xv, B, dims, C, f"enc_img.swin.layers.{i}.downsample.")
                C *= 2
        if not final_norm:
            return xv, dims, C
        x = xv.t
        gam, bet = S.p("enc_img.swin.norm.weight"), S.p("enc_img.swin.norm.bias")
        y, mean, rstd = K.layernorm_fwd(x, gam, bet, 1e-5)
        out = V(y)
        inp = xv

        def bwd():
            dx, _ = K.layernorm_bwd(out.g, x, gam, mean, rstd, S.g("enc_img.swin.norm.weight"), S.g("enc_img.swin.norm.bias"))
            _acc(inp, dx)
        self.tape.append(bwd)
        return out, dims, C
