"""Encoders and the fusion transformer of `VioletEngine` (engine.py): EncVideo / EncTxt assembly (model.py:32-115), HF BertLayer forward +
backward closures, the query-row form of the last layer, `go_cross` (model.py:204-214), `get_att`.  Methods of the engine class."""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import config as CFG
from . import kernels as K
from . import swin_index as SI
from .store import BF16, F32, V, DropScale, _acc, _gout, _h2d, _dev_i32


class FusionMixin:
    # -------------------------------------------------------------- EncVideo / EncTxt  -> one token pool
    def _drop_on(self, site, train):
        return bool(train) and (self._drop_sites is None or site in self._drop_sites)

    def encode(self, img, cov, txt, dp_all, train, odr=None):
        """returns pool V([B*Lv + NT*X, 768]) : rows [0, B*Lv) = feat_img (model.py:71), rest = feat_txt (model.py:107) of the NT =
        txt.shape[0] text sequences (NT = B in pre-training; B*O option sequences in multiple-choice QA)."""
        cfg, S, dev = self.cfg, self.store, self.device
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        Hd = cfg["hidden"]
        sw, dims, C8 = self.swin_forward(img, cov, dp_all)
        self.store.sync_pending()           # everything below reads non-Swin parameters: their AdamW update ran beside the Swin forward
        self.other_ready = None
        hw = dims[1] * dims[2]
        assert dims[1] == H // 32 and dims[2] == W // 32                       # model.py:34 hard-codes //32
        Lv = T * (1 + hw)
        has_fc = "enc_img.fc.weight" in S.index
        f = K.gemm(sw.t, S.b("enc_img.fc.weight"), bias=S.p("enc_img.fc.bias")) if has_fc else sw.t
        pos = S.p("enc_img.emb_pos", (1 + cfg["max_size_patch"] ** 2, Hd))
        ln_ = S.p("enc_img.emb_len", (cfg["max_size_frame"], Hd))
        if T > cfg["max_size_frame"]:
            raise RuntimeError(f"max_size_frame ({cfg['max_size_frame']}) must be >= T ({T})  (model.py:69)")
        if odr is None:
            pre = K.encvideo_assemble(f, S.p("enc_img.emb_cls", (Hd,)), pos, ln_, B, T, hw, Hd)
        else:
            # frame-order variant (model.py:61-67; no caller in the reference sets it, inference surface only): slot i of clip b adds
            # emb_len[i] when odr[b][i] == i, else emb_odr -- one frame table per clip, the same kernel on one clip at a time
            eo = S.p("enc_img.emb_odr", (1, Hd))
            hit = torch.as_tensor([[int(p_) == i for i, p_ in enumerate(o)] for o in odr], device=dev).view(B, T, 1)
            tabs = torch.where(hit, ln_[:T].unsqueeze(0), eo.unsqueeze(0)).contiguous()                  # f32 [B, T, Hd]
            pre = torch.empty((B * T * (1 + hw), Hd), device=dev, dtype=BF16)
            for b in range(B):
                K.encvideo_assemble(f[b * T * hw:(b + 1) * T * hw], S.p("enc_img.emb_cls", (Hd,)), pos, tabs[b], 1, T, hw, Hd,
                                    out=pre[b * T * (1 + hw):(b + 1) * T * (1 + hw)])
        pool = torch.empty((B * Lv + txt.shape[0] * X, Hd), device=dev, dtype=BF16)
        gi, bi = S.p("enc_img.norm.weight"), S.p("enc_img.norm.bias")
        fi, mean_i, rstd_i = K.layernorm_fwd(pre, gi, bi, 1e-5)
        pool[:B * Lv].copy_(fi)
        # text: HF BertEmbeddings (word + position + token_type(0)) -> LayerNorm(1e-12) -> dropout(0.1)
        pt = "enc_txt.emb_txt."
        e = K.bert_embed(txt, S.p(pt + "word_embeddings.weight"), S.p(pt + "position_embeddings.weight"),
                         S.p(pt + "token_type_embeddings.weight")[0])
        gt, bt = S.p(pt + "LayerNorm.weight"), S.p(pt + "LayerNorm.bias")
        ft, mean_t, rstd_t = K.layernorm_fwd(e, gt, bt, CFG.BERT["eps"])
        p_drop = CFG.BERT["hidden_dropout"] if self._drop_on("emb", train) else 0.0
        off_t = self._next_offset(ft.numel())
        self.last_offsets["emb"] = off_t
        if p_drop > 0:
            ft = K.dropout(ft, p_drop, self.seed, off_t)
        pool[B * Lv:].copy_(ft)
        out = V(pool)

        def bwd():
            if odr is not None:
                raise RuntimeError("odr is served on the inference surface only (go_feat / EncVideo.forward); no training path of the reference sets it")
            dpool = out.g                                                       # bf16 [B*Lv + B*X, Hd]
            dft = dpool[B * Lv:]
            if p_drop > 0:
                dft = K.dropout(dft, p_drop, self.seed, off_t)
            de, _ = K.layernorm_bwd(dft, e, gt, mean_t, rstd_t, S.g(pt + "LayerNorm.weight"), S.g(pt + "LayerNorm.bias"))
            K.bert_embed_bwd(txt, de, S.g(pt + "word_embeddings.weight"), S.g(pt + "position_embeddings.weight"),
                             S.g(pt + "token_type_embeddings.weight")[0])
            dpre, _ = K.layernorm_bwd(dpool[:B * Lv], pre, gi, mean_i, rstd_i, S.g("enc_img.norm.weight"), S.g("enc_img.norm.bias"))
            df = K.encvideo_assemble_bwd(dpre, S.g("enc_img.emb_cls", (Hd,)), S.g("enc_img.emb_pos", (1 + cfg["max_size_patch"] ** 2, Hd)),
                                         S.g("enc_img.emb_len", (cfg["max_size_frame"], Hd)), B, T, hw, Hd)
            if has_fc:
                df = self._linear_bwd(df, sw.t, "enc_img.fc.weight", "enc_img.fc.bias")
            _acc(sw, df)
        self.tape.append(bwd)
        return out, Lv, hw

    # -------------------------------------------------------------- fusion encoder
    def _bert_layer(self, xv, nseq, Lq, keymask, l, train, causal_from=0, att_out=None):
        if self.sw.block_abi and self.device.type == "cuda":
            return self._bert_layer_block(xv, nseq, Lq, keymask, l, train, causal_from, att_out)
        return self._bert_layer_calls(xv, nseq, Lq, keymask, l, train, causal_from, att_out)

    def _bert_layer_block(self, xv, nseq, Lq, keymask, l, train, causal_from=0, att_out=None):
        """HF BertLayer `l` through the block-level C ABI (include/vmvm.h vmvm_bert_layer; csrc/blocks.hip fills the same per-kernel
        descriptors as `_bert_layer_calls` below, which stays as the statement of the layer and as the other side of the bit-for-bit
        test): one descriptor + one foreign call for the forward (10 launches) and one for the backward (22 launches, the four weight
        gradients on the engine's second stream).  This function only allocates and names buffers."""
        from . import lib as L
        import ctypes as C
        S, dev = self.store, self.device
        pre = f"trsfr.layer.{l}."
        Hd, nh, F = self.cfg["hidden"], CFG.BERT["heads"], CFG.BERT["ffn"]
        qn = [pre + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        bn = [pre + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        M = nseq * Lq
        x = xv.t
        p_h = CFG.BERT["hidden_dropout"] if train else 0.0
        p_a = CFG.BERT["attn_dropout"] if train else 0.0
        c8 = self.gelu_code8
        d = L.BertLayer()
        d.nseq, d.L, d.hidden, d.heads, d.ffn = nseq, Lq, Hd, nh, F
        wo, w1, w2 = pre + "attention.output.dense.weight", pre + "intermediate.dense.weight", pre + "output.dense.weight"
        d.Wqkv, d.Wo, d.W1, d.W2 = S.fused(S.shadow, qn, (3 * Hd, Hd)).data_ptr(), S.b(wo).data_ptr(), S.b(w1).data_ptr(), S.b(w2).data_ptr()
        wqt = S.bt(qn[0])                                   # the fused [H, 3H] transposed copy (query | key | value adjacent), or none
        d.WqkvT = L.ptr(wqt if (wqt is not None and tuple(wqt.shape) == (Hd, 3 * Hd)) else None)
        d.WoT, d.W1T, d.W2T = (L.ptr(S.bt(n_)) for n_ in (wo, w1, w2))
        d.bqkv = S.fused(S.flat, bn, (3 * Hd,)).data_ptr()
        d.bo, d.b1, d.b2 = (S.p(pre + n_).data_ptr() for n_ in ("attention.output.dense.bias", "intermediate.dense.bias", "output.dense.bias"))
        d.ln1_g, d.ln1_b = S.p(pre + "attention.output.LayerNorm.weight").data_ptr(), S.p(pre + "attention.output.LayerNorm.bias").data_ptr()
        d.ln2_g, d.ln2_b = S.p(pre + "output.LayerNorm.weight").data_ptr(), S.p(pre + "output.LayerNorm.bias").data_ptr()
        d.ln_eps = CFG.BERT["eps"]
        if not S.frozen:
            d.gWqkv, d.gbqkv = S.fused(S.grad, qn, (3 * Hd, Hd)).data_ptr(), S.fused(S.grad, bn, (3 * Hd,)).data_ptr()
            d.gWo, d.gW1, d.gW2 = S.g(wo).data_ptr(), S.g(w1).data_ptr(), S.g(w2).data_ptr()
            d.gbo, d.gb1, d.gb2 = (S.g(pre + n_).data_ptr() for n_ in ("attention.output.dense.bias", "intermediate.dense.bias", "output.dense.bias"))
            d.gln1_g, d.gln1_b = S.g(pre + "attention.output.LayerNorm.weight").data_ptr(), S.g(pre + "attention.output.LayerNorm.bias").data_ptr()
            d.gln2_g, d.gln2_b = S.g(pre + "output.LayerNorm.weight").data_ptr(), S.g(pre + "output.LayerNorm.bias").data_ptr()
        e = torch.empty
        qkv, ctx, lse = e((M, 3 * Hd), device=dev, dtype=BF16), e((M, Hd), device=dev, dtype=BF16), e((nseq, nh, Lq), device=dev, dtype=F32)
        a, x1, mean1, rstd1 = e((M, Hd), device=dev, dtype=BF16), e((M, Hd), device=dev, dtype=BF16), e(M, device=dev, dtype=F32), e(M, device=dev, dtype=F32)
        u, h = e((M, F), device=dev, dtype=torch.uint8 if c8 else BF16), e((M, F), device=dev, dtype=BF16)
        f, x2, mean2, rstd2 = e((M, Hd), device=dev, dtype=BF16), e((M, Hd), device=dev, dtype=BF16), e(M, device=dev, dtype=F32), e(M, device=dev, dtype=F32)
        d.x, d.qkv, d.ctx, d.lse, d.a, d.x1, d.mean1, d.rstd1 = (t_.data_ptr() for t_ in (x, qkv, ctx, lse, a, x1, mean1, rstd1))
        d.u, d.code8, d.h, d.f, d.x2, d.mean2, d.rstd2 = u.data_ptr(), int(c8), h.data_ptr(), f.data_ptr(), x2.data_ptr(), mean2.data_ptr(), rstd2.data_ptr()
        d.keymask, d.causal_from, d.att_colsum = L.ptr(keymask), causal_from, L.ptr(att_out)
        # Philox offsets in the order `_bert_layer_calls` draws them
        d.off_attn = self._next_offset(nseq * nh * Lq * Lq)
        dmask = None
        if p_a > 0 and self.store_drop_mask:
            dmask = K.attention_drop_mask(nseq, Lq, nh, Hd // nh, 1, p_a, dev, causal_from=causal_from, att_colsum=att_out)
        d.drop_mask = L.ptr(dmask)
        d.off_1 = self._next_offset(M * Hd)
        d.off_2 = self._next_offset(M * Hd)
        d.p_hidden, d.p_attn, d.seed = p_h, p_a, self.seed
        x8 = x18 = None
        if self.fp8:
            x8, x18 = e((M, Hd), device=dev, dtype=torch.uint8), e((M, Hd), device=dev, dtype=torch.uint8)
            d.in_fp8, d.Wqkv8, d.W18 = 1, S.fused8(qn, (3 * Hd, Hd)).data_ptr(), S.b8(w1).data_ptr()
            d.x8, d.x18, d.a8_scale, d.w8_scale = x8.data_ptr(), x18.data_ptr(), self.A8_SCALE, S.W8_SCALE
        d.reserve_cus = K.reserve_cus()
        lib = L.load()
        L.check(lib.vmvm_bert_layer_fwd(C.byref(d), L.stream()), "bert_layer_fwd")
        out = V(x2)
        # the descriptor holds raw addresses: every forward buffer the backward re-reads must stay referenced until the closure has run
        saved = (x, qkv, ctx, lse, a, x1, mean1, rstd1, u, h, f, mean2, rstd2, dmask, keymask)

        def bwd():
            assert saved is not None
            dg = out.g
            df, du, dx1, da = e((M, Hd), device=dev, dtype=BF16), e((M, F), device=dev, dtype=BF16), e((M, Hd), device=dev, dtype=BF16), e((M, Hd), device=dev, dtype=BF16)
            dfm = e((M, Hd), device=dev, dtype=BF16) if p_h > 0 else None
            dam = e((M, Hd), device=dev, dtype=BF16) if p_h > 0 else None
            dctx, dqkv, delta = e((M, Hd), device=dev, dtype=BF16), e((M, 3 * Hd), device=dev, dtype=BF16), e((nseq, nh, Lq), device=dev, dtype=F32)
            dx = _gout(xv)
            dx = e((M, Hd), device=dev, dtype=BF16) if dx is None else dx
            d.d_out, d.d_x = dg.data_ptr(), dx.data_ptr()
            d.df, d.dfm, d.du, d.dx1, d.da, d.dam, d.dctx, d.dqkv, d.delta = (L.ptr(t_) for t_ in (df, dfm, du, dx1, da, dam, dctx, dqkv, delta))
            wsm = K._WORKSPACE.get(x.device)
            d.ws_main, d.ws_main_bytes = L.ptr(wsm), (wsm.numel() if wsm is not None else 0)
            side = self.wstream
            d.ws_side, d.ws_side_bytes = (self.workspace_w.data_ptr(), self.workspace_w.numel()) if side is not None else (0, 0)
            d.reserve_cus = K.reserve_cus()
            L.check(lib.vmvm_bert_layer_bwd(C.byref(d), L.stream(), side.cuda_stream if side is not None else None,
                                            self._fork_event() if side is not None else None), "bert_layer_bwd")
            # the side stream reads these until it has passed the four weight gradients (the main-stream buffers die with this closure)
            self._whold((dfm if dfm is not None else df, h, du, x1, dam if dam is not None else da, ctx, dqkv, x))
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def _bert_layer_calls(self, xv, nseq, Lq, keymask, l, train, causal_from=0, att_out=None):
        S, dev = self.store, self.device
        pre = f"trsfr.layer.{l}."
        Hd, nh = self.cfg["hidden"], CFG.BERT["heads"]
        qn = [pre + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        bn = [pre + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        Wqkv, Gqkv = S.fused(S.shadow, qn, (3 * Hd, Hd)), S.fused(S.grad, qn, (3 * Hd, Hd))
        bqkv, gbqkv = S.fused(S.flat, bn, (3 * Hd,)), S.fused(S.grad, bn, (3 * Hd,))
        p_h = CFG.BERT["hidden_dropout"] if train else 0.0
        p_a = CFG.BERT["attn_dropout"] if train else 0.0
        M = nseq * Lq
        x = xv.t
        a8 = 1.0 / (self.A8_SCALE * S.W8_SCALE)
        if self.fp8:
            qkv = K.gemm(K.cast_fp8(x, self.A8_SCALE), S.fused8(qn, (3 * Hd, Hd)), bias=bqkv, fp8=True, alpha=a8)
        else:
            qkv = K.gemm(x, Wqkv, bias=bqkv)
        o_att = self._next_offset(nseq * nh * Lq * Lq)
        akw = dict(q_off=0, k_off=Hd, v_off=2 * Hd, keymask=keymask, dropout_p=p_a, seed=self.seed, offset=o_att, causal_from=causal_from)
        if p_a > 0 and self.store_drop_mask:                     # the forward's keep / drop decisions, read back by both backward kernels
            akw["drop_mask"] = K.attention_drop_mask(nseq, Lq, nh, Hd // nh, 1, p_a, dev, causal_from=causal_from, att_colsum=att_out)
        ctx, lse = K.attention_fwd(qkv, nseq, Lq, nh, Hd // nh, 1, 1.0 / math.sqrt(Hd // nh), att_colsum=att_out, **akw)
        o1 = self._next_offset(M * Hd)
        a = K.gemm(ctx, S.b(pre + "attention.output.dense.weight"), bias=S.p(pre + "attention.output.dense.bias"), resid=x,
                   dropout_p=p_h, seed=self.seed, offset=o1)
        g1, b1 = S.p(pre + "attention.output.LayerNorm.weight"), S.p(pre + "attention.output.LayerNorm.bias")
        x1, mean1, rstd1 = K.layernorm_fwd(a, g1, b1, CFG.BERT["eps"])
        c8 = self.gelu_code8                                     # GELU' saved as an 8-bit code (as in the Swin MLPs)
        u = torch.empty((M, CFG.BERT["ffn"]), device=dev, dtype=torch.uint8 if c8 else BF16)
        if self.fp8:
            h = K.gemm(K.cast_fp8(x1, self.A8_SCALE), S.b8(pre + "intermediate.dense.weight"), bias=S.p(pre + "intermediate.dense.bias"),
                       act=1, out_preact=u, fp8=True, alpha=a8, code8=c8)
        else:
            h = K.gemm(x1, S.b(pre + "intermediate.dense.weight"), bias=S.p(pre + "intermediate.dense.bias"), act=1, out_preact=u, code8=c8)
        o2 = self._next_offset(M * Hd)
        f = K.gemm(h, S.b(pre + "output.dense.weight"), bias=S.p(pre + "output.dense.bias"), resid=x1, dropout_p=p_h, seed=self.seed, offset=o2)
        g2, b2 = S.p(pre + "output.LayerNorm.weight"), S.p(pre + "output.LayerNorm.bias")
        x2, mean2, rstd2 = K.layernorm_fwd(f, g2, b2, CFG.BERT["eps"])
        out = V(x2)

        def bwd():
            df, dfm = K.layernorm_bwd(out.g, f, g2, mean2, rstd2, S.g(pre + "output.LayerNorm.weight"), S.g(pre + "output.LayerNorm.bias"),
                                      want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o2)
            dfm = df if dfm is None else dfm
            du = self._linear_bwd(dfm, h, pre + "output.dense.weight", pre + "output.dense.bias", dx_kw=dict(act=3, aux=u, code8=c8))
            dx1 = self._linear_bwd(du, x1, pre + "intermediate.dense.weight", pre + "intermediate.dense.bias", dx_kw=dict(resid=df))
            da, dam = K.layernorm_bwd(dx1, a, g1, mean1, rstd1, S.g(pre + "attention.output.LayerNorm.weight"),
                                      S.g(pre + "attention.output.LayerNorm.bias"), want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o1)
            dam = da if dam is None else dam
            dctx = self._linear_bwd(dam, ctx, pre + "attention.output.dense.weight", pre + "attention.output.dense.bias")
            dqkv = K.attention_bwd(dctx, qkv, ctx, lse, nseq, Lq, nh, Hd // nh, 1, 1.0 / math.sqrt(Hd // nh), **akw)
            dx = self._linear_bwd(dqkv, x, None, None, w=Wqkv, gw=Gqkv, gb=gbqkv, dx_kw=dict(resid=da, out=_gout(xv)), wT=S.bt(qn[0]))
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def _bert_layer_qrow(self, xv, nseq, Lq, qpos, keymask, l, train):
        """HF BertLayer `l` for sequences of which ONLY the output at position `qpos` is read -- the VTM pass reads the encoder's last
        hidden state at the text [CLS] position (main_pretrain.py:260: out[:, T*(1+hw), :]), so in the LAST fusion layer every other
        query row of those sequences is dead code, forward and backward (their d(out) is zero).  K and V of every position are still
        computed (one GEMM on the key / value rows of the fused QKV weight); the query, the attention row (`vmvm_attn_query_row_*`),
        both dense layers, the FFN and both LayerNorms run on nseq rows instead of nseq * Lq.  Same arithmetic per row as
        `_bert_layer`; returns V([nseq, H])."""
        S, dev = self.store, self.device
        pre = f"trsfr.layer.{l}."
        Hd, nh = self.cfg["hidden"], CFG.BERT["heads"]
        hd = Hd // nh
        qn = [pre + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        bn = [pre + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        Wqkv, Gqkv = S.fused(S.shadow, qn, (3 * Hd, Hd)), S.fused(S.grad, qn, (3 * Hd, Hd))
        bqkv, gbqkv = S.fused(S.flat, bn, (3 * Hd,)), S.fused(S.grad, bn, (3 * Hd,))
        WT = S.bt(qn[0])                                                  # fused W^T [H, 3H] (or None)
        p_h = CFG.BERT["hidden_dropout"] if train else 0.0
        p_a = CFG.BERT["attn_dropout"] if train else 0.0
        scale = 1.0 / math.sqrt(hd)
        x = xv.t                                                          # [nseq * Lq, H]
        rows = self._cached(("qrow", nseq, Lq, qpos), lambda: _dev_i32(np.arange(nseq) * Lq + qpos, dev))
        kv = K.gemm(x, Wqkv[Hd:], bias=bqkv[Hd:])                         # K | V of every position  [nseq * Lq, 2H]
        xc = K.gather_rows(x, rows, nseq)                                 # the query rows  [nseq, H]
        q = K.gemm(xc, Wqkv[:Hd], bias=bqkv[:Hd])
        o_att = self._next_offset(nseq * nh * Lq)
        ctx, pr, prd = K.attn_query_row_fwd(q, kv, nseq, Lq, nh, hd, scale, k_off=0, v_off=Hd, keymask=keymask, dropout_p=p_a, seed=self.seed, offset=o_att)
        o1 = self._next_offset(nseq * Hd)
        a = K.gemm(ctx, S.b(pre + "attention.output.dense.weight"), bias=S.p(pre + "attention.output.dense.bias"), resid=xc,
                   dropout_p=p_h, seed=self.seed, offset=o1)
        g1, b1 = S.p(pre + "attention.output.LayerNorm.weight"), S.p(pre + "attention.output.LayerNorm.bias")
        x1, mean1, rstd1 = K.layernorm_fwd(a, g1, b1, CFG.BERT["eps"])
        u = torch.empty((nseq, CFG.BERT["ffn"]), device=dev, dtype=BF16)
        h = K.gemm(x1, S.b(pre + "intermediate.dense.weight"), bias=S.p(pre + "intermediate.dense.bias"), act=1, out_preact=u)
        o2 = self._next_offset(nseq * Hd)
        f = K.gemm(h, S.b(pre + "output.dense.weight"), bias=S.p(pre + "output.dense.bias"), resid=x1, dropout_p=p_h, seed=self.seed, offset=o2)
        g2, b2 = S.p(pre + "output.LayerNorm.weight"), S.p(pre + "output.LayerNorm.bias")
        x2, mean2, rstd2 = K.layernorm_fwd(f, g2, b2, CFG.BERT["eps"])
        out = V(x2)

        def bwd():
            df, dfm = K.layernorm_bwd(out.g, f, g2, mean2, rstd2, S.g(pre + "output.LayerNorm.weight"), S.g(pre + "output.LayerNorm.bias"),
                                      want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o2)
            dfm = df if dfm is None else dfm
            du = self._linear_bwd(dfm, h, pre + "output.dense.weight", pre + "output.dense.bias", dx_kw=dict(act=3, aux=u))
            dx1 = self._linear_bwd(du, x1, pre + "intermediate.dense.weight", pre + "intermediate.dense.bias", dx_kw=dict(resid=df))
            da, dam = K.layernorm_bwd(dx1, a, g1, mean1, rstd1, S.g(pre + "attention.output.LayerNorm.weight"),
                                      S.g(pre + "attention.output.LayerNorm.bias"), want_dX2=p_h > 0, dropout_p=p_h, seed=self.seed, offset=o1)
            dam = da if dam is None else dam
            dctx = self._linear_bwd(dam, ctx, pre + "attention.output.dense.weight", pre + "attention.output.dense.bias")
            dq, dkv = K.attn_query_row_bwd(dctx, q, kv, pr, prd, nseq, Lq, nh, hd, scale, k_off=0, v_off=Hd)
            # query projection: d(xc) = dq Wq + da (the residual of the attention block); key / value projection: d(x) = dkv Wkv
            dxc = self._linear_bwd(dq, xc, None, None, w=Wqkv[:Hd], gw=Gqkv[:Hd], gb=gbqkv[:Hd], dx_kw=dict(resid=da),
                                   wT=None if WT is None else WT[:, :Hd])
            dx = self._linear_bwd(dkv, x, None, None, w=Wqkv[Hd:], gw=Gqkv[Hd:], gb=gbqkv[Hd:], wT=None if WT is None else WT[:, Hd:], dx_kw=dict(out=_gout(xv)))
            dx.index_add_(0, rows.long(), dxc)                            # (nseq rows; plumbing)
            _acc(xv, dx)
        self.tape.append(bwd)
        return out

    def go_cross(self, pool, idx, keymask, nseq, Lq, train, causal_from=0, att_out=None, qrow_split=None, mid_hook=False):
        """gather the [img;txt] sequences from the token pool and run the 12 fusion layers (model.py:204-214).
        causal_from = Lv: the seq2seq mask of the smtm pass (main_pretrain.py:217-224, model.py:191-199)."""
        Hd = self.cfg["hidden"]
        x = K.gather_rows(pool.t, idx, nseq * Lq)
        xv = V(x)
        cur = xv
        nl = self.cfg["bert_layers"]
        for l in range(nl - 1 if qrow_split is not None else nl):
            if mid_hook and nl >= 4 and l == nl // 2:
                # a closure pushed BEFORE layer nl // 2 runs right AFTER that layer's backward: the gradients of layers nl // 2 .. nl - 1
                # (and of the heads) are final there -- data parallel: their exchange starts now (dist.GradReducer.reduce_other_early)
                self.tape.append(lambda: self.on_fusion_mid_ready() if self.on_fusion_mid_ready is not None else None)
            cur = self._bert_layer(cur, nseq, Lq, keymask, l, train, causal_from, att_out)
        if qrow_split is None:
            return cur, xv, idx
        # last layer: the first n1 sequences in full, of the others only the row at `qpos` (see _bert_layer_qrow)
        n1, qpos = qrow_split
        if causal_from != 0 or att_out is not None:     # _bert_layer_qrow has neither the seq2seq mask nor the attention capture
            raise RuntimeError("go_cross(qrow_split=...) serves the plain key-mask pass only (no causal_from / att_out)")
        if n1 == 0:                               # every sequence: only the row at `qpos` (retrieval / open-ended QA read the text [CLS] state only)
            return (None, self._bert_layer_qrow(cur, nseq, Lq, qpos, keymask, nl - 1, train)), xv, idx
        gbuf = torch.empty_like(cur.t)                                         # d(layer input): the two halves' backward write their rows side by side
        xa = V(cur.t[:n1 * Lq], gbuf[:n1 * Lq])
        xb = V(cur.t[n1 * Lq:], gbuf[n1 * Lq:])
        prev = cur

        def join():                               # runs AFTER the two halves' backward closures
            _acc(prev, gbuf)
        self.tape.append(join)
        out_a = self._bert_layer(xa, n1, Lq, keymask[:n1], nl - 1, train, causal_from, att_out)
        out_b = self._bert_layer_qrow(xb, nseq - n1, Lq, qpos, keymask[n1:], nl - 1, train)
        return (out_a, out_b), xv, idx

    @torch.no_grad()
    def get_att(self, img, txt, mask, train=True, dp_all=None, cov=None):
        """VIOLET_Pretrain.get_att (main_pretrain.py:211-215): one (img_i, txt_i) fusion pass whose attention kernels also
        accumulate the head-averaged column sums of every layer -> (B, T*(1+hw)+X) f32, the sampling weights of the 'am' masking.
        `train` keeps dropout / DropPath on, as the reference calls it from masking() with the model in train mode."""
        dev = self.device
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        saved, self.tape = self.tape, []
        if train and dp_all is None:
            dp_all = self.sample_drop_path(B)
        cov_d = None if cov is None else cov.to(dev, torch.uint8).contiguous()
        pool, Lv, hw = self.encode(img.to(dev, F32).contiguous(), cov_d, txt.to(dev).contiguous(), dp_all, train)
        Lq = Lv + X
        ar_v, ar_t = np.arange(Lv), np.arange(X)
        idx1 = _dev_i32(np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + i * X + ar_t]) for i in range(B)]), dev)
        km1 = torch.cat([torch.ones(B, Lv, dtype=torch.uint8, device=dev), (mask.to(dev) != 0).to(torch.uint8)], 1).contiguous()
        att = torch.zeros((B, Lq), device=dev, dtype=F32)
        self.go_cross(pool, idx1, km1, B, Lq, train, att_out=att)
        self.tape = saved
        return att
