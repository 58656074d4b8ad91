"""Pre-training heads, losses and the step schedule of `VioletEngine` (engine.py): MLM head, VTM head, MVM targets, `forward_backward`
(VIOLET_Pretrain.forward main_pretrain.py:226-267, heads + losses :374-432, 555-567).  Methods of the engine class."""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import config as CFG
from . import kernels as K
from . import lib as L
from . import swin_index as SI
from .store import BF16, F32, V, DropScale, _acc, _gout, _h2d, _dev_i32


class HeadsMixin:
    # -------------------------------------------------------------- MLM head (HF BertOnlyMLMHead), shared by every pass that reads it
    def _mlm_dims(self):
        Vv = self.cfg["vocab"]
        return Vv, -(-Vv // 8) * 8, -(-Vv // 4) * 4          # vocabulary, row pitch of the f32 logits, columns the GEMM writes

    def _mlm_head_fwd(self, rows, n_rows, target, loss, want_grad):
        """dense + GELU + LayerNorm + decoder (+ bias) + cross entropy(ignore -1) on `rows` [n_rows, H] (main_pretrain.py:236,560)."""
        S, dev, Hd = self.store, self.device, self.cfg["hidden"]
        pm = "fc_mtm.predictions."
        Vv, Vpad, Nlog = self._mlm_dims()
        u_ = torch.empty((n_rows, Hd), device=dev, dtype=BF16)
        t_ = K.gemm(rows, S.b(pm + "transform.dense.weight"), bias=S.p(pm + "transform.dense.bias"), act=1, out_preact=u_)
        gm, bm = S.p(pm + "transform.LayerNorm.weight"), S.p(pm + "transform.LayerNorm.bias")
        tn_, mean_, rstd_ = K.layernorm_fwd(t_, gm, bm, CFG.BERT["eps"])
        lg_ = torch.empty((n_rows, Vpad), device=dev, dtype=F32)
        K.gemm(tn_, S.b(pm + "decoder.weight"), N=Nlog, bias=S.p(pm + "bias"), out=lg_)
        dlog_ = K.cross_entropy(lg_, Vv, target, loss, want_grad=want_grad, ld_d=Vpad)
        return dict(r=rows, u=u_, t=t_, tn=tn_, mean=mean_, rstd=rstd_, logits=lg_, dlog=dlog_, n=n_rows)

    def _mlm_head_bwd(self, hd, dx_out=None):
        """head gradients (accumulated into the shared fc_mtm.* tensors); returns / writes d(rows)."""
        S, dev, Hd = self.store, self.device, self.cfg["hidden"]
        pm = "fc_mtm.predictions."
        Vv, Vpad, _ = self._mlm_dims()
        n = hd["n"]
        gm = S.p(pm + "transform.LayerNorm.weight")
        Wdec = S.b(pm + "decoder.weight")
        def dec_wgrad(ws):
            K.colsum(hd["dlog"], S.g(pm + "bias"), accumulate=True, M=n, N=Vpad, workspace=ws)     # pad columns are zero and land in arena padding
            K.gemm(hd["dlog"], hd["tn"], a_kmajor=False, b_kmajor=False, M=Vv, N=Hd, K=n, out=S.g(pm + "decoder.weight"), accumulate=True, workspace=ws)
        self._wgrad_launch(dec_wgrad, (hd["dlog"], hd["tn"]))
        # d(tn) = dlog . W over K = the PADDED vocabulary when that is a whole number of 64-wide K tiles (30522 -> 30528): the GEMM
        # then takes the direct-to-LDS kernel instead of the K % 64 != 0 fallback (497 -> 60 us).  Invariants this relies on:
        #  (i) the pad columns [Vv, Vpad) of dlog are exact zeros (vmvm_cross_entropy writes them);
        #  (ii) the (Vpad - Vv) extra rows of the [Vpad, H] weight view lie INSIDE the bf16 arena (other parameters or its zero tail:
        #       (Vpad - Vv) * H <= ParamStore.TAIL) and are FINITE, so 0 * w = 0 -- checked after every optimizer step by the
        #       clip coefficient being finite (a non-finite parameter makes every loss NaN long before it matters here).
        # 48 output tiles and a 30528-long reduction: as an f32 accumulation the GEMM splits K over the chip (444 -> ~70 us), its
        # partial slabs going through the engine's split-K workspace and a fixed-order reduce (run-to-run deterministic).
        Kdec = Vpad if (Vpad % 64 == 0 and (Vpad - Vv) * Hd <= S.TAIL) else Vv
        dtn32 = torch.zeros((n, Hd), device=dev, dtype=F32)
        K.gemm(hd["dlog"], Wdec, b_kmajor=False, M=n, N=Hd, K=Kdec, out=dtn32, accumulate=True)
        dtn = dtn32.to(BF16)
        dt_, _ = K.layernorm_bwd(dtn, hd["t"], gm, hd["mean"], hd["rstd"], S.g(pm + "transform.LayerNorm.weight"), S.g(pm + "transform.LayerNorm.bias"))
        du_ = K.gelu_bwd(dt_, hd["u"])
        return self._linear_bwd(du_, hd["r"], pm + "transform.dense.weight", pm + "transform.dense.bias",
                                dx_kw=None if dx_out is None else dict(out=dx_out))

    # -------------------------------------------------------------- the trunk both step forms share
    def _forward_trunk(self, batch, negatives, train, dp_all, dropout, backward):
        """encoders -> token pool -> sequence assembly -> ONE fusion pass over the B pass-1 and B*O VTM sequences (+ the smtm pass):
        everything of VIOLET_Pretrain.forward (main_pretrain.py:226-262) in front of the heads.  Returns the record the heads and
        `_backward_trunk` read."""
        cfg, S, dev = self.cfg, self.store, self.device
        img, cov, txt, mask = batch["img"], batch["cov"], batch["txt"], batch["mask"]
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        O = min(B, 4)
        self.tape = []
        if train and dp_all is None:
            dp_all = self.sample_drop_path(B)
        # dropout sites follow `train` unless overridden: dropout = False / True for all of them, or a collection of site names out of
        # {"emb" (BertEmbeddings), "fusion" (the 12 BertLayers), "vtm" (the VTM head's Dropout, main_pretrain.py:146)} -- parity tests
        # switch sites on one group at a time and feed the kernels' own masks to the oracle
        self._drop_sites = None if (dropout is None or isinstance(dropout, bool)) else frozenset(dropout)
        train = train if (dropout is None or self._drop_sites is not None) else bool(dropout)
        feat_target = batch.get("feature_target")
        if feat_target is None and self.feature_teacher is not None:
            feat_target = self.feature_teacher.features(img)     # frozen Swin teacher first: its activations are gone before the student's pile up
        pool, Lv, hw = self.encode(img, cov, txt, dp_all, train)
        Lq = Lv + X
        # ---- sequence assembly indices (pass 1: (img_i, txt_i); pass 2: (img_i, txt_i), (img_i, txt_neg) ...)
        if negatives is None:
            negatives = self.sample_negatives(B)
        # (vectorised: as Python loops over the B * O pairs this block cost the host 12.8 ms per step at B = 32, tools/scratch/lead_probe.py)
        ar_v, ar_t = np.arange(Lv), np.arange(X)

        def seq_rows(pi, pj):                               # rows of the (visual tokens of clip pi | text tokens of clip pj) sequences
            pi, pj = np.asarray(pi, dtype=np.int64), np.asarray(pj, dtype=np.int64)
            return np.concatenate([pi[:, None] * Lv + ar_v[None, :], B * Lv + pj[:, None] * X + ar_t[None, :]], axis=1).reshape(-1)
        idx1 = self._cached(("idx1", B, Lv, X), lambda: seq_rows(np.arange(B), np.arange(B)))
        neg = np.asarray(negatives, dtype=np.int64).reshape(B, -1)[:, :O - 1] if O > 1 else np.zeros((B, 0), np.int64)
        pair_i = np.repeat(np.arange(B), O)
        pair_j = np.concatenate([np.arange(B)[:, None], neg], axis=1).reshape(-1)
        idx2 = seq_rows(pair_i, pair_j)
        use_smtm = "smtm" in cfg.get("pretrain_tasks", ())
        idx1_d = _dev_i32(idx1, dev) if use_smtm else None
        km_txt = (mask != 0).to(torch.uint8)
        km1 = torch.cat([torch.ones(B, Lv, dtype=torch.uint8, device=dev), km_txt], 1).contiguous()
        tj_h = pair_j
        tj = _h2d(torch.from_numpy(tj_h), dev)
        km2 = torch.cat([torch.ones(B * O, Lv, dtype=torch.uint8, device=dev), km_txt[tj]], 1).contiguous()
        txt_off_d = txt_list_d = None
        if backward:                            # CSR of the pass-2 sequences by their text index (the pool gradient gathers through it)
            order = np.argsort(tj_h, kind="stable")
            csr = np.concatenate([np.concatenate([[0], np.cumsum(np.bincount(tj_h, minlength=B))]), order]).astype(np.int32)
            csr_d = _dev_i32(csr, dev)
            txt_off_d, txt_list_d = csr_d[:B + 1], csr_d[B + 1:]

        # ONE fusion pass over the B sequences of pass 1 (model.py:204-214 via main_pretrain.py:233) and the B*O sequences of the VTM
        # pass (:243-259) together: sequences are independent through the encoder, so every layer kernel runs once on (1 + O) * B
        # sequences instead of twice (the B-sequence launches filled 0.2 - 0.6 of a round of the persistent GEMM grids)
        n1, n2 = B, B * O
        idx12_d = _dev_i32(np.concatenate([idx1, idx2]), dev)
        km12 = torch.cat([km1, km2], 0).contiguous()
        # In the LAST layer only the text [CLS] row of the VTM sequences is alive (the VTM head reads nothing else, :260): it runs as
        # `_bert_layer_qrow` on n2 rows instead of n2 * Lq (`go_cross(qrow_split=...)`); results are those of the full layer.
        ntape = len(self.tape)
        qrow = self.sw.qrow                             # (0: the whole last layer for every sequence, for A/B runs)
        out12 = cls_rows = None
        if qrow:
            (out1, out2c), in12, _ = self.go_cross(pool, idx12_d, km12, n1 + n2, Lq, self._drop_on("fusion", train), qrow_split=(n1, Lv), mid_hook=True)
            if backward:
                out1.g = torch.empty_like(out1.t)                                  # the heads write it in place
        else:
            out12, in12, _ = self.go_cross(pool, idx12_d, km12, n1 + n2, Lq, self._drop_on("fusion", train), mid_hook=True)
            cls_rows = self._cached(("cls_rows", B * O, Lq, Lv), lambda: _dev_i32(np.arange(B * O) * Lq + Lv, dev))
            out1, out2c = V(out12.t[:n1 * Lq]), V(K.gather_rows(out12.t[n1 * Lq:], cls_rows, n2))
            if backward:
                out12.g = torch.empty_like(out12.t)
                out1.g = out12.g[:n1 * Lq]
        n_fusion_closures = len(self.tape) - ntape
        out3 = in3 = None
        if use_smtm:                            # third pass under the seq2seq mask (main_pretrain.py:238-240)
            out3, in3, _ = self.go_cross(pool, idx1_d, km1, B, Lq, self._drop_on("fusion", train), causal_from=Lv)
        txt_rows = self._cached(("txt_rows", B, Lv, X), lambda: _dev_i32(np.concatenate([i * Lq + Lv + ar_t for i in range(B)]), dev))
        return dict(B=B, T=T, H=H, W=W, X=X, O=O, Lv=Lv, Lq=Lq, hw=hw, n1=n1, n2=n2, train=train, feat_target=feat_target, pool=pool, in12=in12,
                    out1=out1, out2c=out2c, out12=out12, cls_rows=cls_rows, out3=out3, in3=in3, use_smtm=use_smtm, qrow=qrow,
                    n_fusion_closures=n_fusion_closures, txt_off_d=txt_off_d, txt_list_d=txt_list_d, txt_rows=txt_rows, ar_t=ar_t)

    def _backward_trunk(self, tr, on_other_grads_ready=None):
        """the tape behind the heads: fusion layers (, smtm pass) -> token-pool gradient -> encode -> Video-Swin; `tr` = `_forward_trunk`'s
        record with out1.g / out2c.g (/ out12.g, out3.g) filled by the heads"""
        B, O, Lv, X, Lq, n1 = tr["B"], tr["O"], tr["Lv"], tr["X"], tr["Lq"], tr["n1"]
        n_layers = self.cfg["bert_layers"]
        for _ in range((n_layers if tr["use_smtm"] else 0) + tr["n_fusion_closures"]):
            self.tape.pop()()
        g12 = tr["in12"].g
        tr["pool"].g = K.pool_grad(g12[:n1 * Lq], g12[n1 * Lq:], tr["in3"].g if tr["use_smtm"] else None, B, O, Lv, X, tr["txt_off_d"], tr["txt_list_d"])
        self.tape.pop()()                       # encode backward: text embeddings + EncVideo head -> last non-Swin gradients
        if on_other_grads_ready is not None:
            on_other_grads_ready()              # data-parallel: all-reduce of the non-Swin groups overlaps the Swin backward
        while self.tape:
            self.tape.pop()()
        self._wgrad_join()

    # -------------------------------------------------------------- full step
    def forward_backward(self, batch, negatives=None, train=True, dp_all=None, want_outputs=False, backward=True,
                         dropout=None, on_other_grads_ready=None):
        with L.pin_current():               # (one look-up of torch's current stream per step instead of one per launch, lib.stream)
            return self._forward_backward(batch, negatives, train, dp_all, want_outputs, backward, dropout, on_other_grads_ready)

    def _forward_backward(self, batch, negatives=None, train=True, dp_all=None, want_outputs=False, backward=True,
                          dropout=None, on_other_grads_ready=None):
        """One pass of the hot path.  batch: img f32 (B,T,3,H,W) UN-masked, cov u8 (B,T,h,w), txt i64 (B,X) (masked ids),
        mask i64 (B,X), ans_mtm i64 (B,X).  Returns dict of loss scalars (device f32 tensors) and optional outputs."""
        cfg, S, dev = self.cfg, self.store, self.device
        img, cov, txt, mask, ans_mtm = batch["img"], batch["cov"], batch["txt"], batch["mask"], batch["ans_mtm"]
        Hd = cfg["hidden"]
        tr = self._forward_trunk(batch, negatives, train, dp_all, dropout, backward)
        B, T, H, W, X, O, Lv, Lq, hw, n1, n2 = (tr[k] for k in ("B", "T", "H", "W", "X", "O", "Lv", "Lq", "hw", "n1", "n2"))
        train, feat_target, pool, in12, out1, out2c, out12, cls_rows = (tr[k] for k in ("train", "feat_target", "pool", "in12", "out1", "out2c", "out12", "cls_rows"))
        out3, in3, use_smtm, qrow, txt_rows = (tr[k] for k in ("out3", "in3", "use_smtm", "qrow", "txt_rows"))
        lz = torch.zeros(8, device=dev, dtype=F32)                                          # one fill: the kernels accumulate into their slot
        losses = {k: lz[i:i + 1] for i, k in enumerate(("mtm", "vtm", "mvm", "mvm_pixel", "mvm_vq", "mvm_feature", "mvm_hog", "smtm"))}
        outs = {}

        # ---- MLM head (HF BertOnlyMLMHead; main_pretrain.py:236,560) -- also the head of the smtm pass (:240,:567)
        Vv = cfg["vocab"]
        tgt_m = ans_mtm.reshape(-1).contiguous()

        def mlm_head(outv, loss):
            return self._mlm_head_fwd(K.gather_rows(outv.t, txt_rows, B * X), B * X, tgt_m, loss, backward)

        mlm_head_bwd = self._mlm_head_bwd

        h_mlm = mlm_head(out1, losses["mtm"])
        if use_smtm:
            h_smtm = mlm_head(out3, losses["smtm"])
        if want_outputs:
            outs["out_mtm"] = h_mlm["logits"][:, :Vv].reshape(B, X, Vv)
            if use_smtm:
                outs["out_smtm"] = h_smtm["logits"][:, :Vv].reshape(B, X, Vv)

        # ---- VTM head (main_pretrain.py:146-147,260-262,561)
        r_v = out2c.t                                    # [B*O, H]: the text [CLS] states of the VTM sequences
        p_fc = 0.1 if self._drop_on("vtm", train) else 0.0
        off_fc = self._next_offset(r_v.numel())
        self.last_offsets["vtm"] = off_fc
        r_vd = K.dropout(r_v, p_fc, self.seed, off_fc) if p_fc > 0 else r_v
        h_v = K.gemm(r_vd, S.b("fc.1.weight"), bias=S.p("fc.1.bias"), act=2)
        inv_temp = 1.0 / cfg["temp"]
        lg_v = K.rowdot(h_v, S.p("fc.3.weight", (2 * Hd,)), S.p("fc.3.bias"), inv_temp)           # [B*O]
        dlg_v = K.vtm_ce(lg_v, B, O, losses["vtm"])                                           # loss + its f32 gradient in one launch
        if want_outputs:
            outs["out_vtm"] = lg_v.view(B, O)
            outs["vtm_cls"] = r_v                       # the [CLS] states the VTM head reads (tests: head gradients on the same inputs)

        # ---- MVM pixel head (main_pretrain.py:178-179,420-432)
        ps = cfg["size_patch"]
        h_, w_ = H // ps, W // ps
        targets = cfg["mvm_target"]
        use_pix, use_vq, use_hog = "pixel" in targets, "vq" in targets, "hog" in targets
        vis_rows = self._cached(("vis_rows", B, T, hw, Lq), lambda: _dev_i32(
            np.concatenate([i * Lq + t * (1 + hw) + 1 + np.arange(hw) for i in range(B) for t in range(T)]), dev))
        if use_pix or use_hog:
            r_p = K.gather_rows(out1.t, vis_rows, B * T * hw)
        if use_hog:
            # MVM HOG head (main_pretrain.py:180-183,453-468): 1x1 conv H -> ps*ps + PixelShuffle(ps) -> one map per frame; L1 against
            # the data loader's HOG maps batch["hog"] (B,T,H,W) over pixels of covered patches, / (mask.sum() + 1e-5)
            Whog = S.b("decoder_hog.0.weight", (ps * ps, Hd))
            pred_h = K.gemm(r_p, Whog, bias=S.p("decoder_hog.0.bias"))
            msum_h = (cov.to(F32).sum() * float(ps * ps)).view(1)
            dpred_h = K.pixel_l1(pred_h, batch["hog"].to(F32).contiguous(), cov.reshape(-1), msum_h, losses["mvm_hog"], B, T, h_, w_, ps,
                                 channels=1, inv_div=1.0)
        if use_pix:
            Wpix = S.b("decoder_pixel.0.weight", (3 * ps * ps, Hd))
            pred = K.gemm(r_p, Wpix, bias=S.p("decoder_pixel.0.bias"))
            mask_sum = (cov.to(F32).sum() * float(3 * ps * ps)).view(1)
            dpred = K.pixel_l1(pred, img, cov.reshape(-1), mask_sum, losses["mvm_pixel"], B, T, h_, w_, ps)
            if want_outputs:
                outs["pred_pixel"] = pred
        # ---- MVM vq head (main_pretrain.py:194-209,469-502): frozen dVAE tokens as targets; decoder_vq (1x1 conv H -> 2H) +
        # PixelShuffle(4) + fc_mvm MLP + CE.  Only covered patches carry targets (ans = -1 elsewhere), so the head runs on the
        # covered patches' rows only.  PixelShuffle is folded into a row permutation of the decoder weight: output channel
        # c*16 + (i*4+j) moves to (i*4+j)*96 + c, so one GEMM row is 16 consecutive 96-channel positions.
        n_mp = 0
        if use_vq and "vq_patch_rows" in batch:
            prow, tix = batch["vq_patch_rows"], batch["vq_tok_index"]
            n_mp = int(prow.numel())
        if use_vq and n_mp > 0:
            up = ps // 8
            cq = 2 * Hd // (up * up)
            Vq = cfg.get("size_vq", 8192)
            tokens = batch.get("vq_tokens")
            if tokens is None:
                tokens = self.teacher.extract_vq_token(img.view(B * T, 3, H, W))
            tgt_q = tokens.reshape(-1)[tix].contiguous()
            perm = self._cached(("vq_perm", Hd, up), lambda: torch.from_numpy(
                (np.arange(cq)[None, :] * (up * up) + np.arange(up * up)[:, None]).reshape(-1).astype(np.int64)).to(dev))
            Wq = S.b("decoder_vq.0.weight", (2 * Hd, Hd)).index_select(0, perm)       # (tiny; plumbing)
            bq = S.p("decoder_vq.0.bias").index_select(0, perm)
            r_q = K.gather_rows(out1.t, prow, n_mp)
            y_q = K.gemm(r_q, Wq, bias=bq)                                            # [n_mp, 16*cq]
            x_q = y_q.view(n_mp * up * up, cq)
            p_q = 0.1 if self._drop_on("heads", train) else 0.0
            off_q = self._next_offset(x_q.numel())
            x_qd = K.dropout(x_q, p_q, self.seed, off_q) if p_q > 0 else x_q
            h_q = K.gemm(x_qd, S.b("fc_mvm.1.weight"), bias=S.p("fc_mvm.1.bias"), act=2)
            lg_q = K.gemm(h_q, S.b("fc_mvm.3.weight"), bias=S.p("fc_mvm.3.bias"), out_dtype=F32)
            dlg_q = K.cross_entropy(lg_q, Vq, tgt_q, losses["mvm_vq"], want_grad=backward, ld_d=Vq)
            if want_outputs:
                outs["vq_logits"], outs["vq_targets"] = lg_q, tgt_q
                outs["vq_acc"] = (lg_q.argmax(-1) == tgt_q).float().mean()
        # ---- MVM feature head (main_pretrain.py:153-174,508-545): fc_mvm (Dropout, Linear H -> 2H, ReLU, Linear 2H -> F) on every
        # non-cls visual token; targets = the frozen Swin teacher's features of the UN-masked clip; L1 over covered patches
        use_feat = "3d_feature" in targets or "2d_feature" in targets
        if use_feat:
            tgt_f = feat_target                                                          # bf16 [B*T*hw, F], no grad
            r_f = r_p if (use_pix or use_hog) else K.gather_rows(out1.t, vis_rows, B * T * hw)
            p_f = 0.1 if self._drop_on("heads", train) else 0.0
            off_f = self._next_offset(r_f.numel())
            r_fd = K.dropout(r_f, p_f, self.seed, off_f) if p_f > 0 else r_f
            h_f = K.gemm(r_fd, S.b("fc_mvm.1.weight"), bias=S.p("fc_mvm.1.bias"), act=2)
            pred_f = K.gemm(h_f, S.b("fc_mvm.3.weight"), bias=S.p("fc_mvm.3.bias"))
            cov_sum = cov.to(F32).sum().view(1)
            dpred_f = K.feature_l1(pred_f, tgt_f, cov.reshape(-1), cov_sum, losses["mvm_feature"])
            if want_outputs:
                outs["pred_feature"], outs["feature_target"] = pred_f, tgt_f
        losses["mvm"] = losses["mvm_pixel"] + losses["mvm_vq"] + losses["mvm_feature"] + losses["mvm_hog"]
        if want_outputs:
            outs["out_mvm"] = out1.t.view(B, Lq, Hd)[:, :Lv]
        if not backward:
            self.tape = []
            return losses, outs

        # =============================== backward ===============================
        # heads -> gradients of the two encoder outputs
        use_vis = use_pix or use_feat or use_hog
        npx = B * T * hw if use_vis else 0
        dcat = torch.empty((npx + B * X, Hd), device=dev, dtype=BF16)              # [visual-token rows ; mlm rows]
        vis_filled = False
        if use_pix:
            self._linear_bwd(dpred, r_p, None, None, w=Wpix, gw=S.g("decoder_pixel.0.weight", (3 * ps * ps, Hd)), gb=S.g("decoder_pixel.0.bias"),
                             dx_kw=dict(out=dcat[:npx]), wT=S.bt("decoder_pixel.0.weight"))
            vis_filled = True
        if use_hog:
            dr_h = self._linear_bwd(dpred_h, r_p, None, None, w=Whog, gw=S.g("decoder_hog.0.weight", (ps * ps, Hd)), gb=S.g("decoder_hog.0.bias"),
                                    dx_kw=None if vis_filled else dict(out=dcat[:npx]), wT=S.bt("decoder_hog.0.weight"))
            if vis_filled:
                K.add_bf16(dcat[:npx], dr_h, out=dcat[:npx])
            vis_filled = True
        if use_feat:
            dh_f = self._linear_bwd(dpred_f, h_f, "fc_mvm.3.weight", "fc_mvm.3.bias", dx_kw=dict(act=4, aux=h_f))   # ReLU' folded into the dgrad
            dr_f = self._linear_bwd(dh_f, r_fd, "fc_mvm.1.weight", "fc_mvm.1.bias")
            if p_f > 0:
                dr_f = K.dropout(dr_f, p_f, self.seed, off_f)
            if vis_filled:
                K.add_bf16(dcat[:npx], dr_f, out=dcat[:npx])
            else:
                dcat[:npx].copy_(dr_f)
        mlm_head_bwd(h_mlm, dcat[npx:])
        if use_smtm:
            d3 = torch.empty((B * X, Hd), device=dev, dtype=BF16)
            mlm_head_bwd(h_smtm, d3)
            inv3 = self._cached(("inv3", B, Lq, Lv, X), lambda: self._inverse_rows(B * Lq, [txt_rows]))
            out3.g = K.gather_rows(d3, inv3, B * Lq)
        inv1 = self._cached(("inv1", B, T, hw, Lq, Lv, X, use_vis), lambda: self._inverse_rows(B * Lq, [vis_rows, txt_rows] if use_vis else [txt_rows]))
        K.gather_rows(dcat, inv1, B * Lq, out=out1.g)
        if use_vq and n_mp > 0:
            dh_q = self._linear_bwd(dlg_q, h_q, "fc_mvm.3.weight", "fc_mvm.3.bias", dx_kw=dict(act=4, aux=h_q))     # ReLU' folded into the dgrad
            dx_q = self._linear_bwd(dh_q, x_qd, "fc_mvm.1.weight", "fc_mvm.1.bias")
            if p_q > 0:
                dx_q = K.dropout(dx_q, p_q, self.seed, off_q)
            gWq = torch.zeros((2 * Hd, Hd), device=dev, dtype=F32)
            gbq = torch.zeros(2 * Hd, device=dev, dtype=F32)
            dr_q = self._linear_bwd(dx_q.view(n_mp, 2 * Hd), r_q, None, None, w=Wq, gw=gWq, gb=gbq, wsync=True)    # (gWq / gbq are read right below)
            S.g("decoder_vq.0.weight", (2 * Hd, Hd)).index_add_(0, perm, gWq)          # undo the PixelShuffle row permutation
            S.g("decoder_vq.0.bias").index_add_(0, perm, gbq)
            out1.g.index_add_(0, prow.long(), dr_q)                                     # covered-patch rows (unique) of the fusion output
        # VTM
        # d(vtm)/d(logits) of the (B,O) matrix in f32: positives and negatives of a clip nearly cancel, a bf16-rounded softmax
        # would add rounding noise of the size of the signal (the reference's autocast runs cross_entropy in fp32 as well)
        dh_v = K.rowdot_bwd(h_v, S.p("fc.3.weight", (2 * Hd,)), dlg_v, inv_temp, S.g("fc.3.weight", (2 * Hd,)), S.g("fc.3.bias"), relu_mask=True)
        dr_v = self._linear_bwd(dh_v, r_vd, "fc.1.weight", "fc.1.bias")
        if p_fc > 0:
            dr_v = K.dropout(dr_v, p_fc, self.seed, off_fc)
        out2c.g = dr_v
        if not qrow:
            inv2 = self._cached(("inv2", B * O, Lq, Lv), lambda: self._inverse_rows(B * O * Lq, [cls_rows]))
            K.gather_rows(dr_v, inv2, B * O * Lq, out=out12.g[n1 * Lq:])

        # encoders (tape holds: encode, the merged pass' layers (, the smtm pass' layers)) -> run them back, then gather into the pool
        self._backward_trunk(tr, on_other_grads_ready)
        return losses, outs

    # -------------------------------------------------------------- the step OPEN at the reference's model outputs (autograd interop)
    def forward_open(self, batch, negatives=None, train=True, dp_all=None, dropout=None):
        with L.pin_current():
            return self._forward_open(batch, negatives, train, dp_all, dropout)

    def backward_open(self, tr, d_mtm, d_mvm, d_vtm, d_smtm=None, on_other_grads_ready=None):
        with L.pin_current():
            return self._backward_open(tr, d_mtm, d_mvm, d_vtm, d_smtm, on_other_grads_ready)

    def _forward_open(self, batch, negatives=None, train=True, dp_all=None, dropout=None):
        """VIOLET_Pretrain.forward as the REFERENCE defines it (main_pretrain.py:226-267): the model ends at `out_mtm` (MLM logits
        (B, X, V) f32), `out_mvm` (the fusion encoder's visual-token states (B, T*(1+hw), H) bf16) and `out_vtm` ((B, O) pair scores / temp,
        f32); the losses -- and the MVM decoders, which the reference's agent applies itself (main_pretrain.py:420-432) -- are the
        caller's.  The tape stays armed: `backward_open` takes the three output gradients.  model.VIOLET_Pretrain.forward wraps the pair
        in a torch.autograd.Function."""
        S, dev, Hd = self.store, self.device, self.cfg["hidden"]
        tr = self._forward_trunk(batch, negatives, train, dp_all, dropout, True)
        B, X, O, Lv, Lq = tr["B"], tr["X"], tr["O"], tr["Lv"], tr["Lq"]
        out1, out2c = tr["out1"], tr["out2c"]
        Vv, Vpad, Nlog = self._mlm_dims()

        def head_open(outv):                    # HF BertOnlyMLMHead without the loss
            pm = "fc_mtm.predictions."
            rows = K.gather_rows(outv.t, tr["txt_rows"], B * X)
            u_ = torch.empty((B * X, Hd), device=dev, dtype=BF16)
            t_ = K.gemm(rows, S.b(pm + "transform.dense.weight"), bias=S.p(pm + "transform.dense.bias"), act=1, out_preact=u_)
            tn_, mean_, rstd_ = K.layernorm_fwd(t_, S.p(pm + "transform.LayerNorm.weight"), S.p(pm + "transform.LayerNorm.bias"), CFG.BERT["eps"])
            lg_ = torch.empty((B * X, Vpad), device=dev, dtype=F32)
            K.gemm(tn_, S.b(pm + "decoder.weight"), N=Nlog, bias=S.p(pm + "bias"), out=lg_)
            return dict(r=rows, u=u_, t=t_, tn=tn_, mean=mean_, rstd=rstd_, logits=lg_, dlog=None, n=B * X)
        tr["h_mlm"] = head_open(out1)
        tr["h_smtm"] = head_open(tr["out3"]) if tr["use_smtm"] else None
        # VTM head (main_pretrain.py:146-147,260-262)
        r_v = out2c.t
        p_fc = 0.1 if self._drop_on("vtm", tr["train"]) else 0.0
        off_fc = self._next_offset(r_v.numel())
        self.last_offsets["vtm"] = off_fc
        r_vd = K.dropout(r_v, p_fc, self.seed, off_fc) if p_fc > 0 else r_v
        h_v = K.gemm(r_vd, S.b("fc.1.weight"), bias=S.p("fc.1.bias"), act=2)
        inv_temp = 1.0 / self.cfg["temp"]
        lg_v = K.rowdot(h_v, S.p("fc.3.weight", (2 * Hd,)), S.p("fc.3.bias"), inv_temp)
        tr.update(r_vd=r_vd, h_v=h_v, p_fc=p_fc, off_fc=off_fc, inv_temp=inv_temp)
        outs = {"out_mtm": tr["h_mlm"]["logits"][:, :Vv].reshape(B, X, Vv), "out_mvm": out1.t.view(B, Lq, Hd)[:, :Lv], "out_vtm": lg_v.view(B, O),
                "out_smtm": tr["h_smtm"]["logits"][:, :Vv].reshape(B, X, Vv) if tr["use_smtm"] else None}
        return outs, tr

    def _backward_open(self, tr, d_mtm, d_mvm, d_vtm, d_smtm=None, on_other_grads_ready=None):
        """the backward of `forward_open` from the gradients of its outputs (None = that output did not reach the loss).  Parameter
        gradients are ACCUMULATED into the gradient arena -- the `.grad` views of model.parameters() -- like autograd's AccumulateGrad."""
        S, dev, Hd = self.store, self.device, self.cfg["hidden"]
        B, X, O, Lv, Lq, n1 = tr["B"], tr["X"], tr["O"], tr["Lv"], tr["Lq"], tr["n1"]
        out1, out2c = tr["out1"], tr["out2c"]
        Vv, Vpad, _ = self._mlm_dims()

        def dlog_of(d):                          # (B, X, V) -> bf16 [B*X, Vpad] with zero pad columns (what vmvm_cross_entropy hands the head backward)
            t = torch.zeros((B * X, Vpad), device=dev, dtype=BF16)
            t[:, :Vv].copy_(d.reshape(B * X, Vv))
            return t
        g1 = out1.g.view(B, Lq, Hd)
        if d_mvm is None:
            g1[:, :Lv].zero_()
        else:
            g1[:, :Lv].copy_(d_mvm.reshape(B, Lv, Hd))
        if d_mtm is None:
            g1[:, Lv:].zero_()
        else:
            tr["h_mlm"]["dlog"] = dlog_of(d_mtm)
            g1[:, Lv:].copy_(self._mlm_head_bwd(tr["h_mlm"]).view(B, X, Hd))
        if tr["use_smtm"]:
            g3 = torch.zeros((B, Lq, Hd), device=dev, dtype=BF16)
            if d_smtm is not None:
                tr["h_smtm"]["dlog"] = dlog_of(d_smtm)
                g3[:, Lv:].copy_(self._mlm_head_bwd(tr["h_smtm"]).view(B, X, Hd))
            tr["out3"].g = g3.view(B * Lq, Hd)
        if d_vtm is None:
            dr_v = torch.zeros_like(out2c.t)
        else:
            dlg_v = d_vtm.reshape(-1).to(F32).contiguous()
            dh_v = K.rowdot_bwd(tr["h_v"], S.p("fc.3.weight", (2 * Hd,)), dlg_v, tr["inv_temp"], S.g("fc.3.weight", (2 * Hd,)), S.g("fc.3.bias"), relu_mask=True)
            dr_v = self._linear_bwd(dh_v, tr["r_vd"], "fc.1.weight", "fc.1.bias")
            if tr["p_fc"] > 0:
                dr_v = K.dropout(dr_v, tr["p_fc"], self.seed, tr["off_fc"])
        out2c.g = dr_v
        if not tr["qrow"]:
            inv2 = self._cached(("inv2", B * O, Lq, Lv), lambda: self._inverse_rows(B * O * Lq, [tr["cls_rows"]]))
            K.gather_rows(dr_v, inv2, B * O * Lq, out=tr["out12"].g[n1 * Lq:])
        self._backward_trunk(tr, on_other_grads_ready)
