"""Downstream passes of `VioletEngine` (engine.py): retrieval, open-ended QA, multiple-choice / MLM-generation QA.
Methods of the engine class."""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import config as CFG
from . import kernels as K
from . import swin_index as SI
from .store import BF16, F32, V, DropScale, _acc, _gout, _h2d, _dev_i32


class DownstreamMixin:
    # -------------------------------------------------------------- downstream: text-to-video retrieval
    # -------------------------------------------------------------- scaffolding shared by the downstream passes
    def _begin_pass(self, img, txt, train, dp_all):
        """new tape, DropPath draw, encoders -> (pool, Lv, X, Lq): one token pool of B clips' visual rows, then the text sequences' rows"""
        self.tape = []
        if train and dp_all is None:
            dp_all = self.sample_drop_path(img.shape[0])
        pool, Lv, _ = self.encode(img, None, txt, dp_all, train)
        X = txt.shape[1]
        return pool, Lv, X, Lv + X

    def _fuse_pairs(self, pool, key, pairs, B, mask, Lv, X, train, cls_only):
        """One fusion pass over the (clip i, text sequence j) pairs: rows gathered from the token pool, key mask = ones over the visual part
        + the text's mask.  cls_only: the head reads the text [CLS] state alone, so the last layer runs for that query row only
        (`go_cross(qrow_split=(0, Lv))`).  -> dict(out = V of the full output or of the [CLS] rows, inn, idx, n_closures)."""
        dev = self.device
        ar_v, ar_t = np.arange(Lv), np.arange(X)
        idx_d = self._cached(key, lambda: _dev_i32(np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + j * X + ar_t]) for i, j in pairs]), dev))
        km_txt = (mask != 0).to(torch.uint8)
        tj = [j for _, j in pairs]
        if tj != list(range(km_txt.shape[0])):
            km_txt = km_txt[_h2d(torch.tensor(tj), dev)]
        km = torch.cat([torch.ones(len(pairs), Lv, dtype=torch.uint8, device=dev), km_txt], 1).contiguous()
        ntape = len(self.tape)
        if cls_only:
            (_, out), inn, _ = self.go_cross(pool, idx_d, km, len(pairs), Lv + X, train, qrow_split=(0, Lv))
        else:
            out, inn, _ = self.go_cross(pool, idx_d, km, len(pairs), Lv + X, train)
        return dict(out=out, inn=inn, idx=idx_d, n_closures=len(self.tape) - ntape)

    def _cls_hidden(self, r_v, train):
        """first half of the reference's `fc` heads on the [CLS] states (Dropout(0.1), Linear H -> 2H, ReLU; main_retrieval.py:54-56,
        main_qaoe.py:42-47) -> state for `_cls_hidden_bwd`"""
        S = self.store
        p_fc = 0.1 if train else 0.0
        off_fc = self._next_offset(r_v.numel())
        r_vd = K.dropout(r_v, p_fc, self.seed, off_fc) if p_fc > 0 else r_v
        return dict(h=K.gemm(r_vd, S.b("fc.1.weight"), bias=S.p("fc.1.bias"), act=2), x=r_vd, p=p_fc, off=off_fc)

    def _cls_hidden_bwd(self, st, dh):
        """dh = d(loss)/d(hidden) with the ReLU mask already applied -> d(loss)/d([CLS] states); accumulates fc.1's gradients"""
        dr = self._linear_bwd(dh, st["x"], "fc.1.weight", "fc.1.bias")
        return K.dropout(dr, st["p"], self.seed, st["off"]) if st["p"] > 0 else dr

    def _finish_pass(self, fz, pool, dout):
        """backward of the fusion pass (its closures are the newest on the tape), the token pool's gradient through the gather, the encoders"""
        fz["out"].g = dout
        for _ in range(fz["n_closures"]):
            self.tape.pop()()
        dpool = torch.zeros((pool.t.shape[0], pool.t.shape[1]), device=self.device, dtype=F32)
        K.scatter_add_rows(fz["inn"].g, fz["idx"], dpool)
        pool.g = K.cast_bf16(dpool)
        while self.tape:
            self.tape.pop()()
        self._wgrad_join()

    # -------------------------------------------------------------- downstream: text-video retrieval
    def retrieval_forward_backward(self, img, txt, mask, train=True, backward=True, dp_all=None, dlogits=None):
        """VIOLET_Retrieval.forward + NormSoftmaxLoss (main_retrieval.py:63-85, agent.py:34-50): every (video i, text j) pair of the
        batch goes through the fusion encoder (B*B sequences gathered from one token pool), the `fc` head reads the text [CLS]
        state, and the loss is the symmetric cross entropy of the (B,B) score matrix / temp with the diagonal as targets.
        Returns (loss f32[1], scores (B,B) f32).  `dlogits` (B,B) f32 replaces d(loss)/d(scores / temp) in the backward (tests:
        the loss gradient itself is a difference of nearly equal terms whenever the scores are close, a poor probe of the
        backward path)."""
        cfg, S, dev = self.cfg, self.store, self.device
        B, Hd = img.shape[0], cfg["hidden"]
        pool, Lv, X, Lq = self._begin_pass(img, txt, train, dp_all)
        fz = self._fuse_pairs(pool, ("ret_idx", B, Lv, X), [(i, j) for i in range(B) for j in range(B)], B, mask, Lv, X, train, cls_only=True)   # (:76 reads [CLS] only)
        st = self._cls_hidden(fz["out"].t, train)
        inv_temp = 1.0 / cfg["temp"]
        lg = K.rowdot(st["h"], S.p("fc.3.weight", (2 * Hd,)), S.p("fc.3.bias"), inv_temp).view(B, B)
        tgt = torch.arange(B, dtype=torch.int64, device=dev)
        loss = torch.zeros(1, device=dev, dtype=F32)
        K.cross_entropy(lg.contiguous(), B, tgt, loss, want_grad=False)                              # -mean diag log_softmax over rows
        K.cross_entropy(lg.t().contiguous(), B, tgt, loss, want_grad=False)                          # ... and over columns
        scores = lg * cfg["temp"]
        if not backward:
            self.tape = []
            return loss, scores
        # d(loss)/d(logits) of the (B,B) matrix in f32: the useful part of this gradient is what is left after the rows / columns
        # cancel, a bf16-rounded softmax would bury it (tiny matrix: plumbing)
        eye = torch.eye(B, device=dev, dtype=F32)
        dlg = ((torch.softmax(lg, 1) - eye) / B + (torch.softmax(lg, 0) - eye) / B).reshape(-1).contiguous()
        if dlogits is not None:
            dlg = dlogits.to(dev, F32).reshape(-1).contiguous()
        dh = K.rowdot_bwd(st["h"], S.p("fc.3.weight", (2 * Hd,)), dlg, inv_temp, S.g("fc.3.weight", (2 * Hd,)), S.g("fc.3.bias"), relu_mask=True)
        self._finish_pass(fz, pool, self._cls_hidden_bwd(st, dh))
        return loss, scores

    # -------------------------------------------------------------- downstream: open-ended video QA
    def qaoe_forward_backward(self, img, txt, mask, ans, train=True, backward=True, dp_all=None):
        """VIOLET_QAOE.forward + CrossEntropyLoss(ignore_index=-1) (main_qaoe.py:49-58,72-76, agent.py:57): one (video, question)
        fusion pass, `fc` (Dropout, Linear, ReLU, Linear -> answer vocabulary) on the text [CLS] state.  Returns (loss f32[1],
        logits (B, size_vocab) f32)."""
        cfg, S, dev = self.cfg, self.store, self.device
        B, Hd, NV = img.shape[0], cfg["hidden"], int(cfg["size_vocab"])
        pool, Lv, X, Lq = self._begin_pass(img, txt, train, dp_all)
        fz = self._fuse_pairs(pool, ("qa_idx", B, Lv, X), [(i, i) for i in range(B)], B, mask, Lv, X, train, cls_only=True)
        st = self._cls_hidden(fz["out"].t, train)
        h_v = st["h"]
        NVp = -(-NV // 8) * 8
        logits = torch.zeros((B, NVp), device=dev, dtype=F32)
        K.gemm(h_v, S.b("fc.3.weight"), N=-(-NV // 4) * 4, bias=S.p("fc.3.bias"), out=logits)
        loss = torch.zeros(1, device=dev, dtype=F32)
        dlog = K.cross_entropy(logits, NV, ans.to(dev).reshape(-1).contiguous(), loss, want_grad=backward, ld_d=NVp)
        if not backward:
            self.tape = []
            return loss, logits[:, :NV]
        K.colsum(dlog, S.g("fc.3.bias"), accumulate=True, M=B, N=NVp)
        K.gemm(dlog, h_v, a_kmajor=False, b_kmajor=False, M=NV, N=2 * Hd, K=B, out=S.g("fc.3.weight"), accumulate=True)
        dh = K.gemm(dlog, S.b("fc.3.weight"), b_kmajor=False, M=B, N=2 * Hd, K=NV, act=4, aux=h_v)            # ReLU' folded in
        self._finish_pass(fz, pool, self._cls_hidden_bwd(st, dh))
        return loss, logits[:, :NV]

    # -------------------------------------------------------------- downstream: multiple-choice video QA, MLM-head form
    def qamc_mlm_forward_backward(self, img, txt, mask, mask_ans, train=True, backward=True, dp_all=None):
        """VIOLET_QAMC_MLM_Head.forward + Agent_QAMC_MLM_Head.step (main_qamc_tsv_mlm_head.py:76-109): txt / mask / mask_ans are
        (B, O, X) -- one "question + option_o + [MASK]" sequence per option, labelled true / false token id at the [MASK] position
        and -1 elsewhere.  The clip's video tokens are shared by its O sequences (B*O sequences gathered from one token pool, as the
        retrieval head's pairs), the shared MLM head (`fc_mtm`) reads every text position, cross entropy with ignore_index -1.
        Returns (loss f32[1], logits (B*O*X, vocab) f32 view).  Task token / prompt (`enable_task_token`, `enable_prompt`) are off."""
        cfg, dev = self.cfg, self.device
        B, O, Vv = img.shape[0], int(txt.shape[1]), cfg["vocab"]
        n_seq = B * O
        txt2, mask2 = txt.reshape(n_seq, -1).contiguous(), mask.reshape(n_seq, -1).contiguous()
        pool, Lv, X, Lq = self._begin_pass(img, txt2, train, dp_all)          # pool rows: B*Lv visual, then (B*O)*X text
        fz = self._fuse_pairs(pool, ("qamc_idx", B, O, Lv, X), [(s_ // O, s_) for s_ in range(n_seq)], B, mask2, Lv, X, train, cls_only=False)
        txt_rows = self._cached(("qamc_txt_rows", n_seq, Lv, X), lambda: _dev_i32(np.concatenate([i * Lq + Lv + np.arange(X) for i in range(n_seq)]), dev))
        nr = n_seq * X
        loss = torch.zeros(1, device=dev, dtype=F32)
        hd = self._mlm_head_fwd(K.gather_rows(fz["out"].t, txt_rows, nr), nr, mask_ans.to(dev).reshape(-1).contiguous(), loss, backward)
        lg_ = hd["logits"]
        if not backward:
            self.tape = []
            return loss, lg_[:, :Vv]
        dtxt = self._mlm_head_bwd(hd)
        inv = self._cached(("qamc_inv", n_seq, Lq, Lv, X), lambda: self._inverse_rows(n_seq * Lq, [txt_rows]))
        self._finish_pass(fz, pool, K.gather_rows(dtxt, inv, n_seq * Lq))
        return loss, lg_[:, :Vv]
