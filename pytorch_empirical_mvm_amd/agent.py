"""Training-agent surface of the reference (agent.py Agent_Base, main_pretrain.py:269-619 Agent_Pretrain) over the
HIP engine: masking -> prepare_batch -> step (forward, losses, backward, all-reduce, clip, AdamW, LR schedule)."""
import json
import os
import random

import numpy as np
import torch

from . import config as CFG
from . import dist as D
from . import kernels as K
from . import lib as L


class Agent_Pretrain:
    def __init__(self, args, model):
        self.args, self.model = args, model
        self.engine = model.engine
        self.patch_size = model.patch_size
        self.cls_token_id, self.sep_token_id = CFG.TOKENS["cls"], CFG.TOKENS["sep"]
        self.pad_token_id, self.mask_token_id, self.unk_token_id = CFG.TOKENS["pad"], CFG.TOKENS["mask"], CFG.TOKENS["unk"]
        self.global_step = 0
        self.sched_step = 0                 # WarmupLinearLR.last_epoch: the ctor already performed step 0 (agent.py:13-32)
        self.opt_step = 0
        self.world_size, self.rank = 1, 0
        self.comm = None
        self.log = {d: {} for d in (args.dataset if isinstance(args.dataset, (list, tuple)) else [args.dataset])}
        self._sumsq = torch.zeros(1, device=self.engine.device, dtype=torch.float32)

    # ------------------------------------------------------------------ distributed (agent.py:195-201, utils/dist.py:20-75)
    def prepare_dist_model(self):
        """Replaces DDP(find_unused_parameters=True) / DeepSpeed ZeRO-1: plain data parallel, one process per GPU,
        gradient all-reduce (RCCL over xGMI) of the flat arena, non-Swin half overlapped with the Swin backward."""
        if D.is_initialized():
            self.world_size, self.rank = D.world_size(), D.rank()
            self.comm = D.GradReducer(self.engine.store, self.engine.device)
            if getattr(self.engine, "wstream", None) is not None:
                self.comm.wait_streams.append(self.engine.wstream)      # weight gradients are produced on the engine's second stream
            # the autograd-driven step (model(batch) ... loss.backward(), model._OpenStep) starts the same two exchange phases from inside
            # its backward node: the non-Swin groups behind the fusion backward, the Swin tail behind stage n-2
            self.model._grad_hook = self.comm.reduce_other
            self.model._mid_hook = self.comm.reduce_other_early
            self.model._tail_hook = self.comm.reduce_swin_tail     # (installed on the engine only for the duration of that backward: a bare
            #                                                         engine.forward_backward() -- tools/dp_check.py's per-rank reference -- must not reduce)
            D.broadcast_(self.engine.store.flat)           # identical replicas (DDP broadcasts rank-0 parameters at wrap time)
            self.engine.store.refresh_shadow()
            # DDP also broadcasts the frozen teachers' parameters (they are sub-modules of the wrapped model): without this, ranks
            # seeded differently would regress onto different random teachers
            ft = getattr(self.model, "feature_model", None)
            if ft is not None:
                D.broadcast_(ft.eng.store.flat)
                ft.eng.store.refresh_shadow()
            dl = getattr(self.model, "dalle", None)
            if dl is not None:
                for k in sorted(dl.w):
                    D.broadcast_(dl.w[k])
                dl._refresh()

    def reduce_mean(self, v):
        """agent.py:118-125"""
        if self.world_size < 2:
            return v
        t = torch.tensor([float(v)], device=self.engine.device)
        D.all_reduce_(t)
        return float(t.item()) / self.world_size

    # ------------------------------------------------------------------ masking (main_pretrain.py:276-372)
    @torch.no_grad()
    def masking(self, img, txt, mask, vq=None, p_mask=0.15, materialize=False):
        """Same distributions and RNG sources as the reference (`random.choice`, `torch.rand`, `np.random.randint`);
        returns the reference's keys plus `cov` (B,T,h,w) u8, the patch cover the kernels consume.  `img` / `mvm_mask`
        are only materialised on request -- the HIP path applies the cover while reading the clip."""
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        h, w = H // self.patch_size, W // self.patch_size
        txt = txt.clone()
        spc = (txt == self.cls_token_id) | (txt == self.sep_token_id) | (txt == self.pad_token_id) | (txt == self.mask_token_id)
        ans_mtm = torch.full_like(txt, -1)
        cov = torch.zeros(B, T, h, w, dtype=torch.uint8)
        Lv = (1 + h * w) * T
        # `vq` (dataset-supplied token map (B, T*(1+hw)), -1 at the per-frame cls slots): special visual positions and the MVM answers
        spc_vis = None if vq is None else (vq.cpu() == -1)
        ans_mvm = torch.full((B, T * (1 + h * w)), -1, dtype=torch.long)
        failed_masking = False               # sticky across the batch, as in the reference (:300, :342-343)
        if p_mask > 0:
            for i in range(B):
                mask_type = random.choice(self.args.pretrain_masks)
                sel, cov_i = None, torch.zeros(T, h, w, dtype=torch.uint8)
                if mask_type == "bm":                                    # :304-318
                    sel = (~spc[i].cpu()) & (torch.rand(X) < p_mask)
                    for _ in range(T):
                        t = np.random.randint(1, T) if T > 1 else 1
                        hh, ww = np.random.randint(1, h * 2 // 3), np.random.randint(1, w * 2 // 3)
                        t1, h1, w1 = np.random.randint(0, T - t + 1), np.random.randint(0, h - hh + 1), np.random.randint(0, w - ww + 1)
                        cov_i[t1:t1 + t, h1:h1 + hh, w1:w1 + ww] = 1
                if mask_type == "am":
                    # attention-guided masking (:320-343): positions drawn without replacement with the layer- and head-averaged
                    # attention mass each position receives; special positions (per-frame cls, [CLS]/[SEP]/pad/[MASK]) excluded.
                    # The weights come from the HIP path (attention kernels with the column-sum output, one pass for the whole
                    # batch), the draw is torch.multinomial on the CPU generator.
                    # (:321-324: get_att runs on the whole batch as it stands NOW -- text and clips of the earlier samples already masked)
                    a = self.model.get_att(img, txt, mask, cov=cov)[1][i].detach().float().cpu().clone()
                    spc_v = torch.tensor(sum([[True] + [False] * (h * w) for _ in range(T)], [])) if spc_vis is None else spc_vis[i]
                    a[torch.cat([spc_v, spc[i].cpu()])] = 0.0
                    try:
                        pos = torch.multinomial(a, int((Lv + X) * p_mask)).numpy()
                        sel = torch.zeros(X, dtype=torch.bool)
                        for p_ in pos:
                            if p_ < Lv:
                                i_t, q = p_ // (1 + h * w), p_ % (1 + h * w) - 1
                                cov_i[i_t, q // w, q % w] = 1
                            else:
                                sel[p_ - Lv] = True
                        failed_masking = not bool(sel.any())
                    except Exception:
                        failed_masking = True
                if mask_type == "rm" or failed_masking:                  # :344-352 (also the fallback of a failed 'am' draw)
                    sel = (~spc[i].cpu()) & (torch.rand(X) < p_mask)
                    r = torch.rand((1 + h * w) * T) < p_mask
                    if spc_vis is not None:
                        r = r & ~spc_vis[i]
                    cov_i = r.view(T, 1 + h * w)[:, 1:].reshape(T, h, w).to(torch.uint8)
                cov[i] = cov_i
                if vq is not None:                                       # :356-360 curr_ans_mvm[p] = vq[i][p] at covered positions
                    pos_c = torch.cat([torch.zeros(T, 1, dtype=torch.bool), cov_i.view(T, h * w).bool()], 1).flatten()
                    ans_mvm[i] = torch.where(pos_c, vq[i].cpu().long(), ans_mvm[i])
                sel = sel.to(txt.device)
                ans_mtm[i] = torch.where(sel, txt[i], ans_mtm[i])
                txt[i] = torch.where(sel, torch.full_like(txt[i], self.mask_token_id), txt[i])
        out = {"txt": txt, "mask": mask, "ans_mtm": ans_mtm, "ans_mvm": ans_mvm, "cov": cov, "unmask_img": img}
        if "vq" in self.args.mvm_target:
            out.update(self.vq_index(cov))
        if materialize:
            full = cov.to(img.dtype).to(img.device)[:, :, None, :, None, :, None].expand(-1, -1, 3, -1, 32, -1, 32).reshape(B, T, 3, H, W)
            out["mvm_mask"] = full
            out["img"] = img * (1.0 - full)
        else:
            out["img"] = img
        return out

    @torch.no_grad()
    def masking_device(self, img, txt, mask, vq=None, p_mask=0.15, generator=None, draws=None):
        """masking() on the GPU (libvmvm `vmvm_masking`) -- no host loops, no host sync for the pixel target.
        `img`, `txt`, `mask` are device tensors; the uniform draws come from torch.rand on the device (`generator` = the
        rank's CUDA generator) or are passed explicitly as `draws` = (u_type, u_txt, u_rm, u_bm) (parity tests: the CPU
        oracle's `masking_from_uniform` consumes the same arrays).  Same distributions as the reference ('rm' Bernoulli
        field / 'bm' cuboids / 15 % [MASK]); the random STREAM necessarily differs from the host RNGs of masking()."""
        dev = self.engine.device
        B, T, _, H, W = img.shape
        X = txt.shape[1]
        h, w = H // self.patch_size, W // self.patch_size
        kinds = {"rm": 0, "bm": 1}
        for m in self.args.pretrain_masks:
            if m not in kinds:
                raise NotImplementedError(f"mask type '{m}' is outside the accelerated path (SURVEY 8f.2)")
        types = torch.tensor([kinds[m] for m in self.args.pretrain_masks], dtype=torch.int32)
        if draws is None:
            n = [B, B * X, B * T * (1 + h * w), B * T * 6]
            u = torch.rand(sum(n), device=dev, generator=generator)
            draws = torch.split(u, n)
        u_type, u_txt, u_rm, u_bm = (d.to(dev, torch.float32).contiguous() for d in draws)
        txt = txt.to(dev).clone()
        if p_mask > 0:
            ans_mtm, cov = K.masking(txt, u_type, u_txt, u_rm, u_bm, types, T, h, w, p_mask, CFG.TOKENS)
        else:
            ans_mtm = torch.full_like(txt, -1)
            cov = torch.zeros((B, T, h, w), device=dev, dtype=torch.uint8)
        out = {"img": img, "unmask_img": img, "txt": txt, "mask": mask, "ans_mtm": ans_mtm, "cov": cov,
               "ans_mvm": torch.full((B, T * (1 + h * w)), -1, dtype=torch.long, device=dev)}
        if "vq" in self.args.mvm_target:
            out.update(self.vq_index(cov))              # index lists built on the device (one count read-back)
        return out

    def vq_index(self, cov):
        """Index lists for the vq head (main_pretrain.py:485-488): a vq position is a target iff its 32x32 patch is covered, so the
        head only runs on covered patches.  patch_rows: rows of the MVM output (B*(T*(1+hw)+X) layout of the fusion output);
        tok_index: for each covered patch its 16 token positions (i-major, j-minor) in the (B*T, 28, 28) map.  Built where `cov`
        lives: on the device for masking_device (torch.nonzero reads the count back -- the one host sync of that path; the step's
        GEMM sizes depend on it), in numpy for the host masking path."""
        B, T, h, w = cov.shape
        up = self.patch_size // 8
        vs_h, vs_w = h * up, w * up
        Lq = T * (1 + h * w) + int(self.args.size_txt)
        if cov.is_cuda:
            nz = torch.nonzero(cov)                                       # row-major order = numpy's
            b, t, hh, ww = nz[:, 0], nz[:, 1], nz[:, 2], nz[:, 3]
            rows = (b * Lq + t * (1 + h * w) + 1 + hh * w + ww).to(torch.int32)
            ar = torch.arange(up, device=cov.device)
            ii, jj = ar.repeat_interleave(up).view(1, -1), ar.repeat(up).view(1, -1)
            tok = ((b * T + t)[:, None] * vs_h + (hh[:, None] * up + ii)) * vs_w + ww[:, None] * up + jj
            return {"vq_patch_rows": rows.contiguous(), "vq_tok_index": tok.reshape(-1).contiguous()}
        b, t, hh, ww = np.nonzero(cov.cpu().numpy())
        rows = b * Lq + t * (1 + h * w) + 1 + hh * w + ww
        ii, jj = np.meshgrid(np.arange(up), np.arange(up), indexing="ij")
        tok = ((b * T + t)[:, None] * vs_h + (hh[:, None] * up + ii.reshape(1, -1))) * vs_w + ww[:, None] * up + jj.reshape(1, -1)
        return {"vq_patch_rows": torch.from_numpy(rows.astype(np.int32)), "vq_tok_index": torch.from_numpy(tok.reshape(-1).astype(np.int64))}

    def prepare_batch(self, batch):
        """agent.py:156-159 (move_to_cuda)"""
        dev = self.engine.device
        return {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}

    # ------------------------------------------------------------------ one optimizer step (main_pretrain.py:555-573, agent.py:181-193)
    def current_lrs(self):
        f = CFG.lr_factor(self.sched_step, self.args.max_iter)
        base = self.args.lr
        mul = self.args.vis_backbone_lr_mul
        lrs = [max(1e-8, base * mul * f), max(1e-8, base * f)]
        return [lrs[0], lrs[1], lrs[0], lrs[1]]

    def step(self, batch, is_train=True, negatives=None, dp_all=None, sync=True):
        eng = self.engine
        src_img = batch["unmask_img"] if "unmask_img" in batch else batch["img"]
        b = dict(img=src_img.to(eng.device, torch.float32).contiguous(), cov=batch["cov"].to(eng.device).contiguous(),
                 txt=batch["txt"].to(eng.device).contiguous(), mask=batch["mask"].to(eng.device).contiguous(),
                 ans_mtm=batch["ans_mtm"].to(eng.device).contiguous())
        for k in ("vq_patch_rows", "vq_tok_index", "vq_tokens", "hog"):
            if k in batch and batch[k] is not None:
                b[k] = batch[k].to(eng.device).contiguous()
        if is_train:
            hook = self.comm.reduce_other if self.comm is not None else None
            eng.on_swin_tail_ready = self.comm.reduce_swin_tail if self.comm is not None else None
            eng.on_fusion_mid_ready = self.comm.reduce_other_early if self.comm is not None else None
            try:
                losses, _ = eng.forward_backward(b, negatives=negatives, train=True, dp_all=dp_all, on_other_grads_ready=hook)
            finally:
                eng.on_fusion_mid_ready = None       # (a bare engine.forward_backward() afterwards must not start an exchange)
            self.backward_step()
            self.global_step += 1
        else:
            # evaluation branch of the reference (main_pretrain.py:574-586): accuracies for MLM / VTM, masked L1 for MVM
            losses, outs = eng.forward_backward(b, negatives=negatives, train=False, want_outputs=True, backward=False)
            pred_m = outs["out_mtm"].argmax(-1)
            ans_m = b["ans_mtm"]
            nm = int((ans_m != -1).sum().item())
            ac_mtm = float(((pred_m == ans_m) & (ans_m != -1)).sum().item()) / nm if nm > 0 else -1
            ac_vtm = float((outs["out_vtm"].argmax(-1) == 0).float().mean().item())
            r = {"mtm": ac_mtm, "vtm": ac_vtm}
            if "pixel" in self.args.mvm_target:
                r["mvm_pixel"] = float(losses["mvm_pixel"].item())
            if "hog" in self.args.mvm_target:
                r["mvm_hog"] = float(losses["mvm_hog"].item())
            if "3d_feature" in self.args.mvm_target or "2d_feature" in self.args.mvm_target:
                r["mvm_3d_feature"] = float(losses["mvm_feature"].item())      # the reference reports both under this key (:526,:545)
            if "vq" in self.args.mvm_target:           # accuracy over covered positions (main_pretrain.py:503-506)
                r["mvm_vq"] = float(outs["vq_acc"].item()) if "vq_acc" in outs else -1
            if "out_smtm" in outs:                     # main_pretrain.py:581-584: the seq2seq-masked pass is scored against the MLM answers
                pred_s = outs["out_smtm"].argmax(-1)
                r["smtm"] = float(((pred_s == ans_m) & (ans_m != -1)).sum().item()) / nm if nm > 0 else -1
            return r
        if not sync:
            return losses
        smtm = float(losses["smtm"].item()) if "smtm" in self.args.get("pretrain_tasks", ()) else -1
        return {"mtm": float(losses["mtm"].item()), "mvm": float(losses["mvm"].item()), "vtm": float(losses["vtm"].item()), "smtm": smtm}

    def go_dl(self, ep, dl, is_train):
        """main_pretrain.py:588-610 : one pass over a loader of {img, txt, mask} batches; returns rank-averaged means."""
        ret = {}
        for batch in dl:
            masked = self.masking(batch["img"], batch["txt"], batch["mask"], batch.get("vq"))
            if batch.get("hog") is not None:
                masked["hog"] = batch["hog"]
            r = self.step(self.prepare_batch(masked), is_train)
            for k, v in r.items():
                ret.setdefault(k, []).append(v)
        return {k: self.reduce_mean(float(np.mean([x for x in v if x == x]))) for k, v in ret.items()}

    def evaluate(self, dl):
        """main_pretrain_yaml.py:196-214 : eval mode, mask every batch like training, step(is_train=False), NaN-ignoring means
        averaged over the ranks; back to train mode."""
        self.model.eval()
        ret = {}
        for batch in dl:
            masked = self.masking(batch["img"], batch["txt"], batch["mask"], batch.get("vq"))
            if batch.get("hog") is not None:
                masked["hog"] = batch["hog"]
            r = self.step(self.prepare_batch(masked), is_train=False)
            for k, v in r.items():
                ret.setdefault(k, []).append(v)
        out = {k: self.reduce_mean(float(np.average([x for x in v if x == x]))) for k, v in ret.items()}
        self.model.train()
        return out

    def forward_step(self, batch):
        """agent.py:161-179 : `out = self.model(batch)` -- with grad mode on the outputs carry a grad_fn (model._OpenStep), so the
        reference's own `step` body (losses in plain torch, `backward_step(ls)`) drives this model."""
        return self.model(batch)

    def backward_step(self, loss=None):
        """[loss.backward() ->] all-reduce (rest) -> global grad norm -> clip -> AdamW -> scheduler.step -> zero_grad   (agent.py:181-193).
        `loss` = the reference's call form `backward_step(ls)`: the scalar of a step that went through `forward_step` / `model(batch)`;
        None = the gradients are already in the arena (the fused `step()` below)."""
        if loss is not None:
            loss.backward()
        S = self.engine.store
        S.sync_pending()                                 # (a second backward_step without a forward in between: the previous tail first)
        if self.comm is not None:
            self.comm.reduce_swin_and_wait()
        self.opt_step += 1
        gscale = 1.0 / self.world_size
        self._sumsq.zero_()
        z1 = self.comm is not None and getattr(self.comm, "zero1", False)
        own = self.comm.own if z1 else [(0, S.n_trainable)]       # (ZeRO-1: the parts of every reduction range this rank owns, dist.GradReducer.owned_ranges)
        if self.args.max_grad_norm > 0:
            for oa, oe in own:
                if oe > oa:
                    K.sumsq(S.grad[oa:oe], self._sumsq)
            if z1:
                D.all_reduce_(self._sumsq)               # the global norm: every rank holds the reduced gradient of its parts only
        lrs = self.current_lrs()

        def update(groups):
            for gi in groups:
                ga, ge = S.segments[gi]
                for oa, oe in own:
                    a, e = max(ga, oa), min(ge, oe)      # (ZeRO-1: this rank's parts of the group; otherwise the whole group)
                    if e > a:
                        K.adamw(S.flat[a:e], S.grad[a:e], S.m[a:e], S.v[a:e], S.shadow[a:e], lr=lrs[gi], weight_decay=(self.args.decay if gi < 2 else 0.0),
                                beta1=0.9, beta2=0.98, eps=1e-8, step=self.opt_step, sumsq_t=self._sumsq, max_grad_norm=float(self.args.max_grad_norm),
                                grad_scale=gscale)
        eng = self.engine
        split = getattr(eng, "wstream", None) is not None and eng.sw.opt_overlap and not z1 and getattr(S, "shadow8", None) is None
        if split:
            # The next forward starts with the Video-Swin backbone, which reads Swin parameters only: the update of the other 137 M parameters
            # (fusion encoder, heads, embeddings), their W^T copies and the zeroing of their gradients run on the engine's second stream
            # beside it; engine.encode() waits for `other_ready` before the first non-Swin parameter is read.  (The clip coefficient is
            # read from _sumsq by both halves: the side stream starts behind the norm.)
            # The side stream's half starts BEHIND the Swin half: side by side the two updates only share the HBM they are both bound by, and
            # the Swin half -- the one the next forward waits for -- took 1.2 ms instead of 0.6; behind it, the other half runs beside the
            # forward's first kernels.  (Round 4 kept the other order behind VMVM_OPT_ORDER; its A/B was 0.07-0.16 ms: removed in round 5.)
            update((0, 2))
            eng.wstream.wait_stream(torch.cuda.current_stream())
            with L.on_stream(eng.wstream):
                update((1, 3))
                S.refresh_transposed("other")
                for gi in (1, 3):
                    a, e = S.segments[gi]
                    S.grad[a:e].zero_()
                eng.other_ready = torch.cuda.Event()
                eng.other_ready.record()
                S.pending = eng.other_ready              # (ParamStore.sync_pending: every reader of the non-Swin arena waits for this)
            S.refresh_transposed("swin")                 # (measured in round 5: on the side stream instead -- the backward is their only reader -- the step is the same, 106.74 vs 106.83 ms)
            for gi in (0, 2):
                a, e = S.segments[gi]
                S.grad[a:e].zero_()
            S.grad[S.n_trainable:].zero_()               # the frozen segment + padding tail (S.grad.zero_() of the unsplit path covers them)
        else:
            update(range(4))
            if z1:                                       # the other ranks' updated master shards, then their bf16 compute copies
                for sa, se in self.comm.gather_params(S.flat):
                    K.cast_bf16(S.flat[sa:se], S.shadow[sa:se])
            S.refresh_transposed()
            S.refresh_fp8()                              # (opt-in fp8 forward: the e4m3 weight copy follows the updated bf16 copy)
            S.grad.zero_()
        self.sched_step += 1

    def grad_norm(self):
        return float(torch.sqrt(self._sumsq).item()) / self.world_size

    # ------------------------------------------------------------------ checkpoint surface (main_pretrain.py:612-619, agent.py:127-132)
    def save_training_meta(self):
        if self.rank == 0:
            os.makedirs(self.args.path_output, exist_ok=True)
            json.dump(dict(self.args), open(f"{self.args.path_output}/args.json", "w"), indent=2)
            self.save_model(0)

    def save_model(self, ep, dataset="init", part=0):
        if self.rank == 0:
            os.makedirs(self.args.path_output, exist_ok=True)
            sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
            torch.save(sd, os.path.join(self.args.path_output, f"ckpt_violet_pretrain_{dataset}_{part}_{ep}.pt"))
