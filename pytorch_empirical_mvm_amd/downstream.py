"""Downstream fine-tuning heads on the accelerated encoders.
* Text-to-video retrieval: `VIOLET_Retrieval` / `Agent_Retrieval` (main_retrieval.py:56-110, NormSoftmaxLoss agent.py:34-50) -- the
  batch's B x B (video, text) pairs are B*B gathered sequences of one fusion pass.
* Open-ended video QA: `VIOLET_QAOE` / `Agent_QAOE` (main_qaoe.py:42-90) -- one (video, question) pass, classification over the
  answer vocabulary on the text [CLS] state.
* Multiple-choice video QA, MLM-head form: `VIOLET_QAMC_MLM_Head` / `Agent_QAMC_MLM_Head` (main_qamc_tsv_mlm_head.py:61-123) -- one
  "question + option + [MASK]" sequence per option, the clip's video tokens shared by its options, the pre-training MLM head
  predicts the true / false token at [MASK].
* The two single-sequence MLM-head variants: generative multiple choice `VIOLET_QAMC_MLM_Head_GEN` / `Agent_QAMC_MLM_Head_GEN`
  (main_qamc_tsv_mlm_gen_ans_idx.py:83-125: the candidate answers' token logits at [MASK]) and open-ended QA on the MLM head
  `VIOLET_QAOE_LSMDC` / `Agent_QAOE_LSMDC` = `Agent_QAOE_MLM_Head` (main_qaoe_lsmdc_fib.py:55-115, main_qaoe_tsv_mlm_head.py:101-130:
  top-1 / top-5 accuracy).
Same encoders, token pool and fusion kernels as pre-training; checkpoints interchange through the shared key names."""
import torch

from . import config as CFG
from .agent import Agent_Pretrain
from .model import VIOLET_Pretrain


class VIOLET_Retrieval(VIOLET_Pretrain):
    """VIOLET_Base + `fc` (Dropout, Linear(768, 1536), ReLU, Linear(1536, 1)); checkpoint keys enc_img.* / enc_txt.* / trsfr.* / fc.*"""

    def __init__(self, args, tokzr=None, device="cuda"):
        args = CFG.Args(dict(args))
        args.update(task="retrieval", mvm_target=[])
        super().__init__(args, tokzr, device=device)

    @torch.no_grad()
    def forward(self, img, txt, mask, vid=None):
        """-> (scores (B,B) f32 = fc output of pair (video i, text j), ans = arange(B))   (main_retrieval.py:63-85)"""
        dev = self.engine.device
        _, scores = self.engine.retrieval_forward_backward(img.to(dev, torch.float32).contiguous(), txt.to(dev).contiguous(),
                                                           mask.to(dev).contiguous(), train=self.training, backward=False)
        return scores, torch.arange(img.shape[0], device=dev)


class Agent_Retrieval(Agent_Pretrain):
    """Agent_Retrieval.step (main_retrieval.py:95-110): train -> NormSoftmaxLoss(temp) + the shared backward_step (all-reduce,
    clip, AdamW, schedule); eval -> top-1 accuracy of the row arg-max."""

    def step(self, img, txt, mask, vid=None, is_train=True):
        eng = self.engine
        dev = eng.device
        img, txt, mask = img.to(dev, torch.float32).contiguous(), txt.to(dev).contiguous(), mask.to(dev).contiguous()
        if is_train:
            loss, _ = eng.retrieval_forward_backward(img, txt, mask, train=True, backward=True)
            if self.comm is not None:
                self.comm.reduce_other()
            self.backward_step()
            self.global_step += 1
            return float(loss.item())
        _, scores = eng.retrieval_forward_backward(img, txt, mask, train=False, backward=False)
        return float((scores.argmax(1) == torch.arange(img.shape[0], device=dev)).float().mean().item())


class VIOLET_QAOE(VIOLET_Pretrain):
    """VIOLET_Base + `fc` (Dropout, Linear(768, 1536), ReLU, Linear(1536, size_vocab))   (main_qaoe.py:42-47)"""

    def __init__(self, args, tokzr=None, device="cuda"):
        args = CFG.Args(dict(args))
        args.update(task="qaoe", mvm_target=[])
        assert int(args.get("size_vocab", 0)) > 0, "args.size_vocab (answer vocabulary) is required"
        super().__init__(args, tokzr, device=device)

    @torch.no_grad()
    def forward(self, img, txt, mask, ans):
        dev = self.engine.device
        _, logits = self.engine.qaoe_forward_backward(img.to(dev, torch.float32).contiguous(), txt.to(dev).contiguous(), mask.to(dev).contiguous(),
                                                      ans, train=self.training, backward=False)
        return logits, ans


class Agent_QAOE(Agent_Pretrain):
    """Agent_QAOE.step (main_qaoe.py:72-90): train -> CE(ignore_index=-1) + backward_step; eval -> per-sample correctness list"""

    def step(self, img, txt, mask, ans, is_train=True):
        eng = self.engine
        dev = eng.device
        img, txt, mask = img.to(dev, torch.float32).contiguous(), txt.to(dev).contiguous(), mask.to(dev).contiguous()
        if is_train:
            loss, _ = eng.qaoe_forward_backward(img, txt, mask, ans, train=True, backward=True)
            if self.comm is not None:
                self.comm.reduce_other()
            self.backward_step()
            self.global_step += 1
            return float(loss.item())
        _, logits = eng.qaoe_forward_backward(img, txt, mask, ans, train=False, backward=False)
        return (logits.argmax(1) == ans.to(dev)).float().tolist()


class VIOLET_QAMC_MLM_Head(VIOLET_Pretrain):
    """VIOLET_Base + the pre-training MLM head `fc_mtm` (+ `emb_task`, unused while the task token is off); no `fc`
    (main_qamc_tsv_mlm_head.py:61-71).  A pre-training checkpoint loads through `load_ckpt` (shared key names)."""

    def __init__(self, args, tokzr=None, device="cuda"):
        args = CFG.Args(dict(args))
        args.update(task="qamc_mlm", mvm_target=[])
        if args.get("enable_task_token") or args.get("enable_prompt"):
            raise NotImplementedError("task token / prompt prefixes (model.py:219-293) are outside the accelerated path")
        super().__init__(args, tokzr, device=device)

    @torch.no_grad()
    def forward(self, batch):
        """batch: img (B,T,3,H,W), txt / mask / mask_ans (B,O,X) -> (out (B*O, X, vocab) f32, ans (B,O,X))   (:76-94)"""
        dev = self.engine.device
        txt = batch["txt"]
        B, O, X = txt.shape
        _, lg = self.engine.qamc_mlm_forward_backward(batch["img"].to(dev, torch.float32).contiguous(), txt.to(dev), batch["mask"].to(dev),
                                                      batch["mask_ans"].to(dev), train=self.training, backward=False)
        return lg.view(B * O, X, -1), batch["mask_ans"].to(dev)


class Agent_QAMC_MLM_Head(Agent_Pretrain):
    """Agent_QAMC_MLM_Head.step (main_qamc_tsv_mlm_head.py:100-123): train -> cross entropy over the text positions (ignore -1) + the
    shared backward_step; eval -> per-clip accuracy of arg-max over the options of p_true / (p_true + p_false) at [MASK]."""

    def __init__(self, args, model, true_token_id=2995, false_token_id=6270):       # bert-base-uncased ids of "true" / "false"
        super().__init__(args, model)
        self.true_token_id, self.false_token_id = int(true_token_id), int(false_token_id)

    def current_lrs(self):
        """Agent_QAMC.build_optimizer (main_qamc.py:111-140): the `vis_backbone_lr_mul` group is the parameters whose name starts with
        `fc.` -- none in the MLM-head model -- so every group runs at the base rate."""
        lr = max(1e-8, self.args.lr * CFG.lr_factor(self.sched_step, self.args.max_iter))
        return [lr, lr, lr, lr]

    def step(self, batch, is_train=True):
        eng = self.engine
        dev = eng.device
        img = batch["img"].to(dev, torch.float32).contiguous()
        txt, mask, ans = batch["txt"].to(dev), batch["mask"].to(dev), batch["mask_ans"].to(dev)
        if is_train:
            loss, _ = eng.qamc_mlm_forward_backward(img, txt, mask, ans, train=True, backward=True)
            if self.comm is not None:
                self.comm.reduce_other()
            self.backward_step()
            self.global_step += 1
            return float(loss.item())
        B, O, X = txt.shape
        _, lg = eng.qamc_mlm_forward_backward(img, txt, mask, ans, train=False, backward=False)
        lg = lg.view(B * O, X, -1)
        pt, pf = lg[:, :, self.true_token_id], lg[:, :, self.false_token_id]
        sc = pt / (pt + pf)
        m = ans.reshape(B * O, X)
        sc, am = sc[m != -1].view(B, O), m[m != -1].view(B, O)
        return (sc.argmax(-1) == (am == self.true_token_id).nonzero()[:, 1]).float().tolist()


class VIOLET_QAMC_MLM_Head_GEN(VIOLET_QAMC_MLM_Head):
    """main_qamc_tsv_mlm_gen_ans_idx.py:83-101: ONE "question + options + [MASK]" sequence per clip -- txt / mask / mask_ans (B, X) --
    through the same encoders and MLM head (the multiple-choice pass with a single option per clip)."""

    @torch.no_grad()
    def forward(self, batch):
        """-> (out (B, X, vocab) f32, ans (B, X))"""
        dev = self.engine.device
        txt = batch["txt"]
        B, X = txt.shape
        _, lg = self.engine.qamc_mlm_forward_backward(batch["img"].to(dev, torch.float32).contiguous(), txt.to(dev)[:, None], batch["mask"].to(dev)[:, None],
                                                      batch["mask_ans"].to(dev)[:, None], train=self.training, backward=False)
        return lg.view(B, X, -1), batch["mask_ans"].to(dev)


class VIOLET_QAOE_LSMDC(VIOLET_QAMC_MLM_Head_GEN):
    """main_qaoe_lsmdc_fib.py:55-84 (`size_vocab == -1`: the answer is a token of the MLM vocabulary): same forward as the generative
    multiple-choice model."""


VIOLET_QAOE_MLM_Head = VIOLET_QAOE_LSMDC            # main_qaoe_tsv_mlm_head.py builds VIOLET_QAOE_LSMDC under this task name


class _Agent_MLM_QA(Agent_QAMC_MLM_Head):
    """shared train branch: CE(ignore_index=-1) over the text positions of the single-sequence form + the shared backward_step"""

    def _fwd(self, batch, is_train):
        eng, dev = self.engine, self.engine.device
        img = batch["img"].to(dev, torch.float32).contiguous()
        txt, mask, ans = batch["txt"].to(dev), batch["mask"].to(dev), batch["mask_ans"].to(dev)
        loss, lg = eng.qamc_mlm_forward_backward(img, txt[:, None], mask[:, None], ans[:, None], train=is_train, backward=is_train)
        return loss, lg.view(txt.shape[0], txt.shape[1], -1), ans

    def _train(self, batch):
        loss, _, _ = self._fwd(batch, True)
        if self.comm is not None:
            self.comm.reduce_other()
        self.backward_step()
        self.global_step += 1
        return float(loss.item())


class Agent_QAMC_MLM_Head_GEN(_Agent_MLM_QA):
    """Agent_QAMC_MLM_Head_GEN.step (main_qamc_tsv_mlm_gen_ans_idx.py:103-125).  eval: the RAW logits of the candidate answer tokens
    `ans_tok_ids` at the clip's [MASK] position, divided by their sum, arg-max against batch["ans_idx"] -> per-clip correctness."""

    def __init__(self, args, model, ans_tok_ids):
        super().__init__(args, model)
        self.ans_tok_ids = [int(i) for i in ans_tok_ids]

    def step(self, batch, is_train=True):
        if is_train:
            return self._train(batch)
        _, out, ans = self._fwd(batch, False)
        B = ans.shape[0]
        p = out[:, :, self.ans_tok_ids][ans != -1]
        p = (p / p.sum(dim=-1).view(B, 1)).view(B, -1)
        return (p.argmax(dim=-1) == batch["ans_idx"].to(p.device)).float().tolist()


class Agent_QAOE_LSMDC(_Agent_MLM_QA):
    """Agent_QAOE_LSMDC.step / get_top_k_acc (main_qaoe_lsmdc_fib.py:86-112): train -> {'ls'}; eval -> {'ac_1', 'ac_5'} per-clip lists"""

    def step(self, batch, is_train=True):
        if is_train:
            return {"ls": self._train(batch)}
        _, out, ans = self._fwd(batch, False)
        return {"ac_1": self.get_top_k_acc(out, ans, k=1), "ac_5": self.get_top_k_acc(out, ans, k=5)}

    @staticmethod
    def get_top_k_acc(out, ans, k=5):
        """1.0 per labelled position whose label is among the k largest logits; padded with 0.0 to the batch size"""
        B = out.shape[0]
        sel = ans != -1
        ac = []
        if bool(sel.any()):
            lab = ans[sel].view(-1, 1)
            top = torch.topk(out[sel].view(lab.shape[0], -1), k=k, dim=-1).indices
            ac = (top == lab).any(dim=-1).float().tolist()
        return ac + [0.0] * (B - len(ac))


Agent_QAOE_MLM_Head = Agent_QAOE_LSMDC              # main_qaoe_tsv_mlm_head.py:101-105 adds only the optional freeze
