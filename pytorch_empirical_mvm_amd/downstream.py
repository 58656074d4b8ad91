"""Downstream fine-tuning heads on the accelerated encoders.
* Text-to-video retrieval: `VIOLET_Retrieval` / `Agent_Retrieval` (main_retrieval.py:56-110, NormSoftmaxLoss agent.py:34-50) -- the
  batch's B x B (video, text) pairs are B*B gathered sequences of one fusion pass.
* Open-ended video QA: `VIOLET_QAOE` / `Agent_QAOE` (main_qaoe.py:42-90) -- one (video, question) pass, classification over the
  answer vocabulary on the text [CLS] state.
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
