"""diagnostic: VTM head gradients of the HIP path vs the oracle's head on the same [CLS] rows"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))
arch = dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
for B in (2, 4):
    args = CFG.get_args(vis_backbone_size="tiny", size_frame=2, max_size_frame=6, arch_override=arch, bert_layers=1, size_img=96, temp=1.0)
    model = VIOLET_Pretrain(args, None, device="cuda")
    cfg = R.make_cfg("tiny", T=2, img=96, arch=arch, bert_layers=1, temp=1.0)
    sd = R.make_state_dict(cfg)
    model.load_state_dict(sd)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=3)
    neg = R.vtm_negatives_default(B)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
    eng = model.engine
    eng.store.grad.zero_()
    losses, outs = eng.forward_backward(dict(img=img.cuda(), cov=cov, txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda()),
                                        negatives=neg, train=False, backward=True, want_outputs=True)
    torch.cuda.synchronize()
    x = outs["vtm_cls"].float().cpu()
    O = min(B, 4)
    hp = {k: sd[k].clone().requires_grad_(True) for k in ("fc.1.weight", "fc.1.bias", "fc.3.weight", "fc.3.bias")}
    lg = R.vtm_head(hp, x, 1.0).view(B, O)
    torch.nn.functional.cross_entropy(lg, torch.zeros(B, dtype=torch.long)).backward()
    print("B", B, "logits got", outs["out_vtm"].float().cpu().flatten()[:4].tolist(), "ref", lg.detach().flatten()[:4].tolist())
    print("  row diffs |x_neg-x_pos|/|x|:", float((x[1] - x[0]).norm() / x[0].norm()))
    for k, prm in hp.items():
        got = eng.store.g(k).detach().cpu().double().flatten()
        print(f"  {k}: cos {cos(got, prm.grad):.5f} ratio {float(got.norm() / prm.grad.double().norm()):.4f} |ref| {float(prm.grad.norm()):.3e}")
    # the same with bf16-rounded weights on the oracle side (the engine multiplies bf16 weight copies)
    hp2 = {k: (sd[k].to(torch.bfloat16).float() if k.endswith("1.weight") else sd[k].clone()).requires_grad_(True) for k in hp}
    lg2 = R.vtm_head(hp2, x, 1.0).view(B, O)
    torch.nn.functional.cross_entropy(lg2, torch.zeros(B, dtype=torch.long)).backward()
    for k, prm in hp2.items():
        got = eng.store.g(k).detach().cpu().double().flatten()
        print(f"  [bf16 W1] {k}: cos {cos(got, prm.grad):.5f}")
