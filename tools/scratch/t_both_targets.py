import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
args = CFG.get_args(vis_backbone_size="tiny", size_frame=4, max_size_frame=6, mvm_target=["pixel", "vq"], dvae_hid=64, dvae_vocab=512, dvae_dtype=torch.float32)
model = VIOLET_Pretrain(args, None, device="cuda"); agent = Agent_Pretrain(args, model)
cfg = R.make_cfg("tiny", T=4, mvm_target=["pixel", "vq"], dvae_hid=64, dvae_vocab=512)
sd = R.make_state_dict(cfg); model.load_state_dict(sd)
img, txt, mask = R.make_batch(cfg, 2)
mb = R.default_masking(cfg, img, txt, mask, seed=5)
cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8)
ref = R.pretrain_losses(sd, cfg, mb, negatives=R.vtm_negatives_default(2))
vqi = agent.vq_index(cov)
b = dict(img=img.cuda(), cov=cov.cuda().contiguous(), txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda(),
         vq_patch_rows=vqi["vq_patch_rows"].cuda(), vq_tok_index=vqi["vq_tok_index"].cuda(), vq_tokens=R.vq_tokens(sd, cfg, img).cuda())
ls, _ = model.engine.forward_backward(b, negatives=R.vtm_negatives_default(2), train=False, backward=True)
print("engine mvm", float(ls["mvm"]), "pixel", float(ls["mvm_pixel"]), "vq", float(ls["mvm_vq"]), " oracle mvm", float(ref["mvm"]))
assert abs(float(ls["mvm"]) - float(ref["mvm"])) < 2e-2 * float(ref["mvm"])
print("BOTH OK")
