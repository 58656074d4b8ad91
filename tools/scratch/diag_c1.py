#!/usr/bin/env python3
"""Diagnostic: stage-by-stage comparison of the HIP pretraining forward against the CPU oracle + optimizer trajectory."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import violet_ref as R
from pytorch_empirical_mvm_amd import config as CFG, kernels as K
from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
from pytorch_empirical_mvm_amd.engine import V

def cmp(name, got, ref):
    got, ref = got.float().cpu().double().flatten(), ref.detach().double().flatten()
    cos = float(got @ ref / (got.norm() * ref.norm() + 1e-30))
    print(f"{name:44s} rel_fro={float((got-ref).norm()/(ref.norm()+1e-30)):.4e} max={float((got-ref).abs().max()):.4e} refmax={float(ref.abs().max()):.3e} cos={cos:.6f}", flush=True)

mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
if mode == "fwd":
    size, T, B = "tiny", 4, 2
    args = CFG.get_args(vis_backbone_size=size, size_frame=T, max_size_frame=6)
    model = VIOLET_Pretrain(args, None, device="cuda")
    cfg = R.make_cfg(size, T=T)
    sd = R.make_state_dict(cfg)
    model.load_state_dict(sd)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=3)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
    eng = model.engine
    eng.tape = []
    pool, Lv, hw = eng.encode(img.cuda(), cov, mb["txt"].cuda(), None, False)
    fi, mi = R.enc_video(sd, cfg, mb["img"])
    ft = R.enc_txt(sd, mb["txt"])
    cmp("feat_img", pool.t[:B * Lv], fi)
    cmp("feat_txt", pool.t[B * Lv:], ft)
    X = txt.shape[1]; Lq = Lv + X
    ar_v, ar_t = np.arange(Lv), np.arange(X)
    idx1 = np.concatenate([np.concatenate([i * Lv + ar_v, B * Lv + i * X + ar_t]) for i in range(B)])
    idx1_d = torch.from_numpy(idx1.astype(np.int32)).cuda()
    km1 = torch.cat([torch.ones(B, Lv, dtype=torch.uint8), (mask != 0).to(torch.uint8)], 1).cuda().contiguous()
    x = K.gather_rows(pool.t, idx1_d, B * Lq)
    feat = torch.cat([fi, ft], dim=1)
    cmp("fusion input", x, feat)
    m_all = torch.cat([mi, mask], dim=1)
    add = (1.0 - m_all[:, None, None, :].float()) * torch.finfo(torch.float32).min
    cur = V(x)
    for l in range(12):
        cur = eng._bert_layer(cur, B, Lq, km1, l, False)
        feat = R.bert_layer(sd, f"trsfr.layer.{l}.", feat, add)
        cmp(f"bert layer {l} (all)", cur.t, feat)
        cmp(f"bert layer {l} (visual rows)", cur.t.view(B, Lq, -1)[:, :Lv], feat[:, :Lv])
        cmp(f"bert layer {l} (text rows)", cur.t.view(B, Lq, -1)[:, Lv:], feat[:, Lv:])
else:
    arch = dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    args = CFG.get_args(vis_backbone_size="tiny", size_frame=2, max_size_frame=6, arch_override=arch, bert_layers=1, max_iter=20, lr=5e-5, size_img=96)
    model = VIOLET_Pretrain(args, None, device="cuda")
    cfg = R.make_cfg("tiny", T=2, img=96, arch=arch, bert_layers=1)
    sd = R.make_state_dict(cfg)
    model.load_state_dict(sd)
    img, txt, mask = R.make_batch(cfg, 2)
    mb = R.default_masking(cfg, img, txt, mask, seed=1)
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8)
    neg = R.vtm_negatives_default(2)
    agent = Agent_Pretrain(args, model)
    b = agent.prepare_batch(dict(unmask_img=img, cov=cov.contiguous(), txt=mb["txt"], mask=mask, ans_mtm=mb["ans_mtm"]))
    opt_state = {}
    for step in range(1, 4):
        before = {k: v.clone() for k, v in sd.items()}
        ref = R.train_step(sd, cfg, mb, opt_state, step, 20, negatives=neg, lr=5e-5)
        S = model.engine.store
        before_g = {k: S.p(k).detach().cpu().clone() for k in S.index}
        losses, _ = model.engine.forward_backward(dict(img=b["unmask_img"], cov=b["cov"], txt=b["txt"], mask=b["mask"], ans_mtm=b["ans_mtm"]), negatives=neg, train=False, backward=True)
        grads = {k: S.g(k).detach().cpu().clone() for k in S.index}
        agent.backward_step()
        torch.cuda.synchronize()
        print(f"== step {step}: losses got {[round(float(losses[k].item()),4) for k in ('mtm','vtm','mvm')]} ref {[round(ref[k],4) for k in ('mtm','vtm','mvm')]} gnorm got {agent.grad_norm():.3f} ref {ref['grad_norm']:.3f}")
        rows = []
        for k in S.index:
            if k not in ref["grads"]: continue
            g_ref, g_got = ref["grads"][k].double().flatten(), grads[k].double().flatten()
            cosg = float(g_ref @ g_got / (g_ref.norm() * g_got.norm() + 1e-30))
            rows.append((float(g_got.norm()) / (float(g_ref.norm()) + 1e-30), cosg, float(g_ref.norm()), k))
        rows.sort(key=lambda r: r[1])
        for r in rows[:12]:
            print(f"   grad |got|/|ref|={r[0]:.3f} cos={r[1]:.4f} |ref|={r[2]:.3e} {r[3]}")
        rows.sort(key=lambda r: -abs(np.log(r[0] + 1e-30)))
        for r in rows[:8]:
            print(f"   NORM |got|/|ref|={r[0]:.3f} cos={r[1]:.4f} |ref|={r[2]:.3e} {r[3]}")
