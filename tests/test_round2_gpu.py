"""Round-2 parity cases on the BASELINE configurations ("next round" item 1) and the boundary additions:

* full-size C2 (Swin-B, 8 x 224^2, 432-token fusion sequences) PER-TENSOR gradients against the oracle's autograd -- the 18-deep
  stage 3, 16 / 32 heads, split-K weight gradients at K = 50 176 tokens and the fused bias-gradient column sums end to end;
* full-width config 5 (Swin-L-384, window (8,12,12), 16 x 384^2) step;
* the native dVAE tokenizer against the ORACLE's encoder (not against the package's own torch path);
* RCCL itself: backend nccl at world size 1 with the reducer active (VMVM_FORCE_DIST) == the run without a reducer;
* evaluate() / step(is_train=False) incl. the smtm accuracy; save_model -> load_ckpt with the max_size_frame 6 -> 8 resize rule;
  upstream Video-Swin checkpoint loading (load_checkpoint_3d) and the standalone get_vidswin_model / EncVideo / go_feat surface."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.join(os.path.dirname(__file__), "..")


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def _engine(cfg_args):
    from pytorch_empirical_mvm_amd import config as CFG
    from pytorch_empirical_mvm_amd.model import VIOLET_Pretrain
    args = CFG.get_args(**cfg_args)
    return VIOLET_Pretrain(args, None, device="cuda"), args


@pytest.mark.timeout(1500)
def test_full_size_c2_gradients_vs_oracle():
    """Swin-B, T = 8, 224^2, B = 2: every parameter gradient with norm above 1e-3 of the largest against the CPU oracle's autograd
    at temp = 1.0 (cosine >= 0.997, norm within 2.5 %: the bars README and the header of test_parity_gpu.py state;
    measured minimum 0.9986), the global gradient norm within 1 %; then the
    global gradient norm at the reference's temp = 0.05 (6 %).
    The VTM head `fc.1` / `fc.3`: its gradient is p(neg) * (h(neg) - h(pos)) of two [CLS] states that differ only through the text
    (a difference far below bf16 resolution with closed-form weights), so it is compared with the oracle's head evaluated on the
    HIP path's OWN [CLS] states -- the head's forward / softmax-gradient / backward kernels on identical inputs."""
    from oracle import violet_ref as R
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    B = 2
    for temp in (1.0, 0.05):
        cfg = R.make_cfg("base", T=8, temp=temp)
        model, args = _engine(dict(vis_backbone_size="base", size_frame=8, max_size_frame=8, temp=temp))
        sd = R.make_state_dict(cfg)
        model.load_state_dict(sd)
        img, txt, mask = R.make_batch(cfg, B)
        mb = R.default_masking(cfg, img, txt, mask, seed=3)
        neg = R.vtm_negatives_default(B)
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        ls = R.pretrain_losses(params, cfg, mb, negatives=neg)
        ls["total"].backward()
        cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
        eng = model.engine
        eng.store.grad.zero_()
        losses, outs = eng.forward_backward(dict(img=img.cuda(), cov=cov, txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda()),
                                            negatives=neg, train=False, backward=True, want_outputs=True)
        torch.cuda.synchronize()
        for k in ("mtm", "mvm"):
            assert abs(float(losses[k].item()) - float(ls[k].detach())) <= 2e-2 * abs(float(ls[k].detach())) + 1e-3, (temp, k)
        # VTM head on the engine's own [CLS] rows
        # (fc.1.weight as the bf16 copy the GEMM multiplies: which ReLU units sit at the pos / neg boundary decides the gradient's
        #  direction here; fc.3.weight is a difference of bf16-rounded activations: 0.97; fc.3.bias: analytically zero)
        hp = {k: (sd[k].to(torch.bfloat16).float() if k == "fc.1.weight" else sd[k].clone()).requires_grad_(True)
              for k in ("fc.1.weight", "fc.1.bias", "fc.3.weight", "fc.3.bias")}
        lg = R.vtm_head(hp, outs["vtm_cls"].float().cpu(), temp).view(B, -1)
        torch.nn.functional.cross_entropy(lg, torch.zeros(B, dtype=torch.long)).backward()
        for k, thr in (("fc.1.weight", 0.999), ("fc.1.bias", 0.999), ("fc.3.weight", 0.97)):          # (measured: 0.99995, 0.99996, 0.984)
            got = eng.store.g(k).detach().cpu().double().flatten()
            c = _cos(got, hp[k].grad)
            print(f"\n[vtm head, temp {temp}] {k}: cosine {c:.5f}, norm ratio {float(got.norm() / hp[k].grad.double().norm()):.4f}")
            assert c >= thr and abs(float(got.norm() / hp[k].grad.double().norm()) - 1.0) <= 0.05, (temp, k, c)
        ref_norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None)))
        S = eng.store
        got_norm = float(torch.sqrt((S.grad[:S.n_trainable].double() ** 2).sum()).item())
        # temp = 0.05 multiplies the VTM branch by 20: its gradient (see above: decided by the ReLU units at the pos / neg boundary)
        # dominates the norm, 4 % observed; at temp = 1 the norm agrees within 1 %
        assert abs(got_norm - ref_norm) <= (1e-2 if temp == 1.0 else 6e-2) * ref_norm, (temp, got_norm, ref_norm)
        if temp == 1.0:
            gmax = max(float(p.grad.norm()) for p in params.values() if p.grad is not None)
            bad, checked = [], 0
            for name, p in params.items():
                if p.grad is None or float(p.grad.norm()) < 1e-3 * gmax or name.startswith(("fc.1.", "fc.3.")):
                    continue
                got = eng.store.g(name).detach().cpu().double().flatten()
                ref = p.grad.double().flatten()
                cos, ratio = _cos(got, ref), float(got.norm() / ref.norm())
                checked += 1
                if cos < 0.997 or abs(ratio - 1.0) > 0.025:
                    bad.append((name, round(cos, 4), round(ratio, 3)))
            assert checked > 300 and not bad, (checked, bad[:12])
        del model, eng, params
        torch.cuda.empty_cache()


@pytest.mark.timeout(1200)
def test_full_width_c5_step():
    """BASELINE config 5 at full width (Swin-L, 16 x 384^2 frames, window (8,12,12): 1152-token windows, 2352-token fusion
    sequences on the streaming attention kernels), B = 2: one training step; finite losses, a positive finite gradient norm.  (Rounds 1-4
    ran this twice, with and without the opt-in e4m3 forward GEMMs; that path was removed in round 5 -- DESIGN 7 -- and the full-width
    forward is now pinned to the oracle in tests/test_round5_gpu.py.)"""
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    import bench
    model, args = _engine(dict(vis_backbone_size="large", size_frame=16, max_size_frame=16, size_img=384, max_iter=100, seed=88))
    agent = Agent_Pretrain(args, model)
    agent.sched_step = 10
    img, txt, mask = bench.synth_batch(args, 2, "cuda", 123)
    import random
    random.seed(5); np.random.seed(5); torch.manual_seed(5)
    mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
    model.eval()
    eng = model.engine
    b = dict(img=mb["unmask_img"].float().contiguous(), cov=mb["cov"].contiguous(), txt=mb["txt"], mask=mb["mask"], ans_mtm=mb["ans_mtm"])
    losses, _ = eng.forward_backward(b, negatives=np.array([[1], [0]]), train=False, backward=True)
    agent.backward_step()
    torch.cuda.synchronize()
    res = {k: float(losses[k].item()) for k in ("mtm", "vtm", "mvm")}
    assert all(np.isfinite(v) for v in res.values()), res
    assert np.isfinite(agent.grad_norm()) and agent.grad_norm() > 0
    del model, agent, eng
    torch.cuda.empty_cache()


@pytest.mark.timeout(900)
def test_gelu_code8_step_equals_bf16_preactivation_step():
    """The GELU backward's operand as an 8-bit code of GELU' (vmvm_gemm_desc.aux_code8, default) against the same step with the bf16
    pre-activation saved (gelu_code8=False), full-width C2 model, B = 2, dropout / DropPath off: the forward is the same arithmetic
    (losses equal to 1e-5 relative); every gradient tensor agrees to cosine >= 0.998 and norm within 2.5 % (measured: two thirds of
    the 489 tensors above 0.9995, the worst -- the patch-embedding weight, below all 24 Swin MLPs -- at 0.9986; norms within 1.5 % except
    the first Swin block's norm1.bias / qkv.bias at 1.7 % since round 3; the bar against the fp32 oracle is 0.99 / 5 %).  The multiplier is exact to 0.0025 ABSOLUTE, so the noise is largest where GELU' is small, and it
    accumulates down the network."""
    from oracle import violet_ref as R
    import bench
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    res = {}
    for c8 in (True, False):
        model, args = _engine(dict(vis_backbone_size="base", size_frame=8, max_size_frame=8, max_iter=100, seed=88))
        eng = model.engine
        eng.gelu_code8 = c8
        agent = Agent_Pretrain(args, model)
        img, txt, mask = bench.synth_batch(args, 2, "cuda", 321)
        import random
        random.seed(7); np.random.seed(7); torch.manual_seed(7)
        mb = agent.prepare_batch(agent.masking(img, txt, mask, None))
        model.eval()
        b = dict(img=mb["unmask_img"].float().contiguous(), cov=mb["cov"].contiguous(), txt=mb["txt"], mask=mb["mask"], ans_mtm=mb["ans_mtm"])
        eng.store.grad.zero_()
        losses, _ = eng.forward_backward(b, negatives=np.array([[1], [0]]), train=False, backward=True)
        torch.cuda.synchronize()
        res[c8] = ({k: float(losses[k].item()) for k in ("mtm", "vtm", "mvm")}, {n: eng.store.g(n).detach().float().cpu() for n in eng.store.index
                                                                                   if n not in eng.store.FROZEN})
        del model, agent, eng
        torch.cuda.empty_cache()
    for k in ("mtm", "vtm", "mvm"):
        assert abs(res[True][0][k] - res[False][0][k]) <= 1e-5 * abs(res[False][0][k]) + 1e-7, (k, res[True][0], res[False][0])
    gmax = max(float(g.norm()) for g in res[False][1].values())
    bad, checked = [], 0
    for n, ref in res[False][1].items():
        if float(ref.norm()) < 1e-3 * gmax:
            continue
        got = res[True][1][n]
        cos, ratio = _cos(got, ref), float(got.norm() / ref.norm())
        checked += 1
        if cos < 0.998 or abs(ratio - 1.0) > 0.025:
            bad.append((n, round(cos, 5), round(ratio, 4)))
    assert checked > 300 and not bad, (checked, bad[:12])



@pytest.mark.timeout(600)
def test_dvae_native_tokenizer_vs_oracle_encoder():
    """the full-width tokenizer (n_hid 256, 8192 codes) on the hand-written implicit-GEMM fp16 convolutions against
    the ORACLE's fp32 `dvae_encoder` (the restatement pinned to the reference's Encoder by tests/golden/vq.npz) on the same weights."""
    from oracle import violet_ref as R
    from pytorch_empirical_mvm_amd.dvae import DalleTeacher
    torch.manual_seed(0)
    t = DalleTeacher(256, 8192, device="cuda", seed=3)
    assert t.native
    sd = {k: v.float().cpu() for k, v in t.state_dict().items()}                 # keys dalle.encoder.blocks.*
    img = torch.randn(2, 3, 224, 224).clamp_(-2.1, 2.6)
    x = 0.8 * (img * torch.tensor(R.IMNET_STD).view(1, 3, 1, 1) + torch.tensor(R.IMNET_MEAN).view(1, 3, 1, 1)) + 0.1
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        zr = R.dvae_encoder(sd, dict(dvae_hid=256, dvae_vocab=8192), x).permute(0, 2, 3, 1)          # (N,28,28,V)
    zl = t.logits_native(img.cuda()).view(2, 28, 28, 8192).float().cpu()      # native path: pre-processing fused into the stem kernel
    assert _cos(zl, zr) >= 0.9995, _cos(zl, zr)
    assert float((zl - zr).abs().max()) <= 3e-2 * float(zr.abs().max())
    tok = t.extract_vq_token(img.cuda()).cpu()
    tok_r = R.vq_tokens(sd, dict(dvae_hid=256, dvae_vocab=8192), img.view(1, 2, 3, 224, 224))
    assert float((tok == tok_r).float().mean()) >= 0.98, float((tok == tok_r).float().mean())


@pytest.mark.timeout(900)
@pytest.mark.parametrize("wire", ["bf16", "f32"])
def test_rccl_world1_reducer_equals_no_reducer(wire):
    """RCCL (torch.distributed backend "nccl") initialised at world size 1 with the gradient reducer forced on (side-stream
    all-reduces hooked into the backward at the C2 shapes, B = 4): the reduced gradient equals the one of the run without a reducer,
    and three optimizer steps run through the whole path."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VMVM_FORCE_DIST="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", VMVM_GRAD_WIRE=wire)
    p = subprocess.run([sys.executable, "tools/rccl_smoke.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert f"rccl world-1: wire={wire} identical=True" in p.stdout, p.stdout[-2000:]


def test_evaluate_and_eval_step_with_smtm_accuracy(tmp_path):
    """Agent_Pretrain.evaluate (main_pretrain_yaml.py:196-214) over a small loader; step(is_train=False) returns the reference's
    accuracy keys incl. `smtm` (main_pretrain.py:574-586) and agrees with the accuracies recomputed from the model's outputs."""
    from oracle import violet_ref as R
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    arch = dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    tasks = ["vtm", "mlm", "mvm", "smtm"]
    model, args = _engine(dict(vis_backbone_size="tiny", size_frame=2, max_size_frame=6, arch_override=arch, bert_layers=1, size_img=96,
                               pretrain_tasks=tasks, path_output=str(tmp_path)))
    cfg = R.make_cfg("tiny", T=2, img=96, arch=arch, bert_layers=1, pretrain_tasks=tasks)
    model.load_state_dict(R.make_state_dict(cfg))
    agent = Agent_Pretrain(args, model)
    img, txt, mask = R.make_batch(cfg, 4)
    dl = [dict(img=img[:2], txt=txt[:2], mask=mask[:2]), dict(img=img[2:], txt=txt[2:], mask=mask[2:])]
    import random
    random.seed(0); np.random.seed(0); torch.manual_seed(0)
    r = agent.evaluate(dl)
    assert set(r) >= {"mtm", "vtm", "mvm_pixel", "smtm"}, r
    assert model.training                                    # evaluate() puts the model back into train mode
    assert all(np.isfinite(v) for v in r.values()) and 0.0 <= r["vtm"] <= 1.0 and -1 <= r["smtm"] <= 1.0
    # one batch by hand: accuracies from the returned logits
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    mb = agent.prepare_batch(agent.masking(img[:2], txt[:2], mask[:2], None))
    model.eval()
    r1 = agent.step(mb, is_train=False, negatives=R.vtm_negatives_default(2))
    out = model(dict(img=mb["unmask_img"], cov=mb["cov"], txt=mb["txt"], mask=mb["mask"], ans_mtm=mb["ans_mtm"]), negatives=R.vtm_negatives_default(2))
    ans = mb["ans_mtm"]
    nm = int((ans != -1).sum())
    want = float(((out["out_mtm"].argmax(-1) == ans) & (ans != -1)).sum()) / nm if nm else -1
    assert abs(r1["mtm"] - want) < 1e-6 and abs(r1["vtm"] - float((out["out_vtm"].argmax(-1) == 0).float().mean())) < 1e-6
    model.train()


def test_save_model_load_ckpt_round_trip_with_frame_resize(tmp_path):
    """save_model -> load_ckpt (main_pretrain.py:612-619, model.py:295-353): same-shape tensors come back bit-identical; a checkpoint
    saved at max_size_frame = 6 loads into a max_size_frame = 8 model with emb_len[:, :6] taken from the file and the rest kept."""
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    arch = dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    common = dict(vis_backbone_size="tiny", size_frame=2, arch_override=arch, bert_layers=1, size_img=96, path_output=str(tmp_path), dataset="unit")
    m6, a6 = _engine(dict(common, max_size_frame=6, seed=1))
    ag = Agent_Pretrain(a6, m6)
    ag.save_model(3, "unit", 0)
    path = os.path.join(str(tmp_path), "ckpt_violet_pretrain_unit_0_3.pt")
    assert os.path.exists(path)
    saved = torch.load(path, map_location="cpu")
    assert all(v.device.type == "cpu" for v in saved.values()) and "fc_mtm.predictions.decoder.bias" in saved
    m6b, _ = _engine(dict(common, max_size_frame=6, seed=2))
    m6b.load_ckpt(path)
    for k, v in m6.state_dict().items():
        assert torch.equal(v.cpu(), m6b.state_dict()[k].cpu()), k
    m8, _ = _engine(dict(common, max_size_frame=8, seed=2))
    before = m8.state_dict()["enc_img.emb_len"].cpu().clone()
    m8.load_ckpt(path)
    after = m8.state_dict()["enc_img.emb_len"].cpu()
    assert tuple(after.shape) == (1, 8, 1, 768)
    assert torch.equal(after[:, :6], saved["enc_img.emb_len"]) and torch.equal(after[:, 6:], before[:, 6:])
    assert torch.equal(m8.state_dict()["trsfr.layer.0.output.dense.weight"].cpu(), saved["trsfr.layer.0.output.dense.weight"])
    # the bf16 compute copy follows the loaded parameters
    S = m8.engine.store
    assert torch.equal(S.shadow[:S.total].float(), S.flat[:S.total].to(torch.bfloat16).float())


def test_vq_checkpoint_carries_the_tokenizer(tmp_path):
    """`dalle.encoder.*` travels with state_dict() / load_state_dict() and `dalle_model_path` loads a pickle."""
    arch = dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    common = dict(vis_backbone_size="tiny", size_frame=2, max_size_frame=6, arch_override=arch, bert_layers=1, mvm_target=["vq"], dvae_hid=64, dvae_vocab=512)
    m1, _ = _engine(dict(common, seed=1))
    sd = m1.state_dict()
    keys = [k for k in sd if k.startswith("dalle.encoder.blocks.")]
    assert len(keys) == len(m1.dalle.w) and "dalle.encoder.blocks.output.conv.w" in keys
    m2, _ = _engine(dict(common, seed=2))
    assert not torch.equal(m2.dalle.w["blocks.input.w"].cpu(), m1.dalle.w["blocks.input.w"].cpu())
    missing, unexpected = m2.load_state_dict({k: v.cpu() for k, v in sd.items()})
    assert not unexpected
    for k in m1.dalle.w:
        assert torch.equal(m1.dalle.w[k].cpu(), m2.dalle.w[k].cpu()), k
    pth = os.path.join(str(tmp_path), "encoder.pkl")
    torch.save({k[len("dalle.encoder."):]: v.cpu() for k, v in sd.items() if k.startswith("dalle.encoder.")}, pth)
    m3, _ = _engine(dict(common, seed=3, dalle_model_path=pth))
    assert torch.equal(m3.dalle.w["blocks.group_2.block_1.id_path.w"].cpu(), m1.dalle.w["blocks.group_2.block_1.id_path.w"].cpu())


def test_standalone_vidswin_and_encvideo_surface(tmp_path):
    """get_vidswin_model / load_checkpoint_3d (video_swin.py:573-659) and EncVideo / go_feat (model.py:8-78,174-178): an upstream-style
    `.pth` ({'state_dict': {'backbone.*'}}) loads with the prefix stripped; the standalone backbone's output equals the oracle's
    Swin forward; EncVideo returns (B, T*(1+hw), 768) features + a ones mask."""
    from oracle import violet_ref as R
    from pytorch_empirical_mvm_amd import config as CFG
    from pytorch_empirical_mvm_amd import visbackbone as VB
    arch = dict(embed_dim=32, depths=(1, 1, 1, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    cfg = R.make_cfg("tiny", T=2, img=96, arch=arch, bert_layers=1)
    sd = R.make_state_dict(cfg)
    swin = {k[len("enc_img.swin."):]: v for k, v in sd.items() if k.startswith("enc_img.swin.")}
    pth = os.path.join(str(tmp_path), "swin_upstream.pth")
    torch.save({"state_dict": {"backbone." + k: v for k, v in swin.items()}, "meta": {}}, pth)
    assert set(VB.load_checkpoint_3d(pth)) == set(swin)
    args = CFG.get_args(vis_backbone_size="tiny", size_img=96, arch_override=arch, vis_backbone_init="3d", vis_backbone_pretrained_weight=pth)
    m = VB.get_vidswin_model(args)
    assert m.norm.normalized_shape[0] == 8 * 32
    img, txt, mask = R.make_batch(cfg, 2)
    x = img.transpose(1, 2).contiguous()                          # (B,3,T,H,W) as the reference's backbone takes it
    y = m(x.cuda())
    assert tuple(y.shape) == (2, 256, 2, 3, 3)
    with torch.no_grad():
        yr = R.swin_forward(sd, cfg, x).permute(0, 4, 1, 2, 3)    # oracle: channels-last (B,T,h,w,8E) -> (B,8E,T,h,w)
    assert _cos(y.float().cpu(), yr) >= 0.999
    # vis_backbone_init = "2d": an image-Swin checkpoint ({'model': ...}) inflated along time on the way in (video_swin.py:484-535)
    sd2 = {k: v.clone() for k, v in swin.items()}
    sd2["patch_embed.proj.weight"] = swin["patch_embed.proj.weight"].sum(2)                          # (E,3,4,4): tiling / 2 gives both taps = sum / 2
    for k in [k for k in swin if k.endswith("relative_position_bias_table")]:
        sd2[k] = swin[k][:169].clone()                                                               # one temporal offset's 13 x 13 table
    sd2["layers.0.blocks.0.attn.relative_position_index"] = torch.zeros(49, 49, dtype=torch.long)
    pth2 = os.path.join(str(tmp_path), "swin_2d.pth")
    torch.save({"model": sd2}, pth2)
    m2 = VB.get_vidswin_model(CFG.get_args(vis_backbone_size="tiny", size_img=96, arch_override=arch, vis_backbone_init="2d", vis_backbone_pretrained_weight=pth2))
    got = m2.state_dict()
    w3 = got["patch_embed.proj.weight"].float().cpu()
    assert tuple(w3.shape) == tuple(swin["patch_embed.proj.weight"].shape) and torch.allclose(w3[:, :, 0], w3[:, :, 1])
    assert torch.allclose(w3[:, :, 0], sd2["patch_embed.proj.weight"] / 2, atol=1e-6)
    tb = got["layers.1.blocks.0.attn.relative_position_bias_table"].float().cpu()
    assert tuple(tb.shape) == (15 * 169, 2) and torch.allclose(tb.view(15, 169, 2), sd2["layers.1.blocks.0.attn.relative_position_bias_table"].expand(15, -1, -1), atol=1e-6)
    assert torch.allclose(got["layers.2.blocks.0.mlp.fc1.weight"].float().cpu(), swin["layers.2.blocks.0.mlp.fc1.weight"], atol=1e-6)
    model, _ = _engine(dict(vis_backbone_size="tiny", size_frame=2, max_size_frame=6, arch_override=arch, bert_layers=1, size_img=96))
    model.load_state_dict(sd)
    model.eval()
    feat, fmask = VB.EncVideo(model)(img.cuda())
    assert tuple(feat.shape) == (2, 2 * (1 + 9), 768) and tuple(fmask.shape) == (2, 20) and int(fmask.min()) == 1
    with torch.no_grad():
        fr = R.enc_video(sd, cfg, img)
    fr = fr[0] if isinstance(fr, (tuple, list)) else fr
    assert _cos(feat.float().cpu(), fr.reshape(feat.shape)) >= 0.999
    # frame-order embedding and visual-token mask (model.py:61-67,75): clip 0 in order, clip 1 with its two frames swapped (-> emb_odr twice)
    odr = [[0, 1], [1, 0]]
    vt = torch.ones(2, 2, 10, dtype=torch.long)
    vt[1, 0, 3:] = 0
    feat_o, mask_o = VB.EncVideo(model)(img.cuda(), odr=odr, vt_mask=vt.cuda())
    with torch.no_grad():
        fo, mo = R.enc_video(sd, cfg, img, odr=odr, vt_mask=vt)
    assert _cos(feat_o.float().cpu(), fo.reshape(feat_o.shape)) >= 0.999 and torch.equal(mask_o.cpu(), mo)
    assert torch.equal(feat_o[0], feat[0]) and not torch.equal(feat_o[1], feat[1])                  # only the re-ordered clip changes
    d_ref = (fo - fr.reshape(fo.shape))[1]
    assert _cos((feat_o.float() - feat.float())[1].cpu(), d_ref) >= 0.98                              # ... and by what the oracle says


def test_edge_cases_single_clip_empty_cover_padded_text_single_frame():
    """Edge cases of the step against the oracle on the same inputs: B = 1 (O = min(B, 4) = 1: the VTM logits are (1, 1), its loss 0 --
    main_pretrain.py:243-262), a batch whose patch cover is EMPTY (pixel loss = 0 / (0 + 1e-5), main_pretrain.py:429-430), text rows
    padded down to [CLS] [SEP], a single-frame clip (T = 1: D is padded 1 -> 8 inside the window partition) at 96^2 pixels, and a batch
    without any MLM target (the one place the two differ by design: reference NaN, here 0)."""
    from oracle import violet_ref as R
    arch = dict(embed_dim=32, depths=(1, 1, 2, 1), num_heads=(1, 2, 4, 8), window=(8, 7, 7))
    for (T, img_sz, B, empty_cov, short_txt) in [(2, 96, 1, False, False), (1, 96, 2, False, True), (2, 96, 2, True, False)]:
        model, args = _engine(dict(vis_backbone_size="tiny", size_frame=T, max_size_frame=6, arch_override=arch, bert_layers=1, size_img=img_sz))
        cfg = R.make_cfg("tiny", T=T, img=img_sz, arch=arch, bert_layers=1)
        sd = R.make_state_dict(cfg)
        model.load_state_dict(sd)
        img, txt, mask = R.make_batch(cfg, B)
        if short_txt:                                               # row 0: only [CLS] [SEP] are real tokens
            txt[0, 2:] = 0; mask[0, 2:] = 0; txt[0, 1] = 102
        mb = R.default_masking(cfg, img, txt, mask, seed=5)
        if empty_cov:
            mb["mvm_mask"].zero_(); mb["img"] = mb["unmask_img"].clone()
        neg = R.vtm_negatives_default(B)
        with torch.no_grad():
            ref = R.pretrain_losses(sd, cfg, mb, negatives=neg)
        cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8).cuda().contiguous()
        batch = dict(img=mb["unmask_img"].cuda(), cov=cov, txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda())
        eng = model.engine
        eng.store.grad.zero_()
        losses, _ = eng.forward_backward(batch, negatives=neg, train=False, want_outputs=True, backward=True)
        torch.cuda.synchronize()
        tag = (T, img_sz, B, empty_cov, short_txt)
        for k in ("mtm", "vtm", "mvm"):
            got, want = float(losses[k].item()), float(ref[k])
            assert np.isfinite(got), (tag, k, got)
            if k == "mtm" and int((mb["ans_mtm"] != -1).sum()) == 0:
                # a batch without a single MLM target: the reference's CrossEntropyLoss(ignore_index=-1) averages over an empty set
                # (NaN, and the step's update is lost); vmvm_cross_entropy divides by max(n_valid, 1): loss 0, zero gradient
                assert not np.isfinite(want) and got == 0.0, (tag, got, want)
                continue
            tol = 8e-2 if k == "vtm" else 3e-2 * abs(want) + 2e-3
            assert abs(got - want) <= tol, (tag, k, got, want)
        if B == 1:
            assert abs(float(losses["vtm"].item())) <= 1e-6
        if empty_cov:
            assert float(losses["mvm"].item()) == 0.0
        g = eng.store.grad[:eng.store.n_trainable]
        assert bool(torch.isfinite(g).all()), tag
        if empty_cov:                                               # no covered patch: the pixel head receives no gradient at all
            assert float(eng.store.g("decoder_pixel.0.weight").abs().max()) == 0.0


@pytest.mark.timeout(1500)
def test_full_size_c4_vq_target_step_vs_oracle():
    """BASELINE config 4 at full width on one GPU (Swin-B, 8 x 224^2 frames, vq target with the full dVAE tokenizer: n_hid 256, 8192
    codes, every pass a libvmvm kernel), B = 2, against the CPU oracle on the same weights and batch: token agreement of the native
    fp16 tokenizer with the oracle's fp32 encoder, then -- with the ORACLE's tokens as targets on both sides, so an arg-max near-tie
    does not decide the comparison -- the three losses, the vq-head gradients (cosine >= 0.997, norm +-2.5 %, the bars of the C2 test) and the global norm."""
    from oracle import violet_ref as R
    from pytorch_empirical_mvm_amd.agent import Agent_Pretrain
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    B, T = 2, 8
    cfg = R.make_cfg("base", T=T, mvm_target=["vq"], temp=1.0)
    model, args = _engine(dict(vis_backbone_size="base", size_frame=T, max_size_frame=8, mvm_target=["vq"], temp=1.0))
    assert model.dalle.native
    sd = R.make_state_dict(cfg)
    model.load_state_dict(sd)
    agent = Agent_Pretrain(args, model)
    img, txt, mask = R.make_batch(cfg, B)
    mb = R.default_masking(cfg, img, txt, mask, seed=3)
    neg = R.vtm_negatives_default(B)
    with torch.no_grad():
        tok_ref = R.vq_tokens(sd, cfg, mb["unmask_img"])                                  # (B*T, 28, 28) from the fp32 encoder
    tok = model.dalle.extract_vq_token(img.view(B * T, 3, 224, 224).cuda()).cpu()
    agree = float((tok == tok_ref.view_as(tok)).float().mean())
    assert agree >= 0.98, agree
    mb["vq_tokens"] = tok_ref
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ls = R.pretrain_losses(params, cfg, mb, negatives=neg)
    ls["total"].backward()
    cov = mb["mvm_mask"][:, :, 0, ::32, ::32].to(torch.uint8)
    vqi = agent.vq_index(cov)
    batch = dict(img=img.cuda(), cov=cov.cuda().contiguous(), txt=mb["txt"].cuda(), mask=mask.cuda(), ans_mtm=mb["ans_mtm"].cuda(),
                 vq_patch_rows=vqi["vq_patch_rows"].cuda(), vq_tok_index=vqi["vq_tok_index"].cuda(), vq_tokens=tok_ref.view_as(tok).cuda())
    eng = model.engine
    eng.store.grad.zero_()
    losses, _ = eng.forward_backward(batch, negatives=neg, train=False, backward=True, want_outputs=True)
    torch.cuda.synchronize()
    for k in ("mtm", "vtm", "mvm"):
        got, want = float(losses[k].item()), float(ls[k].detach())
        assert abs(got - want) <= 2e-2 * abs(want) + 2e-3, (k, got, want)
    for k in ("decoder_vq.0.weight", "decoder_vq.0.bias", "fc_mvm.1.weight", "fc_mvm.1.bias", "fc_mvm.3.weight", "fc_mvm.3.bias"):
        got = eng.store.g(k).detach().cpu().double().flatten()
        ref = params[k].grad.double().flatten()
        c, ratio = _cos(got, ref), float(got.norm() / ref.norm())
        print(f"\n[c4 vq head] {k}: cosine {c:.5f}, norm ratio {ratio:.4f}")
        assert c >= 0.997 and abs(ratio - 1.0) <= 0.025, (k, c, ratio)
    ref_norm = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params.values() if p.grad is not None)))
    S = eng.store
    got_norm = float(torch.sqrt((S.grad[:S.n_trainable].double() ** 2).sum()).item())
    assert abs(got_norm - ref_norm) <= 2e-2 * ref_norm, (got_norm, ref_norm)
